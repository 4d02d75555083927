// Development micro-probe: issue cost of the fp64 instructions the base wave's critical stream uses, relative to v_fma_f64
// (one wave per SIMD, 8 independent chains).   hipcc -O3 --offload-arch=gfx950 tools/micro/rate_probe.hip -o build/rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP8_(ASM) \
  asm volatile(ASM : "+v"(a0)); asm volatile(ASM : "+v"(a1)); asm volatile(ASM : "+v"(a2)); asm volatile(ASM : "+v"(a3)); \
  asm volatile(ASM : "+v"(a4)); asm volatile(ASM : "+v"(a5)); asm volatile(ASM : "+v"(a6)); asm volatile(ASM : "+v"(a7));
// 64 instructions per loop iteration, so that the loop branch (an instruction-fetch restart) does not dominate
#define OP8(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM) OP8_(ASM)

template <int OP> __global__ void k_rate(double* out, int iters) {
  double a0 = 1.0 + 0.001 * threadIdx.x, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { OP8("v_fma_f64 %0, %0, 1.0, 0.5") }
    if (OP == 1) { OP8("v_mul_f64 %0, %0, 1.0") }
    if (OP == 2) { OP8("v_add_f64 %0, %0, 1.0") }
    if (OP == 3) { OP8("v_rsq_f64 %0, %0") }
    if (OP == 4) { OP8("v_rcp_f64 %0, %0") }
    if (OP == 5) { OP8("v_rndne_f64 %0, %0") }
    if (OP == 6) { OP8("v_ldexp_f64 %0, %0, 1") }
    if (OP == 7) { OP8("v_max_f64 %0, %0, 1.0") }
    if (OP == 8) { OP8("v_mov_b64 %0, %0") }
    if (OP == 9) { OP8("v_sqrt_f64 %0, %0") }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_cvt(double* out, int iters) {
  double a0 = 1.0 + 0.001 * threadIdx.x, a1 = 1.1, a2 = 1.2, a3 = 1.3;
  int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b1) : "v"(a1));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b3) : "v"(a3));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b1) : "v"(a1));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b3) : "v"(a3));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b1) : "v"(a1));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b3) : "v"(a3));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b1) : "v"(a1));
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b3) : "v"(a3));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = b0 + b1 + b2 + b3;
}

template <int OP> static float run(double* d, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(1024), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  double* d; (void)hipMalloc(&d, 8 * 64 * 1024);
  const int iters = 5000;
  // warm the clocks
  for (int i = 0; i < 5; ++i) run<0>(d, iters);
  const float fma = run<0>(d, iters);
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64", "v_rndne_f64", "v_ldexp_f64", "v_max_f64", "v_mov_b64", "v_sqrt_f64"};
  float t[10];
  t[0] = fma; t[1] = run<1>(d, iters); t[2] = run<2>(d, iters); t[3] = run<3>(d, iters); t[4] = run<4>(d, iters);
  t[5] = run<5>(d, iters); t[6] = run<6>(d, iters); t[7] = run<7>(d, iters); t[8] = run<8>(d, iters); t[9] = run<9>(d, iters);
  for (int i = 0; i < 10; ++i) printf("%-12s %.2f ns per instruction  = %.2f x v_fma_f64\n", names[i], t[i] * 1e6 / (iters * 64.0), t[i] / fma);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k_cvt, dim3(1024), dim3(64), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
  }
  printf("%-12s %.2f ns per instruction  = %.2f x v_fma_f64\n", "v_cvt_i32_f64", best * 1e6 / (iters * 16.0), best * 4.0 / fma);
  return 0;
}
