// Development micro-probe, round 2: issue cadence of fp64 VALU instructions in cycles (s_memtime inside the kernel, so the
// figure is independent of the shader clock): chains = independent dependent-chains per wave (1: every instruction waits for
// the previous one), waves per SIMD 1 or 2, plain FMA / DPP FMA / a mix with quarter-rate v_rsq_f64.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_probe.hip -o build/issue_probe && build/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int CH, int KIND> __global__ void k_issue(double* out, long long* cyc, int iters) {
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
  const double c = 1.0000001, d = 1e-9;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / CH; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c), "v"(d));
        if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 3) asm volatile("v_mov_b64 %0, %0" : "+v"(a[i]));
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CH, int KIND> static double run(double* d, long long* dc, int waves_per_simd) {
  const int iters = 400, wpb = 4 * waves_per_simd;     // one workgroup per CU: 4 or 8 waves
  hipLaunchKernelGGL((k_issue<CH, KIND>), dim3(256), dim3(64 * wpb), 0, 0, d, dc, iters);
  hipLaunchKernelGGL((k_issue<CH, KIND>), dim3(256), dim3(64 * wpb), 0, 0, d, dc, iters);
  std::vector<long long> c(256 * wpb);
  if (hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  double m = 0; for (auto v : c) m += v; m /= c.size();
  return m / (iters * 64.0);   // cycles per instruction of ONE wave
}

int main() {
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 64 * 8 * 256)); CK(hipMalloc(&dc, 8 * 8 * 256));
  const char* kinds[4] = {"v_fma_f64", "v_fmac_f64_dpp", "v_mul_f64", "v_mov_b64"};
  for (int w = 1; w <= 2; ++w) {
    printf("%d wave(s) per SIMD, all CUs busy: cycles between instructions of one wave (SIMD cadence = that / waves)\n", w);
    printf("  %-16s chains=1 %.2f  2 %.2f  4 %.2f  8 %.2f\n", kinds[0], run<1, 0>(d, dc, w), run<2, 0>(d, dc, w), run<4, 0>(d, dc, w), run<8, 0>(d, dc, w));
    printf("  %-16s chains=1 %.2f  2 %.2f  4 %.2f  8 %.2f\n", kinds[1], run<1, 1>(d, dc, w), run<2, 1>(d, dc, w), run<4, 1>(d, dc, w), run<8, 1>(d, dc, w));
    printf("  %-16s chains=1 %.2f  2 %.2f  4 %.2f  8 %.2f\n", kinds[2], run<1, 2>(d, dc, w), run<2, 2>(d, dc, w), run<4, 2>(d, dc, w), run<8, 2>(d, dc, w));
    printf("  %-16s chains=1 %.2f  2 %.2f  4 %.2f  8 %.2f\n", kinds[3], run<1, 3>(d, dc, w), run<2, 3>(d, dc, w), run<4, 3>(d, dc, w), run<8, 3>(d, dc, w));
  }
  // sustained FP64 FMA rate of the whole device (wall clock): 8 independent chains per wave, 2 and 4 waves per SIMD
  for (int w = 2; w <= 4; w += 2) {
    const int iters = 20000, wpb = 4 * w;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_issue<8, 0>), dim3(256), dim3(64 * wpb), 0, 0, d, dc, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_issue<8, 0>), dim3(256), dim3(64 * wpb), 0, 0, d, dc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 2.0 * 64 * 64.0 * iters * wpb * 256;
    std::vector<long long> c(256 * wpb);
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : c) m += v; m /= c.size();
    printf("sustained v_fma_f64, %d waves per SIMD, %.1f ms: %.1f TFLOP/s of 78.6 nominal; s_memtime ticks per wave %.0f = %.3f GHz tick rate; %.2f ticks per SIMD instruction\n",
           w, ms, flops / (ms * 1e-3) / 1e12, m, m / (ms * 1e6), m / (iters * 64.0) / w);
  }
  return 0;
}
