// Development micro-probe (not part of the library): (1) semantics of v_fmac_f64_dpp row_newbcast on gfx950,
// (2) its issue rate against a plain v_fmac_f64, (3) which SIMD each wave of a 6-wave workgroup lands on.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dpp_probe.hip -o build/dpp_probe && build/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

template <int N> __device__ __forceinline__ void fmac_b(double& acc, double c, double x) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(N));
}

__global__ void k_sem(double* out, const double* c, const double* x) {
  const int t = threadIdx.x;
  double acc = 1.0;
  double cv = c[t], xv = x[t];
  fmac_b<3>(acc, cv, xv);      // acc = 1 + c[row*16+3] * x[t]
  fmac_b<15>(acc, cv, xv);     //     + c[row*16+15] * x[t]
  out[t] = acc;
}

template <bool DPP> __global__ void k_rate(double* out, const double* c, int iters) {
  const int t = threadIdx.x;
  double cv = c[t];
  double a0 = 0.1 * t, a1 = 0.2, a2 = 0.3, a3 = 0.4, a4 = 0.5, a5 = 0.6, a6 = 0.7, a7 = 0.8;
  const double x = 1e-9;
  for (int i = 0; i < iters; ++i) {
    if (DPP) {
      fmac_b<1>(a0, cv, x); fmac_b<2>(a1, cv, x); fmac_b<3>(a2, cv, x); fmac_b<4>(a3, cv, x);
      fmac_b<5>(a4, cv, x); fmac_b<6>(a5, cv, x); fmac_b<7>(a6, cv, x); fmac_b<8>(a7, cv, x);
    } else {
      asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a0) : "v"(cv), "v"(x)); asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a1) : "v"(cv), "v"(x));
      asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a2) : "v"(cv), "v"(x)); asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a3) : "v"(cv), "v"(x));
      asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a4) : "v"(cv), "v"(x)); asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a5) : "v"(cv), "v"(x));
      asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a6) : "v"(cv), "v"(x)); asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a7) : "v"(cv), "v"(x));
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// does a VALU instruction get cheaper when only the first 16 lanes are active?  (it does not on gfx950: see main)
__global__ void k_exec(double* out, const double* c, int iters, int active) {
  const int t = threadIdx.x;
  double a0 = 0.1 * t, a1 = 0.2, a2 = 0.3, a3 = 0.4, a4 = 0.5, a5 = 0.6, a6 = 0.7, a7 = 0.8;
  const double cv = c[t], x = 1e-9;
  if (t < active) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_fma(cv, x, a0); a1 = __builtin_fma(cv, x, a1); a2 = __builtin_fma(cv, x, a2); a3 = __builtin_fma(cv, x, a3);
      a4 = __builtin_fma(cv, x, a4); a5 = __builtin_fma(cv, x, a5); a6 = __builtin_fma(cv, x, a6); a7 = __builtin_fma(cv, x, a7);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      asm volatile("" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_place(int* simd, int* cu) {
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    simd[blockIdx.x * (blockDim.x >> 6) + wave] = (hw >> 4) & 3;
    cu[blockIdx.x * (blockDim.x >> 6) + wave] = (hw >> 8) & 15;
  }
  // keep the waves resident for a while so that workgroups do not reuse slots
  double a = threadIdx.x;
  for (int i = 0; i < 20000; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.0) simd[0] = -1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
  double *dc, *dx, *dout;
  std::vector<double> c(64), x(64), out(64);
  for (int i = 0; i < 64; ++i) { c[i] = 100.0 + i; x[i] = 0.5 + 0.25 * i; }
  CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dx, 512)); CK(hipMalloc(&dout, 8 * 64 * 1024));
  CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, x.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dout, dc, dx);
  CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    const int row = t / 16;
    const double ref = 1.0 + c[row * 16 + 3] * x[t] + c[row * 16 + 15] * x[t];
    if (std::fabs(out[t] - ref) > 1e-12 * std::fabs(ref)) { if (bad < 5) printf("lane %d: got %.17g want %.17g\n", t, out[t], ref); ++bad; }
  }
  printf("semantics: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok: D += bcast(S0, lane n of the row) * S1", bad);

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int dpp = 0; dpp < 2; ++dpp) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (dpp) hipLaunchKernelGGL(k_rate<true>, dim3(1024), dim3(64), 0, 0, dout, dc, iters);
      else hipLaunchKernelGGL(k_rate<false>, dim3(1024), dim3(64), 0, 0, dout, dc, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%s: %.3f ms for %d x 8 fmac per wave, one wave per SIMD -> %.2f ns per instruction\n", dpp ? "v_fmac_f64_dpp" : "v_fmac_f64    ", ms, iters, ms * 1e6 / (iters * 8.0));
    }
  }

  for (int active = 64; active >= 16; active -= 16) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_exec, dim3(1024), dim3(64), 0, 0, dout, dc, iters, active);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("v_fma_f64 with %2d active lanes: %.2f ns per instruction\n", active, ms * 1e6 / (iters * 8.0));
    }
  }

  const int nb = 256, wpb = 6;
  int *dsimd, *dcu;
  CK(hipMalloc(&dsimd, nb * wpb * 4)); CK(hipMalloc(&dcu, nb * wpb * 4));
  hipLaunchKernelGGL(k_place, dim3(nb), dim3(64 * wpb), 0, 0, dsimd, dcu);
  std::vector<int> simd(nb * wpb), cu(nb * wpb);
  CK(hipMemcpy(simd.data(), dsimd, nb * wpb * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cu.data(), dcu, nb * wpb * 4, hipMemcpyDeviceToHost));
  int patt[4096] = {0};
  for (int b = 0; b < nb; ++b) {
    int key = 0;
    for (int w = 0; w < wpb; ++w) key = key * 4 + simd[b * wpb + w];
    patt[key]++;
  }
  printf("SIMD of waves 0..5 of a 384-thread workgroup (pattern: count over %d workgroups):\n", nb);
  for (int k = 0; k < 4096; ++k) if (patt[k]) {
    printf("  ");
    for (int w = wpb - 1; w >= 0; --w) printf("%d", (k >> (2 * w)) & 3);
    printf(": %d\n", patt[k]);
  }
  return 0;
}
