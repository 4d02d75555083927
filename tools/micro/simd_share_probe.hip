// Development micro-probe, round 3: aggregate fp64 issue rate of ONE SIMD as a function of how many wavefronts it hosts and of the
// instruction kind the column waves of the pipeline kernel are made of (v_fmac_f64_dpp row_newbcast against plain v_fma_f64).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/simd_share_probe.hip -o build/simd_share_probe && build/simd_share_probe
// Workgroup = 4 W wavefronts (wave i sits on SIMD i & 3: sync_probe.hip), every wavefront issues the same stream of N instructions
// with CH independent accumulator chains; reported: cycles from the first wavefront's start to the last one's end, per
// instruction and SIMD (W N instructions each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// KIND 0: v_fma_f64; 1: v_fmac_f64_dpp row_newbcast:5; 2: alternating 2 dpp : 1 plain (the column step's mix);
// 3: v_fma_f64 with one ds_read_b128 per 24 (the column step's LDS share); 4: v_fma_f64 with a scalar multiplicand
template <int KIND, int CH> __device__ __forceinline__ void body(double (&a)[CH], const double c, const double d, const int addr, double (&l)[2]) {
#pragma unroll
  for (int r = 0; r < 96 / CH; ++r) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const bool dpp = (KIND == 1) || (KIND == 2 && ((r * CH + i) % 3) != 2);
      if (dpp) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c), "v"(d));
      else if (KIND == 4) asm volatile("v_fma_f64 %0, %0, s[40:41], %1" : "+v"(a[i]) : "v"(d));
      else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    }
    if (KIND == 3 && (r * CH) % 24 == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(addr) : "memory");
  }
}

template <int KIND, int CH> __global__ __launch_bounds__(1024) void k_issue(double* out, long long* cyc, int iters) {
  __shared__ double lds[2048];
  __shared__ long long s_t0[16], s_t1[16];
  double a[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
  const double c = 1.0000001, d = 1e-9;
  double l[2] = {0.0, 0.0};
  asm volatile("s_mov_b32 s40, 0\n s_mov_b32 s41, 0x3ff00000" ::: "s40", "s41");
  lds[threadIdx.x & 2047] = 0.0;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) body<KIND, CH>(a, c, d, (threadIdx.x & 63) * 16, l);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  double s = l[0] + l[1];
#pragma unroll
  for (int i = 0; i < CH; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_t0[wave] = t0; s_t1[wave] = t1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long lo = s_t0[0], hi = s_t1[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = s_t0[w] < lo ? s_t0[w] : lo; hi = s_t1[w] > hi ? s_t1[w] : hi; }
    cyc[blockIdx.x] = hi - lo;
  }
}

template <int KIND, int CH> static int run(const char* name, double* d, long long* dc) {
  const int iters = 200;
  printf("  %-46s", name);
  for (int W = 1; W <= 4; ++W) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_issue<KIND, CH>), dim3(256), dim3(256 * W), 0, 0, d, dc, iters);
    std::vector<long long> c(256);
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    double m = 0; for (int i = 0; i < 256; ++i) m += c[i]; m /= 256;
    printf("  W=%d %.2f", W, m / (double(iters) * 96.0 * W));
  }
  printf("   cycles / instruction / SIMD\n");
  return 0;
}

int main() {
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 1024 * 256)); CK(hipMalloc(&dc, 8 * 256));
  run<0, 4>("v_fma_f64, 4 chains", d, dc);
  run<0, 8>("v_fma_f64, 8 chains", d, dc);
  run<0, 16>("v_fma_f64, 16 chains", d, dc);
  run<4, 8>("v_fma_f64 scalar multiplicand, 8 chains", d, dc);
  run<1, 4>("v_fmac_f64_dpp row_newbcast, 4 chains", d, dc);
  run<1, 8>("v_fmac_f64_dpp row_newbcast, 8 chains", d, dc);
  run<1, 16>("v_fmac_f64_dpp row_newbcast, 16 chains", d, dc);
  run<2, 8>("2 dpp : 1 plain, 8 chains", d, dc);
  run<3, 8>("v_fma_f64 + ds_read_b128 / 24, 8 chains", d, dc);
  return 0;
}
