// Development micro-probe (not part of the library), round 2:
//  (1) accuracy of the v_rsq_f64 / v_rcp_f64 seeds and of second- / third-order refinements,
//  (2) SIMD placement of the waves of a 512-thread workgroup when wave 6 exits at once, and that the remaining seven
//      waves still meet at s_barrier,
//  (3) issue cost of LDS instructions (ds_write_b128 / ds_read_b128 / ds_write2_b64) against v_fma_f64 in the same wave,
//  (4) latency of a flag hand-over through LDS between two waves of a workgroup (producer on one SIMD, consumer polling).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/sync_probe.hip -o build/sync_probe && build/sync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// ------------------------------------------------------------------------------------------------ (1) seed accuracy
__global__ void k_seed(const double* x, double* rsq, double* rcp, double* rsq2, double* rsq3, double* rcp2, double* rcp3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  const double y = __builtin_amdgcn_rsq(v);
  rsq[i] = y;
  {   // second order (Newton): y (1 + e/2)
    const double e = __builtin_fma(-v * y, y, 1.0);
    rsq2[i] = __builtin_fma(y * e, 0.5, y);
    const double t = __builtin_fma(0.375, e, 0.5);
    rsq3[i] = __builtin_fma(y * e, t, y);
  }
  const double r = __builtin_amdgcn_rcp(v);
  rcp[i] = r;
  {
    const double e = __builtin_fma(-v, r, 1.0);
    rcp2[i] = __builtin_fma(r, e, r);
    rcp3[i] = __builtin_fma(r, __builtin_fma(e, e, e), r);
  }
}

// ------------------------------------------------------------------------------------------------ (2) placement
__global__ void k_place8(int* simd, int* met) {
  const int wave = threadIdx.x >> 6;
  if (wave == 6) return;                       // leaves before any barrier
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    simd[blockIdx.x * 8 + wave] = (hw >> 4) & 3;
  }
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  double a = threadIdx.x;
  for (int i = 0; i < 2000 * (wave + 1); ++i) a = a * 1.0000001 + 1e-9;   // waves arrive at different times
  if (a == 12345.0) simd[0] = -1;
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, 1);
  __syncthreads();
  if (threadIdx.x == 0) met[blockIdx.x] = cnt;  // 7 if the barrier waited for every remaining wave
}

// ------------------------------------------------------------------------------------------------ (3) LDS issue cost
// MODE 0: 64 v_fma per iteration.  MODE 1: + 4 ds_write_b128.  MODE 2: + 4 ds_read_b128 (results consumed at the end of
// the iteration).  MODE 3: + 4 ds_write2_b64.  MODE 4: + 4 ds_write_b128 with only 16 lanes active.
template <int MODE> __global__ void k_lds(double* out, int iters) {
  __shared__ double buf[64 * 2 * 8];
  const int t = threadIdx.x;
  double a0 = 1.0 + 0.001 * t, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
  double2* p = reinterpret_cast<double2*>(buf) + t;
  double2 r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
  buf[t] = t; buf[t + 64] = t;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a0 = __builtin_fma(a0, 1.0000001, 1e-9); a1 = __builtin_fma(a1, 1.0000001, 1e-9); a2 = __builtin_fma(a2, 1.0000001, 1e-9);
      a3 = __builtin_fma(a3, 1.0000001, 1e-9); a4 = __builtin_fma(a4, 1.0000001, 1e-9); a5 = __builtin_fma(a5, 1.0000001, 1e-9);
      a6 = __builtin_fma(a6, 1.0000001, 1e-9); a7 = __builtin_fma(a7, 1.0000001, 1e-9);
      asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      asm volatile("" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    if (MODE == 1 || (MODE == 4 && t < 16)) {
      p[0] = double2{a0, a1}; p[64] = double2{a2, a3}; p[128] = double2{a4, a5}; p[192] = double2{a6, a7};
    }
    if (MODE == 2) {
      r0 = p[0]; r1 = p[64]; r2 = p[128]; r3 = p[192];
      asm volatile("" : "+v"(r0.x), "+v"(r1.x), "+v"(r2.x), "+v"(r3.x));
    }
    if (MODE == 3) {
      buf[t] = a0; buf[t + 64] = a1; buf[t + 128] = a2; buf[t + 192] = a3;
      buf[t + 256] = a4; buf[t + 320] = a5; buf[t + 384] = a6; buf[t + 448] = a7;
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0.x + r1.x + r2.x + r3.x + buf[(t * 7) & 511];
}

// ------------------------------------------------------------------------------------------------ (4) flag hand-over
// Two waves of a 128-thread workgroup play ping-pong through two LDS words: each round trip is two hand-overs
// (write flag -> the other wave sees it).  Bounded polling.  Between hand-overs each wave issues `work` FMAs.
__global__ void k_pingpong(long long* cycles, int rounds, int* fail) {
  __shared__ volatile int flag[2];
  const int wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) { flag[0] = 0; flag[1] = 0; }
  __syncthreads();
  const long long t0 = clock64();
  int bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (wave == 0) {
      if ((threadIdx.x & 63) == 0) flag[0] = r;
      int spins = 0;
      while (flag[1] < r && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
      if (spins >= (1 << 20)) { bad = 1; break; }
    } else {
      int spins = 0;
      while (flag[0] < r && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
      if (spins >= (1 << 20)) { bad = 1; break; }
      if ((threadIdx.x & 63) == 0) flag[1] = r;
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; if (bad) *fail = 1; }
  if (threadIdx.x == 64 && bad) *fail = 1;
}

// barrier round trip for comparison: `rounds` __syncthreads of an 8-wave workgroup
__global__ void k_barrier(long long* cycles, int rounds) {
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE> static float run_lds(double* d, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(1024), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  // (1)
  const int n = 1 << 20;
  std::vector<double> x(n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (s >> 11) * (1.0 / 9007199254740992.0);
    const int ex = (int)((s >> 3) % 41) - 20;                 // 2^-20 .. 2^20
    x[i] = std::ldexp(1.0 + u, ex);
  }
  double *dx, *d[6];
  CK(hipMalloc(&dx, n * 8));
  for (int k = 0; k < 6; ++k) CK(hipMalloc(&d[k], n * 8));
  CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_seed, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], d[5], n);
  std::vector<double> h(n);
  const char* names[6] = {"v_rsq_f64 seed", "v_rcp_f64 seed", "rsq + 2nd order", "rsq + 3rd order", "rcp + 2nd order", "rcp + 3rd order"};
  for (int k = 0; k < 6; ++k) {
    CK(hipMemcpy(h.data(), d[k], n * 8, hipMemcpyDeviceToHost));
    long double worst = 0;
    for (int i = 0; i < n; ++i) {
      const long double ref = (k == 0 || k == 2 || k == 3) ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
      const long double e = fabsl(((long double)h[i] - ref) / ref);
      if (e > worst) worst = e;
    }
    printf("%-18s max relative error %.3Le = 2^%.1Lf  (%.2Lf ulp of binary64)\n", names[k], worst, log2l(worst), worst / 1.1102230246251565e-16L);
  }

  // (2)
  const int nb = 256;
  int *dsimd, *dmet;
  CK(hipMalloc(&dsimd, nb * 8 * 4)); CK(hipMalloc(&dmet, nb * 4));
  CK(hipMemset(dsimd, 0xff, nb * 8 * 4));
  hipLaunchKernelGGL(k_place8, dim3(nb), dim3(512), 0, 0, dsimd, dmet);
  std::vector<int> simd(nb * 8), met(nb);
  CK(hipMemcpy(simd.data(), dsimd, nb * 8 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(met.data(), dmet, nb * 4, hipMemcpyDeviceToHost));
  {
    int okb = 0;
    std::vector<std::pair<std::string, int>> patt;
    for (int b = 0; b < nb; ++b) {
      if (met[b] == 7) ++okb;
      std::string key;
      for (int w = 0; w < 8; ++w) key += (w == 6) ? 'x' : char('0' + simd[b * 8 + w]);
      bool found = false;
      for (auto& p : patt) if (p.first == key) { ++p.second; found = true; }
      if (!found) patt.push_back({key, 1});
    }
    printf("512-thread workgroup, wave 6 exits at once: barrier met by the 7 others in %d / %d workgroups; SIMD of waves 0..7:\n", okb, nb);
    for (auto& p : patt) printf("  %s: %d\n", p.first.c_str(), p.second);
  }

  // (3)
  double* dout; CK(hipMalloc(&dout, 8 * 64 * 1024));
  const int iters = 2000;
  run_lds<0>(dout, iters);
  const float t0 = run_lds<0>(dout, iters), t1 = run_lds<1>(dout, iters), t2 = run_lds<2>(dout, iters), t3 = run_lds<3>(dout, iters), t4 = run_lds<4>(dout, iters);
  const float fma = t0 / (iters * 64.0f);
  printf("64 v_fma_f64 per iteration: %.2f ns per FMA\n", fma * 1e6);
  printf("+ 4 ds_write_b128 (64 lanes): %.2f FMA slots each\n", (t1 - t0) / (iters * 4.0f) / fma);
  printf("+ 4 ds_read_b128  (64 lanes): %.2f FMA slots each\n", (t2 - t0) / (iters * 4.0f) / fma);
  printf("+ 8 ds_write_b64 / 4 write2 (64 lanes): %.2f FMA slots per pair\n", (t3 - t0) / (iters * 4.0f) / fma);
  printf("+ 4 ds_write_b128 (16 lanes): %.2f FMA slots each\n", (t4 - t0) / (iters * 4.0f) / fma);

  // (4)
  long long* dcy; int* dfail;
  CK(hipMalloc(&dcy, 256 * 8)); CK(hipMalloc(&dfail, 4)); CK(hipMemset(dfail, 0, 4));
  const int rounds = 2000;
  hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(128), 0, 0, dcy, rounds, dfail);
  std::vector<long long> cy(256);
  int fail = 0;
  CK(hipMemcpy(cy.data(), dcy, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost));
  double mean = 0; for (auto c : cy) mean += c; mean /= 256;
  printf("LDS flag ping-pong (2 waves, s_sleep 1 polling): %.0f clock64 ticks per round trip = %.0f per hand-over%s\n", mean / rounds, mean / rounds / 2, fail ? "  [POLL LIMIT HIT]" : "");
  hipLaunchKernelGGL(k_barrier, dim3(256), dim3(512), 0, 0, dcy, rounds);
  CK(hipMemcpy(cy.data(), dcy, 256 * 8, hipMemcpyDeviceToHost));
  mean = 0; for (auto c : cy) mean += c; mean /= 256;
  printf("__syncthreads of an 8-wave workgroup: %.0f clock64 ticks each\n", mean / rounds);
  // clock64 tick rate: time a known-length kernel
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_barrier, dim3(256), dim3(512), 0, 0, dcy, 200000);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(cy.data(), dcy, 256 * 8, hipMemcpyDeviceToHost));
  printf("clock64: %.1f ticks per microsecond\n", cy[0] / (ms * 1e3));
  return 0;
}
