// Development micro-probe, round 3: do two column wavefronts on ONE SIMD run the REAL column step (col_dpp_step, pipe_common.hpp)
// faster together than one after the other?  W wavefronts per SIMD, each `steps` RK4 steps of 4 x 16 column lanes against a static
// coefficient ring in LDS, no synchronisation; reported: cycles from the first wavefront's start to the last one's end per step and
// per wavefront-of-the-SIMD (a lone wavefront's figure if the SIMD merely runs its wavefronts one after the other).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ilowthrustopt_amd/csrc -Ilowthrustopt_amd/csrc/hooks tools/micro/col_share_probe.hip -o build/col_share_probe
#include "pipe_common.hpp"
#include <cstdio>
#include <vector>
using namespace lto;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int ND, int NA, bool LM>
__global__ __launch_bounds__(1024) void k_cols(double* out, long long* cyc, int steps, int prio_mask) {
  constexpr int SD = CoefBySegment::stage_doubles<25>();
  __shared__ double s_coef[4 * 4 * SD];
  __shared__ long long s_t0[16], s_t1[16];
  for (int i = threadIdx.x; i < 4 * 4 * SD; i += blockDim.x) s_coef[i] = 1e-3 * ((i * 7) % 13 - 6);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = lane & 15, seg = (wave & 3) * 4 + (lane >> 4);
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  const ColStepConst k(1e-3, 2.0);
  const double* rec = s_coef + CoefBySegment::lane_base(col, seg);
  if (prio_mask & (1 << (wave >> 2))) __builtin_amdgcn_s_setprio(2);
  __syncthreads();
  const long long t0 = clock64();
  if (col < NA)
    for (int step = 0; step < steps; ++step) col_dpp_step<ND, SD, LM, NA>(rec + ((step & 3) * 4) * SD, k, step, y);
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int r = 0; r < ND; ++r) s += y[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) { s_t0[wave] = t0; s_t1[wave] = t1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long lo = s_t0[0], hi = s_t1[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = s_t0[w] < lo ? s_t0[w] : lo; hi = s_t1[w] > hi ? s_t1[w] : hi; }
    cyc[blockIdx.x] = hi - lo;
  }
}

template <int ND, int NA, bool LM> static int run(const char* name, double* d, long long* dc) {
  const int steps = 256;
  for (int prio = 0; prio < 2; ++prio) {
    printf("  %-40s %-26s", name, prio ? "first wave of a SIMD at prio 2" : "equal priority");
    for (int W = 1; W <= 4; ++W) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_cols<ND, NA, LM>), dim3(256), dim3(256 * W), 0, 0, d, dc, steps, prio);
      std::vector<long long> c(256);
      CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
      double m = 0; for (int i = 0; i < 256; ++i) m += c[i]; m /= 256;
      printf("  W=%d %7.1f", W, m / (double(steps) * W));
    }
    printf("   cycles / step / wavefront\n");
  }
  return 0;
}

int main() {
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 1024 * 256)); CK(hipMalloc(&dc, 8 * 256));
  run<14, 13, false>("14-dim columns (13 lanes of 16)", d, dc);
  run<12, 12, true>("12-dim columns (12 lanes of 16)", d, dc);
  return 0;
}
