// kernels_indirect_pipe.hip -- three-role software pipeline for the fixed-step RK4 STM sweep (BASELINE configs[1]).
//
// At 4 096 segments the chip offers exactly 16 lanes per segment (1 024 SIMDs x 64 lanes) and the sweep lasts as long
// as ONE lane's instruction stream: the per-lane kernel (indirect_kernel.hpp) spends ~60 % of that stream on work
// every column lane of a segment repeats (base trajectory, gravity, control law), and the cooperative kernel
// (kernels_indirect_coop.hip) pays one workgroup barrier per RK stage with the base role's full RHS + coefficient
// build on the critical path.  Here a workgroup (4 wavefronts, one per SIMD) owns 16 segments and runs three roles
// that are skewed by one RK4 STEP each, so they execute concurrently and meet at ONE barrier per step:
//
//   wave 0  base      integrates the ND-dim base state (rhs*_base: the RHS alone -- the shortest instruction stream)
//                     and publishes, per stage, the part of the stage argument the coefficients depend on
//   wave 1  coef      one step behind: lane = (segment, RK stage); builds G, H, U (+ mass couplings) at those
//                     arguments and publishes them
//   waves 2,3 columns two steps behind: lane = (segment, PAIR of STM columns); c' = F(t) c with the coefficients
//                     read from LDS -- no gravity, no control law, no base state in these lanes
//
// Both hand-overs are double-buffered per step in LDS (46 KB for ND = 14); a sweep of n steps takes n + 2 phases.
// Two columns per lane (7 pairs x 16 segments = 112 lanes for ND = 14) is what lets the column role fit the two
// remaining SIMDs, so that every SIMD carries one wave of ~170-190 fp64 instructions per stage instead of the
// per-lane kernel's 281.  Every wavefront executes exactly steps + 2 barriers; nothing spins.
//
// Replaces the serial loop of jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) for the fixed-step setting.
#include "kernels.hpp"

namespace lto {

constexpr int PIPE_SEG = 16;   // segments per workgroup

// The coefficients of an RK stage are functions of the stage argument's position, lambda_v (and, for ND = 14, mass and
// lambda_m) only: that is all the base wave publishes (6 / 8 doubles per stage -- LDS stores of a 64-lane wave cost
// issue time on the critical path), in the order of PipeArg<ND>::idx.
template <int ND> struct PipeArg {
  static constexpr int N = 6;
  static constexpr int idx[6] = {0, 1, 2, 9, 10, 11};
  using Coef = VarCoef12;
};
template <> struct PipeArg<14> {
  static constexpr int N = 8;
  static constexpr int idx[8] = {0, 1, 2, 6, 10, 11, 12, 13};
  using Coef = VarCoef14;
};

template <int ND, int PM>
__device__ __forceinline__ void pipe_base_rhs(const double (&y)[ND], const TrajParams& tp, double (&k)[ND]) {
  if constexpr (ND == 12) rhs12_base<PM>(y, tp, k);
  else rhs14_base<PM>(y, tp, k);
}
// G, H, U (+ mass couplings) at the stage argument `arg` (components outside PipeArg<ND>::idx are unused: the slopes
// this call also produces are dead code)
template <int ND, int PM>
__device__ __forceinline__ void pipe_coef(const double (&arg)[ND], const TrajParams& tp, typename PipeArg<ND>::Coef& vc) {
  double dead[ND];
  if constexpr (ND == 12) rhs12<PM, true>(arg, tp, dead, vc);
  else rhs14<PM, true>(arg, tp, dead, vc);
}
template <int ND>
__device__ __forceinline__ void pipe_col(const typename PipeArg<ND>::Coef& vc, const double w2, const double (&c)[ND],
                                         double (&dc)[ND]) {
  if constexpr (ND == 12) var_col12(vc, w2, c, dc);
  else var_col14(vc, w2, c, dc);
}

template <int ND, int PM>
__global__ __launch_bounds__(256) void k_indirect_pipe(const IndirectArgs a) {
  using Coef = typename PipeArg<ND>::Coef;
  constexpr int NI = PipeArg<ND>::N;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  constexpr int NPAIR = ND / 2;

  // base -> coef: slab = parity * 4 + stage, [value][segment]; slabs 8..10 take the stores of the base wave's three
  // spare lane groups, so that publishing needs no EXEC branch (a branch per stage would cut the base role's one
  // basic block per step into five and keep the scheduler from overlapping the stages' rsqrt / exp chains)
  __shared__ double s_int[8 + 3][NI][PIPE_SEG];
  __shared__ double s_coef[2][4][NC][PIPE_SEG];   // coef -> columns: [step parity][stage][value][segment]

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int seg = lane & (PIPE_SEG - 1), slot = lane >> 4;
  const int s_raw = blockIdx.x * PIPE_SEG + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  const int s = a.order ? a.order[s_lin] : s_lin;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  // mixed-class batch: segments of another control-law class belong to that class's launch; here they run through the
  // barriers without storing
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform

  const int steps = a.steps;
  const double h = span / (double)steps;
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);

  const bool in_range = (s_raw < a.S) && mine;
  const int nphase = steps + 2;   // every role loop below executes exactly nphase barriers
#ifdef PIPE_PROBE
  const int probe = a.max_steps;  // development build only: bit 0 / 1 / 2 switches the base / coef / column work off
#else
  constexpr int probe = 0;
#endif

  if (wave == 0) {
    // -------------------------------------------------------------------- base: step p in phase p
    double y[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
    for (int p = 0; p < nphase; ++p) {
      if (p < steps && !(probe & 1)) {
        const int buf = p & 1;
        double k[ND], yt[ND], acc[ND];
        auto publish = [&](int stage, const double (&arg)[ND]) {
          double* dst = &s_int[slot == 0 ? buf * 4 + stage : 7 + slot][0][seg];
#pragma unroll
          for (int e = 0; e < NI; ++e) dst[e * PIPE_SEG] = arg[PipeArg<ND>::idx[e]];
        };
        publish(0, y);
        pipe_base_rhs<ND, PM>(y, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(1, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(2, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
        publish(3, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
      }
      __syncthreads();
    }
    if (in_range && slot == 0) {
      if (a.defect) {
#pragma unroll
        for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s] = y[c] - a.X[c * a.ldx + node + 1];
      }
      if (a.errors) a.errors[s] = 0.0;
      if (a.nacc) a.nacc[s] = steps;
      if (a.nrej) a.nrej[s] = 0;
    }
  } else if (wave == 1) {
    // -------------------------------------------------------------------- coefficients of step p - 1 in phase p
    // lane = (segment, RK stage): the four stages of a step are built side by side
    for (int p = 0; p < nphase; ++p) {
      if (p >= 1 && p <= steps && !(probe & 2)) {
        const int buf = (p - 1) & 1;
        double arg[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) arg[c] = 0.0;
        const double* src = &s_int[buf * 4 + slot][0][seg];
#pragma unroll
        for (int e = 0; e < NI; ++e) arg[PipeArg<ND>::idx[e]] = src[e * PIPE_SEG];
        Coef vc;
        pipe_coef<ND, PM>(arg, tp, vc);
        const double* o = reinterpret_cast<const double*>(&vc);
        double* dst = &s_coef[buf][slot][0][seg];
#pragma unroll
        for (int e = 0; e < NC; ++e) dst[e * PIPE_SEG] = o[e];
      }
      __syncthreads();
    }
  } else {
    // -------------------------------------------------------------------- columns: step p - 2 in phase p
    const int pair_raw = (wave - 2) * 4 + slot;
    const bool col_lane = pair_raw < NPAIR;
    const int pair = col_lane ? pair_raw : 0;        // spare lanes shadow pair 0 and store nothing
    double y[2][ND];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < ND; ++r) y[j][r] = (r == 2 * pair + j) ? 1.0 : 0.0;
    for (int p = 0; p < nphase; ++p) {
      if (p >= 2 && !(probe & 4)) {
        const int buf = p & 1;
        double acc[2][ND], yt[2][ND];
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          Coef vc;
          double* v = reinterpret_cast<double*>(&vc);
#pragma unroll
#ifdef PIPE_FEWREADS
          for (int e = 0; e < NC; ++e) v[e] = s_coef[buf][0][e][seg];
#else
          for (int e = 0; e < NC; ++e) v[e] = s_coef[buf][stage][e][seg];
#endif
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            double k[ND];
            if (stage == 0) pipe_col<ND>(vc, w2, y[j], k);
            else pipe_col<ND>(vc, w2, yt[j], k);
#pragma unroll
            for (int c = 0; c < ND; ++c) {
              if (stage == 0) { acc[j][c] = __builtin_fma(h6, k[c], y[j][c]); yt[j][c] = __builtin_fma(h2, k[c], y[j][c]); }
              else if (stage == 1) { acc[j][c] = __builtin_fma(h3, k[c], acc[j][c]); yt[j][c] = __builtin_fma(h2, k[c], y[j][c]); }
              else if (stage == 2) { acc[j][c] = __builtin_fma(h3, k[c], acc[j][c]); yt[j][c] = __builtin_fma(h, k[c], y[j][c]); }
              else y[j][c] = __builtin_fma(h6, k[c], acc[j][c]);
            }
          }
        }
      }
      __syncthreads();
    }
    if (in_range && col_lane) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < ND; ++r) a.Phi[(long)((2 * pair + j) * ND + r) * a.ldp + s] = y[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Six-wave form: the column role as ONE column per lane with a DPP row = one segment.
//
// In the four-wave kernel above a column lane reads all 17 / 25 coefficients of its segment from LDS at every stage
// (13 ds_read2_b64 per stage: ~14 issue cycles each, a quarter of the column waves' time).  Here the 16 lanes of a DPP row
// are the 14 (12) STM columns of ONE segment, lane j of the row holds coefficients j and 16 + j of that segment (one
// ds_read2_b64 per stage), and every product  coefficient x column entry  is a v_fmac_f64_dpp with row_newbcast:n --
// the coefficient is read from lane n of the row inside the FMA, no move, no LDS.  A wave now covers 4 segments, so the
// 16 segments of a workgroup need four column waves; the hardware places the waves of a workgroup on SIMDs round-robin
// (measured: tools/micro/dpp_probe.hip), so waves 0, 1, 4, 5 (two per SIMD on two SIMDs) take the columns and waves
// 2 and 3, each alone on its SIMD, the base and coefficient roles.
template <int N>
__device__ __forceinline__ void fmac_b(double& acc, const double c, const double x) {      // acc += c[lane N of the row] * x
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(N));
}
template <int N>
__device__ __forceinline__ void fmac_bn(double& acc, const double c, const double x) {     // acc -= c[lane N of the row] * x
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(N));
}

// cdot = F c for one STM column with the coefficients spread over the row: cA = value (lane), cB = value (16 + lane) of
//   [Gxx Gyy Gzz Gxy Gxz Gyz | Hxx Hyy Hzz Hxy Hxz Hyz | ua ub lx ly] [lz | umx umy umz mm mn Lm Ln Ll]   (VarCoef12 / 14)
// k[7..9] (ND = 14; k[6..8] for ND = 12) return H a + G d, i.e. MINUS the slope (the caller folds the sign into the
// RK update); every other entry is the slope itself.
template <int ND>
__device__ __forceinline__ void col_dpp(const double cA, const double cB, const double w2, const double (&c)[ND], double (&k)[ND]) {
  constexpr int G = (ND == 14) ? 7 : 6;     // first lambda_r row
  constexpr int D = G + 3;                  // first lambda_v row
  const double ax = c[0], ay = c[1], az = c[2];
  const double dx = c[D], dyv = c[D + 1], dz = c[D + 2];
  double ld = 0.0;
  fmac_b<14>(ld, cA, dx); fmac_b<15>(ld, cA, dyv); fmac_b<0>(ld, cB, dz);
  double tl = 0.0;
  fmac_b<13>(tl, cA, ld);
  k[0] = c[3]; k[1] = c[4]; k[2] = c[5];
  double k3 = w2 * c[4], k4 = -w2 * c[3], k5 = 0.0;
  fmac_b<0>(k3, cA, ax); fmac_b<3>(k3, cA, ay); fmac_b<4>(k3, cA, az); fmac_bn<12>(k3, cA, dx); fmac_b<14>(k3, cA, tl);
  fmac_b<3>(k4, cA, ax); fmac_b<1>(k4, cA, ay); fmac_b<5>(k4, cA, az); fmac_bn<12>(k4, cA, dyv); fmac_b<15>(k4, cA, tl);
  fmac_b<4>(k5, cA, ax); fmac_b<5>(k5, cA, ay); fmac_b<2>(k5, cA, az); fmac_bn<12>(k5, cA, dz); fmac_b<0>(k5, cB, tl);
  double s7 = 0.0, s8 = 0.0, s9 = 0.0;
  fmac_b<6>(s7, cA, ax); fmac_b<9>(s7, cA, ay); fmac_b<10>(s7, cA, az); fmac_b<0>(s7, cA, dx); fmac_b<3>(s7, cA, dyv); fmac_b<4>(s7, cA, dz);
  fmac_b<9>(s8, cA, ax); fmac_b<7>(s8, cA, ay); fmac_b<11>(s8, cA, az); fmac_b<3>(s8, cA, dx); fmac_b<1>(s8, cA, dyv); fmac_b<5>(s8, cA, dz);
  fmac_b<10>(s9, cA, ax); fmac_b<11>(s9, cA, ay); fmac_b<8>(s9, cA, az); fmac_b<4>(s9, cA, dx); fmac_b<5>(s9, cA, dyv); fmac_b<2>(s9, cA, dz);
  k[G] = s7; k[G + 1] = s8; k[G + 2] = s9;
  k[D] = __builtin_fma(w2, dyv, -c[G]);
  k[D + 1] = __builtin_fma(-w2, dx, -c[G + 1]);
  k[D + 2] = -c[G + 2];
  if constexpr (ND == 14) {
    const double mu = c[6];
    fmac_b<1>(k3, cB, mu); fmac_b<2>(k4, cB, mu); fmac_b<3>(k5, cB, mu);
    double k6 = 0.0, k13 = 0.0;
    fmac_b<4>(k6, cB, mu); fmac_b<5>(k6, cB, ld);
    fmac_b<6>(k13, cB, mu); fmac_b<7>(k13, cB, ld); fmac_b<8>(k13, cB, c[13]);
    k[6] = k6; k[13] = k13;
  }
  k[3] = k3; k[4] = k4; k[5] = k5;
}

constexpr int PIPE6_LDC = 33;   // doubles per (stage, segment) coefficient record: 32 + 1 so that the coefficient wave's
                                // stores (one record per lane) and the column lanes' loads (one row per record) are
                                // both free of bank conflicts

template <int ND, int PM>
__global__ __launch_bounds__(384) void k_indirect_pipe6(const IndirectArgs a) {
  using Coef = typename PipeArg<ND>::Coef;
  constexpr int NI = PipeArg<ND>::N;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  constexpr int G = (ND == 14) ? 7 : 6;

  __shared__ double s_int[8 + 3][NI][PIPE_SEG];             // base -> coef, as in k_indirect_pipe
  __shared__ double s_coef[2][4][PIPE_SEG][PIPE6_LDC];      // coef -> columns: [step parity][stage][segment][value]

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // base / coefficient waves: lane = (segment, lane group) as in k_indirect_pipe.  Column wave cw: lane = (row, column),
  // segment = 4 cw + row.
  const bool col_wave = (wave != 2 && wave != 3);
  const int cw = wave < 2 ? wave : wave - 2;                  // waves 0, 1, 4, 5 -> 0, 1, 2, 3
  const int seg = col_wave ? cw * 4 + (lane >> 4) : (lane & (PIPE_SEG - 1));
  const int slot = lane >> 4;                                 // lane group (base / coef waves)
  const int col = lane & 15;                                  // STM column (column waves)
  const int s_raw = blockIdx.x * PIPE_SEG + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;
  const int s = a.order ? a.order[s_lin] : s_lin;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform

  const int steps = a.steps;
  const double h = span / (double)steps;
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  const bool in_range = (s_raw < a.S) && mine;
  const int nphase = steps + 2;
#ifdef PIPE_PROBE
  const int probe = a.max_steps;  // development build only: bit 0 / 1 / 2 switches the base / coef / column work off
#else
  constexpr int probe = 0;
#endif

  if (wave == 2) {
    // -------------------------------------------------------------------- base: step p in phase p
    double y[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
    for (int p = 0; p < nphase; ++p) {
      if (p < steps && !(probe & 1)) {
        const int buf = p & 1;
        double k[ND], yt[ND], acc[ND];
        auto publish = [&](int stage, const double (&arg)[ND]) {
          double* dst = &s_int[slot == 0 ? buf * 4 + stage : 7 + slot][0][seg];
#pragma unroll
          for (int e = 0; e < NI; ++e) dst[e * PIPE_SEG] = arg[PipeArg<ND>::idx[e]];
        };
        publish(0, y);
        pipe_base_rhs<ND, PM>(y, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(1, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(2, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
        publish(3, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
      }
      __syncthreads();
    }
    if (in_range && slot == 0) {
      if (a.defect) {
#pragma unroll
        for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s] = y[c] - a.X[c * a.ldx + node + 1];
      }
      if (a.errors) a.errors[s] = 0.0;
      if (a.nacc) a.nacc[s] = steps;
      if (a.nrej) a.nrej[s] = 0;
    }
  } else if (wave == 3) {
    // -------------------------------------------------------------------- coefficients of step p - 1 in phase p
    for (int p = 0; p < nphase; ++p) {
      if (p >= 1 && p <= steps && !(probe & 2)) {
        const int buf = (p - 1) & 1;
        double arg[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) arg[c] = 0.0;
        const double* src = &s_int[buf * 4 + slot][0][seg];
#pragma unroll
        for (int e = 0; e < NI; ++e) arg[PipeArg<ND>::idx[e]] = src[e * PIPE_SEG];
        Coef vc;
        pipe_coef<ND, PM>(arg, tp, vc);
        const double* o = reinterpret_cast<const double*>(&vc);
        double* dst = &s_coef[buf][slot][seg][0];
#pragma unroll
        for (int e = 0; e < NC; ++e) dst[e] = o[e];
      }
      __syncthreads();
    }
  } else {
    // -------------------------------------------------------------------- columns: step p - 2 in phase p
    double y[ND];
#pragma unroll
    for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
    for (int p = 0; p < nphase; ++p) {
      if (p >= 2 && !(probe & 4)) {
        const int buf = p & 1;
        double acc[ND], yt[ND], k[ND];
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          const double* rec = &s_coef[buf][stage][seg][col];
          const double cA = rec[0], cB = rec[16];
          if (stage == 0) col_dpp<ND>(cA, cB, w2, y, k);
          else col_dpp<ND>(cA, cB, w2, yt, k);
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            const bool neg = (c >= G && c < G + 3);          // k holds minus the slope in the lambda_r rows
            const double b6 = neg ? -h6 : h6, b3 = neg ? -h3 : h3, a2 = neg ? -h2 : h2, a1 = neg ? -h : h;
            if (stage == 0) { acc[c] = __builtin_fma(b6, k[c], y[c]); yt[c] = __builtin_fma(a2, k[c], y[c]); }
            else if (stage == 1) { acc[c] = __builtin_fma(b3, k[c], acc[c]); yt[c] = __builtin_fma(a2, k[c], y[c]); }
            else if (stage == 2) { acc[c] = __builtin_fma(b3, k[c], acc[c]); yt[c] = __builtin_fma(a1, k[c], y[c]); }
            else y[c] = __builtin_fma(b6, k[c], acc[c]);
          }
        }
      }
      __syncthreads();
    }
    if (in_range && col < ND) {
#pragma unroll
      for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + s] = y[r];
    }
  }
}

template <int ND, int PM>
static hipError_t launch_pipe6_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + PIPE_SEG - 1) / PIPE_SEG);
  hipLaunchKernelGGL((k_indirect_pipe6<ND, PM>), grid, dim3(384), 0, st, a);
  return hipGetLastError();
}

template <int ND>
static hipError_t launch_pipe6_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe6_one<ND, PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe6_one<ND, PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe6_one<ND, PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe6_one<ND, PM_PGEN>(a, st);
  return e;
}

hipError_t launch_indirect_stm_pipe6(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe6_pm<12>(pm, a, st);
  if (ndim == 14) return launch_pipe6_pm<14>(pm, a, st);
  return hipErrorInvalidValue;
}

template <int ND, int PM>
static hipError_t launch_pipe_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + PIPE_SEG - 1) / PIPE_SEG);
  hipLaunchKernelGGL((k_indirect_pipe<ND, PM>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int ND>
static hipError_t launch_pipe_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe_one<ND, PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe_one<ND, PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe_one<ND, PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe_one<ND, PM_PGEN>(a, st);
  return e;
}

// RK4 only (the 13-stage methods use the cooperative kernel); steps >= 1.
hipError_t launch_indirect_stm_pipe(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe_pm<12>(pm, a, st);
  if (ndim == 14) return launch_pipe_pm<14>(pm, a, st);
  return hipErrorInvalidValue;
}

}  // namespace lto
