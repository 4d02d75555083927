// kernels_direct.hip -- batched two-sided mid-point shooting of the given-thrust CRTBP system.
//
// Replaces defectCalc / jacobianCalc / the tf partial of src/multiShoot_CRTBP_direct.jl
// (:66-109, :111-143, :503-516).  Per segment the reference runs ode7_8 twice (forward half from node i,
// backward half from node i+1 with flipped velocity, same step grid :84,:95); here the two halves are
// adjacent lanes (lane parity = direction) and meet through a DPP/permute exchange, so the mid-point
// state never leaves registers.
//
// Jacobian: instead of 2(nstate+3) = 18-20 perturbed re-propagations per segment (forward differences,
// pert = 1e-8, :123-143) each lane integrates the base half-arc plus ONE sensitivity column of
// [Phi | Psi] (Phi = dx/dx0, Psi = dx/dcontrol) with the same RKF7(8) tableau and grid, i.e. the exact
// derivative of the discrete map.  Column index = blockIdx.y (wave-uniform).
#include "kernels.hpp"
#include "rk.hpp"

namespace lto {

template <int NS>
struct SysDirect {
  static constexpr int DIM = NS;
  DirectLane L;
  __device__ __forceinline__ void rhs(const double (&x)[NS], double (&k)[NS]) const {
    VarCoef6 vc;
    rhs_direct<NS, false>(x, L, k, vc);
  }
};

// base (NS) + one sensitivity column (NS)
template <int NS>
struct SysDirectVar {
  static constexpr int DIM = 2 * NS;
  DirectLane L;
  double ux, uy, uz;  // unit forcing direction of a control column (0 for state columns)
  double fm;          // d mdot / d control_j (7-state control columns)
  __device__ __forceinline__ void rhs(const double (&y)[2 * NS], double (&k)[2 * NS]) const {
    double xb[NS], kb[NS], c[NS], dc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { xb[i] = y[i]; c[i] = y[NS + i]; }
    VarCoef6 vc;
    rhs_direct<NS, true>(xb, L, kb, vc);
    var_col_direct<NS>(vc, L.w2, c, ux * vc.k_over_m, uy * vc.k_over_m, uz * vc.k_over_m, fm, dc);
#pragma unroll
    for (int i = 0; i < NS; ++i) { k[i] = kb[i]; k[NS + i] = dc[i]; }
  }
};

// Common per-lane setup: lane -> (segment, direction), initial half-arc state and DirectLane.
template <int NS>
__device__ __forceinline__ void direct_setup(const DirectArgs& a, int& s, int& dir, double& hhalf,
                                             double& span_total, double (&x)[NS], DirectLane& L, double& nc) {
  const int gid = blockIdx.x * 64 + threadIdx.x;
  s = gid >> 1;
  dir = gid & 1;
  const int sc = (s < a.S) ? s : a.S - 1;  // inactive lanes shadow the last segment (keeps exchanges defined)
  const int traj = sc / a.seg_per_traj;
  const int i = sc - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i + dir;
  const long tg = (long)traj * a.t_stride;
  hhalf = 0.5 * (a.t[tg + i + 1] - a.t[tg + i]);            // t_mid - t_i            (direct.jl:70)
  span_total = a.t[tg + a.n_nodes - 1] - a.t[tg];           // tf - t0                (direct.jl:506-510)
#pragma unroll
  for (int c = 0; c < NS; ++c) x[c] = a.X[c * a.ldx + node];
  if (dir) { x[3] = -x[3]; x[4] = -x[4]; x[5] = -x[5]; }     // reverse velocity        (direct.jl:92)
  const double td = dir ? -1.0 : 1.0;
  L.MU = a.MU;
  L.w2 = 2.0 * td;
  L.cx = a.U[0 * a.ldu + node]; L.cy = a.U[1 * a.ldu + node]; L.cz = a.U[2 * a.ldu + node];
  L.kk = a.kk;
  { const double k6 = L.kk * 1e-3; L.tx = L.cx * k6; L.ty = L.cy * k6; L.tz = L.cz * k6; }   // NS = 6: control * kk / 1000.0
  nc = sqrt(__builtin_fma(L.cx, L.cx, __builtin_fma(L.cy, L.cy, L.cz * L.cz)));
  L.mdot = -td * nc / a.isp_g0 * a.TU;                       // prop_EP_deriv.jl:42
}

__device__ __forceinline__ double xchg1(double v) { return __shfl_xor(v, 1); }

// K3: defect + RKF7(8) error estimate.
template <int NS>
__global__ __launch_bounds__(64) void k_direct_defect(const DirectArgs a) {
  int s, dir; double hhalf, span_total, nc;
  double x[NS];
  SysDirect<NS> sys;
  direct_setup<NS>(a, s, dir, hhalf, span_total, x, sys.L, nc);
  const double h = hhalf / (double)a.half_steps;
  double maxErr = 0.0;
  for (int k = 0; k < a.half_steps; ++k) {
    double xn[NS];
    const double delta = rkf78_step<SysDirect<NS>, NS>(sys, h, x, xn);
    maxErr = fmax(maxErr, delta);
#pragma unroll
    for (int c = 0; c < NS; ++c) x[c] = xn[c];
  }
  if (dir) { x[3] = -x[3]; x[4] = -x[4]; x[5] = -x[5]; }     // direct.jl:98
  const bool active = s < a.S;   // recomputed here: no per-lane boolean is kept live across the integrator
  double d[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) d[c] = x[c] - xchg1(x[c]);    // fwd lane: state_for - stateF_back  (:101)
  const double e = fmax(maxErr, xchg1(maxErr));              // :104
  if (active && dir == 0) {
    if (a.mid) {                                             // node meshRefine_direct inserts (direct.jl:651-660)
#pragma unroll
      for (int c = 0; c < NS; ++c) a.mid[c * a.ldm + s] = x[c];
    }
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < NS; ++c) a.defect[c * a.ldd + s] = d[c];
    }
    if (a.errors) a.errors[s] = e;
  }
}

// K4: one sensitivity column per lane.  blockIdx.y = j: j < NS -> d/d(initial state j); j >= NS ->
// d/d(control j-NS).  Output block layout (variable order of direct.jl:125):
//   cols [0,NS)        d defect / d x_i     = Phi_f
//   cols [NS,2NS)      d defect / d x_{i+1} = -R Phi_b R      R = diag(1,1,1,-1,-1,-1[,1])
//   cols [2NS,2NS+3)   d defect / d u_i     = Psi_f
//   cols [2NS+3,2NS+6) d defect / d u_{i+1} = -R Psi_b
template <int NS>
__global__ __launch_bounds__(64) void k_direct_jacobian(const DirectArgs a) {
  int s, dir; double hhalf, span_total, nc;
  double x[NS];
  SysDirectVar<NS> sys;
  direct_setup<NS>(a, s, dir, hhalf, span_total, x, sys.L, nc);
  const int j = blockIdx.y;
  const bool is_ctrl = j >= NS;
  const int jc = j - NS;
  sys.ux = (is_ctrl && jc == 0) ? 1.0 : 0.0;
  sys.uy = (is_ctrl && jc == 1) ? 1.0 : 0.0;
  sys.uz = (is_ctrl && jc == 2) ? 1.0 : 0.0;
  sys.fm = 0.0;
  if (NS == 7 && is_ctrl) {
    // d mdot / d c_j = -td TU / (Isp g0) c_j / |c|; at c = 0 the one-sided value the reference's forward
    // difference sees (d|c|/dc_j = 1).
    const double cj = (jc == 0) ? sys.L.cx : (jc == 1 ? sys.L.cy : sys.L.cz);
    const double dn = (nc > 0.0) ? cj / nc : 1.0;
    sys.fm = -(0.5 * sys.L.w2) * dn / a.isp_g0 * a.TU;
  }
  double y[2 * NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) { y[c] = x[c]; y[NS + c] = (c == j) ? 1.0 : 0.0; }

  const double h = hhalf / (double)a.half_steps;
  double maxErr = 0.0;
  for (int k = 0; k < a.half_steps; ++k) {
    double yn[2 * NS];
    const double delta = rkf78_step<SysDirectVar<NS>, NS>(sys, h, y, yn);
    maxErr = fmax(maxErr, delta);
#pragma unroll
    for (int c = 0; c < 2 * NS; ++c) y[c] = yn[c];
  }

  const bool active = s < a.S;   // recomputed here: no per-lane boolean is kept live across the integrator
  // sensitivity column -> Jacobian block column
  if (active && a.Jac) {
    const int col = is_ctrl ? (2 * NS + 3 * dir + jc) : (NS * dir + j);
    const double rj = (!is_ctrl && j >= 3 && j < 6) ? -1.0 : 1.0;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      const double rr = (r >= 3 && r < 6) ? -1.0 : 1.0;
      const double v = dir ? -(rr * rj) * y[NS + r] : y[NS + r];
      a.Jac[(long)(col * NS + r) * a.ldj + s] = v;
    }
  }

  if (j == 0) {  // wave-uniform: the column-0 lanes also emit defect, errors and the tf partial
    double f[NS];
    {
      double xb[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) xb[c] = y[c];
      VarCoef6 vc;
      rhs_direct<NS, false>(xb, sys.L, f, vc);
    }
    double xe[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) xe[c] = y[c];
    if (dir) {
      xe[3] = -xe[3]; xe[4] = -xe[4]; xe[5] = -xe[5];
      f[3] = -f[3]; f[4] = -f[4]; f[5] = -f[5];   // R f_b
    }
    const double scale = hhalf / span_total;        // d(half length)/d tf = hhalf / (tf - t0)
    const double e = fmax(maxErr, xchg1(maxErr));
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      const double d = xe[c] - xchg1(xe[c]);
      const double g = (f[c] - xchg1(f[c])) * scale;
      if (active && dir == 0) {
        if (a.defect) a.defect[c * a.ldd + s] = d;
        if (a.dtf) a.dtf[c * a.ldd + s] = g;
      }
    }
    if (active && dir == 0 && a.errors) a.errors[s] = e;
  }
}

// (A wave-specialised form with one barrier per RK stage -- base wave + five column waves per 16 segments -- measured no faster than
// the per-lane kernel and slower than the pipeline below at every size; AUTO never chose it.  Removed in round 3.)
// K4'': software-pipelined Jacobian kernel (BASELINE configs[2]).  In the per-lane kernel every sensitivity column lane
// re-integrates the half-arc base state (RHS, gravity gradient and the stage arguments of the base: ~85 of its ~125
// instructions per RK stage), nine times per arc (arc = segment x direction); the cooperative kernel removes that but
// meets at a barrier every RK stage with a half-empty base wave on the critical path.  Here a workgroup owns 32 segments
// = 64 arcs and runs, skewed by one RKF7(8) STEP:
//   wave 0          base role, lane = arc (all 64 lanes busy): the NS-dim half-arc; publishes the variational
//                   coefficients (G, k/m [, dv/dm]) of all 13 stages of its current step to LDS
//   waves 1..NS+3   wave w = sensitivity column w - 1 of [Phi | Psi] for all 64 arcs (wave-uniform column: no divergence,
//                   LDS reads with lane = arc are conflict-free); one step behind, c' = A(t) c + forcing only
// One barrier per step (half_steps + 1 phases), coefficient slabs double-buffered (93 / 133 KB of LDS: one workgroup per CU,
// three waves per SIMD).  Per arc and stage ~9.3 wave-instructions against ~17.6 in the per-lane kernel.  Same tableau, step
// grid, arithmetic per column and output layout as k_direct_jacobian.
// Wave w of a workgroup runs on SIMD w mod 4 (tools/micro/sync_probe.hip).  Twelve wavefronts are launched so that the column
// waves sit three to a SIMD (~2 150 instructions per step) and the base wave (~1 300, the longest dependent stream) has
// SIMD 0 to itself (NS = 6: waves 4 and 8 leave at once) or shares it with one column wave (NS = 7: ten columns, wave 8
// leaves).  With ten waves in launch order SIMD 0 carried the base wave AND two column waves: 2 700 instructions per step
// against 1 400-2 150 on the others, and the step barrier waited for it (95.5 -> 79.7 us at 16 384 segments).
template <int NS>
__device__ __forceinline__ int direct_pipe_column_of_wave(const int w) {   // -1: leaves at once
  if (NS == 6) return (w & 3) == 0 ? -1 : (w & 3) - 1 + 3 * (w >> 2);       // waves 1 2 3 | 5 6 7 | 9 10 11 -> columns 0..8
  return (w == 8) ? -1 : (w < 8 ? w - 1 : w - 2);                           // waves 1..7, 9..11 -> columns 0..9
}

template <int NS>
__global__ __launch_bounds__(768) void k_direct_jacobian_pipe(const DirectArgs a) {
  constexpr int ARCS = 64;
  constexpr int NC = (NS == 7) ? 10 : 7;       // doubles handed over per (arc, stage)
  __shared__ double s_coef[2][13][NC][ARCS];
  __shared__ double s_x[2 * NS + 1][ARCS];     // mid-point exchange: [xe (NS) | R f (NS) | maxErr][arc]
  __shared__ double s_keep[ARCS];              // per arc: what only the epilogue needs

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), arc = threadIdx.x & 63;
  const bool is_base = (wave == 0);
  const int jw = is_base ? 0 : direct_pipe_column_of_wave<NS>(wave);
  if (jw < 0) return;                          // before any barrier: the hardware barrier counts the waves still alive
  const int j = jw;                            // sensitivity column of this wave
  const int seg = arc >> 1, dir = arc & 1;     // forward / backward half-arcs of a segment are lane neighbours
  const int s_raw = blockIdx.x * (ARCS / 2) + seg;
  const int s = s_raw < a.S ? s_raw : a.S - 1; // shadow lanes repeat the last segment, store nothing (the epilogue decides from the lane id again)

  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i + dir;
  const long tg = (long)traj * a.t_stride;
  const double hhalf = 0.5 * (a.t[tg + i + 1] - a.t[tg + i]);            // direct.jl:70
  const double span_total = a.t[tg + a.n_nodes - 1] - a.t[tg];           // direct.jl:506-510
  // d(half length) / d tf is needed after the step loop only: formed here and parked in LDS, so that neither it nor the three grid
  // values it comes from occupy registers through the loop (at 168 registers per lane they went to scratch: 10 MB of HBM writes
  // per launch at 16 384 segments)
  if (is_base) s_keep[arc] = hhalf / span_total;
  const double td = dir ? -1.0 : 1.0;
  DirectLane L;
  L.MU = a.MU; L.w2 = 2.0 * td; L.kk = a.kk;
  L.cx = a.U[0 * a.ldu + node]; L.cy = a.U[1 * a.ldu + node]; L.cz = a.U[2 * a.ldu + node];
  { const double k6 = L.kk * 1e-3; L.tx = L.cx * k6; L.ty = L.cy * k6; L.tz = L.cz * k6; }   // NS = 6: control * kk / 1000.0
  const double nc = sqrt(__builtin_fma(L.cx, L.cx, __builtin_fma(L.cy, L.cy, L.cz * L.cz)));
  L.mdot = -td * nc / a.isp_g0 * a.TU;                                   // prop_EP_deriv.jl:42

  double y[NS], K[13][NS];
  const bool is_ctrl = j >= NS;
  const int jc = j - NS;
  double fx = 0.0, fy = 0.0, fz = 0.0, fm = 0.0;
  if (is_base) {
#pragma unroll
    for (int c = 0; c < NS; ++c) y[c] = a.X[c * a.ldx + node];
    if (dir) { y[3] = -y[3]; y[4] = -y[4]; y[5] = -y[5]; }               // direct.jl:92
  } else {
#pragma unroll
    for (int c = 0; c < NS; ++c) y[c] = (c == j) ? 1.0 : 0.0;
    fx = (is_ctrl && jc == 0) ? 1.0 : 0.0;
    fy = (is_ctrl && jc == 1) ? 1.0 : 0.0;
    fz = (is_ctrl && jc == 2) ? 1.0 : 0.0;
    if (NS == 7 && is_ctrl) {
      const double cj = (jc == 0) ? L.cx : (jc == 1 ? L.cy : L.cz);
      const double dn = (nc > 0.0) ? cj / nc : 1.0;
      fm = -td * dn / a.isp_g0 * a.TU;
    }
  }

  const int steps = a.half_steps;
  const double h = hhalf / (double)steps;
  double maxErr = 0.0;
  // one RKF7(8) step of this lane's NS components; slope(st, arg, out) evaluates stage st
  // NS = 7: slope by slope with the tableau coefficient materialised where it is used (coef_here, rk.hpp): with seven components,
  // ten column waves and the mass couplings the compiler otherwise hoists the 64-bit literals of the unrolled stage loop, runs out
  // of scalar registers and parks two dozen of them in scratch (48 dwords, reloaded every step).
  constexpr bool BY_SLOPE = (NS == 7);
  // NS = 7 also forms the arguments of the last three stages, the weighted sum and the error term EAGERLY: with seven components
  // ten slopes are alive at stage 12 (K0, K3 ... K11; 182 registers with y and the argument, against 168 per lane at three waves
  // per SIMD -- 13 dwords went to scratch).  From stage EAGER - 1 on every sum that still needs an old slope is carried as an
  // accumulator instead: behind stage 8's argument the sums of stages 9 ... 12, of the weights and of the error term take K0 ... K7 (in
  // the tableau's order, so every chain of FMAs is the one it was: bit-identical), then each new slope is fed to them as it appears
  // and dies.  Scratch per lane by threshold: 7: 20 B, 8 and 9: none, 10: 12 B, 11: 20 B, none of it (before): 56 B.
  constexpr int EAGER = 9;
  auto rk_step = [&](auto&& slope) {
    double accE[13 - EAGER][NS], accB[NS], gerr[NS];
    const double cerr = __ddiv_rn(__dmul_rn(h, 41.0), 840.0);             // rkf78_err_term's scale (ode.jl:940)
    auto feed = [&](const int k) {                                        // slope k enters every eager sum that uses it
#pragma unroll
      for (int t = EAGER; t < 13; ++t)
        if (t > k && TabRKF78::A[t][k] != 0.0) {
          const double w = coef_here(TabRKF78::A[t][k]);
#pragma unroll
          for (int c = 0; c < NS; ++c) accE[t - EAGER][c] = __builtin_fma(w, K[k][c], accE[t - EAGER][c]);
        }
      if (TabRKF78::B[k] != 0.0) {
        const double w = coef_here(TabRKF78::B[k]);
#pragma unroll
        for (int c = 0; c < NS; ++c) accB[c] = __builtin_fma(w, K[k][c], accB[c]);
      }
      if (TabRKF78::E[k] != 0.0 && is_base) {                            // the four terms of rkf78_err_term, one at a time in its order
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          const double term = __dmul_rn(cerr, K[k][c]);
          gerr[c] = (k == 0) ? term : __dadd_rn(gerr[c], TabRKF78::E[k] > 0.0 ? term : -term);
        }
      }
    };
#pragma unroll
    for (int st = 0; st < 13; ++st) {
      double arg[NS];
      if constexpr (BY_SLOPE) {
        if (st < EAGER) {
          double acc[NS];
#pragma unroll
          for (int c = 0; c < NS; ++c) acc[c] = 0.0;
#pragma unroll
          for (int k = 0; k < st; ++k)
            if (TabRKF78::A[st][k] != 0.0) {
              const double w = coef_here(TabRKF78::A[st][k]);
#pragma unroll
              for (int c = 0; c < NS; ++c) acc[c] = __builtin_fma(w, K[k][c], acc[c]);
            }
#pragma unroll
          for (int c = 0; c < NS; ++c) arg[c] = (st == 0) ? y[c] : __builtin_fma(h, acc[c], y[c]);
        } else {
#pragma unroll
          for (int c = 0; c < NS; ++c) arg[c] = __builtin_fma(h, accE[st - EAGER][c], y[c]);
        }
        if (st == EAGER - 1) {
#pragma unroll
          for (int t = 0; t < 13 - EAGER; ++t)
#pragma unroll
            for (int c = 0; c < NS; ++c) accE[t][c] = 0.0;
#pragma unroll
          for (int c = 0; c < NS; ++c) { accB[c] = 0.0; gerr[c] = 0.0; }
#pragma unroll
          for (int k = 0; k < EAGER - 1; ++k) feed(k);
        }
      } else {
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < st; ++k)
            if (TabRKF78::A[st][k] != 0.0) acc = __builtin_fma(TabRKF78::A[st][k], K[k][c], acc);
          arg[c] = (st == 0) ? y[c] : __builtin_fma(h, acc, y[c]);
        }
      }
      slope(st, arg, K[st]);
      if constexpr (BY_SLOPE) {
        if (st >= EAGER - 1) feed(st);
      }
    }
    if constexpr (BY_SLOPE) {
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        if (is_base) maxErr = fmax(maxErr, fabs(gerr[c]));
        y[c] = __builtin_fma(h, accB[c], y[c]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 13; ++k)
          if (TabRKF78::B[k] != 0.0) acc = __builtin_fma(TabRKF78::B[k], K[k][c], acc);
        if (is_base) maxErr = fmax(maxErr, fabs(rkf78_err_term(h, K[0][c], K[10][c], K[11][c], K[12][c])));
        y[c] = __builtin_fma(h, acc, y[c]);
      }
    }
  };

  for (int p = 0; p < steps + 1; ++p) {
    if (is_base) {
      if (p < steps) {
        double (*slab)[NC][ARCS] = s_coef[p & 1];
        rk_step([&](int st, const double (&arg)[NS], double (&out)[NS]) {
          VarCoef6 vc;
          rhs_direct<NS, true>(arg, L, out, vc);
          slab[st][0][arc] = vc.Gxx; slab[st][1][arc] = vc.Gyy; slab[st][2][arc] = vc.Gzz;
          slab[st][3][arc] = vc.Gxy; slab[st][4][arc] = vc.Gxz; slab[st][5][arc] = vc.Gyz;
          slab[st][6][arc] = vc.k_over_m;
          if (NS == 7) { slab[st][7][arc] = vc.dvdm_x; slab[st][8][arc] = vc.dvdm_y; slab[st][9][arc] = vc.dvdm_z; }
        });
      }
    } else if (p >= 1) {
      const double (*slab)[NC][ARCS] = s_coef[(p - 1) & 1];
      rk_step([&](int st, const double (&arg)[NS], double (&out)[NS]) {
        VarCoef6 vc;
        vc.Gxx = slab[st][0][arc]; vc.Gyy = slab[st][1][arc]; vc.Gzz = slab[st][2][arc];
        vc.Gxy = slab[st][3][arc]; vc.Gxz = slab[st][4][arc]; vc.Gyz = slab[st][5][arc];
        vc.k_over_m = slab[st][6][arc];
        if (NS == 7) { vc.dvdm_x = slab[st][7][arc]; vc.dvdm_y = slab[st][8][arc]; vc.dvdm_z = slab[st][9][arc]; }
        var_col_direct<NS>(vc, L.w2, arg, fx * vc.k_over_m, fy * vc.k_over_m, fz * vc.k_over_m, fm, out);
      });
    }
    __syncthreads();
  }

  // Epilogue.  Who this lane is comes from the lane id again (two instructions) rather than from registers kept through the loop.
  const int arc_e = __lane_id();
  const int dir_e = arc_e & 1;
  const int s_e = blockIdx.x * (ARCS / 2) + (arc_e >> 1);
  const bool writer_e = s_e < a.S;
  if (!is_base) {
    if (writer_e && a.Jac) {
      const int col = is_ctrl ? (2 * NS + 3 * dir_e + jc) : (NS * dir_e + j);
      const double rj = (!is_ctrl && j >= 3 && j < 6) ? -1.0 : 1.0;
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const double rr = (r >= 3 && r < 6) ? -1.0 : 1.0;
        a.Jac[(long)(col * NS + r) * a.ldj + s_e] = dir_e ? -(rr * rj) * y[r] : y[r];
      }
    }
  } else {
    // forward and backward halves meet: publish (x_end, R f(x_end), maxErr), combine on the forward lane
    double f[NS];
    VarCoef6 vc;
    rhs_direct<NS, false>(y, L, f, vc);
    if (dir_e) { y[3] = -y[3]; y[4] = -y[4]; y[5] = -y[5]; f[3] = -f[3]; f[4] = -f[4]; f[5] = -f[5]; }   // direct.jl:98
#pragma unroll
    for (int c = 0; c < NS; ++c) { s_x[c][arc_e] = y[c]; s_x[NS + c][arc_e] = f[c]; }
    s_x[2 * NS][arc_e] = maxErr;
  }
  __syncthreads();
  if (is_base && writer_e && dir_e == 0) {
    const double scale = s_keep[arc_e];          // d(half length) / d tf
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      if (a.defect) a.defect[c * a.ldd + s_e] = s_x[c][arc_e] - s_x[c][arc_e + 1];                            // :101
      if (a.dtf) a.dtf[c * a.ldd + s_e] = (s_x[NS + c][arc_e] - s_x[NS + c][arc_e + 1]) * scale;
    }
    if (a.errors) a.errors[s_e] = fmax(s_x[2 * NS][arc_e], s_x[2 * NS][arc_e + 1]);                            // :104
  }
}

hipError_t launch_direct_jacobian_pipe(int nstate, const DirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  dim3 grid((a.S + 31) / 32);
  if (nstate == 6) hipLaunchKernelGGL((k_direct_jacobian_pipe<6>), grid, dim3(768), 0, st, a);
  else if (nstate == 7) hipLaunchKernelGGL((k_direct_jacobian_pipe<7>), grid, dim3(768), 0, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_direct_defect(int nstate, const DirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  dim3 grid((2 * (long)a.S + 63) / 64);
  if (nstate == 6) hipLaunchKernelGGL((k_direct_defect<6>), grid, dim3(64), 0, st, a);
  else if (nstate == 7) hipLaunchKernelGGL((k_direct_defect<7>), grid, dim3(64), 0, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_direct_jacobian(int nstate, const DirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  dim3 grid((2 * (long)a.S + 63) / 64, nstate + 3);
  if (nstate == 6) hipLaunchKernelGGL((k_direct_jacobian<6>), grid, dim3(64), 0, st, a);
  else if (nstate == 7) hipLaunchKernelGGL((k_direct_jacobian<7>), grid, dim3(64), 0, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace lto
