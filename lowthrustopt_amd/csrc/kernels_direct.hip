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
