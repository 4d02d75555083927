// kernels_indirect_lane.hip -- RK4 STM sweep for batches that fill the chip many times over: lane = ONE WHOLE SEGMENT.
//
// Replaces jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) with fixed-step RK4 at BASELINE configs[3] sizes (256 levels x
// 1 024 segments).  Once every SIMD has wavefronts to spare, only the number of instructions issued per segment counts.  The
// three-role pipelines (kernels_indirect_pipe48.hip) spend 118 wave-instructions per segment and RK4 step: a DPP row of 16 lanes
// carries the 12 columns of a segment (a quarter of the column lanes idle), the roles meet at a barrier per step, coefficients
// travel through LDS.  Here a lane owns its segment outright:
//
//   per RK4 step   base trajectory, four stage evaluations (rhs12_base: the pipelines' base role, same operations, same bits)
//                  -> the four stage matrices G, H, U from the stage arguments' position and lambda_v (rhs12<PM, true>: the
//                     pipelines' coefficient role), scaled by the stage weights, 68 doubles in registers
//                  -> the twelve STM columns one after the other through those four matrices (col_reg_step: col_dpp_step of
//                     pipe_common.hpp with the coefficient in a register instead of a DPP broadcast -- the same FMAs in the same
//                     order, so Phi equals the pipelines' bit for bit).
//   No LDS, no barrier, no idle lane, nothing computed twice: ~60 wave-instructions per segment and step.
//
// The price is state: Phi is 144 doubles per lane, next to 68 doubles of stage matrices and ~50 of a column in flight.  The kernel
// runs ONE wavefront per SIMD (512 registers per lane: 256 addressable + 256 accumulation registers): six columns live in
// registers -- the compiler parks them in the accumulation registers and moves one in and out per step, ~830 v_accvgpr moves per
// step against ~3 800 FP64 instructions -- and six in LDS (36 KB per wavefront, four wavefronts per CU; 16-byte pieces, one per lane
// and access).  More register columns spill to scratch, more LDS columns do not fit four wavefronts per CU (LANE_COLS_IN_REGISTERS).
// So it needs 64 segments per SIMD to fill the chip: AUTO compares its rounds of 256 x CUs segments with the pipelines' (lto_api.hip).
// The stage arguments pass through an empty asm before the matrices are built from them: otherwise the compiler merges that
// evaluation into the base stage's and keeps ~20 by-products per stage alive across the step (+110 registers).
// 12-dim; every control-law class; RK4 with any number of steps.
#include "kernels.hpp"
#include "pipe_common.hpp"

namespace lto {

// One RK4 stage of one STM column, coefficients in registers: out = init + a F(arg) arg, all of vc but lhat pre-multiplied by the
// stage weight a, aw2 = a 2 omega.  Operation for operation col_dpp_stage<12> (pipe_common.hpp).
__device__ __forceinline__ void col_reg_stage(const VarCoef12& v, const double aw2, const double a, const double (&arg)[12],
                                              const double (&init)[12], double (&out)[12]) {
  const double ax = arg[0], ay = arg[1], az = arg[2];
  const double dx = arg[9], dyv = arg[10], dz = arg[11];
  double ld = __builtin_fma(v.lx, dx, 0.0);
  ld = __builtin_fma(v.ly, dyv, ld);
  ld = __builtin_fma(v.lz, dz, ld);
  const double tl = __builtin_fma(v.ub, ld, 0.0);              // a ub (lhat . d)
  out[0] = __builtin_fma(a, arg[3], init[0]); out[1] = __builtin_fma(a, arg[4], init[1]); out[2] = __builtin_fma(a, arg[5], init[2]);
  double o3 = __builtin_fma(aw2, arg[4], init[3]), o4 = __builtin_fma(-aw2, arg[3], init[4]), o5 = init[5];
  o3 = __builtin_fma(v.Gxx, ax, o3); o3 = __builtin_fma(v.Gxy, ay, o3); o3 = __builtin_fma(v.Gxz, az, o3); o3 = __builtin_fma(-v.ua, dx, o3); o3 = __builtin_fma(v.lx, tl, o3);
  o4 = __builtin_fma(v.Gxy, ax, o4); o4 = __builtin_fma(v.Gyy, ay, o4); o4 = __builtin_fma(v.Gyz, az, o4); o4 = __builtin_fma(-v.ua, dyv, o4); o4 = __builtin_fma(v.ly, tl, o4);
  o5 = __builtin_fma(v.Gxz, ax, o5); o5 = __builtin_fma(v.Gyz, ay, o5); o5 = __builtin_fma(v.Gzz, az, o5); o5 = __builtin_fma(-v.ua, dz, o5); o5 = __builtin_fma(v.lz, tl, o5);
  double o7 = init[6], o8 = init[7], o9 = init[8];             // lambda_r rows: minus (H a + G d)
  o7 = __builtin_fma(-v.Hxx, ax, o7); o7 = __builtin_fma(-v.Hxy, ay, o7); o7 = __builtin_fma(-v.Hxz, az, o7);
  o7 = __builtin_fma(-v.Gxx, dx, o7); o7 = __builtin_fma(-v.Gxy, dyv, o7); o7 = __builtin_fma(-v.Gxz, dz, o7);
  o8 = __builtin_fma(-v.Hxy, ax, o8); o8 = __builtin_fma(-v.Hyy, ay, o8); o8 = __builtin_fma(-v.Hyz, az, o8);
  o8 = __builtin_fma(-v.Gxy, dx, o8); o8 = __builtin_fma(-v.Gyy, dyv, o8); o8 = __builtin_fma(-v.Gyz, dz, o8);
  o9 = __builtin_fma(-v.Hxz, ax, o9); o9 = __builtin_fma(-v.Hyz, ay, o9); o9 = __builtin_fma(-v.Hzz, az, o9);
  o9 = __builtin_fma(-v.Gxz, dx, o9); o9 = __builtin_fma(-v.Gyz, dyv, o9); o9 = __builtin_fma(-v.Gzz, dz, o9);
  out[3] = o3; out[4] = o4; out[5] = o5;
  out[6] = o7; out[7] = o8; out[8] = o9;
  out[9] = __builtin_fma(aw2, dyv, __builtin_fma(-a, arg[6], init[9]));
  out[10] = __builtin_fma(-aw2, dx, __builtin_fma(-a, arg[7], init[10]));
  out[11] = __builtin_fma(-a, arg[8], init[11]);
}

// One RK4 step of one column, carrying u = 3^k y: col_dpp_step<12> of pipe_common.hpp with register coefficients; its rescaling by
// 3^-256 every 256 steps is done by the caller for all columns at once (the same multiplication on the same value, one branch per
// step instead of twelve).
__device__ __forceinline__ void col_reg_step(const VarCoef12 (&vc)[4], const ColStepConst& k, double (&y)[12]) {
  double B[12], V1[12], V2[12], V3[12];
  col_reg_stage(vc[0], k.h2w, k.h2, y, y, V1);
#pragma unroll
  for (int c = 0; c < 12; ++c) B[c] = V1[c] - y[c];
  col_reg_stage(vc[1], k.h2w, k.h2, V1, y, V2);
#pragma unroll
  for (int c = 0; c < 12; ++c) B[c] = __builtin_fma(2.0, V2[c], B[c]);
  col_reg_stage(vc[2], k.hw, k.h, V2, y, V3);
#pragma unroll
  for (int c = 0; c < 12; ++c) B[c] += V3[c];
  col_reg_stage(vc[3], k.h2w, k.h2, V3, B, y);
}

constexpr int LANE_COLS_IN_REGISTERS = 6;

template <int PM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_indirect_lane(const IndirectArgs a) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(tp.p) != PM) return;   // mixed-class batch: another launch owns this trajectory
  const int steps = a.steps;
  const double h = (a.t[tg + 1] - a.t[tg]) / (double)steps;
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  const ColStepConst kc(h, 2.0 * tp.omega);

  double y[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) y[c] = a.X[c * a.ldx + node];
  // Phi: columns 0 .. NREG-1 in registers (the compiler parks them in the accumulation registers), the others in LDS, one 16-byte
  // piece per lane and access (conflict-free), read back through an offset the compiler cannot see through (it would otherwise
  // forward the stored registers across the loop and keep them alive -- 144 + 68 + 36 doubles do not fit 512 registers)
  constexpr int NREG = LANE_COLS_IN_REGISTERS, NLDS = 12 - NREG;
  __shared__ double2 s_park[(NLDS > 0 ? NLDS : 1) * 6][64];
  int lane_opaque = threadIdx.x;
  asm volatile("" : "+v"(lane_opaque));
  double phi[NREG > 0 ? NREG : 1][12];
#pragma unroll
  for (int col = 0; col < 12; ++col)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double e0 = (2 * q == col) ? 1.0 : 0.0, e1 = (2 * q + 1 == col) ? 1.0 : 0.0;
      if (col < NREG) { phi[col][2 * q] = e0; phi[col][2 * q + 1] = e1; }
      else s_park[(col - NREG) * 6 + q][threadIdx.x] = make_double2(e0, e1);
    }

  for (int step = 0; step < steps; ++step) {
    // base trajectory (the pipelines' base role): one RK4 step; of every stage argument the position and lambda_v are kept
    double arg[4][6];
    {
      double k[12], yt[12], acc[12];
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[0][c] = y[c]; arg[0][3 + c] = y[9 + c]; }
      rhs12_base<PM>(y, tp, k);
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[1][c] = yt[c]; arg[1][3 + c] = yt[9 + c]; }
      rhs12_base<PM>(yt, tp, k);
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[2][c] = yt[c]; arg[2][3 + c] = yt[9 + c]; }
      rhs12_base<PM>(yt, tp, k);
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[3][c] = yt[c]; arg[3][3 + c] = yt[9 + c]; }
      rhs12_base<PM>(yt, tp, k);
#pragma unroll
      for (int c = 0; c < 12; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
    }
    // the four stage matrices (the pipelines' coefficient role): rhs12<PM, true> on the stage argument with everything but the
    // position and lambda_v zero, its slopes dead; G, H, ua, ub times the stage weight (h/2, h/2, h, h/2: col_dpp_step's scheme)
    VarCoef12 vc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double ya[12], dead[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) ya[c] = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(arg[j][c]));   // opaque: the compiler otherwise merges this evaluation into the
                                                                         // base stage's and keeps ~20 by-products per stage alive across the step
#pragma unroll
      for (int c = 0; c < 3; ++c) { ya[c] = arg[j][c]; ya[9 + c] = arg[j][3 + c]; }
      __builtin_amdgcn_sched_barrier(0);           // one stage's matrices at a time
      rhs12<PM, true>(ya, tp, dead, vc[j]);
      const double as = (j == 2) ? h : h2;
      double* o = reinterpret_cast<double*>(&vc[j]);
#pragma unroll
      for (int e = 0; e < 14; ++e) o[e] = o[e] * as;
    }
#pragma unroll
    for (int col = 0; col < 12; ++col) {
      __builtin_amdgcn_sched_barrier(0);           // one column at a time: interleaving two of them costs more registers than there are
      if (col < NREG) col_reg_step(vc, kc, phi[col]);
      else {
        double u[12];
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double2 t = s_park[(col - NREG) * 6 + q][lane_opaque]; u[2 * q] = t.x; u[2 * q + 1] = t.y; }
        col_reg_step(vc, kc, u);
#pragma unroll
        for (int q = 0; q < 6; ++q) s_park[(col - NREG) * 6 + q][threadIdx.x] = make_double2(u[2 * q], u[2 * q + 1]);
      }
    }
    if (((step + 1) & (COL_RESCALE_EVERY - 1)) == 0) {       // every 256 steps: u <- 3^-256 u, all columns (pipe_common.hpp COL_RESCALE)
#pragma unroll
      for (int col = 0; col < NREG; ++col)
#pragma unroll
        for (int c = 0; c < 12; ++c) phi[col][c] *= COL_RESCALE;
#pragma unroll
      for (int j = 0; j < NLDS * 6; ++j) {
        double2 t = s_park[j][lane_opaque];
        t.x *= COL_RESCALE; t.y *= COL_RESCALE;
        s_park[j][threadIdx.x] = t;
      }
    }
  }

  const unsigned off = (unsigned)s << 3;               // byte offset of this lane inside a row of a struct-of-arrays output
  if (a.defect) {
    char* drow = (char*)a.defect;
#pragma unroll
    for (int c = 0; c < 12; ++c) *(double*)(drow + (long)c * a.ldd * 8 + off) = y[c] - a.X[c * a.ldx + node + 1];
  }
  if (a.errors) a.errors[s] = 0.0;
  if (a.nacc) a.nacc[s] = steps;
  if (a.nrej) a.nrej[s] = 0;
  char* prow = (char*)a.Phi;
  const long row_bytes = a.ldp * 8;
#pragma unroll
  for (int col = 0; col < 12; ++col)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      double2 t;
      if (col < NREG) t = make_double2(phi[col][2 * q], phi[col][2 * q + 1]);
      else t = s_park[(col - NREG) * 6 + q][lane_opaque];
      *(double*)(prow + off) = t.x * a.stm_scale;
      prow += row_bytes;
      *(double*)(prow + off) = t.y * a.stm_scale;
      prow += row_bytes;
    }
}

template <int PM>
static hipError_t launch_lane_one(const IndirectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_indirect_lane<PM>), dim3((a.S + 63) / 64), dim3(64), 0, st, a);
  return hipGetLastError();
}

bool indirect_stm_lane_available(int ndim, int method, long S) { return ndim == 12 && method == M_RK4 && S < (1L << 29); }

hipError_t launch_indirect_stm_lane(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  if (a0.steps < 1 || !a0.Phi || a0.order) return hipErrorInvalidValue;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_lane_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_lane_one<PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_lane_one<PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_lane_one<PM_PGEN>(a, st);
  return e;
}

}  // namespace lto
