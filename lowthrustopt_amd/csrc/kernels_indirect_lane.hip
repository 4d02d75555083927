// kernels_indirect_lane.hip -- RK4 STM sweep for batches that fill the chip many times over: lane = ONE WHOLE SEGMENT.
//
// Replaces jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) with fixed-step RK4 at BASELINE configs[3] sizes (256 levels x
// 1 024 segments).  Once every SIMD has wavefronts to spare, only the number of instructions issued per segment counts.  The
// three-role pipelines (kernels_indirect_pipe48.hip) spend 118 wave-instructions per segment and RK4 step: a DPP row of 16 lanes
// carries the 12 columns of a segment (a quarter of the column lanes idle), the roles meet at a barrier per step, coefficients
// travel through LDS.  Here a lane owns its segment outright:
//
//   per RK4 step   base trajectory, four stage evaluations (rhs12_base_parts: the pipelines' base role, same operations, same bits,
//                     keeping of every evaluation the by-products the variational coefficients are made of)
//                  -> the four stage matrices G, H, U from the stage arguments' position and lambda_v and those by-products
//                     (coef12_from_parts: no second set of reciprocal square roots, no second control law), scaled by the stage
//                     weights, 68 doubles in addressable registers
//                  -> the twelve STM columns one after the other through those four matrices (col_reg_step: col_dpp_step of
//                     pipe_common.hpp with the coefficient in a register instead of a DPP broadcast -- the same FMAs in the same order).
//   No barrier, no idle lane, nothing computed twice: 61 wave-instructions per segment and step (3 913 per wavefront).
//   The defect equals the pipelines' bit for bit (same base arithmetic); Phi agrees with theirs to round-off (~1e-15 of max |Phi|):
//   the pipelines' coefficient role re-evaluates the control law with another exponential routine, this kernel reuses the base
//   evaluation's (round 5's form, which re-evaluated like the pipelines, was bit-identical and 11 % slower).
//
// The price is state: Phi is 144 doubles per lane, next to 68 doubles of stage matrices and four 12-vectors of a column in flight.
// The kernel runs ONE wavefront per SIMD (512 registers per lane: 256 addressable + 256 accumulation registers) and says where
// everything lives (round 6; before, the allocator chose, parked matrices as well as Phi and spilled to scratch inside the loop):
// EIGHT columns of Phi in the accumulation registers by inline asm (192 of 256), moved in and out around their own RK4 step; the base
// state parked there too during the column phase (24); four columns in LDS (24 KB per wavefront, four wavefronts per CU; 16-byte
// pieces, one per lane and access).  490 registers, no scratch, no spill in any control-law class; nine columns would spill
// (LANE_COLS_IN_REGISTERS).  It needs 64 segments per SIMD to fill the chip: AUTO compares its rounds of 256 x CUs segments with the
// pipelines' (lto_api.hip).  The stage arguments pass through an empty asm before the matrices are built from them: otherwise the
// compiler keeps ~20 more by-products per stage alive across the step.
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

#ifndef LANE_NREG
#define LANE_NREG 8
#endif
constexpr int LANE_COLS_IN_REGISTERS = LANE_NREG;

// Explicit parking (round 6; LANE_EXPLICIT_PARK, default 3 -- the lower levels are kept for A/B builds, profiles/r06_probe_lane_parking.txt):
//   1  the register columns of Phi live in the ACCUMULATION registers by inline asm with "a" constraints and pass through the
//      addressable registers only for their own RK4 step (24 reads before it, 24 writes behind it);
//   2  the base state is parked the same way across the column phase (without that the allocator keeps it addressable through the
//      phase and spills ~100 registers elsewhere);
//   3  the four stage matrices are built from the by-products of the base evaluations (rhs12_base_parts -> coef12_from_parts: no
//      second set of reciprocal square roots and exponentials per step, 270 instructions fewer) -- affordable only now that nothing
//      spills: 28 more doubles live across the base phase.
// Left to itself (level 0) the allocator parks whatever it happens to choose -- stage matrices and work values as well as Phi -- and
// spills four doubles to scratch inside the loop; a lone wavefront per SIMD cannot hide those round trips.  C4: 2.41 -> 2.16 ms on the
// same box with eight columns in registers (level 2: 2.31; level 3 with six columns: 2.19).
#ifndef LANE_EXPLICIT_PARK
#define LANE_EXPLICIT_PARK 3
#endif
struct AccDouble { int lo, hi; };
__device__ __forceinline__ void acc_put(AccDouble& a, const double v) {
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"(__double2loint(v)));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"(__double2hiint(v)));
}
__device__ __forceinline__ double acc_get(const AccDouble& a) {
  int lo, hi;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
  return __hiloint2double(hi, lo);
}

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
  // Phi: columns 0 .. NREG-1 in the accumulation registers (PHI_PUT / PHI_GET), the others in LDS, one 16-byte piece per lane and
  // access (conflict-free), read back through an offset the compiler cannot see through (it would otherwise forward the stored
  // registers across the loop and keep them alive)
  constexpr int NREG = LANE_COLS_IN_REGISTERS, NLDS = 12 - NREG;
  __shared__ double2 s_park[(NLDS > 0 ? NLDS : 1) * 6][64];
  int lane_opaque = threadIdx.x;
  asm volatile("" : "+v"(lane_opaque));
#if LANE_EXPLICIT_PARK
  AccDouble phi[NREG > 0 ? NREG : 1][12];
#define PHI_PUT(col, r, v) acc_put(phi[col][r], (v))
#define PHI_GET(col, r) acc_get(phi[col][r])
#else
  double phi[NREG > 0 ? NREG : 1][12];
#define PHI_PUT(col, r, v) (phi[col][r] = (v))
#define PHI_GET(col, r) (phi[col][r])
#endif
#pragma unroll
  for (int col = 0; col < 12; ++col)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double e0 = (2 * q == col) ? 1.0 : 0.0, e1 = (2 * q + 1 == col) ? 1.0 : 0.0;
      if (col < NREG) { PHI_PUT(col, 2 * q, e0); PHI_PUT(col, 2 * q + 1, e1); }
      else s_park[(col - NREG) * 6 + q][threadIdx.x] = make_double2(e0, e1);
    }

  for (int step = 0; step < steps; ++step) {
    // base trajectory (the pipelines' base role): one RK4 step; of every stage argument the position and lambda_v are kept
    double arg[4][6];
#if LANE_EXPLICIT_PARK >= 3
    BaseParts12 bparts[4];
#endif
    {
      double k[12], yt[12], acc[12];
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[0][c] = y[c]; arg[0][3 + c] = y[9 + c]; }
#if LANE_EXPLICIT_PARK >= 3
      rhs12_base_parts<PM>(y, tp, k, bparts[0]);
#else
      rhs12_base<PM>(y, tp, k);
#endif
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[1][c] = yt[c]; arg[1][3 + c] = yt[9 + c]; }
#if LANE_EXPLICIT_PARK >= 3
      rhs12_base_parts<PM>(yt, tp, k, bparts[1]);
#else
      rhs12_base<PM>(yt, tp, k);
#endif
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[2][c] = yt[c]; arg[2][3 + c] = yt[9 + c]; }
#if LANE_EXPLICIT_PARK >= 3
      rhs12_base_parts<PM>(yt, tp, k, bparts[2]);
#else
      rhs12_base<PM>(yt, tp, k);
#endif
#pragma unroll
      for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
#pragma unroll
      for (int c = 0; c < 3; ++c) { arg[3][c] = yt[c]; arg[3][3 + c] = yt[9 + c]; }
#if LANE_EXPLICIT_PARK >= 3
      rhs12_base_parts<PM>(yt, tp, k, bparts[3]);
#else
      rhs12_base<PM>(yt, tp, k);
#endif
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
#if LANE_EXPLICIT_PARK >= 3
      coef12_from_parts(ya[0], ya[1], ya[2], ya[9], ya[10], ya[11], bparts[j], tp.MU, vc[j]);
      (void)dead;
#else
      rhs12<PM, true>(ya, tp, dead, vc[j]);
#endif
      const double as = (j == 2) ? h : h2;
      double* o = reinterpret_cast<double*>(&vc[j]);
#pragma unroll
      for (int e = 0; e < 14; ++e) o[e] = o[e] * as;
    }
#ifndef LANE_PARK_BASE
#define LANE_PARK_BASE (LANE_EXPLICIT_PARK >= 2)
#endif
#if LANE_PARK_BASE
    AccDouble ypark[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) acc_put(ypark[c], y[c]);
#endif
#pragma unroll
    for (int col = 0; col < 12; ++col) {
      __builtin_amdgcn_sched_barrier(0);           // one column at a time: interleaving two of them costs more registers than there are
#if LANE_EXPLICIT_PARK
      if (col < NREG) {
        double u[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) u[c] = PHI_GET(col, c);
        col_reg_step(vc, kc, u);
#pragma unroll
        for (int c = 0; c < 12; ++c) PHI_PUT(col, c, u[c]);
      }
#else
      if (col < NREG) col_reg_step(vc, kc, phi[col]);
#endif
      else {
        double u[12];
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double2 t = s_park[(col - NREG) * 6 + q][lane_opaque]; u[2 * q] = t.x; u[2 * q + 1] = t.y; }
        col_reg_step(vc, kc, u);
#pragma unroll
        for (int q = 0; q < 6; ++q) s_park[(col - NREG) * 6 + q][threadIdx.x] = make_double2(u[2 * q], u[2 * q + 1]);
      }
    }
#if LANE_PARK_BASE
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 12; ++c) y[c] = acc_get(ypark[c]);
#endif
    if (((step + 1) & (COL_RESCALE_EVERY - 1)) == 0) {       // every 256 steps: u <- 3^-256 u, all columns (pipe_common.hpp COL_RESCALE)
#pragma unroll
      for (int col = 0; col < NREG; ++col)
#pragma unroll
        for (int c = 0; c < 12; ++c) PHI_PUT(col, c, PHI_GET(col, c) * COL_RESCALE);
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
      if (col < NREG) t = make_double2(PHI_GET(col, 2 * q), PHI_GET(col, 2 * q + 1));
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
