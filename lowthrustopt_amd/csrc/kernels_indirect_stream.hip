// kernels_indirect_stream.hip -- the HBM-bound corner of the indirect STM sweep: ONE RK4 step per segment, full 12x12 STM out.
//
// Replaces jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) for plans with RK4 x 1 (the fine-mesh limit: SURVEY 8d's "HBM
// evidence point", arithmetic intensity 4.4 flop/B against a machine balance of 9.8).  What bounds it is the 1 248 bytes per
// segment that must leave the chip (Phi 1 152 + defect 96), so the mapping is chosen for bytes and instruction count, not for
// parallel slack:
//
//   lane = ONE WHOLE SEGMENT.  The base trajectory's four stage evaluations run once per segment and leave the variational
//   coefficients of all four stages in registers (4 x VarCoef12 = 68 doubles); the twelve STM columns are then advanced one after
//   the other through those four matrices (36 doubles of column state at a time) and stored as they finish.  Nothing is computed
//   twice: the per-(segment, column-group) lanes of k_indirect<12,PM,RK4,3> re-run the base stages and the coefficient build in
//   every one of their four lanes (150 M wave-instructions per 1 048 576 segments; this form ~45 M), and their four column groups
//   are four workgroups on four XCDs, each reading both nodes from HBM again (profiles/r04z_hbm_ndim12_pmc.json: 0.64 GB fetched
//   for 0.11 GB of nodes).
//
//   HBM: every load / store instruction of a wavefront moves 512 contiguous bytes (struct-of-arrays, segment index fastest): node
//   i and node i + 1 come from the same cache lines, each line of Phi and of the defect is written exactly once, whole.  Stores
//   are fire-and-forget (nothing waits for them) and NONTEMPORAL: 1.3 GB of Phi per million segments that nobody re-reads would
//   otherwise be written into the L2 and evicted from it line by line (0.310 -> 0.254 ms per 1 048 576 segments, 4.9 -> 6.0 TB/s
//   algorithmic; profiles/r05_probe_stream.txt).  Two wavefronts per SIMD (<= 256 VGPRs) overlap one segment block's loads with
//   the other's arithmetic (one per SIMD with 512 registers: the same time).
//
// A column starts at the unit vector only because steps == 1; plans with 2 ... 5 RK4 steps keep the per-column-group kernel.
// Arithmetic per entry = rhs12<PM, true> / var_col12 / the RK4 update of rk4_step, in the same order as k_indirect<12,PM,RK4,COLS>.
#include "kernels.hpp"
#include "rk.hpp"

namespace lto {

template <int PM, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_indirect_stream(const IndirectArgs a) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(tp.p) != PM) return;   // mixed-class batch: another launch owns this trajectory
  const double w2 = 2.0 * tp.omega;

  double y[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) y[c] = a.X[c * a.ldx + node];
  const double span = a.t[tg + 1] - a.t[tg];
  const double h = span / (double)a.steps;             // steps == 1 (the launcher checks)
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);

  // Base trajectory: one RK4 step.  Only what the variational coefficients depend on is kept of every stage argument -- its
  // position and lambda_v, 6 doubles -- and the coefficients are built from those AFTER the step (rhs12<PM, true> once more per
  // stage with its slopes dead: the same instructions on the same operands, so the same bits as one fused evaluation): with the
  // coefficients of earlier stages live across the later evaluations the base phase needs ~280 registers, this way ~200.
  double arg[4][6];
  {
    double k[12], yt[12], acc[12];
    VarCoef12 none;
#pragma unroll
    for (int c = 0; c < 3; ++c) { arg[0][c] = y[c]; arg[0][3 + c] = y[9 + c]; }
    rhs12<PM, false>(y, tp, k, none);
#pragma unroll
    for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { arg[1][c] = yt[c]; arg[1][3 + c] = yt[9 + c]; }
    rhs12<PM, false>(yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { arg[2][c] = yt[c]; arg[2][3 + c] = yt[9 + c]; }
    rhs12<PM, false>(yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < 12; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) { arg[3][c] = yt[c]; arg[3][3 + c] = yt[9 + c]; }
    rhs12<PM, false>(yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < 12; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
  }
  const unsigned off = (unsigned)s << 3;               // byte offset of this lane inside a row of a struct-of-arrays output (S < 2^29)
  if (a.defect) {
    // node i + 1 is read only now: its lines are (all but one) the lines node i came from, and twelve more live doubles across the
    // base stages would not fit two wavefronts per SIMD
    char* drow = (char*)a.defect;
#pragma unroll
    for (int c = 0; c < 12; ++c) __builtin_nontemporal_store(y[c] - a.X[c * a.ldx + node + 1], (double*)(drow + (long)c * a.ldd * 8 + off));
  }
  if (a.errors) a.errors[s] = 0.0;
  if (a.nacc) a.nacc[s] = a.steps;
  if (a.nrej) a.nrej[s] = 0;

  // the four stage matrices
  VarCoef12 vc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double ya[12], dead[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) ya[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(arg[j][c]));   // opaque: or the compiler merges this evaluation back into the step's
#pragma unroll
    for (int c = 0; c < 3; ++c) { ya[c] = arg[j][c]; ya[9 + c] = arg[j][3 + c]; }
    rhs12<PM, true>(ya, tp, dead, vc[j]);
  }
  // the twelve columns, one after the other, through them
  char* prow = (char*)a.Phi;                            // row (col * 12 + r) of Phi starts at prow: uniform, advanced by scalar adds
  const long row_bytes = a.ldp * 8;
  for (int col = 0; col < 12; ++col) {
    double c0[12], ct[12], dc[12], acc[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) c0[r] = (r == col) ? 1.0 : 0.0;
    var_col12(vc[0], w2, c0, dc);
#pragma unroll
    for (int r = 0; r < 12; ++r) { acc[r] = __builtin_fma(h6, dc[r], c0[r]); ct[r] = __builtin_fma(h2, dc[r], c0[r]); }
    var_col12(vc[1], w2, ct, dc);
#pragma unroll
    for (int r = 0; r < 12; ++r) { acc[r] = __builtin_fma(h3, dc[r], acc[r]); ct[r] = __builtin_fma(h2, dc[r], c0[r]); }
    var_col12(vc[2], w2, ct, dc);
#pragma unroll
    for (int r = 0; r < 12; ++r) { acc[r] = __builtin_fma(h3, dc[r], acc[r]); ct[r] = __builtin_fma(h, dc[r], c0[r]); }
    var_col12(vc[3], w2, ct, dc);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      __builtin_nontemporal_store(__builtin_fma(h6, dc[r], acc[r]), (double*)(prow + off));
      prow += row_bytes;
    }
  }
}

template <int PM>
static hipError_t launch_stream_one(const IndirectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_indirect_stream<PM, 2>), dim3((a.S + 63) / 64), dim3(64), 0, st, a);
  return hipGetLastError();
}

bool indirect_stm_stream_available(int ndim, int method, int steps, long S) {
  return ndim == 12 && method == M_RK4 && steps == 1 && S < (1L << 29);
}

hipError_t launch_indirect_stm_stream(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  if (a0.steps != 1 || !a0.Phi) return hipErrorInvalidValue;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_stream_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_stream_one<PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_stream_one<PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_stream_one<PM_PGEN>(a, st);
  return e;
}

}  // namespace lto
