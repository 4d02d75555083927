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
//
// ND = 14 (round 6; BASELINE configs[1]'s system, 1 680 bytes of Phi and defect per segment): the same mapping with rhs14 / var_col14.
// Four VarCoef14 are 100 doubles and a column in flight 42, so the lane takes the whole register file of its SIMD (one wavefront per
// SIMD, 512 registers -- for the 12-dim form measured to cost nothing, see above); of a stage argument the coefficients need position,
// mass, lambda_v and lambda_m (8 doubles).
#include "kernels.hpp"
#include "rk.hpp"

namespace lto {

// rows of a stage argument that the variational coefficients depend on (rhs12 / rhs14 with VAR)
template <int ND> struct StreamKeep;
template <> struct StreamKeep<12> { static constexpr int N = 6;  static constexpr int idx[6] = {0, 1, 2, 9, 10, 11}; };
template <> struct StreamKeep<14> { static constexpr int N = 8;  static constexpr int idx[8] = {0, 1, 2, 6, 10, 11, 12, 13}; };

template <int ND, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_indirect_stream(const IndirectArgs a) {
  constexpr int PM = PM_ANY;                           // the law is chosen per trajectory at run time: one kernel per dimension (round 6)
  using Keep = StreamKeep<ND>;
  using Coef = typename std::conditional<ND == 12, VarCoef12, VarCoef14>::type;
  constexpr int NK = Keep::N;
  auto rhs = [](auto var, const double (&yy)[ND], const TrajParams& tp, double (&dy)[ND], Coef& vc) {
    constexpr bool VAR = decltype(var)::value;
    if constexpr (ND == 12) rhs12<PM, VAR>(yy, tp, dy, vc);
    else rhs14<PM, VAR>(yy, tp, dy, vc);
  };
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;

  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
  const double span = a.t[tg + 1] - a.t[tg];
  const double h = span / (double)a.steps;             // steps == 1 (the launcher checks)
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);

  // Base trajectory: one RK4 step.  Only what the variational coefficients depend on is kept of every stage argument -- its
  // position and lambda_v, 6 doubles (ND = 14: + mass and lambda_m) -- and the coefficients are built from those AFTER the step (the
  // right-hand side once more per stage with its slopes dead: the same instructions on the same operands, so the same bits as one
  // fused evaluation): with the coefficients of earlier stages live across the later evaluations the base phase needs ~280
  // registers (ND = 12), this way ~200.
  double arg[4][NK];
  {
    double k[ND], yt[ND], acc[ND];
    Coef none;
#pragma unroll
    for (int c = 0; c < NK; ++c) arg[0][c] = y[Keep::idx[c]];
    rhs(std::false_type{}, y, tp, k, none);
#pragma unroll
    for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < NK; ++c) arg[1][c] = yt[Keep::idx[c]];
    rhs(std::false_type{}, yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < NK; ++c) arg[2][c] = yt[Keep::idx[c]];
    rhs(std::false_type{}, yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
#pragma unroll
    for (int c = 0; c < NK; ++c) arg[3][c] = yt[Keep::idx[c]];
    rhs(std::false_type{}, yt, tp, k, none);
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
  }
  const unsigned off = (unsigned)s << 3;               // byte offset of this lane inside a row of a struct-of-arrays output (S < 2^29)
  if (a.defect) {
    // node i + 1 is read only now: its lines are (all but one) the lines node i came from, and twelve more live doubles across the
    // base stages would not fit two wavefronts per SIMD
    char* drow = (char*)a.defect;
#pragma unroll
    for (int c = 0; c < ND; ++c) __builtin_nontemporal_store(y[c] - a.X[c * a.ldx + node + 1], (double*)(drow + (long)c * a.ldd * 8 + off));
  }
  if (a.errors) a.errors[s] = 0.0;
  if (a.nacc) a.nacc[s] = a.steps;
  if (a.nrej) a.nrej[s] = 0;

  // the four stage matrices
  Coef vc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double ya[ND], dead[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) ya[c] = 0.0;
#pragma unroll
    for (int c = 0; c < NK; ++c) asm volatile("" : "+v"(arg[j][c]));   // opaque: or the compiler merges this evaluation back into the step's
#pragma unroll
    for (int c = 0; c < NK; ++c) ya[Keep::idx[c]] = arg[j][c];
    rhs(std::true_type{}, ya, tp, dead, vc[j]);
  }
  // the columns, one after the other, through them
  char* prow = (char*)a.Phi;                            // row (col * ND + r) of Phi starts at prow: uniform, advanced by scalar adds
  const long row_bytes = a.ldp * 8;
  auto var_col = [&](const Coef& v, const double (&c)[ND], double (&dc)[ND]) {
    if constexpr (ND == 12) var_col12(v, w2, c, dc);
    else var_col14(v, w2, c, dc);
  };
  for (int col = 0; col < ND; ++col) {
    double c0[ND], ct[ND], dc[ND], acc[ND];
#pragma unroll
    for (int r = 0; r < ND; ++r) c0[r] = (r == col) ? 1.0 : 0.0;
    var_col(vc[0], c0, dc);
#pragma unroll
    for (int r = 0; r < ND; ++r) { acc[r] = __builtin_fma(h6, dc[r], c0[r]); ct[r] = __builtin_fma(h2, dc[r], c0[r]); }
    var_col(vc[1], ct, dc);
#pragma unroll
    for (int r = 0; r < ND; ++r) { acc[r] = __builtin_fma(h3, dc[r], acc[r]); ct[r] = __builtin_fma(h2, dc[r], c0[r]); }
    var_col(vc[2], ct, dc);
#pragma unroll
    for (int r = 0; r < ND; ++r) { acc[r] = __builtin_fma(h3, dc[r], acc[r]); ct[r] = __builtin_fma(h, dc[r], c0[r]); }
    var_col(vc[3], ct, dc);
#pragma unroll
    for (int r = 0; r < ND; ++r) {
      __builtin_nontemporal_store(__builtin_fma(h6, dc[r], acc[r]), (double*)(prow + off));
      prow += row_bytes;
    }
  }
}

bool indirect_stm_stream_available(int ndim, int method, int steps, long S) {
  return (ndim == 12 || ndim == 14) && method == M_RK4 && steps == 1 && S < (1L << 29);
}

// One launch whatever the batch's control laws (PM_ANY): mixed-class batches used to be one launch per class.
hipError_t launch_indirect_stm_stream(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  (void)pm;
  if (a.S <= 0) return hipSuccess;
  if (a.steps != 1 || !a.Phi || (ndim != 12 && ndim != 14)) return hipErrorInvalidValue;
  const dim3 grid((a.S + 63) / 64);
  if (ndim == 12) hipLaunchKernelGGL((k_indirect_stream<12, 2>), grid, dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_indirect_stream<14, 1>), grid, dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace lto
