// kernels.hpp -- argument blocks and launch entry points shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "dynamics.hpp"

namespace lto {

// integrator ids (== LTO_* in include/lto.h)
enum Method : int { M_RK4 = 0, M_RKF78_FIXED = 1, M_RKF78_ADAPTIVE = 2, M_DOP853_ADAPTIVE = 3 };

// Device layout (include/lto.h, "device-resident API"): node j = b*n_nodes + k, segment
// s = b*seg_per_traj + i, component-major with leading dimensions ld*.
struct IndirectArgs {
  const double* X; long ldx;       // [ndim][ldx] nodes (ndim = 12 or 14)
  const double* t; int t_stride;   // t[b*t_stride + k]; t_stride = n_nodes (per-trajectory grids) or 0 (shared)
  const TrajParams* tp; int tp_stride;  // tp[b*tp_stride]; 1 or 0
  int n_nodes, seg_per_traj, S;
  int steps;                       // fixed-step methods
  double rtol, atol; int max_steps;  // adaptive methods
  double* defect; long ldd;        // [12][ldd] or null
  double* errors;                  // [S] or null
  double* Phi; long ldp;           // [144][ldp] (col*12+row) or null
  int* nacc; int* nrej;            // [S] adaptive step counters or null
  const int* order;                // [S] or null: lane -> segment map of adaptive sweeps (lto_indirect_plan_rebalance)
  int xcd_ranges;                  // 1: `order` is the windowed kind (LTO_ORDER_WINDOW below): workgroup b works on unit xcd_unit(b, grid), not b
  int class_filter;                // set by the launchers: 1 = this launch handles only trajectories of the kernel's p-class
  double stm_scale;                // 3^-(steps mod 256): the DPP column lanes of the pipeline kernels carry 3^k Phi (pipe_common.hpp)
  double* h_first;                 // [S] or null: step size of the segment's first ACCEPTED trial step (written by the two-lane adaptive kernels)
  int warm;                        // 1: start every segment from h_first (the previous sweep of this kind) instead of Hairer's rule
  // Record staging of rebalanced sweeps (round 4; kernels_indirect_coop2.hip, kernels_indirect_defect2.hip): with a balanced lane
  // order neighbouring lanes hold unrelated segments, and every element of the struct-of-arrays operands is an 8-byte access to
  // its own cache line (5 - 8 x the algorithmic traffic).  When these are set the kernel reads a node as ONE record
  //   Xa[node * NODE_REC + c], c < 12;  Xa[node * NODE_REC + 12] = the node's time
  // and writes a segment's results as records  Da[s * 12 + c],  Pa[s * 144 + col * 12 + row];  the plan converts between the
  // caller's arrays and the records with coalesced transposes before / after the sweep (lto_api.hip).  Null: the arrays above.
  const double* Xa;
  double* Da;
  double* Pa;
};
constexpr int NODE_REC = 16;       // doubles per node record: 12 components, the time, padding to 128 bytes
// operand access of the kernels that support record staging
__device__ __forceinline__ double arg_node(const IndirectArgs& a, const int c, const long node) {
  return a.Xa ? a.Xa[node * NODE_REC + c] : a.X[c * a.ldx + node];
}
__device__ __forceinline__ double arg_span(const IndirectArgs& a, const long node, const long tg) {
  return a.Xa ? a.Xa[(node + 1) * NODE_REC + 12] - a.Xa[node * NODE_REC + 12] : a.t[tg + 1] - a.t[tg];
}
__device__ __forceinline__ void put_defect(const IndirectArgs& a, const int c, const int s, const double v) {
  if (a.Da) a.Da[(long)s * 12 + c] = v; else a.defect[c * a.ldd + s] = v;
}
__device__ __forceinline__ void put_phi(const IndirectArgs& a, const int col, const int row, const int s, const double v) {
  if (a.Pa) a.Pa[(long)s * 144 + col * 12 + row] = v; else a.Phi[(long)(col * 12 + row) * a.ldp + s] = v;
}

struct DirectArgs {
  const double* X; long ldx;       // [nstate][ldx]
  const double* U; long ldu;       // [3][ldu]  thrust, N
  const double* t; int t_stride;
  double MU, kk, isp_g0, TU;       // kk = TU^2/DU/1e3; isp_g0 = Isp * 9.81
  int n_nodes, seg_per_traj, S;
  int half_steps;                  // nsteps - 1 RKF7(8) steps per half segment
  double* defect; long ldd;        // [nstate][ldd] or null
  double* errors;                  // [S] or null
  double* Jac; long ldj;           // [nstate*nvar][ldj] (col*nstate+row) or null
  double* dtf;                     // [nstate][ldd] or null
  double* mid; long ldm;           // [nstate][ldm] or null: forward half-arc end state x(t_i + h_i/2; x_i, u_i)
};

static inline bool single_class(int pm) { return (pm & (pm - 1)) == 0; }   // pm: bit mask of p-classes

// Launchers return hipSuccess or the launch error.  `pm` is a bit mask of the PMode classes present in the batch (bit c = class c), `method` a Method.
hipError_t launch_indirect_defect(int pm, int method, const IndirectArgs& a, hipStream_t st);
// cols_per_lane in {1,2,3}; 0 = choose from S.
hipError_t launch_indirect_stm(int pm, int method, int cols_per_lane, const IndirectArgs& a, hipStream_t st);
// dense output: segment s is sampled at td[first[s] .. first[s+1]); Y is SoA [ndim][ldy]
struct DenseArgs {
  const int* first;        // [S+1] prefix offsets into td / columns of Y
  const double* td;        // [n_samples] sample times
  double* Y; long ldy;     // [ND][ldy]
  double* final_state;     // [ND][n_batch] or null: x(t_n) of every trajectory
};
hipError_t launch_indirect_dense(int ndim, int pm, int method, const IndirectArgs& a, const DenseArgs& d, hipStream_t st);
hipError_t launch_indirect14_defect(int pm, int method, const IndirectArgs& a, hipStream_t st);
hipError_t launch_indirect14_stm(int pm, int method, int cols_per_lane, const IndirectArgs& a, hipStream_t st);
// wave-specialised STM kernel (kernels_indirect_coop.hip): base wave + column waves per 16 segments
hipError_t launch_indirect_stm_coop(int ndim, int pm, int method, const IndirectArgs& a, hipStream_t st);
// the same with every 12-component state split over two lanes (kernels_indirect_coop2.hip): 12-dim, DOP853 adaptive only
hipError_t launch_indirect_stm_coop2(int pm, const IndirectArgs& a, hipStream_t st);
// ... and its 14-dim form (kernels_indirect_coop2_14.hip): states split 7 + 7, thirteen columns; the always-thrust-limited laws (p = 0, 1) only
hipError_t launch_indirect_stm_coop2_14(int pm, const IndirectArgs& a, hipStream_t st);
bool indirect_stm_coop2_14_available(int pm);
// defect-only sweep with two lanes per segment (kernels_indirect_defect2.hip): 12-dim, DOP853 adaptive only
hipError_t launch_indirect_defect2(int pm, const IndirectArgs& a, hipStream_t st);
// ... with four lanes per segment (same file): while the chip has a SIMD per 16 segments to spare
hipError_t launch_indirect_defect4(int pm, const IndirectArgs& a, hipStream_t st);
// ... on the 14-dim system (same file; p = 0 / 1 batches, see indirect_stm_coop2_14_available)
hipError_t launch_indirect14_defect4(int pm, const IndirectArgs& a, hipStream_t st);
// three-role pipeline, fixed-step RK4 only: base wave, coefficient wave and column waves per 16 segments, skewed by one RK4 step.
// Eight-wave form (kernels_indirect_pipe8.hip): one STM column per lane with the coefficients broadcast inside the FMA (v_fmac_f64_dpp
// row_newbcast), two RK4 steps per phase, a fourth of the column work alternates between two SIMDs, base role with paired stages
hipError_t launch_indirect_stm_pipe8(int ndim, int pm, const IndirectArgs& a, hipStream_t st);
// large batches (kernels_indirect_pipe48.hip): 48 segments and 16 waves per workgroup, base lane = segment, DPP column rows
hipError_t launch_indirect_stm_pipe48(int ndim, int pm, const IndirectArgs& a, bool seg44, hipStream_t st);
hipError_t launch_indirect_stm_pipe32(int ndim, int pm, const IndirectArgs& a, hipStream_t st);   // kernels_indirect_pipe32.hip
// one RK4 step, lane = whole segment with all twelve STM columns (kernels_indirect_stream.hip): the HBM-bound corner of the sweep
hipError_t launch_indirect_stm_stream(int ndim, int pm, const IndirectArgs& a, hipStream_t st);
bool indirect_stm_stream_available(int ndim, int method, int steps, long S);
// RK4, any number of steps, lane = whole segment with the full STM (kernels_indirect_lane.hip): batches that fill the chip many times over
hipError_t launch_indirect_stm_lane(int pm, const IndirectArgs& a, hipStream_t st);
bool indirect_stm_lane_available(int ndim, int method, long S);
bool indirect_stm_pipe32_available(int ndim, int pm);
hipError_t launch_direct_defect(int nstate, const DirectArgs& a, hipStream_t st);
hipError_t launch_direct_jacobian(int nstate, const DirectArgs& a, hipStream_t st);
// base wave + one wave per sensitivity column for 32 segments, skewed by one RKF7(8) step (one barrier per step)
hipError_t launch_direct_jacobian_pipe(int nstate, const DirectArgs& a, hipStream_t st);

// Newton step of the indirect method on the device (kernels_bvp.hip): structured orthogonal cyclic reduction.
size_t bvp_workspace_doubles(int n_nodes, int n_batch);
hipError_t launch_bvp_solve(const double* Phi, long ldp, const double* defect, long ldd, int n_nodes, int n_batch,
                            int adjoints_only, double* workspace, double* delta, long ldx, hipStream_t st);
hipError_t launch_axpy(const double* x, const double* d, double alpha, double* y, long count, hipStream_t st);

hipError_t launch_pack_soa(const double* aos, int ndim, long count, double* soa, long ld, hipStream_t st);
hipError_t launch_unpack_soa(const double* soa, long ld, int ndim, long count, double* aos, hipStream_t st);
// two pack / unpack jobs in one launch
hipError_t launch_pack_soa2(const double* aos_a, int ndim_a, long count_a, double* soa_a, long ld_a, const double* aos_b, int ndim_b,
                            long count_b, double* soa_b, long ld_b, hipStream_t st);
hipError_t launch_unpack_soa2(const double* soa_a, long ld_a, int ndim_a, long count_a, double* aos_a, const double* soa_b, long ld_b,
                              int ndim_b, long count_b, double* aos_b, hipStream_t st);
constexpr int LTO_ORDER_BINS = 1024;   // int workspace launch_segment_order needs
// Windowed lane order (round 5): segments are ordered by step count INSIDE windows of LTO_ORDER_WINDOW consecutive segments, the
// windows by their slowest segment, heaviest first, dealt to the eight XCDs in turn (each XCD's windows contiguous in the order).
// With the sweep kernels' workgroups mapped to contiguous ranges per XCD (xcd_unit below), the wavefronts that share a window run
// on ONE XCD at about the same time: its L2 sees every line of the window's nodes, defects and Phi whole, so the sweep reads and
// writes the caller's struct-of-arrays operands directly -- no record passes (kernels.hpp IndirectArgs::Xa / Da / Pa).
constexpr int LTO_ORDER_WINDOW = 1024;       // largest window (one thread per segment in k_order_window)
constexpr int LTO_XCDS = 8;
// Window size for a batch of S segments: as close to 1 024 as gives every XCD the same number of windows -- 8 k windows for
// k = ceil(S / 8 192), a multiple of 16 segments (the interleave unit) -- so that the XCDs' lists line up with the contiguous
// eighths of the order the kernels' workgroups are mapped to (xcd_unit below).  S = 65 536, 262 144, 20 x 4 096: 1 024.
inline int order_window_size(long S) {
  const long k = (S + 8191) / 8192 > 0 ? (S + 8191) / 8192 : 1;
  long w = (S + 8 * k - 1) / (8 * k);
  w = (w + 15) / 16 * 16;
  return (int)(w < 16 ? 16 : (w > LTO_ORDER_WINDOW ? LTO_ORDER_WINDOW : w));
}
inline long order_windows(long S) { const long w = order_window_size(S); return (S + w - 1) / w; }
// ints of workspace behind the S entries of an order array: window-local order [S], per window (key, destination, slot) [3 nwin], bins
inline size_t order_workspace_ints(long S) { return (size_t)S + 3 * (size_t)order_windows(S) + LTO_ORDER_BINS + 64; }
inline size_t order_bytes(long S) { return sizeof(int) * ((size_t)S + order_workspace_ints(S)); }
hipError_t launch_segment_order_windowed(const int* nacc, const int* nrej, int S, int weave, int* work, int* order, hipStream_t st);
// Workgroup b of a grid of nb -> the unit it works on, such that the workgroups an XCD receives (round-robin dispatch: b mod 8) own a
// CONTIGUOUS range of units: the nodes two neighbouring units share, and the windows of an ordered sweep, then sit in one L2.
// A bijection of [0, nb) for every nb.
__device__ __forceinline__ int xcd_unit_of(const int b, const int nb) {
  const int x = b % LTO_XCDS, j = b / LTO_XCDS;
  const int chunk = nb / LTO_XCDS, rem = nb % LTO_XCDS;
  return x * chunk + (x < rem ? x : rem) + j;
}
// Only the windowed order wants the ranges: in natural order (slow segments cluster along a trajectory) and in the global order
// (heaviest first) a contiguous eighth per XCD would be an eighth of very different weight.
#define xcd_unit(a, b, nb) ((a).xcd_ranges ? xcd_unit_of((b), (nb)) : (b))
// node records for the staged sweeps: Xa[j][0..11] = X[c][j], Xa[j][12] = the node's time (t[b * t_stride + k], j = b n_nodes + k)
hipError_t launch_node_records(const double* X, long ldx, const double* t, int t_stride, int n_nodes, long J, double* Xa, hipStream_t st);
hipError_t launch_step_stats(const int* nacc, const int* nrej, int S, unsigned long long* acc, long long* host_out, hipStream_t st);
hipError_t launch_segment_order(const int* nacc, const int* nrej, int S, int* bins, int* order, hipStream_t st);
hipError_t launch_trial_points(const double* X, const double* d, long ld, int ndim, int n, int nb, int na, const double* alphas,
                               double* Xt, long ldt, hipStream_t st);
hipError_t launch_axpy_traj(const double* x, const double* d, const double* alpha, double* y, long ld, int ndim, int n, int nb,
                            hipStream_t st);
hipError_t launch_soc_mask(const double* mx, const double* act, double thr, double* step, int nb, hipStream_t st);
hipError_t launch_pick_alpha(const double* ss, const double* alphas, int na, const double* act, const double* search, double* step, int nb,
                             const double* mxt, double* mx, hipStream_t st);
hipError_t launch_take_trial(const double* trial, long ldt, const double* ss, const double* act, const double* search, int na, int seg,
                             int ndim, int nb, double* defect, long ldd, const double* alphas, double* step, const double* mxt, double* mx,
                             hipStream_t st);   // step != nullptr: also k_pick_alpha's outputs (one launch for both)
hipError_t launch_iter_report(const double* a, int na, const double* b, int nb, double* host_dev, long long* seq_dev, long long seq, hipStream_t st);
hipError_t launch_end_states(double* X, long ld, int n, int nb, int nrow, double* saved, int restore, hipStream_t st);
hipError_t launch_defect_norms(const double* defect, long ldd, int ndim, int seg_per_traj, int n_batch, double* sumsq,
                               double* maxabs, hipStream_t st);

}  // namespace lto
