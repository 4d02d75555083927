// kernels_bvp.hip -- the Newton step of indirect multiple shooting on the device (SURVEY "next" row N1).
//
// The reference solves  Jac_full * dx = -defect  with a sparse QR on the host (`-Jac_sparse \ defect_vec`,
// src/multiShoot_CRTBP_indirect.jl:181-182), where row block i of Jac_full is [Phi_i | -I] (:123) and the columns
// of the two fixed end states are zeroed (:141-142).  That is a block-bidiagonal two-point boundary value system:
//       A_i d_i + B_i d_{i+1} = r_i,   A_i = Phi_i, B_i = -I, r_i = -defect_i,   i = 0 .. S-1.
// It is solved here by STRUCTURED ORTHOGONAL cyclic reduction: at every level adjacent block rows (2j, 2j+1) are
// stacked and the 24 x 12 column block of their shared unknown [B_2j; A_2j+1] is triangularised by Householder
// reflections; the top 12 rows define that unknown (kept for back-substitution), the bottom 12 rows are a new
// block row coupling the two outer unknowns.  log2(S) levels of independent 24 x 37 problems, one wavefront each
// (lane = matrix column, 24 doubles in registers, reflector broadcast by v_readlane).  Only orthogonal
// transformations are used, so -- unlike condensing (products of STMs) -- the elimination is backward stable for
// long, unstable trajectories.  The reflectors are stored, so a second right-hand side (the second-order-correction
// re-solve, :190-214) costs one cheap pass.
//
// Node eliminated at level l by pair j:  mid = (2j+1) 2^l,  left = 2j 2^l,  right = min((2j+2) 2^l, n-1).
#include "kernels.hpp"

namespace lto {

// Two variants (template parameter NU = free unknowns per node):
//   NU = 12  the square system of the regular iterations (all 12 components of the interior nodes free, the two end
//            states fixed): 24 x 36 stacks, 12 reflections eliminate the shared unknown, the other 12 rows ARE the new
//            block row.
//   NU = 6   flag_adjointsOnly (indirect.jl:169-178): only the costates are free, A_i = Phi_i[:, 7:12], B_i = -I[:, 7:12];
//            the system is over-determined (12(n-1) equations, 6n unknowns) and is solved in the least-squares sense,
//            as `\` does: the 24 x 18 stack is triangularised completely (18 reflections); rows 0-5 define the shared
//            unknown, rows 6-17 are the new (upper-trapezoidal) block row, rows 18-23 carry only residual and drop out.
template <int NU> struct BvpDims {
  static constexpr int NK = (NU == 12) ? 12 : 18;            // Householder reflections per pair
  static constexpr int NCOLS = 3 * NU + 1;                   // mid | left | right | rhs
  static constexpr int ROW = 24 * NU + 12;                   // A (12 x NU), B (12 x NU), r (12)
  static constexpr int REC_R = 0, REC_CA = NU * NU, REC_CB = 2 * NU * NU, REC_G = 3 * NU * NU;
  static constexpr int REC_V = 3 * NU * NU + NU, REC_TAU = REC_V + 24 * NK, REC = REC_TAU + NK;
};

struct BvpArgs {
  int n_nodes, n_batch, S_traj;       // S_traj = n_nodes - 1
  double* rows0; double* rows1;       // ping-pong block rows  [n_batch][S_traj][ROW]
  double* rec;                        // [n_batch][n_nodes][REC]  (entries 1 .. n_nodes-2 used)
  double* delta; long ldx;            // SoA [12][ldx], node j = b*n_nodes + k
};

// level-0 rows from the STM sweep's outputs
template <int NU>
__global__ __launch_bounds__(256) void k_bvp_init(const double* __restrict__ Phi, long ldp, const double* __restrict__ defect,
                                                  long ldd, BvpArgs a) {
  using D = BvpDims<NU>;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long S_total = (long)a.S_traj * a.n_batch;
  if (idx >= S_total * D::ROW) return;
  const long s = idx / D::ROW;
  const int e = (int)(idx - s * D::ROW);
  const int i = (int)(s % a.S_traj);
  double v;
  if (e < 12 * NU) {                   // A: columns of Phi_i that belong to free unknowns
    const int c = e / 12, r = e % 12;
    const int pc = (NU == 12) ? c : 6 + c;
    v = (NU == 12 && i == 0 && c < 6) ? 0.0 : Phi[(long)(pc * 12 + r) * ldp + s];   // fixed initial state (:141)
  } else if (e < 24 * NU) {            // B = -I restricted to the free unknowns
    const int c = (e - 12 * NU) / 12, r = (e - 12 * NU) % 12;
    const int pc = (NU == 12) ? c : 6 + c;
    v = (r == pc && !(NU == 12 && i == a.S_traj - 1 && c < 6)) ? -1.0 : 0.0;       // fixed final state (:142)
  } else {
    v = -defect[(long)(e - 24 * NU) * ldd + s];
  }
  a.rows0[idx] = v;
}

template <int NU>
__global__ __launch_bounds__(256) void k_bvp_init_rhs(const double* __restrict__ defect, long ldd, BvpArgs a) {
  using D = BvpDims<NU>;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over S_total * 12
  const long S_total = (long)a.S_traj * a.n_batch;
  if (idx >= S_total * 12) return;
  const long s = idx / 12;
  const int c = (int)(idx - s * 12);
  a.rows0[s * D::ROW + 24 * NU + c] = -defect[(long)c * ldd + s];
}

// One wavefront per pair (j = pair index or the carry slot, b = trajectory, c = lane).  Lane c < NCOLS holds column c
// of the stack.
template <int NU>
__device__ __forceinline__ void bvp_reduce_pair(const BvpArgs& a, int level, int M, const double* cur, double* nxt, const int j,
                                                const int b, const int c) {
  using D = BvpDims<NU>;
  const int npairs = M / 2;
  const double* rows = cur + (long)b * a.S_traj * D::ROW;
  double* out = nxt + (long)b * a.S_traj * D::ROW;
  if (j >= npairs) {                   // odd row carried to the next level unchanged
    const double* src = rows + (long)(M - 1) * D::ROW;
    double* dst = out + (long)npairs * D::ROW;
    for (int e = c; e < D::ROW; e += 64) dst[e] = src[e];
    return;
  }
  const double* top = rows + (long)(2 * j) * D::ROW;
  const double* bot = top + D::ROW;
  double col[24];
#pragma unroll
  for (int r = 0; r < 24; ++r) col[r] = 0.0;
  if (c < NU) {                        // shared unknown: [B_top; A_bot]
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = top[12 * NU + c * 12 + r]; col[12 + r] = bot[c * 12 + r]; }
  } else if (c < 2 * NU) {             // left unknown: [A_top; 0]
#pragma unroll
    for (int r = 0; r < 12; ++r) col[r] = top[(c - NU) * 12 + r];
  } else if (c < 3 * NU) {             // right unknown: [0; B_bot]
#pragma unroll
    for (int r = 0; r < 12; ++r) col[12 + r] = bot[12 * NU + (c - 2 * NU) * 12 + r];
  } else if (c == 3 * NU) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = top[24 * NU + r]; col[12 + r] = bot[24 * NU + r]; }
  }
  double tau_mine = 0.0;
#pragma unroll
  for (int k = 0; k < D::NK; ++k) {
    // reflector from column k (computed in every lane, only lane k's is used)
    // (four partial sums: the 23-term chain is otherwise the longest dependent stretch of the reflection; reciprocal and
    // reciprocal square root by Newton refinement of the hardware seeds, ~1 ulp, instead of the IEEE sequences: ~20 against
    // ~100 dependent instructions per reflection, twelve reflections per pair)
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
    for (int r = k + 1; r < 24; ++r) {
      if (((r - k - 1) & 3) == 0) q0 = __builtin_fma(col[r], col[r], q0);
      else if (((r - k - 1) & 3) == 1) q1 = __builtin_fma(col[r], col[r], q1);
      else if (((r - k - 1) & 3) == 2) q2 = __builtin_fma(col[r], col[r], q2);
      else q3 = __builtin_fma(col[r], col[r], q3);
    }
    const double xn2 = (q0 + q1) + (q2 + q3);
    const double alpha = col[k];
    const double n2 = __builtin_fma(alpha, alpha, xn2);
    const bool trivial = (xn2 == 0.0);
    const double nrm = trivial ? fabs(alpha) : n2 * rsqrt_nr(n2);
    const double beta = (alpha >= 0.0) ? -nrm : nrm;
    const double tau_k = trivial ? 0.0 : (beta - alpha) * rcp_nr(beta);
    const double scl = trivial ? 0.0 : rcp_nr(alpha - beta);
    if (c == k) {
      tau_mine = tau_k;
      if (!trivial) {
        col[k] = beta;
#pragma unroll
        for (int r = k + 1; r < 24; ++r) col[r] *= scl;
      }
    }
    const double tau_b = __shfl(tau_k, k);
    double v[24];
#pragma unroll
    for (int r = k + 1; r < 24; ++r) v[r] = __shfl(col[r], k);
    if (c > k && c < D::NCOLS) {
      double w0 = col[k], w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
      for (int r = k + 1; r < 24; ++r) {
        if (((r - k - 1) & 3) == 0) w0 = __builtin_fma(v[r], col[r], w0);
        else if (((r - k - 1) & 3) == 1) w1 = __builtin_fma(v[r], col[r], w1);
        else if (((r - k - 1) & 3) == 2) w2 = __builtin_fma(v[r], col[r], w2);
        else w3 = __builtin_fma(v[r], col[r], w3);
      }
      double w = (w0 + w1) + (w2 + w3);
      w *= tau_b;
      col[k] -= w;
#pragma unroll
      for (int r = k + 1; r < 24; ++r) col[r] = __builtin_fma(-w, v[r], col[r]);
    }
  }
  const int mid = (2 * j + 1) << level;
  double* rec = a.rec + ((long)b * a.n_nodes + mid) * D::REC;
  double* nr = out + (long)j * D::ROW;
  // reflectors: lane k < NK holds v_k below the diagonal
  if (c < D::NK) {
#pragma unroll
    for (int r = 0; r < 24; ++r) rec[D::REC_V + c * 24 + r] = (r > c) ? col[r] : (r == c ? 1.0 : 0.0);
    rec[D::REC_TAU + c] = tau_mine;
  }
  // rows 0 .. NU-1: the eliminated unknown; rows NU .. NU+11: the new block row (entries below the diagonal of a
  // triangularised column are reflector storage, i.e. structural zeros of the matrix)
  if (c < NU) {
#pragma unroll
    for (int r = 0; r < NU; ++r) rec[D::REC_R + c * NU + r] = (r <= c) ? col[r] : 0.0;
  } else if (c < 2 * NU) {
#pragma unroll
    for (int r = 0; r < NU; ++r) rec[D::REC_CA + (c - NU) * NU + r] = col[r];
#pragma unroll
    for (int r = 0; r < 12; ++r) nr[(c - NU) * 12 + r] = (c < D::NK && NU + r > c) ? 0.0 : col[NU + r];
  } else if (c < 3 * NU) {
#pragma unroll
    for (int r = 0; r < NU; ++r) rec[D::REC_CB + (c - 2 * NU) * NU + r] = col[r];
#pragma unroll
    for (int r = 0; r < 12; ++r) nr[12 * NU + (c - 2 * NU) * 12 + r] = (c < D::NK && NU + r > c) ? 0.0 : col[NU + r];
  } else if (c == 3 * NU) {
#pragma unroll
    for (int r = 0; r < NU; ++r) rec[D::REC_G + r] = col[r];
#pragma unroll
    for (int r = 0; r < 12; ++r) nr[24 * NU + r] = col[NU + r];
  }
}

template <int NU>
__global__ __launch_bounds__(64) void k_bvp_reduce(BvpArgs a, int level, int M, const double* __restrict__ cur, double* __restrict__ nxt) {
  bvp_reduce_pair<NU>(a, level, M, cur, nxt, blockIdx.x, blockIdx.y, threadIdx.x);
}

// Re-apply the stored reflectors to a new right-hand side.  One lane per pair (+ carry).
template <int NU>
__device__ __forceinline__ void bvp_reduce_rhs_pair(const BvpArgs& a, int level, int M, const double* cur, double* nxt, const int j,
                                                    const int b) {
  using D = BvpDims<NU>;
  const int npairs = M / 2;
  const double* rows = cur + (long)b * a.S_traj * D::ROW;
  double* out = nxt + (long)b * a.S_traj * D::ROW;
  if (j == npairs && (M & 1)) {
    for (int r = 0; r < 12; ++r) out[(long)npairs * D::ROW + 24 * NU + r] = rows[(long)(M - 1) * D::ROW + 24 * NU + r];
    return;
  }
  if (j >= npairs) return;
  double x[24];
#pragma unroll
  for (int r = 0; r < 12; ++r) { x[r] = rows[(long)(2 * j) * D::ROW + 24 * NU + r]; x[12 + r] = rows[(long)(2 * j + 1) * D::ROW + 24 * NU + r]; }
  const int mid = (2 * j + 1) << level;
  double* rec = a.rec + ((long)b * a.n_nodes + mid) * D::REC;
#pragma unroll
  for (int k = 0; k < D::NK; ++k) {
    const double* v = rec + D::REC_V + k * 24;
    double w = x[k];
#pragma unroll
    for (int r = k + 1; r < 24; ++r) w = __builtin_fma(v[r], x[r], w);
    w *= rec[D::REC_TAU + k];
    x[k] -= w;
#pragma unroll
    for (int r = k + 1; r < 24; ++r) x[r] = __builtin_fma(-w, v[r], x[r]);
  }
#pragma unroll
  for (int r = 0; r < NU; ++r) rec[D::REC_G + r] = x[r];
#pragma unroll
  for (int r = 0; r < 12; ++r) out[(long)j * D::ROW + 24 * NU + r] = x[NU + r];
}

template <int NU>
__global__ __launch_bounds__(64) void k_bvp_reduce_rhs(BvpArgs a, int level, int M, const double* __restrict__ cur, double* __restrict__ nxt) {
  bvp_reduce_rhs_pair<NU>(a, level, M, cur, nxt, blockIdx.x * 64 + threadIdx.x, blockIdx.y);
}

// Last level: one row  A d_first + B d_last = r.  NU = 12: the end states are fixed, 12 x 12 system for the two
// end-node costate updates.  NU = 6: 12 x 12 system for the costates of the first and last node.  Gaussian elimination
// with partial pivoting, one lane per trajectory.
template <int NU>
__device__ __forceinline__ void bvp_final_one(const BvpArgs& a, const double* cur, const int b, const int lane) {
  // One wavefront per trajectory: lane r < 12 holds row r of the augmented 12 x 13 system in registers; pivot search,
  // row swap and elimination go through cross-lane shuffles (no scratch arrays).
  using D = BvpDims<NU>;
  const double* row = cur + (long)b * a.S_traj * D::ROW;
  const int r = lane < 12 ? lane : 11;
  constexpr int off = (NU == 12) ? 6 : 0;      // free columns of each block
  double m[13];
#pragma unroll
  for (int c = 0; c < 6; ++c) { m[c] = row[(off + c) * 12 + r]; m[6 + c] = row[12 * NU + (off + c) * 12 + r]; }
  m[12] = row[24 * NU + r];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    // partial pivoting: row with the largest |m[.][k]| among rows >= k (ties -> lowest row)
    double best = (lane >= k && lane < 12) ? fabs(m[k]) : -1.0;
    int piv = lane;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const int op = __shfl_xor(piv, o);
      if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
    }
    piv = __shfl(piv, 0);                       // lanes 0..15 agree after the butterfly
    // swap rows k and piv
    const int src = (lane == k) ? piv : (lane == piv ? k : lane);
#pragma unroll
    for (int c = 0; c < 13; ++c) m[c] = __shfl(m[c], src);
    const double pk = __shfl(m[k], k);
    const double f = (lane > k && lane < 12) ? m[k] / pk : 0.0;
#pragma unroll
    for (int c = k + 1; c < 13; ++c) m[c] = __builtin_fma(-f, __shfl(m[c], k), m[c]);
  }
  // back substitution: x[k] = (m[k][12] - sum_{c>k} m[k][c] x[c]) / m[k][k], broadcast as it is formed
  double x[12];
#pragma unroll
  for (int k = 11; k >= 0; --k) {
    double sacc = m[12];
#pragma unroll
    for (int c = k + 1; c < 12; ++c) sacc = __builtin_fma(-m[c], x[c], sacc);
    x[k] = __shfl(sacc / m[k], k);
  }
  if (lane == 0) {
    const long n0 = (long)b * a.n_nodes, n1 = n0 + a.n_nodes - 1;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      a.delta[(long)c * a.ldx + n0] = 0.0;
      a.delta[(long)c * a.ldx + n1] = 0.0;
      a.delta[(long)(6 + c) * a.ldx + n0] = x[c];
      a.delta[(long)(6 + c) * a.ldx + n1] = x[6 + c];
    }
  }
}

// Back-substitution at one level: d_mid = R^{-1} (g - Ca d_left - Cb d_right).  One lane per pair.
template <int NU>
__device__ __forceinline__ void bvp_backsub_pair(const BvpArgs& a, int level, int M, const int j, const int b) {
  using D = BvpDims<NU>;
  if (j >= M / 2) return;
  const int mid = (2 * j + 1) << level, left = (2 * j) << level;
  int right = (2 * j + 2) << level;
  if (right > a.n_nodes - 1) right = a.n_nodes - 1;
  const double* rec = a.rec + ((long)b * a.n_nodes + mid) * D::REC;
  const long nb = (long)b * a.n_nodes;
  constexpr int off = 12 - NU;                 // NU = 6: unknowns are components 6..11 (the costates)
  double dl[NU], dr[NU], x[NU];
#pragma unroll
  for (int c = 0; c < NU; ++c) { dl[c] = a.delta[(long)(off + c) * a.ldx + nb + left]; dr[c] = a.delta[(long)(off + c) * a.ldx + nb + right]; }
#pragma unroll
  for (int r = 0; r < NU; ++r) {
    double s = rec[D::REC_G + r];
#pragma unroll
    for (int c = 0; c < NU; ++c) s -= rec[D::REC_CA + c * NU + r] * dl[c] + rec[D::REC_CB + c * NU + r] * dr[c];
    x[r] = s;
  }
#pragma unroll
  for (int k = NU - 1; k >= 0; --k) {
    double s = x[k];
#pragma unroll
    for (int c = k + 1; c < NU; ++c) s -= rec[D::REC_R + c * NU + k] * x[c];
    x[k] = s / rec[D::REC_R + k * NU + k];
  }
#pragma unroll
  for (int c = 0; c < NU; ++c) a.delta[(long)(off + c) * a.ldx + nb + mid] = x[c];
  if (NU == 6) {
#pragma unroll
    for (int c = 0; c < 6; ++c) a.delta[(long)c * a.ldx + nb + mid] = 0.0;   // states are not updated
  }
}

// The same with SIXTEEN lanes per pair (lane r = row r of the unknown): the record's matrices are read as rows of NU
// consecutive doubles instead of one 6 KB record per lane (4 096 segments: a level took 15 us with one lane per pair, the
// loads of 64 lanes going to 64 different records).  Lane r forms s_r = g_r - (Ca d_left + Cb d_right)_r, then the triangular
// solve runs column by column: lane k divides, broadcasts x_k inside the 16-lane group, the lanes above it update their s.
template <int NU>
__device__ __forceinline__ void bvp_backsub_pair16(const BvpArgs& a, int level, int M, const int j, const int b, const int r) {
  using D = BvpDims<NU>;
  if (j >= M / 2) return;               // uniform for the 16 lanes of a pair
  const int mid = (2 * j + 1) << level, left = (2 * j) << level;
  int right = (2 * j + 2) << level;
  if (right > a.n_nodes - 1) right = a.n_nodes - 1;
  const double* rec = a.rec + ((long)b * a.n_nodes + mid) * D::REC;
  const long nb = (long)b * a.n_nodes;
  constexpr int off = 12 - NU;
  const int rr = r < NU ? r : NU - 1;   // lanes NU..15 shadow the last row (no stores)
  double s = rec[D::REC_G + rr];
#pragma unroll
  for (int c = 0; c < NU; ++c) {
    const double dl = a.delta[(long)(off + c) * a.ldx + nb + left], dr = a.delta[(long)(off + c) * a.ldx + nb + right];
    s -= rec[D::REC_CA + c * NU + rr] * dl + rec[D::REC_CB + c * NU + rr] * dr;
  }
  double x = 0.0;
#pragma unroll
  for (int k = NU - 1; k >= 0; --k) {
    const double xk = __shfl(s / rec[D::REC_R + k * NU + k], k, 16);      // lane k of the group holds the finished s_k
    if (rr == k) x = xk;
    s = __builtin_fma(-rec[D::REC_R + k * NU + (rr < k ? rr : 0)], (rr < k) ? xk : 0.0, s);
  }
  if (r < NU) a.delta[(long)(off + r) * a.ldx + nb + mid] = x;
  if (NU == 6 && r < 6) a.delta[(long)r * a.ldx + nb + mid] = 0.0;       // states are not updated
}

// one level: 16 lanes per pair, four pairs per wavefront
template <int NU>
__global__ __launch_bounds__(64) void k_bvp_backsub(BvpArgs a, int level, int M) {
  bvp_backsub_pair16<NU>(a, level, M, blockIdx.x * 4 + (threadIdx.x >> 4), blockIdx.y, threadIdx.x & 15);
}

// Two levels of the reduction in one launch (a level is ~15 us of launch + latency whatever its size, and 4 096 segments
// are twelve levels).  Workgroup g: its two wavefronts reduce the pairs 2g and 2g + 1 of `level` (cur -> nxt), then --
// block-scope barrier: both rows were written by this workgroup -- wavefront 0 reduces pair g of level + 1 (nxt -> out).
// `out` is a third buffer: other workgroups may still be reading cur.
template <int NU>
__global__ __launch_bounds__(128) void k_bvp_reduce2(BvpArgs a, int level, int M, const double* __restrict__ cur, double* nxt, double* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = blockIdx.x, b = blockIdx.y;
  const int M1 = M / 2 + (M & 1);
  const int j = 2 * g + wave;
  if (j < M1) bvp_reduce_pair<NU>(a, level, M, cur, nxt, j, b, lane);
  __syncthreads();
  if (wave == 0 && g < M1 / 2 + (M1 & 1)) bvp_reduce_pair<NU>(a, level + 1, M1, nxt, out, g, b, lane);
}
// the same for a new right-hand side: one lane does the pairs 2g, 2g + 1 of `level` and then pair g of level + 1
template <int NU>
__global__ __launch_bounds__(64) void k_bvp_reduce_rhs2(BvpArgs a, int level, int M, const double* __restrict__ cur, double* nxt, double* out) {
  const int g = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  const int M1 = M / 2 + (M & 1);
  if (g >= M1 / 2 + (M1 & 1)) return;
  bvp_reduce_rhs_pair<NU>(a, level, M, cur, nxt, 2 * g, b);
  bvp_reduce_rhs_pair<NU>(a, level, M, cur, nxt, 2 * g + 1, b);
  __threadfence_block();               // this lane reads back what it has just written
  bvp_reduce_rhs_pair<NU>(a, level + 1, M1, nxt, out, g, b);
}
// Two levels of the back-substitution: the 16-lane group that forms the unknown of pair g at level + 1 then forms those of the
// pairs 2g and 2g + 1 of `level`, whose outer unknowns are that one and unknowns of earlier launches.
template <int NU>
__global__ __launch_bounds__(64) void k_bvp_backsub2(BvpArgs a, int level, int M) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 4), r = threadIdx.x & 15, b = blockIdx.y;
  const int M1 = M / 2 + (M & 1);
  if (g >= M1 / 2 + (M1 & 1)) return;  // uniform for the 16 lanes of a group
  bvp_backsub_pair16<NU>(a, level + 1, M1, g, b, r);
  __threadfence_block();               // the group reads back the unknown it has just stored
  bvp_backsub_pair16<NU>(a, level, M, 2 * g, b, r);
  bvp_backsub_pair16<NU>(a, level, M, 2 * g + 1, b, r);
}

// Tail of the reduction in ONE launch: once a level has at most 16 block rows (8 pairs = 8 wavefronts, 2 per SIMD: the full register budget) the
// remaining levels, the final 12 x 12 solve and the matching back-substitution levels run inside one 512-thread
// workgroup per trajectory, separated by __syncthreads() (block-scope ordering of the global-memory block rows).
// A 30-node problem (the reference demo) is then init + this kernel instead of 12 dependent tiny launches.
constexpr int BVP_TAIL_MAX = 16;

template <int NU>
__global__ __launch_bounds__(512) void k_bvp_tail(BvpArgs a, int level0, int M0, double* cur, double* nxt, int factor) {
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int M = M0, level = level0;
  int Ms[8];
  while (M > 1) {
    Ms[level - level0] = M;
    const int npairs = M / 2, carry = M & 1;
    if (factor) { if (wave < npairs + carry) bvp_reduce_pair<NU>(a, level, M, cur, nxt, wave, b, lane); }
    else if (tid < npairs + carry) bvp_reduce_rhs_pair<NU>(a, level, M, cur, nxt, tid, b);
    __syncthreads();
    double* t = cur; cur = nxt; nxt = t;
    M = npairs + carry;
    ++level;
  }
  if (wave == 0) bvp_final_one<NU>(a, cur, b, lane);
  __syncthreads();
  for (int l = level - 1; l >= level0; --l) {
    bvp_backsub_pair16<NU>(a, l, Ms[l - level0], tid >> 4, b, tid & 15);     // at most 8 pairs: 32 groups of 16 lanes
    __syncthreads();
  }
}

// y = x + alpha * d  (elementwise over an SoA [rows][ld] block): trial points, SOC accumulation
__global__ __launch_bounds__(256) void k_axpy(const double* __restrict__ x, const double* __restrict__ d, double alpha,
                                              double* __restrict__ y, long count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) y[i] = __builtin_fma(alpha, d[i], x[i]);
}

size_t bvp_workspace_doubles(int n_nodes, int n_batch) {
  // sized for the larger (NU = 12) variant; the adjoints-only variant uses a prefix of the same workspace
  // three buffers of block rows (a two-level launch reads one and writes the other two; same per-trajectory pitch), the records
  return (size_t)3 * (n_nodes - 1) * n_batch * BvpDims<12>::ROW + (size_t)n_nodes * n_batch * BvpDims<12>::REC;
}

template <int NU>
static hipError_t bvp_solve_impl(const double* Phi, long ldp, const double* defect, long ldd, int n_nodes, int n_batch,
                                 double* workspace, double* delta, long ldx, hipStream_t st) {
  using D = BvpDims<NU>;
  BvpArgs a;
  a.n_nodes = n_nodes; a.n_batch = n_batch; a.S_traj = n_nodes - 1;
  const size_t rows_sz = (size_t)a.S_traj * n_batch * D::ROW;
  a.rows0 = workspace; a.rows1 = workspace + rows_sz; a.rec = workspace + 3 * rows_sz;
  double* rows2 = workspace + 2 * rows_sz;
  a.delta = delta; a.ldx = ldx;
  const long S_total = (long)a.S_traj * n_batch;
  if (Phi) {
    const long cnt = S_total * D::ROW;
    hipLaunchKernelGGL((k_bvp_init<NU>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, Phi, ldp, defect, ldd, a);
  } else {
    const long cnt = S_total * 12;
    hipLaunchKernelGGL((k_bvp_init_rhs<NU>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, defect, ldd, a);
  }
  double* cur = a.rows0;
  double* nxt = a.rows1;
  double* spare = rows2;
  int M = a.S_traj, level = 0;
  int Ms[40];
  while (M > BVP_TAIL_MAX) {
    const int M1 = M / 2 + (M & 1);
    if (M1 > BVP_TAIL_MAX) {           // two levels in one launch: cur -> nxt -> spare
      const int M2 = M1 / 2 + (M1 & 1);
      Ms[level] = M; Ms[level + 1] = M1;
      if (Phi) hipLaunchKernelGGL((k_bvp_reduce2<NU>), dim3(M2, n_batch), dim3(128), 0, st, a, level, M, cur, nxt, spare);
      else hipLaunchKernelGGL((k_bvp_reduce_rhs2<NU>), dim3((M2 + 63) / 64, n_batch), dim3(64), 0, st, a, level, M, cur, nxt, spare);
      double* t = cur; cur = spare; spare = t;
      M = M2;
      level += 2;
    } else {
      Ms[level] = M;
      if (Phi) hipLaunchKernelGGL((k_bvp_reduce<NU>), dim3(M1, n_batch), dim3(64), 0, st, a, level, M, cur, nxt);
      else hipLaunchKernelGGL((k_bvp_reduce_rhs<NU>), dim3((M1 + 63) / 64, n_batch), dim3(64), 0, st, a, level, M, cur, nxt);
      double* t = cur; cur = nxt; nxt = t;
      M = M1;
      ++level;
    }
  }
  // remaining levels, final solve and their back-substitution in one launch per trajectory
  hipLaunchKernelGGL((k_bvp_tail<NU>), dim3(n_batch), dim3(512), 0, st, a, level, M, cur, nxt, Phi ? 1 : 0);
  int l = level - 1;
  for (; l >= 1; l -= 2) {             // levels l and l - 1 in one launch
    const int M1 = Ms[l];
    const int groups = M1 / 2 + (M1 & 1);
    hipLaunchKernelGGL((k_bvp_backsub2<NU>), dim3((groups + 3) / 4, n_batch), dim3(64), 0, st, a, l - 1, Ms[l - 1]);
  }
  if (l == 0) {
    const int npairs = Ms[0] / 2;
    hipLaunchKernelGGL((k_bvp_backsub<NU>), dim3((npairs + 3) / 4, n_batch), dim3(64), 0, st, a, 0, Ms[0]);
  }
  return hipGetLastError();
}

// Factor (if Phi != null) or re-apply to a new rhs (Phi == null), then solve: delta[12][ldx] (SoA, node-indexed).
hipError_t launch_bvp_solve(const double* Phi, long ldp, const double* defect, long ldd, int n_nodes, int n_batch,
                            int adjoints_only, double* workspace, double* delta, long ldx, hipStream_t st) {
  return adjoints_only ? bvp_solve_impl<6>(Phi, ldp, defect, ldd, n_nodes, n_batch, workspace, delta, ldx, st)
                       : bvp_solve_impl<12>(Phi, ldp, defect, ldd, n_nodes, n_batch, workspace, delta, ldx, st);
}

hipError_t launch_axpy(const double* x, const double* d, double alpha, double* y, long count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, x, d, alpha, y, count);
  return hipGetLastError();
}

}  // namespace lto
