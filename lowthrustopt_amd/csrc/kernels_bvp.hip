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
// long, unstable trajectories.  The orthogonal factor of every pair is stored as an explicit 24 x 24 matrix (formed for
// free by 24 lanes the stack leaves idle), so a second right-hand side (the second-order-correction re-solve, :190-214) is
// one small matrix-vector product per pair and level.
//
// Round 4: SIXTEEN consecutive block rows per workgroup and FOUR levels per launch -- the eight pairs of the first level in
// eight wavefronts, the rows of the levels in between in LDS, one block row per workgroup out -- with the level-0 rows read
// straight from the sweep's outputs.  4 096 segments: chunk, chunk, tail (16 rows: four levels, the 12 x 12 solve, four levels
// of back-substitution), two back-substitution launches of four levels each: 5 launches for 12 + 12 levels (round 3: 10
// launches, two levels each, the rows of every level through HBM, and a re-solve that re-applied 12 reflectors per pair in ONE
// lane reading its own 6 KB record).
//
// Node eliminated at level l by pair j:  mid = (2j+1) 2^l,  left = 2j 2^l,  right = min((2j+2) 2^l, n-1).
#include "kernels.hpp"
#include <type_traits>

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
  static constexpr int NLANES = NCOLS + 24;                  // ... | the 24 columns of the identity, which end up as Q^T
  static constexpr int ROW = 24 * NU + 12;                   // A (12 x NU), B (12 x NU), r (12)
  static constexpr int REC_R = 0, REC_CA = NU * NU, REC_CB = 2 * NU * NU, REC_G = 3 * NU * NU;
  static constexpr int REC_QT = 3 * NU * NU + NU, REC = REC_QT + 24 * 24;    // QT[c][r] = (Q^T)_{r c}: lane r reads consecutive doubles
  static_assert(NLANES <= 64, "one wavefront per pair");
};
constexpr int BVP_CHUNK = 16;      // block rows per workgroup of a chunk launch (four levels)
constexpr int BVP_CHUNK_LEVELS = 4;

struct BvpArgs {
  int n_nodes, n_batch, S_traj;       // S_traj = n_nodes - 1
  double* rec;                        // [n_batch][n_nodes][REC]  (entries 1 .. n_nodes-2 used)
  double* delta; long ldx;            // SoA [12][ldx], node j = b*n_nodes + k
  const double* Phi; long ldp;        // the sweep's outputs: level-0 rows are read from them directly
  const double* defect; long ldd;
};

// ---- the stack of a pair, lane c = column c: [mid | left | right | rhs | identity]
// from two block rows in memory (global or LDS)
template <int NU>
__device__ __forceinline__ void bvp_stack_rows(const double* top, const double* bot, const int c, double (&col)[24]) {
  using D = BvpDims<NU>;
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
  } else if (c < D::NLANES) {
#pragma unroll
    for (int r = 0; r < 24; ++r) col[r] = (r == c - D::NCOLS) ? 1.0 : 0.0;
  }
}
// element e of the level-0 block row of segment i of trajectory b (s = b S_traj + i): [A | B | r] from the sweep's outputs
template <int NU>
__device__ __forceinline__ double bvp_row0_A(const BvpArgs& a, const long s, const int i, const int c, const int r) {
  const int pc = (NU == 12) ? c : 6 + c;                                            // columns of Phi_i that belong to free unknowns
  return (NU == 12 && i == 0 && c < 6) ? 0.0 : a.Phi[(long)(pc * 12 + r) * a.ldp + s];   // fixed initial state (:141)
}
template <int NU>
__device__ __forceinline__ double bvp_row0_B(const BvpArgs& a, const int i, const int c, const int r) {
  const int pc = (NU == 12) ? c : 6 + c;                                            // -I restricted to the free unknowns
  return (r == pc && !(NU == 12 && i == a.S_traj - 1 && c < 6)) ? -1.0 : 0.0;       // fixed final state (:142)
}
// ... the stack of the level-0 pair (2j, 2j + 1) without materialising the rows
template <int NU>
__device__ __forceinline__ void bvp_stack_sweep(const BvpArgs& a, const int b, const int j, const int c, double (&col)[24]) {
  using D = BvpDims<NU>;
  const int it = 2 * j, ib = 2 * j + 1;
  const long st = (long)b * a.S_traj + it, sb = st + 1;
#pragma unroll
  for (int r = 0; r < 24; ++r) col[r] = 0.0;
  if (c < NU) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = bvp_row0_B<NU>(a, it, c, r); col[12 + r] = bvp_row0_A<NU>(a, sb, ib, c, r); }
  } else if (c < 2 * NU) {
#pragma unroll
    for (int r = 0; r < 12; ++r) col[r] = bvp_row0_A<NU>(a, st, it, c - NU, r);
  } else if (c < 3 * NU) {
#pragma unroll
    for (int r = 0; r < 12; ++r) col[12 + r] = bvp_row0_B<NU>(a, ib, c - 2 * NU, r);
  } else if (c == 3 * NU) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = -a.defect[(long)r * a.ldd + st]; col[12 + r] = -a.defect[(long)r * a.ldd + sb]; }
  } else if (c < D::NLANES) {
#pragma unroll
    for (int r = 0; r < 24; ++r) col[r] = (r == c - D::NCOLS) ? 1.0 : 0.0;
  }
}
// a level-0 row that has no partner (odd count): materialised as it is carried up
template <int NU>
__device__ __forceinline__ void bvp_row_from_sweep(const BvpArgs& a, const int b, const int i, double* dst, const int lane) {
  using D = BvpDims<NU>;
  const long s = (long)b * a.S_traj + i;
  for (int e = lane; e < D::ROW; e += 64) {
    double v;
    if (e < 12 * NU) v = bvp_row0_A<NU>(a, s, i, e / 12, e % 12);
    else if (e < 24 * NU) v = bvp_row0_B<NU>(a, i, (e - 12 * NU) / 12, (e - 12 * NU) % 12);
    else v = -a.defect[(long)(e - 24 * NU) * a.ldd + s];
    dst[e] = v;
  }
}

// x of lane K (a compile-time lane) in every lane: two v_readlane_b32 into a scalar register pair, which the following FMAs read
// directly.  (HIP's __shfl is an LDS-crossbar ds_bpermute_b32 per dword: 444 of them and their waits per pair reduction.)
template <int K>
__device__ __forceinline__ double lane_bcast(const double x) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), K), hi = __builtin_amdgcn_readlane(__double2hiint(x), K);
  return __hiloint2double(hi, lo);
}
// x of lane K of the own 16-lane row (DPP row_newbcast)
template <int K>
__device__ __forceinline__ double row_bcast(const double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x150 + K, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x150 + K, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void bvp_static_for(F&& f) {
  if constexpr (FIRST < LAST) {
    f(std::integral_constant<int, FIRST>{});
    bvp_static_for<FIRST + 1, LAST>(f);
  }
}

// ---- NK Householder reflections on the stack (one wavefront, lane = column; the reflector of step k is lane k's column,
// broadcast by v_readlane), then the record of the eliminated node `rec` and the new block row `nr`
template <int NU>
__device__ __forceinline__ void bvp_reflect_store(double (&col)[24], const int c, double* rec, double* nr) {
  using D = BvpDims<NU>;
  bvp_static_for<0, D::NK>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    // Reflection k: H = I - g v v^T with v = (alpha - beta, x_{k+1}, ..., x_23) taken from lane k's column AS IT IS, beta =
    // -sign(alpha) |(alpha, x)|, g = 1 / (beta (beta - alpha)).  Round 4: the reflector is not normalised (nobody stores it any
    // more -- the record keeps R and the explicit Q^T), so lane k's column is broadcast first (v_readlane, overlapping the norm's
    // chain), every lane forms the same norm and g from the broadcast values (one reciprocal square root and ONE reciprocal by
    // Newton refinement of the hardware seeds, ~1 ulp), and the 23 scaling multiplications and the broadcast of tau are gone:
    // ~130 instead of ~165 instructions per reflection, twelve (eighteen) reflections per pair.
    double x[24];
#pragma unroll
    for (int r = k + 1; r < 24; ++r) x[r] = lane_bcast<k>(col[r]);
    const double alpha = lane_bcast<k>(col[k]);
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
#pragma unroll
    for (int r = k + 1; r < 24; ++r) {
      if (((r - k - 1) & 3) == 0) q0 = __builtin_fma(x[r], x[r], q0);
      else if (((r - k - 1) & 3) == 1) q1 = __builtin_fma(x[r], x[r], q1);
      else if (((r - k - 1) & 3) == 2) q2 = __builtin_fma(x[r], x[r], q2);
      else q3 = __builtin_fma(x[r], x[r], q3);
    }
    const double xn2 = (q0 + q1) + (q2 + q3);
    const bool trivial = (xn2 == 0.0);                 // nothing below the diagonal: H = I
    const double n2 = __builtin_fma(alpha, alpha, xn2);
    const double nrm = n2 * rsqrt_nr(n2);
    const double beta = (alpha >= 0.0) ? -nrm : nrm;
    const double vk = alpha - beta;
    const double g = trivial ? 0.0 : rcp_nr(beta * (beta - alpha));
    if (c == k && !trivial) col[k] = beta;
    if (c > k && c < D::NLANES) {
      double w0 = vk * col[k], w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll
      for (int r = k + 1; r < 24; ++r) {
        if (((r - k - 1) & 3) == 0) w0 = __builtin_fma(x[r], col[r], w0);
        else if (((r - k - 1) & 3) == 1) w1 = __builtin_fma(x[r], col[r], w1);
        else if (((r - k - 1) & 3) == 2) w2 = __builtin_fma(x[r], col[r], w2);
        else w3 = __builtin_fma(x[r], col[r], w3);
      }
      const double w = ((w0 + w1) + (w2 + w3)) * g;
      col[k] = __builtin_fma(-w, vk, col[k]);
#pragma unroll
      for (int r = k + 1; r < 24; ++r) col[r] = __builtin_fma(-w, x[r], col[r]);
    }
  });
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
  } else if (c < D::NLANES) {          // column c' of the identity has become Q^T e_c'
#pragma unroll
    for (int r = 0; r < 24; ++r) rec[D::REC_QT + (c - D::NCOLS) * 24 + r] = col[r];
  }
}

// Node eliminated at level l by pair j
__device__ __forceinline__ int bvp_mid(const int j, const int level) { return (2 * j + 1) << level; }

// ---- level-0 rows into a buffer (only when the whole problem fits the tail launch: at most 16 segments)
template <int NU>
__global__ __launch_bounds__(64) void k_bvp_rows0(BvpArgs a, double* rows) {
  using D = BvpDims<NU>;
  const int i = blockIdx.x, b = blockIdx.y;
  bvp_row_from_sweep<NU>(a, b, i, rows + ((long)b * a.S_traj + i) * D::ROW, threadIdx.x);
}
template <int NU>
__global__ __launch_bounds__(256) void k_bvp_rhs0(BvpArgs a, double* rhs) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over S_total * 12
  const long S_total = (long)a.S_traj * a.n_batch;
  if (idx >= S_total * 12) return;
  const long s = idx / 12;
  const int c = (int)(idx - s * 12);
  rhs[idx] = -a.defect[(long)c * a.ldd + s];
}

// ---- FOUR levels of the reduction in one launch: workgroup g of trajectory b owns the rows [16 g, 16 g + 16) of level
// `level0` (M0 rows; FIRST: level 0, read from the sweep's outputs) and everything that grows out of them: 8, 4, 2, 1 pairs in
// as many wavefronts, the rows in between in LDS, its one row of level level0 + 4 to out[g].  A row without a partner (odd
// count) is carried up unchanged.
template <int NU, bool FIRST>
__global__ __launch_bounds__(512) void k_bvp_chunk(BvpArgs a, const int level0, const int M0, const double* __restrict__ cur, double* __restrict__ out) {
  using D = BvpDims<NU>;
  __shared__ double bufA[8][D::ROW];
  __shared__ double bufB[4][D::ROW];
  const int g = blockIdx.x, b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* rows = cur + (long)b * a.S_traj * D::ROW;
  double* orow = out + ((long)b * a.S_traj + g) * D::ROW;
  int M = M0;
#pragma unroll
  for (int l = 0; l < BVP_CHUNK_LEVELS; ++l) {
    const int width = 8 >> l;                    // pairs of this workgroup at this level
    const int j = g * width + wave;              // global pair index at this level
    if (wave < width && 2 * j < M) {
      double* dst = (l == 0) ? bufA[wave] : (l == 1) ? bufB[wave] : (l == 2) ? bufA[wave] : orow;
      const double* top = (l == 0) ? rows + (long)(2 * j) * D::ROW : (l == 1) ? bufA[2 * wave] : (l == 2) ? bufB[2 * wave] : bufA[0];
      const double* bot = (l == 0) ? top + D::ROW : (l == 1) ? bufA[2 * wave + 1] : (l == 2) ? bufB[2 * wave + 1] : bufA[1];
      if (2 * j + 1 < M) {
        double col[24];
        if (FIRST && l == 0) bvp_stack_sweep<NU>(a, b, j, lane, col);
        else bvp_stack_rows<NU>(top, bot, lane, col);
        bvp_reflect_store<NU>(col, lane, a.rec + ((long)b * a.n_nodes + bvp_mid(j, level0 + l)) * D::REC, dst);
      } else if (FIRST && l == 0) {
        bvp_row_from_sweep<NU>(a, b, 2 * j, dst, lane);
      } else {
        for (int e = lane; e < D::ROW; e += 64) dst[e] = top[e];
      }
    }
    __syncthreads();
    M = (M + 1) / 2;
  }
}

// ---- a new right-hand side through the stored orthogonal factors (the second-order-correction re-solve): per pair
// x' = Q^T [r_top; r_bot], 32 lanes (lane r = row r), the first NU entries are the eliminated node's g, the next 12 the new
// row's right-hand side.  Right-hand sides travel in their own small buffers [n_batch][S_traj][12].
template <int NU>
__device__ __forceinline__ void bvp_rhs_pair(const double* xt, const double* xb, const double* rec_c, double* rec_g, double* nr, const int r) {
  using D = BvpDims<NU>;
  const int rr = r < 24 ? r : 23;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int c = 0; c < 12; c += 2) {
    acc0 = __builtin_fma(rec_c[D::REC_QT + c * 24 + rr], xt[c], acc0);
    acc1 = __builtin_fma(rec_c[D::REC_QT + (c + 1) * 24 + rr], xt[c + 1], acc1);
  }
#pragma unroll
  for (int c = 0; c < 12; c += 2) {
    acc0 = __builtin_fma(rec_c[D::REC_QT + (12 + c) * 24 + rr], xb[c], acc0);
    acc1 = __builtin_fma(rec_c[D::REC_QT + (13 + c) * 24 + rr], xb[c + 1], acc1);
  }
  const double x = acc0 + acc1;
  if (r < NU) rec_g[D::REC_G + r] = x;
  else if (r < NU + 12) nr[r - NU] = x;
}
template <int NU, bool FIRST>
__global__ __launch_bounds__(256) void k_bvp_chunk_rhs(BvpArgs a, const int level0, const int M0, const double* __restrict__ cur, double* __restrict__ out) {
  using D = BvpDims<NU>;
  __shared__ double x0[16][12];      // FIRST: this workgroup's level-0 right-hand sides
  __shared__ double xA[8][12];
  __shared__ double xB[4][12];
  const int g = blockIdx.x, b = blockIdx.y, grp = threadIdx.x >> 5, r = threadIdx.x & 31;
  const double* rows = cur + (long)b * a.S_traj * 12;
  double* orow = out + ((long)b * a.S_traj + g) * 12;
  int M = M0;
  if (FIRST) {
    if (threadIdx.x < 16 * 12) {
      const int i = threadIdx.x / 12, c = threadIdx.x % 12;
      const int ig = g * BVP_CHUNK + i;
      x0[i][c] = (ig < M0) ? -a.defect[(long)c * a.ldd + (long)b * a.S_traj + ig] : 0.0;
    }
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < BVP_CHUNK_LEVELS; ++l) {
    const int width = 8 >> l;
    const int j = g * width + grp;
    if (grp < width && 2 * j < M) {
      double* dst = (l == 0) ? xA[grp] : (l == 1) ? xB[grp] : (l == 2) ? xA[grp] : orow;
      const double* top = (l == 0) ? (FIRST ? x0[2 * grp] : rows + (long)(2 * j) * 12) : (l == 1) ? xA[2 * grp] : (l == 2) ? xB[2 * grp] : xA[0];
      const double* bot = (l == 0) ? (FIRST ? x0[2 * grp + 1] : top + 12) : (l == 1) ? xA[2 * grp + 1] : (l == 2) ? xB[2 * grp + 1] : xA[1];
      if (2 * j + 1 < M) {
        double* rec = a.rec + ((long)b * a.n_nodes + bvp_mid(j, level0 + l)) * D::REC;
        bvp_rhs_pair<NU>(top, bot, rec, rec, dst, r);
      } else if (r < 12) {
        dst[r] = top[r];
      }
    }
    __syncthreads();
    M = (M + 1) / 2;
  }
}

// Last level: one row  A d_first + B d_last = r.  NU = 12: the end states are fixed, 12 x 12 system for the two
// end-node costate updates.  NU = 6: 12 x 12 system for the costates of the first and last node.  Gaussian elimination
// with partial pivoting: one wavefront per trajectory, lane r < 12 holds row r of the augmented 12 x 13 system in registers;
// pivot search by DPP, elimination through v_readlane (no scratch arrays, no LDS crossbar).  `rhs` != null: the row's
// right-hand side comes from there (re-solve).
template <int NU>
__device__ __forceinline__ void bvp_final_one(const BvpArgs& a, const double* row, const double* rhs, const int b, const int lane) {
  const int r = lane < 12 ? lane : 11;
  constexpr int off = (NU == 12) ? 6 : 0;      // free columns of each block
  double m[13];
#pragma unroll
  for (int c = 0; c < 6; ++c) { m[c] = row[(off + c) * 12 + r]; m[6 + c] = row[12 * NU + (off + c) * 12 + r]; }
  m[12] = rhs ? rhs[r] : row[24 * NU + r];
  // No row ever moves (round 3 swapped rows through 13 ds_bpermute per step): a lane remembers that its row has been a pivot,
  // the pivot row of step k is read where it sits (v_readlane with a scalar lane), and the back-substitution visits the rows
  // in pivot order.  Pivot search: all-reduce over the 16-lane row by DPP rotations (max is idempotent), ties to the lowest row.
  bool used = lane >= 12;
  int pivs[12];
  bvp_static_for<0, 12>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double best = used ? -1.0 : fabs(m[k]);
    int piv = lane & 15;
    auto step = [&](auto rot_c) {
      constexpr int ctrl = 0x120 + decltype(rot_c)::value;            // row_ror:n
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(best), ctrl, 0xF, 0xF, false);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(best), ctrl, 0xF, 0xF, false);
      const double ob = __hiloint2double(hi, lo);
      const int op = __builtin_amdgcn_mov_dpp(piv, ctrl, 0xF, 0xF, false);
      if (ob > best || (ob == best && op < piv)) { best = ob; piv = op; }
    };
    step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 4>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 1>{});
    const int pv = __builtin_amdgcn_readfirstlane(piv);               // every lane of row 0 holds the same answer
    pivs[k] = pv;
    auto from_piv = [&](const double x) {
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), pv), __builtin_amdgcn_readlane(__double2loint(x), pv));
    };
    const double pk = from_piv(m[k]);
    const double f = (!used && lane != pv) ? m[k] * rcp_nr(pk) : 0.0;
#pragma unroll
    for (int c = k + 1; c < 13; ++c) m[c] = __builtin_fma(-f, from_piv(m[c]), m[c]);
    used = used || (lane == pv);
  });
  // back substitution in pivot order: x[k] from the row that was the pivot of column k, broadcast as it is formed
  double x[12];
  bvp_static_for<0, 12>([&](auto kc) {
    constexpr int k = 11 - decltype(kc)::value;
    double sacc = m[12];
#pragma unroll
    for (int c = k + 1; c < 12; ++c) sacc = __builtin_fma(-m[c], x[c], sacc);
    const double q = sacc / m[k];
    x[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), pivs[k]), __builtin_amdgcn_readlane(__double2loint(q), pivs[k]));
  });
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

// Back-substitution of one pair: d_mid = R^{-1} (g - Ca d_left - Cb d_right), SIXTEEN lanes per pair (lane r = row r of the
// unknown): the record's matrices are read as rows of NU consecutive doubles.  Lane r forms s_r = g_r - (Ca d_left + Cb
// d_right)_r, then the triangular solve runs column by column: lane k divides, broadcasts x_k inside the 16-lane group, the
// lanes above it update their s.
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
  // the reciprocal of this lane's diagonal entry up front (hardware seed + Newton step, ~1 ulp), so that a step of the dependent
  // chain is one multiplication, one broadcast and one FMA instead of an IEEE division sequence of ~30 instructions
  const double rdiag = rcp_nr(rec[D::REC_R + rr * NU + rr]);
  double x = 0.0;
  bvp_static_for<0, NU>([&](auto kc) {
    constexpr int k = NU - 1 - decltype(kc)::value;
    const double xk = row_bcast<k>(s * rdiag);                            // lane k of the group holds the finished s_k
    if (rr == k) x = xk;
    s = __builtin_fma(-rec[D::REC_R + k * NU + (rr < k ? rr : 0)], (rr < k) ? xk : 0.0, s);
  });
  if (r < NU) a.delta[(long)(off + r) * a.ldx + nb + mid] = x;
  if (NU == 6 && r < 6) a.delta[(long)r * a.ldx + nb + mid] = 0.0;       // states are not updated
}

// ---- FOUR levels of the back-substitution in one launch, the mirror image of k_bvp_chunk: workgroup g forms the unknowns its
// chunk eliminated at levels level0 + 3 .. level0 (1, 2, 4, 8 pairs, sixteen lanes each); their outer unknowns are its own of
// the level before or those of earlier launches.  (Workgroup-scope ordering of the unknowns through global memory.)
template <int NU>
__global__ __launch_bounds__(128) void k_bvp_backchunk(BvpArgs a, const int level0, const int M0) {
  const int g = blockIdx.x, b = blockIdx.y, grp = threadIdx.x >> 4, r = threadIdx.x & 15;
  int Ms[BVP_CHUNK_LEVELS];
  Ms[0] = M0;
#pragma unroll
  for (int l = 1; l < BVP_CHUNK_LEVELS; ++l) Ms[l] = (Ms[l - 1] + 1) / 2;
#pragma unroll
  for (int l = BVP_CHUNK_LEVELS - 1; l >= 0; --l) {
    const int width = 8 >> l;
    if (grp < width) bvp_backsub_pair16<NU>(a, level0 + l, Ms[l], g * width + grp, b, r);
    __syncthreads();
  }
}

// ---- the top of the tree in ONE launch: at most 16 block rows -- the remaining levels (8 pairs = 8 wavefronts, 2 per SIMD: the
// full register budget), the final 12 x 12 solve and the matching back-substitution levels inside one 512-thread workgroup per
// trajectory, separated by __syncthreads() (workgroup-scope ordering of the global-memory block rows).  A 30-node problem
// (the reference demo) is a chunk launch, this one and a back-substitution launch.
// factor = 0: the re-solve; `rows` then only supplies the last level's matrix (kept from the factorisation), the right-hand
// sides travel through rhs_cur / rhs_nxt.
template <int NU>
__global__ __launch_bounds__(512) void k_bvp_tail(BvpArgs a, int level0, int M0, double* cur, double* nxt, double* rhs_cur, double* rhs_nxt, int factor) {
  using D = BvpDims<NU>;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int M = M0, level = level0;
  int Ms[8];
  while (M > 1) {
    Ms[level - level0] = M;
    const int npairs = M / 2, carry = M & 1;
    if (factor) {
      const double* rows = cur + (long)b * a.S_traj * D::ROW;
      double* orows = nxt + (long)b * a.S_traj * D::ROW;
      if (wave < npairs) {
        double col[24];
        bvp_stack_rows<NU>(rows + (long)(2 * wave) * D::ROW, rows + (long)(2 * wave + 1) * D::ROW, lane, col);
        bvp_reflect_store<NU>(col, lane, a.rec + ((long)b * a.n_nodes + bvp_mid(wave, level)) * D::REC, orows + (long)wave * D::ROW);
      } else if (wave == npairs && carry) {
        for (int e = lane; e < D::ROW; e += 64) orows[(long)npairs * D::ROW + e] = rows[(long)(M - 1) * D::ROW + e];
      }
    } else {
      const double* rows = rhs_cur + (long)b * a.S_traj * 12;
      double* orows = rhs_nxt + (long)b * a.S_traj * 12;
      const int grp = tid >> 5, r = tid & 31;
      if (grp < npairs) {
        double* rec = a.rec + ((long)b * a.n_nodes + bvp_mid(grp, level)) * D::REC;
        bvp_rhs_pair<NU>(rows + (long)(2 * grp) * 12, rows + (long)(2 * grp + 1) * 12, rec, rec, orows + (long)grp * 12, r);
      } else if (grp == npairs && carry && r < 12) {
        orows[(long)npairs * 12 + r] = rows[(long)(M - 1) * 12 + r];
      }
    }
    __syncthreads();
    double* t = cur; cur = nxt; nxt = t;
    t = rhs_cur; rhs_cur = rhs_nxt; rhs_nxt = t;
    M = npairs + carry;
    ++level;
  }
  if (wave == 0) bvp_final_one<NU>(a, cur + (long)b * a.S_traj * D::ROW, factor ? nullptr : rhs_cur + (long)b * a.S_traj * 12, b, lane);
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

// Workspace: the rows of the levels that cross launches (level 4: a sixteenth of the segments, level 8: ...; the level-0 rows
// of a problem of at most 16 segments), two of them; two right-hand-side buffers; the records; the last level's row kept
// for the re-solve.  Sized for the larger (NU = 12) variant; the adjoints-only variant uses a prefix of every part.
static size_t bvp_rows_doubles(int n_nodes, int n_batch) { return (size_t)(n_nodes - 1) * n_batch * BvpDims<12>::ROW; }
size_t bvp_workspace_doubles(int n_nodes, int n_batch) {
  // (the level-4 buffer needs a sixteenth of this, but a trajectory's rows keep the pitch S_traj * ROW in every buffer: simple
  // indexing for 4 096 segments x 2.4 KB x 2 = 20 MB)
  return 2 * bvp_rows_doubles(n_nodes, n_batch) + 2 * (size_t)(n_nodes - 1) * n_batch * 12 + (size_t)n_nodes * n_batch * BvpDims<12>::REC;
}

template <int NU>
static hipError_t bvp_solve_impl(const double* Phi, long ldp, const double* defect, long ldd, int n_nodes, int n_batch,
                                 double* workspace, double* delta, long ldx, hipStream_t st) {
  BvpArgs a;
  a.n_nodes = n_nodes; a.n_batch = n_batch; a.S_traj = n_nodes - 1;
  const size_t rows_sz = bvp_rows_doubles(n_nodes, n_batch), rhs_sz = (size_t)a.S_traj * n_batch * 12;
  double* rowsA = workspace;
  double* rowsB = workspace + rows_sz;
  double* rhsA = workspace + 2 * rows_sz;
  double* rhsB = rhsA + rhs_sz;
  a.rec = rhsB + rhs_sz;
  a.delta = delta; a.ldx = ldx;
  a.Phi = Phi; a.ldp = ldp; a.defect = defect; a.ldd = ldd;
  const bool factor = (Phi != nullptr);
  // The re-solve walks the same sequence of buffer swaps as the factorisation, so its tail finds the last level's row -- the
  // matrix of the final 12 x 12 system, which only the factorisation forms -- where that pass left it.
  int M = a.S_traj, level = 0;
  int Ms[48];
  double* cur = rowsA;
  double* nxt = rowsB;
  double* rcur = rhsA;
  double* rnxt = rhsB;
  bool first = true;
  if (M <= BVP_CHUNK) {                // the whole problem fits the tail: materialise its level-0 rows
    if (factor) hipLaunchKernelGGL((k_bvp_rows0<NU>), dim3(M, n_batch), dim3(64), 0, st, a, cur);
    else hipLaunchKernelGGL((k_bvp_rhs0<NU>), dim3((unsigned)(((long)M * n_batch * 12 + 255) / 256)), dim3(256), 0, st, a, rcur);
  }
  while (M > BVP_CHUNK) {
    const int Mn = (M + BVP_CHUNK - 1) / BVP_CHUNK;
    int m = M;
    for (int l = 0; l < BVP_CHUNK_LEVELS; ++l) { Ms[level + l] = m; m = (m + 1) / 2; }
    if (factor) {
      if (first) hipLaunchKernelGGL((k_bvp_chunk<NU, true>), dim3(Mn, n_batch), dim3(512), 0, st, a, level, M, cur, nxt);
      else hipLaunchKernelGGL((k_bvp_chunk<NU, false>), dim3(Mn, n_batch), dim3(512), 0, st, a, level, M, cur, nxt);
    } else {
      if (first) hipLaunchKernelGGL((k_bvp_chunk_rhs<NU, true>), dim3(Mn, n_batch), dim3(256), 0, st, a, level, M, rcur, rnxt);
      else hipLaunchKernelGGL((k_bvp_chunk_rhs<NU, false>), dim3(Mn, n_batch), dim3(256), 0, st, a, level, M, rcur, rnxt);
    }
    { double* t = cur; cur = nxt; nxt = t; }
    { double* t = rcur; rcur = rnxt; rnxt = t; }
    first = false;
    M = Mn;
    level += BVP_CHUNK_LEVELS;
  }
  // remaining levels, final solve and their back-substitution in one launch per trajectory
  hipLaunchKernelGGL((k_bvp_tail<NU>), dim3(n_batch), dim3(512), 0, st, a, level, M, cur, nxt, rcur, rnxt, factor ? 1 : 0);
  for (int l = level - BVP_CHUNK_LEVELS; l >= 0; l -= BVP_CHUNK_LEVELS) {
    const int groups = (Ms[l] + BVP_CHUNK - 1) / BVP_CHUNK;
    hipLaunchKernelGGL((k_bvp_backchunk<NU>), dim3(groups, n_batch), dim3(128), 0, st, a, l, Ms[l]);
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
