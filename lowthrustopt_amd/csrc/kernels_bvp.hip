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

// per block row: A (144, column-major), B (144), r (12)
constexpr int ROW_DOUBLES = 300;
// per eliminated node: R (144, upper triangle used), Ca (144), Cb (144), g (12), V (24 x 12 reflectors), tau (12)
constexpr int REC_R = 0, REC_CA = 144, REC_CB = 288, REC_G = 432, REC_V = 444, REC_TAU = 732, REC_DOUBLES = 744;

struct BvpArgs {
  int n_nodes, n_batch, S_traj;       // S_traj = n_nodes - 1
  double* rows0; double* rows1;       // ping-pong block rows  [n_batch][S_traj][ROW_DOUBLES]
  double* rec;                        // [n_batch][n_nodes][REC_DOUBLES]  (entries 1 .. n_nodes-2 used)
  double* delta; long ldx;            // SoA [12][ldx], node j = b*n_nodes + k
};

// level-0 rows from the STM sweep's outputs
__global__ __launch_bounds__(256) void k_bvp_init(const double* __restrict__ Phi, long ldp, const double* __restrict__ defect,
                                                  long ldd, BvpArgs a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over S_total * 300
  const long S_total = (long)a.S_traj * a.n_batch;
  if (idx >= S_total * ROW_DOUBLES) return;
  const long s = idx / ROW_DOUBLES;
  const int e = (int)(idx - s * ROW_DOUBLES);
  const int i = (int)(s % a.S_traj);
  double v;
  if (e < 144) {                       // A = Phi_i, fixed initial state: columns 0..5 of the first block zeroed
    const int c = e / 12;
    v = (i == 0 && c < 6) ? 0.0 : Phi[(long)e * ldp + s];
  } else if (e < 288) {                // B = -I, fixed final state: columns 0..5 of the last block zeroed
    const int c = (e - 144) / 12, r = (e - 144) % 12;
    v = (r == c && !(i == a.S_traj - 1 && c < 6)) ? -1.0 : 0.0;
  } else {
    v = -defect[(long)(e - 288) * ldd + s];
  }
  a.rows0[idx] = v;
}

// right-hand side only (re-solve with the stored factorisation)
__global__ __launch_bounds__(256) void k_bvp_init_rhs(const double* __restrict__ defect, long ldd, BvpArgs a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over S_total * 12
  const long S_total = (long)a.S_traj * a.n_batch;
  if (idx >= S_total * 12) return;
  const long s = idx / 12;
  const int c = (int)(idx - s * 12);
  a.rows0[s * ROW_DOUBLES + 288 + c] = -defect[(long)c * ldd + s];
}

// One wavefront per pair.  grid = (pairs + carry, n_batch), block = 64.
__global__ __launch_bounds__(64) void k_bvp_reduce(BvpArgs a, int level, int M, const double* __restrict__ cur, double* __restrict__ nxt) {
  const int j = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  const int npairs = M / 2;
  const double* rows = cur + (long)b * a.S_traj * ROW_DOUBLES;
  double* out = nxt + (long)b * a.S_traj * ROW_DOUBLES;
  if (j >= npairs) {                   // odd row carried to the next level unchanged
    const double* src = rows + (long)(M - 1) * ROW_DOUBLES;
    double* dst = out + (long)npairs * ROW_DOUBLES;
    for (int e = c; e < ROW_DOUBLES; e += 64) dst[e] = src[e];
    return;
  }
  const double* top = rows + (long)(2 * j) * ROW_DOUBLES;
  const double* bot = top + ROW_DOUBLES;
  double col[24];
#pragma unroll
  for (int r = 0; r < 24; ++r) col[r] = 0.0;
  if (c < 12) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = top[144 + c * 12 + r]; col[12 + r] = bot[c * 12 + r]; }
  } else if (c < 24) {
#pragma unroll
    for (int r = 0; r < 12; ++r) col[r] = top[(c - 12) * 12 + r];
  } else if (c < 36) {
#pragma unroll
    for (int r = 0; r < 12; ++r) col[12 + r] = bot[144 + (c - 24) * 12 + r];
  } else if (c == 36) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { col[r] = top[288 + r]; col[12 + r] = bot[288 + r]; }
  }
  double tau_mine = 0.0;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    // reflector from column k (computed in every lane, only lane k's is used)
    double xn2 = 0.0;
#pragma unroll
    for (int r = k + 1; r < 24; ++r) xn2 = __builtin_fma(col[r], col[r], xn2);
    const double alpha = col[k];
    const double nrm = sqrt(__builtin_fma(alpha, alpha, xn2));
    const double beta = (alpha >= 0.0) ? -nrm : nrm;
    const bool trivial = (xn2 == 0.0);
    const double tau_k = trivial ? 0.0 : (beta - alpha) / beta;
    const double scl = trivial ? 0.0 : 1.0 / (alpha - beta);
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
    if (c > k && c <= 36) {
      double w = col[k];
#pragma unroll
      for (int r = k + 1; r < 24; ++r) w = __builtin_fma(v[r], col[r], w);
      w *= tau_b;
      col[k] -= w;
#pragma unroll
      for (int r = k + 1; r < 24; ++r) col[r] = __builtin_fma(-w, v[r], col[r]);
    }
  }
  const int mid = (2 * j + 1) << level;
  double* rec = a.rec + ((long)b * a.n_nodes + mid) * REC_DOUBLES;
  double* nr = out + (long)j * ROW_DOUBLES;
  if (c < 12) {
#pragma unroll
    for (int r = 0; r < 12; ++r) rec[REC_R + c * 12 + r] = (r <= c) ? col[r] : 0.0;
#pragma unroll
    for (int r = 0; r < 24; ++r) rec[REC_V + c * 24 + r] = (r > c) ? col[r] : (r == c ? 1.0 : 0.0);
    rec[REC_TAU + c] = tau_mine;
  } else if (c < 24) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { rec[REC_CA + (c - 12) * 12 + r] = col[r]; nr[(c - 12) * 12 + r] = col[12 + r]; }
  } else if (c < 36) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { rec[REC_CB + (c - 24) * 12 + r] = col[r]; nr[144 + (c - 24) * 12 + r] = col[12 + r]; }
  } else if (c == 36) {
#pragma unroll
    for (int r = 0; r < 12; ++r) { rec[REC_G + r] = col[r]; nr[288 + r] = col[12 + r]; }
  }
}

// Re-apply the stored reflectors to a new right-hand side.  One lane per pair (+ carry).
__global__ __launch_bounds__(64) void k_bvp_reduce_rhs(BvpArgs a, int level, int M, const double* __restrict__ cur, double* __restrict__ nxt) {
  const int j = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  const int npairs = M / 2;
  const double* rows = cur + (long)b * a.S_traj * ROW_DOUBLES;
  double* out = nxt + (long)b * a.S_traj * ROW_DOUBLES;
  if (j == npairs && (M & 1)) {
    for (int r = 0; r < 12; ++r) out[(long)npairs * ROW_DOUBLES + 288 + r] = rows[(long)(M - 1) * ROW_DOUBLES + 288 + r];
    return;
  }
  if (j >= npairs) return;
  double x[24];
#pragma unroll
  for (int r = 0; r < 12; ++r) { x[r] = rows[(long)(2 * j) * ROW_DOUBLES + 288 + r]; x[12 + r] = rows[(long)(2 * j + 1) * ROW_DOUBLES + 288 + r]; }
  const int mid = (2 * j + 1) << level;
  double* rec = a.rec + ((long)b * a.n_nodes + mid) * REC_DOUBLES;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const double* v = rec + REC_V + k * 24;
    double w = x[k];
#pragma unroll
    for (int r = k + 1; r < 24; ++r) w = __builtin_fma(v[r], x[r], w);
    w *= rec[REC_TAU + k];
    x[k] -= w;
#pragma unroll
    for (int r = k + 1; r < 24; ++r) x[r] = __builtin_fma(-w, v[r], x[r]);
  }
#pragma unroll
  for (int r = 0; r < 12; ++r) { rec[REC_G + r] = x[r]; out[(long)j * ROW_DOUBLES + 288 + r] = x[12 + r]; }
}

// Last level: one row  A d_first + B d_last = r  with d_first[0:6] = d_last[0:6] = 0  ->  12 x 12 system for the
// two end-node costate updates (Gaussian elimination with partial pivoting).  One lane per trajectory.
__global__ __launch_bounds__(64) void k_bvp_final(BvpArgs a, const double* __restrict__ cur) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.n_batch) return;
  const double* row = cur + (long)b * a.S_traj * ROW_DOUBLES;
  double Mx[12][13];
  for (int r = 0; r < 12; ++r) {
    for (int c = 0; c < 6; ++c) { Mx[r][c] = row[(6 + c) * 12 + r]; Mx[r][6 + c] = row[144 + (6 + c) * 12 + r]; }
    Mx[r][12] = row[288 + r];
  }
  for (int k = 0; k < 12; ++k) {
    int piv = k;
    double best = fabs(Mx[k][k]);
    for (int r = k + 1; r < 12; ++r) if (fabs(Mx[r][k]) > best) { best = fabs(Mx[r][k]); piv = r; }
    if (piv != k) for (int c = k; c < 13; ++c) { const double t = Mx[k][c]; Mx[k][c] = Mx[piv][c]; Mx[piv][c] = t; }
    const double inv = 1.0 / Mx[k][k];
    for (int r = k + 1; r < 12; ++r) {
      const double f = Mx[r][k] * inv;
      for (int c = k + 1; c < 13; ++c) Mx[r][c] -= f * Mx[k][c];
    }
  }
  double x[12];
  for (int k = 11; k >= 0; --k) {
    double s = Mx[k][12];
    for (int c = k + 1; c < 12; ++c) s -= Mx[k][c] * x[c];
    x[k] = s / Mx[k][k];
  }
  const long n0 = (long)b * a.n_nodes, n1 = n0 + a.n_nodes - 1;
  for (int c = 0; c < 6; ++c) {
    a.delta[(long)c * a.ldx + n0] = 0.0;
    a.delta[(long)c * a.ldx + n1] = 0.0;
    a.delta[(long)(6 + c) * a.ldx + n0] = x[c];
    a.delta[(long)(6 + c) * a.ldx + n1] = x[6 + c];
  }
}

// Back-substitution at one level: d_mid = R^{-1} (g - Ca d_left - Cb d_right).  One lane per pair.
__global__ __launch_bounds__(64) void k_bvp_backsub(BvpArgs a, int level, int M) {
  const int j = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (j >= M / 2) return;
  const int mid = (2 * j + 1) << level, left = (2 * j) << level;
  int right = (2 * j + 2) << level;
  if (right > a.n_nodes - 1) right = a.n_nodes - 1;
  const double* rec = a.rec + ((long)b * a.n_nodes + mid) * REC_DOUBLES;
  const long nb = (long)b * a.n_nodes;
  double dl[12], dr[12], x[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) { dl[c] = a.delta[(long)c * a.ldx + nb + left]; dr[c] = a.delta[(long)c * a.ldx + nb + right]; }
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    double s = rec[REC_G + r];
#pragma unroll
    for (int c = 0; c < 12; ++c) s -= rec[REC_CA + c * 12 + r] * dl[c] + rec[REC_CB + c * 12 + r] * dr[c];
    x[r] = s;
  }
#pragma unroll
  for (int k = 11; k >= 0; --k) {
    double s = x[k];
#pragma unroll
    for (int c = k + 1; c < 12; ++c) s -= rec[REC_R + c * 12 + k] * x[c];
    x[k] = s / rec[REC_R + k * 12 + k];
  }
#pragma unroll
  for (int c = 0; c < 12; ++c) a.delta[(long)c * a.ldx + nb + mid] = x[c];
}

// y = x + alpha * d  (elementwise over an SoA [rows][ld] block): trial points, SOC accumulation
__global__ __launch_bounds__(256) void k_axpy(const double* __restrict__ x, const double* __restrict__ d, double alpha,
                                              double* __restrict__ y, long count) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) y[i] = __builtin_fma(alpha, d[i], x[i]);
}

size_t bvp_workspace_doubles(int n_nodes, int n_batch) {
  return (size_t)2 * (n_nodes - 1) * n_batch * ROW_DOUBLES + (size_t)n_nodes * n_batch * REC_DOUBLES;
}

// Factor (if Phi != null) or re-apply to a new rhs (Phi == null), then solve: delta[12][ldx] (SoA, node-indexed).
hipError_t launch_bvp_solve(const double* Phi, long ldp, const double* defect, long ldd, int n_nodes, int n_batch,
                            double* workspace, double* delta, long ldx, hipStream_t st) {
  BvpArgs a;
  a.n_nodes = n_nodes; a.n_batch = n_batch; a.S_traj = n_nodes - 1;
  const size_t rows_sz = (size_t)a.S_traj * n_batch * ROW_DOUBLES;
  a.rows0 = workspace; a.rows1 = workspace + rows_sz; a.rec = workspace + 2 * rows_sz;
  a.delta = delta; a.ldx = ldx;
  const long S_total = (long)a.S_traj * n_batch;
  if (Phi) {
    const long cnt = S_total * ROW_DOUBLES;
    hipLaunchKernelGGL(k_bvp_init, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, Phi, ldp, defect, ldd, a);
  } else {
    const long cnt = S_total * 12;
    hipLaunchKernelGGL(k_bvp_init_rhs, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, defect, ldd, a);
  }
  double* cur = a.rows0;
  double* nxt = a.rows1;
  int M = a.S_traj, level = 0;
  int Ms[40];
  while (M > 1) {
    Ms[level] = M;
    const int npairs = M / 2, carry = M & 1;
    if (Phi) hipLaunchKernelGGL(k_bvp_reduce, dim3(npairs + carry, n_batch), dim3(64), 0, st, a, level, M, cur, nxt);
    else hipLaunchKernelGGL(k_bvp_reduce_rhs, dim3((npairs + carry + 63) / 64, n_batch), dim3(64), 0, st, a, level, M, cur, nxt);
    double* t = cur; cur = nxt; nxt = t;
    M = npairs + carry;
    ++level;
  }
  hipLaunchKernelGGL(k_bvp_final, dim3((n_batch + 63) / 64), dim3(64), 0, st, a, cur);
  for (int l = level - 1; l >= 0; --l) {
    const int npairs = Ms[l] / 2;
    hipLaunchKernelGGL(k_bvp_backsub, dim3((npairs + 63) / 64, n_batch), dim3(64), 0, st, a, l, Ms[l]);
  }
  return hipGetLastError();
}

hipError_t launch_axpy(const double* x, const double* d, double alpha, double* y, long count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_axpy, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, x, d, alpha, y, count);
  return hipGetLastError();
}

}  // namespace lto
