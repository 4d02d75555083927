// kernels_util.hip -- layout and reduction kernels around the propagators (HBM-bound, trivial work).
//
//  pack / unpack : Julia column-major [ndim x count] (node-contiguous AoS, 8*ndim bytes per node)
//                  <-> component-major SoA [ndim][ld].  A tile of up to 256 nodes is staged through LDS so both
//                  the HBM reads and the HBM writes are unit-stride across the wavefront.
//  defect_norms  : per trajectory sum(defect.^2) (line-search cost, multiShoot_CRTBP_indirect.jl:240,
//                  multiShoot_CRTBP_direct.jl:424) and max|defect| (convergence test, :331 / :588).
#include <algorithm>
#include "kernels.hpp"

namespace lto {

constexpr int THREADS = 256;
constexpr int MAXDIM = 512;
constexpr int LDS_BUDGET_DOUBLES = 8192;  // 64 KiB tile

struct PackJob { const double* aos; double* soa; long count, ld; int ndim, tn, tiles; };

// nodes per tile: as many as fit the LDS budget, at most 256, a multiple of 32 when possible -- and fewer when that leaves
// the launch under 512 workgroups: the AoS side may be page-locked host memory behind the link (lto_api.hip, stage_in /
// stage_out), where every trip of a thread's copy loop is a round trip of ~2 us, so a small batch wants one or two elements
// per thread and its tiles spread over the chip (12 x 4 097 doubles read from the host: 20.5 us with 17 tiles of 256 nodes).
static int tile_nodes(int ndim, long count) {
  int n = LDS_BUDGET_DOUBLES / (ndim + 1);
  if (n > 256) n = 256;
  if (n >= 32) n &= ~31;
  while (n >= 64 && (count + n - 1) / n < 512) n >>= 1;
  return n < 1 ? 1 : n;
}

// LDS tile [tn][ndim+1]: the +1 pad breaks the regular row stride (bank conflicts).
__device__ __forceinline__ void pack_tile(double* tile, const double* __restrict__ aos, int ndim, long count, double* __restrict__ soa,
                                          long ld, int tn, long tile_id) {
  const long j0 = tile_id * tn;
  const int nj = (int)((count - j0 < tn) ? (count - j0) : tn);
  const int pitch = ndim + 1;
  const double* src = aos + j0 * ndim;
  for (int e = threadIdx.x; e < nj * ndim; e += THREADS) {  // unit-stride HBM read
    const int j = e / ndim, c = e - j * ndim;
    tile[j * pitch + c] = src[e];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nj * ndim; e += THREADS) {  // runs of nj contiguous doubles per component
    const int c = e / nj, j = e - c * nj;
    soa[c * ld + j0 + j] = tile[j * pitch + c];
  }
}

__device__ __forceinline__ void unpack_tile(double* tile, const double* __restrict__ soa, long ld, int ndim, long count,
                                            double* __restrict__ aos, int tn, long tile_id) {
  const long j0 = tile_id * tn;
  const int nj = (int)((count - j0 < tn) ? (count - j0) : tn);
  const int pitch = ndim + 1;
  for (int e = threadIdx.x; e < nj * ndim; e += THREADS) {
    const int c = e / nj, j = e - c * nj;
    tile[j * pitch + c] = soa[c * ld + j0 + j];
  }
  __syncthreads();
  double* dst = aos + j0 * ndim;
  for (int e = threadIdx.x; e < nj * ndim; e += THREADS) {
    const int j = e / ndim, c = e - j * ndim;
    dst[e] = tile[j * pitch + c];
  }
}

__global__ __launch_bounds__(THREADS) void k_pack(const double* __restrict__ aos, int ndim, long count,
                                                  double* __restrict__ soa, long ld, int tn) {
  extern __shared__ double tile[];
  pack_tile(tile, aos, ndim, count, soa, ld, tn, blockIdx.x);
}

__global__ __launch_bounds__(THREADS) void k_unpack(const double* __restrict__ soa, long ld, int ndim, long count,
                                                    double* __restrict__ aos, int tn) {
  extern __shared__ double tile[];
  unpack_tile(tile, soa, ld, ndim, count, aos, tn, blockIdx.x);
}

// Two layout jobs in one launch (the host-pointer API's page-locked path: node array + time grid in, STM + defect out;
// behind the link every launch is a few microseconds of latency that two jobs can share).  Workgroups [0, a.tiles) serve
// job a, the rest job b.
__global__ __launch_bounds__(THREADS) void k_pack2(const PackJob a, const PackJob b) {
  extern __shared__ double tile[];
  if ((int)blockIdx.x < a.tiles) pack_tile(tile, a.aos, a.ndim, a.count, a.soa, a.ld, a.tn, blockIdx.x);
  else pack_tile(tile, b.aos, b.ndim, b.count, b.soa, b.ld, b.tn, (long)blockIdx.x - a.tiles);
}

__global__ __launch_bounds__(THREADS) void k_unpack2(const PackJob a, const PackJob b) {
  extern __shared__ double tile[];
  if ((int)blockIdx.x < a.tiles) unpack_tile(tile, a.soa, a.ld, a.ndim, a.count, const_cast<double*>(a.aos), a.tn, blockIdx.x);
  else unpack_tile(tile, b.soa, b.ld, b.ndim, b.count, const_cast<double*>(b.aos), b.tn, (long)blockIdx.x - a.tiles);
}

// One workgroup per trajectory; NaN-propagating max (a NaN defect must surface, status_flag = 2 path).
// 1 024 lanes, every lane fetches the ndim rows of a segment before it uses any of them (round 3: 256 lanes, one load in flight
// each: 26 us for the 12 x 4 096 defect block of one trajectory, three times per Newton iteration).
constexpr int NORM_THREADS = 1024;
__global__ __launch_bounds__(NORM_THREADS) void k_defect_norms(const double* __restrict__ defect, long ldd, int ndim,
                                                               int seg_per_traj, double* __restrict__ sumsq,
                                                               double* __restrict__ maxabs) {
  const int b = blockIdx.x;
  double ss0 = 0.0, ss1 = 0.0, mx = 0.0;
  bool bad = false;
  const double* base = defect + (long)b * seg_per_traj;
  constexpr int NR = 16;               // rows fetched together (the sweeps' blocks have 6 .. 14)
  for (int i = threadIdx.x; i < seg_per_traj; i += NORM_THREADS) {
    for (int c0 = 0; c0 < ndim; c0 += NR) {
      double v[NR];
#pragma unroll
      for (int c = 0; c < NR; ++c) v[c] = (c0 + c < ndim) ? base[(long)(c0 + c) * ldd + i] : 0.0;
#pragma unroll
      for (int c = 0; c < NR; c += 2) {
        ss0 = __builtin_fma(v[c], v[c], ss0);
        ss1 = __builtin_fma(v[c + 1], v[c + 1], ss1);
      }
#pragma unroll
      for (int c = 0; c < NR; ++c) { bad |= (v[c] != v[c]); mx = fmax(mx, fabs(v[c])); }
    }
  }
  double ss = ss0 + ss1;
  __shared__ double s_ss[NORM_THREADS / 64], s_mx[NORM_THREADS / 64];
  __shared__ int s_bad[NORM_THREADS / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_xor(ss, o);
    mx = fmax(mx, __shfl_xor(mx, o));
    bad |= (bool)__shfl_xor((int)bad, o);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_ss[w] = ss; s_mx[w] = mx; s_bad[w] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0, m = 0.0;
    bool any_bad = false;
#pragma unroll
    for (int k = 0; k < NORM_THREADS / 64; ++k) { t += s_ss[k]; m = fmax(m, s_mx[k]); any_bad |= (bool)s_bad[k]; }
    if (sumsq) sumsq[b] = t;
    if (maxabs) maxabs[b] = any_bad ? __builtin_nan("") : m;
  }
}

// ---- load balancing of adaptive sweeps: segment order by the step counts of the previous sweep, heaviest first.
// Counting sort on key = min(accepted + rejected, ORDER_BINS - 1): histogram, descending exclusive scan, scatter.
// The order inside a bin is whatever the atomics produce; nothing depends on it (each segment's result is its own).
constexpr int ORDER_BINS = 1024;

__device__ __forceinline__ int order_key(const int* nacc, const int* nrej, int s) {
  const int k = nacc[s] + nrej[s];
  return k < 0 ? 0 : (k >= ORDER_BINS ? ORDER_BINS - 1 : k);
}

__global__ __launch_bounds__(256) void k_order_hist(const int* nacc, const int* nrej, int S, int* bins) {
  __shared__ int h[ORDER_BINS];
  for (int i = threadIdx.x; i < ORDER_BINS; i += 256) h[i] = 0;
  __syncthreads();
  for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) atomicAdd(&h[order_key(nacc, nrej, s)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < ORDER_BINS; i += 256)
    if (h[i]) atomicAdd(&bins[i], h[i]);
}

// bins[k] <- number of segments with a key > k (start offset of bin k in descending order); one workgroup.
__global__ __launch_bounds__(ORDER_BINS) void k_order_scan(int* bins) {
  __shared__ int a[ORDER_BINS];
  const int i = threadIdx.x;                    // position i holds key ORDER_BINS-1-i
  const int mine = bins[ORDER_BINS - 1 - i];
  a[i] = mine;
  __syncthreads();
  for (int off = 1; off < ORDER_BINS; off <<= 1) {
    const int v = (i >= off) ? a[i - off] : 0;
    __syncthreads();
    a[i] += v;
    __syncthreads();
  }
  bins[ORDER_BINS - 1 - i] = a[i] - mine;
}

// Scatter with one global atomic per (workgroup, chunk, key present): every lane takes its rank inside the chunk from an LDS
// counter, one lane per key reserves the chunk's range.  Round 4: a line search's 81 920 segments have about ten distinct keys, and
// one global atomic per segment on ten addresses took ~0.5 ms.
__global__ __launch_bounds__(256) void k_order_scatter(const int* nacc, const int* nrej, int S, int* cursor, int* order) {
  __shared__ int cnt[ORDER_BINS];
  __shared__ int base[ORDER_BINS];
  for (int i = threadIdx.x; i < ORDER_BINS; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int s0 = blockIdx.x * 256; s0 < S; s0 += gridDim.x * 256) {      // uniform per workgroup
    const int s = s0 + threadIdx.x;
    int key = -1, rank = 0;
    if (s < S) { key = order_key(nacc, nrej, s); rank = atomicAdd(&cnt[key], 1); }
    __syncthreads();
    if (key >= 0 && rank == 0) base[key] = atomicAdd(&cursor[key], cnt[key]);     // the key's first lane reserves for all of them
    __syncthreads();
    if (key >= 0) {
      const int pos = base[key] + rank;
      if (pos >= 0 && pos < S) order[pos] = s;  // always true for a consistent histogram; keeps a stale one harmless
    }
    __syncthreads();
    if (key >= 0 && rank == 0) cnt[key] = 0;
    __syncthreads();
  }
}

hipError_t launch_segment_order(const int* nacc, const int* nrej, int S, int* bins, int* order, hipStream_t st) {
  if (S <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(bins, 0, sizeof(int) * ORDER_BINS, st);
  if (e != hipSuccess) return e;
  int blocks = (S + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_order_hist, dim3(blocks), dim3(256), 0, st, nacc, nrej, S, bins);
  hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_BINS), 0, st, bins);
  hipLaunchKernelGGL(k_order_scatter, dim3(blocks), dim3(256), 0, st, nacc, nrej, S, bins, order);
  return hipGetLastError();
}

// ---- windowed order (kernels.hpp LTO_ORDER_WINDOW): three small launches.
// (1) one workgroup per window: counting sort of its <= 1 024 segments by step count, heaviest first, into `local` (window-major);
//     the window's key = its largest step count.
__global__ __launch_bounds__(LTO_ORDER_WINDOW) void k_order_window(const int* nacc, const int* nrej, int S, int W, int* local, int* wkey) {
  static_assert(LTO_ORDER_WINDOW == ORDER_BINS, "one thread per bin in the scan");
  __shared__ int h[ORDER_BINS];
  __shared__ int a[ORDER_BINS];
  __shared__ int top;
  const int i = threadIdx.x;                    // 1 024 threads: one per bin; the first W of them also own a segment
  const int s = blockIdx.x * W + i;
  h[i] = 0;
  if (i == 0) top = 0;
  __syncthreads();
  int key = -1, rank = 0;
  if (i < W && s < S) { key = order_key(nacc, nrej, s); rank = atomicAdd(&h[key], 1); atomicMax(&top, key); }
  __syncthreads();
  const int mine = h[ORDER_BINS - 1 - i];       // position i holds key ORDER_BINS-1-i: descending
  a[i] = mine;
  __syncthreads();
  for (int off = 1; off < ORDER_BINS; off <<= 1) {
    const int v = (i >= off) ? a[i - off] : 0;
    __syncthreads();
    a[i] += v;
    __syncthreads();
  }
  h[ORDER_BINS - 1 - i] = a[i] - mine;          // start of bin (ORDER_BINS-1-i) inside the window
  __syncthreads();
  if (key >= 0) local[blockIdx.x * W + h[key] + rank] = s;
  if (i == 0) wkey[blockIdx.x] = top;
}
// (2) one workgroup: rank the windows by key (counting sort, heaviest first; a short last window ranks last) and give each its
//     place: the window of rank r goes to XCD r mod 8 as that XCD's (r / 8)-th window, the XCDs' lists one after the other.
//     Inside a list the full windows are INTERLEAVED in groups of `weave` at the granularity of 16 segments (a wavefront of the
//     four-lane defect kernel, a workgroup of the cooperative kernel): first the heaviest 16 segments of every window of the group,
//     then their second-heaviest 16, ... -- the workgroups an XCD starts first are then the heavy ones of ALL its windows (longest
//     processing time first: what keeps the sweep from ending on a heavy wavefront that started late), while the group's windows
//     (weave x ~200 KB of nodes and defects) stay inside the XCD's 4 MB of L2.
//     wdst[w] = first position of w's group (or of w itself: the short window, never interleaved); winfo[w] = group size << 8 | slot.
__global__ __launch_bounds__(ORDER_BINS) void k_order_place(const int* wkey, int nwin, int S, int W, int weave, int* wdst, int* winfo) {
  // weave = 0: every window of an XCD's list in ONE group up to 16 windows (a sweep of one or two rounds must not end on the heavy
  // wavefronts of a small last group: the line search's 80 windows took 85 instead of 67 us with groups of 8 + 2), groups of 16 beyond
  if (weave <= 0) { const int nx = (nwin + LTO_XCDS - 1) / LTO_XCDS; weave = nx < 16 ? (nx < 1 ? 1 : nx) : 16; }
  __shared__ int h[ORDER_BINS];
  __shared__ int a[ORDER_BINS];
  const int i = threadIdx.x;
  const bool has_short = (S % W) != 0;
  const int nfull = has_short ? nwin - 1 : nwin;              // full windows are ranked 0 .. nfull-1, the short one is rank nwin-1
  h[i] = 0;
  __syncthreads();
  for (int w = i; w < nfull; w += ORDER_BINS) atomicAdd(&h[wkey[w]], 1);
  __syncthreads();
  const int mine = h[ORDER_BINS - 1 - i];
  a[i] = mine;
  __syncthreads();
  for (int off = 1; off < ORDER_BINS; off <<= 1) {
    const int v = (i >= off) ? a[i - off] : 0;
    __syncthreads();
    a[i] += v;
    __syncthreads();
  }
  h[ORDER_BINS - 1 - i] = a[i] - mine;          // first rank of the windows with key (ORDER_BINS-1-i); bumped as ranks are handed out
  __syncthreads();
  const int xs = (nwin - 1) % LTO_XCDS;          // the XCD whose list ends with the short window (if there is one)
  const int deficit = nwin * W - S;
  auto list_len = [&](const int x) { return (nwin - x + LTO_XCDS - 1) / LTO_XCDS; };        // windows with rank = x mod 8
  auto list_start = [&](const int x) {
    int before = 0;
    for (int q = 0; q < x; ++q) before += list_len(q) * W - ((has_short && q == xs) ? deficit : 0);
    return before;
  };
  for (int w = i; w < nwin; w += ORDER_BINS) {
    const bool is_short = has_short && w == nwin - 1;
    const int r = is_short ? nwin - 1 : atomicAdd(&h[wkey[w]], 1);
    const int x = r % LTO_XCDS, j = r / LTO_XCDS;
    const int nf = list_len(x) - ((has_short && x == xs) ? 1 : 0);                           // full windows of this list
    if (is_short) { wdst[w] = list_start(x) + nf * W; winfo[w] = 0; continue; }
    const int g = j / weave, m = j % weave;
    const int mg = (nf - g * weave) < weave ? (nf - g * weave) : weave;
    wdst[w] = list_start(x) + g * weave * W;
    winfo[w] = (mg << 8) | m;
  }
}
// (3) one workgroup per window: its local order to its place
__global__ __launch_bounds__(256) void k_order_copy(const int* local, const int* wdst, const int* winfo, int S, int W, int* order) {
  const int w = blockIdx.x;
  const int n = (S - w * W) < W ? (S - w * W) : W;
  const int dst = wdst[w], info = winfo[w];
  const int mg = info >> 8, m = info & 255;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int pos = (mg == 0) ? dst + i : dst + ((i >> 4) * mg + m) * 16 + (i & 15);
    if (pos >= 0 && pos < S) order[pos] = local[w * W + i];
  }
}

hipError_t launch_segment_order_windowed(const int* nacc, const int* nrej, int S, int weave, int* work, int* order, hipStream_t st) {
  if (S <= 0) return hipSuccess;
  const int W = order_window_size(S);
  const int nwin = (S + W - 1) / W;
  if (weave < 0 || weave > 255) return hipErrorInvalidValue;      // 0 = choose (k_order_place)
  int* local = work;
  int* wkey = work + S;
  int* wdst = wkey + nwin;
  int* winfo = wdst + nwin;
  hipLaunchKernelGGL(k_order_window, dim3(nwin), dim3(LTO_ORDER_WINDOW), 0, st, nacc, nrej, S, W, local, wkey);
  hipLaunchKernelGGL(k_order_place, dim3(1), dim3(ORDER_BINS), 0, st, wkey, nwin, S, W, weave, wdst, winfo);
  hipLaunchKernelGGL(k_order_copy, dim3(nwin), dim3(256), 0, st, local, wdst, winfo, S, W, order);
  return hipGetLastError();
}

// ---- driver-loop helpers (lto_indirect_solve[_batch]): SoA in / SoA out, node index j = b*n + k (trajectory b, node k)
// Xt[c][(b*na + a)*n + k] = X[c][b*n + k] + alphas[a] d[c][b*n + k]: the na trial trajectories of lineSearch
// (indirect.jl:227-233) of every trajectory of the batch
__global__ __launch_bounds__(256) void k_trial_points(const double* X, const double* d, long ld, int ndim, int n, int nb, int na,
                                                      const double* alphas, double* Xt, long ldt) {
  const long total = (long)ndim * nb * na * n;
  for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += gridDim.x * 256L) {
    const int k = (int)(q % n);
    const long r = q / n;
    const int a = (int)(r % na);
    const int b = (int)((r / na) % nb);
    const int c = (int)(r / ((long)na * nb));
    const long src = c * ld + (long)b * n + k;
    Xt[c * ldt + ((long)b * na + a) * n + k] = __builtin_fma(alphas[a], d[src], X[src]);
  }
}

// y[c][b*n + k] = x[c][b*n + k] + alpha[b] d[c][b*n + k]  (per-trajectory step length; alpha = 0 freezes a trajectory)
__global__ __launch_bounds__(256) void k_axpy_traj(const double* x, const double* d, const double* alpha, double* y, long ld, int ndim,
                                                   int n, int nb) {
  const long per = (long)nb * n;
  const long total = (long)ndim * per;
  for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += gridDim.x * 256L) {
    const long j = q % per;
    const int c = (int)(q / per);
    const long idx = c * ld + j;
    const double al = alpha[j / n];
    y[idx] = (al == 0.0) ? x[idx] : __builtin_fma(al, d[idx], x[idx]);      // a frozen trajectory keeps its values whatever d holds (NaN x 0 is NaN)
  }
}

// Decisions of the Newton loop taken on the device (lto_indirect_solve_batch), so that the host reads back once per iteration:
// second-order-correction mask  step[b] = 1 if trajectory b is active and max |xc_update| < thr (indirect.jl:190), else 0
__global__ void k_soc_mask(const double* mx, const double* act, double thr, double* step, int nb) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) step[b] = (act[b] != 0.0 && mx[b] == mx[b] && mx[b] < thr) ? 1.0 : 0.0;
}
// step length  step[b] = alphas[first minimiser of ss[b*na .. ]] where the line search is on (lineSearch, :244-245: `alpha[er .==
// minimum(er)][1]`, the comparison `<` skips NaN trials exactly as the host loop did), 1 for the other active trajectories, 0 for frozen ones.
// act / search == nullptr: every trajectory searches.  With `mxt` (max |defect| of every trial trajectory) the chosen trial's maximum
// goes to mx[b]: the defect check after the update (:328-331) is the line search's own sweep at that trial point (k_take_trial).
__device__ inline int first_minimiser(const double* e, int na) {
  int best = 0;
  for (int a = 1; a < na; ++a) if (e[a] < e[best]) best = a;
  return best;
}
__global__ void k_pick_alpha(const double* ss, const double* alphas, int na, const double* act, const double* search, double* step, int nb,
                             const double* mxt, double* mx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const bool on = !act || act[b] != 0.0;
  double s = on ? 1.0 : 0.0;
  int chosen = -1;
  if (on && (!search || search[b] != 0.0)) {
    chosen = first_minimiser(ss + (long)b * na, na);
    s = alphas[chosen];
  }
  step[b] = s;
  if (mxt && mx && chosen >= 0) mx[b] = mxt[(long)b * na + chosen];
}
// defect[c][b*seg + i] = trial[c][(b*na + chosen)*seg + i] for the trajectories whose line search chose a trial point: XC_all +
// xc_update*alpha pinned at its end states (:304, :324-325) IS that trial point (the update's end-state rows are zero and both are one
// fma), so its defect sweep (:328) would repeat the trial's lane for lane.
__global__ __launch_bounds__(256) void k_take_trial(const double* __restrict__ trial, long ldt, const double* __restrict__ ss,
                                                    const double* __restrict__ act, const double* __restrict__ search, int na, int seg,
                                                    int ndim, int nb, double* __restrict__ defect, long ldd, const double* __restrict__ alphas,
                                                    double* __restrict__ step, const double* __restrict__ mxt, double* __restrict__ mx) {
  for (int b = blockIdx.y; b < nb; b += gridDim.y) {
    const bool on = !act || act[b] != 0.0;
    const bool chosen = on && (!search || search[b] != 0.0);
    const int a = chosen ? first_minimiser(ss + (long)b * na, na) : 0;
    if (step && blockIdx.x == 0 && threadIdx.x == 0) {      // k_pick_alpha's part, when this launch stands for both
      step[b] = chosen ? alphas[a] : (on ? 1.0 : 0.0);
      if (mxt && mx && chosen) mx[b] = mxt[(long)b * na + a];
    }
    if (!chosen) continue;
    const long src = ((long)b * na + a) * seg, dst = (long)b * seg;
    const long total = (long)ndim * seg;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += gridDim.x * 256L) {
      const int c = (int)(q / seg);
      const int i = (int)(q % seg);
      defect[c * ldd + dst + i] = trial[c * ldt + src + i];
    }
  }
}
// The iteration's scalars straight into page-locked host memory (no copy-engine operation, no stream synchronisation): the host
// polls the sequence word, which is written last behind a system-scope fence.
__global__ void k_iter_report(const double* a, int na_, const double* b, int nb_, volatile double* host, volatile long long* seq_word, long long seq) {
  const int i = threadIdx.x;
  for (int k = i; k < na_; k += blockDim.x) host[k] = a[k];
  for (int k = i; k < nb_; k += blockDim.x) host[na_ + k] = b[k];
  __threadfence_system();
  __syncthreads();
  if (i == 0) *seq_word = seq;
}
hipError_t launch_soc_mask(const double* mx, const double* act, double thr, double* step, int nb, hipStream_t st) {
  if (nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_soc_mask, dim3((nb + 255) / 256), dim3(256), 0, st, mx, act, thr, step, nb);
  return hipGetLastError();
}
hipError_t launch_pick_alpha(const double* ss, const double* alphas, int na, const double* act, const double* search, double* step, int nb,
                             const double* mxt, double* mx, hipStream_t st) {
  if (nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pick_alpha, dim3((nb + 255) / 256), dim3(256), 0, st, ss, alphas, na, act, search, step, nb, mxt, mx);
  return hipGetLastError();
}
hipError_t launch_take_trial(const double* trial, long ldt, const double* ss, const double* act, const double* search, int na, int seg,
                             int ndim, int nb, double* defect, long ldd, const double* alphas, double* step, const double* mxt, double* mx,
                             hipStream_t st) {
  if (nb <= 0 || seg <= 0) return hipSuccess;
  long blocks = ((long)ndim * seg + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(k_take_trial, dim3((unsigned)blocks, (unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, st, trial, ldt, ss, act, search, na, seg, ndim, nb, defect, ldd, alphas, step, mxt, mx);
  return hipGetLastError();
}
hipError_t launch_iter_report(const double* a, int na, const double* b, int nb, double* host_dev, long long* seq_dev, long long seq, hipStream_t st) {
  hipLaunchKernelGGL(k_iter_report, dim3(1), dim3(256), 0, st, a, na, b, nb, (volatile double*)host_dev, (volatile long long*)seq_dev, seq);
  return hipGetLastError();
}

// save (dir = 0) or restore (dir = 1) the first `nrow` rows of node 0 and node n-1 of every trajectory
// (indirect.jl:270-271, :324-325); saved [nb][2 nrow]
__global__ void k_end_states(double* X, long ld, int n, int nrow, double* saved, int dir) {
  const int b = blockIdx.x;
  const int r = threadIdx.x;
  if (r >= 2 * nrow) return;
  const long idx = (long)(r % nrow) * ld + (long)b * n + (r < nrow ? 0 : n - 1);
  double* sv = saved + (long)b * 2 * nrow;
  if (dir) X[idx] = sv[r]; else sv[r] = X[idx];
}

hipError_t launch_trial_points(const double* X, const double* d, long ld, int ndim, int n, int nb, int na, const double* alphas,
                               double* Xt, long ldt, hipStream_t st) {
  const long total = (long)ndim * nb * na * n;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_trial_points, dim3((unsigned)blocks), dim3(256), 0, st, X, d, ld, ndim, n, nb, na, alphas, Xt, ldt);
  return hipGetLastError();
}

hipError_t launch_axpy_traj(const double* x, const double* d, const double* alpha, double* y, long ld, int ndim, int n, int nb,
                            hipStream_t st) {
  const long total = (long)ndim * nb * n;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_axpy_traj, dim3((unsigned)blocks), dim3(256), 0, st, x, d, alpha, y, ld, ndim, n, nb);
  return hipGetLastError();
}

hipError_t launch_end_states(double* X, long ld, int n, int nb, int nrow, double* saved, int restore, hipStream_t st) {
  if (nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_end_states, dim3(nb), dim3(64), 0, st, X, ld, n, nrow, saved, restore);
  return hipGetLastError();
}

// Sum and maximum of the trial steps of the last adaptive sweep, straight into page-locked host memory (no copy, no
// synchronisation): the plan's NEXT defect sweep reads them -- whatever has arrived by then -- to tell a workload whose segments
// all take about the same number of steps (throughput-bound: fewer lanes per segment) from one with a long tail (the slowest
// segment sets the time: more lanes per segment).  acc[0] = sum, acc[1] = max, acc[2] = blocks done (device scratch, left zeroed).
__global__ __launch_bounds__(1024) void k_step_stats(const int* __restrict__ nacc, const int* __restrict__ nrej, int S, unsigned long long* acc,
                                                     volatile long long* host_out) {
  long long sum = 0;
  int mx = 0;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < S; i += gridDim.x * 1024) {
    const int k = nacc[i] + nrej[i];
    sum += k;
    mx = k > mx ? k : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    const int om = __shfl_xor(mx, o);
    mx = om > mx ? om : mx;
  }
  __shared__ long long s_sum[16];
  __shared__ int s_mx[16];
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_mx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {               // one pair of atomics per block (1 024 contended atomics on two addresses took ~30 us)
    long long bs = 0;
    int bm = 0;
    for (int k = 0; k < 16; ++k) { bs += s_sum[k]; bm = s_mx[k] > bm ? s_mx[k] : bm; }
    atomicAdd(&acc[0], (unsigned long long)bs);
    atomicMax(&acc[1], (unsigned long long)bm);
    __threadfence();
    if (atomicAdd(&acc[2], 1ull) + 1ull == gridDim.x) {            // last block: publish and reset
      __threadfence();
      const unsigned long long s = atomicExch(&acc[0], 0ull), m = atomicExch(&acc[1], 0ull);
      acc[2] = 0ull;
      host_out[1] = (long long)m;
      host_out[2] = (long long)S;
      __threadfence_system();
      host_out[0] = (long long)s;                                   // the sum last: a reader that sees it sees a whole record
    }
  }
}
hipError_t launch_step_stats(const int* nacc, const int* nrej, int S, unsigned long long* acc, long long* host_out, hipStream_t st) {
  if (S <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)((S + 4095) / 4096 > 64 ? 64 : (S + 4095) / 4096);
  hipLaunchKernelGGL(k_step_stats, dim3(blocks), dim3(1024), 0, st, nacc, nrej, S, acc, (volatile long long*)host_out);
  return hipGetLastError();
}

// Node records of the staged (rebalanced) adaptive sweeps: 64 nodes per workgroup through an LDS tile, unit stride on both sides.
__global__ __launch_bounds__(256) void k_node_records(const double* __restrict__ X, long ldx, const double* __restrict__ t, int t_stride,
                                                      int n_nodes, long J, double* __restrict__ Xa) {
  __shared__ double tile[64][NODE_REC + 1];
  const long j0 = (long)blockIdx.x * 64;
  for (int q = threadIdx.x; q < 64 * 13; q += 256) {
    const int c = q >> 6, i = q & 63;
    const long j = j0 + i;
    double v = 0.0;
    if (j < J) {
      if (c < 12) v = X[c * ldx + j];
      else { const long b = j / n_nodes; v = t[b * t_stride + (j - b * n_nodes)]; }
    }
    tile[i][c] = v;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * NODE_REC; q += 256) {
    const int i = q / NODE_REC, c = q % NODE_REC;
    if (j0 + i < J) Xa[(j0 + i) * NODE_REC + c] = (c < 13) ? tile[i][c] : 0.0;
  }
}
hipError_t launch_node_records(const double* X, long ldx, const double* t, int t_stride, int n_nodes, long J, double* Xa, hipStream_t st) {
  if (J <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_node_records, dim3((unsigned)((J + 63) / 64)), dim3(256), 0, st, X, ldx, t, t_stride, n_nodes, J, Xa);
  return hipGetLastError();
}

hipError_t launch_pack_soa(const double* aos, int ndim, long count, double* soa, long ld, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  if (ndim < 1 || ndim > MAXDIM) return hipErrorInvalidValue;
  const int tn = tile_nodes(ndim, count);
  const unsigned blocks = (unsigned)((count + tn - 1) / tn);
  hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(THREADS), sizeof(double) * tn * (ndim + 1), st, aos, ndim, count, soa, ld, tn);
  return hipGetLastError();
}

hipError_t launch_unpack_soa(const double* soa, long ld, int ndim, long count, double* aos, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  if (ndim < 1 || ndim > MAXDIM) return hipErrorInvalidValue;
  const int tn = tile_nodes(ndim, count);
  const unsigned blocks = (unsigned)((count + tn - 1) / tn);
  hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(THREADS), sizeof(double) * tn * (ndim + 1), st, soa, ld, ndim, count, aos, tn);
  return hipGetLastError();
}

static bool pack_job(PackJob& j, const double* aos, int ndim, long count, double* soa, long ld) {
  if (ndim < 1 || ndim > MAXDIM || count <= 0) return false;
  j.aos = aos; j.soa = soa; j.ndim = ndim; j.count = count; j.ld = ld;
  j.tn = tile_nodes(ndim, count);
  j.tiles = (int)((count + j.tn - 1) / j.tn);
  return true;
}

hipError_t launch_pack_soa2(const double* aos_a, int ndim_a, long count_a, double* soa_a, long ld_a, const double* aos_b, int ndim_b,
                            long count_b, double* soa_b, long ld_b, hipStream_t st) {
  PackJob a, b;
  if (!pack_job(a, aos_a, ndim_a, count_a, soa_a, ld_a) || !pack_job(b, aos_b, ndim_b, count_b, soa_b, ld_b)) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * (size_t)std::max(a.tn * (a.ndim + 1), b.tn * (b.ndim + 1));
  hipLaunchKernelGGL(k_pack2, dim3((unsigned)(a.tiles + b.tiles)), dim3(THREADS), lds, st, a, b);
  return hipGetLastError();
}

hipError_t launch_unpack_soa2(const double* soa_a, long ld_a, int ndim_a, long count_a, double* aos_a, const double* soa_b, long ld_b,
                              int ndim_b, long count_b, double* aos_b, hipStream_t st) {
  PackJob a, b;
  if (!pack_job(a, aos_a, ndim_a, count_a, const_cast<double*>(soa_a), ld_a) ||
      !pack_job(b, aos_b, ndim_b, count_b, const_cast<double*>(soa_b), ld_b)) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * (size_t)std::max(a.tn * (a.ndim + 1), b.tn * (b.ndim + 1));
  hipLaunchKernelGGL(k_unpack2, dim3((unsigned)(a.tiles + b.tiles)), dim3(THREADS), lds, st, a, b);
  return hipGetLastError();
}

hipError_t launch_defect_norms(const double* defect, long ldd, int ndim, int seg_per_traj, int n_batch, double* sumsq,
                               double* maxabs, hipStream_t st) {
  if (n_batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_defect_norms, dim3(n_batch), dim3(NORM_THREADS), 0, st, defect, ldd, ndim, seg_per_traj, sumsq, maxabs);
  return hipGetLastError();
}

}  // namespace lto
