// lto_api.hip -- the C ABI of include/lto.h: contexts, plans, host-pointer and device-resident sweeps.
//
// No C++ exception crosses the ABI (everything below is noexcept by construction: no STL that
// throws on the hot path, allocation failures are turned into LTO_EHIP).  No signal handlers, no
// global state besides what HIP itself keeps.  A context belongs to one device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>


#include "../../include/lto.h"
#include "kernels.hpp"
#include "hostbuf.hpp"

using namespace lto;

// RK4 STM sweeps with >= 6 steps, 12-dim: microseconds per round of the form whose lane is a whole segment (kernels_indirect_lane.hip;
// a round = 256 segments per CU) at 64 steps on MI355X, for AUTO's comparison with the pipelines' round costs (default of
// lto_ctx::lane_round_us; lto_calibrate_kernels measures it on the context's own device).
static const double kLaneRoundUs = 505.0;      // round 6 (explicit register parking, matrices from the base evaluations' by-products): was 590
// us per round at 64 steps, MI355X: [12-dim | 14-dim][eight-wave (16 x CUs) | 48-segment (48 x CUs) | per-lane with 3 columns (64 x CUs) |
// 44-segment (44 x CUs) | 32-segment (32 x CUs)]; lto_calibrate_kernels replaces them with the context's own device's
static const double kRoundCostDefault[2][5] = {{63.0, 165.0, 246.0, 139.0, 111.0}, {72.0, 191.0, 1e300, 1e300, 128.0}};

struct lto_ctx {
  int device;
  int cu_count;    // compute units of the device: the kernel choice works in rounds of workgroups per CU
  hipStream_t stream;
  bool timing;
  hipEvent_t ev0, ev1;
  bool ev_valid;
  // grow-only device arena for the host-pointer API
  char* arena;
  size_t arena_bytes;
  size_t arena_top;
  // small cache of device blocks for plan-owned buffers: the host-pointer API builds a plan per call, and a
  // hipMalloc/hipFree pair costs more than a 29-segment sweep
  struct { void* ptr; size_t bytes; } pool[8];
  // lane order of the last large adaptive sweep made through the host-pointer API (which builds a plan per call):
  // consecutive Newton iterations sweep the same problem, so the previous call's step counts balance this one.
  // A stale order is still a valid permutation -- it can only cost speed, never correctness.
  // One slot per kind of order (round 6; advisor finding: defect sweeps want the windowed order, STM sweeps / Newton steps the global
  // one, and a loop that alternates defectCalc and jacobianCalc at the same size evicted the other call's order every time).
  int* order_cache[3];   // [kind]: [order_S + workspace] (kernels.hpp order_bytes); kind 1 global, 2 windowed (slot 0 unused)
  long order_S[3];
  int order_ndim[3];
  // plans of the host-pointer API, kept between calls (a Newton iteration calls with the same shapes and parameters
  // every time: no parameter upload, no device allocation per call); owned by the context
  struct HostPlan {
    lto_indirect_plan* plan;
    int ndim, n_nodes, n_batch, n_prm;
    lto_integrator integ;
    lto_params* prm;       // [n_prm] copy of the caller's parameters (the key)
    unsigned long stamp;   // last use
  } host_plans[4];
  unsigned long stamp;
  // lifetime: plans handed to the caller keep the context alive.  lto_destroy with such plans outstanding (a garbage
  // collector runs finalizers in any order) only marks the context; the last lto_*_plan_destroy frees it.
  int live_plans;
  bool closing;
  bool free_claimed = false;   // somebody is freeing this context (ctx_release)
  // page-locked blocks handed out by lto_host_alloc.  The GPU addresses them directly, so the host-pointer API reads and
  // writes a caller's buffer that lies inside one of them in place: the AoS <-> SoA kernels are the transfer, and no
  // copy-engine operation (about 10 us of latency each) is queued.
  // A block keeps its context alive the way a plan does (lto_destroy defers while any is outstanding); `dev` is null for a
  // block the device cannot address directly (still page-locked: the copy engine moves it).  The list has its own lock:
  // a garbage collector may free a block from another thread while a sweep looks one up.
  struct Pinned { char* host; char* dev; size_t bytes; };
  lto::HostList<Pinned> pinned;
  std::mutex pinned_mu;
  double last_call_ms;     // wall time of the last host-pointer call, entry to return (lto_last_call_ms)
  int last_call_order;     // lane order the last host-pointer indirect call swept with: 0 natural, 1 global, 2 windowed (lto_last_call_order)
  // landing block of the Newton loop's per-iteration scalars (lto_indirect_solve_batch): page-locked, mapped, written by
  // k_iter_report; word 0 is the sequence number the host polls, the values follow.  Grow-only; absent = copy + synchronise.
  // AUTO's cost table: microseconds per ROUND of each RK4 STM family at 64 steps, [ndim == 14][family] with family 0 = eight-wave
  // pipeline (rounds of 16 x CUs segments), 1 = 48-segment pipeline (48 x CUs), 2 = per-lane with three columns (64 x CUs; 12-dim
  // only), 3 = 44-segment form of the large-batch pipeline (44 x CUs; 12-dim only), 4 = 32-segment / twelve-wave pipeline (32 x CUs).
  // Defaults: MI355X, profiles/r04z; lto_calibrate_kernels replaces them with this device's own.
  double round_cost[2][5];
  double lane_round_us;    // the whole-segment lanes (kernels_indirect_lane.hip, 12-dim): us per round of 256 x CUs segments at 64 steps
  bool calibrated;
  double* rep_host;
  double* rep_dev;
  size_t rep_doubles;
  long long rep_seq;
  char err[512];
};

struct lto_indirect_plan {
  lto_ctx* ctx;
  int ndim, n_nodes, n_batch, S;
  int pm;           // bit mask of the PMode classes present in the batch
  int n_prm;        // 1 or n_batch
  lto_integrator integ;
  TrajParams* d_tp;
  int* d_nacc;
  int* d_nrej;
  int* d_order;     // [S] lane -> segment map of adaptive sweeps + LTO_ORDER_BINS ints of sort workspace (lazily allocated)
  int use_order;
  int order_kind;      // what d_order holds: 1 = the global order (record staging), 2 = the windowed order (kernels.hpp LTO_ORDER_WINDOW)
  int order_borrowed;  // d_order belongs to the context's cache
  int swept;           // an adaptive sweep has filled the step counters
  int cols_per_lane;
  int kernel;       // LTO_KERNEL_*
  int last_kernel;  // family the last STM sweep ran (AUTO resolved)
  int p48_form;     // large-batch pipeline, 12-dim: 0 = the form with the cheaper rounds, 44 / 48 = that form (calibration)
  double* d_bvp;    // workspace of the device Newton solve (lazily allocated)
  size_t bvp_bytes;
  int bvp_variant;  // -1 none, 0 square system, 1 adjoints-only least squares: what the stored factorisation is
  // warm start of the adaptive controllers (lto_indirect_plan_set_warm_start): first accepted step size of every segment in the
  // last STM sweep / defect-only sweep (they control different error norms, hence two arrays; lazily allocated)
  int warm_start;
  int defect_lanes;         // lanes per segment of the defect-only sweep with the reference's setting: 0 = choose, 1, 2, 4
  double* d_hfirst[2];      // [0] STM sweeps, [1] defect-only sweeps
  int hfirst_valid[2];
  // record staging of rebalanced sweeps (kernels.hpp, IndirectArgs::Xa / Da / Pa): allocated with the lane order
  // trial-step statistics of the last defect-only sweep (k_step_stats): [sum, max, S] in page-locked host memory the kernel writes
  long long* h_stats;       // host view (nullptr: not available)
  long long* h_stats_dev;   // device view of the same block
  unsigned long long* d_stats_acc;   // [3] device scratch
  int stats_age;            // qualifying sweeps so far
  hipEvent_t stats_ev;      // recorded behind every k_step_stats launch
  int stats_pending;        // a k_step_stats launch has not been consumed yet
  int stats_lanes;          // the statistics' verdict, latched when they are consumed: 0 none (size thresholds), 1 or 2 lanes per segment
  double* d_xa;             // [n_nodes n_batch][NODE_REC]
  double* d_da;             // [S][12]
  double* d_pa;             // [S][144]: only for plans that run STM sweeps (stage_alloc's need_phi)
  int stm_swept;            // an STM sweep has run on this plan
  int stage_failed;         // an allocation of record staging failed: the sweeps gather from the caller's arrays (lto_indirect_plan_staging)
  int out_blocks;           // LTO_LAYOUT_BLOCKS: Phi [S][144] and defect [S][12] per-segment blocks instead of struct-of-arrays (lto_indirect_plan_set_output_layout)
};

struct lto_direct_plan {
  lto_ctx* ctx;
  int nstate, n_nodes, n_batch, S, nsteps;
  lto_direct_params prm;
  int kernel;       // LTO_KERNEL_*
};

namespace {

int set_err(lto_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    if (e != hipSuccess) std::snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    else std::snprintf(c->err, sizeof c->err, "%s", what);
  }
  return code;
}

#define LTO_HIP(c, call)                                              \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) return set_err((c), LTO_EHIP, #call, e_);   \
  } while (0)

int bind_device(lto_ctx* c) {
  LTO_HIP(c, hipSetDevice(c->device));
  return LTO_OK;
}

// ---- arena: reset at the start of each host-pointer call, bump-allocated, 256-B aligned
int arena_reserve(lto_ctx* c, size_t bytes) {
  if (bytes <= c->arena_bytes) return LTO_OK;
  if (c->arena) { LTO_HIP(c, hipStreamSynchronize(c->stream)); LTO_HIP(c, hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
  size_t want = bytes + bytes / 4 + (1u << 20);
  LTO_HIP(c, hipMalloc((void**)&c->arena, want));
  c->arena_bytes = want;
  return LTO_OK;
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
template <class T>
T* arena_take(lto_ctx* c, size_t count) {
  T* p = (T*)(c->arena + c->arena_top);
  c->arena_top += al256(count * sizeof(T));
  return p;
}

// ---- device block cache (see lto_ctx::pool)
hipError_t pool_alloc(lto_ctx* c, void** out, size_t bytes) {
  int best = -1;
  for (int i = 0; i < 8; ++i)
    if (c->pool[i].ptr && c->pool[i].bytes >= bytes && (best < 0 || c->pool[i].bytes < c->pool[best].bytes)) best = i;
  if (best >= 0 && c->pool[best].bytes <= 4 * bytes + 4096) {
    *out = c->pool[best].ptr;
    c->pool[best].ptr = nullptr;
    return hipSuccess;
  }
  return hipMalloc(out, bytes < 256 ? 256 : bytes);
}
void pool_free(lto_ctx* c, void* ptr, size_t bytes) {
  if (!ptr) return;
  if (bytes < 256) bytes = 256;
  int slot = -1;
  for (int i = 0; i < 8; ++i) if (!c->pool[i].ptr) { slot = i; break; }
  if (slot < 0) {  // evict the smallest cached block
    slot = 0;
    for (int i = 1; i < 8; ++i) if (c->pool[i].bytes < c->pool[slot].bytes) slot = i;
    (void)hipFree(c->pool[slot].ptr);
  }
  c->pool[slot].ptr = ptr;
  c->pool[slot].bytes = bytes;
}

// Reference validity rule for p (stateCostate_deriv.jl:36-53): p == 0, p == 1 or p > 1.
bool p_valid(double p) { return p == 0.0 || p == 1.0 || p > 1.0; }

int make_traj_params(lto_ctx* c, int ndim, const lto_params* prm, int n, TrajParams* out, int* pm_out) {
  int pm = 0;
  for (int i = 0; i < n; ++i) {
    const lto_params& q = prm[i];
    if (!p_valid(q.p)) return set_err(c, LTO_EBADP, "Invalid value of p!");
    TrajParams t;
    // ndim = 12: `mass` is the constant spacecraft mass.  ndim = 14: mass is state[7] and the slot carries Isp.
    t.accel_limit = (ndim == 12) ? q.thrustLimit / q.mass / 1e3 * (q.TU * q.TU) / q.DU : 0.0;  // stateCostate_deriv.jl:33
    t.cT = q.thrustLimit / 1e3 * (q.TU * q.TU) / q.DU;
    t.kappa_td = (ndim == 14) ? q.time_direction * 1e3 * q.DU / (q.TU * q.mass * 9.81) : 0.0;
    t.inv_2rho = 1.0 / (2.0 * q.rho);
    t.inv_rho = 1.0 / q.rho;
    t.p = q.p;
    t.inv_p = (q.p != 0.0) ? 1.0 / q.p : 0.0;
    t.inv_pm1 = (q.p > 1.0) ? 1.0 / (q.p - 1.0) : 0.0;
    t.omega = q.time_direction;
    t.MU = q.MU;
    out[i] = t;
    pm |= 1 << p_class(q.p);
  }
  *pm_out = pm;                                  // bit mask of the control-law classes present
  return LTO_OK;
}

int check_integ(lto_ctx* c, const lto_integrator* ig) {
  if (!ig) return set_err(c, LTO_ENULL, "integrator is NULL");
  switch (ig->method) {
    case LTO_RK4:
    case LTO_RKF78_FIXED:
      if (ig->steps < 1) return set_err(c, LTO_EINVAL, "fixed-step integrator needs steps >= 1");
      return LTO_OK;
    case LTO_RKF78_ADAPTIVE:
      if (!(ig->rtol > 0.0)) return set_err(c, LTO_EINVAL, "adaptive integrator needs rtol > 0");
      return LTO_OK;
    case LTO_DOP853_ADAPTIVE:
      if (!(ig->rtol > 0.0) || !(ig->atol >= 0.0)) return set_err(c, LTO_EINVAL, "adaptive integrator needs rtol > 0, atol >= 0");
      return LTO_OK;
  }
  return set_err(c, LTO_EINVAL, "unknown integrator method");
}

void timing_begin(lto_ctx* c, hipStream_t st) {
  if (c->timing) { (void)hipEventRecord(c->ev0, st); }
}
void timing_end(lto_ctx* c, hipStream_t st) {
  if (c->timing) { (void)hipEventRecord(c->ev1, st); c->ev_valid = true; }
}

}  // namespace

extern "C" {

static void plan_free(lto_indirect_plan* p);
static void ctx_free(lto_ctx* c);
static int host_plan_acquire(lto_ctx* c, int ndim, int n_nodes, int n_batch, const lto_params* prm, int n_prm,
                             const lto_integrator* integ, lto_indirect_plan** out);

int lto_version(void) { return LTO_VERSION; }

int lto_create(lto_ctx** out, int device_id) {
  if (!out) return LTO_ENULL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LTO_ENODEVICE;
  if (device_id < 0 || device_id >= n) return LTO_EINVAL;
  lto_ctx* c = new (std::nothrow) lto_ctx();   // value-initialised: every scalar member zero, the vector empty
  if (!c) return LTO_EHIP;
  c->device = device_id;
  c->cu_count = 0;
  std::memcpy(c->round_cost, kRoundCostDefault, sizeof kRoundCostDefault);
  c->lane_round_us = kLaneRoundUs;
  if (hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess) { c->cu_count = 0; (void)hipGetLastError(); }
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return LTO_EHIP;
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return LTO_EHIP;
  }
  *out = c;
  return LTO_OK;
}

// page-locked blocks -> owning context (lto_host_free may come without the handle, from any thread)
static std::mutex g_blocks_mu;
struct HostBlock { void* ptr; lto_ctx* owner; };
static lto::HostList<HostBlock> g_blocks;          // a handful of entries: linear search
static lto_ctx* host_block_take(void* ptr) {       // under g_blocks_mu: the owner of `ptr`, the entry removed; nullptr if unknown
  for (size_t k = 0; k < g_blocks.size(); ++k)
    if (g_blocks[k].ptr == ptr) { lto_ctx* o = g_blocks[k].owner; g_blocks.erase_at(k); return o; }
  return nullptr;
}
static void host_block_forget(void* ptr) { std::lock_guard<std::mutex> lk(g_blocks_mu); (void)host_block_take(ptr); }
static bool ctx_has_blocks(lto_ctx* c) { std::lock_guard<std::mutex> lk(c->pinned_mu); return !c->pinned.empty(); }

// A context has three kinds of owners: its handle (until lto_destroy), its plans, its page-locked blocks; garbage collectors
// release them in any order and from any thread (lto_host_free takes no handle).  Who frees the context is decided under ONE
// lock, and exactly once (advisor finding, round 3: two threads could both see "last owner" and free it twice).
static std::mutex g_life_mu;
enum CtxOwner { OWNER_HANDLE, OWNER_PLAN, OWNER_BLOCK };
static void ctx_plan_added(lto_ctx* c) { std::lock_guard<std::mutex> lk(g_life_mu); ++c->live_plans; }
static bool ctx_is_closing(lto_ctx* c) { std::lock_guard<std::mutex> lk(g_life_mu); return c->closing; }
// the caller has given up an owner of kind `what` (a block: already removed from c->pinned); true = the caller frees the context
static bool ctx_release(lto_ctx* c, CtxOwner what) {
  std::lock_guard<std::mutex> lk(g_life_mu);
  if (what == OWNER_HANDLE) c->closing = true;
  if (what == OWNER_PLAN) --c->live_plans;
  if (!c->closing || c->live_plans > 0 || c->free_claimed || ctx_has_blocks(c)) return false;
  c->free_claimed = true;
  return true;
}

void lto_destroy(lto_ctx* c) {
  if (!c) return;
  if (ctx_release(c, OWNER_HANDLE)) ctx_free(c);      // otherwise: freed by the last lto_*_plan_destroy / lto_host_free
}

static void ctx_free(lto_ctx* c) {
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& h : c->host_plans) { if (h.plan) plan_free(h.plan); std::free(h.prm); h.plan = nullptr; h.prm = nullptr; }
  for (const lto_ctx::Pinned& b : c->pinned) { host_block_forget(b.host); (void)hipHostFree(b.host); }   // none left on the deferred path
  c->pinned.clear();
  if (c->arena) (void)hipFree(c->arena);
  for (int k = 0; k < 3; ++k) if (c->order_cache[k]) (void)hipFree(c->order_cache[k]);
  if (c->rep_host) (void)hipHostFree(c->rep_host);
  for (int i = 0; i < 8; ++i) if (c->pool[i].ptr) (void)hipFree(c->pool[i].ptr);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* lto_last_error(const lto_ctx* c) { return c ? c->err : "null context"; }

void* lto_ctx_stream(lto_ctx* c) { return c ? (void*)c->stream : nullptr; }
int lto_ctx_device(const lto_ctx* c) { return c ? c->device : -1; }

int lto_set_timing(lto_ctx* c, int enabled) {
  if (!c) return LTO_ENULL;
  c->timing = enabled != 0;
  c->ev_valid = false;
  return LTO_OK;
}

double lto_last_kernel_ms(lto_ctx* c) {
  if (!c || !c->ev_valid) return -1.0;
  float ms = -1.0f;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

double lto_last_call_ms(const lto_ctx* c) { return c ? c->last_call_ms : -1.0; }
int lto_last_call_order(const lto_ctx* c) { return c ? c->last_call_order : LTO_ENULL; }

// entry-to-return wall time of a host-pointer call, kept in the context
struct CallTimer {
  lto_ctx* c;
  std::chrono::steady_clock::time_point t0;
  explicit CallTimer(lto_ctx* ctx) : c(ctx), t0(std::chrono::steady_clock::now()) {}
  ~CallTimer() { if (c) c->last_call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

/* ------------------------------------------------------------------------------ indirect plans */

// plan construction without lifetime bookkeeping (the library's own short-lived and cached plans)
static int plan_build(lto_ctx* c, int ndim, int n_nodes, int n_batch, const lto_params* prm, int n_prm,
                      const lto_integrator* integ, lto_indirect_plan** out) {
  if (!c || !out) return LTO_ENULL;
  *out = nullptr;
  if (!prm) return set_err(c, LTO_ENULL, "prm is NULL");
  if (ndim != 12 && ndim != 14) return set_err(c, LTO_EINVAL, "ndim must be 12 (or 14: mass + mass costate extension)");
  if (n_nodes < 2 || n_batch < 1) return set_err(c, LTO_EINVAL, "need n_nodes >= 2 and n_batch >= 1");
  if (n_prm != 1 && n_prm != n_batch) return set_err(c, LTO_EINVAL, "n_prm must be 1 or n_batch");
  if ((long)(n_nodes - 1) * n_batch > 0x7fffffffL) return set_err(c, LTO_EINVAL, "too many segments");
  int rc = check_integ(c, integ);
  if (rc) return rc;
  rc = bind_device(c);
  if (rc) return rc;
  TrajParams* h = (TrajParams*)std::malloc(sizeof(TrajParams) * (size_t)n_prm);
  if (!h) return set_err(c, LTO_EHIP, "host allocation failed");
  int pm = 0;
  rc = make_traj_params(c, ndim, prm, n_prm, h, &pm);
  if (rc) { std::free(h); return rc; }
  lto_indirect_plan* p = new (std::nothrow) lto_indirect_plan();
  if (!p) { std::free(h); return set_err(c, LTO_EHIP, "host allocation failed"); }
  std::memset(p, 0, sizeof *p);
  p->ctx = c; p->ndim = ndim; p->n_nodes = n_nodes; p->n_batch = n_batch; p->S = (n_nodes - 1) * n_batch;
  p->pm = pm; p->n_prm = n_prm; p->integ = *integ; p->bvp_variant = -1;
  if (p->integ.max_steps <= 0) p->integ.max_steps = 100000;
  hipError_t e = pool_alloc(c, (void**)&p->d_tp, sizeof(TrajParams) * (size_t)n_prm);
  if (e == hipSuccess) e = hipMemcpy(p->d_tp, h, sizeof(TrajParams) * (size_t)n_prm, hipMemcpyHostToDevice);
  std::free(h);
  const bool adaptive = integ->method == LTO_RKF78_ADAPTIVE || integ->method == LTO_DOP853_ADAPTIVE;
  if (e == hipSuccess && adaptive) {
    e = pool_alloc(c, (void**)&p->d_nacc, sizeof(int) * (size_t)p->S);
    if (e == hipSuccess) e = pool_alloc(c, (void**)&p->d_nrej, sizeof(int) * (size_t)p->S);
  }
  if (e != hipSuccess) {
    plan_free(p);
    return set_err(c, LTO_EHIP, "plan allocation", e);
  }
  // Page-locked landing place of the trial-step statistics that steer AUTO's lanes per segment (lto_indirect_defect_dev): here, not
  // in the first sweep that wants it -- a sweep may be inside a caller's graph capture, where nothing may be allocated.
  if (ndim == 12 && integ->method == LTO_DOP853_ADAPTIVE && p->S >= 64L * c->cu_count) {
    void* hp = nullptr; void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess &&
        pool_alloc(c, (void**)&p->d_stats_acc, sizeof(unsigned long long) * 4) == hipSuccess &&
        hipMemset(p->d_stats_acc, 0, sizeof(unsigned long long) * 4) == hipSuccess) {
      std::memset(hp, 0, 64);
      p->h_stats = (long long*)hp; p->h_stats_dev = (long long*)dp;
      if (hipEventCreateWithFlags(&p->stats_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); (void)hipHostFree(hp);
        p->h_stats = nullptr; p->stats_ev = nullptr;
      }
    } else {                       // no statistics: AUTO keeps its size thresholds
      (void)hipGetLastError();
      if (hp) (void)hipHostFree(hp);
      p->h_stats = nullptr;
    }
  }
  *out = p;
  return LTO_OK;
}

int lto_indirect_plan_create(lto_ctx* c, int ndim, int n_nodes, int n_batch, const lto_params* prm, int n_prm,
                             const lto_integrator* integ, lto_indirect_plan** out) {
  const int rc = plan_build(c, ndim, n_nodes, n_batch, prm, n_prm, integ, out);
  if (rc == LTO_OK) ctx_plan_added(c);
  return rc;
}

// The caller may have launched sweeps of this plan on its own streams: the plan's device blocks go back to the
// context's block cache (pool_free) and may be handed to the next plan at once, so everything in flight on the device
// has to finish first.  Destroying a plan is rare; the library's own short-lived plans use plan_free after
// synchronising the one stream they used.
void lto_indirect_plan_destroy(lto_indirect_plan* p) {
  if (!p) return;
  lto_ctx* c = p->ctx;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  plan_free(p);
  if (ctx_release(c, OWNER_PLAN)) ctx_free(c);
}

static void plan_free(lto_indirect_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  pool_free(p->ctx, p->d_tp, sizeof(TrajParams) * (size_t)p->n_prm);
  pool_free(p->ctx, p->d_nacc, sizeof(int) * (size_t)p->S);
  pool_free(p->ctx, p->d_nrej, sizeof(int) * (size_t)p->S);
  if (!p->order_borrowed) pool_free(p->ctx, p->d_order, order_bytes(p->S));
  pool_free(p->ctx, p->d_bvp, p->bvp_bytes);
  for (int k = 0; k < 2; ++k) pool_free(p->ctx, p->d_hfirst[k], sizeof(double) * (size_t)p->S);
  if (p->h_stats) (void)hipHostFree(p->h_stats);
  if (p->stats_ev) (void)hipEventDestroy(p->stats_ev);
  pool_free(p->ctx, p->d_stats_acc, sizeof(unsigned long long) * 4);
  pool_free(p->ctx, p->d_xa, sizeof(double) * NODE_REC * (size_t)p->n_nodes * p->n_batch);
  pool_free(p->ctx, p->d_da, sizeof(double) * 12 * (size_t)p->S);
  pool_free(p->ctx, p->d_pa, sizeof(double) * 144 * (size_t)p->S);
  delete p;
}

const int* lto_indirect_plan_steps_accepted(const lto_indirect_plan* p) { return p ? p->d_nacc : nullptr; }
const int* lto_indirect_plan_steps_rejected(const lto_indirect_plan* p) { return p ? p->d_nrej : nullptr; }

int lto_indirect_plan_copy_steps(lto_indirect_plan* p, void* stream, int* accepted, int* rejected) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  if (!p->d_nacc || !p->d_nrej) return set_err(c, LTO_EINVAL, "fixed-step plan has no step counters");
  int rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (accepted) e = hipMemcpyAsync(accepted, p->d_nacc, sizeof(int) * (size_t)p->S, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && rejected) e = hipMemcpyAsync(rejected, p->d_nrej, sizeof(int) * (size_t)p->S, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "copy step counters", e);
  return LTO_OK;
}

// Lane order of the adaptive sweeps.  Round 5: ordered inside windows of consecutive segments, the windows dealt to the XCDs
// (kernels.hpp LTO_ORDER_WINDOW) -- the sweeps then gather from and scatter to the caller's arrays inside one L2 and need no record
// staging.  LTO_ORDER_MODE=global in the environment (development switch, read once per process): the global order of round 4 with
// its record staging.
// A plan that runs STM sweeps keeps the global order and its records: with sixteen workgroup-rounds per CU the sweep's time is the
// sum of its rounds, longest-processing-time-first over ALL workgroups is what keeps that sum short, and the windowed order costs
// 17 % of time there (1.75 against 1.50 ms at C5 + STM) for its 2.6 x less traffic.  A defect-only plan (C5 itself, the line
// search's trial plan) takes the windowed order: same time, a third of the traffic, no record passes.
static int order_kind_for(bool stm) {
  static const int forced = [] {
    const char* e = std::getenv("LTO_ORDER_MODE");
    return !e ? 0 : std::strcmp(e, "global") == 0 ? 1 : std::strcmp(e, "windowed") == 0 ? 2 : 0;
  }();
  return forced ? forced : (stm ? 1 : 2);
}
static int order_weave() {
  static const int w = [] { const char* e = std::getenv("LTO_ORDER_WEAVE"); const int v = e ? std::atoi(e) : 0; return (v >= 1 && v <= 255) ? v : 0; }();      // 0 = the kernel's choice
  return w;
}
static hipError_t segment_order(int kind, const int* nacc, const int* nrej, long S, int* work, int* order, hipStream_t st) {
  return kind == 2 ? launch_segment_order_windowed(nacc, nrej, (int)S, order_weave(), work, order, st)
                   : launch_segment_order(nacc, nrej, (int)S, work, order, st);
}

// Record staging (12-dim plans with the reference's integrator setting): the buffers come with the lane order, outside any sweep.
static bool stage_capable(const lto_indirect_plan* p) { return p->ndim == 12 && p->integ.method == LTO_DOP853_ADAPTIVE; }
// need_phi: the plan runs STM sweeps, so the [S][144] Phi records are wanted too.  A defect-only plan (the line search's S x 20
// trial plan) never gets them: at 256 x 1 024 x 20 segments they would pin 6 GB nothing reads (advisor finding, round 4).  An
// allocation that fails switches staging off for what it was for -- the sweeps then gather from the caller's arrays as before --
// and is reported: lto_indirect_plan_staging() carries the bit, lto_last_error() the text (the call still returns LTO_OK).
static int stage_alloc(lto_indirect_plan* p, bool need_phi) {
  if (!stage_capable(p)) return LTO_OK;
  lto_ctx* c = p->ctx;
  const bool own = !p->out_blocks;       // LTO_LAYOUT_BLOCKS: the caller's Phi / defect arrays ARE the records
  struct { double** ptr; size_t n; bool want; } want[3] = {{&p->d_xa, (size_t)NODE_REC * p->n_nodes * p->n_batch, true},
                                                           {&p->d_da, (size_t)12 * p->S, own}, {&p->d_pa, (size_t)144 * p->S, need_phi && own}};
  for (auto& w : want) {
    if (*w.ptr || !w.want) continue;
    hipError_t e = pool_alloc(c, (void**)w.ptr, sizeof(double) * w.n);
    if (e != hipSuccess) {
      *w.ptr = nullptr; (void)hipGetLastError();
      p->stage_failed = 1;
      std::snprintf(c->err, sizeof c->err, "note: record staging of ordered sweeps is off for this plan (%zu bytes: %s); results are unaffected",
                    sizeof(double) * w.n, hipGetErrorString(e));
      return LTO_OK;
    }
  }
  return LTO_OK;
}

int lto_indirect_plan_staging(const lto_indirect_plan* p) {
  if (!p) return 0;
  if (p->out_blocks) return (p->d_xa ? 3 : 0) | (p->stage_failed ? 4 : 0);      // node records only: results go straight to the caller's blocks
  return ((p->d_xa && p->d_da) ? 1 : 0) | (p->d_pa ? 2 : 0) | (p->stage_failed ? 4 : 0);
}

int lto_indirect_plan_set_output_layout(lto_indirect_plan* p, int layout) {
  if (!p) return LTO_ENULL;
  if (layout != LTO_LAYOUT_SOA && layout != LTO_LAYOUT_BLOCKS) return set_err(p->ctx, LTO_EINVAL, "layout must be LTO_LAYOUT_SOA or LTO_LAYOUT_BLOCKS");
  if (layout == LTO_LAYOUT_BLOCKS && !stage_capable(p))
    return set_err(p->ctx, LTO_EUNSUPPORTED, "LTO_LAYOUT_BLOCKS is built for 12-dim DOP853_ADAPTIVE plans (the kernels that write per-segment records)");
  p->out_blocks = layout == LTO_LAYOUT_BLOCKS;
  return LTO_OK;
}

int lto_indirect_plan_rebalance(lto_indirect_plan* p, void* stream) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  if (!p->d_nacc || !p->d_nrej) return set_err(c, LTO_EINVAL, "fixed-step plan: every segment takes the same number of steps");
  if (!p->swept) return set_err(c, LTO_EINVAL, "no sweep has run on this plan yet: there are no step counts to balance by");
  int rc = bind_device(c);
  if (rc) return rc;
  if (!p->d_order) {
    hipError_t e = pool_alloc(c, (void**)&p->d_order, order_bytes(p->S));
    if (e != hipSuccess) return set_err(c, LTO_EHIP, "order allocation", e);
  }
  const int kind = order_kind_for(p->stm_swept != 0);
  hipError_t e = segment_order(kind, p->d_nacc, p->d_nrej, p->S, p->d_order + p->S, p->d_order, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_segment_order", e);
  p->use_order = 1;
  p->order_kind = kind;
  return kind == 2 ? LTO_OK : stage_alloc(p, p->stm_swept != 0);
}

int lto_indirect_plan_reset_order(lto_indirect_plan* p) {
  if (!p) return LTO_ENULL;
  p->use_order = 0;
  return LTO_OK;
}

int lto_indirect_plan_set_kernel(lto_indirect_plan* p, int kernel) {
  if (!p) return LTO_ENULL;
  // (selectors 3 and 4 are not indirect families: 3 = LTO_KERNEL_DIRECT_PIPE, the direct plans' pipelined Jacobian kernel; 4 is unassigned)
  if (kernel != LTO_KERNEL_AUTO && kernel != LTO_KERNEL_PER_LANE && kernel != LTO_KERNEL_COOP && kernel != LTO_KERNEL_LANE &&
      kernel != LTO_KERNEL_PIPE8 && kernel != LTO_KERNEL_COOP2 && kernel != LTO_KERNEL_PIPE48 && kernel != LTO_KERNEL_PIPE32)
    return set_err(p->ctx, LTO_EINVAL, "kernel must be LTO_KERNEL_AUTO, _PER_LANE, _COOP, _PIPE8, _COOP2, _PIPE48, _PIPE32 or _LANE");
  if (kernel == LTO_KERNEL_LANE && !indirect_stm_lane_available(p->ndim, p->integ.method, p->S))
    return set_err(p->ctx, LTO_EINVAL, "LTO_KERNEL_LANE is built for 12-dim RK4 plans");
  if (kernel == LTO_KERNEL_COOP2 && (p->integ.method != LTO_DOP853_ADAPTIVE || (p->ndim != 12 && !indirect_stm_coop2_14_available(p->pm))))
    return set_err(p->ctx, LTO_EINVAL, "LTO_KERNEL_COOP2 is built for DOP853_ADAPTIVE plans: 12-dim, and 14-dim with p = 0 or p = 1");
  if ((kernel == LTO_KERNEL_PIPE8 || kernel == LTO_KERNEL_PIPE48 || kernel == LTO_KERNEL_PIPE32) && p->integ.method != LTO_RK4)
    return set_err(p->ctx, LTO_EINVAL, "the pipeline kernels are built for fixed-step RK4 plans");
  if (kernel == LTO_KERNEL_PIPE32 && !indirect_stm_pipe32_available(p->ndim, p->pm))
    return set_err(p->ctx, LTO_EINVAL, "LTO_KERNEL_PIPE32 is built for 12-dim plans and for 14-dim plans with p = 0 or p = 1");
  p->kernel = kernel;
  return LTO_OK;
}

int lto_indirect_plan_last_kernel(const lto_indirect_plan* p) { return p ? p->last_kernel : LTO_KERNEL_AUTO; }

int lto_indirect_plan_set_defect_lanes(lto_indirect_plan* p, int lanes) {
  if (!p) return LTO_ENULL;
  if (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 4) return set_err(p->ctx, LTO_EINVAL, "defect lanes must be 0 (choose), 1, 2 or 4");
  const bool quad14 = p->ndim == 14 && p->integ.method == LTO_DOP853_ADAPTIVE && indirect_stm_coop2_14_available(p->pm);
  if (lanes > 1 && !(p->ndim == 12 && p->integ.method == LTO_DOP853_ADAPTIVE) && !(lanes == 4 && quad14))
    return set_err(p->ctx, LTO_EINVAL, "two and four lanes per segment are built for 12-dim DOP853_ADAPTIVE plans (the reference's integrator setting); "
                                       "four also for 14-dim DOP853_ADAPTIVE plans with p = 0 or p = 1");
  p->defect_lanes = lanes;
  return LTO_OK;
}

int lto_indirect_plan_set_warm_start(lto_indirect_plan* p, int on) {
  if (!p) return LTO_ENULL;
  if (on && !(p->ndim == 12 && p->integ.method == LTO_DOP853_ADAPTIVE))
    return set_err(p->ctx, LTO_EINVAL, "warm start is built for 12-dim DOP853_ADAPTIVE plans (the reference's integrator setting)");
  if (on) {
    // Both arrays are allocated and zeroed HERE, not inside the first warm sweep (advisor finding, round 3): a sweep may be part
    // of a caller's graph capture, where nothing may be allocated, and a segment a sweep skips (zero span, another launch's
    // control-law class) must leave a value the next sweep recognises as "none" -- 0 -- not whatever the pool handed out.
    int rc = bind_device(p->ctx);
    if (rc) return rc;
    for (int k = 0; k < 2; ++k) {
      if (p->d_hfirst[k]) continue;
      hipError_t e = pool_alloc(p->ctx, (void**)&p->d_hfirst[k], sizeof(double) * (size_t)p->S);
      if (e == hipSuccess) e = hipMemset(p->d_hfirst[k], 0, sizeof(double) * (size_t)p->S);
      if (e != hipSuccess) { p->d_hfirst[k] = nullptr; return set_err(p->ctx, LTO_EHIP, "warm-start array", e); }
      p->hfirst_valid[k] = 0;
    }
  }
  p->warm_start = on ? 1 : 0;
  if (!on) { p->hfirst_valid[0] = 0; p->hfirst_valid[1] = 0; }
  return LTO_OK;
}

// The h_first array of sweep kind `which` (0 STM, 1 defect-only); args get it with the warm flag.  The caller marks the array valid
// (warm_filled) only once the sweep that fills it has been launched successfully.
static int warm_args(lto_indirect_plan* p, int which, bool kernel_records, IndirectArgs* a) {
  a->h_first = nullptr; a->warm = 0;
  if (!p->warm_start || !kernel_records || !p->d_hfirst[which]) return LTO_OK;
  a->h_first = p->d_hfirst[which];
  a->warm = p->hfirst_valid[which];
  return LTO_OK;
}
static void warm_filled(lto_indirect_plan* p, int which, const IndirectArgs& a) {
  if (a.h_first) p->hfirst_valid[which] = 1;          // stream order: the next sweep of this kind reads what this one wrote
}

int lto_indirect_plan_set_cols_per_lane(lto_indirect_plan* p, int cols) {
  if (!p) return LTO_ENULL;
  if (cols == 12 || cols == 14) {
    if (cols != p->ndim || !indirect_stm_stream_available(p->ndim, p->integ.method, p->integ.steps, p->S))
      return set_err(p->ctx, LTO_EINVAL, "cols_per_lane = ndim (12 or 14: the whole STM in the segment's lane) is built for RK4 plans with ONE step per segment");
  } else if (cols != 0 && cols != 1 && cols != 2 && cols != 3) return set_err(p->ctx, LTO_EINVAL, "cols_per_lane must be 0, 1, 2, 3 or the plan's dimension");
  if (p->ndim == 14 && cols == 3) return set_err(p->ctx, LTO_EUNSUPPORTED, "14 STM columns do not split into groups of 3: use 0 (auto), 1 or 2");
  if (p->ndim == 12 && cols == 2) return set_err(p->ctx, LTO_EUNSUPPORTED, "two columns per lane are not built for 12-dim plans (removed in round 6: one column wins up to 8 192 segments, three above): use 0 (auto), 1 or 3");
  p->cols_per_lane = cols;
  return LTO_OK;
}

static int fill_indirect_args(lto_indirect_plan* p, const double* X, long ldx, const double* t, int n_tgrids,
                              IndirectArgs* a) {
  lto_ctx* c = p->ctx;
  if (!X || !t) return set_err(c, LTO_ENULL, "X or t is NULL");
  const long J = (long)p->n_nodes * p->n_batch;
  if (ldx < J) return set_err(c, LTO_EINVAL, "ldx smaller than n_nodes*n_batch");
  if (n_tgrids != 1 && n_tgrids != p->n_batch) return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  std::memset(a, 0, sizeof *a);
  a->X = X; a->ldx = ldx; a->t = t; a->t_stride = (n_tgrids == 1) ? 0 : p->n_nodes;
  a->tp = p->d_tp; a->tp_stride = (p->n_prm == 1) ? 0 : 1;
  a->n_nodes = p->n_nodes; a->seg_per_traj = p->n_nodes - 1; a->S = p->S;
  a->steps = p->integ.steps; a->rtol = p->integ.rtol; a->atol = p->integ.atol; a->max_steps = p->integ.max_steps;
  a->nacc = p->d_nacc; a->nrej = p->d_nrej;
  a->order = p->use_order ? p->d_order : nullptr;
  a->xcd_ranges = (p->use_order && p->order_kind == 2) ? 1 : 0;
  a->stm_scale = std::pow(3.0, -(double)(p->integ.steps > 0 ? p->integ.steps % 256 : 0));   // pipe_common.hpp COL_RESCALE_EVERY
  p->swept = 1;                                    // every caller launches a sweep right after a successful fill
  return LTO_OK;
}

int lto_indirect_defect_dev(lto_indirect_plan* p, void* stream, const double* X, long ldx, const double* t,
                            int n_tgrids, double* defect, long ldd, double* errors) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  IndirectArgs a;
  int rc = fill_indirect_args(p, X, ldx, t, n_tgrids, &a);
  if (rc) return rc;
  if (!defect) return set_err(c, LTO_ENULL, "defect is NULL");
  if (ldd < p->S && !p->out_blocks) return set_err(c, LTO_EINVAL, "ldd smaller than the segment count");
  a.defect = defect; a.ldd = ldd; a.errors = errors;
  rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  timing_begin(c, st);
  // The reference's setting (12-dim, DOP853).  Two lanes per segment (tools/probe_defect2.py: 29 segments 99 -> 73 us, 4 096:
  // 119 -> 88 us, 65 536 ordered: 0.43 -> 0.32 ms, 262 144: 0.44 -> 0.38 ms; 524 288: 0.60 -> 0.73 ms, so one lane beyond);
  // round 3: four lanes per segment (a DPP quad, 16 segments per wavefront) up to eight wavefronts per SIMD -- 4 096 segments:
  // 90 -> 77 us, 65 536 ordered: 0.31 -> 0.27 ms, 131 072: 0.32 -> 0.30 ms; 262 144: 0.39 -> 0.52 ms, so two lanes there.
  // LTO_KERNEL_PER_LANE / LTO_KERNEL_COOP2 on the plan, or lto_indirect_plan_set_defect_lanes, force one form.
  const bool ref_setting = p->ndim == 12 && p->integ.method == LTO_DOP853_ADAPTIVE;
  // the same integrator setting on the 14-dim system (always-thrust-limited laws): the quad form while the chip has a SIMD per 16
  // segments to spare, as for 12-dim (round 6)
  const bool quad14 = p->ndim == 14 && p->integ.method == LTO_DOP853_ADAPTIVE && indirect_stm_coop2_14_available(p->pm);
  int lanes = 1;
  if (quad14) lanes = p->defect_lanes ? p->defect_lanes : ((p->kernel == LTO_KERNEL_AUTO || p->kernel == LTO_KERNEL_COOP2) && (long)(p->S + 15) / 16 <= 32L * c->cu_count) ? 4 : 1;
  if (ref_setting) {
    if (p->defect_lanes) lanes = p->defect_lanes;
    else if (p->kernel == LTO_KERNEL_COOP2) lanes = 2;
    else if (p->kernel == LTO_KERNEL_AUTO) {
      lanes = ((long)(p->S + 15) / 16 <= 32L * c->cu_count) ? 4 : (p->S <= 262144 ? 2 : 1);
      // Those thresholds come from the C5 study, where the slowest segment takes 8 x the mean number of trial steps and sets the
      // sweep's time: more lanes per segment = a shorter stream for it.  A sweep whose segments all take about the same number
      // of steps (the 20 trial trajectories of a line search) is throughput-bound once the chip is full, and fewer lanes per
      // segment issue fewer instructions per segment (tools/probe_linesearch_lanes.py, 20 x 4 096 segments: 166 / 147 / 124 us
      // with 4 / 2 / 1 lanes; 20 x 1 024: 73 / 61 / 94).  The previous sweep's statistics say which case this is.
      // The verdict is the same in every run of the same call sequence (advisor finding, round 4: it used to be "whatever has
      // arrived by then", read while the kernel might still be writing): statistics are consumed only behind the event recorded
      // after k_step_stats -- the host waits for it here, i.e. for the EARLIER sweep that launched it, which a Newton loop has
      // long read back -- then latched in the plan until the next statistics launch is consumed.  Inside a graph capture nothing
      // may be waited for: the latched verdict stands.
      if (p->h_stats && p->S >= 64L * c->cu_count) {
        if (p->stats_pending) {
          hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
          if (cap == hipStreamCaptureStatusNone && hipEventSynchronize(p->stats_ev) == hipSuccess) {
            p->stats_pending = 0;
            const long long sum = p->h_stats[0], mx = p->h_stats[1], cnt = p->h_stats[2];
            p->stats_lanes = 0;
            if (cnt == p->S && sum > 0 && mx * (long long)p->S <= 3 * sum)      // max <= 3 x mean: no tail worth shortening
              p->stats_lanes = (p->S <= 160L * c->cu_count) ? 2 : 1;
          }
        }
        if (p->stats_lanes) lanes = p->stats_lanes;
      }
    }
  }
  if (p->out_blocks && lanes == 1) lanes = 2;          // the one-lane kernel writes struct-of-arrays only
  rc = warm_args(p, 1, lanes > 1, &a);
  if (rc) return rc;
  // balanced lane order: nodes in, defects out as records (IndirectArgs::Xa / Da), coalesced transposes either side of the sweep
  // (LTO_LAYOUT_BLOCKS: the caller's defect array is the record array -- no transpose behind the sweep, whatever the order)
  const bool blocks = p->out_blocks != 0;
  const bool staged = a.order && p->order_kind == 1 && lanes > 1 && p->d_xa && (blocks || p->d_da);
  if (staged) {
    hipError_t q = launch_node_records(X, ldx, t, a.t_stride, p->n_nodes, (long)p->n_nodes * p->n_batch, p->d_xa, st);
    if (q != hipSuccess) return set_err(c, LTO_EHIP, "launch_node_records", q);
    a.Xa = p->d_xa; a.Da = p->d_da;
  }
  if (blocks) a.Da = defect;
  hipError_t e = lanes == 4        ? (p->ndim == 12 ? launch_indirect_defect4(p->pm, a, st) : launch_indirect14_defect4(p->pm, a, st))
                 : lanes == 2      ? launch_indirect_defect2(p->pm, a, st)
                 : (p->ndim == 12) ? launch_indirect_defect(p->pm, p->integ.method, a, st)
                                   : launch_indirect14_defect(p->pm, p->integ.method, a, st);
  if (e == hipSuccess && staged && !blocks) e = launch_pack_soa(p->d_da, 12, p->S, defect, ldd, st);
  timing_end(c, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_indirect_defect", e);
  warm_filled(p, 1, a);
  if (ref_setting && p->kernel == LTO_KERNEL_AUTO && !p->defect_lanes && p->S >= 64L * c->cu_count) {
    // statistics for the next sweep's choice (a few us, stream-ordered, written by the kernel itself into page-locked memory)
    // not after every sweep (the extra launch and its host write cost ~10 us): after the first two, then every sixteenth
    const int age = p->stats_age++;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;        // (an event recorded inside a capture cannot be waited for later)
    if (p->h_stats && (age < 2 || (age & 15) == 0) && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone &&
        launch_step_stats(p->d_nacc, p->d_nrej, p->S, p->d_stats_acc, p->h_stats_dev, st) == hipSuccess &&
        hipEventRecord(p->stats_ev, st) == hipSuccess)
      p->stats_pending = 1;
  }
  p->swept = 1;
  return LTO_OK;
}

/* What LTO_KERNEL_AUTO resolves to for an STM sweep: a pure function of the plan's shape and the cost table (no device, no context
 * state), so that the choice at sizes this build never ran on -- the per-rank batches of an 8-GPU run -- can be pinned by a CPU test
 * (tests/test_auto_kernel.py).  cost: us per round at 64 steps of the eight-wave / 48-segment / (unused) / 44-segment / 32-segment
 * pipelines for this dimension; per_lane3_us: the 12-dim per-lane kernel with three columns, rounds of 64 x CUs (only for RK4 with
 * 2 ... 5 steps); lane_us: the whole-segment lanes, rounds of 256 x CUs.
 * RK4 with >= 6 steps: the pipelines -- the eight-wave form while the batch is one round of it (16 segments per CU), above that the
 * family whose rounds are cheapest for THIS segment count, a partly filled round costing a whole one.  (Round 6: the per-lane kernel
 * left this table -- its rounds, 246 us per 64 x CUs, never beat the 32-segment pipeline's, 2 x 111 us.)  RK4 with fewer steps: the
 * per-lane kernel (fill and drain phases outweigh the pipelines' shorter phase), on a full chip its whole-segment forms.  13-stage
 * methods: the cooperative kernels; 12-dim DOP853 (the reference's setting) the two-lanes-per-state form. */
static int auto_stm_kernel(int ndim, int method, int steps, int pm, long S, long cus, bool ordered, int cols_per_lane, const double* cost,
                           double per_lane3_us, double lane_us) {
  const auto rounds = [&](long per_round) { return (double)((S + per_round - 1) / per_round); };
  if (method != LTO_RK4)      // DOP853: the two-lanes-per-state forms (12-dim; 14-dim for batches of the always-thrust-limited laws, round 6)
    return (method == LTO_DOP853_ADAPTIVE && (ndim == 12 || indirect_stm_coop2_14_available(pm))) ? LTO_KERNEL_COOP2 : LTO_KERNEL_COOP;
  if (steps < 6) {
    if (steps >= 2 && indirect_stm_lane_available(ndim, method, S) && !ordered && cols_per_lane == 0 &&
        rounds(256 * cus) * lane_us < rounds(64 * cus) * per_lane3_us)
      return LTO_KERNEL_LANE;
    return LTO_KERNEL_PER_LANE;
  }
  if (S <= 16 * cus) return LTO_KERNEL_PIPE8;
  const double t8 = rounds(16 * cus) * cost[0];
  const double t48 = std::min(rounds(48 * cus) * cost[1], ndim == 12 ? rounds(44 * cus) * cost[3] : 1e300);
  const double t32 = indirect_stm_pipe32_available(ndim, pm) ? rounds(32 * cus) * cost[4] : 1e300;
  int kern = (t32 < t8 && t32 < t48) ? LTO_KERNEL_PIPE32 : (t48 <= t8 ? LTO_KERNEL_PIPE48 : LTO_KERNEL_PIPE8);
  // the whole-segment lanes (kernels_indirect_lane.hip, 12-dim): rounds of 256 x CUs segments -- four wavefronts of 64 per CU, one per
  // SIMD.  A partly filled round costs a whole one, so the pipelines keep the sizes just above a multiple of their own, smaller rounds.
  if (indirect_stm_lane_available(ndim, method, S) && !ordered && rounds(256 * cus) * lane_us < std::min(std::min(t8, t48), t32)) kern = LTO_KERNEL_LANE;
  return kern;
}

int lto_indirect_auto_kernel(int ndim, int method, int steps, double p, long n_segments, int n_cus, int ordered) {
  if ((ndim != 12 && ndim != 14) || method < LTO_RK4 || method > LTO_DOP853_ADAPTIVE || n_segments < 1 || n_cus < 1) return LTO_EINVAL;
  if (!(p == 0.0 || p >= 1.0)) return LTO_EINVAL;           // the reference's error("Invalid value of p!") is a run-time code; here: not a plan
  const int pm = 1 << p_class(p);
  return auto_stm_kernel(ndim, method, steps, pm, n_segments, n_cus, ordered != 0, 0, kRoundCostDefault[ndim == 14 ? 1 : 0], kRoundCostDefault[0][2], kLaneRoundUs);
}

// One-step RK4 STM sweeps (SURVEY 8d's HBM-bound corner): from this many segments AUTO's per-lane family runs the form whose lane is
// a whole segment (kernels_indirect_stream.hip): one wavefront of 64 segments per SIMD of an MI355X.  Below, the per-(segment,
// column group) lanes fill the chip with four to twelve times the wavefronts and the sweep is latency-bound either way.
static const long kStreamMinSegments = 65536;

int lto_indirect_jacobian_dev(lto_indirect_plan* p, void* stream, const double* X, long ldx, const double* t,
                              int n_tgrids, double* Phi, long ldp, double* defect, long ldd) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  IndirectArgs a;
  int rc = fill_indirect_args(p, X, ldx, t, n_tgrids, &a);
  if (rc) return rc;
  if (!Phi) return set_err(c, LTO_ENULL, "Phi is NULL");
  if (!p->out_blocks && (ldp < p->S || (defect && ldd < p->S))) return set_err(c, LTO_EINVAL, "ldp/ldd smaller than the segment count");
  a.Phi = Phi; a.ldp = ldp; a.defect = defect; a.ldd = ldd;
  rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  timing_begin(c, st);
  // Kernel choice (DESIGN.md "Kernels"; measured on MI355X with tools/probe_kernels.py, profiles/r03_probe_kernels.txt).
  // RK4 with >= 6 steps per segment: the three-role pipeline kernels.  A workgroup of the eight-wave form owns 16 segments and a
  // CU holds one (91 KB of LDS), so up to 16 x CUs segments (4 096 on MI355X) the sweep is one round -- 14-dim 76 us, 12-dim 66 us
  // against 106 / 89 us (four-wave form, removed), 173 / 136 us per-lane, 238 / 116 us cooperative -- and above that every family
  // runs in rounds of the segments the chip holds at once, a partly filled round costing a whole one: the family with the
  // cheapest rounds for THIS segment count wins (lto_ctx::round_cost: us per round at 64 steps; the ratios do not depend on the step count): the
  // eight-wave form in rounds of 16 x CUs, the 48-segment / 16-wave form in rounds of 48 x CUs (12-dim: 44 x CUs), for 12-dim also the per-lane
  // kernel with 3 columns per lane in rounds of 64 x CUs.  13-stage methods: the wave-specialised kernel (DOP853 @1e-13,
  // 4 096 segments: 0.32 ms vs 1.9 ms per-lane), for the reference's setting (12-dim, DOP853) its two-lanes-per-state form.
  int kern = p->kernel;
  if (kern == LTO_KERNEL_AUTO)
    kern = auto_stm_kernel(p->ndim, p->integ.method, p->integ.steps, p->pm, p->S, c->cu_count > 0 ? c->cu_count : 256, p->use_order != 0, p->cols_per_lane,
                           c->round_cost[p->ndim == 14 ? 1 : 0], c->round_cost[0][2], c->lane_round_us);
  // Families that are gone since round 6 resolve to the one that took over (results agree to round-off, lto_indirect_plan_last_kernel
  // says what ran): the 13-stage methods have no per-lane STM form any more, RK4 no cooperative form, and 12-dim DOP853 only the
  // two-lanes-per-state cooperative form.
  if (p->integ.method != LTO_RK4 && kern == LTO_KERNEL_PER_LANE) kern = LTO_KERNEL_COOP;
  if (p->integ.method == LTO_RK4 && kern == LTO_KERNEL_COOP)
    kern = auto_stm_kernel(p->ndim, LTO_RK4, p->integ.steps < 6 ? 6 : p->integ.steps, p->pm, p->S, c->cu_count > 0 ? c->cu_count : 256, p->use_order != 0, 0,
                           c->round_cost[p->ndim == 14 ? 1 : 0], c->round_cost[0][2], c->lane_round_us);
  if (p->integ.method == LTO_DOP853_ADAPTIVE && p->ndim == 12 && kern == LTO_KERNEL_COOP) kern = LTO_KERNEL_COOP2;
  // the large-batch pipeline has two forms for 12-dim (48 or 44 segments per workgroup, kernels_indirect_pipe48.hip): the cheaper
  // rounds for this segment count, whether AUTO or the caller chose the family
  bool seg44 = false;
  if (kern == LTO_KERNEL_PIPE48 && p->ndim == 12) {
    const long cus = c->cu_count > 0 ? c->cu_count : 256;
    const double* cost = c->round_cost[0];
    const auto rounds = [&](long per_round) { return (double)((p->S + per_round - 1) / per_round); };
    seg44 = p->p48_form ? p->p48_form == 44 : rounds(44 * cus) * cost[3] < rounds(48 * cus) * cost[1];
  }
  p->last_kernel = kern;
  rc = warm_args(p, 0, kern == LTO_KERNEL_COOP2, &a);
  if (rc) return rc;
  p->stm_swept = 1;
  const bool blocks = p->out_blocks != 0;
  if (blocks && kern != LTO_KERNEL_COOP2) return set_err(c, LTO_EUNSUPPORTED, "LTO_LAYOUT_BLOCKS needs the two-lanes-per-state cooperative kernel (LTO_KERNEL_AUTO or _COOP2)");
  if (!blocks && a.order && p->order_kind == 1 && kern == LTO_KERNEL_COOP2 && p->d_xa && p->d_da && !p->d_pa && !p->stage_failed) {
    // the lane order was made before this plan's first STM sweep: the Phi records come now -- unless the stream is being captured
    // (an allocation may not happen there; this sweep then runs unstaged and a later one outside a capture allocates)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) (void)stage_alloc(p, true);
    else (void)hipGetLastError();
  }
  // LTO_LAYOUT_BLOCKS: the caller's Phi / defect arrays are the per-segment records the kernel writes (IndirectArgs::Pa / Da) -- no
  // record arrays of the plan's own and no transposes behind the sweep, with or without a lane order
  const bool staged = a.order && p->order_kind == 1 && kern == LTO_KERNEL_COOP2 && p->d_xa && (blocks || (p->d_da && p->d_pa));
  if (staged) {
    hipError_t q = launch_node_records(X, ldx, t, a.t_stride, p->n_nodes, (long)p->n_nodes * p->n_batch, p->d_xa, st);
    if (q != hipSuccess) return set_err(c, LTO_EHIP, "launch_node_records", q);
    a.Xa = p->d_xa; a.Pa = p->d_pa;
    if (a.defect) a.Da = p->d_da;
  }
  if (blocks) { a.Pa = Phi; if (a.defect) a.Da = defect; }
  hipError_t e;
  if (kern == LTO_KERNEL_COOP) e = launch_indirect_stm_coop(p->ndim, p->pm, p->integ.method, a, st);
  else if (kern == LTO_KERNEL_COOP2) e = (p->ndim == 12) ? launch_indirect_stm_coop2(p->pm, a, st) : launch_indirect_stm_coop2_14(p->pm, a, st);
  else if (kern == LTO_KERNEL_PIPE8) e = launch_indirect_stm_pipe8(p->ndim, p->pm, a, st);
  else if (kern == LTO_KERNEL_PIPE48) e = launch_indirect_stm_pipe48(p->ndim, p->pm, a, seg44, st);
  else if (kern == LTO_KERNEL_PIPE32) e = launch_indirect_stm_pipe32(p->ndim, p->pm, a, st);
  else if (kern == LTO_KERNEL_LANE) e = launch_indirect_stm_lane(p->pm, a, st);
  else if (!a.order && (p->cols_per_lane == p->ndim || (p->cols_per_lane == 0 && p->S >= kStreamMinSegments &&
                                                        indirect_stm_stream_available(p->ndim, p->integ.method, p->integ.steps, p->S))))
    e = launch_indirect_stm_stream(p->ndim, p->pm, a, st);   // one RK4 step on a full chip: lane = segment, HBM-bound (kernels_indirect_stream.hip)
  else e = (p->ndim == 12) ? launch_indirect_stm(p->pm, p->integ.method, p->cols_per_lane, a, st)
                           : launch_indirect14_stm(p->pm, p->integ.method, p->cols_per_lane, a, st);
  if (e == hipSuccess && staged && !blocks) e = launch_pack_soa(p->d_pa, 144, p->S, a.Phi, a.ldp, st);
  if (e == hipSuccess && staged && !blocks && a.Da) e = launch_pack_soa(p->d_da, 12, p->S, a.defect, a.ldd, st);
  timing_end(c, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_indirect_stm", e);
  warm_filled(p, 0, a);
  p->swept = 1;
  return LTO_OK;
}

/* ------------------------------------------------------------------------------ device Newton solve */
int lto_indirect_newton_solve_dev(lto_indirect_plan* p, void* stream, const double* Phi, long ldp, const double* defect,
                                  long ldd, int adjoints_only, double* delta, long ldx) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  if (p->ndim != 12) return set_err(c, LTO_EUNSUPPORTED, "device Newton solve is built for ndim = 12");
  if (p->out_blocks) return set_err(c, LTO_EUNSUPPORTED, "device Newton solve reads struct-of-arrays Phi / defect: use a plan with LTO_LAYOUT_SOA");
  if (!defect || !delta) return set_err(c, LTO_ENULL, "defect or delta is NULL");
  if (ldd < p->S || (Phi && ldp < p->S) || ldx < (long)p->n_nodes * p->n_batch) return set_err(c, LTO_EINVAL, "leading dimension too small");
  int rc = bind_device(c);
  if (rc) return rc;
  if (!p->d_bvp) {
    if (!Phi) return set_err(c, LTO_EINVAL, "no factorisation yet: the first solve needs Phi");
    p->bvp_bytes = sizeof(double) * bvp_workspace_doubles(p->n_nodes, p->n_batch);
    hipError_t e = pool_alloc(c, (void**)&p->d_bvp, p->bvp_bytes);
    if (e != hipSuccess) { p->d_bvp = nullptr; return set_err(c, LTO_EHIP, "newton workspace", e); }
  }
  const int variant = adjoints_only ? 1 : 0;
  if (!Phi && p->bvp_variant != variant) return set_err(c, LTO_EINVAL, "re-solve requested for a variant that was not factored");
  hipError_t e = launch_bvp_solve(Phi, ldp, defect, ldd, p->n_nodes, p->n_batch, variant, p->d_bvp, delta, ldx, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_bvp_solve", e);
  if (Phi) p->bvp_variant = variant;
  return LTO_OK;
}

int lto_axpy_dev(lto_ctx* c, void* stream, const double* x, const double* d, double alpha, double* y, long count) {
  if (!c) return LTO_ENULL;
  if (!x || !d || !y) return set_err(c, LTO_ENULL, "x, d or y is NULL");
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = launch_axpy(x, d, alpha, y, count, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_axpy", e);
  return LTO_OK;
}

int lto_trial_points_dev(lto_ctx* c, void* stream, const double* X, const double* delta, long ld, int ndim, int n_nodes, int n_batch,
                         int n_alpha, const double* alphas, double* Xt, long ldt) {
  if (!c) return LTO_ENULL;
  if (!X || !delta || !alphas || !Xt) return set_err(c, LTO_ENULL, "X, delta, alphas or Xt is NULL");
  if (ndim < 1 || n_nodes < 1 || n_batch < 1 || n_alpha < 1) return set_err(c, LTO_EINVAL, "ndim, n_nodes, n_batch and n_alpha must be positive");
  if (ld < (long)n_nodes * n_batch || ldt < (long)n_nodes * n_batch * n_alpha) return set_err(c, LTO_EINVAL, "leading dimension too small");
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = launch_trial_points(X, delta, ld, ndim, n_nodes, n_batch, n_alpha, alphas, Xt, ldt, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_trial_points", e);
  return LTO_OK;
}

// Scalars of the Newton loop to the host: a[0..na) then b[0..nb) into out.  With the mapped landing block one small kernel
// writes them and the host polls the sequence word (a few microseconds after the kernel); a stream that has drained without
// the word arriving is an error.  Without the block: two copies and a stream synchronisation (about 30 us).
static bool report_reserve(lto_ctx* c, size_t doubles) {
  if (c->rep_host && c->rep_doubles >= doubles) return true;
  if (c->rep_host) { (void)hipDeviceSynchronize(); (void)hipHostFree(c->rep_host); c->rep_host = nullptr; c->rep_doubles = 0; }   // (any stream may have carried the last report)
  void* hp = nullptr; void* dp = nullptr;
  const size_t want = doubles + 64;
  if (hipHostMalloc(&hp, sizeof(double) * (want + 1), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (hp) (void)hipHostFree(hp);
    return false;
  }
  std::memset(hp, 0, sizeof(double) * (want + 1));
  c->rep_host = (double*)hp; c->rep_dev = (double*)dp; c->rep_doubles = want; c->rep_seq = 0;
  return true;
}
static int read_scalars(lto_ctx* c, hipStream_t st, const double* a, int na, const double* b, int nb, double* out) {
  hipError_t e;
  if (c->rep_host && c->rep_doubles >= (size_t)(na + nb)) {
    const long long seq = ++c->rep_seq;
    e = launch_iter_report(a, na, b, nb, c->rep_dev + 1, (long long*)c->rep_dev, seq, st);
    if (e != hipSuccess) return set_err(c, LTO_EHIP, "report", e);
    volatile long long* w = (volatile long long*)c->rep_host;
    // busy poll (a look at the stream every 16 k reads) for the first 5 ms -- the usual case is microseconds behind the last kernel of
    // an iteration the host enqueued in a fraction of its run time -- then a look and a short sleep per read, so that a sweep that
    // takes seconds does not hold a core at 100 %.  (Counting reads instead of time sent a 0.45 ms iteration into the sleeps: a
    // cached read takes a nanosecond.)
    const auto t_start = std::chrono::steady_clock::now();
    bool slow = false;
    for (unsigned long spin = 1;; ++spin) {
      if (*w == seq) break;
      if (slow || (spin & 0x3fff) == 0) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipErrorNotReady) {
          if (slow) std::this_thread::sleep_for(std::chrono::microseconds(20));
          else slow = std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(5);
          continue;
        }
        if (q == hipSuccess && *w == seq) break;
        return set_err(c, LTO_EHIP, "report: the stream drained without the iteration's scalars", q);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(out, c->rep_host + 1, sizeof(double) * (size_t)(na + nb));
    return LTO_OK;
  }
  e = hipMemcpyAsync(out, a, sizeof(double) * (size_t)na, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && nb > 0) e = hipMemcpyAsync(out + na, b, sizeof(double) * (size_t)nb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e == hipSuccess ? LTO_OK : set_err(c, LTO_EHIP, "norm", e);
}

int lto_line_search_pick_dev(lto_ctx* c, void* stream, const double* sumsq, const double* maxabs, const double* alphas, int n_alpha,
                             const double* trial_defect, long ldt, int ndim, int seg_per_traj, int n_batch, double* step,
                             double* maxabs_out, double* defect, long ldd) {
  if (!c) return LTO_ENULL;
  if (!sumsq || !alphas || !step) return set_err(c, LTO_ENULL, "sumsq, alphas or step is NULL");
  if ((maxabs == nullptr) != (maxabs_out == nullptr) || (trial_defect == nullptr) != (defect == nullptr))
    return set_err(c, LTO_ENULL, "maxabs / maxabs_out and trial_defect / defect come in pairs");
  if (n_alpha < 1 || n_batch < 1 || ndim < 1 || seg_per_traj < 1) return set_err(c, LTO_EINVAL, "n_alpha, n_batch, ndim and seg_per_traj must be positive");
  if (defect && (ldt < (long)seg_per_traj * n_batch * n_alpha || ldd < (long)seg_per_traj * n_batch)) return set_err(c, LTO_EINVAL, "leading dimension too small");
  int rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = defect ? launch_take_trial(trial_defect, ldt, sumsq, nullptr, nullptr, n_alpha, seg_per_traj, ndim, n_batch, defect, ldd, alphas,
                                            step, maxabs, maxabs_out, st)
                        : launch_pick_alpha(sumsq, alphas, n_alpha, nullptr, nullptr, step, n_batch, maxabs, maxabs_out, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "line search pick", e);
  return LTO_OK;
}

int lto_read_scalars_dev(lto_ctx* c, void* stream, const double* a, int na, const double* b, int nb, double* out) {
  if (!c) return LTO_ENULL;
  if (!a || !out || (nb > 0 && !b)) return set_err(c, LTO_ENULL, "a, b or out is NULL");
  if (na < 1 || nb < 0 || (long)na + nb > (1L << 20)) return set_err(c, LTO_EINVAL, "need 1 <= na, 0 <= nb, na + nb <= 2^20");
  int rc = bind_device(c);
  if (rc) return rc;
  (void)report_reserve(c, (size_t)na + nb);
  return read_scalars(c, (hipStream_t)stream, a, na, b, nb, out);
}

/* AUTO's cost table measured on this device: one full round of every RK4 STM family and dimension (16 / 48 / 64 x CUs segments,
 * 64 RK4 steps, one state near the L2 halo orbits in every segment -- fixed-step kernels do the same work whatever the data), after 30 ms of
 * sweeps so that the clocks have settled; the median of five launches. */
int lto_calibrate_kernels(lto_ctx* c) {
  if (!c) return LTO_ENULL;
  int rc = bind_device(c);
  if (rc) return rc;
  const long cus = c->cu_count > 0 ? c->cu_count : 256;
  const long per_round[5] = {16 * cus, 48 * cus, 64 * cus, 44 * cus, 32 * cus};
  const int family_kernel[5] = {LTO_KERNEL_PIPE8, LTO_KERNEL_PIPE48, LTO_KERNEL_PER_LANE, LTO_KERNEL_PIPE48, LTO_KERNEL_PIPE32};
  const long lane_round = 256 * cus;                 // the whole-segment lanes' round (12-dim): the largest batch measured
  const long Smax = lane_round, nmax = Smax + 1;
  hipStream_t st = c->stream;
  LTO_HIP(c, hipStreamSynchronize(st));
  rc = arena_reserve(c, al256(sizeof(double) * 14 * nmax) + al256(sizeof(double) * nmax) + al256(sizeof(double) * 196 * Smax) + al256(sizeof(double) * 14 * Smax) + 4096);
  if (rc) return rc;
  c->arena_top = 0;
  double* d_X = arena_take<double>(c, (size_t)14 * nmax);
  double* d_t = arena_take<double>(c, (size_t)nmax);
  double* d_phi = arena_take<double>(c, (size_t)196 * Smax);
  double* d_def = arena_take<double>(c, (size_t)14 * Smax);
  // a state near the Earth-Moon L2 halo family (0.17 DU from the Moon), small costates; 1 000 kg / lambda_m = 0.1 for the 14-row layout
  const double x12[12] = {1.1599795702248494, 0.0097200000000000, -0.1240184140575570, 0.0087153964800000, -0.2085329310256100, 0.0105833000000000,
                          0.01, -0.02, 0.015, 0.02, 0.01, -0.01};
  lto::HostBuf<double> hX((size_t)14 * nmax), ht((size_t)nmax);
  if (!hX.ok() || !ht.ok()) return set_err(c, LTO_ENOMEM, "lto_calibrate_kernels: out of host memory");
  for (long k = 0; k < nmax; ++k) ht[k] = 0.02 * (double)k;
  hipEvent_t e0, e1;
  LTO_HIP(c, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return set_err(c, LTO_EHIP, "hipEventCreate"); }
  lto_params prm = {0.012150585609624, 384400.0, 375190.25852, 0.05, 1000.0, 1.0, 1.0, 1.0};
  lto_integrator integ; std::memset(&integ, 0, sizeof integ);
  integ.method = LTO_RK4; integ.steps = 64;
  double measured[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 1e300, 1e300, 0}};
  double measured_lane = 0.0;
  for (int di = 0; di < 2 && rc == LTO_OK; ++di) {
    const int nd = di ? 14 : 12;
    for (long k = 0; k < nmax; ++k)
      for (int r = 0; r < nd; ++r) {
        double v;
        if (nd == 12) v = x12[r];
        else v = (r < 6) ? x12[r] : (r == 6) ? 1000.0 : (r < 13) ? x12[r - 1] : 0.1;
        hX[(size_t)r * nmax + k] = v;
      }
    prm.mass = di ? 3000.0 : 1000.0;                 // 14-row layout: the slot carries Isp
    hipError_t e = hipMemcpyAsync(d_X, hX.data(), sizeof(double) * nd * nmax, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, ht.data(), sizeof(double) * nmax, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "calibration upload", e); break; }
    for (int f = 0; f < 6 && rc == LTO_OK; ++f) {     // f = 5: the whole-segment lanes (12-dim only)
      if (nd == 14 && (f == 2 || f == 3 || f == 5)) continue;
      const long S = (f == 5) ? lane_round : per_round[f];      // one full round: with 44 x CUs segments the 44-form is the cheaper one, with 48 x CUs the 48-form
      lto_indirect_plan* p = nullptr;
      rc = plan_build(c, nd, (int)(S + 1), 1, &prm, 1, &integ, &p);
      if (rc) break;
      p->kernel = (f == 5) ? LTO_KERNEL_LANE : family_kernel[f];
      p->p48_form = (f == 3) ? 44 : 48;
      if (f == 2) p->cols_per_lane = 3;
      auto sweep = [&]() { return lto_indirect_jacobian_dev(p, st, d_X, nmax, d_t, 1, d_phi, S, d_def, S); };
      if (di == 0 && f == 0) {                      // let the clocks settle: ~30 ms of sweeps
        const auto t0 = std::chrono::steady_clock::now();
        while (rc == LTO_OK && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.03) {
          for (int q = 0; q < 16 && rc == LTO_OK; ++q) rc = sweep();
          if (rc == LTO_OK && hipStreamSynchronize(st) != hipSuccess) rc = set_err(c, LTO_EHIP, "calibration warm-up");
        }
      }
      double ms[5];
      for (int q = 0; q < 2 && rc == LTO_OK; ++q) rc = sweep();
      for (int q = 0; q < 5 && rc == LTO_OK; ++q) {
        float m = 0.0f;
        if (hipEventRecord(e0, st) != hipSuccess) { rc = set_err(c, LTO_EHIP, "hipEventRecord"); break; }
        rc = sweep();
        if (rc == LTO_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&m, e0, e1) != hipSuccess))
          rc = set_err(c, LTO_EHIP, "calibration timing");
        ms[q] = m;
      }
      plan_free(p);
      if (rc == LTO_OK) {
        std::sort(ms, ms + 5);
        if (f == 5) measured_lane = ms[2] * 1e3 * (64.0 / integ.steps);
        else measured[di][f] = ms[2] * 1e3 * (64.0 / integ.steps);
      }
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc != LTO_OK) return rc;
  for (int di = 0; di < 2; ++di)
    for (int f = 0; f < 5; ++f)
      if (!(measured[di][f] > 0.0)) return set_err(c, LTO_EHIP, "calibration returned a non-positive time");
  if (!(measured_lane > 0.0)) return set_err(c, LTO_EHIP, "calibration returned a non-positive time");
  std::memcpy(c->round_cost, measured, sizeof measured);
  c->lane_round_us = measured_lane;
  c->calibrated = true;
  return LTO_OK;
}

double lto_kernel_lane_round_us(const lto_ctx* c) { return c ? c->lane_round_us : 0.0; }

int lto_kernel_round_costs(const lto_ctx* c, int ndim, double* us_per_round, int* calibrated) {
  if (!c || !us_per_round) return LTO_ENULL;
  if (ndim != 12 && ndim != 14) return LTO_EINVAL;
  for (int f = 0; f < 5; ++f) us_per_round[f] = c->round_cost[ndim == 14 ? 1 : 0][f];
  if (ndim == 14) us_per_round[2] = us_per_round[3] = -1.0;           // not candidates
  if (calibrated) *calibrated = c->calibrated ? 1 : 0;
  return LTO_OK;
}

/* Host-pointer API: adopt / refresh the context's cached lane order (see lto_ctx::order_cache).  Below these sizes
 * one round of wavefronts / workgroups covers the chip and the order cannot matter. */
// Round 4: defect-only sweeps from 16 384 segments (was 131 072).  Every wavefront of such a sweep is resident at once, but a
// wavefront lasts as long as its slowest segment and holds its registers and issue slots until then: with the lanes ordered, the
// line search's 20 x 4 096 segments take 69 instead of 152 us, 20 x 1 024 take 50 instead of 63 (tools/probe_linesearch_lanes.py).
static const long kOrderMinStm = 8192, kOrderMinDefect = 16384;

static bool host_order_wanted(const lto_indirect_plan* p, bool stm) {
  return p->d_nacc && p->S >= (stm ? kOrderMinStm : kOrderMinDefect);
}

static void host_order_adopt(lto_ctx* c, lto_indirect_plan* p, bool stm) {
  c->last_call_order = 0;
  if (p->order_borrowed) { p->d_order = nullptr; p->use_order = 0; p->order_borrowed = 0; }   // cached plan: the context's order may have moved
  const int kind = order_kind_for(stm);
  if (!host_order_wanted(p, stm) || !c->order_cache[kind] || c->order_S[kind] != p->S || c->order_ndim[kind] != p->ndim) return;
  p->d_order = c->order_cache[kind]; p->order_borrowed = 1; p->use_order = 1; p->order_kind = kind;
  c->last_call_order = kind;
  if (kind == 1) (void)stage_alloc(p, stm);
}

static void host_order_refresh(lto_ctx* c, lto_indirect_plan* p, bool stm, hipStream_t st) {
  if (!host_order_wanted(p, stm)) return;
  const int kind = order_kind_for(stm);
  if (!c->order_cache[kind] || c->order_S[kind] != p->S) {
    if (p->use_order) return;                      // (cannot happen: adoption requires a matching cache)
    if (c->order_cache[kind]) { (void)hipStreamSynchronize(st); (void)hipFree(c->order_cache[kind]); c->order_cache[kind] = nullptr; }
    if (hipMalloc((void**)&c->order_cache[kind], order_bytes(p->S)) != hipSuccess) {
      c->order_cache[kind] = nullptr; (void)hipGetLastError();
      return;                                      // balancing is an optimisation: carry on without it
    }
    c->order_S[kind] = p->S;
  }
  c->order_ndim[kind] = p->ndim;
  if (segment_order(kind, p->d_nacc, p->d_nrej, p->S, c->order_cache[kind] + p->S, c->order_cache[kind], st) != hipSuccess) {
    (void)hipGetLastError();
    c->order_S[kind] = 0;                          // never adopt a half-written order
  }
}

/* One Newton iteration of multiShoot_CRTBP_indirect on the device (indirect.jl:290-296; both settings of flag_adjointsOnly):
 * jacobianCalc + the least-squares step of optimizeTraj_OLS (:181-182) + its second-order correction (:190-214).
 * Only XC, t go up and xc_update, defect come down; Phi never leaves HBM. */
int lto_indirect_newton_step(lto_ctx* c, int ndim, int n_nodes, int n_batch, const double* XC, const double* t, int n_tgrids,
                             const lto_params* prm, int n_prm, const lto_integrator* integ, int flag_adjointsOnly,
                             double soc_threshold, double* xc_update, double* defect) {
  if (!c) return LTO_ENULL;
  if (!XC || !t || !xc_update) return set_err(c, LTO_ENULL, "XC, t or xc_update is NULL");
  if (ndim != 12) return set_err(c, LTO_EUNSUPPORTED, "device Newton step is built for ndim = 12");
  if (n_tgrids != 1 && n_tgrids != n_batch) return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  lto_indirect_plan* p = nullptr;
  int rc = host_plan_acquire(c, ndim, n_nodes, n_batch, prm, n_prm, integ, &p);   // cached between calls, owned by the context
  if (rc) return rc;
  const long J = (long)n_nodes * n_batch, S = p->S;
  const size_t need = al256(sizeof(double) * 12 * J) * 5 + al256(sizeof(double) * n_nodes * n_tgrids) +
                      al256(sizeof(double) * 12 * S) * 3 + al256(sizeof(double) * 144 * S) + 16384;
  rc = arena_reserve(c, need);
  if (rc) return rc;
  c->arena_top = 0;
  double* d_aos = arena_take<double>(c, (size_t)12 * J);
  double* d_X = arena_take<double>(c, (size_t)12 * J);
  double* d_X2 = arena_take<double>(c, (size_t)12 * J);
  double* d_del = arena_take<double>(c, (size_t)12 * J);
  double* d_del2 = arena_take<double>(c, (size_t)12 * J);
  double* d_t = arena_take<double>(c, (size_t)n_nodes * n_tgrids);
  double* d_def = arena_take<double>(c, (size_t)12 * S);
  double* d_def2 = arena_take<double>(c, (size_t)12 * S);
  double* d_def_aos = arena_take<double>(c, (size_t)12 * S);
  double* d_phi = arena_take<double>(c, (size_t)144 * S);
  hipStream_t st = c->stream;
  hipError_t e = hipMemcpyAsync(d_aos, XC, sizeof(double) * 12 * J, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_t, t, sizeof(double) * n_nodes * n_tgrids, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_pack_soa(d_aos, 12, J, d_X, J, st);
  if (e != hipSuccess) { (void)hipStreamSynchronize(st); return set_err(c, LTO_EHIP, "stage in", e); }
  host_order_adopt(c, p, true);
  rc = lto_indirect_jacobian_dev(p, st, d_X, J, d_t, n_tgrids, d_phi, S, d_def, S);
  if (rc == LTO_OK) host_order_refresh(c, p, true, st);
  if (rc == LTO_OK) rc = lto_indirect_newton_solve_dev(p, st, d_phi, S, d_def, S, flag_adjointsOnly, d_del, J);
  double* h_del = nullptr;
  if (rc == LTO_OK) {
    h_del = (double*)std::malloc(sizeof(double) * 12 * (size_t)J);
    if (!h_del) rc = set_err(c, LTO_EHIP, "host allocation failed");
  }
  if (rc == LTO_OK) {
    e = hipMemcpyAsync(h_del, d_del, sizeof(double) * 12 * J, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "newton step", e);
  }
  if (rc == LTO_OK) {
    double mx = 0.0;
    bool finite = true;
    for (long k = 0; k < 12 * J; ++k) { const double v = std::fabs(h_del[k]); if (!(v == v)) finite = false; if (v > mx) mx = v; }
    if (finite && mx < soc_threshold) {   // :190  norm(xc_update, Inf) < 1e-1
      e = launch_axpy(d_X, d_del, 1.0, d_X2, 12 * J, st);
      if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "axpy", e);
      if (rc == LTO_OK) rc = lto_indirect_defect_dev(p, st, d_X2, J, d_t, n_tgrids, d_def2, S, nullptr);
      if (rc == LTO_OK) rc = lto_indirect_newton_solve_dev(p, st, nullptr, 0, d_def2, S, flag_adjointsOnly, d_del2, J);
      if (rc == LTO_OK) {
        e = launch_axpy(d_del, d_del2, 1.0, d_del, 12 * J, st);
        if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "axpy", e);
      }
    }
  }
  if (rc == LTO_OK) {
    e = launch_unpack_soa(d_del, J, 12, J, d_aos, st);
    if (e == hipSuccess) e = hipMemcpyAsync(xc_update, d_aos, sizeof(double) * 12 * J, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && defect) {
      e = launch_unpack_soa(d_def, S, 12, S, d_def_aos, st);
      if (e == hipSuccess) e = hipMemcpyAsync(defect, d_def_aos, sizeof(double) * 12 * S, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
  } else {
    (void)hipStreamSynchronize(st);
  }
  std::free(h_del);
  return rc;
}

/* Whole Newton loop of multiShoot_CRTBP_indirect (src/multiShoot_CRTBP_indirect.jl:254-345) with the trajectories
 * resident in HBM: per iteration one STM sweep, the structured least-squares step (+ second-order correction), the
 * 20-point line search as ONE batched sweep after iteration 3, end-state pinning and the defect check.  Only scalars
 * cross PCIe inside the loop (per trajectory: max|xc_update|, 20 sums of squares, max|defect|).
 * n_batch independent problems (homotopy levels, thrust levels, different guesses) run the loop side by side: every
 * device operation covers the whole batch; a trajectory that has left the reference loop (converged, NaN, iteration
 * limit) is frozen by a zero step length and its results are kept. */
int lto_indirect_solve_batch(lto_ctx* c, int ndim, int n_nodes, int n_batch, const double* XC_in, const double* t, int n_tgrids,
                             const lto_params* prm, int n_prm, const lto_integrator* integ, int flag_adjointsOnly, int maxIter,
                             double* XC_out, double* defect, int* status_flag, int* iterations, double* history) {
  if (!c) return LTO_ENULL;
  if (!XC_in || !t || !prm || !integ || !XC_out || !status_flag) return set_err(c, LTO_ENULL, "XC_in, t, prm, integ, XC_out or status_flag is NULL");
  if (ndim != 12) return set_err(c, LTO_EUNSUPPORTED, "the device Newton loop is built for ndim = 12");
  if (maxIter < 0) return set_err(c, LTO_EINVAL, "maxIter must be >= 0");
  if (n_batch < 1 || n_nodes < 2) return set_err(c, LTO_EINVAL, "need n_nodes >= 2 and n_batch >= 1");
  if ((n_tgrids != 1 && n_tgrids != n_batch) || (n_prm != 1 && n_prm != n_batch)) return set_err(c, LTO_EINVAL, "n_tgrids / n_prm must be 1 or n_batch");
  constexpr int NA = 20;                                   // LinRange(0.1, 1, 20), :227
  const int B = n_batch;
  if ((long)B * NA * (n_nodes - 1) > 0x7fffffffL) return set_err(c, LTO_EINVAL, "too many line-search segments");
  // parameters / time grids of the B*NA line-search trial trajectories: trajectory b's, NA times
  lto::HostBuf<lto_params> prm_l;
  lto::HostBuf<double> t_l;
  if (n_prm != 1) {
    if (!prm_l.alloc((size_t)B * NA)) return set_err(c, LTO_ENOMEM, "lto_indirect_solve_batch: out of host memory");
    for (int b = 0; b < B; ++b) for (int a = 0; a < NA; ++a) prm_l[(size_t)b * NA + a] = prm[b];
  }
  if (n_tgrids != 1) {
    if (!t_l.alloc((size_t)B * NA * n_nodes)) return set_err(c, LTO_ENOMEM, "lto_indirect_solve_batch: out of host memory");
    for (int b = 0; b < B; ++b) for (int a = 0; a < NA; ++a)
      std::memcpy(&t_l[((size_t)b * NA + a) * n_nodes], t + (size_t)b * n_nodes, sizeof(double) * n_nodes);
  }
  lto_indirect_plan* p = nullptr;
  lto_indirect_plan* pl = nullptr;
  int rc = plan_build(c, 12, n_nodes, B, prm, n_prm, integ, &p);
  if (rc) return rc;
  rc = plan_build(c, 12, n_nodes, B * NA, n_prm == 1 ? prm : prm_l.data(), n_prm == 1 ? 1 : B * NA, integ, &pl);
  if (rc) { plan_free(p); return rc; }
  const long n = n_nodes, J = n * B, S = (n - 1) * B;
  const int ntl = (n_tgrids == 1) ? 1 : B * NA;
  const size_t n_small = (size_t)12 * B + NA + 6 * (size_t)B + 2 * (size_t)NA * B + 64;
  const size_t need = al256(sizeof(double) * 12 * J) * 5 + al256(sizeof(double) * 12 * J * NA) + al256(sizeof(double) * n * n_tgrids) +
                      al256(sizeof(double) * n * ntl) + al256(sizeof(double) * 12 * S) * 4 + al256(sizeof(double) * 12 * S * NA) +
                      al256(sizeof(double) * 144 * S) + al256(sizeof(double) * n_small) + 65536;
  rc = arena_reserve(c, need);
  if (rc) { plan_free(pl); plan_free(p); return rc; }
  c->arena_top = 0;
  double* d_aos = arena_take<double>(c, (size_t)12 * J);
  double* d_X = arena_take<double>(c, (size_t)12 * J);
  double* d_X2 = arena_take<double>(c, (size_t)12 * J);
  double* d_del = arena_take<double>(c, (size_t)12 * J);
  double* d_del2 = arena_take<double>(c, (size_t)12 * J);
  double* d_Xt = arena_take<double>(c, (size_t)12 * J * NA);
  double* d_t = arena_take<double>(c, (size_t)n * n_tgrids);
  double* d_tl = (n_tgrids == 1) ? d_t : arena_take<double>(c, (size_t)n * ntl);
  double* d_def = arena_take<double>(c, (size_t)12 * S);
  double* d_def2 = arena_take<double>(c, (size_t)12 * S);
  double* d_defj = arena_take<double>(c, (size_t)12 * S);   // the STM sweep's own defect (right-hand side of the step); d_def stays defectCalc's
  double* d_def_aos = arena_take<double>(c, (size_t)12 * S);
  double* d_deft = arena_take<double>(c, (size_t)12 * S * NA);
  double* d_phi = arena_take<double>(c, (size_t)144 * S);
  double* d_small = arena_take<double>(c, n_small);
  double* d_saved = d_small;                               // [B][12] pinned end states
  double* d_alphas = d_saved + (size_t)12 * B;             // [NA]   trial step lengths
  double* d_step = d_alphas + NA;                          // [B]    step length / SOC mask per trajectory
  double* d_mx = d_step + B;                               // [B]    per-trajectory max norms
  double* d_ss = d_mx + B;                                 // [NA*B] per-trial sums of squares
  double* d_act = d_ss + (size_t)NA * B;                   // [B]    1 = trajectory still in its loop
  double* d_search = d_act + B;                            // [B]    1 = line search on (iteration > 3)
  double* d_mxdel = d_search + B;                          // [B]    max |xc_update| of the iteration
  double* d_mxt = d_mxdel + B;                             // [NA*B] per-trial max |defect|
  (void)report_reserve(c, (size_t)3 * B);
  hipStream_t st = c->stream;
  double alphas[NA];
  for (int a = 0; a < NA; ++a) alphas[a] = 0.1 + (1.0 - 0.1) / (NA - 1) * a;
  alphas[NA - 1] = 1.0;
  lto::HostBuf<double> h_mx(B), h_er(B, 1.0), h_step(B), h_back((size_t)3 * B), h_act(B, -1.0), h_search(B, -1.0);   // er = 1.0: :279
  bool soc_speculative = false;
  unsigned trial_sweeps = 0;
  lto::HostBuf<int> it(B, 0), status(B, 0);
  lto::HostBuf<char> active(B, 1);
  if (!h_mx.ok() || !h_er.ok() || !h_step.ok() || !h_back.ok() || !h_act.ok() || !h_search.ok() || !it.ok() || !status.ok() || !active.ok()) {
    plan_free(pl); plan_free(p);
    return set_err(c, LTO_ENOMEM, "lto_indirect_solve_batch: out of host memory");
  }

  hipError_t e = hipMemcpyAsync(d_aos, XC_in, sizeof(double) * 12 * J, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_t, t, sizeof(double) * n * n_tgrids, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && n_tgrids != 1) e = hipMemcpyAsync(d_tl, t_l.data(), sizeof(double) * n * ntl, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_alphas, alphas, sizeof alphas, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_pack_soa(d_aos, 12, J, d_X, J, st);
  if (e == hipSuccess) e = launch_end_states(d_X, J, n_nodes, B, 6, d_saved, 0, st);           // state_0, state_f  (:270-271)
  if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage in", e);

  // per-trajectory max |v| of an SoA block [rows][ld], `per` columns per trajectory -> host (NaN-propagating)
  auto max_abs = [&](const double* v, long ld, long per, double* out) -> int {
    hipError_t q = launch_defect_norms(v, ld, 12, (int)per, B, nullptr, d_mx, st);
    if (q == hipSuccess) q = hipMemcpyAsync(out, d_mx, sizeof(double) * B, hipMemcpyDeviceToHost, st);
    if (q == hipSuccess) q = hipStreamSynchronize(st);
    return q == hipSuccess ? LTO_OK : set_err(c, LTO_EHIP, "norm", q);
  };
  auto any_active = [&]() { for (int b = 0; b < B; ++b) if (active[b]) return true; return false; };

  if (rc == LTO_OK) rc = lto_indirect_defect_dev(p, st, d_X, J, d_t, n_tgrids, d_def, S, nullptr);      // :274
  while (rc == LTO_OK && any_active()) {
    // `while er > 1e-10` (:280) + the iteration limit (:281-286), trajectory by trajectory
    for (int b = 0; b < B; ++b) {
      if (!active[b]) continue;
      if (!(h_er[b] > 1e-10)) { active[b] = 0; continue; }            // converged, or NaN (the comparison is false)
      if (++it[b] > maxIter) { status[b] = 1; active[b] = 0; }
    }
    if (!any_active()) break;
    // Round 4: the loop's decisions are taken on the device -- the second-order-correction mask from max |xc_update| (:190) and the
    // line search's first minimiser (:244-245) -- so the host reads back ONCE per iteration (max |defect|, the step lengths and
    // max |xc_update| together) instead of three times.  While the last known max |xc_update| of some active trajectory is
    // >= 0.1 the correction is still decided on the host (one more read-back, but a defect sweep and a re-solve whose result would
    // be discarded are not launched); once every active trajectory has been below, it is computed for all and applied by mask.
    bool flags_changed = false;
    for (int b = 0; b < B; ++b) {
      const double fa = active[b] ? 1.0 : 0.0, fs = (active[b] && it[b] > 3) ? 1.0 : 0.0;
      if (fa != h_act[b] || fs != h_search[b]) { h_act[b] = fa; h_search[b] = fs; flags_changed = true; }
    }
    if (flags_changed) {
      e = hipMemcpyAsync(d_act, h_act.data(), sizeof(double) * B, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(d_search, h_search.data(), sizeof(double) * B, hipMemcpyHostToDevice, st);
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "flag upload", e); break; }
    }
    rc = lto_indirect_jacobian_dev(p, st, d_X, J, d_t, n_tgrids, d_phi, S, d_defj, S);             // :290
    // large adaptive problems: the next sweeps of this plan run with the lanes ordered by this sweep's step counts
    if (rc == LTO_OK && host_order_wanted(p, true)) rc = lto_indirect_plan_rebalance(p, st);
    if (rc == LTO_OK) rc = lto_indirect_newton_solve_dev(p, st, d_phi, S, d_defj, S, flag_adjointsOnly, d_del, J);  // :182
    if (rc != LTO_OK) break;
    e = launch_defect_norms(d_del, J, 12, (int)n, B, nullptr, d_mxdel, st);                          // max |xc_update| per trajectory
    if (e == hipSuccess) e = launch_soc_mask(d_mxdel, d_act, 1e-1, d_step, B, st);                   // second-order correction, :190-214
    if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "soc mask", e); break; }
    bool soc = true;
    if (!soc_speculative) {                                // early iterations: read max |xc_update| and skip the work if nobody needs it
      rc = read_scalars(c, st, d_mxdel, B, nullptr, 0, h_mx.data());
      if (rc != LTO_OK) break;
      soc = false;
      for (int b = 0; b < B; ++b) soc |= (active[b] && h_mx[b] == h_mx[b] && h_mx[b] < 1e-1);
    }
    if (soc) {
      e = launch_axpy(d_X, d_del, 1.0, d_X2, 12 * J, st);
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "axpy", e); break; }
      rc = lto_indirect_defect_dev(p, st, d_X2, J, d_t, n_tgrids, d_def2, S, nullptr);
      if (rc == LTO_OK) rc = lto_indirect_newton_solve_dev(p, st, nullptr, 0, d_def2, S, flag_adjointsOnly, d_del2, J);
      if (rc != LTO_OK) break;
      e = launch_axpy_traj(d_del, d_del2, d_step, d_del, J, 12, n_nodes, B, st);                      // masked: step = 0 keeps d_del
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "axpy", e); break; }
    }
    bool search = false, all_search = true;
    for (int b = 0; b < B; ++b) if (active[b]) { search |= it[b] > 3; all_search &= it[b] > 3; }
    if (search) {                                          // :300-302: the 20 trial trajectories of every problem, one sweep
      e = launch_trial_points(d_X, d_del, J, 12, n_nodes, B, NA, d_alphas, d_Xt, J * NA, st);
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "trial points", e); break; }
      rc = lto_indirect_defect_dev(pl, st, d_Xt, J * NA, d_tl, ntl, d_deft, S * NA, nullptr);
      // the next trial sweeps run with the lanes ordered by this one's step counts; near convergence the counts hardly move, so the
      // order (always a valid permutation, whatever its age) is renewed every fourth sweep only
      if (rc == LTO_OK && host_order_wanted(pl, false) && (trial_sweeps++ & 3) == 0) rc = lto_indirect_plan_rebalance(pl, st);
      if (rc != LTO_OK) break;
      e = launch_defect_norms(d_deft, S * NA, 12, n_nodes - 1, B * NA, d_ss, d_mxt, st);           // sum(defect.^2), :240 (+ max |defect|)
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "line search", e); break; }
    }
    // alpha (:244-245), 1, or 0 (frozen).  When every active trajectory searched, the same launch takes the chosen trial's max
    // |defect| and defect block: CHECK UPDATE (:328-331) without a sweep -- the new XC_all is the chosen trial point bit for bit
    // (same fma, the update's end-state rows are zero), so defectCalc there is the lanes of the line search's sweep that integrated it.
    const bool reuse = search && all_search;
    e = reuse ? launch_take_trial(d_deft, S * NA, d_ss, d_act, d_search, NA, n_nodes - 1, 12, B, d_def, S, d_alphas, d_step, d_mxt, d_mx, st)
              : launch_pick_alpha(d_ss, d_alphas, NA, d_act, d_search, d_step, B, nullptr, nullptr, st);
    if (e == hipSuccess) e = launch_axpy_traj(d_X, d_del, d_step, d_X, J, 12, n_nodes, B, st);      // :304
    if (e == hipSuccess) e = launch_end_states(d_X, J, n_nodes, B, 6, d_saved, 1, st);             // :324-325
    if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "update", e); break; }
    if (!reuse) {
      rc = lto_indirect_defect_dev(p, st, d_X, J, d_t, n_tgrids, d_def, S, nullptr);               // :328
      if (rc != LTO_OK) break;
      e = launch_defect_norms(d_def, S, 12, (int)(n - 1), B, nullptr, d_mx, st);                     // :331
      if (e != hipSuccess) { rc = set_err(c, LTO_EHIP, "norm", e); break; }
    }
    // one read-back: [step | max |defect|] are adjacent in the small block, max |xc_update| follows the flags
    rc = read_scalars(c, st, d_step, 2 * B, d_mxdel, B, h_back.data());
    if (rc != LTO_OK) break;
    soc_speculative = true;
    for (int b = 0; b < B; ++b) {
      h_step[b] = h_back[b]; h_mx[b] = h_back[B + b];
      const double md = h_back[2 * B + b];
      if (active[b] && !(md < 1e-1)) soc_speculative = false;        // somebody is still taking big steps (or NaN): decide on the host next time
    }
    for (int b = 0; b < B; ++b) {
      if (!active[b]) continue;
      h_er[b] = h_mx[b];
      if (history && it[b] <= maxIter) {
        history[((size_t)b * maxIter + (it[b] - 1)) * 2] = h_er[b];
        history[((size_t)b * maxIter + (it[b] - 1)) * 2 + 1] = h_step[b];
      }
      if (h_er[b] > 1e3) it[b] += 100;                     // "Not likely to converge. Aborting." (:333-336)
    }
  }
  if (rc == LTO_OK) {
    e = launch_unpack_soa(d_X, J, 12, J, d_aos, st);
    if (e == hipSuccess) e = hipMemcpyAsync(XC_out, d_aos, sizeof(double) * 12 * J, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && defect) {
      e = launch_unpack_soa(d_def, S, 12, S, d_def_aos, st);
      if (e == hipSuccess) e = hipMemcpyAsync(defect, d_def_aos, sizeof(double) * 12 * S, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = max_abs(d_def, S, n - 1, h_mx.data()) == LTO_OK ? hipSuccess : hipErrorUnknown;
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
    // :339-341 flags a NaN trajectory; a NaN defect leaves the loop the same way (NaN > 1e-10 is false), so both
    // report status 2 here, as drivers.multiShoot_CRTBP_indirect does
    if (rc == LTO_OK)
      for (int b = 0; b < B; ++b)
        if (XC_out[(size_t)12 * n * b] != XC_out[(size_t)12 * n * b] || h_mx[b] != h_mx[b]) status[b] = 2;
  } else {
    (void)hipStreamSynchronize(st);
  }
  for (int b = 0; b < B; ++b) { status_flag[b] = status[b]; if (iterations) iterations[b] = it[b]; }
  plan_free(pl);
  plan_free(p);
  return rc;
}

int lto_indirect_solve(lto_ctx* c, int ndim, int n_nodes, const double* XC_in, const double* t, const lto_params* prm,
                       const lto_integrator* integ, int flag_adjointsOnly, int maxIter, double* XC_out, double* defect,
                       int* status_flag, int* iterations, double* history) {
  return lto_indirect_solve_batch(c, ndim, n_nodes, 1, XC_in, t, 1, prm, 1, integ, flag_adjointsOnly, maxIter, XC_out, defect,
                                  status_flag, iterations, history);
}

/* ------------------------------------------------------------------------------ dense output */
int lto_indirect_dense_dev(lto_indirect_plan* p, void* stream, const double* X, long ldx, const double* t, int n_tgrids,
                           const int* first, const double* t_samples, double* Y, long ldy, double* final_state) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  IndirectArgs a;
  int rc = fill_indirect_args(p, X, ldx, t, n_tgrids, &a);
  if (rc) return rc;
  if (!first || !t_samples || !Y) return set_err(c, LTO_ENULL, "first, t_samples or Y is NULL");
  // dense output is built for what densify needs (HelperFunctions.jl:51-101 re-propagates with the solver of the sweep: the 12-dim
  // system, DOP853 for Vern8) and for the contract's RK4; round 6 removed the 24 other instantiations, which nothing ran
  if (p->ndim != 12 || (p->integ.method != LTO_RK4 && p->integ.method != LTO_DOP853_ADAPTIVE))
    return set_err(c, LTO_EUNSUPPORTED, "dense output is built for ndim = 12 with LTO_RK4 or LTO_DOP853_ADAPTIVE");
  rc = bind_device(c);
  if (rc) return rc;
  DenseArgs d;
  d.first = first; d.td = t_samples; d.Y = Y; d.ldy = ldy; d.final_state = final_state;
  hipStream_t st = (hipStream_t)stream;
  timing_begin(c, st);
  hipError_t e = launch_indirect_dense(p->ndim, p->pm, p->integ.method, a, d, st);
  timing_end(c, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_indirect_dense", e);
  p->swept = 1;
  return LTO_OK;
}

/* ------------------------------------------------------------------------------ direct plans */
static int direct_plan_build(lto_ctx* c, int nstate, int n_nodes, int n_batch, int nsteps, const lto_direct_params* prm,
                             lto_direct_plan** out) {
  if (!c || !out) return LTO_ENULL;
  *out = nullptr;
  if (!prm) return set_err(c, LTO_ENULL, "prm is NULL");
  if (nstate != 6 && nstate != 7) return set_err(c, LTO_EINVAL, "nstate must be 6 or 7");
  if (n_nodes < 2 || n_batch < 1) return set_err(c, LTO_EINVAL, "need n_nodes >= 2 and n_batch >= 1");
  if (nsteps < 2) return set_err(c, LTO_EINVAL, "nsteps (grid points per half segment) must be >= 2");
  if ((long)(n_nodes - 1) * n_batch > 0x3fffffffL) return set_err(c, LTO_EINVAL, "too many segments");
  lto_direct_plan* p = new (std::nothrow) lto_direct_plan();
  if (!p) return set_err(c, LTO_EHIP, "host allocation failed");
  p->ctx = c; p->nstate = nstate; p->n_nodes = n_nodes; p->n_batch = n_batch; p->S = (n_nodes - 1) * n_batch;
  p->nsteps = nsteps; p->prm = *prm; p->kernel = LTO_KERNEL_AUTO;
  *out = p;
  return LTO_OK;
}

// user-visible direct plans keep the context alive like indirect ones (lto_destroy)
int lto_direct_plan_create(lto_ctx* c, int nstate, int n_nodes, int n_batch, int nsteps, const lto_direct_params* prm,
                           lto_direct_plan** out) {
  const int rc = direct_plan_build(c, nstate, n_nodes, n_batch, nsteps, prm, out);
  if (rc == LTO_OK) ctx_plan_added(c);
  return rc;
}

void lto_direct_plan_destroy(lto_direct_plan* p) {
  if (!p) return;
  lto_ctx* c = p->ctx;
  delete p;
  if (ctx_release(c, OWNER_PLAN)) ctx_free(c);
}

int lto_direct_plan_set_kernel(lto_direct_plan* p, int kernel) {
  if (!p) return LTO_ENULL;
  if (kernel == LTO_KERNEL_COOP)
    return set_err(p->ctx, LTO_EINVAL, "the wave-specialised direct Jacobian kernel was removed in round 3 (never faster than _PER_LANE or _PIPE)");
  if (kernel != LTO_KERNEL_AUTO && kernel != LTO_KERNEL_PER_LANE && kernel != LTO_KERNEL_DIRECT_PIPE)
    return set_err(p->ctx, LTO_EINVAL, "kernel must be LTO_KERNEL_AUTO, _PER_LANE or _DIRECT_PIPE");
  p->kernel = kernel;
  return LTO_OK;
}

static int fill_direct_args(lto_direct_plan* p, const double* X, long ldx, const double* U, long ldu, const double* t,
                            int n_tgrids, DirectArgs* a) {
  lto_ctx* c = p->ctx;
  if (!X || !U || !t) return set_err(c, LTO_ENULL, "X, U or t is NULL");
  const long J = (long)p->n_nodes * p->n_batch;
  if (ldx < J || ldu < J) return set_err(c, LTO_EINVAL, "ldx/ldu smaller than n_nodes*n_batch");
  if (n_tgrids != 1 && n_tgrids != p->n_batch) return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  std::memset(a, 0, sizeof *a);
  a->X = X; a->ldx = ldx; a->U = U; a->ldu = ldu; a->t = t; a->t_stride = (n_tgrids == 1) ? 0 : p->n_nodes;
  a->MU = p->prm.MU;
  a->kk = (p->prm.TU * p->prm.TU) / p->prm.DU / 1e3;  // N/kg -> DU/TU^2   (prop_EP_deriv.jl:32)
  a->isp_g0 = p->prm.Isp * 9.81;                       // prop_EP_deriv.jl:41-42
  a->TU = p->prm.TU;
  a->n_nodes = p->n_nodes; a->seg_per_traj = p->n_nodes - 1; a->S = p->S;
  a->half_steps = p->nsteps - 1;
  return LTO_OK;
}

static int direct_defect_launch(lto_direct_plan* p, void* stream, const double* X, long ldx, const double* U, long ldu,
                                const double* t, int n_tgrids, double* defect, long ldd, double* errors, double* mid,
                                long ldm);

int lto_direct_defect_dev(lto_direct_plan* p, void* stream, const double* X, long ldx, const double* U, long ldu,
                          const double* t, int n_tgrids, double* defect, long ldd, double* errors) {
  if (!p) return LTO_ENULL;
  if (!defect) return set_err(p->ctx, LTO_ENULL, "defect is NULL");
  return direct_defect_launch(p, stream, X, ldx, U, ldu, t, n_tgrids, defect, ldd, errors, nullptr, 0);
}

int lto_direct_midpoints_dev(lto_direct_plan* p, void* stream, const double* X, long ldx, const double* U, long ldu,
                             const double* t, int n_tgrids, double* x_mid, long ldm, double* defect, long ldd,
                             double* errors) {
  if (!p) return LTO_ENULL;
  if (!x_mid) return set_err(p->ctx, LTO_ENULL, "x_mid is NULL");
  if (ldm < p->S) return set_err(p->ctx, LTO_EINVAL, "ldm smaller than the segment count");
  return direct_defect_launch(p, stream, X, ldx, U, ldu, t, n_tgrids, defect, ldd, errors, x_mid, ldm);
}

static int direct_defect_launch(lto_direct_plan* p, void* stream, const double* X, long ldx, const double* U, long ldu,
                                const double* t, int n_tgrids, double* defect, long ldd, double* errors, double* mid,
                                long ldm) {
  lto_ctx* c = p->ctx;
  DirectArgs a;
  int rc = fill_direct_args(p, X, ldx, U, ldu, t, n_tgrids, &a);
  if (rc) return rc;
  if (defect && ldd < p->S) return set_err(c, LTO_EINVAL, "ldd smaller than the segment count");
  a.defect = defect; a.ldd = ldd; a.errors = errors; a.mid = mid; a.ldm = ldm;
  rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  timing_begin(c, st);
  hipError_t e = launch_direct_defect(p->nstate, a, st);
  timing_end(c, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_direct_defect", e);
  return LTO_OK;
}

int lto_direct_jacobian_dev(lto_direct_plan* p, void* stream, const double* X, long ldx, const double* U, long ldu,
                            const double* t, int n_tgrids, double* Jac, long ldj, double* dtf, double* defect, long ldd,
                            double* errors) {
  if (!p) return LTO_ENULL;
  lto_ctx* c = p->ctx;
  DirectArgs a;
  int rc = fill_direct_args(p, X, ldx, U, ldu, t, n_tgrids, &a);
  if (rc) return rc;
  if (!Jac) return set_err(c, LTO_ENULL, "Jac is NULL");
  if (ldj < p->S || ((defect || dtf) && ldd < p->S)) return set_err(c, LTO_EINVAL, "ldj/ldd smaller than the segment count");
  a.Jac = Jac; a.ldj = ldj; a.dtf = dtf; a.defect = defect; a.ldd = ldd; a.errors = errors;
  rc = bind_device(c);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  timing_begin(c, st);
  // measured (MI355X, bench.py --workload c3, ms per sweep; per-lane / wave-specialised / software-pipelined):
  //   29 segments 0.032 / - / 0.047;  2 048: 0.036 / - / 0.049;  4 096: 0.069 / - / 0.051;  8 192: 0.108 / - / 0.055;
  //   16 384 (BASELINE configs[2]): 0.181 / 0.176 / 0.106;  65 536: 0.592 / - / 0.383
  // The pipelined kernel does ~half the arithmetic (the half-arc base state is integrated once per arc, not once per
  // sensitivity column) but needs 10 waves of one workgroup resident per 32 segments: it wins once the per-lane kernel no
  // longer fits the chip in one round.
  int kern = p->kernel;
  if (kern == LTO_KERNEL_AUTO) kern = (p->S >= 3072) ? LTO_KERNEL_DIRECT_PIPE : LTO_KERNEL_PER_LANE;
  hipError_t e = (kern == LTO_KERNEL_DIRECT_PIPE) ? launch_direct_jacobian_pipe(p->nstate, a, st) : launch_direct_jacobian(p->nstate, a, st);
  timing_end(c, st);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_direct_jacobian", e);
  return LTO_OK;
}

/* ------------------------------------------------------------------------------ utilities */
int lto_pack_soa_dev(lto_ctx* c, void* stream, const double* aos, int ndim, long count, double* soa, long ld) {
  if (!c) return LTO_ENULL;
  if (!aos || !soa) return set_err(c, LTO_ENULL, "aos or soa is NULL");
  if (ndim < 1 || count < 0 || ld < count) return set_err(c, LTO_EINVAL, "bad pack dimensions");
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = launch_pack_soa(aos, ndim, count, soa, ld, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_pack_soa", e);
  return LTO_OK;
}

int lto_unpack_soa_dev(lto_ctx* c, void* stream, const double* soa, long ld, int ndim, long count, double* aos) {
  if (!c) return LTO_ENULL;
  if (!aos || !soa) return set_err(c, LTO_ENULL, "aos or soa is NULL");
  if (ndim < 1 || count < 0 || ld < count) return set_err(c, LTO_EINVAL, "bad unpack dimensions");
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = launch_unpack_soa(soa, ld, ndim, count, aos, (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_unpack_soa", e);
  return LTO_OK;
}

int lto_defect_norms_dev(lto_ctx* c, void* stream, const double* defect, long ldd, int ndim, int seg_per_traj,
                         int n_batch, double* sumsq, double* maxabs) {
  if (!c) return LTO_ENULL;
  if (!defect) return set_err(c, LTO_ENULL, "defect is NULL");
  if (ndim < 1 || seg_per_traj < 1 || n_batch < 1 || ldd < (long)seg_per_traj * n_batch)
    return set_err(c, LTO_EINVAL, "bad norm dimensions");
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = launch_defect_norms(defect, ldd, ndim, seg_per_traj, n_batch, sumsq, maxabs,
                                     (hipStream_t)stream);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "launch_defect_norms", e);
  return LTO_OK;
}

/* ------------------------------------------------------------------------------ host-pointer API
 * H2D (Julia layout) -> pack to SoA -> sweep -> unpack -> D2H, all on the context's stream, then one
 * stream synchronise.  The caller's buffers are only touched inside the call.  Pass buffers from lto_host_alloc
 * (page-locked) and the copies are plain DMA at link speed; pageable buffers are staged by the HIP runtime. */

// End of a host-pointer call: poll the stream for up to ~1 ms before blocking in the runtime.  A 4 096-segment sweep is
// over in 0.2 ms, and the wake-up of a blocked hipStreamSynchronize is a visible part of that.
static hipError_t stream_wait(hipStream_t st) {
  const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
  do {
    for (int k = 0; k < 16; ++k) {
      const hipError_t q = hipStreamQuery(st);
      if (q != hipErrorNotReady) return q;
    }
  } while (std::chrono::steady_clock::now() < give_up);
  return hipStreamSynchronize(st);
}

// Device view of a caller's buffer that lies wholly inside a block from lto_host_alloc; nullptr for any other memory.
static double* pinned_view(lto_ctx* c, const double* host, size_t bytes) {
  const char* h = (const char*)host;
  std::lock_guard<std::mutex> lk(c->pinned_mu);
  for (const lto_ctx::Pinned& b : c->pinned) {
    if (b.dev && h >= b.host && bytes <= b.bytes && (size_t)(h - b.host) <= b.bytes - bytes) return (double*)(b.dev + (h - b.host));
  }
  return nullptr;
}
// host AoS [ndim x count] -> device SoA rows of pitch ld.  Page-locked source: the pack kernel reads it over the link;
// otherwise a copy into d_aos first.
static hipError_t stage_in(lto_ctx* c, const double* host, int ndim, long count, double* d_aos, double* d_soa, long ld,
                           hipStream_t st) {
  if (const double* z = pinned_view(c, host, sizeof(double) * (size_t)ndim * count)) return launch_pack_soa(z, ndim, count, d_soa, ld, st);
  hipError_t e = hipMemcpyAsync(d_aos, host, sizeof(double) * (size_t)ndim * count, hipMemcpyHostToDevice, st);
  return e == hipSuccess ? launch_pack_soa(d_aos, ndim, count, d_soa, ld, st) : e;
}
// device SoA -> host AoS [ndim x count]; the unpack kernel writes a page-locked destination directly.
static hipError_t stage_out(lto_ctx* c, const double* d_soa, long ld, int ndim, long count, double* d_aos, double* host,
                            hipStream_t st) {
  if (double* z = pinned_view(c, host, sizeof(double) * (size_t)ndim * count)) return launch_unpack_soa(d_soa, ld, ndim, count, z, st);
  hipError_t e = launch_unpack_soa(d_soa, ld, ndim, count, d_aos, st);
  return e == hipSuccess ? hipMemcpyAsync(host, d_aos, sizeof(double) * (size_t)ndim * count, hipMemcpyDeviceToHost, st) : e;
}
// plain vectors (time grids, per-segment error estimates): a one-row pack / unpack is a copy kernel
static hipError_t vec_in(lto_ctx* c, const double* host, long count, double* dev, hipStream_t st) {
  if (const double* z = pinned_view(c, host, sizeof(double) * (size_t)count)) return launch_pack_soa(z, 1, count, dev, count, st);
  return hipMemcpyAsync(dev, host, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st);
}
static hipError_t vec_out(lto_ctx* c, const double* dev, long count, double* host, hipStream_t st) {
  if (double* z = pinned_view(c, host, sizeof(double) * (size_t)count)) return launch_unpack_soa(dev, count, 1, count, z, st);
  return hipMemcpyAsync(host, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st);
}

// The plan of a host-pointer call: looked up in the context's small cache by (shape, integrator, parameter values),
// built on a miss (least recently used entry replaced).  Owned by the context.
static int host_plan_acquire(lto_ctx* c, int ndim, int n_nodes, int n_batch, const lto_params* prm, int n_prm,
                             const lto_integrator* integ, lto_indirect_plan** out) {
  *out = nullptr;
  if (!prm || !integ) return set_err(c, LTO_ENULL, "prm or integrator is NULL");
  if (n_prm != 1 && n_prm != n_batch) return set_err(c, LTO_EINVAL, "n_prm must be 1 or n_batch");
  lto_ctx::HostPlan* slot = nullptr;               // an empty entry, else the least recently used one
  // the key compares the integrator field by field (the struct has padding a caller need not initialise) with the
  // max_steps default applied, so that 0 and 100000 share a plan
  const int want_max = integ->max_steps <= 0 ? 100000 : integ->max_steps;
  for (auto& h : c->host_plans) {
    const bool same_integ = h.plan && h.integ.method == integ->method && h.integ.steps == integ->steps && h.integ.rtol == integ->rtol &&
                            h.integ.atol == integ->atol && (h.integ.max_steps <= 0 ? 100000 : h.integ.max_steps) == want_max;
    if (h.plan && h.ndim == ndim && h.n_nodes == n_nodes && h.n_batch == n_batch && h.n_prm == n_prm && same_integ &&
        std::memcmp(h.prm, prm, sizeof(lto_params) * (size_t)n_prm) == 0) {
      h.stamp = ++c->stamp;
      *out = h.plan;
      return LTO_OK;
    }
    if (!slot || (slot->plan && (!h.plan || h.stamp < slot->stamp))) slot = &h;
  }
  lto_ctx::HostPlan* lru = slot;
  lto_indirect_plan* p = nullptr;
  int rc = plan_build(c, ndim, n_nodes, n_batch, prm, n_prm, integ, &p);
  if (rc) return rc;
  lto_params* key = (lto_params*)std::malloc(sizeof(lto_params) * (size_t)n_prm);
  if (!key) { plan_free(p); return set_err(c, LTO_EHIP, "host allocation failed"); }
  std::memcpy(key, prm, sizeof(lto_params) * (size_t)n_prm);
  if (lru->plan) {                                  // the evicted plan's blocks are recycled: nothing of it may be in flight
    (void)hipStreamSynchronize(c->stream);
    plan_free(lru->plan);
    std::free(lru->prm);
  }
  lru->plan = p; lru->prm = key; lru->ndim = ndim; lru->n_nodes = n_nodes; lru->n_batch = n_batch; lru->n_prm = n_prm;
  lru->integ = *integ; lru->stamp = ++c->stamp;
  *out = p;
  return LTO_OK;
}

int lto_host_alloc(lto_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return LTO_ENULL;
  *out = nullptr;
  int rc = bind_device(c);
  if (rc) return rc;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return set_err(c, LTO_EHIP, "hipHostMalloc", e);
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, *out, 0) != hipSuccess) { dev = nullptr; (void)hipGetLastError(); }   // still page-locked: the copy engine moves it
  bool listed;
  { std::lock_guard<std::mutex> lk(c->pinned_mu); listed = c->pinned.push({(char*)*out, (char*)dev, bytes ? bytes : 1}); }
  if (listed) {
    std::lock_guard<std::mutex> lk(g_blocks_mu);
    listed = g_blocks.push({*out, c});
  }
  if (!listed) {                                   // out of host memory for the bookkeeping: no block
    { std::lock_guard<std::mutex> lk(c->pinned_mu);
      for (size_t k = 0; k < c->pinned.size(); ++k) if (c->pinned[k].host == (char*)*out) { c->pinned.erase_at(k); break; } }
    (void)hipHostFree(*out);
    *out = nullptr;
    return set_err(c, LTO_ENOMEM, "lto_host_alloc: out of host memory");
  }
  return LTO_OK;
}

// ctx may be NULL (a finalizer that no longer has the handle): the owner is looked up.  Freeing the last block of a context
// whose lto_destroy was deferred completes that destroy.
int lto_host_free(lto_ctx* c, void* ptr) {
  if (!ptr) return LTO_OK;
  lto_ctx* owner = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_blocks_mu);
    owner = host_block_take(ptr);
  }
  if (!owner) return c ? set_err(c, LTO_EINVAL, "lto_host_free: not a block from lto_host_alloc (or freed twice)") : LTO_EINVAL;
  // The entry is neutralised FIRST (no device alias, no size: pinned_view skips it), so that no host-pointer call on another thread
  // can be handed the device view of memory about to be freed; it stays in the list -- and keeps its context alive -- as a "dying"
  // entry while the device work drains and the block is freed, and only that dying entry is erased afterwards: a concurrent
  // lto_host_alloc that is given the same address again adds a LIVE entry with the same .host, which must survive (advisor
  // finding, round 4).
  {
    std::lock_guard<std::mutex> lk(owner->pinned_mu);
    for (auto& b : owner->pinned)
      if (b.host == (char*)ptr && b.dev) { b.dev = nullptr; b.bytes = 0; break; }
  }
  (void)hipSetDevice(owner->device);
  if (!ctx_is_closing(owner)) (void)hipStreamSynchronize(owner->stream);   // a sweep may still be writing the block in place
  else (void)hipDeviceSynchronize();
  const hipError_t e = hipHostFree(ptr);
  {
    std::lock_guard<std::mutex> lk(owner->pinned_mu);
    for (size_t k = 0; k < owner->pinned.size(); ++k)
      if (owner->pinned[k].host == (char*)ptr && !owner->pinned[k].dev) { owner->pinned.erase_at(k); break; }
  }
  if (ctx_release(owner, OWNER_BLOCK)) { ctx_free(owner); return e == hipSuccess ? LTO_OK : LTO_EHIP; }
  if (e != hipSuccess) return set_err(owner, LTO_EHIP, "hipHostFree", e);
  return LTO_OK;
}

int lto_indirect_defect(lto_ctx* c, int ndim, int n_nodes, int n_batch, const double* XC, const double* t, int n_tgrids,
                        const lto_params* prm, int n_prm, const lto_integrator* integ, double* defect, double* errors) {
  CallTimer call_timer(c);
  if (!c) return LTO_ENULL;
  if (!XC || !t || !defect) return set_err(c, LTO_ENULL, "XC, t or defect is NULL");
  if (n_tgrids != 1 && n_tgrids != n_batch) return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  lto_indirect_plan* p = nullptr;
  int rc = host_plan_acquire(c, ndim, n_nodes, n_batch, prm, n_prm, integ, &p);   // cached between calls, owned by the context
  if (rc) return rc;
  const long J = (long)n_nodes * n_batch, S = p->S;
  const size_t need = al256(sizeof(double) * ndim * J) * 2 + al256(sizeof(double) * n_nodes * n_tgrids) +
                      al256(sizeof(double) * ndim * S) * 2 + al256(sizeof(double) * S) + 4096;
  rc = arena_reserve(c, need);
  if (rc) return rc;
  c->arena_top = 0;
  double* d_aos = arena_take<double>(c, (size_t)ndim * J);
  double* d_X = arena_take<double>(c, (size_t)ndim * J);
  double* d_t = arena_take<double>(c, (size_t)n_nodes * n_tgrids);
  double* d_def = arena_take<double>(c, (size_t)ndim * S);
  double* d_def_aos = arena_take<double>(c, (size_t)ndim * S);
  double* d_err = arena_take<double>(c, (size_t)S);
  hipStream_t st = c->stream;
  hipError_t e = stage_in(c, XC, ndim, J, d_aos, d_X, J, st);
  if (e == hipSuccess) e = vec_in(c, t, (long)n_nodes * n_tgrids, d_t, st);
  if (e != hipSuccess) { (void)hipStreamSynchronize(st); return set_err(c, LTO_EHIP, "stage in", e); }
  host_order_adopt(c, p, false);
  rc = lto_indirect_defect_dev(p, st, d_X, J, d_t, n_tgrids, d_def, S, errors ? d_err : nullptr);
  if (rc == LTO_OK) host_order_refresh(c, p, false, st);
  if (rc == LTO_OK) {
    e = stage_out(c, d_def, S, ndim, S, d_def_aos, defect, st);
    if (e == hipSuccess && errors) e = vec_out(c, d_err, S, errors, st);
    if (e == hipSuccess) e = stream_wait(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
  } else {
    (void)hipStreamSynchronize(st);
  }
  return rc;
}

int lto_indirect_jacobian(lto_ctx* c, int ndim, int n_nodes, int n_batch, const double* XC, const double* t, int n_tgrids,
                          const lto_params* prm, int n_prm, const lto_integrator* integ, double* Phi, double* defect) {
  CallTimer call_timer(c);
  if (!c) return LTO_ENULL;
  if (!XC || !t || !Phi) return set_err(c, LTO_ENULL, "XC, t or Phi is NULL");
  if (n_tgrids != 1 && n_tgrids != n_batch) return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  lto_indirect_plan* p = nullptr;
  int rc = host_plan_acquire(c, ndim, n_nodes, n_batch, prm, n_prm, integ, &p);   // cached between calls, owned by the context
  if (rc) return rc;
  const long J = (long)n_nodes * n_batch, S = p->S;
  const int nn = ndim * ndim;
  const size_t need = al256(sizeof(double) * ndim * J) * 2 + al256(sizeof(double) * n_nodes * n_tgrids) +
                      al256(sizeof(double) * ndim * S) * 2 + al256(sizeof(double) * nn * S) * 2 + 4096;
  rc = arena_reserve(c, need);
  if (rc) return rc;
  c->arena_top = 0;
  double* d_aos = arena_take<double>(c, (size_t)ndim * J);
  double* d_X = arena_take<double>(c, (size_t)ndim * J);
  double* d_t = arena_take<double>(c, (size_t)n_nodes * n_tgrids);
  double* d_def = arena_take<double>(c, (size_t)ndim * S);
  double* d_def_aos = arena_take<double>(c, (size_t)ndim * S);
  double* d_phi = arena_take<double>(c, (size_t)nn * S);
  double* d_phi_aos = arena_take<double>(c, (size_t)nn * S);
  hipStream_t st = c->stream;
  // all operands page-locked: node array and time grid come in with one launch, STM and defect leave with one
  const long nt = (long)n_nodes * n_tgrids;
  const double* zX = pinned_view(c, XC, sizeof(double) * (size_t)ndim * J);
  const double* zt = pinned_view(c, t, sizeof(double) * (size_t)nt);
  double* zPhi = pinned_view(c, Phi, sizeof(double) * (size_t)nn * S);
  double* zdef = defect ? pinned_view(c, defect, sizeof(double) * (size_t)ndim * S) : nullptr;
  hipError_t e;
  if (zX && zt) {
    e = launch_pack_soa2(zX, ndim, J, d_X, J, zt, 1, nt, d_t, nt, st);
  } else {
    e = stage_in(c, XC, ndim, J, d_aos, d_X, J, st);
    if (e == hipSuccess) e = vec_in(c, t, nt, d_t, st);
  }
  if (e != hipSuccess) { (void)hipStreamSynchronize(st); return set_err(c, LTO_EHIP, "stage in", e); }
  host_order_adopt(c, p, true);
  rc = lto_indirect_jacobian_dev(p, st, d_X, J, d_t, n_tgrids, d_phi, S, d_def, S);
  if (rc == LTO_OK) host_order_refresh(c, p, true, st);
  if (rc == LTO_OK) {
    if (zPhi && zdef) {
      e = launch_unpack_soa2(d_phi, S, nn, S, zPhi, d_def, S, ndim, S, zdef, st);
    } else {
      e = stage_out(c, d_phi, S, nn, S, d_phi_aos, Phi, st);
      if (e == hipSuccess && defect) e = stage_out(c, d_def, S, ndim, S, d_def_aos, defect, st);
    }
    if (e == hipSuccess) e = stream_wait(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
  } else {
    (void)hipStreamSynchronize(st);
  }
  return rc;
}

/* densify of src/HelperFunctions.jl:51-101 for one trajectory: t_dense = LinRange(t[1], t[end], n_desired); every
 * segment is re-propagated and sampled at the t_dense points inside [t_i, t_{i+1}); the final propagated state is
 * appended (:94-97).  XC_dense [ndim x n_desired], t_dense [n_desired]. */
int lto_indirect_densify(lto_ctx* c, int ndim, int n_nodes, const double* XC, const double* t, const lto_params* prm,
                         const lto_integrator* integ, int n_desired, double* XC_dense, double* t_dense) {
  if (!c) return LTO_ENULL;
  if (!XC || !t || !XC_dense || !t_dense) return set_err(c, LTO_ENULL, "XC, t, XC_dense or t_dense is NULL");
  if (n_desired < 2) return set_err(c, LTO_EINVAL, "n_desired must be >= 2");
  lto_indirect_plan* p = nullptr;
  int rc = plan_build(c, ndim, n_nodes, 1, prm, 1, integ, &p);
  if (rc) return rc;
  const int S = p->S;
  int* h_first = (int*)std::malloc(sizeof(int) * (size_t)(S + 1));
  if (!h_first) { plan_free(p); return set_err(c, LTO_EHIP, "host allocation failed"); }
  const double t0 = t[0], tn = t[n_nodes - 1];
  for (int k = 0; k < n_desired; ++k) {
    const double tau = (double)k / (double)(n_desired - 1);
    t_dense[k] = (1.0 - tau) * t0 + tau * tn;
  }
  // samples of segment i: t_dense in [t_i, t_{i+1}); the last grid point (== t_n) is served by the final state
  int j = 0;
  for (int i = 0; i < S; ++i) {
    h_first[i] = j;
    while (j < n_desired - 1 && t_dense[j] < t[i + 1]) ++j;
  }
  h_first[S] = n_desired - 1;
  const long J = n_nodes;
  const size_t need = al256(sizeof(double) * ndim * J) * 2 + al256(sizeof(double) * n_nodes) + al256(sizeof(int) * (S + 1)) +
                      al256(sizeof(double) * n_desired) * 2 + al256(sizeof(double) * ndim * n_desired) * 2 + 8192;
  rc = arena_reserve(c, need);
  if (rc) { std::free(h_first); plan_free(p); return rc; }
  c->arena_top = 0;
  double* d_aos = arena_take<double>(c, (size_t)ndim * J);
  double* d_X = arena_take<double>(c, (size_t)ndim * J);
  double* d_t = arena_take<double>(c, (size_t)n_nodes);
  int* d_first = arena_take<int>(c, (size_t)S + 1);
  double* d_td = arena_take<double>(c, (size_t)n_desired);
  double* d_Y = arena_take<double>(c, (size_t)ndim * n_desired);
  double* d_Yaos = arena_take<double>(c, (size_t)ndim * n_desired);
  hipStream_t st = c->stream;
  hipError_t e = hipMemcpyAsync(d_aos, XC, sizeof(double) * ndim * J, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_t, t, sizeof(double) * n_nodes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_first, h_first, sizeof(int) * (S + 1), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_td, t_dense, sizeof(double) * n_desired, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_pack_soa(d_aos, ndim, J, d_X, J, st);
  if (e != hipSuccess) { (void)hipStreamSynchronize(st); std::free(h_first); plan_free(p); return set_err(c, LTO_EHIP, "stage in", e); }
  // the final state lands in the last column of Y: final_state[c * n_batch + traj] with ld = n_desired, offset n_desired-1
  // is not expressible through the [ND][n_batch] layout, so take it into the tail of d_Yaos and splice on the host side
  double* d_final = d_Yaos;   // [ndim] (n_batch = 1); overwritten by the unpack afterwards, so copy it out first
  rc = lto_indirect_dense_dev(p, st, d_X, J, d_t, 1, d_first, d_td, d_Y, n_desired, d_final);
  if (rc == LTO_OK) {
    // splice: Y[c][n_desired-1] = final[c]
    for (int cc = 0; cc < ndim && e == hipSuccess; ++cc)
      e = hipMemcpyAsync(d_Y + (size_t)cc * n_desired + (n_desired - 1), d_final + cc, sizeof(double), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = launch_unpack_soa(d_Y, n_desired, ndim, n_desired, d_Yaos, st);
    if (e == hipSuccess) e = hipMemcpyAsync(XC_dense, d_Yaos, sizeof(double) * ndim * n_desired, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = stream_wait(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
  } else {
    (void)hipStreamSynchronize(st);
  }
  std::free(h_first);
  plan_free(p);
  return rc;
}

static int direct_host(lto_ctx* c, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                       int n_tgrids, int nsteps, const lto_direct_params* prm, double* Jac_temp, double* ddefect_dtf,
                       double* defect, double* errors, bool want_jac, double* x_mid = nullptr) {
  CallTimer call_timer(c);
  lto_direct_plan* p = nullptr;
  int rc = direct_plan_build(c, nstate, n_nodes, n_batch, nsteps, prm, &p);
  if (rc) return rc;
  if (n_tgrids != 1 && n_tgrids != n_batch) { delete p; return set_err(c, LTO_EINVAL, "n_tgrids must be 1 or n_batch"); }
  const long J = (long)n_nodes * n_batch, S = p->S;
  const int nvar = 2 * (nstate + 3), nj = nstate * nvar;
  size_t need = al256(sizeof(double) * nstate * J) * 2 + al256(sizeof(double) * 3 * J) * 2 +
                al256(sizeof(double) * n_nodes * n_tgrids) + al256(sizeof(double) * nstate * S) * 4 +
                al256(sizeof(double) * S) + 8192;
  if (want_jac) need += al256(sizeof(double) * nj * S) * 2;
  rc = arena_reserve(c, need);
  if (rc) { delete p; return rc; }
  c->arena_top = 0;
  double* d_xa = arena_take<double>(c, (size_t)nstate * J);
  double* d_X = arena_take<double>(c, (size_t)nstate * J);
  double* d_ua = arena_take<double>(c, (size_t)3 * J);
  double* d_U = arena_take<double>(c, (size_t)3 * J);
  double* d_t = arena_take<double>(c, (size_t)n_nodes * n_tgrids);
  double* d_def = arena_take<double>(c, (size_t)nstate * S);
  double* d_def_aos = arena_take<double>(c, (size_t)nstate * S);
  double* d_dtf = arena_take<double>(c, (size_t)nstate * S);
  double* d_dtf_aos = arena_take<double>(c, (size_t)nstate * S);
  double* d_err = arena_take<double>(c, (size_t)S);
  double* d_jac = want_jac ? arena_take<double>(c, (size_t)nj * S) : nullptr;
  double* d_jac_aos = want_jac ? arena_take<double>(c, (size_t)nj * S) : nullptr;
  hipStream_t st = c->stream;
  hipError_t e = stage_in(c, X, nstate, J, d_xa, d_X, J, st);
  if (e == hipSuccess) e = stage_in(c, U, 3, J, d_ua, d_U, J, st);
  if (e == hipSuccess) e = vec_in(c, t, (long)n_nodes * n_tgrids, d_t, st);
  if (e != hipSuccess) { delete p; return set_err(c, LTO_EHIP, "stage in", e); }
  if (want_jac)
    rc = lto_direct_jacobian_dev(p, st, d_X, J, d_U, J, d_t, n_tgrids, d_jac, S, d_dtf, d_def, S, d_err);
  else   // the dtf staging buffers are free on this path: they carry the mid-point states
    rc = direct_defect_launch(p, st, d_X, J, d_U, J, d_t, n_tgrids, d_def, S, d_err, x_mid ? d_dtf : nullptr, S);
  if (rc == LTO_OK) {
    if (x_mid) e = stage_out(c, d_dtf, S, nstate, S, d_dtf_aos, x_mid, st);
    if (e == hipSuccess && defect) e = stage_out(c, d_def, S, nstate, S, d_def_aos, defect, st);
    if (e == hipSuccess && errors) e = vec_out(c, d_err, S, errors, st);
    if (e == hipSuccess && want_jac) {
      e = stage_out(c, d_jac, S, nj, S, d_jac_aos, Jac_temp, st);
      if (e == hipSuccess && ddefect_dtf) e = stage_out(c, d_dtf, S, nstate, S, d_dtf_aos, ddefect_dtf, st);
    }
    if (e == hipSuccess) e = stream_wait(st);
    if (e != hipSuccess) rc = set_err(c, LTO_EHIP, "stage out", e);
  } else {
    (void)hipStreamSynchronize(st);
  }
  delete p;
  return rc;
}

int lto_direct_defect(lto_ctx* c, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                      int n_tgrids, int nsteps, const lto_direct_params* prm, double* defect, double* errors) {
  if (!c) return LTO_ENULL;
  if (!X || !U || !t || !defect) return set_err(c, LTO_ENULL, "X, U, t or defect is NULL");
  return direct_host(c, nstate, n_nodes, n_batch, X, U, t, n_tgrids, nsteps, prm, nullptr, nullptr, defect, errors, false);
}

int lto_direct_jacobian(lto_ctx* c, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                        int n_tgrids, int nsteps, const lto_direct_params* prm, double* Jac_temp, double* ddefect_dtf,
                        double* defect, double* errors) {
  if (!c) return LTO_ENULL;
  if (!X || !U || !t || !Jac_temp) return set_err(c, LTO_ENULL, "X, U, t or Jac_temp is NULL");
  return direct_host(c, nstate, n_nodes, n_batch, X, U, t, n_tgrids, nsteps, prm, Jac_temp, ddefect_dtf, defect, errors, true);
}

int lto_direct_midpoints(lto_ctx* c, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                         int n_tgrids, int nsteps, const lto_direct_params* prm, double* x_mid, double* defect, double* errors) {
  if (!c) return LTO_ENULL;
  if (!X || !U || !t || !x_mid) return set_err(c, LTO_ENULL, "X, U, t or x_mid is NULL");
  return direct_host(c, nstate, n_nodes, n_batch, X, U, t, n_tgrids, nsteps, prm, nullptr, nullptr, defect, errors, false, x_mid);
}

}  // extern "C"
