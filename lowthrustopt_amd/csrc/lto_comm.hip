// lto_comm.hip -- the one exchange step of the path: RCCL collectives over xGMI on device-resident defect slabs.
//
// Segments are independent given their nodes, so a sweep shards with no data-path collective; what every rank needs
// afterwards is the FULL defect vector (or its norms) for the convergence test and the line-search decision of the Newton
// loop (src/multiShoot_CRTBP_indirect.jl:240 sum(defect.^2), :331 norm(defect, Inf)).  This file puts that exchange in
// the product, device to device:
//
//   lto_comm_*         one process per GPU (torchrun / MPI ranks): ncclCommInitRank on the context's device with an id the
//                      launcher distributes; all-gather of equal slabs and in-place all-reduce (sum / max) on a caller stream
//   lto_group_comm_*   one host process, several GPUs (lto_group): ncclCommInitAll over the group's devices; the same two
//                      collectives on every context's stream inside ncclGroupStart / End.  A group that repeats a device
//                      (how a 1-GPU box exercises the sharding) cannot form an RCCL clique: there the gather is device
//                      copies ordered by events, the reduce a small kernel per buffer -- same results, no RCCL call.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): the library loads and every other entry point works on a machine
// without it; the collectives then return LTO_EUNSUPPORTED.  When PyTorch is in the process its librccl is already mapped
// and the loader hands back that one (same SONAME), so both speak to the same runtime.
// The payload is tiny (12 x S/G doubles per rank: 49 KB at 4 096 segments on 8 GPUs), the collective is latency-bound.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "../../include/lto.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      q.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (q.lib) break;
    }
    if (!q.lib) return q;
    auto sym = [&](const char* n) { return dlsym(q.lib, n); };
    q.GetUniqueId = (decltype(q.GetUniqueId))sym("ncclGetUniqueId");
    q.CommInitRank = (decltype(q.CommInitRank))sym("ncclCommInitRank");
    q.CommInitAll = (decltype(q.CommInitAll))sym("ncclCommInitAll");
    q.CommDestroy = (decltype(q.CommDestroy))sym("ncclCommDestroy");
    q.AllGather = (decltype(q.AllGather))sym("ncclAllGather");
    q.AllReduce = (decltype(q.AllReduce))sym("ncclAllReduce");
    q.GroupStart = (decltype(q.GroupStart))sym("ncclGroupStart");
    q.GroupEnd = (decltype(q.GroupEnd))sym("ncclGroupEnd");
    q.GetErrorString = (decltype(q.GetErrorString))sym("ncclGetErrorString");
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommInitAll && q.CommDestroy && q.AllGather && q.AllReduce && q.GroupStart &&
           q.GroupEnd && q.GetErrorString;
    return q;
  }();
  return r;
}

// in-place element-wise reduction of `n` buffers of `count` doubles that live on ONE device (repeated-device groups)
__global__ void k_reduce_buffers(double* const* bufs, int n, long count, int op) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double v = bufs[0][i];
  for (int k = 1; k < n; ++k) {
    const double w = bufs[k][i];
    if (op == LTO_COMM_SUM) v += w;
    else v = (v != v || w != w) ? (v + w) : (w > v ? w : v);   // NaN-propagating max, as norm(defect, Inf) in Julia
  }
  for (int k = 0; k < n; ++k) bufs[k][i] = v;
}

}  // namespace

struct lto_comm {
  int device = 0, world = 1, rank = 0;
  ncclComm_t comm = nullptr;
  char err[512] = {0};
};

struct lto_group_comm {
  std::vector<lto_ctx*> ctx;         // borrowed from the group
  std::vector<int> device;
  std::vector<ncclComm_t> comm;      // empty when the group repeats a device
  std::vector<hipEvent_t> ev;        // one per context: producer stream -> consumer streams (copy path)
  double** d_ptrs = nullptr;         // device array of buffer pointers for k_reduce_buffers (copy path)
  bool clique = false;
  char err[512] = {0};
};

namespace {

template <class T>
int comm_fail(T* c, int code, const char* what, ncclResult_t r) {
  std::snprintf(c->err, sizeof c->err, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "rccl error");
  return code;
}
template <class T>
int comm_fail(T* c, int code, const char* what) {
  std::snprintf(c->err, sizeof c->err, "%s", what);
  return code;
}

}  // namespace

extern "C" {

int lto_comm_available(void) { return rccl().ok ? 1 : 0; }

int lto_comm_unique_id(void* id128) {
  if (!id128) return LTO_ENULL;
  if (!rccl().ok) return LTO_EUNSUPPORTED;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return LTO_EHIP;
  static_assert(sizeof id == LTO_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof id);
  return LTO_OK;
}

int lto_comm_create(lto_ctx* ctx, int world, int rank, const void* id128, lto_comm** out) {
  if (!ctx || !id128 || !out) return LTO_ENULL;
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return LTO_EINVAL;
  if (!rccl().ok) return LTO_EUNSUPPORTED;
  lto_comm* c = new (std::nothrow) lto_comm();
  if (!c) return LTO_EHIP;
  c->device = lto_ctx_device(ctx); c->world = world; c->rank = rank;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  if (hipSetDevice(c->device) != hipSuccess) { delete c; return LTO_EHIP; }
  const ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return LTO_EHIP; }
  *out = c;
  return LTO_OK;
}

void lto_comm_destroy(lto_comm* c) {
  if (!c) return;
  if (c->comm) { (void)hipSetDevice(c->device); (void)rccl().CommDestroy(c->comm); }
  delete c;
}

const char* lto_comm_last_error(const lto_comm* c) { return c ? c->err : "null communicator"; }
int lto_comm_size(const lto_comm* c) { return c ? c->world : 0; }
int lto_comm_rank(const lto_comm* c) { return c ? c->rank : -1; }

/* recv [world][count] <- every rank's send [count]; asynchronous on `stream` of this rank's device. */
int lto_comm_allgather_dev(lto_comm* c, void* stream, const double* send, double* recv, long count) {
  if (!c || !send || !recv) return LTO_ENULL;
  if (count < 0) return comm_fail(c, LTO_EINVAL, "count < 0");
  if (count == 0) return LTO_OK;
  if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, LTO_EHIP, "hipSetDevice");
  const ncclResult_t r = rccl().AllGather(send, recv, (size_t)count, ncclDouble, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return comm_fail(c, LTO_EHIP, "ncclAllGather", r);
  return LTO_OK;
}

/* buf [count] <- sum / max over ranks, in place; asynchronous on `stream`.  (RCCL's max does not promise NaN
 * propagation: reduce a NaN count next to the maxima if the driver's status_flag = 2 path matters.) */
int lto_comm_allreduce_dev(lto_comm* c, void* stream, double* buf, long count, int op) {
  if (!c || !buf) return LTO_ENULL;
  if (count < 0 || (op != LTO_COMM_SUM && op != LTO_COMM_MAX)) return comm_fail(c, LTO_EINVAL, "bad count or op");
  if (count == 0) return LTO_OK;
  if (hipSetDevice(c->device) != hipSuccess) return comm_fail(c, LTO_EHIP, "hipSetDevice");
  const ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, op == LTO_COMM_SUM ? ncclSum : ncclMax, c->comm,
                                          (hipStream_t)stream);
  if (r != ncclSuccess) return comm_fail(c, LTO_EHIP, "ncclAllReduce", r);
  return LTO_OK;
}

/* ------------------------------------------------------------------------------ one process, several GPUs */
int lto_group_comm_create(lto_group* g, lto_group_comm** out) {
  if (!g || !out) return LTO_ENULL;
  *out = nullptr;
  const int n = lto_group_size(g);
  if (n < 1) return LTO_EINVAL;
  lto_group_comm* q = new (std::nothrow) lto_group_comm();
  if (!q) return LTO_EHIP;
  bool distinct = true;
  for (int k = 0; k < n; ++k) {
    lto_ctx* c = lto_group_ctx(g, k);
    q->ctx.push_back(c);
    q->device.push_back(lto_ctx_device(c));
    for (int m = 0; m < k; ++m) distinct &= q->device[m] != q->device[k];
  }
  q->ev.resize(n, nullptr);
  for (int k = 0; k < n; ++k) {
    if (hipSetDevice(q->device[k]) != hipSuccess || hipEventCreateWithFlags(&q->ev[k], hipEventDisableTiming) != hipSuccess) {
      lto_group_comm_destroy(q);
      return LTO_EHIP;
    }
  }
  if (distinct && n > 1 && rccl().ok) {
    q->comm.resize(n, nullptr);
    if (rccl().CommInitAll(q->comm.data(), n, q->device.data()) != ncclSuccess) { q->comm.clear(); lto_group_comm_destroy(q); return LTO_EHIP; }
    q->clique = true;
  } else if (distinct && n > 1) {
    lto_group_comm_destroy(q);
    return LTO_EUNSUPPORTED;            // several GPUs but no RCCL in the process
  } else {
    bool same = true;
    for (int k = 1; k < n; ++k) same &= q->device[k] == q->device[0];
    if (!same) { lto_group_comm_destroy(q); return LTO_EUNSUPPORTED; }   // partly repeated device lists: neither a clique nor one device
    if (hipSetDevice(q->device[0]) != hipSuccess || hipMalloc((void**)&q->d_ptrs, sizeof(double*) * n) != hipSuccess) {
      lto_group_comm_destroy(q);
      return LTO_EHIP;
    }
  }
  *out = q;
  return LTO_OK;
}

void lto_group_comm_destroy(lto_group_comm* q) {
  if (!q) return;
  for (size_t k = 0; k < q->comm.size(); ++k)
    if (q->comm[k]) { (void)hipSetDevice(q->device[k]); (void)rccl().CommDestroy(q->comm[k]); }
  for (size_t k = 0; k < q->ev.size(); ++k)
    if (q->ev[k]) { (void)hipSetDevice(q->device[k]); (void)hipEventDestroy(q->ev[k]); }
  if (q->d_ptrs) { (void)hipSetDevice(q->device[0]); (void)hipFree(q->d_ptrs); }
  delete q;
}

const char* lto_group_comm_last_error(const lto_group_comm* q) { return q ? q->err : "null group communicator"; }
int lto_group_comm_uses_rccl(const lto_group_comm* q) { return q && q->clique ? 1 : 0; }

/* recv[k] [n][count] <- send[m] [count] of every member m, for every member k; member k's operands live on its device and
 * the work is enqueued on its context's stream (lto_ctx_stream), after whatever produced send[k] there. */
int lto_group_comm_allgather_dev(lto_group_comm* q, const double* const* send, double* const* recv, long count) {
  if (!q || !send || !recv) return LTO_ENULL;
  if (count < 0) return comm_fail(q, LTO_EINVAL, "count < 0");
  const int n = (int)q->ctx.size();
  if (count == 0) return LTO_OK;
  for (int k = 0; k < n; ++k) if (!send[k] || !recv[k]) return comm_fail(q, LTO_ENULL, "send[k] or recv[k] is NULL");
  if (q->clique) {
    ncclResult_t r = rccl().GroupStart();
    for (int k = 0; k < n && r == ncclSuccess; ++k) {
      if (hipSetDevice(q->device[k]) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipSetDevice");
      r = rccl().AllGather(send[k], recv[k], (size_t)count, ncclDouble, q->comm[k], (hipStream_t)lto_ctx_stream(q->ctx[k]));
    }
    const ncclResult_t e = rccl().GroupEnd();
    if (r != ncclSuccess) return comm_fail(q, LTO_EHIP, "ncclAllGather", r);
    if (e != ncclSuccess) return comm_fail(q, LTO_EHIP, "ncclGroupEnd", e);
    return LTO_OK;
  }
  // one device, several contexts: every stream publishes an event after its producer, every stream waits for all of them
  // and copies the slabs into its own receive buffer
  if (hipSetDevice(q->device[0]) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipSetDevice");
  for (int k = 0; k < n; ++k)
    if (hipEventRecord(q->ev[k], (hipStream_t)lto_ctx_stream(q->ctx[k])) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipEventRecord");
  for (int k = 0; k < n; ++k) {
    hipStream_t st = (hipStream_t)lto_ctx_stream(q->ctx[k]);
    for (int m = 0; m < n; ++m) {
      if (m != k && hipStreamWaitEvent(st, q->ev[m], 0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipStreamWaitEvent");
      if (hipMemcpyAsync(recv[k] + (long)m * count, send[m], sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return comm_fail(q, LTO_EHIP, "hipMemcpyAsync");
    }
  }
  return LTO_OK;
}

/* buf[k] [count] <- sum / max over members, in place on every member. */
int lto_group_comm_allreduce_dev(lto_group_comm* q, double* const* buf, long count, int op) {
  if (!q || !buf) return LTO_ENULL;
  if (count < 0 || (op != LTO_COMM_SUM && op != LTO_COMM_MAX)) return comm_fail(q, LTO_EINVAL, "bad count or op");
  const int n = (int)q->ctx.size();
  if (count == 0) return LTO_OK;
  for (int k = 0; k < n; ++k) if (!buf[k]) return comm_fail(q, LTO_ENULL, "buf[k] is NULL");
  if (q->clique) {
    ncclResult_t r = rccl().GroupStart();
    for (int k = 0; k < n && r == ncclSuccess; ++k) {
      if (hipSetDevice(q->device[k]) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipSetDevice");
      r = rccl().AllReduce(buf[k], buf[k], (size_t)count, ncclDouble, op == LTO_COMM_SUM ? ncclSum : ncclMax, q->comm[k],
                           (hipStream_t)lto_ctx_stream(q->ctx[k]));
    }
    const ncclResult_t e = rccl().GroupEnd();
    if (r != ncclSuccess) return comm_fail(q, LTO_EHIP, "ncclAllReduce", r);
    if (e != ncclSuccess) return comm_fail(q, LTO_EHIP, "ncclGroupEnd", e);
    return LTO_OK;
  }
  // one device: stream 0 waits for every producer, reduces all buffers in place, the others wait for stream 0
  if (hipSetDevice(q->device[0]) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipSetDevice");
  hipStream_t s0 = (hipStream_t)lto_ctx_stream(q->ctx[0]);
  for (int k = 1; k < n; ++k) {
    if (hipEventRecord(q->ev[k], (hipStream_t)lto_ctx_stream(q->ctx[k])) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipEventRecord");
    if (hipStreamWaitEvent(s0, q->ev[k], 0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipStreamWaitEvent");
  }
  if (hipMemcpyAsync(q->d_ptrs, buf, sizeof(double*) * (size_t)n, hipMemcpyHostToDevice, s0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipMemcpyAsync");
  if (hipStreamSynchronize(s0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipStreamSynchronize");   // `buf` is the caller's host array
  hipLaunchKernelGGL(k_reduce_buffers, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s0, (double* const*)q->d_ptrs, n, count, op);
  if (hipGetLastError() != hipSuccess) return comm_fail(q, LTO_EHIP, "k_reduce_buffers");
  if (hipEventRecord(q->ev[0], s0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipEventRecord");
  for (int k = 1; k < n; ++k)
    if (hipStreamWaitEvent((hipStream_t)lto_ctx_stream(q->ctx[k]), q->ev[0], 0) != hipSuccess) return comm_fail(q, LTO_EHIP, "hipStreamWaitEvent");
  return LTO_OK;
}

}  // extern "C"
