// lto_group.hip -- several GPUs behind ONE host process (SURVEY 8b "multi-GPU handled inside one call via one host
// thread per device"): the single-process Julia host of INTEGRATION.md has no torch.distributed to shard with.
//
// A group owns one lto_ctx per entry of device_ids.  Every call splits the sweep into contiguous shards -- whole
// trajectories when the batch is at least as large as the group, otherwise segment blocks of the single trajectory
// with a one-node halo (segment i reads nodes i and i+1 only: multiShoot_CRTBP_indirect.jl:71-86,
// multiShoot_CRTBP_direct.jl:77-105) -- and runs the blocking single-device entry point of each shard on its own
// host thread and context.  Julia's column-major outputs are segment-major, so every shard writes one contiguous
// slab of the caller's arrays: the "gather" is the D2H copy itself, no collective and no staging.
// Device ids may repeat (two shards on one GPU); that is how the tests exercise the sharding on a 1-GPU box.
#include <cstdio>
#include <cstring>
#include <new>
#include <pthread.h>

#include "../../include/lto.h"
#include "hostbuf.hpp"

struct lto_group {
  lto::HostList<lto_ctx*> ctx;
  char err[512];
};

namespace {

struct Shard { long first; long count; };   // units: trajectories or segments

// contiguous near-equal partition of `total` units over at most `parts` shards (no empty shard); sh.ok() is false when out of memory
struct Shards {
  lto::HostBuf<Shard> sh;
  size_t count = 0;
  bool ok() const { return sh.ok(); }
  size_t size() const { return count; }
  const Shard& operator[](size_t k) const { return sh[k]; }
};
void partition(long total, int parts, Shards& out) {
  const long g = parts < total ? parts : total;
  if (!out.sh.alloc((size_t)g)) return;
  long first = 0;
  for (long r = 0; r < g; ++r) {
    const long cnt = total / g + (r < total % g ? 1 : 0);
    out.sh[r] = {first, cnt};
    first += cnt;
  }
  out.count = (size_t)g;
}

int fail(lto_group* g, int code, const char* msg) {
  std::snprintf(g->err, sizeof g->err, "%s", msg);
  return code;
}

// run fn(shard index) on one thread per shard; first non-zero return code wins, its context's message is kept.  Threads are
// pthreads, not std::thread: a thread that cannot be created is a return value here (LTO_ENOMEM after the started ones have been
// joined), where std::thread's constructor would throw -- in a library built without exception support that ends the process.
template <class F>
struct ShardJob { F* fn; size_t k; int rc; };
template <class F>
void* shard_main(void* arg) {
  ShardJob<F>* j = (ShardJob<F>*)arg;
  j->rc = (*j->fn)(j->k);
  return nullptr;
}
template <class F>
int run_shards(lto_group* g, const Shards& sh, F fn) {
  if (!sh.ok()) return fail(g, LTO_ENOMEM, "out of host memory");
  const size_t n = sh.size();
  lto::HostBuf<ShardJob<F>> job(n);
  lto::HostBuf<pthread_t> th(n);
  if (!job.ok() || !th.ok()) return fail(g, LTO_ENOMEM, "out of host memory");
  size_t started = 1;
  int spawn_err = 0;
  for (size_t k = 1; k < n; ++k) {
    job[k].fn = &fn; job[k].k = k; job[k].rc = 0;
    spawn_err = pthread_create(&th[k], nullptr, shard_main<F>, &job[k]);
    if (spawn_err) break;
    started = k + 1;
  }
  job[0].rc = spawn_err ? 0 : fn(0);               // nothing is computed for a call that cannot run all of its shards
  for (size_t k = 1; k < started; ++k) (void)pthread_join(th[k], nullptr);
  if (spawn_err) {
    std::snprintf(g->err, sizeof g->err, "could not start the host thread of shard %zu (pthread_create: error %d)", started, spawn_err);
    return LTO_ENOMEM;
  }
  for (size_t k = 0; k < n; ++k)
    if (job[k].rc) {
      std::snprintf(g->err, sizeof g->err, "shard %zu (device context %zu): %s", k, k, lto_last_error(g->ctx[k]));
      return job[k].rc;
    }
  g->err[0] = 0;
  return LTO_OK;
}

bool by_batch(const lto_group* g, int n_batch) { return n_batch > 1 || g->ctx.size() == 1; }

}  // namespace

extern "C" {

int lto_group_create(int n_devices, const int* device_ids, lto_group** out) {
  if (!out) return LTO_ENULL;
  *out = nullptr;
  if (n_devices < 1 || !device_ids) return LTO_EINVAL;
  lto_group* g = new (std::nothrow) lto_group();
  if (!g) return LTO_ENOMEM;
  g->err[0] = 0;
  for (int k = 0; k < n_devices; ++k) {
    lto_ctx* c = nullptr;
    const int rc = lto_create(&c, device_ids[k]);
    if (rc) {
      for (lto_ctx* q : g->ctx) lto_destroy(q);
      delete g;
      return rc;
    }
    if (!g->ctx.push(c)) {
      lto_destroy(c);
      for (lto_ctx* q : g->ctx) lto_destroy(q);
      delete g;
      return LTO_ENOMEM;
    }
  }
  *out = g;
  return LTO_OK;
}

void lto_group_destroy(lto_group* g) {
  if (!g) return;
  for (lto_ctx* c : g->ctx) lto_destroy(c);
  delete g;
}

const char* lto_group_last_error(const lto_group* g) { return g ? g->err : "null group"; }

int lto_group_size(const lto_group* g) { return g ? (int)g->ctx.size() : 0; }

lto_ctx* lto_group_ctx(lto_group* g, int k) { return (g && k >= 0 && k < (int)g->ctx.size()) ? g->ctx[k] : nullptr; }

int lto_group_indirect_defect(lto_group* g, int ndim, int n_nodes, int n_batch, const double* XC, const double* t, int n_tgrids,
                              const lto_params* prm, int n_prm, const lto_integrator* integ, double* defect, double* errors) {
  if (!g) return LTO_ENULL;
  if (!XC || !t || !prm || !integ || !defect) return fail(g, LTO_ENULL, "XC, t, prm, integ or defect is NULL");
  if (n_nodes < 2 || n_batch < 1 || (ndim != 12 && ndim != 14)) return fail(g, LTO_EINVAL, "bad ndim / n_nodes / n_batch");
  if ((n_tgrids != 1 && n_tgrids != n_batch) || (n_prm != 1 && n_prm != n_batch)) return fail(g, LTO_EINVAL, "n_tgrids / n_prm must be 1 or n_batch");
  const long S = n_nodes - 1;
  if (by_batch(g, n_batch)) {
    Shards sh;
    partition(n_batch, (int)g->ctx.size(), sh);
    return run_shards(g, sh, [&](size_t k) {
      const long b = sh[k].first, nb = sh[k].count;
      return lto_indirect_defect(g->ctx[k], ndim, n_nodes, (int)nb, XC + (long)ndim * n_nodes * b, t + (n_tgrids == 1 ? 0 : n_nodes * b),
                                 n_tgrids == 1 ? 1 : (int)nb, prm + (n_prm == 1 ? 0 : b), n_prm == 1 ? 1 : (int)nb, integ,
                                 defect + (long)ndim * S * b, errors ? errors + S * b : nullptr);
    });
  }
  Shards sh;
  partition(S, (int)g->ctx.size(), sh);
  return run_shards(g, sh, [&](size_t k) {
    const long s0 = sh[k].first, cnt = sh[k].count;
    return lto_indirect_defect(g->ctx[k], ndim, (int)cnt + 1, 1, XC + (long)ndim * s0, t + s0, 1, prm, 1, integ, defect + (long)ndim * s0,
                               errors ? errors + s0 : nullptr);
  });
}

int lto_group_indirect_jacobian(lto_group* g, int ndim, int n_nodes, int n_batch, const double* XC, const double* t, int n_tgrids,
                                const lto_params* prm, int n_prm, const lto_integrator* integ, double* Phi, double* defect) {
  if (!g) return LTO_ENULL;
  if (!XC || !t || !prm || !integ || !Phi) return fail(g, LTO_ENULL, "XC, t, prm, integ or Phi is NULL");
  if (n_nodes < 2 || n_batch < 1 || (ndim != 12 && ndim != 14)) return fail(g, LTO_EINVAL, "bad ndim / n_nodes / n_batch");
  if ((n_tgrids != 1 && n_tgrids != n_batch) || (n_prm != 1 && n_prm != n_batch)) return fail(g, LTO_EINVAL, "n_tgrids / n_prm must be 1 or n_batch");
  const long S = n_nodes - 1, nn = (long)ndim * ndim;
  if (by_batch(g, n_batch)) {
    Shards sh;
    partition(n_batch, (int)g->ctx.size(), sh);
    return run_shards(g, sh, [&](size_t k) {
      const long b = sh[k].first, nb = sh[k].count;
      return lto_indirect_jacobian(g->ctx[k], ndim, n_nodes, (int)nb, XC + (long)ndim * n_nodes * b, t + (n_tgrids == 1 ? 0 : n_nodes * b),
                                   n_tgrids == 1 ? 1 : (int)nb, prm + (n_prm == 1 ? 0 : b), n_prm == 1 ? 1 : (int)nb, integ,
                                   Phi + nn * S * b, defect ? defect + (long)ndim * S * b : nullptr);
    });
  }
  Shards sh;
  partition(S, (int)g->ctx.size(), sh);
  return run_shards(g, sh, [&](size_t k) {
    const long s0 = sh[k].first, cnt = sh[k].count;
    return lto_indirect_jacobian(g->ctx[k], ndim, (int)cnt + 1, 1, XC + (long)ndim * s0, t + s0, 1, prm, 1, integ, Phi + nn * s0,
                                 defect ? defect + (long)ndim * s0 : nullptr);
  });
}

int lto_group_direct_defect(lto_group* g, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                            int n_tgrids, int nsteps, const lto_direct_params* prm, double* defect, double* errors) {
  if (!g) return LTO_ENULL;
  if (!X || !U || !t || !prm || !defect) return fail(g, LTO_ENULL, "X, U, t, prm or defect is NULL");
  if (n_nodes < 2 || n_batch < 1 || (nstate != 6 && nstate != 7)) return fail(g, LTO_EINVAL, "bad nstate / n_nodes / n_batch");
  if (n_tgrids != 1 && n_tgrids != n_batch) return fail(g, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  const long S = n_nodes - 1;
  if (by_batch(g, n_batch)) {
    Shards sh;
    partition(n_batch, (int)g->ctx.size(), sh);
    return run_shards(g, sh, [&](size_t k) {
      const long b = sh[k].first, nb = sh[k].count;
      return lto_direct_defect(g->ctx[k], nstate, n_nodes, (int)nb, X + (long)nstate * n_nodes * b, U + 3L * n_nodes * b,
                               t + (n_tgrids == 1 ? 0 : n_nodes * b), n_tgrids == 1 ? 1 : (int)nb, nsteps, prm,
                               defect + (long)nstate * S * b, errors ? errors + S * b : nullptr);
    });
  }
  Shards sh;
  partition(S, (int)g->ctx.size(), sh);
  return run_shards(g, sh, [&](size_t k) {
    const long s0 = sh[k].first, cnt = sh[k].count;
    return lto_direct_defect(g->ctx[k], nstate, (int)cnt + 1, 1, X + (long)nstate * s0, U + 3 * s0, t + s0, 1, nsteps, prm,
                             defect + (long)nstate * s0, errors ? errors + s0 : nullptr);
  });
}

int lto_group_direct_jacobian(lto_group* g, int nstate, int n_nodes, int n_batch, const double* X, const double* U, const double* t,
                              int n_tgrids, int nsteps, const lto_direct_params* prm, double* Jac_temp, double* ddefect_dtf,
                              double* defect, double* errors) {
  if (!g) return LTO_ENULL;
  if (!X || !U || !t || !prm || !Jac_temp) return fail(g, LTO_ENULL, "X, U, t, prm or Jac_temp is NULL");
  if (n_nodes < 2 || n_batch < 1 || (nstate != 6 && nstate != 7)) return fail(g, LTO_EINVAL, "bad nstate / n_nodes / n_batch");
  if (n_tgrids != 1 && n_tgrids != n_batch) return fail(g, LTO_EINVAL, "n_tgrids must be 1 or n_batch");
  const long S = n_nodes - 1, nj = (long)nstate * 2 * (nstate + 3);
  if (by_batch(g, n_batch)) {
    Shards sh;
    partition(n_batch, (int)g->ctx.size(), sh);
    return run_shards(g, sh, [&](size_t k) {
      const long b = sh[k].first, nb = sh[k].count;
      return lto_direct_jacobian(g->ctx[k], nstate, n_nodes, (int)nb, X + (long)nstate * n_nodes * b, U + 3L * n_nodes * b,
                                 t + (n_tgrids == 1 ? 0 : n_nodes * b), n_tgrids == 1 ? 1 : (int)nb, nsteps, prm, Jac_temp + nj * S * b,
                                 ddefect_dtf ? ddefect_dtf + (long)nstate * S * b : nullptr,
                                 defect ? defect + (long)nstate * S * b : nullptr, errors ? errors + S * b : nullptr);
    });
  }
  Shards sh;
  partition(S, (int)g->ctx.size(), sh);
  const double span_total = t[n_nodes - 1] - t[0];
  return run_shards(g, sh, [&](size_t k) {
    const long s0 = sh[k].first, cnt = sh[k].count;
    double* dtf = ddefect_dtf ? ddefect_dtf + (long)nstate * s0 : nullptr;
    const int rc = lto_direct_jacobian(g->ctx[k], nstate, (int)cnt + 1, 1, X + (long)nstate * s0, U + 3 * s0, t + s0, 1, nsteps, prm,
                                       Jac_temp + nj * s0, dtf, defect ? defect + (long)nstate * s0 : nullptr,
                                       errors ? errors + s0 : nullptr);
    if (rc == LTO_OK && dtf) {
      // the tf partial scales every segment by h_i / (tf - t0) (direct.jl:506-510); a shard saw its own span
      const double scale = (t[s0 + cnt] - t[s0]) / span_total;
      for (long q = 0; q < (long)nstate * cnt; ++q) dtf[q] *= scale;
    }
    return rc;
  });
}

}  // extern "C"
