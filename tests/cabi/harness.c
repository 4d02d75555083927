/* Plain-C caller of include/lto.h, the way a Julia `ccall` (or any C host) uses the boundary: no Python, no torch.
 *
 *   harness                       no arguments: version, error codes of the misuse paths; exits 0 when the library behaves
 *                                 (with or without a device: without one lto_create must fail with LTO_ENODEVICE).
 *   harness in.bin out.bin        reads  int32 ndim, n_nodes, method, steps; double rtol, atol; 8 doubles lto_params;
 *                                        XC [ndim x n_nodes]; t [n_nodes]
 *                                 writes defect [ndim x S], errors [S] (lto_indirect_defect, pageable buffers),
 *                                        Phi [ndim x ndim x S], defect [ndim x S] (lto_indirect_jacobian, page-locked buffers
 *                                        from lto_host_alloc)
 * Built by tests/test_cabi_c_harness.py with  gcc -std=c99 -pedantic -Wall -Werror  (the header must be valid C99). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lto.h"

static int fail(const char* what, int rc, const lto_ctx* c) {
  fprintf(stderr, "harness: %s -> %d (%s)\n", what, rc, c ? lto_last_error(c) : "");
  return 1;
}

static int self_check(void) {
  lto_ctx* c = NULL;
  int rc;
  if (lto_version() != LTO_VERSION) return fail("lto_version", lto_version(), NULL);
  if (lto_create(NULL, 0) != LTO_ENULL) return fail("lto_create(NULL)", 0, NULL);
  /* host logic that needs no device (round 6): what LTO_KERNEL_AUTO resolves to, by the MI355X cost table */
  if (lto_indirect_auto_kernel(14, LTO_RK4, 64, 1.0, 4096, 256, 0) != LTO_KERNEL_PIPE8) return fail("auto kernel, contract size", 0, NULL);
  if (lto_indirect_auto_kernel(14, LTO_RK4, 64, 1.0, 262144, 256, 0) != LTO_KERNEL_PIPE32) return fail("auto kernel, 14-dim C4 size", 0, NULL);
  if (lto_indirect_auto_kernel(12, LTO_RK4, 64, 1.0, 262144, 256, 0) != LTO_KERNEL_LANE) return fail("auto kernel, C4", 0, NULL);
  if (lto_indirect_auto_kernel(12, LTO_RK4, 64, 1.0, 32768, 256, 0) != LTO_KERNEL_PIPE48) return fail("auto kernel, C4 per rank of eight", 0, NULL);
  if (lto_indirect_auto_kernel(12, LTO_DOP853_ADAPTIVE, 0, 1.0, 65536, 256, 1) != LTO_KERNEL_COOP2) return fail("auto kernel, C5", 0, NULL);
  if (lto_indirect_auto_kernel(14, LTO_DOP853_ADAPTIVE, 0, 2.0, 4096, 256, 0) != LTO_KERNEL_COOP) return fail("auto kernel, 14-dim p = 2", 0, NULL);
  if (lto_indirect_auto_kernel(13, LTO_RK4, 64, 1.0, 4096, 256, 0) != LTO_EINVAL) return fail("auto kernel accepted ndim 13", 0, NULL);
  if (lto_indirect_auto_kernel(12, LTO_RK4, 64, 0.5, 4096, 256, 0) != LTO_EINVAL) return fail("auto kernel accepted p = 0.5", 0, NULL);
  if (lto_comm_rccl_ranks(NULL) != LTO_ENULL || lto_last_call_order(NULL) != LTO_ENULL) return fail("NULL handles accepted", 0, NULL);
  printf("host logic: LTO_KERNEL_AUTO resolves as documented\n");
  rc = lto_create(&c, 0);
  if (rc == LTO_ENODEVICE) { printf("no device: lto_create refused (LTO_ENODEVICE), no CPU fallback\n"); return 0; }
  if (rc != LTO_OK) return fail("lto_create", rc, c);
  {
    lto_integrator integ = {LTO_RK4, 4, 0.0, 0.0, 0};
    lto_params prm = {0.0121505856, 384400.0, 375190.2589, 0.05, 1000.0, 1.0, 1.0, 1.0};
    double XC[24] = {0}, t[2] = {0.0, 0.1}, d[12];
    if (lto_indirect_defect(c, 12, 2, 1, NULL, t, 1, &prm, 1, &integ, d, NULL) != LTO_ENULL) return fail("NULL XC accepted", 0, c);
    if (lto_indirect_defect(c, 13, 2, 1, XC, t, 1, &prm, 1, &integ, d, NULL) >= 0) return fail("ndim 13 accepted", 0, c);
    prm.p = 0.5;   /* the reference: error("Invalid value of p!") */
    if (lto_indirect_defect(c, 12, 2, 1, XC, t, 1, &prm, 1, &integ, d, NULL) != LTO_EBADP) return fail("p = 0.5 accepted", 0, c);
  }
  lto_destroy(c);
  printf("device present: misuse paths return their codes\n");
  return 0;
}

int main(int argc, char** argv) {
  int hdr[4], rc, ndim, n, S;
  double tol[2];
  lto_params prm;
  lto_integrator integ;
  lto_ctx* c = NULL;
  double *XC, *t, *defect, *errors, *pXC, *pt, *pPhi, *pdef;
  void* blk;
  FILE* f;
  if (argc < 3) return self_check();
  f = fopen(argv[1], "rb");
  if (!f) return fail("open input", 0, NULL);
  if (fread(hdr, sizeof(int), 4, f) != 4 || fread(tol, sizeof(double), 2, f) != 2 || fread(&prm, sizeof prm, 1, f) != 1)
    return fail("read header", 0, NULL);
  ndim = hdr[0]; n = hdr[1]; S = n - 1;
  integ.method = hdr[2]; integ.steps = hdr[3]; integ.rtol = tol[0]; integ.atol = tol[1]; integ.max_steps = 0;
  XC = (double*)malloc(sizeof(double) * (size_t)ndim * n);
  t = (double*)malloc(sizeof(double) * (size_t)n);
  defect = (double*)malloc(sizeof(double) * (size_t)ndim * S);
  errors = (double*)malloc(sizeof(double) * (size_t)S);
  if (!XC || !t || !defect || !errors) return fail("malloc", 0, NULL);
  if (fread(XC, sizeof(double), (size_t)ndim * n, f) != (size_t)ndim * n || fread(t, sizeof(double), (size_t)n, f) != (size_t)n)
    return fail("read arrays", 0, NULL);
  fclose(f);

  rc = lto_create(&c, 0);
  if (rc) return fail("lto_create", rc, c);
  rc = lto_indirect_defect(c, ndim, n, 1, XC, t, 1, &prm, 1, &integ, defect, errors);
  if (rc) return fail("lto_indirect_defect", rc, c);
  /* page-locked operands, carved out of one block */
  rc = lto_host_alloc(c, sizeof(double) * ((size_t)ndim * n + n + (size_t)ndim * ndim * S + (size_t)ndim * S), &blk);
  if (rc) return fail("lto_host_alloc", rc, c);
  pXC = (double*)blk; pt = pXC + (size_t)ndim * n; pPhi = pt + n; pdef = pPhi + (size_t)ndim * ndim * S;
  memcpy(pXC, XC, sizeof(double) * (size_t)ndim * n);
  memcpy(pt, t, sizeof(double) * (size_t)n);
  rc = lto_indirect_jacobian(c, ndim, n, 1, pXC, pt, 1, &prm, 1, &integ, pPhi, pdef);
  if (rc) return fail("lto_indirect_jacobian", rc, c);

  f = fopen(argv[2], "wb");
  if (!f) return fail("open output", 0, NULL);
  fwrite(defect, sizeof(double), (size_t)ndim * S, f);
  fwrite(errors, sizeof(double), (size_t)S, f);
  fwrite(pPhi, sizeof(double), (size_t)ndim * ndim * S, f);
  fwrite(pdef, sizeof(double), (size_t)ndim * S, f);
  fclose(f);
  rc = lto_host_free(c, blk);
  if (rc) return fail("lto_host_free", rc, c);
  lto_destroy(c);
  free(XC); free(t); free(defect); free(errors);
  return 0;
}
