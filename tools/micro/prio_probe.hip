// Development micro-probe, round 3: what a LONE wavefront on a SIMD pays per instruction of the kinds the base role of the
// pipeline kernel is made of, and what a second, low-priority wavefront on the same SIMD gets / costs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/prio_probe.hip -o build/prio_probe && build/prio_probe
// Workgroup = 8 wavefronts (wave i and i + 4 share a SIMD: sync_probe.hip); wave 2 is the measured one, wave 6 the helper.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// KIND: 0 fma x4 chains; 1 fma + v_mov_b32_dpp quad_perm pairs (1 pair per 4 fma); 2 fma + v_cndmask pairs; 3 fma + ds_write_b64 (1 per 8 fma);
// 4 fma + ds_write2_b64 (1 per 8 fma); 5 fma + v_rsq_f64 (1 per 8 fma); 6 fma + v_rcp_f64 (1 per 8); 7 fma + s_mov (1 per 4 fma)
// 8 fma + v_mov_b64 (1 per 4 fma); 9 fma + v_mov_b32_dpp row_newbcast... (1 pair per 4)
template <int KIND> __device__ __forceinline__ void body(double (&a)[4], const double c, const double d, double* lds, int& x, int& y) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    if (KIND == 1) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
                     asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(x)); }
    if (KIND == 2) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y) : "v"(x)); }
    if (KIND == 3 && (r & 1)) asm volatile("ds_write_b64 %0, %1" :: "v"(x), "v"(a[0]) : "memory");
    if (KIND == 4 && (r & 1)) asm volatile("ds_write2_b64 %0, %1, %2 offset1:16" :: "v"(x), "v"(a[0]), "v"(a[1]) : "memory");
    if (KIND == 5 && (r & 1)) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[3]));
    if (KIND == 6 && (r & 1)) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[3]));
    if (KIND == 7) asm volatile("s_mov_b32 s40, 5" ::: "s40");
    if (KIND == 8) asm volatile("v_mov_b64 %0, %1" : "=v"(a[3]) : "v"(a[2]));
    if (KIND == 10 && (r & 1)) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[3]));
  }
}

template <int KIND> __global__ __launch_bounds__(512) void k_prio(double* out, long long* cyc, int iters, int helper, int prio_base) {
  __shared__ double lds[4096];
  const int wave = threadIdx.x >> 6;
  if (wave != 2 && !(wave == 6 && helper)) return;
  double a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
  const double c = 1.0000001, d = 1e-9;
  int x = (threadIdx.x & 63) * 8, y = 3;
  long long t0, t1;
  if (wave == 2) {
    if (prio_base) __builtin_amdgcn_s_setprio(3);
    t0 = clock64();
    for (int it = 0; it < iters; ++it) body<KIND>(a, c, d, lds, x, y);
    t1 = clock64();
  } else {
    // helper: plain FMA stream, low priority, runs longer than the base wave (helper = multiple of iters)
    t0 = clock64();
    for (int it = 0; it < iters * 2; ++it) body<0>(a, c, d, lds, x, y);
    t1 = clock64();
  }
  double s = a[0] + a[1] + a[2] + a[3] + x + y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 2 + (wave == 6)] = t1 - t0;
}

template <int KIND> static int run(const char* name, int extra_per_iter, double* d, long long* dc) {
  const int iters = 300;
  for (int mode = 0; mode < 3; ++mode) {        // 0 lone; 1 helper, equal priority; 2 helper, base at priority 3
    CK(hipMemset(dc, 0, 8 * 2 * 256));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_prio<KIND>), dim3(256), dim3(512), 0, 0, d, dc, iters, mode > 0, mode == 2);
    std::vector<long long> c(512);
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    double mb = 0, mh = 0; for (int i = 0; i < 256; ++i) { mb += c[2 * i]; mh += c[2 * i + 1]; } mb /= 256; mh /= 256;
    const double n_base = iters * (64.0 + extra_per_iter);
    printf("  %-34s %-28s base %.2f ticks/instr (%.0f instr)", name, mode == 0 ? "lone" : mode == 1 ? "helper, equal priority" : "helper, base at priority 3", mb / n_base, n_base);
    if (mode) printf("   helper %.2f ticks/instr over its own run", mh / (iters * 2 * 64.0));
    printf("\n");
  }
  return 0;
}

int main() {
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 512 * 256)); CK(hipMalloc(&dc, 8 * 2 * 256));
  run<0>("fma x4 chains", 0, d, dc);
  run<1>("+ 2 v_mov_b32_dpp quad_perm / 4 fma", 32, d, dc);
  run<2>("+ 2 v_cndmask_b32 / 4 fma", 32, d, dc);
  run<3>("+ ds_write_b64 / 8 fma", 8, d, dc);
  run<4>("+ ds_write2_b64 / 8 fma", 8, d, dc);
  run<5>("+ v_rsq_f64 / 8 fma", 8, d, dc);
  run<6>("+ v_rcp_f64 / 8 fma", 8, d, dc);
  run<10>("+ v_sqrt_f64 / 8 fma", 8, d, dc);
  run<7>("+ s_mov_b32 / 4 fma", 16, d, dc);
  run<8>("+ v_mov_b64 / 4 fma", 16, d, dc);
  return 0;
}
