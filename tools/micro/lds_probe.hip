// Development micro-probe, round 4: what LDS traffic and LDS-flag hand-overs cost a wavefront's instruction stream, for the
// flag-paced cooperative kernel (kernels_indirect_coop3.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/lds_probe.hip -o build/lds_probe && build/lds_probe
// Part 1: issue cost per LDS instruction mixed into an FMA stream (one LDS instruction per 8 FMAs), lone wave and with a second
//         wave on the same SIMD (wave i and i + 4 share a SIMD: sync_probe.hip).
// Part 2: round trip of a hand-over between two wavefronts of a workgroup: (a) s_barrier, (b) data + flag in LDS, consumer polls,
//         on the same SIMD and on different SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// KIND: 0 fma only; 1 ds_write_b64; 2 ds_write2_b64; 3 ds_write_b128; 4 ds_read_b64 (waited at the end of the iteration);
// 5 ds_read2_b64; 6 ds_read_b128; 7 ds_write_b64 x2 addressed [seg][slot] (bank-conflicting stride 32 doubles); 8 ds_write_b128 stride 34
template <int KIND> __device__ __forceinline__ void body(double (&a)[4], const double c, const double d, const int addr, double (&r)[8]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    }
    if (KIND == 1) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(a[0]) : "memory");
    if (KIND == 2) asm volatile("ds_write2_b64 %0, %1, %2 offset1:64" :: "v"(addr), "v"(a[0]), "v"(a[1]) : "memory");
    if (KIND == 3 || KIND == 8) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 v; v.x = a[0]; v.y = a[1];
      asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
    }
    if (KIND == 4) asm volatile("ds_read_b64 %0, %1" : "=v"(r[q]) : "v"(addr) : "memory");
    if (KIND == 5) { typedef double d2 __attribute__((ext_vector_type(2))); d2 v; asm volatile("ds_read2_b64 %0, %1 offset1:64" : "=v"(v) : "v"(addr) : "memory"); r[q] = v.x + 0 * v.y; }
    if (KIND == 6) { typedef double d2 __attribute__((ext_vector_type(2))); d2 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); r[q] = v.x + 0 * v.y; }
    if (KIND == 7) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(a[0]) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KIND> __global__ __launch_bounds__(512) void k_lds(double* out, long long* cyc, int iters, int helper) {
  __shared__ __attribute__((aligned(16))) double lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave != 2 && !(wave == 6 && helper)) return;
  double a[4], r[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = 1.0 + 0.001 * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = 0.0;
  const double c = 1.0000001, d = 1e-9;
  int addr = (KIND == 3 || KIND == 6) ? lane * 16 : lane * 8;
  if (KIND == 7) addr = (lane >> 2) * 32 * 8 + (lane & 3) * 8;
  if (KIND == 8) addr = (lane >> 2) * 34 * 8 + (lane & 3) * 16;
  addr += (wave == 6) ? 32768 : 0;
  long long t0 = clock64();
  if (wave == 2) { for (int it = 0; it < iters; ++it) body<KIND>(a, c, d, addr, r); }
  else { for (int it = 0; it < iters * 2; ++it) body<0>(a, c, d, addr, r); }
  long long t1 = clock64();
  double s = a[0] + a[1] + a[2] + a[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (lane == 0) cyc[blockIdx.x * 2 + (wave == 6)] = t1 - t0;
}

template <int KIND> static int run(const char* name, double* d, long long* dc) {
  const int iters = 300;
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(dc, 0, 8 * 2 * 256));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_lds<KIND>), dim3(256), dim3(512), 0, 0, d, dc, iters, mode);
    std::vector<long long> c(512);
    CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
    double mb = 0, mh = 0; for (int i = 0; i < 256; ++i) { mb += c[2 * i]; mh += c[2 * i + 1]; } mb /= 256; mh /= 256;
    printf("  %-40s %-8s %8.1f ticks per iteration of 64 fma + 8 LDS instr", name, mode == 0 ? "lone" : "helper", mb / iters);
    if (mode) printf("   helper %.2f ticks/fma", mh / (iters * 2 * 64.0));
    printf("\n");
  }
  return 0;
}

// ---- hand-over round trips.  Wave A (wave wa) and wave B (wave wb) ping-pong `rounds` times; each leg hands over NV doubles per
// lane.  MODE 0: __syncthreads of two waves (the other six left); MODE 1: data + flag in LDS, the consumer polls the flag, then reads.
// MODE 2: as 1 but the consumer issues flag read and data reads together and re-reads if the flag was stale (optimistic).
template <int MODE, int NV> __global__ __launch_bounds__(512) void k_hand(double* out, long long* cyc, int rounds, int wa, int wb) {
  __shared__ double buf[2][NV][64];
  __shared__ int flag[2];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x < 2) flag[threadIdx.x] = 0;
  __syncthreads();
  if (wave != wa && wave != wb) return;
  const int me = (wave == wb);
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = lane + i;
  long long t0 = clock64();
  if (MODE == 0) {
    for (int r = 0; r < rounds; ++r) {
      if ((r & 1) == me) {
#pragma unroll
        for (int i = 0; i < NV; ++i) buf[0][i][lane] = v[i];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
      if ((r & 1) != me) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = buf[0][i][lane] + 1.0;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int r = 0; r < rounds; ++r) {
      if ((r & 1) == me) {      // producer of this round
#pragma unroll
        for (int i = 0; i < NV; ++i) buf[me][i][lane] = v[i];
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&flag[me], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      } else {
        const int o = 1 - me;
        if (MODE == 1) {
          int spins = 0;
          while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&flag[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < r + 1 && ++spins < (1 << 16)) {}
          if (spins >= (1 << 16)) break;      // never hang: a lost hand-over ends the measurement
          asm volatile("" ::: "memory");
#pragma unroll
          for (int i = 0; i < NV; ++i) v[i] = buf[o][i][lane] + 1.0;
        } else {
          int spins = 0, f;
          do {
            f = __hip_atomic_load(&flag[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *(volatile double*)&buf[o][i][lane] + 1.0;
          } while (__builtin_amdgcn_readfirstlane(f) < r + 1 && ++spins < (1 << 16));
          if (spins >= (1 << 16)) break;
        }
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 2 + me] = t1 - t0;
}

template <int MODE, int NV> static int run_hand(const char* name, int wa, int wb, double* d, long long* dc) {
  const int rounds = 2000;
  CK(hipMemset(dc, 0, 8 * 2 * 256));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_hand<MODE, NV>), dim3(256), dim3(512), 0, 0, d, dc, rounds, wa, wb);
  std::vector<long long> c(512);
  CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  double m = 0; for (int i = 0; i < 256; ++i) m += c[2 * i]; m /= 256;
  printf("  %-44s waves %d,%d  NV=%d: %7.1f ticks per one-way hand-over\n", name, wa, wb, NV, m / rounds);
  return 0;
}

int main() {
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 512 * 256)); CK(hipMalloc(&dc, 8 * 2 * 256));
  printf("part 1: 64 fma + 8 LDS instructions per iteration (fma-only iteration first)\n");
  run<0>("fma only", d, dc);
  run<1>("ds_write_b64", d, dc);
  run<2>("ds_write2_b64", d, dc);
  run<3>("ds_write_b128", d, dc);
  run<7>("ds_write_b64 [seg][32] layout", d, dc);
  run<8>("ds_write_b128 [seg][34] layout", d, dc);
  run<4>("ds_read_b64", d, dc);
  run<5>("ds_read2_b64", d, dc);
  run<6>("ds_read_b128", d, dc);
  printf("part 2: hand-over between two wavefronts\n");
  run_hand<0, 4>("two barriers per hand-over", 2, 6, d, dc);
  run_hand<0, 4>("two barriers per hand-over", 2, 3, d, dc);
  run_hand<1, 4>("flag, poll then read", 2, 6, d, dc);
  run_hand<1, 4>("flag, poll then read", 2, 3, d, dc);
  run_hand<2, 4>("flag, optimistic read", 2, 6, d, dc);
  run_hand<2, 4>("flag, optimistic read", 2, 3, d, dc);
  run_hand<1, 16>("flag, poll then read", 2, 3, d, dc);
  run_hand<2, 16>("flag, optimistic read", 2, 3, d, dc);
  return 0;
}
