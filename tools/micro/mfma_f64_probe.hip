// Development micro-probe, round 5: the FP64 matrix pipe next to the FP64 vector pipe.
//   (1) operand layout of v_mfma_f64_4x4x4f64 (four independent 4x4x4 products per instruction), found by experiment:
//       A = indicator of lane la, B = indicator of lane lb  ->  which lanes of D are 1
//   (2) issue cadence in s_memtime ticks per instruction: MFMA alone (1, 2, 4 independent accumulators; 1 and 2 waves per SIMD),
//       MFMA with R independent v_fma_f64 of the SAME wave behind each one, and an MFMA-only wave beside a VALU-only wave on one SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_probe.hip -o build/mfma_f64_probe && build/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_layout(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}

// MODE 0: every wave: NACC independent MFMA accumulators + R fmas behind each MFMA.  MODE 1: even waves MFMA only, odd waves FMA only.
template <int NACC, int R, int MODE> __global__ void k_rate(double* out, long long* cyc, int iters) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  double f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + 0.001 * (threadIdx.x + i);
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x, c = 1.0000001, d = 1e-9;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || (wave & 4) == 0;     // waves 0-3 / 4-7 of a workgroup sit on SIMDs 0-3: pairs (w, w+4) share a SIMD
  const bool do_fma = MODE == 0 || (wave & 4) != 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (do_mfma) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
          if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[(i * R + q) & 7]) : "v"(c), "v"(d));
          }
        }
      }
      if (MODE == 1 && do_fma) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[q]) : "v"(c), "v"(d));
      }
    }
  }
  const long long t1 = clock64();
  double s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int NACC, int R, int MODE> static void run(double* d, long long* dc, int waves_per_simd, const char* what) {
  const int iters = 200, wpb = 4 * waves_per_simd;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<NACC, R, MODE>), dim3(256), dim3(64 * wpb), 0, 0, d, dc, iters);
  std::vector<long long> c(256 * wpb);
  if (hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
  if (MODE == 0) {
    double m = 0; for (auto v : c) m += v; m /= c.size();
    const double per_group = m / (iters * 16.0 * NACC);      // ticks of one wave per (1 MFMA + R FMA)
    printf("  %-58s %d wave(s)/SIMD: %.1f ticks per [MFMA + %d FMA] of one wave = %.1f per SIMD\n", what, waves_per_simd, per_group, R, per_group / waves_per_simd);
  } else {
    double mm = 0, mf = 0; int nm = 0, nf = 0;
    for (size_t k = 0; k < c.size(); ++k) { if (((k % wpb) & 4) == 0) { mm += c[k]; ++nm; } else { mf += c[k]; ++nf; } }
    printf("  %-58s MFMA wave: %.1f ticks per MFMA; FMA wave beside it: %.2f ticks per v_fma_f64\n", what, mm / nm / (iters * 16.0 * NACC), mf / nf / (iters * 16.0 * 8));
  }
}

int main() {
  double *dA, *dB, *dD;
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 512));
  // layout: lane la of A = 1, lane lb of B = 1
  int a_blk[64], a_i[64], a_k[64], b_blk[64], b_k[64], b_j[64];
  std::vector<std::vector<int>> hit(64 * 64);
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double A[64] = {0}, B[64] = {0}, D[64];
      A[la] = 1.0; B[lb] = 1.0;
      CK(hipMemcpy(dA, A, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, 512, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      CK(hipMemcpy(D, dD, 512, hipMemcpyDeviceToHost));
      for (int l = 0; l < 64; ++l) if (D[l] != 0.0) hit[la * 64 + lb].push_back(l);
    }
  (void)a_blk; (void)a_i; (void)a_k; (void)b_blk; (void)b_k; (void)b_j;
  printf("v_mfma_f64_4x4x4f64 layout: for A-lane la, the B-lanes lb it meets and the D-lanes that receive the product\n");
  for (int la = 0; la < 64; la += 1) {
    if (!(la < 20 || la % 16 == 0)) continue;
    printf("  la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (!hit[la * 64 + lb].empty()) { printf(" lb=%d->D{", lb); for (int l : hit[la * 64 + lb]) printf("%d ", l); printf("}"); }
    printf("\n");
  }
  double* d; long long* dc;
  CK(hipMalloc(&d, 8 * 64 * 8 * 256)); CK(hipMalloc(&dc, 8 * 8 * 256));
  printf("cadence (s_memtime ticks), every CU busy:\n");
  for (int w = 1; w <= 2; ++w) {
    run<1, 0, 0>(d, dc, w, "MFMA only, 1 accumulator (dependent chain)");
    run<2, 0, 0>(d, dc, w, "MFMA only, 2 accumulators");
    run<4, 0, 0>(d, dc, w, "MFMA only, 4 accumulators");
    run<4, 1, 0>(d, dc, w, "4 accumulators, 1 v_fma_f64 behind each MFMA");
    run<4, 2, 0>(d, dc, w, "4 accumulators, 2 v_fma_f64 behind each MFMA");
    run<4, 3, 0>(d, dc, w, "4 accumulators, 3 v_fma_f64 behind each MFMA");
    run<4, 4, 0>(d, dc, w, "4 accumulators, 4 v_fma_f64 behind each MFMA");
    run<4, 6, 0>(d, dc, w, "4 accumulators, 6 v_fma_f64 behind each MFMA");
  }
  run<4, 0, 1>(d, dc, 2, "2 waves/SIMD: one MFMA-only (4 acc), one FMA-only (8 chains)");
  run<1, 0, 1>(d, dc, 2, "2 waves/SIMD: one MFMA-only (1 acc), one FMA-only (8 chains)");
  return 0;
}
