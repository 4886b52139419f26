// v_mfma_f64_16x16x4_f64 issue-rate probe, round 5: what bounds the instruction on this pool?
//   * operands: one (a, b) pair shared by all accumulators (the round-1 probe) vs an own pair per accumulator,
//     from registers, or re-read from LDS before every MFMA (the shape of jac_gram's inner loop);
//   * occupancy: 1 .. 4 waves per SIMD (workgroups of 256 threads, 1 .. 4 per CU);
//   * data: zeros vs O(1) values (power -> clock);
//   * the clock the chip holds: s_memtime (shader clock) against the wall time of the launch.
// Build: hipcc -O3 --offload-arch=gfx950 scratch/mfma_peak3.hip -o scratch/bin/mfma_peak3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// dependent chains: R back-to-back MFMAs on the SAME accumulator before the next accumulator is touched (what the
// complex products of jac_gram / jac_apply issue: two MFMAs into gre, two into gim, ...)
template <int NACC, int R>
__global__ __launch_bounds__(256) void kchain(double* out, unsigned long long* cyc, int iters, double scale) {
  f64x4 acc[NACC];
  double a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = f64x4{0, 0, 0, 0};
    a[i] = scale * (threadIdx.x * 1e-3 + i);
    b[i] = scale * (blockIdx.x * 1e-3 + 1.0 + 0.5 * i);
  }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + r) % NACC], b[i], acc[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same for v_mfma_f64_4x4x4_4b_f64 (one double of C per lane): does the accumulator change cost anything there?
template <int NACC, int R>
__global__ __launch_bounds__(256) void kchain4(double* out, int iters, double scale) {
  double acc[NACC], a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = 0.0;
    a[i] = scale * (threadIdx.x * 1e-3 + i);
    b[i] = scale * (blockIdx.x * 1e-3 + 1.0 + 0.5 * i);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(i + r) % NACC], b[i], acc[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int R>
void runchain4(int wg_per_cu, int iters, double scale) {
  const int blocks = 256 * wg_per_cu;
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kchain4<NACC, R><<<blocks, 256>>>(out, 10, scale);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kchain4<NACC, R><<<blocks, 256>>>(out, iters, scale);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 512.0 * NACC * R * (double)iters * 4.0 * blocks;
  printf("4x4x4 chain R %d NACC %2d waves/SIMD %d: %8.3f ms  %6.1f TF  (%.1f cycles per MFMA and SIMD at 2.35 GHz)\n", R, NACC, wg_per_cu, ms,
         fl / ms / 1e9, ms * 1e-3 * 2.35e9 / ((double)NACC * R * iters * wg_per_cu));
  hipFree(out);
}

template <int NACC, int R>
void runchain(int wg_per_cu, int iters, double scale) {
  const int blocks = 256 * wg_per_cu;
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  unsigned long long* cyc; hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kchain<NACC, R><<<blocks, 256>>>(out, cyc, 10, scale);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kchain<NACC, R><<<blocks, 256>>>(out, cyc, iters, scale);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = 2048.0 * NACC * R * (double)iters * 4.0 * blocks;
  printf("chain R %d NACC %2d waves/SIMD %d: %8.3f ms  %6.1f TF  (%.1f cycles per MFMA and SIMD at 2.35 GHz)\n", R, NACC, wg_per_cu, ms,
         fl / ms / 1e9, ms * 1e-3 * 2.35e9 / ((double)NACC * R * iters * wg_per_cu));
  hipFree(out); hipFree(cyc);
}

template <int NACC, int MODE>   // MODE 0: shared operands, 1: own operands (registers), 2: own operands re-read from LDS
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters, double scale) {
  __shared__ double lds[2 * 16 * 64];
  f64x4 acc[NACC];
  double a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) {
    acc[i] = f64x4{0, 0, 0, 0};
    a[i] = scale * (threadIdx.x * 1e-3 + i);
    b[i] = scale * (blockIdx.x * 1e-3 + 1.0 + 0.5 * i);
  }
  if (MODE == 2) {
    for (int i = threadIdx.x; i < 2 * 16 * 64; i += 256) lds[i] = scale * (1.0 + 1e-3 * i);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0);
      else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
      else {
        const double x = lds[(i & 15) * 64 + lane], y = lds[(16 + (i & 15)) * 64 + ((lane + it) & 63)];
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
void run(int wg_per_cu, int iters, double scale) {
  const int blocks = 256 * wg_per_cu;
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  unsigned long long* cyc; hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, MODE><<<blocks, 256>>>(out, cyc, 10, scale);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC, MODE><<<blocks, 256>>>(out, cyc, iters, scale);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double cmax = 0; for (auto c : h) cmax = cmax > (double)c ? cmax : (double)c;
  const double fl = 2048.0 * NACC * (double)iters * 4.0 * blocks;
  // cycles per MFMA per SIMD: a SIMD hosts wg_per_cu waves, each issuing NACC * iters MFMAs
  const double cyc_per_mfma = cmax / ((double)NACC * iters * wg_per_cu);
  printf("mode %d NACC %2d waves/SIMD %d scale %.0f: %8.3f ms  %6.1f TF  counter %.3g ticks (%.1f per MFMA and SIMD; counter runs at %.3f GHz)\n",
         MODE, NACC, wg_per_cu, scale, ms, fl / ms / 1e9, cmax, cyc_per_mfma, cmax / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  for (int w : {1, 2, 4}) {
    runchain<8, 1>(w, 10000 / w, 1.0);
    runchain<8, 2>(w, 5000 / w, 1.0);
    runchain<8, 4>(w, 2500 / w, 1.0);
    runchain<4, 8>(w, 2500 / w, 1.0);
    runchain<2, 16>(w, 2500 / w, 1.0);
    runchain<1, 32>(w, 2500 / w, 1.0);
  }
  for (int w : {1, 2, 3}) {
    runchain4<32, 1>(w, 10000 / w, 1.0);
    runchain4<16, 2>(w, 10000 / w, 1.0);
    runchain4<8, 4>(w, 10000 / w, 1.0);
    runchain4<4, 8>(w, 10000 / w, 1.0);
    runchain4<1, 32>(w, 10000 / w, 1.0);
  }
  if (argc > 1) return 0;
  for (double scale : {1.0, 0.0}) {
    for (int w : {1, 2, 3, 4}) {
      run<4, 0>(w, 20000 / w, scale);
      run<4, 1>(w, 20000 / w, scale);
      run<8, 1>(w, 10000 / w, scale);
      run<16, 1>(w, 5000 / w, scale);
      run<8, 2>(w, 10000 / w, scale);
    }
  }
  run<8, 1>(2, 200000, 1.0);   // sustained (2 s): what the clock settles at
  return 0;
}
