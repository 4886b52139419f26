// raw v_mfma_f64_16x16x4_f64 issue rate probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 2048.0 * NACC * iters * 4.0 * blocks;
  printf("NACC %2d blocks %5d: %.3f ms  %.1f TF\n", NACC, blocks, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int b : {256, 512, 1024, 2048}) { run<1>(b, 20000); run<4>(b, 5000); run<8>(b, 2500); run<16>(b, 1250); }
  // long run to see sustained clocks
  run<8>(2048, 50000);
  return 0;
}
