// issue-rate probe for v_mfma_f64_4x4x4_4b_f64 and for 16x16x4 with VALU interleaved
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(double* out, int iters) {
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
template <typename F>
void run(const char* name, F launch, double flop_per_thread_iter_x64, int blocks, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); launch(iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = flop_per_thread_iter_x64 * iters * 4.0 * blocks;
  printf("%-28s blocks %5d: %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
}
int main() {
  double* out; (void)hipMalloc(&out, sizeof(double) * 4096 * 256);
  for (int b : {512, 1024, 2048}) {
    run("4x4x4_4b NACC=8", [&](int it) { k4<8><<<b, 256>>>(out, it); }, 512.0 * 8, b, 4000);
    run("4x4x4_4b NACC=16", [&](int it) { k4<16><<<b, 256>>>(out, it); }, 512.0 * 16, b, 2000);
    run("16x16x4 NACC=8", [&](int it) { k16<8><<<b, 256>>>(out, it); }, 2048.0 * 8, b, 2000);
  }
  return 0;
}
