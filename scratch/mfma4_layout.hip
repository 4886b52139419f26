// Decode the operand/result layout of v_mfma_f64_4x4x4_4b_f64 by indicator inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k(double* out) {
  const int l = threadIdx.x, w = blockIdx.x;  // w = la * 64 + lb
  const int la = w >> 6, lb = w & 63;
  double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
  double d = 0;
  d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, CBSZ, ABID, 0);
  out[(size_t)w * 64 + l] = d;
}
template <int CBSZ, int ABID>
void run(const char* name) {
  double* dout; (void)hipMalloc(&dout, 4096 * 64 * 8);
  k<CBSZ, ABID><<<4096, 64>>>(dout);
  std::vector<double> h(4096 * 64);
  (void)hipMemcpy(h.data(), dout, 4096 * 64 * 8, hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  // for each A lane: which B lanes pair with it, and to which output lane
  for (int la : {0, 1, 2, 3, 4, 5, 8, 12, 16, 17, 20, 32, 48, 63}) {
    printf("A lane %2d:", la);
    int n = 0;
    for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) { if (n < 18) printf(" (B%d->D%d)", lb, l); ++n; }
    printf("  [%d]\n", n);
  }
  (void)hipFree(dout);
}
int main() {
  run<0, 0>("cbsz=0");
  run<2, 0>("cbsz=2 abid=0");
  run<2, 1>("cbsz=2 abid=1");
  return 0;
}
