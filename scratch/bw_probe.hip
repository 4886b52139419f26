// access-pattern bandwidth probe for the symmetric matvec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d2 __attribute__((address_space(1)))* gp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: one wave per row, lanes stride along the row
__global__ __launch_bounds__(256) void patA(const d2* A, int n, int nmat, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= n) return;
  gp a = (gp)(A + (size_t)blockIdx.y * n * n + (size_t)row * n);
  double s = 0;
  for (int c = lane; c < n; c += 64) { d2 v = a[c]; s += v.x + v.y; }
  if (s == 1.2345) out[0] = s;
}
// B: one wave per R rows, per iteration R loads at the same column chunk; MODE 1 staggers the chunk per row
template <int R, int MODE>
__global__ __launch_bounds__(256) void patB(const d2* A, int n, int nmat, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 4 + wave) * R;
  if (r0 >= n) return;
  gp a = (gp)(A + (size_t)blockIdx.y * n * n);
  const int nch = (n + 63) / 64;
  double s = 0;
#pragma unroll 1
  for (int t = 0; t < nch; ++t) {
    d2 v[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      int tt = MODE ? (t + rr * 3) % nch : t;
      int c = min(tt * 64 + lane, n - 1);
      v[rr] = a[(size_t)min(r0 + rr, n - 1) * n + c];
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) s += v[rr].x + v[rr].y;
  }
  if (s == 1.2345) out[0] = s;
}
int main() {
  for (int n : {1024, 1218, 900}) {
    const int nmat = 96;
    size_t bytes = (size_t)nmat * n * n * 16;
    d2* A; double* out;
    CK(hipMalloc(&A, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(A, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name) {
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("n=%4d %-28s %.1f us  %.2f TB/s\n", n, name, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
    };
    time([&] { patA<<<dim3((n + 3) / 4, nmat), 256>>>(A, n, nmat, out); }, "A wave/row");
    time([&] { patB<8, 0><<<dim3((n + 31) / 32, nmat), 256>>>(A, n, nmat, out); }, "B 8 rows same chunk");
    time([&] { patB<8, 1><<<dim3((n + 31) / 32, nmat), 256>>>(A, n, nmat, out); }, "B 8 rows staggered");
    time([&] { patB<4, 0><<<dim3((n + 15) / 16, nmat), 256>>>(A, n, nmat, out); }, "B 4 rows same chunk");
    time([&] { patB<2, 0><<<dim3((n + 7) / 8, nmat), 256>>>(A, n, nmat, out); }, "B 2 rows same chunk");
    time([&] { patB<16, 0><<<dim3((n + 63) / 64, nmat), 256>>>(A, n, nmat, out); }, "B 16 rows same chunk");
    (void)hipFree(A); (void)hipFree(out);
  }
  return 0;
}
