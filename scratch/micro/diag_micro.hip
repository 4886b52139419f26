// Micro-benchmark (development aid): where do the ~29 us of a 32-row substitution launch go?
// variants: 0 = loads + stores only, 1 = + LDS fill and barrier, 2 = + substitution (the library's kernel body)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double2 cplx;
constexpr int NB = 32;
struct desc { const cplx* L; int ldl; int n; cplx* B; int ldb; int nrhs; };
__device__ inline cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
template <int V>
__global__ __launch_bounds__(256) void k(const desc* ds, int s) {
  __shared__ cplx Lk[NB][NB + 1];
  const desc d = ds[blockIdx.y];
  const int k0 = s * NB;
  if (k0 >= d.n) return;
  const int nb = min(NB, d.n - k0);
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (V >= 1) {
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
      int r = idx / NB, c = idx % NB;
      Lk[r][c] = (r < nb && c < nb) ? d.L[(size_t)(k0 + r) * d.ldl + k0 + c] : make_double2(0.0, 0.0);
    }
    __syncthreads();
  }
  if (col >= d.nrhs) return;
  cplx x[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) x[r] = (r < nb) ? d.B[(size_t)(k0 + r) * d.ldb + col] : make_double2(0.0, 0.0);
  if (V >= 2) {
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (r < nb) {
        const double iv = 1.0 / Lk[r][r].x;
        x[r] = make_double2(x[r].x * iv, x[r].y * iv);
#pragma unroll
        for (int j = r + 1; j < NB; ++j) x[j] = csub(x[j], cmul(Lk[j][r], x[r]));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (r < nb) d.B[(size_t)(k0 + r) * d.ldb + col] = x[r];
}
int main() {
  const int nmat = 129;
  std::vector<int> ns(nmat);
  size_t tot = 0;
  for (int i = 0; i < nmat; ++i) { ns[i] = 1218 - 8 * i; tot += (size_t)ns[i] * ns[i]; }
  cplx *L, *B;
  hipMalloc(&L, tot * sizeof(cplx));
  hipMalloc(&B, tot * sizeof(cplx));
  hipMemset(L, 0, tot * sizeof(cplx));
  hipMemset(B, 0, tot * sizeof(cplx));
  std::vector<desc> h(nmat);
  size_t o = 0;
  for (int i = 0; i < nmat; ++i) { h[i] = desc{L + o, ns[i], ns[i], B + o, ns[i], ns[i]}; o += (size_t)ns[i] * ns[i]; }
  desc* d;
  hipMalloc(&d, nmat * sizeof(desc));
  hipMemcpy(d, h.data(), nmat * sizeof(desc), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(5, nmat);
  for (int v = 0; v < 3; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int s = 0; s < 20; ++s) {
        if (v == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, d, s);
        if (v == 1) hipLaunchKernelGGL(k<1>, grid, dim3(256), 0, 0, d, s);
        if (v == 2) hipLaunchKernelGGL(k<2>, grid, dim3(256), 0, 0, d, s);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("variant %d: %.1f us per launch (20 launches back to back)\n", v, ms * 1000 / 20);
    }
  }
  // one empty-ish kernel for the launch floor
  return 0;
}
