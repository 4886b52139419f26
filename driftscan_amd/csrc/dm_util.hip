// dm_util.hip — small memory-bound helpers (gfx950): transposes, identity,
// Hermitian symmetrisation, reductions.  All are plain coalesced streaming
// kernels; none is on the critical path.
#include "dm_common.h"
#include "dm_kernels.h"

namespace {

__global__ void ctrans_kernel(const cplx* __restrict__ src, int lds, cplx* __restrict__ dst, int ldd, int rows,
                              int cols) {
  __shared__ cplx tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int r = by + j, c = bx + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * lds + c] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int r = bx + j, c = by + tx;  // dst is cols x rows
    if (r < cols && c < rows) {
      cplx v = tile[tx][j];
      dst[(size_t)r * ldd + c] = make_double2(v.x, -v.y);
    }
  }
}

__global__ void identity_kernel(cplx* a, int ld, int n) {
  const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && c < n) a[(size_t)r * ld + c] = make_double2(r == c ? 1.0 : 0.0, 0.0);
}

// lower triangle <- average with conj of upper, then mirror
__global__ void hermitize_kernel(cplx* a, int ld, int n) {
  const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || c >= n || c > r) return;
  cplx lo = a[(size_t)r * ld + c], up = a[(size_t)c * ld + r];
  cplx v = make_double2(0.5 * (lo.x + up.x), 0.5 * (lo.y - up.y));
  if (r == c) v.y = 0.0;
  a[(size_t)r * ld + c] = v;
  a[(size_t)c * ld + r] = make_double2(v.x, -v.y);
}

}  // namespace

int dm_conj_transpose(dm_ctx* ctx, const cplx* src, int lds, cplx* dst, int ldd, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return DM_OK;
  hipLaunchKernelGGL(ctrans_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, ctx->stream, src, lds,
                     dst, ldd, rows, cols);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_set_identity(dm_ctx* ctx, cplx* a, int ld, int n) {
  if (n <= 0) return DM_OK;
  hipLaunchKernelGGL(identity_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, a, ld, n);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_hermitize(dm_ctx* ctx, cplx* a, int ld, int n) {
  if (n <= 0) return DM_OK;
  hipLaunchKernelGGL(hermitize_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, a, ld, n);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_fill_zero(dm_ctx* ctx, void* p, size_t bytes) {
  if (bytes == 0) return DM_OK;
  DM_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
  return DM_OK;
}
