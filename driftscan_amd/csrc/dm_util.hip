// dm_util.hip — small memory-bound helpers (gfx950): transposes, identity,
// Hermitian symmetrisation, reductions.  All are plain coalesced streaming
// kernels; none is on the critical path.
#include "dm_common.h"
#include "dm_kernels.h"

#include <algorithm>

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

// ---- batched variants: one launch for a whole list of matrices --------------------------------
struct tdesc { const cplx* src; int lds; cplx* dst; int ldd; int rows; int cols; };

__global__ void ctrans_batched_kernel(const tdesc* __restrict__ ds) {
  __shared__ cplx tile[32][33];
  const tdesc d = ds[blockIdx.z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  if (bx >= d.cols || by >= d.rows) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int r = by + j, c = bx + tx;
    tile[j][tx] = (r < d.rows && c < d.cols) ? d.src[(size_t)r * d.lds + c] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int r = bx + j, c = by + tx;
    if (r < d.cols && c < d.rows) {
      cplx v = tile[tx][j];
      d.dst[(size_t)r * d.ldd + c] = make_double2(v.x, -v.y);
    }
  }
}

struct mdesc { cplx* a; int ld; int n; };

__global__ void identity_batched_kernel(const mdesc* __restrict__ ds) {
  const mdesc d = ds[blockIdx.z];
  const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < d.n && c < d.n) d.a[(size_t)r * d.ld + c] = make_double2(r == c ? 1.0 : 0.0, 0.0);
}

__global__ void hermitize_batched_kernel(const mdesc* __restrict__ ds) {
  const mdesc d = ds[blockIdx.z];
  const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d.n || c >= d.n || c > r) return;
  cplx lo = d.a[(size_t)r * d.ld + c], up = d.a[(size_t)c * d.ld + r];
  cplx v = make_double2(0.5 * (lo.x + up.x), 0.5 * (lo.y - up.y));
  if (r == c) v.y = 0.0;
  d.a[(size_t)r * d.ld + c] = v;
  d.a[(size_t)c * d.ld + r] = make_double2(v.x, -v.y);
}

struct cdesc { const char* src; char* dst; size_t bytes; };
__global__ void copy_batched_kernel(const cdesc* __restrict__ ds) {
  const cdesc d = ds[blockIdx.y];
  const size_t n16 = d.bytes / 16;
  const double2* s = reinterpret_cast<const double2*>(d.src);
  double2* t = reinterpret_cast<double2*>(d.dst);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) t[i] = s[i];
  if (blockIdx.x == 0)
    for (size_t i = n16 * 16 + threadIdx.x; i < d.bytes; i += blockDim.x) d.dst[i] = d.src[i];
}

}  // namespace

int dm_conj_transpose_batched(dm_ctx* ctx, const std::vector<dm_tdesc>& v) {
  if (v.empty()) return DM_OK;
  if (v.size() > 32768) {  // the descriptor index rides in gridDim.z
    for (size_t i0 = 0; i0 < v.size(); i0 += 32768) {
      std::vector<dm_tdesc> part(v.begin() + i0, v.begin() + std::min(v.size(), i0 + 32768));
      DM_TRY(dm_conj_transpose_batched(ctx, part));
    }
    return DM_OK;
  }
  std::vector<tdesc> h(v.size());
  int mr = 0, mc = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    h[i] = tdesc{v[i].src, v[i].lds, v[i].dst, v[i].ldd, v[i].rows, v[i].cols};
    mr = std::max(mr, v[i].rows);
    mc = std::max(mc, v[i].cols);
  }
  if (mr == 0 || mc == 0) return DM_OK;
  tdesc* d = dm_ws_upload(ctx, h);
  if (!d) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_UTIL, ctrans_batched_kernel, dim3((mc + 31) / 32, (mr + 31) / 32, (unsigned)v.size()), dim3(256), 0,
                     ctx->stream, d);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

static int launch_mdesc(dm_ctx* ctx, const std::vector<dm_mat>& v, int which) {
  if (v.empty()) return DM_OK;
  std::vector<mdesc> h(v.size());
  int mn = 0;
  for (size_t i = 0; i < v.size(); ++i) { h[i] = mdesc{v[i].p, v[i].ld, v[i].n}; mn = std::max(mn, v[i].n); }
  if (mn == 0) return DM_OK;
  mdesc* d = dm_ws_upload(ctx, h);
  if (!d) return DM_ENOMEM;
  dim3 grid((mn + 255) / 256, mn, (unsigned)v.size());
  if (which == 0) DM_PLAUNCH(ctx, DM_PROF_UTIL, identity_batched_kernel, grid, dim3(256), 0, ctx->stream, d);
  else DM_PLAUNCH(ctx, DM_PROF_UTIL, hermitize_batched_kernel, grid, dim3(256), 0, ctx->stream, d);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_set_identity_batched(dm_ctx* ctx, const std::vector<dm_mat>& v) { return launch_mdesc(ctx, v, 0); }
int dm_hermitize_batched(dm_ctx* ctx, const std::vector<dm_mat>& v) { return launch_mdesc(ctx, v, 1); }

int dm_copy_batched(dm_ctx* ctx, const std::vector<dm_cdesc>& v) {
  if (v.empty()) return DM_OK;
  std::vector<cdesc> h(v.size());
  size_t mb = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    h[i] = cdesc{reinterpret_cast<const char*>(v[i].src), reinterpret_cast<char*>(v[i].dst), v[i].bytes};
    mb = std::max(mb, v[i].bytes);
  }
  if (mb == 0) return DM_OK;
  cdesc* d = dm_ws_upload(ctx, h);
  if (!d) return DM_ENOMEM;
  const unsigned gx = (unsigned)std::min<size_t>(64, (mb / 16 + 255) / 256 + 1);
  DM_PLAUNCH(ctx, DM_PROF_UTIL, copy_batched_kernel, dim3(gx, (unsigned)v.size()), dim3(256), 0, ctx->stream, d);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_conj_transpose(dm_ctx* ctx, const cplx* src, int lds, cplx* dst, int ldd, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return DM_OK;
  DM_PLAUNCH(ctx, DM_PROF_UTIL, ctrans_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, ctx->stream, src, lds,
                     dst, ldd, rows, cols);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_set_identity(dm_ctx* ctx, cplx* a, int ld, int n) {
  if (n <= 0) return DM_OK;
  DM_PLAUNCH(ctx, DM_PROF_UTIL, identity_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, a, ld, n);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_hermitize(dm_ctx* ctx, cplx* a, int ld, int n) {
  if (n <= 0) return DM_OK;
  DM_PLAUNCH(ctx, DM_PROF_UTIL, hermitize_kernel, dim3((n + 255) / 256, n), dim3(256), 0, ctx->stream, a, ld, n);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_fill_zero(dm_ctx* ctx, void* p, size_t bytes) {
  if (bytes == 0) return DM_OK;
  dm_prof_scope ps(ctx, DM_PROF_UTIL, 0.0);
  DM_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
  return DM_OK;
}
