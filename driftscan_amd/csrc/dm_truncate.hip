// dm_truncate.hip — bit truncation of the beam-transfer blocks before they are written
// (drift/core/beamtransfer.py:641-646: caput.truncate.bit_truncate_max_complex on the rows of the
// m-ordered array, last axis = l).  HBM-bound streaming kernel on the blocks where they lie after
// BT-gen: one wavefront per row (a run of l values, 2-16 KB), two passes — the row maximum of |z|, then
// the rounding of every real and imaginary part — the second pass is served by L2.
//
// Algorithmic bytes per row: read 16 L + write 16 L (the second read hits L2).
//
// The arithmetic is restated in oracle/truncate.py with the same IEEE operations in the same order
// (no contraction into FMAs here), so the two agree bit for bit.  caput itself is not part of this
// image: the restatement follows its published contract (error bound per element, as many trailing
// zero mantissa bits as that bound allows) and is declared unpinned in DESIGN.md.
#include "dm_common.h"
#include "dm_kernels.h"
#include "../../include/driftmi.h"

namespace {

// Round x to the nearest multiple of 2^(k+1) units of its last place, where 2^k is the largest power of
// two not above err / ulp(x): |result - x| <= 2^k ulp <= err, ties to even.
__device__ __forceinline__ double bit_truncate_f64(double x, double err) {
  if (!(err > 0.0) || x == 0.0) return x;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const int ex = (int)((bits >> 52) & 0x7ffull);
  if (ex == 0x7ff) return x;  // inf / nan
  unsigned long long man = bits & 0x000fffffffffffffull;
  int e;
  if (ex == 0) {
    e = -1074;
  } else {
    e = ex - 1075;
    man |= 1ull << 52;
  }
  const double errs = ldexp(err, -e);  // the error budget in units of the last place of x
  if (!(errs >= 1.0)) return x;
  const unsigned long long errm = errs >= 4611686018427387904.0 ? (1ull << 62) : (unsigned long long)errs;
  const int k = 63 - __clzll((long long)errm);
  const unsigned long long q = 1ull << (k + 1), half = 1ull << k;
  const unsigned long long r = man & (q - 1ull);
  unsigned long long base = man - r;
  if (r > half || (r == half && ((base >> (k + 1)) & 1ull))) base += q;
  const double out = ldexp((double)base, e);
  return (bits >> 63) ? -out : out;
}

__device__ __forceinline__ double abs2_nofma(cplx z) { return __dadd_rn(__dmul_rn(z.x, z.x), __dmul_rn(z.y, z.y)); }

__global__ __launch_bounds__(256) void truncate_rows_kernel(cplx* __restrict__ a, long long nrows, int ncols, long long ld,
                                                            double prec, double prec_max_row) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  cplx* p = a + row * ld;
  double mx = 0.0;
  for (int j = lane; j < ncols; j += 64) mx = fmax(mx, abs2_nofma(dm_ldg(p, j)));
  mx = dm_wave_max(mx);
  const double floor_err = __dmul_rn(prec_max_row, sqrt(mx));
  for (int j = lane; j < ncols; j += 64) {
    const cplx z = dm_ldg(p, j);
    const double err = fmax(__dmul_rn(prec, sqrt(abs2_nofma(z))), floor_err);
    dm_stg(p, j, make_double2(bit_truncate_f64(z.x, err), bit_truncate_f64(z.y, err)));
  }
}

}  // namespace

extern "C" int dm_bit_truncate_max_complex(dm_ctx* ctx, void* data_dev, int64_t nrows, int ncols, int64_t ld, double prec,
                                           double prec_max_row) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, data_dev && nrows >= 0 && ncols >= 0 && ld >= ncols && prec >= 0.0 && prec_max_row >= 0.0);
  if (nrows == 0 || ncols == 0) return DM_OK;
  const long long nblk = (nrows + 3) / 4;
  DM_ARG(ctx, nblk <= 0x7fffffffLL);
  hipLaunchKernelGGL(truncate_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, reinterpret_cast<cplx*>(data_dev),
                     (long long)nrows, ncols, (long long)ld, prec, prec_max_row);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}
