// dm_fisher.hip — exact per-m Fisher matrix of the band powers (quadratic power-spectrum estimator).
//
// Replaces PSExact._work_fisher_bias_m + makeproj (drift/core/psestimation.py:672-699, :775-815):
//   C_a  = E (B C_l^a B^H) E^H                      band a projected into the KL basis
//   F_ab = sum_ij C_a[i][j] C_b[j][i] / ((lam_i + 1)(lam_j + 1))
// With Et = diag((lam + 1)^-1/2) E and D_a = Et (B C_l^a B^H) Et^H (Hermitian) this is
//   F_ab = sum_ij D_a[i][j] conj(D_b[i][j]) = (D D^H)_ab,   D = [vec(D_0); vec(D_1); ...]
// i.e. per band one covariance projection (dm_project_cov, K = L) and two grouped ZGEMMs, then
// one Gram product over the vectorised D_a — all on the fp64 matrix cores, every m-block of the
// batch in the same launches.
#include "dm_common.h"
#include "dm_kernels.h"
#include "../../include/driftmi.h"

#include <algorithm>
#include <vector>

namespace {

struct scale_desc { const cplx* E; cplx* Et; const double* lam; int rows; int cols; };

// Et[r][:] = E[r][:] / sqrt(lam[r] + 1)
__global__ __launch_bounds__(256) void fisher_scale_rows_kernel(const scale_desc* __restrict__ ds) {
  const scale_desc d = ds[blockIdx.y];
  const size_t tot = (size_t)d.rows * d.cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d.cols);
    const double s = 1.0 / sqrt(d.lam[r] + 1.0);
    const cplx v = d.E[i];
    d.Et[i] = make_double2(v.x * s, v.y * s);
  }
}

struct sum_desc { const cplx* part; cplx* out; int nchunk; int n; };

// out[i] = sum_c part[c * n + i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void fisher_sum_chunks_kernel(const sum_desc* __restrict__ ds) {
  const sum_desc d = ds[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n) return;
  cplx acc = make_double2(0.0, 0.0);
  for (int c = 0; c < d.nchunk; ++c) acc = cadd(acc, d.part[(size_t)c * d.n + i]);
  d.out[i] = acc;
}

}  // namespace

extern "C" int dm_fisher(dm_ctx* ctx, int nblk, int F, int K, int P, int L, const void* beam_svd_dev,
                         const int* svnum_host, const int* l0_host, int nbands, const double* cl_bands_dev,
                         const void* evecs_dev, const int64_t* evecs_off_host, const int* nmodes_host,
                         const double* evals_dev, const int64_t* evals_off_host, void* fisher_dev, int cl_symmetric) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && K > 0 && P > 0 && L > 0 && nbands > 0 && beam_svd_dev && svnum_host &&
                  cl_bands_dev && evecs_dev && evecs_off_host && nmodes_host && evals_dev && evals_off_host &&
                  fisher_dev);
  if (nblk == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  const cplx* evecs = reinterpret_cast<const cplx*>(evecs_dev);
  cplx* fisher = reinterpret_cast<cplx*>(fisher_dev);
  DM_TRY(dm_fill_zero(ctx, fisher, sizeof(cplx) * (size_t)nblk * nbands * nbands));

  std::vector<int> ndof(nblk, 0);
  std::vector<int64_t> offS(nblk), offT(nblk), offD(nblk);
  size_t totS = 0, totT = 0, totD = 0;
  int maxnm = 0;
  for (int b = 0; b < nblk; ++b) {
    for (int f = 0; f < F; ++f) ndof[b] += svnum_host[b * F + f];
    const size_t n = ndof[b], nm = std::max(nmodes_host[b], 0);
    offS[b] = (int64_t)totS; totS += n * n;
    offT[b] = (int64_t)totT; totT += nm * n;
    offD[b] = (int64_t)totD; totD += nm * nm * (size_t)nbands;
    maxnm = std::max(maxnm, (int)nm);
  }
  if (maxnm == 0) { dm_ws_release(ctx, mark); return DM_OK; }
  cplx* S = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totS, 1));
  cplx* Et = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totT, 1));
  cplx* T = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totT, 1));
  cplx* D = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totD, 1));
  if (!S || !Et || !T || !D) return DM_ENOMEM;

  // Et = diag((lam + 1)^-1/2) E
  {
    std::vector<scale_desc> sd;
    size_t maxel = 0;
    for (int b = 0; b < nblk; ++b) {
      if (nmodes_host[b] <= 0 || ndof[b] <= 0) continue;
      sd.push_back(scale_desc{evecs + evecs_off_host[b], Et + offT[b], evals_dev + evals_off_host[b], nmodes_host[b],
                              ndof[b]});
      maxel = std::max(maxel, (size_t)nmodes_host[b] * ndof[b]);
    }
    scale_desc* d_sd = dm_ws_upload(ctx, sd);
    if (!d_sd) return DM_ENOMEM;
    const unsigned gx = (unsigned)std::min<size_t>((maxel + 255) / 256, 1024);
    DM_PLAUNCH(ctx, DM_PROF_UTIL, fisher_scale_rows_kernel, dim3(gx, (unsigned)sd.size()), dim3(256), 0, ctx->stream, d_sd);
  }
  for (int a = 0; a < nbands; ++a) {
    // S_b = B C_l^a B^H (temperature block only: makeproj calls project_matrix_sky_to_svd(temponly=True))
    DM_TRY(dm_project_cov(ctx, nblk, F, K, P, L, beam_svd_dev, svnum_host, l0_host,
                          cl_bands_dev + (size_t)a * F * F * L, 1, nullptr, S, offS.data(), 1 | (cl_symmetric ? 2 : 0)));
    std::vector<dm_gemm_desc> g1, g2;
    for (int b = 0; b < nblk; ++b) {
      const int n = ndof[b], nm = nmodes_host[b];
      if (nm <= 0 || n <= 0) continue;
      g1.push_back(dm_gemm_make(Et + offT[b], n, 1, false, S + offS[b], n, 1, false, T + offT[b], n, nm, n, n));
      g2.push_back(dm_gemm_make(T + offT[b], n, 1, false, Et + offT[b], 1, n, true,
                                D + offD[b] + (size_t)a * nm * nm, nm, nm, nm, n));
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g1));
    DM_TRY(dm_gemm_grouped_launch(ctx, g2));
  }
  // F_b = D_b D_b^H over the vectorised bands; the long contraction is cut into chunks so that a
  // block contributes many tiles, the chunk sums are added in a fixed order
  {
    const int CH = 16384;
    std::vector<dm_gemm_desc> g;
    std::vector<sum_desc> sd;
    size_t totP = 0;
    std::vector<size_t> offP(nblk);
    std::vector<int> nch(nblk, 0);
    for (int b = 0; b < nblk; ++b) {
      const size_t kk = (size_t)std::max(nmodes_host[b], 0) * std::max(nmodes_host[b], 0);
      nch[b] = (int)((kk + CH - 1) / CH);
      offP[b] = totP;
      totP += (size_t)nch[b] * nbands * nbands;
    }
    cplx* Pp = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totP, 1));
    if (!Pp) return DM_ENOMEM;
    for (int b = 0; b < nblk; ++b) {
      const size_t kk = (size_t)std::max(nmodes_host[b], 0) * std::max(nmodes_host[b], 0);
      if (kk == 0) continue;
      const cplx* Db = D + offD[b];
      for (int c = 0; c < nch[b]; ++c) {
        const size_t k0 = (size_t)c * CH;
        const int kc = (int)std::min<size_t>(CH, kk - k0);
        // Gram of the rows: A(a, k) = D[a][k0 + k], B(k, b') = conj(D[b'][k0 + k])
        g.push_back(dm_gemm_make(Db + k0, (int)kk, 1, false, Db + k0, 1, (int)kk, true,
                                 Pp + offP[b] + (size_t)c * nbands * nbands, nbands, nbands, nbands, kc));
      }
      sd.push_back(sum_desc{Pp + offP[b], fisher + (size_t)b * nbands * nbands, nch[b], nbands * nbands});
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g));
    if (!sd.empty()) {
      sum_desc* d_sd = dm_ws_upload(ctx, sd);
      if (!d_sd) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_UTIL, fisher_sum_chunks_kernel, dim3((nbands * nbands + 255) / 256, (unsigned)sd.size()), dim3(256), 0,
                         ctx->stream, d_sd);
    }
  }
  DM_HIP(ctx, hipGetLastError());
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}
