// dm_kl.hip — covariance projection into the SVD basis and the generalised
// Hermitian eigenproblem of the KL transform, batched over m-blocks.
//
//   dm_project_cov   drift/core/beamtransfer.py:1135-1188  project_matrix_sky_to_svd
//   dm_project_diag  drift/core/beamtransfer.py:1190-1231  project_matrix_diagonal_telescope_to_svd
//   dm_regularise    drift/core/kltransform.py:288-290     diag += reg * max(N)
//   dm_eigh_gen      drift/core/kltransform.py:55-121      eigh_gen (zhegvd + non-PD rescue)
//
// The projections are grouped ZGEMMs on the fp64 matrix cores: one 64x64-tiled
// launch per non-zero polarisation pair covers every (f, f') block of every m.
// The reference multiplies all P^2 pairs including the all-zero ones
// (beamtransfer.py:1168-1169); the caller passes a mask of the non-zero pairs.
#include "dm_common.h"
#include "dm_kernels.h"
#include "../../include/driftmi.h"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace {

struct blk_desc {
  cplx* a; int n;
};

constexpr int BLK_SPLIT = 16;  // workgroups per matrix for the streaming reductions below

__device__ __forceinline__ bool lex_gt(double are, double aim, double bre, double bim) {
  return are > bre || (are == bre && aim > bim);
}

// numpy's max of a complex array is lexicographic in (re, im).  Stage 1: BLK_SPLIT partial
// maxima per matrix; stage 2 folds them and shifts the diagonal.
__global__ __launch_bounds__(256) void lexmax_partial_kernel(const blk_desc* __restrict__ bd, double2* __restrict__ part) {
  const blk_desc d = bd[blockIdx.y];
  __shared__ double sre[256], sim[256];
  double bre = -INFINITY, bim = -INFINITY;
  const size_t tot = (size_t)d.n * d.n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)BLK_SPLIT * 256) {
    cplx v = d.a[i];
    if (lex_gt(v.x, v.y, bre, bim)) { bre = v.x; bim = v.y; }
  }
  sre[threadIdx.x] = bre;
  sim[threadIdx.x] = bim;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      double ore = sre[threadIdx.x + s], oim = sim[threadIdx.x + s];
      if (lex_gt(ore, oim, sre[threadIdx.x], sim[threadIdx.x])) {
        sre[threadIdx.x] = ore;
        sim[threadIdx.x] = oim;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * BLK_SPLIT + blockIdx.x] = make_double2(sre[0], sim[0]);
}

__global__ __launch_bounds__(256) void lexmax_adddiag_kernel(const blk_desc* __restrict__ bd, const double2* __restrict__ part,
                                                             double reg) {
  const blk_desc d = bd[blockIdx.y];
  if (d.n == 0) return;
  double bre = -INFINITY, bim = -INFINITY;
  for (int s = 0; s < BLK_SPLIT; ++s) {
    const double2 v = part[(size_t)blockIdx.y * BLK_SPLIT + s];
    if (lex_gt(v.x, v.y, bre, bim)) { bre = v.x; bim = v.y; }
  }
  const double are = reg * bre, aim = reg * bim;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < d.n) {
    cplx v = d.a[(size_t)i * d.n + i];
    d.a[(size_t)i * d.n + i] = make_double2(v.x + are, v.y + aim);
  }
}

// flag[b] stays 1 only if every element of block b is exactly zero (flag preset to 1)
__global__ __launch_bounds__(256) void flag_set_kernel(int* __restrict__ flag, int n, int val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = val;
}
__global__ __launch_bounds__(256) void allzero_kernel(const blk_desc* __restrict__ bd, int* __restrict__ flag) {
  const blk_desc d = bd[blockIdx.y];
  const size_t tot = (size_t)d.n * d.n;
  int mine = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)BLK_SPLIT * 256) {
    cplx v = d.a[i];
    if (v.x != 0.0 || v.y != 0.0) mine = 1;
  }
  if (mine) flag[blockIdx.y] = 0;  // benign race: every writer stores the same value
}

__global__ void add_diag_kernel(const blk_desc* __restrict__ bd, const double* __restrict__ val) {
  const blk_desc d = bd[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) d.a[(size_t)i * d.n + i].x += val[blockIdx.y];
}


int copy_async(dm_ctx* ctx, cplx* dst, const cplx* src, size_t n) {
  if (n == 0) return DM_OK;
  DM_HIP(ctx, hipMemcpyAsync(dst, src, n * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
  return DM_OK;
}

}  // namespace

extern "C" {

// out[b] (ndof_b x ndof_b, blocks laid out back to back) += sum over the masked pol pairs of
//   (B[f,:n_f,pi,:] * cl[pi,pj,f,f',:]) B[f',:n_f',pj,:]^H
int dm_project_cov(dm_ctx* ctx, int nblk, int F, int K, int P, int L, const void* beam_svd_dev,
                   const int* svnum_host, const int* l0_host, const double* cl_pfl_dev, int npol,
                   const int* polmask_host, void* out_dev, const int64_t* out_off_host, int zero_first) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && K > 0 && P > 0 && L > 0 && beam_svd_dev && svnum_host && cl_pfl_dev &&
                  out_dev && out_off_host && npol >= 1 && npol <= P);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  const cplx* beam = reinterpret_cast<const cplx*>(beam_svd_dev);
  cplx* out = reinterpret_cast<cplx*>(out_dev);
  const int PL = P * L;
  std::vector<std::vector<int>> bounds(nblk, std::vector<int>(F + 1, 0));
  for (int b = 0; b < nblk; ++b) {
    for (int f = 0; f < F; ++f) bounds[b][f + 1] = bounds[b][f] + svnum_host[b * F + f];
  }
  // One product per (block, row frequency): the columns run over ALL frequencies of the block
  // (N = ndof) and are gathered from the (f', mode) rows of beam_svd, each carrying the weights
  // C_l(f, f') of its own frequency on the contraction index.  Against one product per (f, f')
  // pair this is F times fewer descriptors and tiles that are full in N (the per-frequency
  // blocks are only <= K <= 92 wide).  The first pol pair overwrites (beta = 0).
  std::vector<int2> gat;
  std::vector<size_t> goff((size_t)nblk * npol, 0);
  for (int b = 0; b < nblk; ++b) {
    const int l0 = l0_host ? std::min(std::max(l0_host[b], 0), L) : 0;
    for (int pj = 0; pj < npol; ++pj) {
      goff[(size_t)b * npol + pj] = gat.size();
      for (int fj = 0; fj < F; ++fj)
        for (int r = 0; r < svnum_host[b * F + fj]; ++r)
          gat.push_back(make_int2((int)(((size_t)fj * K + r) * PL + (size_t)pj * L + l0), fj * L + l0));
    }
  }
  int2* d_gat = dm_ws_upload(ctx, gat);
  if (!d_gat) return DM_ENOMEM;
  // `zero_first` carries two flag bits: 1 = clear the outputs first, 2 = the caller has verified
  // cl[pi,pj,f,f',l] == cl[pi,pj,f',f,l] (frequency-symmetric covariance)
  bool first_pair = (zero_first & 1) != 0;
  // When the output is overwritten and only diagonal pol pairs contribute (the usual sky models: C_l of
  // (T,T), (Q,Q), (U,U)), every pair's contribution is Hermitian: block (f, f') is the conjugate transpose of
  // (f', f).  Only the frequency blocks f' >= f are formed (half of the products) and mirrored afterwards.
  // (the reference, beamtransfer.py:1135-1188, accepts any array: without the symmetry bit every block is formed)
  bool herm = (zero_first & 1) != 0 && (zero_first & 2) != 0 && !getenv("DM_COV_FULL");
  for (int pi = 0; pi < npol && herm; ++pi)
    for (int pj = 0; pj < npol; ++pj)
      if (pi != pj && !(polmask_host && !polmask_host[pi * P + pj])) { herm = false; break; }
  for (int pi = 0; pi < npol; ++pi)
    for (int pj = 0; pj < npol; ++pj) {
      if (polmask_host && !polmask_host[pi * P + pj]) continue;
      const double beta = first_pair ? 0.0 : 1.0;
      std::vector<dm_gemm_desc> g;
      for (int b = 0; b < nblk; ++b) {
        const int ndof = bounds[b][F];
        if (ndof == 0) continue;
        const int l0 = l0_host ? std::min(std::max(l0_host[b], 0), L) : 0;
        if (L - l0 <= 0) {
          if (first_pair) DM_TRY(dm_fill_zero(ctx, out + out_off_host[b], sizeof(cplx) * (size_t)ndof * ndof));
          continue;
        }
        cplx* ob = out + out_off_host[b];
        const cplx* Bb = beam + ((size_t)b * F * K) * PL;
        for (int fi = 0; fi < F; ++fi) {
          const int ni = svnum_host[b * F + fi];
          if (ni == 0) continue;
          const int c0 = herm ? bounds[b][fi] : 0;  // first column formed
          const cplx* Ai = beam + (((size_t)b * F + fi) * K) * PL + (size_t)pi * L + l0;
          const double* cl = cl_pfl_dev + (((size_t)pi * P + pj) * F + fi) * F * L;
          dm_gemm_desc d = dm_gemm_make(Ai, PL, 1, false, Bb, 1, PL, true, ob + (size_t)bounds[b][fi] * ndof + c0, ndof, ni,
                                        ndof - c0, L - l0, 1.0, beta, cl, DM_GEMM_B_GATHER);
          d.bgather = d_gat + goff[(size_t)b * npol + pj] + c0;
          g.push_back(d);
        }
      }
      DM_TRY(dm_gemm_grouped_launch(ctx, g));
      first_pair = false;
    }
  if (herm && !first_pair) {
    // lower frequency blocks <- conjugate transposes of the upper ones (disjoint source and destination)
    std::vector<dm_tdesc> tr;
    for (int b = 0; b < nblk; ++b) {
      const int ndof = bounds[b][F];
      const int l0 = l0_host ? std::min(std::max(l0_host[b], 0), L) : 0;
      if (ndof == 0 || L - l0 <= 0) continue;
      cplx* ob = out + out_off_host[b];
      for (int fi = 0; fi < F; ++fi) {
        const int ni = svnum_host[b * F + fi], r0 = bounds[b][fi], c1 = bounds[b][fi + 1];
        if (ni == 0 || c1 >= ndof) continue;
        tr.push_back(dm_tdesc{ob + (size_t)r0 * ndof + c1, ndof, ob + (size_t)c1 * ndof + r0, ndof, ni, ndof - c1});
      }
    }
    DM_TRY(dm_conj_transpose_batched(ctx, tr));
  }
  if (first_pair)  // every pair masked: the contract is still out = 0
    for (int b = 0; b < nblk; ++b)
      if (bounds[b][F] > 0)
        DM_TRY(dm_fill_zero(ctx, out + out_off_host[b], sizeof(cplx) * (size_t)bounds[b][F] * bounds[b][F]));
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// out[b] block-diagonal: block f (+)= alpha * (U[f,:n_f,:] * d[f,:]) U[f,:n_f,:]^H
int dm_project_diag(dm_ctx* ctx, int nblk, int F, int K, int T, const void* beam_ut_dev, const int* svnum_host,
                    const double* dmat_dev, double alpha, void* out_dev, const int64_t* out_off_host,
                    int accumulate) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && K > 0 && T > 0 && beam_ut_dev && svnum_host && dmat_dev && out_dev &&
                  out_off_host);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  const cplx* ut = reinterpret_cast<const cplx*>(beam_ut_dev);
  cplx* out = reinterpret_cast<cplx*>(out_dev);
  std::vector<dm_gemm_desc> g;
  for (int b = 0; b < nblk; ++b) {
    int ndof = 0;
    for (int f = 0; f < F; ++f) ndof += svnum_host[b * F + f];
    if (!accumulate) DM_TRY(dm_fill_zero(ctx, out + out_off_host[b], sizeof(cplx) * (size_t)ndof * ndof));
    int off = 0;
    for (int f = 0; f < F; ++f) {
      const int n = svnum_host[b * F + f];
      if (n == 0) continue;
      const cplx* U = ut + (((size_t)b * F + f) * K) * T;
      g.push_back(dm_gemm_make(U, T, 1, false, U, 1, T, true, out + out_off_host[b] + (size_t)off * ndof + off, ndof,
                               n, n, T, alpha, accumulate ? 1.0 : 0.0, dmat_dev + (size_t)f * T));
      off += n;
    }
  }
  int rc = dm_gemm_grouped_launch(ctx, g);
  dm_ws_release(ctx, mark);
  return rc;
}

// diag(N_b) += reg * max(N_b)  with numpy's complex ordering (kltransform.py:289-290)
int dm_regularise(dm_ctx* ctx, int nblk, const int* n_host, void* mats_dev, const int64_t* off_host, double reg) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && n_host && mats_dev && off_host);
  if (nblk == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<blk_desc> bd(nblk);
  for (int b = 0; b < nblk; ++b) bd[b] = blk_desc{reinterpret_cast<cplx*>(mats_dev) + off_host[b], n_host[b]};
  blk_desc* d = dm_ws_upload(ctx, bd);
  if (!d) return DM_ENOMEM;
  double2* part = dm_ws_alloc_t<double2>(ctx, (size_t)nblk * BLK_SPLIT);
  if (!part) return DM_ENOMEM;
  int maxn = 0;
  for (int b = 0; b < nblk; ++b) maxn = std::max(maxn, n_host[b]);
  DM_PLAUNCH(ctx, DM_PROF_UTIL, lexmax_partial_kernel, dim3(BLK_SPLIT, nblk), dim3(256), 0, ctx->stream, d, part);
  DM_PLAUNCH(ctx, DM_PROF_UTIL, lexmax_adddiag_kernel, dim3((std::max(maxn, 1) + 255) / 256, nblk), dim3(256), 0, ctx->stream, d,
                     part, reg);
  DM_HIP(ctx, hipGetLastError());
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// Generalised Hermitian-definite eigenproblem A v = lambda B v for a batch of pencils.
// A and B are destroyed.  evals (ascending) at evals_dev + evoff[b]; evecs at
// evecs_dev + off[b] as an (n x n) row-major matrix whose ROWS are the modes, i.e.
// the reference's `evecs.T.conj()` (kltransform.py:345).  add_const_host[b] is the
// diagonal shift applied to B by the non-positive-definite rescue (0 normally).
// Synchronises.
int dm_eigh_gen(dm_ctx* ctx, int nblk, const int* n_host, void* A_dev, void* B_dev, const int64_t* off_host,
                double* evals_dev, const int64_t* evoff_host, void* evecs_dev, double* add_const_host,
                int* sweeps_host, int cut_mode, double cut_value, int* nkeep_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && n_host && A_dev && B_dev && off_host && evals_dev && evoff_host && evecs_dev &&
                  add_const_host && cut_mode >= 0 && cut_mode <= 2);
  if (sweeps_host) *sweeps_host = 0;
  if (nkeep_host)
    for (int b = 0; b < nblk; ++b) nkeep_host[b] = n_host[b];
  if (nblk == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  cplx* A = reinterpret_cast<cplx*>(A_dev);
  cplx* B = reinterpret_cast<cplx*>(B_dev);
  cplx* E = reinterpret_cast<cplx*>(evecs_dev);

  size_t tot = 0;
  int maxn = 0;
  for (int b = 0; b < nblk; ++b) {
    tot += (size_t)n_host[b] * n_host[b];
    maxn = std::max(maxn, n_host[b]);
    add_const_host[b] = 0.0;
  }
  // local offsets (dense, back to back) for the scratch copies
  std::vector<size_t> loff(nblk);
  {
    size_t o = 0;
    for (int b = 0; b < nblk; ++b) { loff[b] = o; o += (size_t)n_host[b] * n_host[b]; }
  }
  cplx* Lw = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));   // Cholesky factors
  cplx* Tw = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));   // transposes / C
  cplx* Ww = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));   // accumulated unitary
  double* evw = dm_ws_alloc_t<double>(ctx, (size_t)nblk * std::max(maxn, 1));
  int* info_dev = dm_ws_alloc_t<int>(ctx, nblk);
  int* zflag_dev = dm_ws_alloc_t<int>(ctx, nblk);
  if (!Lw || !Tw || !Ww || !evw || !info_dev || !zflag_dev) return DM_ENOMEM;

  // ---- all-zero shortcut (kltransform.py:81-85)
  std::vector<blk_desc> bdA(nblk);
  for (int b = 0; b < nblk; ++b) bdA[b] = blk_desc{A + off_host[b], n_host[b]};
  blk_desc* d_bdA = dm_ws_upload(ctx, bdA);
  if (!d_bdA) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_UTIL, flag_set_kernel, dim3((nblk + 255) / 256), dim3(256), 0, ctx->stream, zflag_dev, nblk, 1);
  DM_PLAUNCH(ctx, DM_PROF_UTIL, allzero_kernel, dim3(BLK_SPLIT, nblk), dim3(256), 0, ctx->stream, d_bdA, zflag_dev);
  // ---- Cholesky of B (on a copy, the rescue needs B itself)
  auto factor_launch = [&](const std::vector<int>& blks) -> int {
    std::vector<dm_mat> mats;
    std::vector<dm_cdesc> cp;
    for (int b : blks) {
      cp.push_back(dm_cdesc{B + off_host[b], Lw + loff[b], sizeof(cplx) * (size_t)n_host[b] * n_host[b]});
      mats.push_back(dm_mat{Lw + loff[b], n_host[b], n_host[b]});
    }
    DM_TRY(dm_copy_batched(ctx, cp));
    return dm_potrf_batched(ctx, mats, info_dev);
  };
  auto factor = [&](const std::vector<int>& blks, std::vector<int>& info) -> int {
    DM_TRY(factor_launch(blks));
    info.resize(blks.size());
    if (!blks.empty()) DM_TRY(dm_download(ctx, info.data(), info_dev, sizeof(int) * blks.size()));
    return DM_OK;
  };
  // The flags of the all-zero test and of the factorisation come back in ONE wait: the factorisation of every
  // non-empty block is queued before anything is read (an all-zero block fails it harmlessly — its factor is never
  // used), and the host builds the launch chains of the two triangular solves while the GPU is busy with it.
  auto trsm_lists = [&](const std::vector<int>& blks, std::vector<dm_trsm_problem>& t1, std::vector<dm_trsm_problem>& t2) {
    t1.clear();
    t2.clear();
    for (int b : blks) {
      t1.push_back(dm_trsm_problem{Lw + loff[b], n_host[b], n_host[b], A + off_host[b], n_host[b], n_host[b]});
      t2.push_back(dm_trsm_problem{Lw + loff[b], n_host[b], n_host[b], Tw + loff[b], n_host[b], n_host[b]});
    }
  };
  std::vector<int> cand;
  for (int b = 0; b < nblk; ++b)
    if (n_host[b] > 0) cand.push_back(b);
  static const bool host_times = getenv("DM_TIME_HOST") != nullptr;  // debugging aid: host time of the planning steps
  const auto ht0 = std::chrono::steady_clock::now();
  DM_TRY(factor_launch(cand));
  const auto ht1 = std::chrono::steady_clock::now();
  dm_trsm_plan plan1, plan2;
  {
    std::vector<dm_trsm_problem> t1, t2;
    trsm_lists(cand, t1, t2);
    DM_TRY(dm_trsm_plan_build(ctx, t1, false, false, plan1));   // X = L^-1 A
    DM_TRY(dm_trsm_plan_build(ctx, t2, false, true, plan2));    // Y = L^-1 X^H, on and above the diagonal
  }
  const auto ht2 = std::chrono::steady_clock::now();
  std::vector<int> zflag(nblk);
  DM_TRY(dm_download(ctx, zflag.data(), zflag_dev, sizeof(int) * nblk));
  if (host_times) {
    const auto ht3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "HOSTTIME eigh_gen: potrf plan+launch %.3f ms, trsm plans %.3f ms, wait for flags %.3f ms\n", ms(ht0, ht1),
            ms(ht1, ht2), ms(ht2, ht3));
  }
  std::vector<int> info_c(cand.size());
  if (!cand.empty()) DM_TRY(dm_download(ctx, info_c.data(), info_dev, sizeof(int) * cand.size()));

  std::vector<int> work;  // blocks that go through the solver
  std::vector<int> info;
  for (size_t i = 0; i < cand.size(); ++i) {
    const int b = cand[i];
    if (zflag[b]) {
      DM_TRY(dm_fill_zero(ctx, evals_dev + evoff_host[b], sizeof(double) * n_host[b]));
      DM_TRY(dm_set_identity(ctx, E + off_host[b], n_host[b], n_host[b]));
      if (nkeep_host && cut_mode) {  // all eigenvalues are 0: searchsorted gives 0 or n
        const int cut = 0.0 >= cut_value ? 0 : n_host[b];
        nkeep_host[b] = cut_mode == 1 ? n_host[b] - cut : cut;
      }
    } else {
      work.push_back(b);
      info.push_back(info_c[i]);
    }
  }
  if (work.size() != cand.size()) {  // some block took the shortcut: the solves run on the others only
    std::vector<dm_trsm_problem> t1, t2;
    trsm_lists(work, t1, t2);
    DM_TRY(dm_trsm_plan_build(ctx, t1, false, false, plan1));
    DM_TRY(dm_trsm_plan_build(ctx, t2, false, true, plan2));
  }
  std::vector<int> bad;
  for (size_t i = 0; i < work.size(); ++i)
    if (info[i] != 0) bad.push_back(work[i]);
  if (!bad.empty()) {
    // rescue (kltransform.py:101-111): add 1e-15 ev_max - 2 ev_min + 1e-60 to diag(B)
    std::vector<dm_jac_herm_problem> hp;
    for (int b : bad) {
      DM_TRY(copy_async(ctx, Tw + loff[b], B + off_host[b], (size_t)n_host[b] * n_host[b]));
      DM_TRY(dm_hermitize(ctx, Tw + loff[b], n_host[b], n_host[b]));
      hp.push_back(dm_jac_herm_problem{Tw + loff[b], n_host[b], Ww + loff[b], n_host[b], n_host[b]});
    }
    DM_TRY(dm_herm_eig_tridiag(ctx, hp, evw, std::max(maxn, 1)));
    std::vector<double> hev((size_t)bad.size() * std::max(maxn, 1));
    DM_TRY(dm_download(ctx, hev.data(), evw, sizeof(double) * hev.size()));
    std::vector<double> shifts(nblk, 0.0);
    for (size_t i = 0; i < bad.size(); ++i) {
      const int b = bad[i];
      const double* ev = &hev[i * std::max(maxn, 1)];
      double mn = ev[0], mx = ev[0];
      for (int k = 1; k < n_host[b]; ++k) { mn = std::min(mn, ev[k]); mx = std::max(mx, ev[k]); }
      add_const_host[b] = 1e-15 * mx - 2.0 * mn + 1e-60;
      shifts[b] = add_const_host[b];
    }
    std::vector<blk_desc> bdB(bad.size());
    std::vector<double> sh(bad.size());
    for (size_t i = 0; i < bad.size(); ++i) {
      bdB[i] = blk_desc{B + off_host[bad[i]], n_host[bad[i]]};
      sh[i] = shifts[bad[i]];
    }
    blk_desc* d_bdB = dm_ws_upload(ctx, bdB);
    double* d_sh = dm_ws_upload(ctx, sh);
    if (!d_bdB || !d_sh) return DM_ENOMEM;
    DM_PLAUNCH(ctx, DM_PROF_UTIL, add_diag_kernel, dim3((maxn + 255) / 256, (unsigned)bad.size()), dim3(256), 0, ctx->stream,
                       d_bdB, d_sh);
    std::vector<int> info2;
    DM_TRY(factor(bad, info2));
    for (size_t i = 0; i < bad.size(); ++i)
      if (info2[i] != 0) {
        ctx->err = "dm_eigh_gen: B is not positive definite even after the diagonal rescue";
        dm_ws_release(ctx, mark);
        return info2[i];  // > 0: LAPACK-style numerical failure
      }
  }

  // ---- C = L^-1 A L^-H
  {
    DM_TRY(dm_trsm_plan_run(ctx, plan1));  // X = L^-1 A
    std::vector<dm_tdesc> tr;
    for (int b : work) tr.push_back(dm_tdesc{A + off_host[b], n_host[b], Tw + loff[b], n_host[b], n_host[b], n_host[b]});
    DM_TRY(dm_conj_transpose_batched(ctx, tr));
    // Y = L^-1 X^H = C^H = C, on and above the diagonal only: the eigensolver reads the upper
    // triangle (LAPACK's zhegst/zheevd with one `uplo` do the same), so no symmetrisation pass either
    DM_TRY(dm_trsm_plan_run(ctx, plan2));
  }

  // ---- Hermitian eigendecomposition C = W^H diag(ev) W.  The eigenvalues come back to the host
  // once (natural order); the selection callback sorts them ascending (LAPACK convention), applies
  // the optional threshold cut and tells the solver which eigenvectors to back-transform and in
  // which order, so W arrives sorted and — with a cut — only nkeep rows are ever formed.
  std::vector<int> i_ev(nblk, 0);              // first kept row (cut_mode 1) / one past the last kept row (cut_mode 2)
  std::vector<std::vector<double>> evsorted(nblk);
  {
    std::vector<dm_jac_herm_problem> hp;
    for (int b : work) hp.push_back(dm_jac_herm_problem{Tw + loff[b], n_host[b], Ww + loff[b], n_host[b], n_host[b]});
    dm_eig_select sel;
    sel.pick = [&](int p, const double* ev, int n, std::vector<int>& cols) {
      const int b = work[p];
      std::vector<int> order(n);
      for (int i = 0; i < n; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ev[x] < ev[y]; });
      evsorted[b].resize(n);
      for (int i = 0; i < n; ++i) evsorted[b][i] = ev[order[i]];
      // np.searchsorted(evals, cut_value): first index with evals[i] >= cut_value
      const int cut = (int)(std::lower_bound(evsorted[b].begin(), evsorted[b].end(), cut_value) - evsorted[b].begin());
      i_ev[b] = cut;
      if (cut_mode == 1) cols.assign(order.begin() + cut, order.end());
      else if (cut_mode == 2) cols.assign(order.begin(), order.begin() + cut);
      else cols = order;
    };
    DM_TRY(dm_herm_eig_tridiag(ctx, hp, evw, std::max(maxn, 1), &sel));
    // sorted eigenvalues back to the device: one copy when the blocks are laid out back to back
    // (the usual case), else one per block
    bool contiguous = !work.empty();
    for (size_t i = 0; i + 1 < work.size() && contiguous; ++i)
      contiguous = evoff_host[work[i + 1]] == evoff_host[work[i]] + n_host[work[i]];
    if (contiguous) {
      std::vector<double> flat;
      for (int b : work) flat.insert(flat.end(), evsorted[b].begin(), evsorted[b].end());
      DM_TRY(dm_upload(ctx, evals_dev + evoff_host[work.front()], flat.data(), sizeof(double) * flat.size()));
    }
    for (size_t i = 0; i < work.size(); ++i) {
      const int b = work[i];
      if (nkeep_host) nkeep_host[b] = sel.nsel[i];
      if (!contiguous)
        DM_TRY(dm_upload(ctx, evals_dev + evoff_host[b], evsorted[b].data(), sizeof(double) * n_host[b]));
    }
  }

  // ---- back-transformation: rows of E = rows of W times L^-1  <=>  E^H = L^-H W^H, kept rows only;
  // they land at their sorted positions ([i_ev, n) or [0, i_ev)), the other rows of E are zeroed
  {
    std::vector<dm_trsm_problem> t3;
    std::vector<dm_tdesc> tr1, tr2;
    // rows that are not formed are zero: one memset over the span of the blocks when they are
    // stored back to back, else one per block
    bool span_ok = cut_mode != 0 && !work.empty();
    for (size_t i = 0; i + 1 < work.size() && span_ok; ++i)
      span_ok = off_host[work[i + 1]] == off_host[work[i]] + (int64_t)n_host[work[i]] * n_host[work[i]];
    if (span_ok) {
      const int bl = work.back();
      DM_TRY(dm_fill_zero(ctx, E + off_host[work.front()],
                          sizeof(cplx) * (size_t)(off_host[bl] + (int64_t)n_host[bl] * n_host[bl] - off_host[work.front()])));
    }
    for (size_t i = 0; i < work.size(); ++i) {
      const int b = work[i];
      const int n = n_host[b];
      const int nk = cut_mode == 1 ? n - i_ev[b] : (cut_mode == 2 ? i_ev[b] : n);
      const int row0 = cut_mode == 1 ? i_ev[b] : 0;
      if (nk < n && !span_ok) DM_TRY(dm_fill_zero(ctx, E + off_host[b], sizeof(cplx) * (size_t)n * n));
      if (nk <= 0) continue;
      tr1.push_back(dm_tdesc{Ww + loff[b], n, Tw + loff[b], nk, nk, n});           // W (nk x n) -> W^H (n x nk)
      t3.push_back(dm_trsm_problem{Lw + loff[b], n, n, Tw + loff[b], nk, nk});
      tr2.push_back(dm_tdesc{Tw + loff[b], nk, E + off_host[b] + (size_t)row0 * n, n, n, nk});  // -> rows of E
    }
    DM_TRY(dm_conj_transpose_batched(ctx, tr1));
    DM_TRY(dm_trsm_left_lower_batched(ctx, t3, true));
    DM_TRY(dm_conj_transpose_batched(ctx, tr2));
  }
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// One-call drivers: KLTransform._transform_m and DoubleKL._transform_m for a batch of m-blocks
// (drift/core/kltransform.py:258-355, drift/core/doublekl.py:30-87).  They compose the entry points above; the
// Python classes call those one by one (they also serve `inverse`, the files, the projections), a host without the
// Python layer calls these.
// ---------------------------------------------------------------------------------------------------------------
static int kl_covariances(dm_ctx* ctx, int nblk, int F, int K, int P, int L, int T, const void* beam_svd_dev,
                          const void* beam_ut_dev, const int* svnum_host, const int* l0_host, const double* cl_sg_dev,
                          const int* sg_mask_host, int sg_symmetric, const double* cl_fg_dev, const int* fg_mask_host,
                          int fg_symmetric, const double* npower_dev, double noise_scale, double regulariser,
                          const std::vector<int>& ndof, const std::vector<int64_t>& off, cplx* S, cplx* N) {
  // S = B C_sg B^H (kltransform.py:281); N = B C_fg B^H or 0 (:283-286); diag(N) += reg * max(N) (:289-290);
  // N += noise_scale * U diag(noisepower) U^H (:292-303)
  DM_TRY(dm_project_cov(ctx, nblk, F, K, P, L, beam_svd_dev, svnum_host, l0_host, cl_sg_dev, P, sg_mask_host, S, off.data(),
                        1 | (sg_symmetric ? 2 : 0)));
  if (cl_fg_dev) {
    DM_TRY(dm_project_cov(ctx, nblk, F, K, P, L, beam_svd_dev, svnum_host, l0_host, cl_fg_dev, P, fg_mask_host, N, off.data(),
                          1 | (fg_symmetric ? 2 : 0)));
  } else {
    for (int b = 0; b < nblk; ++b) DM_TRY(dm_fill_zero(ctx, N + off[b], sizeof(cplx) * (size_t)ndof[b] * ndof[b]));
  }
  DM_TRY(dm_regularise(ctx, nblk, ndof.data(), N, off.data(), regulariser));
  DM_TRY(dm_project_diag(ctx, nblk, F, K, T, beam_ut_dev, svnum_host, npower_dev, noise_scale, N, off.data(), 1));
  return DM_OK;
}

int dm_kl_m(dm_ctx* ctx, int nblk, int F, int K, int P, int L, int T, const void* beam_svd_dev, const void* beam_ut_dev,
            const int* svnum_host, const int* l0_host, const double* cl_sg_dev, const int* sg_mask_host, int sg_symmetric,
            const double* cl_fg_dev, const int* fg_mask_host, int fg_symmetric, const double* npower_dev, double noise_scale,
            double regulariser, int cut_mode, double cut_value, double* evals_dev, const int64_t* evoff_host, void* evecs_dev,
            const int64_t* off_host, double* add_const_host, int* nkeep_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && K > 0 && P > 0 && L > 0 && T > 0 && beam_svd_dev && beam_ut_dev && svnum_host && cl_sg_dev &&
                  npower_dev && evals_dev && evoff_host && evecs_dev && off_host && add_const_host);
  if (nblk == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);
  std::vector<int> ndof(nblk, 0);
  std::vector<int64_t> off(nblk);
  size_t tot = 0;
  for (int b = 0; b < nblk; ++b) {
    for (int f = 0; f < F; ++f) ndof[b] += svnum_host[b * F + f];
    off[b] = (int64_t)tot;
    tot += (size_t)ndof[b] * ndof[b];
  }
  cplx* S = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  cplx* N = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  if (!S || !N) return DM_ENOMEM;
  DM_TRY(kl_covariances(ctx, nblk, F, K, P, L, T, beam_svd_dev, beam_ut_dev, svnum_host, l0_host, cl_sg_dev, sg_mask_host,
                        sg_symmetric, cl_fg_dev, fg_mask_host, fg_symmetric, npower_dev, noise_scale, regulariser, ndof, off, S, N));
  // the caller's block offsets may differ from the dense local ones: solve in place, then copy the modes out
  int sweeps = 0;
  cplx* E = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  if (!E) return DM_ENOMEM;
  const int rc = dm_eigh_gen(ctx, nblk, ndof.data(), S, N, off.data(), evals_dev, evoff_host, E, add_const_host, &sweeps, cut_mode,
                             cut_value, nkeep_host);
  if (rc != DM_OK) return rc;
  std::vector<dm_cdesc> cp;
  for (int b = 0; b < nblk; ++b)
    if (ndof[b] > 0)
      cp.push_back(dm_cdesc{E + off[b], reinterpret_cast<cplx*>(evecs_dev) + off_host[b], sizeof(cplx) * (size_t)ndof[b] * ndof[b]});
  DM_TRY(dm_copy_batched(ctx, cp));
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DM_OK;
}

int dm_doublekl_m(dm_ctx* ctx, int nblk, int F, int K, int P, int L, int T, const void* beam_svd_dev, const void* beam_ut_dev,
                  const int* svnum_host, const int* l0_host, const double* cl_sg_dev, const int* sg_mask_host, int sg_symmetric,
                  const double* cl_fg_dev, const int* fg_mask_host, int fg_symmetric, const double* npower_dev,
                  double floor_scale, double regulariser, double foreground_threshold, int cut_mode, double cut_value,
                  double* f_evals_dev, double* evals_dev, const int64_t* evoff_host, void* modes_dev, const int64_t* off_host,
                  int* nmodes_host, int* nkeep_host, double* add_const_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && K > 0 && P > 0 && L > 0 && T > 0 && beam_svd_dev && beam_ut_dev && svnum_host && cl_sg_dev &&
                  npower_dev && f_evals_dev && evals_dev && evoff_host && modes_dev && off_host && nmodes_host && add_const_host);
  if (nblk == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);
  std::vector<int> ndof(nblk, 0);
  std::vector<int64_t> off(nblk);
  size_t tot = 0;
  for (int b = 0; b < nblk; ++b) {
    for (int f = 0; f < F; ++f) ndof[b] += svnum_host[b * F + f];
    off[b] = (int64_t)tot;
    tot += (size_t)ndof[b] * ndof[b];
  }
  cplx* S = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  cplx* N = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  cplx* S2 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  cplx* N2 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  cplx* E1 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  if (!S || !N || !S2 || !N2 || !E1) return DM_ENOMEM;
  // ---- stage 1: signal against foregrounds, thermal noise down at the 1 mK floor (doublekl.py:44-52)
  DM_TRY(kl_covariances(ctx, nblk, F, K, P, L, T, beam_svd_dev, beam_ut_dev, svnum_host, l0_host, cl_sg_dev, sg_mask_host,
                        sg_symmetric, cl_fg_dev, fg_mask_host, fg_symmetric, npower_dev, floor_scale, regulariser, ndof, off, S, N));
  // stage 2 wants the same S and N with the thermal term at full strength: eigh_gen destroys its inputs, keep copies
  DM_HIP(ctx, hipMemcpyAsync(S2, S, sizeof(cplx) * tot, hipMemcpyDeviceToDevice, ctx->stream));
  DM_HIP(ctx, hipMemcpyAsync(N2, N, sizeof(cplx) * tot, hipMemcpyDeviceToDevice, ctx->stream));
  DM_TRY(dm_project_diag(ctx, nblk, F, K, T, beam_ut_dev, svnum_host, npower_dev, 1.0 - floor_scale, N2, off.data(), 1));
  std::vector<int> keep(nblk, 0);
  int sweeps = 0;
  // modes with S/F strictly above the threshold (doublekl.py:56-60): eigenvalues ascend, so they are the trailing rows
  int rc = dm_eigh_gen(ctx, nblk, ndof.data(), S, N, off.data(), f_evals_dev, evoff_host, E1, add_const_host, &sweeps, 1,
                       std::nextafter(foreground_threshold, INFINITY), keep.data());
  if (rc != DM_OK) return rc;
  // ---- stage 2 in the kept subspace (doublekl.py:70-80)
  std::vector<int> n2(nblk);
  std::vector<int64_t> off2(nblk), toff(nblk);
  size_t tot2 = 0, ttot = 0;
  for (int b = 0; b < nblk; ++b) {
    n2[b] = keep[b];
    off2[b] = (int64_t)tot2;
    tot2 += (size_t)n2[b] * n2[b];
    toff[b] = (int64_t)ttot;
    ttot += (size_t)n2[b] * ndof[b];
  }
  cplx* cs = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot2, 1));
  cplx* cn = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot2, 1));
  cplx* E2 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot2, 1));
  cplx* tmp = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(2 * ttot, 1));
  double* ev2 = dm_ws_alloc_t<double>(ctx, std::max<size_t>((size_t)nblk * std::max(K * F, 1), 1));
  if (!cs || !cn || !E2 || !tmp || !ev2) return DM_ENOMEM;
  {
    std::vector<dm_gemm_desc> g1, g2;
    for (int b = 0; b < nblk; ++b) {
      const int n = ndof[b], r = n2[b];
      if (r <= 0) continue;
      const cplx* Ek = E1 + off[b] + (size_t)(n - r) * n;   // kept rows of E1 (r x n)
      for (int j = 0; j < 2; ++j) {
        cplx* Tj = tmp + (size_t)j * ttot + toff[b];
        g1.push_back(dm_gemm_make(Ek, n, 1, false, (j ? N2 : S2) + off[b], n, 1, false, Tj, n, r, n, n));        // E C
        g2.push_back(dm_gemm_make(Tj, n, 1, false, Ek, 1, n, true, (j ? cn : cs) + off2[b], r, r, r, n));          // (E C) E^H
      }
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g1));
    DM_TRY(dm_gemm_grouped_launch(ctx, g2));
  }
  std::vector<int64_t> evoff2(nblk);
  {
    int64_t o = 0;
    for (int b = 0; b < nblk; ++b) { evoff2[b] = o; o += n2[b]; }
  }
  std::vector<double> ac2(nblk, 0.0);
  std::vector<int> nk2(nblk, 0);
  rc = dm_eigh_gen(ctx, nblk, n2.data(), cs, cn, off2.data(), ev2, evoff2.data(), E2, ac2.data(), &sweeps, cut_mode, cut_value,
                   nk2.data());
  if (rc != DM_OK) return rc;
  // ---- modes = E2 . E1[kept] (doublekl.py:80), stage-2 eigenvalues to the caller's layout
  {
    std::vector<dm_gemm_desc> g;
    std::vector<dm_cdesc> cp;
    for (int b = 0; b < nblk; ++b) {
      const int n = ndof[b], r = n2[b];
      nmodes_host[b] = r;
      if (nkeep_host) nkeep_host[b] = nk2[b];
      if (r <= 0) continue;
      const cplx* Ek = E1 + off[b] + (size_t)(n - r) * n;
      g.push_back(dm_gemm_make(E2 + off2[b], r, 1, false, Ek, n, 1, false, reinterpret_cast<cplx*>(modes_dev) + off_host[b], n, r,
                               n, r));
      cp.push_back(dm_cdesc{ev2 + evoff2[b], evals_dev + evoff_host[b], sizeof(double) * (size_t)r});
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g));
    DM_TRY(dm_copy_batched(ctx, cp));
  }
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DM_OK;
}

}  // extern "C"
