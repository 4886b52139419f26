// dm_api.hip — extern "C" entry points of libdriftmi (see include/driftmi.h).
#include "dm_common.h"
#include "dm_kernels.h"
#include "../../include/driftmi.h"

extern "C" {

int dm_zgemm_strided_batched(dm_ctx* ctx, int M, int N, int K, double alpha, const void* A, int rsA, int csA,
                             int conjA, int64_t strideA, const void* B, int rsB, int csB, int conjB,
                             int64_t strideB, double beta, void* C, int ldc, int64_t strideC,
                             const double* kscale, int64_t stride_kscale, int batch) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, M >= 0 && N >= 0 && K >= 0 && batch >= 0 && A && B && C);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<dm_gemm_desc> g;
  g.reserve(batch);
  for (int b = 0; b < batch; ++b) {
    g.push_back(dm_gemm_make(reinterpret_cast<const cplx*>(A) + b * strideA, rsA, csA, conjA != 0,
                             reinterpret_cast<const cplx*>(B) + b * strideB, rsB, csB, conjB != 0,
                             reinterpret_cast<cplx*>(C) + b * strideC, ldc, M, N, K, alpha, beta,
                             kscale ? kscale + b * stride_kscale : nullptr));
  }
  int rc = dm_gemm_grouped_launch(ctx, g);
  dm_ws_release(ctx, mark);
  return rc;
}

int dm_zgemm_grouped(dm_ctx* ctx, int nprob, const dm_zgemm_problem* probs) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nprob >= 0 && (nprob == 0 || probs));
  dm_ws_scope ws_scope__(ctx);
  std::vector<dm_gemm_desc> g;
  g.reserve(nprob);
  for (int i = 0; i < nprob; ++i) {
    const dm_zgemm_problem& p = probs[i];
    DM_ARG(ctx, p.M >= 0 && p.N >= 0 && p.K >= 0);
    if (p.M == 0 || p.N == 0) continue;
    DM_ARG(ctx, p.A && p.B && p.C);
    g.push_back(dm_gemm_make(reinterpret_cast<const cplx*>(p.A), p.rsA, p.csA, p.conjA != 0, p.B, p.rsB, p.csB,
                             p.conjB != 0, reinterpret_cast<cplx*>(p.C), p.ldc, p.M, p.N, p.K, p.alpha, p.beta));
  }
  return dm_gemm_grouped_launch(ctx, g);
}

int dm_zpotrf_batched(dm_ctx* ctx, int n, void* A, int ld, int64_t stride, int batch, int* info_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, n >= 0 && batch >= 0 && A && info_host);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<dm_mat> mats(batch);
  for (int b = 0; b < batch; ++b) mats[b] = dm_mat{reinterpret_cast<cplx*>(A) + b * stride, ld, n};
  int* info_dev = dm_ws_alloc_t<int>(ctx, batch > 0 ? batch : 1);
  if (!info_dev) return DM_ENOMEM;
  int rc = dm_potrf_batched(ctx, mats, info_dev);
  if (rc == DM_OK) rc = dm_download(ctx, info_host, info_dev, sizeof(int) * batch);
  dm_ws_release(ctx, mark);
  return rc;
}

int dm_ztrsm_left_lower_batched(dm_ctx* ctx, int n, int nrhs, const void* L, int ldl, int64_t strideL, void* B,
                                int ldb, int64_t strideB, int conjtrans, int batch) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, n >= 0 && nrhs >= 0 && batch >= 0 && L && B);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<dm_trsm_problem> ps(batch);
  for (int b = 0; b < batch; ++b)
    ps[b] = dm_trsm_problem{reinterpret_cast<const cplx*>(L) + b * strideL, ldl, n,
                            reinterpret_cast<cplx*>(B) + b * strideB, ldb, nrhs};
  int rc = dm_trsm_left_lower_batched(ctx, ps, conjtrans != 0);
  dm_ws_release(ctx, mark);
  return rc;
}

int dm_jacobi_rows_batched(dm_ctx* ctx, int rows, int cols, int gc0, int gc1, void* Z, int ld, int64_t stride,
                           int batch, double* sigma_dev, int* sweeps_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, rows >= 0 && cols >= 0 && gc0 >= 0 && gc1 <= cols && gc0 <= gc1 && Z && sigma_dev && batch >= 0);
  std::vector<dm_jac_problem> ps(batch);
  for (int b = 0; b < batch; ++b)
    ps[b] = dm_jac_problem{reinterpret_cast<cplx*>(Z) + b * stride, ld, 0, rows, cols, gc0, gc1};
  return dm_jacobi_rows(ctx, ps, sigma_dev, rows > 0 ? rows : 1, sweeps_host);
}

int dm_jacobi_herm_batched(dm_ctx* ctx, int n, void* C, int ldc, int64_t strideC, void* W, int ldw,
                           int64_t strideW, int batch, double* evals_dev, int* sweeps_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, n >= 0 && C && W && evals_dev && batch >= 0);
  for (int b = 0; b < batch; ++b) DM_TRY(dm_set_identity(ctx, reinterpret_cast<cplx*>(W) + b * strideW, ldw, n));
  std::vector<dm_jac_herm_problem> ps(batch);
  for (int b = 0; b < batch; ++b)
    ps[b] = dm_jac_herm_problem{reinterpret_cast<cplx*>(C) + b * strideC, ldc,
                                reinterpret_cast<cplx*>(W) + b * strideW, ldw, n};
  return dm_jacobi_herm(ctx, ps, evals_dev, n > 0 ? n : 1, sweeps_host);
}

int dm_herm_eig_batched(dm_ctx* ctx, int n, void* C, int ldc, int64_t strideC, void* W, int ldw, int64_t strideW,
                        int batch, double* evals_dev) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, n >= 0 && C && W && evals_dev && batch >= 0);
  std::vector<dm_jac_herm_problem> ps(batch);
  for (int b = 0; b < batch; ++b)
    ps[b] = dm_jac_herm_problem{reinterpret_cast<cplx*>(C) + b * strideC, ldc,
                                reinterpret_cast<cplx*>(W) + b * strideW, ldw, n};
  return dm_herm_eig_tridiag(ctx, ps, evals_dev, n > 0 ? n : 1);
}

}  // extern "C"
