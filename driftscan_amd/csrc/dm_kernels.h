// dm_kernels.h — internal (non-ABI) interfaces between the translation units of
// libdriftmi.  All launches go to ctx->stream; nothing here synchronises unless
// it says so.
#pragma once

#include "dm_common.h"

#include <deque>
#include <memory>
#include <functional>

// ---------------------------------------------------------------------------
// grouped ZGEMM (dm_gemm.hip)
// ---------------------------------------------------------------------------
enum {
  DM_GEMM_CONJ_A = 1,  // conjugate the elements of A as they are read
  DM_GEMM_CONJ_B = 2,  // conjugate the elements of B as they are read
  DM_GEMM_B_REAL = 4,  // B points at real doubles (complex x real product)
  DM_GEMM_LOWER = 8,   // only compute 64x64 tiles on or below the block diagonal
  DM_GEMM_ALL_REAL = 16,  // A, B and C are arrays of doubles (strides in doubles)
  DM_GEMM_UPPER = 32,  // only compute 64x64 tiles on or above the block diagonal
  DM_GEMM_UPPER128 = 128,  // with DM_GEMM_UPPER: also the tile left of an odd diagonal tile (whole 128 x 128 diagonal blocks)
  DM_GEMM_B_GATHER = 64,  // columns of B are gathered through desc.bgather and carry their own K weights
};

struct dm_gemm_desc {
  const void* A;         // cplx, viewed as (M x K) through (rsA, csA)
  const void* B;         // cplx (or double with DM_GEMM_B_REAL), viewed as (K x N) through (rsB, csB)
  void* C;               // cplx, row-major, leading dimension ldc
  const double* kscale;  // optional real weights over K (applied to A), or nullptr
  const int2* bgather;   // DM_GEMM_B_GATHER: per column n of B, (x = element offset of its K-run relative to B,
                         //   y = offset of its weight run relative to kscale): B(k, n) = B[x + k rsB] * kscale[y + k]
  int M, N, K;
  int rsA, csA, rsB, csB, ldc;
  int csc;               // element stride between the columns of C (1 unless the caller interleaves outputs; complex kernels only)
  int flags;
  double alpha, beta;
  double alpha_im;  // imaginary part of alpha (0 for the usual real scaling)
};

struct dm_gemm_tile {
  int desc, tm, tn;
};

int dm_gemm_grouped_launch(dm_ctx* ctx, const std::vector<dm_gemm_desc>& descs);

// The same in two steps, for chains of launches whose host images can travel in one copy (dm_gemm.hip).
struct dm_gemm_plan {
  std::vector<char> blob;   // descriptors + tile lists as the kernels read them
  size_t desc_bytes = 0, n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  double fl_c = 0.0, fl_r = 0.0, fl_d = 0.0, fl_g = 0.0;
  bool use4 = true, deep = false, wide_r = false;
  int log_kmin = 0, log_kmax = 0, log_mmax = 0, log_nmax = 0, log_rmw = 0;
  size_t log_ndesc = 0;
  double log_full = 0.0;
};
int dm_gemm_plan_build(const std::vector<dm_gemm_desc>& descs, dm_gemm_plan& plan);
int dm_gemm_plans_upload(dm_ctx* ctx, const std::vector<const dm_gemm_plan*>& plans, std::vector<const char*>& dev);
int dm_gemm_plan_run(dm_ctx* ctx, const dm_gemm_plan& plan, const char* dblob);

// A chain of dependent launches (grouped products with other kernels in between) whose descriptors depend on sizes only:
// the driver's loops run twice — a dry pass that records the plan of every product, one staged copy of all of them, then
// the pass that launches.  `gemm(fill)` calls `fill(descs)` in the dry pass only.
struct dm_gemm_chain {
  dm_ctx* ctx;
  bool dry = true;
  std::deque<dm_gemm_plan> plans;
  std::vector<const char*> dev;
  size_t cur = 0;
  explicit dm_gemm_chain(dm_ctx* c) : ctx(c) {}
  template <typename F>
  int gemm(F&& fill) {
    if (dry) {
      std::vector<dm_gemm_desc> g;
      fill(g);
      plans.emplace_back();
      return dm_gemm_plan_build(g, plans.back());
    }
    const size_t i = cur++;
    return dm_gemm_plan_run(ctx, plans[i], dev[i]);
  }
  int upload() {
    std::vector<const dm_gemm_plan*> pp;
    for (const auto& pl : plans) pp.push_back(&pl);
    dry = false;
    cur = 0;
    return dm_gemm_plans_upload(ctx, pp, dev);
  }
};

static inline dm_gemm_desc dm_gemm_make(const cplx* A, int rsA, int csA, bool conjA, const void* B, int rsB, int csB,
                                        bool conjB, cplx* C, int ldc, int M, int N, int K, double alpha = 1.0,
                                        double beta = 0.0, const double* kscale = nullptr, int extra_flags = 0) {
  dm_gemm_desc d;
  d.A = A; d.B = B; d.C = C; d.kscale = kscale;
  d.M = M; d.N = N; d.K = K;
  d.rsA = rsA; d.csA = csA; d.rsB = rsB; d.csB = csB; d.ldc = ldc;
  d.csc = 1;
  d.flags = (conjA ? DM_GEMM_CONJ_A : 0) | (conjB ? DM_GEMM_CONJ_B : 0) | extra_flags;
  d.alpha = alpha; d.beta = beta; d.alpha_im = 0.0;
  d.bgather = nullptr;
  return d;
}

// ---------------------------------------------------------------------------
// block Jacobi engines (dm_jacobi.hip)
// ---------------------------------------------------------------------------
// One "problem" = one matrix whose rows [row0, row0+nrows) are orthogonalised
// (one-sided) with respect to the columns [gc0, gc1); the unitary row mixing is
// applied to all ncols columns, so passengers (accumulated U^H, projected beams)
// ride along.  For the two-sided (Hermitian) mode the matrix is nrows x nrows in
// columns [gc0, gc0+nrows) and both rows and columns are transformed.
struct dm_jac_problem {
  cplx* Z;        // base of the row-major matrix
  int ld;         // leading dimension (elements)
  int row0;       // first participating row
  int nrows;      // participating rows
  int ncols;      // columns carried along by the row mixing
  int gc0, gc1;   // Gram column range (one-sided) / Hermitian block origin (two-sided)
};

// Householder columns whose squared norm is below this are left alone (tau = 0): sqrt(|alpha|^2 + |x|^2) of such a
// column underflows towards 0 and tau = (beta - alpha) / beta turns into 0/0.  Matrices with exactly zero rows get there:
// the rounding residue of one reflector is reflected again by the next sweep of the bulge chase (1e-16, 1e-32, ... of
// the norm of the matrix), and after ten sweeps the residue of the residue is below 1e-160.  LAPACK's zlarfg rescales
// instead; here the column is at most 1e-145 in magnitude and dropping it is a backward error far below rounding.
constexpr double DM_REFL_TINY = 1e-290;

// Orthogonalise rows (one-sided).  On return sigma[p][0..nrows) holds the row
// norms over the Gram columns, rows sorted by descending norm (rows physically
// permuted).  `sigma` is a device array with `sigma_stride` doubles per problem.
// Synchronises the stream (sweep control needs the convergence flags).
// Options: `unconverged` — the caller knows the rows are not orthogonal yet, the measuring pass is skipped;
// `drop_below` — rows whose norm ends up below drop_below * (largest row norm) after the preconditioner are
// of no interest to the caller (SVD1 discards everything under 1e-10 sigma_0, beamtransfer.py:826): they are
// kept out of the sweeps and of the deeper preconditioner levels (their norms are still reported; by Weyl's
// inequality leaving them out moves the other singular values by at most their Frobenius norm).
struct dm_jac_rows_opts {
  bool unconverged = false;
  double drop_below = 0.0;
  bool one_stage_eig = false;   // the Gram eigenproblems of the preconditioner levels on the one-stage tridiagonalisation
  // > 0: the caller only splits the rows at subspace_cut * (largest row norm) and keeps one side as a SUBSPACE (the image
  // of SVD1, the null space of SVD2: whatever orthonormal basis the next phase is handed, its products are the same).
  // The preconditioner then stops at the first level whose Gram eigenvalues resolve the cut (a level with largest
  // eigenvalue e_top places a row norm s to eps e_top / s^2 relative), and no Jacobi sweep follows: the rows are
  // unitary mixtures of the input rows either way, sorted by norm; what a converged SVD would add is the order
  // INSIDE the two sides.
  double subspace_cut = 0.0;
  // ... how far above the cut the LAST level may begin (in units of the cut): a pair of rows (i, j) on the two sides of the
  // cut comes out of a level whose largest row norm is s_top with a residual angle eps s_top^2 / |s_i^2 - s_j^2|, and
  // what leaks across the cut is that angle times the row.  SVD2 (cut 1e-4, rows of all sizes next to it: 5e-9 of
  // sigma_max in the final spectrum when the split is left to the first level) asks for 100: the level that places the
  // cut holds the rows below 1e-2 of the largest, angles <= 1e-12 / (relative gap) as from a converged SVD; SVD1 (cut
  // 1e-10: whatever leaks is of that size itself) takes the regular level boundaries.
  double subspace_margin = 3.2e5;
};
int dm_jacobi_rows(dm_ctx* ctx, const std::vector<dm_jac_problem>& probs, double* sigma, int sigma_stride,
                   int* sweeps_out = nullptr, const dm_jac_rows_opts* opts = nullptr);

// Diagonalise Hermitian matrices (two-sided): C <- W C W^H (diagonal), with the
// unitary W (rows) accumulated into `W` (nrows x nrows, row-major, ld = ldw),
// which must be initialised to the identity by the caller (or any unitary to
// compose with).  evals[p][0..n) = diagonal, NOT sorted.  Synchronises.
struct dm_jac_herm_problem {
  cplx* C; int ldc;
  cplx* W; int ldw;
  int n;
};
int dm_jacobi_herm(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* evals, int evals_stride,
                   int* sweeps_out = nullptr);

// Same contract as dm_jacobi_herm (C destroyed, W rows = eigenvectors^H, evals unsorted;
// W need not be initialised) through Householder tridiagonalisation + implicit QL +
// compact-WY back-transformation (dm_tridiag.hip).  Returns > 0 if QL fails.  Synchronises.
// Optional eigenvector selection: after the tridiagonal eigen-solve the eigenvalues of problem p
// (natural, unsorted order) are handed to `pick`, which returns the indices of the eigenvectors to
// back-transform in the order the rows of W shall have.  Only those rows of W are written
// (rows [0, nsel[p])); the back-transformation and everything downstream then cost nsel/n of the
// full amount.  `evals` still receives ALL eigenvalues in natural order.  `pick` is called from several host threads
// at once, each call with a different p: it may only touch state that belongs to problem p.
struct dm_eig_select {
  std::function<void(int p, const double* ev, int n, std::vector<int>& cols)> pick;
  std::vector<int> nsel;  // out
};
int dm_herm_eig_tridiag(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* evals, int evals_stride,
                        dm_eig_select* sel = nullptr);

// Permute the rows [row0, row0+nrows) x [0, ncols) of each problem so that the
// device keys (key_stride doubles per problem) end up sorted; keys are sorted too.
int dm_sort_rows_by_key(dm_ctx* ctx, const std::vector<dm_jac_problem>& probs, double* key, int key_stride,
                        bool descending);

// ---------------------------------------------------------------------------
// batched Cholesky / triangular solves (dm_chol.hip)
// ---------------------------------------------------------------------------
struct dm_mat {
  cplx* p; int ld; int n;
};
// In-place lower Cholesky of Hermitian matrices (lower triangle referenced;
// strictly-upper part is zeroed on exit).  info[i] (device ints) = 0 or the
// 1-based order of the first non-positive leading minor, as LAPACK zpotrf.
int dm_potrf_batched(dm_ctx* ctx, const std::vector<dm_mat>& mats, int* info_dev);
// Solve L X = B (conjtrans=false) or L^H X = B (conjtrans=true) in place in B
// (n x nrhs, row-major, ldb).
struct dm_trsm_problem {
  const cplx* L; int ldl; int n;
  cplx* B; int ldb; int nrhs;
};
int dm_trsm_left_lower_batched(dm_ctx* ctx, const std::vector<dm_trsm_problem>& probs, bool conjtrans, bool upper_only = false);
// the same in two steps: build = host-only dry pass over the chain of launches, run = one descriptor copy + the launches
struct dm_gemm_chain;
struct dm_trsm_plan {
  std::vector<dm_trsm_problem> probs;
  bool conjtrans = false, upper_only = false, empty = true;
  std::shared_ptr<dm_gemm_chain> chain;
};
int dm_trsm_plan_build(dm_ctx* ctx, const std::vector<dm_trsm_problem>& probs, bool conjtrans, bool upper_only, dm_trsm_plan& plan);
int dm_trsm_plan_run(dm_ctx* ctx, dm_trsm_plan& plan);

// ---------------------------------------------------------------------------
// small utility kernels (dm_util.hip)
// ---------------------------------------------------------------------------
int dm_conj_transpose(dm_ctx* ctx, const cplx* src, int lds, cplx* dst, int ldd, int rows, int cols);
int dm_set_identity(dm_ctx* ctx, cplx* a, int ld, int n);
int dm_hermitize(dm_ctx* ctx, cplx* a, int ld, int n);  // a <- (a + a^H)/2, real diagonal
int dm_fill_zero(dm_ctx* ctx, void* p, size_t bytes);
// one-launch versions for lists of matrices / buffers
struct dm_tdesc { const cplx* src; int lds; cplx* dst; int ldd; int rows; int cols; };
struct dm_cdesc { const void* src; void* dst; size_t bytes; };
int dm_conj_transpose_batched(dm_ctx* ctx, const std::vector<dm_tdesc>& v);
int dm_set_identity_batched(dm_ctx* ctx, const std::vector<dm_mat>& v);
int dm_hermitize_batched(dm_ctx* ctx, const std::vector<dm_mat>& v);
int dm_copy_batched(dm_ctx* ctx, const std::vector<dm_cdesc>& v);
