// dm_tridiag.hip — batched Hermitian eigensolver: Householder tridiagonalisation,
// implicit-shift QL on the real tridiagonal, and compact-WY back-transformation.
//
// This is the zheevd step of scipy.linalg.eigh(A, B) (drift/core/kltransform.py:89)
// for every m-block at once.  All matrices of the batch advance in lock-step, so a
// launch always carries (#matrices x #row tiles) workgroups:
//
//   T1  tridiagonalisation (LAPACK zhetrd/zlatrd recurrences, panels of 64 reflectors, upper triangle only):
//         trd_symv  p = A v reading each stored element once + Householder scalars  (HBM-bound: the roofline of T1)
//         trd_wx    w = tau p - (tau/2)(p^H v) v and the next column, one pass over the panel
//         her2k     A -= [V W][W V]^H once per panel, K = 128                      (grouped ZGEMM, MFMA)
//   T2  ql_kernel   implicit QL with Wilkinson shifts, one wave (lane 0) per matrix; the
//                   Givens rotations are recorded sweep by sweep instead of being applied
//   T3  rot_apply   the recorded rotations are applied to Z = I, one thread per ROW of Z,
//                   sixteen consecutive sweeps pipelined through a register window so
//                   that each pass over the row does 16 sweeps of work (HBM-bound / 16)
//   T4  back-transformation X = H_0 ... H_{n-2} Z in compact-WY blocks             (grouped ZGEMM, MFMA)
//
// Flop count ~ (16/3 + 8 + ...) n^3 against the ~1400 n^3 the two-sided block-Jacobi
// solver needs on the graded spectra of KL problems; accuracy is LAPACK's (backward
// stable, |d lambda| ~ eps |lambda_max|).
#include "dm_common.h"
#include "dm_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <thread>

// The panel width is a compile-time constant of the kernels (LDS arrays, T-factor layout): the body is
// compiled twice.  Narrow panels halve the traffic on the panel vectors V, W (a third of what trd_symv and
// trd_wx move once the trailing matrices are a few hundred rows) and win up to n ~ 2000 — the Gram matrices
// of the SVD preconditioner, the KL problems of config 2: 864 x 512 matrices 0.465 -> 0.409 s; wide panels
// keep the her2k updates at K = 128 where the MFMA products dominate (n = 12 000: 2.23 s against 2.34 s).
#define DM_TNB 32
#define DM_TRD_NS dm_trd32
#include "dm_tridiag_impl.h"
#undef DM_TNB
#undef DM_TRD_NS
#define DM_TNB 64
#define DM_TRD_NS dm_trd64
#include "dm_tridiag_impl.h"
#undef DM_TNB
#undef DM_TRD_NS

int dm_herm_eig_tridiag(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* evals, int evals_stride,
                        dm_eig_select* sel) {
  int maxn = 0;
  for (const auto& p : probs) maxn = std::max(maxn, p.n);
  int width = maxn <= 2048 ? 32 : 64;
  if (getenv("DM_TRD_SIZES")) {  // debugging aid: the batch composition
    fprintf(stderr, "[dm_herm_eig_tridiag] %zu problems, n =", probs.size());
    for (const auto& p : probs) fprintf(stderr, " %d", p.n);
    fprintf(stderr, "\n");
  }
  // the two-stage reduction (dm_sbr_impl.h) lives in the 32-wide instantiation: its bandwidth is the panel width
  // (the policy in herm_eig_tridiag decides; these are the batches it can say yes to)
  {
    size_t totn = 0;
    for (const auto& p : probs) totn += p.n;
    const char* e = getenv("DM_TRD_TWOSTAGE");
    const int mode = ctx->trd_mode_override >= 0 ? ctx->trd_mode_override : (e ? atoi(e) : -1);
    static const bool ts_mid = !getenv("DM_TRD_TS_MID") || atoi(getenv("DM_TRD_TS_MID")) != 0;
    if (mode == 1 || (mode != 0 && ((maxn >= 3500 && totn >= 24000) || (ts_mid && maxn >= 2400 && totn >= 48000) || maxn >= 14000)))
      width = 32;
  }
  if (const char* e = getenv("DM_TRD_PANEL")) width = atoi(e) == 64 ? 64 : 32;
  return width == 32 ? dm_trd32::herm_eig_tridiag(ctx, probs, evals, evals_stride, sel)
                     : dm_trd64::herm_eig_tridiag(ctx, probs, evals, evals_stride, sel);
}
