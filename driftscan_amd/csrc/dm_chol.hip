// dm_chol.hip — batched blocked Cholesky and triangular solves (gfx950).
//
// These are the zpotrf / zhegst pieces of scipy.linalg.eigh(A, B)
// (drift/core/kltransform.py:89): N = L L^H, C = L^-1 S L^-H, and the
// back-transformation of the eigenvectors with L^-H.
//
// Right-looking blocked Cholesky with NB = 32:
//   potf2  : one workgroup factors the 32x32 diagonal block in LDS        (VALU)
//   trsm   : rows below the block are solved against L_kk^H, 64 rows/WG  (VALU, LDS)
//   update : A22 -= L21 L21^H on the matrix cores (grouped ZGEMM, lower tiles only)
// Every step is one launch for the whole batch of matrices.
//
// The triangular solves are left-looking: for block row k one grouped ZGEMM
// subtracts L[k, <k] X[<k] and a substitution kernel solves the 32x32 diagonal
// system.  Plain substitution (not inverted diagonal blocks) keeps the solve
// backward stable, which matters here: N is conditioned like 1e10.
#include "dm_common.h"
#include "dm_kernels.h"

#include <algorithm>

namespace {

constexpr int NB = 32;

struct chol_desc {
  cplx* A; int ld; int n;
};

// ---- diagonal block factorisation ------------------------------------------------
// grid = batch; 256 threads.  info[b] = first failing 1-based index (sticky).
__global__ __launch_bounds__(256) void potf2_kernel(const chol_desc* __restrict__ ds, int k0, int* __restrict__ info) {
  // One barrier per column: the trailing update reads the UNSCALED column j of T (every thread forms 1 / d_jj itself),
  // the scaled column goes to a second array — no thread waits for a square root or for a scaled column.
  __shared__ cplx T[NB][NB + 1];
  __shared__ cplx Lo[NB][NB + 1];
  const chol_desc d = ds[blockIdx.x];
  if (k0 >= d.n) return;
  if (info[blockIdx.x] != 0) return;
  const int nb = min(NB, d.n - k0);
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    T[r][c] = (r < nb && c < nb && c <= r) ? d.A[(size_t)(k0 + r) * d.ld + k0 + c] : make_double2(0.0, 0.0);
    Lo[r][c] = make_double2(0.0, 0.0);
  }
  int fail = 0;
  for (int j = 0; j < nb; ++j) {
    __syncthreads();   // column j of T is final
    const double djj = T[j][j].x;
    if (!(djj > 0.0) || !isfinite(djj)) {  // every thread reads the same value: a uniform exit
      fail = j + 1;
      break;
    }
    const double inv = 1.0 / djj, invs = 1.0 / sqrt(djj);
    if (tid >= j && tid < nb) Lo[tid][j] = tid == j ? make_double2(sqrt(djj), 0.0) : cscale(T[tid][j], invs);
    // trailing update of the lower triangle: T[r][c] -= T[r][j] conj(T[c][j]) / d_jj,  j < c <= r
    for (int idx = tid; idx < NB * NB; idx += 256) {
      int r = idx / NB, c = idx % NB;
      if (c > j && c <= r && r < nb) T[r][c] = csub(T[r][c], cscale(cmulc(T[r][j], T[c][j]), inv));
    }
  }
  if (fail) {
    if (tid == 0) info[blockIdx.x] = k0 + fail;
    return;
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    if (r < nb && c < nb) {
      cplx v = (c <= r) ? Lo[r][c] : make_double2(0.0, 0.0);
      if (c == r) v.y = 0.0;
      d.A[(size_t)(k0 + r) * d.ld + k0 + c] = v;
    }
  }
}

// ---- panel: rows below the diagonal block, X L_kk^H = A  -> X = A L_kk^-H ----------
// grid = (row tiles of 64, batch).  Each thread-quad handles one row.
__global__ __launch_bounds__(256) void panel_trsm_kernel(const chol_desc* __restrict__ ds, int k0,
                                                         const int* __restrict__ info) {
  __shared__ cplx Lk[NB][NB + 1];
  __shared__ cplx R[64][NB + 1];
  const chol_desc d = ds[blockIdx.y];
  if (k0 + NB >= d.n) return;
  if (info[blockIdx.y] != 0) return;
  const int r0 = k0 + NB + blockIdx.x * 64;
  if (r0 >= d.n) return;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    Lk[r][c] = d.A[(size_t)(k0 + r) * d.ld + k0 + c];
  }
  for (int idx = tid; idx < 64 * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    R[r][c] = (r0 + r < d.n) ? d.A[(size_t)(r0 + r) * d.ld + k0 + c] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  // row x: x[c] = (a[c] - sum_{j<c} x[j] conj(L[c][j])) / L[c][c]; one thread per row, the row in registers and the
  // substitution column-oriented (x[c] leaves all later entries at once: independent multiply-adds, not one chain)
  if (tid < 64) {
    const int r = tid;
    cplx x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = R[r][c];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      x[c] = cscale(x[c], 1.0 / Lk[c][c].x);
#pragma unroll
      for (int j = c + 1; j < NB; ++j) x[j] = csub(x[j], cmulc(x[c], Lk[j][c]));
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) R[r][c] = x[c];
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    if (r0 + r < d.n) d.A[(size_t)(r0 + r) * d.ld + k0 + c] = R[r][c];
  }
}

__global__ void zero_upper_kernel(const chol_desc* __restrict__ ds) {
  const chol_desc d = ds[blockIdx.z];
  const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < d.n && c < d.n && c > r) d.A[(size_t)r * d.ld + c] = make_double2(0.0, 0.0);
}

// ---- diagonal-block substitution for the triangular solves --------------------------
struct trsm_desc {
  const cplx* L; int ldl; int n; cplx* B; int ldb; int nrhs;
};

// Solve L_kk X_k = B_k (forward) or L_kk^H X_k = B_k (backward) for one block row.
// grid = (column tiles of 256, batch); one thread per right-hand-side column.
template <bool CONJTRANS>
__global__ __launch_bounds__(256) void diag_solve_kernel(const trsm_desc* __restrict__ ds, int s, int nblk, int upper_only) {
  __shared__ cplx Lk[NB][NB + 1];
  const trsm_desc d = ds[blockIdx.y];
  int k0;
  if (!CONJTRANS) {
    k0 = s * NB;
  } else {
    // smaller problems start later so that all finish with block row 0 on the last step
    const int pn = (d.n + NB - 1) / NB;
    if (s < nblk - pn) return;
    k0 = (pn - 1 - (s - (nblk - pn))) * NB;
  }
  if (k0 >= d.n || k0 < 0) return;
  const int nb = min(NB, d.n - k0);
  const int col = blockIdx.x * 256 + threadIdx.x;
  // upper_only: only the columns >= the first row of the 64-row super block are wanted (and valid)
  const int first_col = upper_only ? (k0 / (2 * NB)) * (2 * NB) : 0;
  if ((int)(blockIdx.x + 1) * 256 <= first_col) return;   // (uniform over the workgroup)
  for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
    int r = idx / NB, c = idx % NB;
    Lk[r][c] = (r < nb && c < nb) ? d.L[(size_t)(k0 + r) * d.ldl + k0 + c] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  if (col >= d.nrhs || (col / 64 + 1) * 64 <= first_col) return;
  cplx x[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) x[r] = (r < nb) ? d.B[(size_t)(k0 + r) * d.ldb + col] : make_double2(0.0, 0.0);
  // Column-oriented substitution: once x[r] is final it is taken out of every later row at once — 31, 30, ... independent
  // multiply-adds per step instead of one accumulator walking along a row (a chain of 496 dependent complex products,
  // which set the duration of the kernel).  Rows beyond nb are zero in Lk and in x.
  if (!CONJTRANS) {
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (r < nb) {
        x[r] = cscale(x[r], 1.0 / Lk[r][r].x);
#pragma unroll
        for (int j = r + 1; j < NB; ++j) x[j] = csub(x[j], cmul(Lk[j][r], x[r]));
      }
    }
  } else {
    // (L^H)[j][r] = conj(L[r][j]), upper triangular: back substitution
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) {
      const int r = NB - 1 - rr;
      if (r < nb) {
        x[r] = cscale(x[r], 1.0 / Lk[r][r].x);
#pragma unroll
        for (int j = 0; j < r; ++j) x[j] = csub(x[j], cmul(cconj(Lk[r][j]), x[r]));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (r < nb) d.B[(size_t)(k0 + r) * d.ldb + col] = x[r];
}

}  // namespace

int dm_potrf_batched(dm_ctx* ctx, const std::vector<dm_mat>& mats, int* info_dev) {
  const int nbatch = (int)mats.size();
  if (nbatch == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<chol_desc> ds(nbatch);
  int maxn = 0;
  for (int i = 0; i < nbatch; ++i) {
    ds[i] = chol_desc{mats[i].p, mats[i].ld, mats[i].n};
    maxn = std::max(maxn, mats[i].n);
  }
  chol_desc* dd = dm_ws_upload(ctx, ds);
  if (!dd) return DM_ENOMEM;
  DM_HIP(ctx, hipMemsetAsync(info_dev, 0, sizeof(int) * nbatch, ctx->stream));
  // Two-level blocking: the LDS factor/substitution kernels work on 32-wide blocks, but the big
  // trailing update runs once per 64 columns so that the MFMA GEMM sees K = 64 and full tiles.
  // (two passes: the descriptors of all updates are recorded and sent in one copy, then everything is launched)
  dm_gemm_chain chain(ctx);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) DM_TRY(chain.upload());
    for (int k0 = 0; k0 < maxn; k0 += 2 * NB) {
      for (int half = 0; half < 2; ++half) {
        const int kk = k0 + half * NB;
        if (kk >= maxn) break;
        if (!chain.dry) DM_PLAUNCH(ctx, DM_PROF_CHOL, potf2_kernel, dim3(nbatch), dim3(256), 0, ctx->stream, dd, kk, info_dev);
        if (kk + NB >= maxn) break;
        const int rt = (maxn - kk - NB + 63) / 64;
        if (!chain.dry) DM_PLAUNCH(ctx, DM_PROF_CHOL, panel_trsm_kernel, dim3(rt, nbatch), dim3(256), 0, ctx->stream, dd, kk, info_dev);
        if (half == 0) {
          // update only the next 32 columns so that the second factor step sees current data:
          // A[kk+NB:, kk+NB:kk+2NB] -= L[kk+NB:, kk:kk+NB] L[kk+NB:kk+2NB, kk:kk+NB]^H
          DM_TRY(chain.gemm([&](std::vector<dm_gemm_desc>& g) {
            for (int i = 0; i < nbatch; ++i) {
              const int rem = mats[i].n - kk - NB;
              if (rem <= 0) continue;
              cplx* L21 = mats[i].p + (size_t)(kk + NB) * mats[i].ld + kk;
              cplx* A22 = mats[i].p + (size_t)(kk + NB) * mats[i].ld + (kk + NB);
              g.push_back(dm_gemm_make(L21, mats[i].ld, 1, false, L21, 1, mats[i].ld, true, A22, mats[i].ld, rem,
                                       std::min(NB, rem), NB, -1.0, 1.0));
            }
          }));
        }
      }
      // trailing update with both 32-wide panels at once (K = 64), lower tiles only
      const int k2 = k0 + 2 * NB;
      if (k2 < maxn) {
        DM_TRY(chain.gemm([&](std::vector<dm_gemm_desc>& g) {
          for (int i = 0; i < nbatch; ++i) {
            const int rem = mats[i].n - k2;
            if (rem <= 0) continue;
            cplx* L21 = mats[i].p + (size_t)k2 * mats[i].ld + k0;
            cplx* A22 = mats[i].p + (size_t)k2 * mats[i].ld + k2;
            g.push_back(dm_gemm_make(L21, mats[i].ld, 1, false, L21, 1, mats[i].ld, true, A22, mats[i].ld, rem, rem,
                                     2 * NB, -1.0, 1.0, nullptr, DM_GEMM_LOWER));
          }
        }));
      }
    }
  }
  DM_PLAUNCH(ctx, DM_PROF_CHOL, zero_upper_kernel, dim3((maxn + 255) / 256, maxn, nbatch), dim3(256), 0, ctx->stream, dd);
  DM_HIP(ctx, hipGetLastError());
  (void)mark;  // descriptors stay allocated until the caller releases its own mark
  return DM_OK;
}

namespace {
// one pass over the launches of a batched solve: the dry pass of `chain` records the grouped products, the other launches
int trsm_pass(dm_ctx* ctx, const std::vector<dm_trsm_problem>& probs, bool conjtrans, bool upper_only, dm_gemm_chain& chain,
              const trsm_desc* dd) {
  const int nbatch = (int)probs.size();
  int maxn = 0, maxrhs = 0;
  for (int i = 0; i < nbatch; ++i) {
    maxn = std::max(maxn, probs[i].n);
    maxrhs = std::max(maxrhs, probs[i].nrhs);
  }
  const int ct = (maxrhs + 255) / 256;   // column tiles of the substitution kernel
  const int NB2 = 2 * NB;
  const int nblk = (maxn + NB2 - 1) / NB2;   // 64-row super blocks
  const int nblk32 = (maxn + NB - 1) / NB;   // the substitution kernels still count 32-row blocks
  for (int s = 0; s < nblk; ++s) {
    // (1) GEMM updates in the order of the recursive algorithm (solve the first half, update the
    // second half with ONE product, solve the second half), unrolled: before super block u is
    // solved, the 2^ctz(u) blocks starting at u receive the contribution of the 2^ctz(u) blocks
    // solved just before.  Same flops as the left-looking sweep, but half of them sit in one
    // (n/2 x nrhs x n/2) product, a quarter in two (n/4 x nrhs x n/4) products, ... instead of
    // n/64 skinny 64-row products with K up to n.
    DM_TRY(chain.gemm([&](std::vector<dm_gemm_desc>& g) {
    for (int i = 0; i < nbatch; ++i) {
      const dm_trsm_problem& P = probs[i];
      const int pn = (P.n + NB2 - 1) / NB2;
      if (!conjtrans) {
        if (s == 0 || s >= pn) continue;
        const int size = s & -s;                  // 2^ctz(s)
        const int r0 = s * NB2, r1 = std::min((s + size) * NB2, P.n);
        const int c0 = (s - size) * NB2;
        const int j0 = upper_only ? std::min(r0, P.nrhs) : 0;  // first right-hand-side column that matters
        if (P.nrhs - j0 <= 0) continue;
        g.push_back(dm_gemm_make(P.L + (size_t)r0 * P.ldl + c0, P.ldl, 1, false, P.B + (size_t)c0 * P.ldb + j0, P.ldb, 1,
                                 false, P.B + (size_t)r0 * P.ldb + j0, P.ldb, r1 - r0, P.nrhs - j0, size * NB2, -1.0,
                                 1.0));
      } else {
        if (s < nblk - pn) continue;  // smaller problems start later so that all finish together
        const int u = s - (nblk - pn);
        if (u == 0) continue;
        const int kb = pn - 1 - u;
        const int size = u & -u;
        const int t0 = std::max(kb - size + 1, 0) * NB2, t1 = (kb + 1) * NB2;  // target rows
        const int k1 = t1, k2 = std::min((kb + 1 + size) * NB2, P.n);         // rows solved just before
        g.push_back(dm_gemm_make(P.L + (size_t)k1 * P.ldl + t0, 1, P.ldl, true, P.B + (size_t)k1 * P.ldb, P.ldb, 1,
                                 false, P.B + (size_t)t0 * P.ldb, P.ldb, t1 - t0, P.nrhs, k2 - k1, -1.0, 1.0));
      }
    }
    }));
    // (2) inside the super block: substitution on one 32-row half, a small GEMM, the other half
    for (int half = 0; half < 2; ++half) {
      // forward: halves in order 0, 1 ; backward: 1, 0
      const int h = conjtrans ? 1 - half : half;
      // 32-block step index understood by diag_solve_kernel
      int s32;
      if (!conjtrans) s32 = 2 * s + h;
      else s32 = 2 * s + half;  // backward kernel maps its step to the block row itself
      if (chain.dry) {
      } else if (!conjtrans)
        DM_PLAUNCH(ctx, DM_PROF_CHOL, diag_solve_kernel<false>, dim3(ct, nbatch), dim3(256), 0, ctx->stream, dd, s32, nblk32,
                           upper_only ? 1 : 0);
      else
        DM_PLAUNCH(ctx, DM_PROF_CHOL, diag_solve_kernel<true>, dim3(ct, nbatch), dim3(256), 0, ctx->stream, dd, s32, 2 * nblk, 0);
      if (half == 0) {
        DM_TRY(chain.gemm([&](std::vector<dm_gemm_desc>& g2) {
        for (int i = 0; i < nbatch; ++i) {
          const dm_trsm_problem& P = probs[i];
          const int pn = (P.n + NB2 - 1) / NB2;
          if (!conjtrans) {
            const int k0 = s * NB2, k1 = k0 + NB;
            if (k1 >= P.n) continue;
            const int nb = std::min(NB, P.n - k1);
            // B[k1:k1+nb] -= L[k1:k1+nb, k0:k1] X[k0:k1]
            const int j0 = upper_only ? std::min(k0, P.nrhs) : 0;
            if (P.nrhs - j0 <= 0) continue;
            g2.push_back(dm_gemm_make(P.L + (size_t)k1 * P.ldl + k0, P.ldl, 1, false, P.B + (size_t)k0 * P.ldb + j0,
                                      P.ldb, 1, false, P.B + (size_t)k1 * P.ldb + j0, P.ldb, nb, P.nrhs - j0, NB, -1.0,
                                      1.0));
          } else {
            if (s < nblk - pn) continue;
            const int kb = pn - 1 - (s - (nblk - pn));
            const int k0 = kb * NB2, k1 = k0 + NB;
            if (k1 >= P.n) continue;
            const int nb2 = std::min(NB, P.n - k1);
            // B[k0:k1] -= (L[k1:k1+nb2, k0:k1])^H X[k1:k1+nb2]
            g2.push_back(dm_gemm_make(P.L + (size_t)k1 * P.ldl + k0, 1, P.ldl, true, P.B + (size_t)k1 * P.ldb, P.ldb,
                                      1, false, P.B + (size_t)k0 * P.ldb, P.ldb, NB, P.nrhs, nb2, -1.0, 1.0));
          }
        }
        }));
      }
    }
  }
  return DM_OK;
}
}  // namespace

// The products of a solve depend on sizes and addresses only: `build` runs the dry pass (host work that can be done
// while the GPU is still busy with whatever decides if the solve happens at all), `run` sends the descriptors and launches.
int dm_trsm_plan_build(dm_ctx* ctx, const std::vector<dm_trsm_problem>& probs, bool conjtrans, bool upper_only, dm_trsm_plan& plan) {
  // upper_only (forward solves with nrhs == n only): X = L^-1 B is wanted on and above the diagonal only —
  // row block i then needs the rows j < i at columns >= i, which lie above the diagonal too, so
  // every update and substitution simply starts at the first column of its row block (1/3 of the flops).
  if (conjtrans) upper_only = false;
  plan.probs = probs;
  plan.conjtrans = conjtrans;
  plan.upper_only = upper_only;
  plan.chain.reset(new dm_gemm_chain(ctx));
  plan.empty = true;
  for (const auto& P : probs)
    if (P.n > 0 && P.nrhs > 0) plan.empty = false;
  if (plan.empty) return DM_OK;
  return trsm_pass(ctx, plan.probs, conjtrans, upper_only, *plan.chain, nullptr);
}

int dm_trsm_plan_run(dm_ctx* ctx, dm_trsm_plan& plan) {
  if (plan.empty || !plan.chain) return DM_OK;
  const int nbatch = (int)plan.probs.size();
  std::vector<trsm_desc> ds(nbatch);
  for (int i = 0; i < nbatch; ++i)
    ds[i] = trsm_desc{plan.probs[i].L, plan.probs[i].ldl, plan.probs[i].n, plan.probs[i].B, plan.probs[i].ldb, plan.probs[i].nrhs};
  trsm_desc* dd = dm_ws_upload(ctx, ds);
  if (!dd) return DM_ENOMEM;
  DM_TRY(plan.chain->upload());
  DM_TRY(trsm_pass(ctx, plan.probs, plan.conjtrans, plan.upper_only, *plan.chain, dd));
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_trsm_left_lower_batched(dm_ctx* ctx, const std::vector<dm_trsm_problem>& probs, bool conjtrans, bool upper_only) {
  dm_trsm_plan plan;
  DM_TRY(dm_trsm_plan_build(ctx, probs, conjtrans, upper_only, plan));
  return dm_trsm_plan_run(ctx, plan);
}
