// dm_jacobi.hip — batched block-Jacobi engines on gfx950.
//
// Two drivers share three kernels:
//
//   one-sided ("rows")  : SVD of a wide/tall matrix — the rows of Z are unitarily
//       mixed until mutually orthogonal over a chosen set of columns, every other
//       column rides along as a passenger.  Replaces scipy.linalg.svd in
//       matrix_image / matrix_nullspace (drift/core/beamtransfer.py:74, :113) and,
//       through the passenger trick, also the projection GEMMs that follow them
//       (beamtransfer.py:831, :850-851, :866, :877).
//   two-sided ("herm")  : eigendecomposition of a Hermitian matrix; replaces the
//       zheevd step of scipy.linalg.eigh(A, B) (drift/core/kltransform.py:89).
//
// Structure of one round (all matrices of the batch advance in lock-step):
//   1. jac_gram      G = X X^H for every disjoint pair of 32-row blocks      (MFMA)
//      (two-sided: G is read straight from the diagonal blocks of C)
//   2. jac_inner     2-sided cyclic Jacobi on each 64x64 G inside LDS -> Q    (VALU+LDS)
//   3. jac_apply     rows(pair) <- Q^H rows(pair) over all columns           (MFMA)
// and a tournament of nb-1 rounds makes a sweep.  Because every m-block and every
// frequency is an independent problem, one launch carries (#pairs x #problems)
// workgroups — that is what fills 256 CUs with matrices this small.
//
// Accuracy: the inner solver is a Jacobi method too, so graded Gram blocks keep
// their relative accuracy (an eigh-style inner solver does not: see DESIGN.md §5).
#include "dm_common.h"
#include "dm_kernels.h"

#include <algorithm>
#include <map>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int JB = 32;       // rows per block
constexpr int JP = 64;       // rows per pair
constexpr int GP = 65;       // LDS pitch (complex elements) of the 64x64 inner matrices
constexpr int JNT = 1024;    // threads of jac_inner: 2 waves/SIMD hide the LDS latency of the rotation updates (1 WG/CU fits)
constexpr int QP = 80;       // LDS pitch (doubles) of the Q^H planes in jac_apply
constexpr int XP = 17;       // LDS pitch (doubles) of the gram staging planes

struct jac_item {
  cplx* Z;      // matrix base
  int ld;       // leading dimension
  int ra, na;   // first row / valid rows of block A
  int rb, nb;   // first row / valid rows of block B (nb == 0: single block)
  int c0, c1;   // columns transformed by jac_apply: [c0, c1)
  int g0, g1;   // Gram columns [g0, g1)  (two-sided: g0 = column origin of the Hermitian block)
  int prob;     // problem index (convergence bookkeeping)
  int q;        // slot in the Q / G buffers
  int fa, fb;   // one-sided mode: slots of the two blocks in the "rows mutually orthogonal" flags (jac_gram)
};

__device__ __forceinline__ int item_row(const jac_item& it, int k) {
  // k in [0,64): local row -> global row or -1
  if (k < JB) return (k < it.na) ? it.ra + k : -1;
  k -= JB;
  return (k < it.nb) ? it.rb + k : -1;
}

// ---------------------------------------------------------------------------
// 1. Gram of a row-block pair (one-sided mode)
// ---------------------------------------------------------------------------
// G = X X^H of the 64 rows of a pair is Hermitian, and most of it is known before it is computed:
//   FULL   the 10 tiles (16 x 16) on and above the diagonal, dealt 3/3/2/2 to the four waves, the lower ones mirrored —
//          taken the first time a block is met in a sweep (`blk_ok` clear);
//   CROSS  once both blocks have been through a pair solve in this sweep their rows are mutually orthogonal to the inner
//          solver's tolerance — the two diagonal 32 x 32 blocks of G are diagonal matrices, the squared row norms
//          (recomputed from the rows: fp64 sums on the vector ALU while the tiles are staged) — and only the 32 x 32
//          block A B^H is formed: one tile per wave, a quarter of the matrix-core work of a full Gram.
// `blk_ok[fa]`, `blk_ok[fb]`: set by jac_inner when it has solved (or found solved) a pair holding the block; cleared by
// the driver at the start of every sweep, so each block's own Gram is re-measured from the rows once per sweep.
__device__ __constant__ unsigned char JG_TI[4][3] = {{0, 0, 0}, {0, 1, 1}, {1, 2, 0}, {2, 3, 0}};
__device__ __constant__ unsigned char JG_TJ[4][3] = {{0, 1, 2}, {3, 1, 2}, {3, 2, 0}, {3, 3, 0}};
__device__ __constant__ unsigned char JG_NT[4] = {3, 3, 2, 2};

__global__ __launch_bounds__(256) void jac_gram_kernel(const jac_item* __restrict__ items,
                                                       const int* __restrict__ active, cplx* __restrict__ Gbuf,
                                                       const int* __restrict__ blk_ok, int allow_cross,
                                                       unsigned long long* __restrict__ flopctr) {
  __shared__ double Xre[JP * XP], Xim[JP * XP];
  __shared__ double nrm2[JP];
  const jac_item it = items[blockIdx.x];
  if (!active[it.prob]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fk = lane >> 4;
  const bool cross = allow_cross && it.nb > 0 && blk_ok[it.fa] && blk_ok[it.fb];   // uniform over the workgroup
  if (flopctr && tid == 0) {
    const unsigned long long kk = (unsigned long long)(it.g1 - it.g0), nr = it.na + it.nb;
    // algorithmic flops: the Hermitian Gram on and above the diagonal, or the cross block plus the row norms
    atomicAdd(flopctr, cross ? (8ull * it.na * it.nb + 4ull * nr) * kk : 4ull * nr * (nr + 1ull) * kk);
  }

  // tiles of this wave
  int nt, ti[3], tj[3];
  if (cross) {
    nt = 1;
    ti[0] = wave >> 1; tj[0] = 2 + (wave & 1);
    ti[1] = ti[2] = tj[1] = tj[2] = 0;
  } else {
    nt = JG_NT[wave];
#pragma unroll
    for (int t = 0; t < 3; ++t) { ti[t] = JG_TI[wave][t]; tj[t] = JG_TJ[wave][t]; }
  }
  dm_f64x4 gre[3], gim[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    gre[t] = dm_f64x4{0, 0, 0, 0};
    gim[t] = dm_f64x4{0, 0, 0, 0};
  }

  // staging map: 64 rows x 16 cols, k fastest (each row segment = 256 contiguous bytes)
  int srow[4], scol[4], grow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + 256 * i;
    scol[i] = idx & 15;
    srow[i] = idx >> 4;
    grow[i] = item_row(it, srow[i]);
  }
  cplx r[4];
  double nacc[4] = {0.0, 0.0, 0.0, 0.0};
  // global_load (not flat: a flat load also counts on LGKM, so every LDS wait would drain the prefetch) from a clamped,
  // always valid address; rows and columns outside the item read as zero
  // (the loaded values are masked when they are staged, a stage later: a select right behind the load would wait for it)
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = it.g0 + k0 + scol[i];
      r[i] = dm_ldg(it.Z, (size_t)(grow[i] >= 0 ? grow[i] : it.ra) * it.ld + min(c, it.g1 - 1));
    }
  };
  const int nk = (it.g1 - it.g0 + 15) / 16;
  if (nk > 0) load(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = grow[i] >= 0 && it.g0 + kt * 16 + scol[i] < it.g1;
      const double xr = ok ? r[i].x : 0.0, xi = ok ? r[i].y : 0.0;
      Xre[srow[i] * XP + scol[i]] = xr;
      Xim[srow[i] * XP + scol[i]] = xi;
      nacc[i] = fma(xr, xr, fma(xi, xi, nacc[i]));
    }
    __syncthreads();
    if (kt + 1 < nk) load((kt + 1) * 16);
    // one accumulator at a time, its eight MFMAs of this stage back to back: v_mfma_f64_16x16x4_f64 runs at 69 cycles in
    // chains of 8 on one accumulator against 83 in chains of 2 and 105 with the accumulators in rotation
    // (scratch/mfma_peak3.hip: SrcC forwarded from the previous MFMA instead of read from the register file)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < nt) {   // wave-uniform
        double a_re[4], a_im[4], x_re[4], x_im[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ra = (ti[t] * 16 + fi) * XP + kk * 4 + fk;
          const int rb = (tj[t] * 16 + fi) * XP + kk * 4 + fk;
          a_re[kk] = Xre[ra]; a_im[kk] = Xim[ra];
          x_re[kk] = Xre[rb]; x_im[kk] = Xim[rb];
        }
        // G[i][j] += a_i * conj(x_j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          gre[t] = dm_mfma(a_re[kk], x_re[kk], gre[t]);
          gre[t] = dm_mfma(a_im[kk], x_im[kk], gre[t]);
        }
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler interleaves independent chains otherwise)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          gim[t] = dm_mfma(a_im[kk], x_re[kk], gim[t]);
          gim[t] = dm_mfma(-a_re[kk], x_im[kk], gim[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  cplx* G = Gbuf + (size_t)it.q * JP * JP;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t < nt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = ti[t] * 16 + (lane >> 4) + 4 * rr;
        const int col = tj[t] * 16 + (lane & 15);
        const cplx v = make_double2(gre[t][rr], gim[t][rr]);
        if (ti[t] != tj[t]) {
          G[row * JP + col] = v;
          G[col * JP + row] = make_double2(v.x, -v.y);
        } else if (row <= col) {   // diagonal tile: the upper part and its mirror (one writer per entry)
          G[row * JP + col] = row == col ? make_double2(v.x, 0.0) : v;
          if (row != col) G[col * JP + row] = make_double2(v.x, -v.y);
        }
      }
    }
  }
  if (cross) {
    // the two diagonal 32 x 32 blocks: squared row norms on the diagonal, zero elsewhere
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double v = nacc[i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (scol[i] == 0) nrm2[srow[i]] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < JP * JP / 2; idx += 256) {
      const int blk = idx >> 10, rr = (idx >> 5) & 31, cc = idx & 31;   // 2 blocks x 32 x 32
      const int row = blk * 32 + rr, col = blk * 32 + cc;
      G[row * JP + col] = make_double2(rr == cc ? nrm2[row] : 0.0, 0.0);
    }
  }
}

// ---------------------------------------------------------------------------
// 2. inner solver: two-sided cyclic Jacobi on a 64x64 Hermitian matrix in LDS
// ---------------------------------------------------------------------------
// offmax[prob] receives (atomicMax on the bit pattern of a non-negative double)
// the largest relative off-diagonal |g_ij| / sqrt(g_ii g_jj) seen *before* the
// solve: the sweep-level convergence measure.
template <bool HERM>
__global__ __launch_bounds__(JNT) void jac_inner_kernel(const jac_item* __restrict__ items,
                                                        const int* __restrict__ active,
                                                        const double* __restrict__ absfloor_p,
                                                        const cplx* __restrict__ Gbuf, cplx* __restrict__ Qbuf,
                                                        unsigned long long* __restrict__ offmax,
                                                        int* __restrict__ skip, double tol_outer,
                                                        double tol_inner, int measure_only,
                                                        int* __restrict__ blk_ok) {
  extern __shared__ __align__(16) unsigned char smem[];
  cplx* G = reinterpret_cast<cplx*>(smem);
  cplx* Q = G + JP * GP;
  double* rc = reinterpret_cast<double*>(Q + JP * GP);  // [32] cos
  cplx* rs = reinterpret_cast<cplx*>(rc + 32);           // [32] sin * phase
  int* ctl = reinterpret_cast<int*>(rs + 32);            // [0] rotations this sweep, [1..32] pair active
  double* red = reinterpret_cast<double*>(ctl + 40);     // [4] reduction scratch

  const jac_item it = items[blockIdx.x];
  if (!active[it.prob]) return;
  const int tid = threadIdx.x;
  const double absfloor = absfloor_p ? absfloor_p[it.prob] : 0.0;

  // ---- load G (Hermitian by construction), Q = I
  for (int idx = tid; idx < JP * JP; idx += JNT) {
    int r = idx >> 6, c = idx & 63;
    cplx v;
    if (HERM) {
      int gr = item_row(it, r), gc = item_row(it, c);
      // two-sided: rows/cols index the same Hermitian matrix (column origin g0 - row origin folded by caller)
      v = (gr >= 0 && gc >= 0) ? it.Z[(size_t)gr * it.ld + (it.g0 + gc)] : make_double2(0.0, 0.0);
    } else {
      v = Gbuf[(size_t)it.q * JP * JP + idx];
    }
    G[r * GP + c] = v;
    Q[r * GP + c] = make_double2(r == c ? 1.0 : 0.0, 0.0);
  }
  __syncthreads();
  if (HERM) {
    // enforce exact Hermitian symmetry from the lower triangle
    for (int idx = tid; idx < JP * JP; idx += JNT) {
      int r = idx >> 6, c = idx & 63;
      if (r < c) {
        cplx v = G[c * GP + r];
        G[r * GP + c] = make_double2(v.x, -v.y);
      } else if (r == c) {
        G[r * GP + c].y = 0.0;
      }
    }
    __syncthreads();
  }

  // ---- convergence measure before the solve
  double mo = 0.0;
  for (int idx = tid; idx < JP * JP; idx += JNT) {
    int r = idx >> 6, c = idx & 63;
    if (r < c) {
      cplx g = G[r * GP + c];
      double ag = sqrt(g.x * g.x + g.y * g.y);
      double dd = fabs(G[r * GP + r].x * G[c * GP + c].x);
      if (ag > absfloor && dd > 0.0) mo = fmax(mo, ag / sqrt(dd));
    }
  }
  mo = dm_wave_max(mo);
  if ((tid & 63) == 0) red[tid >> 6] = mo;
  __syncthreads();
  {
    double m2 = red[0];
    for (int w = 1; w < JNT / 64; ++w) m2 = fmax(m2, red[w]);
    mo = m2;
  }
  if (tid == 0) {
    atomicMax(&offmax[it.prob], (unsigned long long)__double_as_longlong(mo));
    skip[it.q] = (mo <= tol_outer || measure_only) ? 1 : 0;
    if (!HERM && blk_ok && !measure_only) {
      // whether the pair is solved below or was found orthogonal already: after this item the rows of each of its
      // blocks are mutually orthogonal to tolerance (what jac_gram's CROSS mode relies on)
      blk_ok[it.fa] = 1;
      if (it.nb > 0) blk_ok[it.fb] = 1;
    }
  }
  if (mo <= tol_outer || measure_only) return;  // uniform across the block

  // ---- cyclic Jacobi, 63 parallel steps of 32 disjoint rotations per sweep
  for (int sweep = 0; sweep < 24; ++sweep) {
    if (tid == 0) { ctl[0] = 0; ctl[33] = 0; }
    __syncthreads();
    for (int step = 0; step < 63; ++step) {
      // phase A: rotation parameters
      if (tid < 32) {
        int p, q;
        if (tid == 0) { p = 63; q = step; }
        else { p = (step + tid) % 63; q = (step + 63 - tid) % 63; }
        cplx g = G[p * GP + q];
        double a = G[p * GP + p].x, b = G[q * GP + q].x;
        double ag = sqrt(g.x * g.x + g.y * g.y);
        int act = 0;
        double c = 1.0;
        cplx sp = make_double2(0.0, 0.0);
        // (A norm-wise test |g|^2 <= floor max(a, b) — LAPACK's level of accuracy — was tried here and is NOT
        // enough: beam_svd . pinv(beam_svd) = I needs the small rows to relative accuracy, kappa = 1/svcut.)
        if (ag > 0.0 && ag > absfloor && ag > tol_inner * sqrt(fabs(a * b))) {
          double zeta = (b - a) / (2.0 * ag);
          double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          c = 1.0 / sqrt(1.0 + t * t);
          double s = c * t;
          sp = make_double2(s * g.x / ag, s * g.y / ag);  // s * phase
          act = 1;
        }
        rc[tid] = c;
        rs[tid] = sp;
        ctl[1 + tid] = act;
        if (act) ctl[0] = 1;  // benign race: all writers store 1
      }
      __syncthreads();
      // phase B: G <- J^H G J and Q <- Q J for the 32 disjoint rotations of this step, with
      //   J restricted to pair t = [[c, sp], [-conj(sp), c]].  G is updated by 2x2 blocks:
      //   G'[A][B] = J_A^H G[A][B] J_B touches only the block itself, so the pairs A <= B are
      //   transformed once and mirrored (Hermitian) — a quarter of the LDS reads of separate
      //   column and row passes, and one barrier less per step.
      auto pair_of = [&](int t, int& pp, int& qq) {
        if (t == 0) { pp = 63; qq = step; }
        else { pp = (step + t) % 63; qq = (step + 63 - t) % 63; }
      };
#pragma unroll 2
      for (int i = 0; i < 1024 / JNT; ++i) {
        const int u = tid + JNT * i;
        const int ta = u >> 5, tb = u & 31;
        if (ta > tb) continue;
        const int aa = ctl[1 + ta], ab = ctl[1 + tb];
        if (!aa && !ab) continue;
        int pa, qa, pb, qb;
        pair_of(ta, pa, qa);
        pair_of(tb, pb, qb);
        const double ca = rc[ta], cb = rc[tb];
        const cplx sa = rs[ta], sb = rs[tb];
        if (ta == tb) {
          // diagonal block of an active pair: the rotation annihilates the off-diagonal element
          const cplx gpp = G[pa * GP + pa], gqq = G[qa * GP + qa], gpq = G[pa * GP + qa];
          // (J^H X J)_pp = c^2 gpp + |s|^2 gqq - 2 c Re(conj(sp) ... ) — formed through the two half steps
          // X1 = X J (columns), X2 = J^H X1 (rows), keeping only the real diagonal
          const cplx gqp = make_double2(gpq.x, -gpq.y);
          const cplx x_pp = csub(cscale(gpp, ca), cmulc(gpq, sa));   // c gpp - conj(sp) gpq
          const cplx x_pq = cadd(cmul(sa, gpp), cscale(gpq, ca));     // sp gpp + c gpq
          const cplx x_qp = csub(cscale(gqp, ca), cmulc(gqq, sa));   // c gqp - conj(sp) gqq
          const cplx x_qq = cadd(cmul(sa, gqp), cscale(gqq, ca));     // sp gqp + c gqq
          const cplx n_pp = csub(cscale(x_pp, ca), cmul(sa, x_qp));   // c x_pp - sp x_qp
          const cplx n_qq = cadd(cmulc(x_pq, sa), cscale(x_qq, ca));  // conj(sp) x_pq + c x_qq
          G[pa * GP + pa] = make_double2(n_pp.x, 0.0);
          G[qa * GP + qa] = make_double2(n_qq.x, 0.0);
          G[pa * GP + qa] = make_double2(0.0, 0.0);
          G[qa * GP + pa] = make_double2(0.0, 0.0);
          continue;
        }
        const cplx g00 = G[pa * GP + pb], g01 = G[pa * GP + qb], g10 = G[qa * GP + pb], g11 = G[qa * GP + qb];
        // columns: [x0 x1] = [g0 g1] J_B
        const cplx x00 = csub(cscale(g00, cb), cmulc(g01, sb)), x01 = cadd(cmul(sb, g00), cscale(g01, cb));
        const cplx x10 = csub(cscale(g10, cb), cmulc(g11, sb)), x11 = cadd(cmul(sb, g10), cscale(g11, cb));
        // rows: [n0; n1] = J_A^H [x0; x1] :  n0 = c x0 - sp x1,  n1 = conj(sp) x0 + c x1
        const cplx n00 = csub(cscale(x00, ca), cmul(sa, x10)), n01 = csub(cscale(x01, ca), cmul(sa, x11));
        const cplx n10 = cadd(cmulc(x00, sa), cscale(x10, ca)), n11 = cadd(cmulc(x01, sa), cscale(x11, ca));
        G[pa * GP + pb] = n00; G[pa * GP + qb] = n01; G[qa * GP + pb] = n10; G[qa * GP + qb] = n11;
        G[pb * GP + pa] = make_double2(n00.x, -n00.y);
        G[qb * GP + pa] = make_double2(n01.x, -n01.y);
        G[pb * GP + qa] = make_double2(n10.x, -n10.y);
        G[qb * GP + qa] = make_double2(n11.x, -n11.y);
      }
#pragma unroll 2
      for (int i = 0; i < 2048 / JNT; ++i) {
        int u = tid + JNT * i;
        int t = u & 31, r = u >> 5;
        if (!ctl[1 + t]) continue;
        int pp, qq;
        pair_of(t, pp, qq);
        double c = rc[t];
        cplx sp = rs[t];
        cplx qp = Q[r * GP + pp], qv = Q[r * GP + qq];
        Q[r * GP + pp] = csub(cscale(qp, c), cmulc(qv, sp));
        Q[r * GP + qq] = cadd(cmul(sp, qp), cscale(qv, c));
      }
      __syncthreads();
    }
    if (ctl[0] == 0) break;
    // Would the next sweep rotate anything?  It applies the test of phase A to every pair, and a sweep that finds no
    // pair active changes nothing — so one parallel pass over G with the same test decides, instead of 63 steps of two
    // barriers each that only confirm convergence (the same Q bit for bit; a pair solve is 2 - 3 rotating sweeps, the
    // confirming one was a quarter to a third of the kernel).
    {
      bool need = false;
      for (int idx = tid; idx < JP * JP; idx += JNT) {
        const int r = idx >> 6, c = idx & 63;
        if (r < c) {
          const cplx g = G[r * GP + c];
          const double ag = sqrt(g.x * g.x + g.y * g.y);
          if (ag > 0.0 && ag > absfloor && ag > tol_inner * sqrt(fabs(G[r * GP + r].x * G[c * GP + c].x))) need = true;
        }
      }
      if (need) ctl[33] = 1;  // benign race: all writers store 1
    }
    __syncthreads();
    const int go = ctl[33];
    __syncthreads();
    if (!go) break;
  }

  cplx* Qo = Qbuf + (size_t)it.q * JP * JP;
  for (int idx = tid; idx < JP * JP; idx += JNT) {
    int r = idx >> 6, c = idx & 63;
    Qo[idx] = Q[r * GP + c];
  }
}

// ---------------------------------------------------------------------------
// 3. rows(pair) <- Q^H rows(pair), in place, all passenger columns
// ---------------------------------------------------------------------------
// grid = (items, column chunks).  Each wave takes 16 columns at a time: the
// 64x16 strip is pulled into registers (it is the MFMA B operand as it stands),
// multiplied by Q^H from LDS, and written back — so the update is in place.
constexpr int APPLY_CHUNK = 1024;  // columns per workgroup (Q^H, 64 KB, is re-read per chunk: 6 % on top of the rows)
constexpr int APPLY_NT = 512;      // eight waves share one Q^H in LDS (82 KB: one workgroup per CU): two waves per SIMD

__global__ __launch_bounds__(APPLY_NT) void jac_apply_kernel(const jac_item* __restrict__ items,
                                                             const int* __restrict__ active,
                                                             const int* __restrict__ skip,
                                                             const cplx* __restrict__ Qbuf,
                                                             unsigned long long* __restrict__ flopctr) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* Are = reinterpret_cast<double*>(smem);  // A = Q^H : Are[k][i] = Re conj(Q[k][i]) = Re Q[k][i]
  double* Aim = Are + JP * QP;                    //            Aim[k][i] = -Im Q[k][i]
  const jac_item it = items[blockIdx.x];
  if (!active[it.prob] || skip[it.q]) return;
  const int cbeg = it.c0 + blockIdx.y * APPLY_CHUNK;
  if (cbeg >= it.c1) return;
  const int cend = min(it.c1, cbeg + APPLY_CHUNK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (flopctr && tid == 0) {
    const unsigned long long nr = it.na + it.nb;
    atomicAdd(flopctr, 8ull * nr * nr * (unsigned long long)(cend - cbeg));
  }

  const cplx* Qi = Qbuf + (size_t)it.q * JP * JP;
  for (int idx = tid; idx < JP * JP; idx += APPLY_NT) {
    int k = idx >> 6, i = idx & 63;
    cplx v = dm_ldg(Qi, idx);  // Q[k][i]
    Are[k * QP + i] = v.x;
    Aim[k * QP + i] = -v.y;
  }
  __syncthreads();

  const int fj = lane & 15, fk = lane >> 4;
  // (lane feeds the k-steps k = 4*ks + fk: global row item_row(it, k))
  // The strip a wave works on next is requested BEFORE the MFMAs of the current one (a wave used to load, compute and
  // store in turn: with one or two waves per SIMD the matrix pipe idled through every load — 2.5 TB/s of row traffic).
  constexpr int CSTEP = (APPLY_NT / 64) * 16;
  cplx cur[16], nxt[16];
  auto load_strip = [&](int c, cplx (&v)[16]) {
    const int colc = min(c + fj, cend - 1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int kr = item_row(it, ks * 4 + fk);
      // global_load from a clamped (always valid) address, no branch per load; flat loads would also count on LGKM and
      // every wait for an LDS read of Q^H would drain the prefetch
      // No mask (a select behind the load would wait for it before the MFMAs): a row outside the item meets an exactly
      // zero column of Q^H (jac_inner never rotates a row whose Gram entries are all zero: Q stays the identity there)
      // and is not stored; a column outside the chunk only feeds its own, unstored, output column.
      v[ks] = dm_ldg(it.Z, (size_t)(kr >= 0 ? kr : it.ra) * it.ld + colc);
    }
  };
  int c = cbeg + wave * 16;
  if (c < cend) load_strip(c, cur);
  for (; c < cend; c += CSTEP) {
    const int col = c + fj;
    const bool cok = col < cend;
    if (c + CSTEP < cend) load_strip(c + CSTEP, nxt);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      dm_f64x4 ore = {0, 0, 0, 0}, oim = {0, 0, 0, 0};
      // 8 MFMAs of an accumulator back to back (SrcC forwarding: 69.5 cycles per v_mfma_f64_16x16x4_f64 in chains of 8
      // against 83 with re / im alternating in pairs, 107 at one wave per SIMD; scratch/mfma_peak3.hip)
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        double a_re[4], a_im[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int ra = ((kq * 4 + k4) * 4 + fk) * QP + mt * 16 + fj;
          a_re[k4] = Are[ra];
          a_im[k4] = Aim[ra];
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          ore = dm_mfma(a_re[k4], cur[kq * 4 + k4].x, ore);
          ore = dm_mfma(-a_im[k4], cur[kq * 4 + k4].y, ore);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          oim = dm_mfma(a_re[k4], cur[kq * 4 + k4].y, oim);
          oim = dm_mfma(a_im[k4], cur[kq * 4 + k4].x, oim);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        int grow = item_row(it, mt * 16 + (lane >> 4) + 4 * rr);
        if (cok && grow >= 0) dm_stg(it.Z, (size_t)grow * it.ld + col, make_double2(ore[rr], oim[rr]));
      }
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) cur[ks] = nxt[ks];
  }
}

// ---------------------------------------------------------------------------
// helpers: norms, ranking, row gather, transpose, diagonal
// ---------------------------------------------------------------------------
struct jac_pdesc {
  cplx* Z; int ld; int row0; int nrows; int c0; int c1; int g0; int g1;
};

// one wave per row: sigma[p*stride + i] = || Z[row0+i, g0:g1] ||
__global__ void jac_rownorm_kernel(const jac_pdesc* __restrict__ pd, double* __restrict__ sigma, int stride,
                                   int maxrows) {
  const int p = blockIdx.y;
  const jac_pdesc d = pd[p];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= d.nrows) return;
  const cplx* z = d.Z + (size_t)(d.row0 + row) * d.ld;
  // scaled accumulation is unnecessary: entries are O(1e4) at most and fp64 range is ample
  double s = 0.0;
  for (int c = d.g0 + lane; c < d.g1; c += 64) s += cabs2(z[c]);
  s = dm_wave_sum(s);
  if (lane == 0) sigma[(size_t)p * stride + row] = sqrt(s);
}

// floor[p] = factor * max_i key[p][i]^2  (rows whose mutual Gram entries fall below
// this are numerical noise: singular values under ~4 eps sigma_max)
__global__ void jac_floor_kernel(const double* __restrict__ key, int stride, const int* __restrict__ nrows_p,
                                 double* __restrict__ floor_out, double factor) {
  const int p = blockIdx.x;
  const int n = nrows_p[p];
  __shared__ double red[4];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, key[(size_t)p * stride + i]);
  m = dm_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double mm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    floor_out[p] = factor * mm * mm;
  }
}

// rank[i] = position of row i in descending (or ascending) order of key, stable
__global__ void jac_rank_kernel(const double* __restrict__ key, int stride, const int* __restrict__ nrows_p,
                                int* __restrict__ rank, int descending) {
  const int p = blockIdx.y;
  const int n = nrows_p[p];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* k = key + (size_t)p * stride;
  const double ki = k[i];
  int r = 0;
  for (int j = 0; j < n; ++j) {
    double kj = k[j];
    bool before = descending ? (kj > ki) : (kj < ki);
    if (before || (kj == ki && j < i)) ++r;
  }
  rank[(size_t)p * stride + i] = r;
}

// dst[p][rank[i], c0:c1] = src row i ; sorted keys written too
__global__ void jac_gather_rows_kernel(const jac_pdesc* __restrict__ pd, const int* __restrict__ rank, int stride,
                                       cplx* __restrict__ tmp, const size_t* __restrict__ tmp_off,
                                       const double* __restrict__ key, double* __restrict__ key_sorted) {
  const int p = blockIdx.z;
  const jac_pdesc d = pd[p];
  const int row = blockIdx.y;
  if (row >= d.nrows) return;
  const int w = d.c1 - d.c0;
  const int dst = rank[(size_t)p * stride + row];
  const cplx* s = d.Z + (size_t)(d.row0 + row) * d.ld + d.c0;
  cplx* t = tmp + tmp_off[p] + (size_t)dst * w;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < w; c += gridDim.x * blockDim.x) t[c] = s[c];
  if (blockIdx.x == 0 && threadIdx.x == 0) key_sorted[(size_t)p * stride + dst] = key[(size_t)p * stride + row];
}

__global__ void jac_scatter_back_kernel(const jac_pdesc* __restrict__ pd, const cplx* __restrict__ tmp,
                                        const size_t* __restrict__ tmp_off) {
  const int p = blockIdx.z;
  const jac_pdesc d = pd[p];
  const int row = blockIdx.y;
  if (row >= d.nrows) return;
  const int w = d.c1 - d.c0;
  cplx* s = d.Z + (size_t)(d.row0 + row) * d.ld + d.c0;
  const cplx* t = tmp + tmp_off[p] + (size_t)row * w;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < w; c += gridDim.x * blockDim.x) s[c] = t[c];
}

// ---- block rotation between the rows above a level boundary (A) and the rows below it (B) ----
struct jac_clean_desc {
  cplx* Z; int ld; int row0; int ra; int rb; int ncols;
  cplx* theta;          // rb x ra, row-major
  const double* anorm;  // ra row norms of A over the Gram columns
  cplx* tmp;            // (ra + rb) x ncols: P2 (ra rows) then P1 (rb rows)
};
// theta[j][i] = -c[j][i] / ||a_i||^2   (c = B A^H on entry)
__global__ void jac_theta_kernel(const jac_clean_desc* __restrict__ ds) {
  const jac_clean_desc d = ds[blockIdx.z];
  const int j = blockIdx.y;
  if (j >= d.rb) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.ra; i += gridDim.x * blockDim.x) {
    const double nrm = d.anorm[i];
    const double w = nrm > 0.0 ? -1.0 / (nrm * nrm) : 0.0;
    cplx c = d.theta[(size_t)j * d.ra + i];
    d.theta[(size_t)j * d.ra + i] = make_double2(c.x * w, c.y * w);
  }
}
// tmp row r <- A_r - tmp_r / 2 (r < ra)   |   B_r + tmp_r / 2 (r >= ra)
__global__ void jac_clean_mix_kernel(const jac_clean_desc* __restrict__ ds) {
  const jac_clean_desc d = ds[blockIdx.z];
  const int r = blockIdx.y;
  if (r >= d.ra + d.rb) return;
  const double h = r < d.ra ? -0.5 : 0.5;
  const cplx* z = d.Z + (size_t)(d.row0 + r) * d.ld;
  cplx* t = d.tmp + (size_t)r * d.ncols;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.ncols; c += gridDim.x * blockDim.x) {
    const cplx a = z[c], b = t[c];
    t[c] = make_double2(a.x + h * b.x, a.y + h * b.y);
  }
}

// out = conj(in)^T for square n x n blocks, batched through descriptors
struct jac_tdesc {
  const cplx* src; int lds; cplx* dst; int ldd; int n;
};
__global__ void jac_ctrans_kernel(const jac_tdesc* __restrict__ td, const int* __restrict__ active) {
  __shared__ cplx tile[32][33];
  const jac_tdesc d = td[blockIdx.z];
  if (active && !active[blockIdx.z]) return;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  if (bx >= d.n || by >= d.n) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = by + j, c = bx + tx;
    tile[j][tx] = (r < d.n && c < d.n) ? d.src[(size_t)r * d.lds + c] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int r = bx + j, c = by + tx;
    if (r < d.n && c < d.n) {
      cplx v = tile[tx][j];
      d.dst[(size_t)r * d.ldd + c] = make_double2(v.x, -v.y);
    }
  }
}

// evals[p][i] = Re C[i][i]; also absfloor[p] = 8 eps max|diag| on first call
__global__ void jac_diag_kernel(const jac_tdesc* __restrict__ td, double* __restrict__ evals, int stride) {
  const jac_tdesc d = td[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) evals[(size_t)blockIdx.y * stride + i] = d.src[(size_t)i * d.lds + i].x;
}

__global__ void jac_absmax_kernel(const jac_tdesc* __restrict__ td, double* __restrict__ out, double factor) {
  // one block per problem: max |C_ij| over the whole matrix, times factor
  const jac_tdesc d = td[blockIdx.x];
  __shared__ double red[4];
  double m = 0.0;
  for (size_t idx = threadIdx.x; idx < (size_t)d.n * d.n; idx += blockDim.x) {
    int r = idx / d.n, c = idx % d.n;
    cplx v = d.src[(size_t)r * d.lds + c];
    m = fmax(m, fmax(fabs(v.x), fabs(v.y)));
  }
  m = dm_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = factor * fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// round-robin tournament over nb (even) players; returns rounds x (nb/2) pairs
void tournament(int nb, std::vector<std::vector<std::pair<int, int>>>& rounds) {
  rounds.clear();
  if (nb < 2) return;
  std::vector<int> idx(nb);
  for (int i = 0; i < nb; ++i) idx[i] = i;
  for (int r = 0; r < nb - 1; ++r) {
    std::vector<std::pair<int, int>> pr;
    for (int i = 0; i < nb / 2; ++i) {
      int a = idx[i], b = idx[nb - 1 - i];
      pr.emplace_back(std::min(a, b), std::max(a, b));
    }
    rounds.push_back(pr);
    int last = idx[nb - 1];
    for (int i = nb - 1; i > 1; --i) idx[i] = idx[i - 1];
    idx[1] = last;
  }
}

constexpr size_t INNER_LDS = (size_t)2 * JP * GP * sizeof(cplx) + 32 * sizeof(double) + 32 * sizeof(cplx) +
                             40 * sizeof(int) + 4 * sizeof(double) + 64 + 16 * sizeof(double);  // + room for JNT / 64 reduction slots
constexpr size_t APPLY_LDS = (size_t)2 * JP * QP * sizeof(double);

bool g_attr_set = false;
int set_attrs(dm_ctx* ctx) {
  if (g_attr_set) return DM_OK;
  DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(jac_inner_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)INNER_LDS));
  DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(jac_inner_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)INNER_LDS));
  DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(jac_apply_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)APPLY_LDS));
  g_attr_set = true;
  return DM_OK;
}

// Build the per-round item lists for a set of (possibly differently sized)
// problems.  `make_item(prob, blockA, blockB_or_-1, slot)` fills one item.
struct round_plan {
  std::vector<jac_item> items;         // all rounds, concatenated
  std::vector<int> round_begin;        // size nrounds + 1
  int max_items_per_round = 0;
};

template <typename F>
void plan_rounds(const std::vector<int>& nrows, F make_item, round_plan& plan) {
  // A batch holds thousands of problems of a handful of distinct block counts (configs[1]: 2064 chains of 3 blocks): the
  // round-robin schedule is made once per block count and shared — this planning ran three to six times per SVD stage, a
  // millisecond each with a vector of vectors of vectors per problem.
  const int np = (int)nrows.size();
  typedef std::vector<std::vector<std::pair<int, int>>> sched_t;
  std::map<int, sched_t> by_nb;                 // key: padded (even) block count; 1: the single-block schedule
  std::vector<const sched_t*> sched(np, nullptr);
  std::vector<int> nbs(np, 0);
  int maxrounds = 0;
  size_t nitems = 0;
  for (int p = 0; p < np; ++p) {
    const int nb = (nrows[p] + JB - 1) / JB;
    nbs[p] = nb;
    if (nb <= 0) continue;
    const int key = nb == 1 ? 1 : nb + (nb & 1);
    auto it = by_nb.find(key);
    if (it == by_nb.end()) {
      sched_t sc;
      if (nb == 1) sc = {{{0, -1}}};
      else tournament(key, sc);
      it = by_nb.emplace(key, std::move(sc)).first;
    }
    sched[p] = &it->second;
    maxrounds = std::max(maxrounds, (int)it->second.size());
    nitems += it->second.size() * (size_t)std::max(1, key / 2);
  }
  plan.items.clear();
  plan.items.reserve(nitems);
  plan.round_begin.assign(1, 0);
  plan.max_items_per_round = 0;
  for (int r = 0; r < maxrounds; ++r) {
    int slot = 0;
    for (int p = 0; p < np; ++p) {
      if (!sched[p] || r >= (int)sched[p]->size()) continue;
      const bool multi = sched[p]->size() > 1;
      for (const auto& pr0 : (*sched[p])[r]) {
        const int second = pr0.second >= nbs[p] ? -1 : pr0.second;   // phantom block of an odd count
        if (second < 0 && multi) continue;  // bye: a lone block needs no work this round
        plan.items.push_back(make_item(p, pr0.first, second, slot));
        ++slot;
      }
    }
    plan.max_items_per_round = std::max(plan.max_items_per_round, slot);
    plan.round_begin.push_back((int)plan.items.size());
  }
}

}  // namespace

// ===========================================================================
// one-sided driver
// ===========================================================================
int dm_jacobi_rows(dm_ctx* ctx, const std::vector<dm_jac_problem>& probs, double* sigma, int sigma_stride,
                   int* sweeps_out, const dm_jac_rows_opts* opts) {
  dm_jac_rows_opts O = opts ? *opts : dm_jac_rows_opts();
  if (getenv("DM_JAC_MEASURE")) O.unconverged = false;   // debugging switches
  if (getenv("DM_JAC_NO_DROP")) O.drop_below = 0.0;
  const bool full_gram = getenv("DM_JAC_FULL_GRAM") != nullptr;
  // DM_DEBUG: wall-clock of the phases of this call (synchronises at every mark)
  const bool dbg_t = getenv("DM_DEBUG") != nullptr;
  const bool dbg_sync = getenv("DM_DEBUG_NOSYNC") == nullptr;
  auto dbg_now = [&]() { if (dbg_sync) (void)hipStreamSynchronize(ctx->stream); return std::chrono::steady_clock::now(); };
  auto dbg_t0 = dbg_t ? dbg_now() : std::chrono::steady_clock::time_point();
  auto dbg_mark = [&](const char* what) {
    if (!dbg_t) return;
    auto t1 = dbg_now();
    fprintf(stderr, "[jacobi_rows]   %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - dbg_t0).count());
    dbg_t0 = t1;
  };
  const int np = (int)probs.size();
  if (sweeps_out) *sweeps_out = 0;
  if (np == 0) return DM_OK;
  DM_TRY(set_attrs(ctx));
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;

  std::vector<int> nrows(np);
  int maxrows = 0, maxcols = 0;
  for (int p = 0; p < np; ++p) {
    nrows[p] = probs[p].nrows;
    maxrows = std::max(maxrows, nrows[p]);
    maxcols = std::max(maxcols, probs[p].ncols);
  }
  DM_ARG(ctx, maxrows <= sigma_stride);
  if (maxrows == 0) {  // nothing to orthogonalise anywhere (e.g. every row of every block was cut by the caller)
    dm_ws_release(ctx, mark);
    return DM_OK;
  }

  round_plan plan;
  std::vector<int> nrows_eff(nrows);  // rows that take part in the sweeps (all of them unless drop_below cuts the tail)
  // one "rows mutually orthogonal" flag per 32-row block of every problem (jac_gram's CROSS mode)
  std::vector<int> flag0(np + 1, 0);
  for (int p = 0; p < np; ++p) flag0[p + 1] = flag0[p] + (nrows[p] + JB - 1) / JB;
  auto build_plan = [&]() {
    plan_rounds(nrows_eff, [&](int p, int ba, int bb, int slot) {
      const dm_jac_problem& P = probs[p];
      jac_item it;
      it.Z = P.Z; it.ld = P.ld;
      it.ra = P.row0 + ba * JB; it.na = std::min(JB, nrows_eff[p] - ba * JB);
      if (bb >= 0) { it.rb = P.row0 + bb * JB; it.nb = std::min(JB, nrows_eff[p] - bb * JB); }
      else { it.rb = 0; it.nb = 0; }
      it.c0 = 0; it.c1 = P.ncols; it.g0 = P.gc0; it.g1 = P.gc1;
      it.prob = p; it.q = slot;
      it.fa = flag0[p] + ba; it.fb = bb >= 0 ? flag0[p] + bb : it.fa;
      return it;
    }, plan);
  };
  build_plan();

  std::vector<jac_pdesc> pd(np);
  std::vector<size_t> toff(np);
  size_t ttot = 0;
  for (int p = 0; p < np; ++p) {
    pd[p] = jac_pdesc{probs[p].Z, probs[p].ld, probs[p].row0, probs[p].nrows, 0, probs[p].ncols, probs[p].gc0,
                      probs[p].gc1};
    toff[p] = ttot;
    ttot += (size_t)probs[p].nrows * probs[p].ncols;
  }

  jac_item* d_items = dm_ws_upload(ctx, plan.items);
  jac_pdesc* d_pd = dm_ws_upload(ctx, pd);
  int* d_nrows = dm_ws_upload(ctx, nrows);
  size_t* d_toff = dm_ws_upload(ctx, toff);
  std::vector<int> active(np, 1);
  int* d_active = dm_ws_upload(ctx, active);
  unsigned long long* d_off = dm_ws_alloc_t<unsigned long long>(ctx, np);
  const size_t nslots = (size_t)std::max(plan.max_items_per_round, 1);
  cplx* d_G = dm_ws_alloc_t<cplx>(ctx, nslots * JP * JP);
  cplx* d_Q = dm_ws_alloc_t<cplx>(ctx, nslots * JP * JP);
  int* d_skip = dm_ws_alloc_t<int>(ctx, nslots);
  int* d_rank = dm_ws_alloc_t<int>(ctx, (size_t)np * sigma_stride);
  double* d_key = dm_ws_alloc_t<double>(ctx, (size_t)np * sigma_stride);
  cplx* d_tmp = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(ttot, 1));
  double* d_floor = dm_ws_alloc_t<double>(ctx, np);
  const size_t nflags = (size_t)std::max(flag0[np], 1);
  int* d_ok = dm_ws_alloc_t<int>(ctx, nflags);
  if (!d_items || !d_pd || !d_nrows || !d_toff || !d_active || !d_off || !d_G || !d_Q || !d_skip || !d_rank ||
      !d_key || !d_tmp || !d_floor || !d_ok)
    return DM_ENOMEM;
  // DM_JAC_CROSS=0: every pair Gram in full (upper tiles), as before round 5
  static const int allow_cross = getenv("DM_JAC_CROSS") ? atoi(getenv("DM_JAC_CROSS")) : 1;
  DM_HIP(ctx, hipMemsetAsync(d_ok, 0, sizeof(int) * nflags, ctx->stream));

  // noise floor for Gram entries: (4 eps)^2 * (largest row norm)^2.  Rows that are pure
  // rounding residue (rank-deficient inputs) can never be made mutually orthogonal to
  // relative accuracy — there are more of them than dimensions left — so pairs of such
  // rows are left alone, exactly the level at which LAPACK's backward error sits.
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rownorm_kernel, dim3((maxrows + 3) / 4, np), dim3(256), 0, ctx->stream, d_pd, d_key,
                     sigma_stride, maxrows);
  {
    const double e4 = 4.0 * 2.220446049250313e-16;
    DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_floor_kernel, dim3(np), dim3(256), 0, ctx->stream, d_key, sigma_stride, d_nrows,
                       d_floor, e4 * e4);
  }
  // Rows count as orthogonal at |cos| <= K u, K the length of the inner products and u the unit roundoff —
  // the computed Gram entry of two exactly orthogonal rows is no smaller than that (LAPACK's zgesvj stops
  // at the same level, TOL = CTOL * EPS with CTOL = M) — but never looser than 1e-13 (K <= 900).
  int maxk = 0;
  for (int p = 0; p < np; ++p) maxk = std::max(maxk, probs[p].gc1 - probs[p].gc0);
  const double tol_outer = std::max(1e-13, maxk * 1.1102230246251565e-16), tol_inner = 1e-15;
  int nrounds = (int)plan.round_begin.size() - 1;
  std::vector<unsigned long long> h_off(np);
  // Measuring pass (no rotations): problems whose rows are already orthogonal to tolerance — the
  // pseudo-inverse pass of an unpolarised telescope sees exactly the rows the previous pass
  // produced — skip the preconditioner and the sweeps.  (Re-diagonalising their Gram matrix would
  // even hurt: it is only accurate to eps sigma_1^2 and disturbs the small rows.)
  if (O.unconverged) {
    for (int p = 0; p < np; ++p) active[p] = nrows[p] > 1 ? 1 : 0;
    DM_TRY(dm_upload(ctx, d_active, active.data(), sizeof(int) * np));
  } else {
    DM_HIP(ctx, hipMemsetAsync(d_off, 0, sizeof(unsigned long long) * np, ctx->stream));
    for (int r = 0; r < nrounds; ++r) {
      const int nb = plan.round_begin[r], ni = plan.round_begin[r + 1] - nb;
      if (ni == 0) continue;
      hipLaunchKernelGGL(jac_gram_kernel, dim3(ni), dim3(256), 0, ctx->stream, d_items + nb, d_active, d_G,
                         (const int*)d_ok, 0, (unsigned long long*)nullptr);
      hipLaunchKernelGGL(jac_inner_kernel<false>, dim3(ni), dim3(JNT), INNER_LDS, ctx->stream, d_items + nb, d_active,
                         (const double*)d_floor, d_G, d_Q, d_off, d_skip, tol_outer, tol_inner, 1, (int*)nullptr);
    }
    DM_HIP(ctx, hipGetLastError());
    DM_TRY(dm_download(ctx, h_off.data(), d_off, sizeof(unsigned long long) * np));
    for (int p = 0; p < np; ++p) {
      double mo;
      std::memcpy(&mo, &h_off[p], sizeof(double));
      active[p] = (nrows[p] > 1 && mo > tol_outer) ? 1 : 0;
    }
    DM_TRY(dm_upload(ctx, d_active, active.data(), sizeof(int) * np));
  }
  dbg_mark("setup + measuring pass");
  // Preconditioner: one Hermitian eigendecomposition of the full Gram matrix G = X X^H of every
  // (still active) problem (batched tridiagonal solver) followed by Z <- W Z.  On its own this would only be
  // accurate to eps ||X||^2 (the Gram squares the condition number), but it brings every pair of
  // rows to |cos| <~ eps sigma_1^2 / (sigma_i sigma_j), from where the Jacobi sweeps below — which recompute
  // the Gram blocks from the rows themselves and therefore keep full relative accuracy — converge
  // in two or three sweeps instead of a dozen on the graded spectra of beam matrices.
  // That bound says nothing about pairs with sigma_i sigma_j <~ eps sigma_1^2: the directions below
  // ~1e-8 sigma_1 come out of the first level as an arbitrary mixture (a polarised beam block has several
  // hundred of them, spread over 8 more decades: 30 sweeps).  So the preconditioner is applied again to
  // just those rows — their own Gram matrix resolves another 8 decades relative to THEIR largest norm —
  // and once more below that; each level costs a fraction of one sweep.
  std::vector<char> placed(np, 0);   // subspace mode: a level of the preconditioner placed the cut of this problem
  {
    std::vector<size_t> goff(np);
    size_t gtot = 0;
    for (int p = 0; p < np; ++p) { goff[p] = gtot; gtot += (size_t)nrows[p] * nrows[p]; }
    cplx* Gm = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(gtot, 1));
    cplx* Wm = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(gtot, 1));
    double* evp = dm_ws_alloc_t<double>(ctx, (size_t)np * sigma_stride);
    if (!Gm || !Wm || !evp) return DM_ENOMEM;
    std::vector<int> sub0(np, 0), lvl_on(active);  // first row of the current level's sub-block, per problem
    std::vector<double> ev0(np, 0.0);              // largest Gram eigenvalue of level 0 (sigma_1^2)
    std::vector<double> hev;
    int max_levels = 4;
    if (const char* e = getenv("DM_JAC_PRECOND_LEVELS")) max_levels = std::max(0, std::min(6, atoi(e)));
    const bool clean = !getenv("DM_JAC_NO_CLEAN");
    int level_min_rows = 2 * JP;
    if (const char* e = getenv("DM_JAC_LEVEL_MIN_ROWS")) level_min_rows = std::max(1, atoi(e));
    for (int level = 0; level < max_levels; ++level) {
      if (level > 0 && clean) {
        // The rows of this level still carry components along the rows above them (A) of absolute size
        // ~eps sigma_1^2 / sigma_i — as large as the rows themselves.  Remove them BEFORE looking at the
        // level's own Gram matrix, with the block rotation W = [[I - T^H T / 2, -T^H], [T, I - T T^H / 2]],
        // T = -(B A^H) diag(||a_i||^-2): the rows of A are mutually orthogonal already, |T_ji| <= eps
        // sigma_1^2 / sigma_i^2 <= 2e-7 by the choice of the level boundary, so W is unitary to ||T||^4.
        std::vector<jac_clean_desc> cd;
        std::vector<jac_pdesc> pa;
        std::vector<dm_gemm_desc> gC, gP1, gP2, gA, gB;
        int maxra = 0, maxrb = 0;
        for (int p = 0; p < np; ++p) {
          const dm_jac_problem& P = probs[p];
          const int ra = sub0[p], rb = P.nrows - sub0[p];
          if (!lvl_on[p] || ra < 1 || rb < 1) continue;
          cplx* Arow = P.Z + (size_t)P.row0 * P.ld;
          cplx* Brow = P.Z + (size_t)(P.row0 + ra) * P.ld;
          cplx* th = Gm + goff[p];                         // rb x ra  (<= nrows^2 / 4)
          cplx* t2 = d_tmp + toff[p];                      // P2: ra x ncols
          cplx* t1 = t2 + (size_t)ra * P.ncols;            // P1: rb x ncols
          cd.push_back(jac_clean_desc{P.Z, P.ld, P.row0, ra, rb, P.ncols, th, nullptr, t2});  // anorm set below
          jac_pdesc d = pd[p];
          d.nrows = ra;
          pa.push_back(d);
          gC.push_back(dm_gemm_make(Brow + P.gc0, P.ld, 1, false, Arow + P.gc0, 1, P.ld, true, th, ra, rb, ra,
                                    P.gc1 - P.gc0));
          gP1.push_back(dm_gemm_make(th, ra, 1, false, Arow, P.ld, 1, false, t1, P.ncols, rb, P.ncols, ra));
          gP2.push_back(dm_gemm_make(th, 1, ra, true, Brow, P.ld, 1, false, t2, P.ncols, ra, P.ncols, rb));
          gA.push_back(dm_gemm_make(th, 1, ra, true, t1, P.ncols, 1, false, Arow, P.ld, ra, P.ncols, rb, -1.0, 1.0));
          gB.push_back(dm_gemm_make(th, ra, 1, false, t2, P.ncols, 1, false, Brow, P.ld, rb, P.ncols, ra, 1.0, 1.0));
          maxra = std::max(maxra, ra);
          maxrb = std::max(maxrb, rb);
        }
        if (!cd.empty()) {
          // row norms of A: one output row per (compacted) descriptor
          double* an = dm_ws_alloc_t<double>(ctx, cd.size() * (size_t)sigma_stride);
          if (!an) return DM_ENOMEM;
          for (size_t k = 0; k < cd.size(); ++k) cd[k].anorm = an + k * (size_t)sigma_stride;
          jac_pdesc* d_pa = dm_ws_upload(ctx, pa);
          jac_clean_desc* d_cd = dm_ws_upload(ctx, cd);
          if (!d_pa || !d_cd) return DM_ENOMEM;
          DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rownorm_kernel, dim3((maxra + 3) / 4, (unsigned)pa.size()), dim3(256), 0, ctx->stream, d_pa,
                             an, sigma_stride, maxra);
          DM_TRY(dm_gemm_grouped_launch(ctx, gC));
          DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_theta_kernel, dim3((maxra + 255) / 256, maxrb, (unsigned)cd.size()), dim3(256), 0,
                             ctx->stream, d_cd);
          DM_TRY(dm_gemm_grouped_launch(ctx, gP1));
          DM_TRY(dm_gemm_grouped_launch(ctx, gP2));
          const int gxm = std::max(1, std::min(8, (maxcols + 255) / 256));
          DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_clean_mix_kernel, dim3(gxm, maxra + maxrb, (unsigned)cd.size()), dim3(256), 0, ctx->stream,
                             d_cd);
          DM_TRY(dm_gemm_grouped_launch(ctx, gA));
          DM_TRY(dm_gemm_grouped_launch(ctx, gB));
          DM_HIP(ctx, hipGetLastError());
        }
      }
      std::vector<dm_gemm_desc> g;
      std::vector<dm_jac_herm_problem> hp;
      std::vector<int> who;
      for (int p = 0; p < np; ++p) {
        const dm_jac_problem& P = probs[p];
        const int ns = P.nrows - sub0[p];
        if (!lvl_on[p] || ns < 1) continue;
        const cplx* X = P.Z + (size_t)(P.row0 + sub0[p]) * P.ld + P.gc0;
        // the eigensolver reads the upper triangle only (LAPACK's uplo = 'U'): half of the Gram product
        g.push_back(dm_gemm_make(X, P.ld, 1, false, X, 1, P.ld, true, Gm + goff[p], ns, ns, ns, P.gc1 - P.gc0, 1.0, 0.0,
                                 nullptr, full_gram ? 0 : DM_GEMM_UPPER));
        hp.push_back(dm_jac_herm_problem{Gm + goff[p], ns, Wm + goff[p], ns, ns});
        who.push_back(p);
      }
      if (hp.empty()) break;
      DM_TRY(dm_gemm_grouped_launch(ctx, g));
      // evals land at consecutive strides of the *compacted* problem list
      dbg_mark("level: cleaning + Gram");
      {
        const int keep_mode = ctx->trd_mode_override;
        if (O.one_stage_eig) ctx->trd_mode_override = 0;
        const int rc_eig = dm_herm_eig_tridiag(ctx, hp, evp, sigma_stride);
        ctx->trd_mode_override = keep_mode;
        DM_TRY(rc_eig);
      }
      dbg_mark("level: eigensolver");
      std::vector<dm_jac_problem> sp;
      for (auto& h : hp) sp.push_back(dm_jac_problem{h.W, h.ldw, 0, h.n, h.n, 0, 0});
      DM_TRY(dm_sort_rows_by_key(ctx, sp, evp, sigma_stride, true));  // largest eigenvalue first
      // Z <- W Z through the temporary, then back
      std::vector<dm_gemm_desc> ga;
      std::vector<jac_pdesc> pda;   // the scatter only touches the rows that were transformed
      std::vector<size_t> toffa;
      int subrows = 0;
      for (size_t k = 0; k < who.size(); ++k) {
        const int p = who[k];
        const dm_jac_problem& P = probs[p];
        const int ns = hp[k].n;
        ga.push_back(dm_gemm_make(hp[k].W, ns, 1, false, P.Z + (size_t)(P.row0 + sub0[p]) * P.ld, P.ld, 1, false,
                                  d_tmp + toff[p], P.ncols, ns, P.ncols, ns));
        jac_pdesc d = pd[p];
        d.row0 = P.row0 + sub0[p];
        d.nrows = ns;
        pda.push_back(d);
        toffa.push_back(toff[p]);
        subrows = std::max(subrows, ns);
      }
      DM_TRY(dm_gemm_grouped_launch(ctx, ga));
      jac_pdesc* d_pda = dm_ws_upload(ctx, pda);
      size_t* d_toffa = dm_ws_upload(ctx, toffa);
      if (!d_pda || !d_toffa) return DM_ENOMEM;
      const int gx0 = std::max(1, std::min(8, (maxcols + 255) / 256));
      DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_scatter_back_kernel, dim3(gx0, subrows, (unsigned)pda.size()), dim3(256), 0, ctx->stream,
                         d_pda, d_tmp, d_toffa);
      DM_HIP(ctx, hipGetLastError());
      dbg_mark("level: sort + W Z");
      if (level + 1 == max_levels) break;
      // next level: the rows whose Gram eigenvalue fell below 1e-9 of this level's largest (sigma below
      // 3e-5 of it: a margin of three decades above what this level resolves, and the bound on the block
      // rotation above), unless they already sit at the rounding floor of the whole matrix
      hev.resize(who.size() * (size_t)sigma_stride);
      DM_TRY(dm_download(ctx, hev.data(), evp, sizeof(double) * hev.size()));
      bool more = false;
      for (size_t k = 0; k < who.size(); ++k) {
        const int p = who[k];
        const double* ev = &hev[k * (size_t)sigma_stride];
        const int ns = hp[k].n;
        if (level == 0) ev0[p] = ev[0];
        lvl_on[p] = 0;
        if (!(ev[0] > 0.0)) continue;
        int i = 0;
        // (subspace mode: the level below begins at margin x cut of the largest row norm if that is higher than the
        // regular boundary — the rows next to the cut then get a level of their own)
        const double sm2 = O.subspace_cut > 0.0 ? O.subspace_margin * O.subspace_margin * O.subspace_cut * O.subspace_cut : 0.0;
        const double bound = std::max(1e-9 * ev[0], level == 0 ? sm2 * ev[0] : 0.0);
        while (i < ns && ev[i] >= bound) ++i;
        const double e4 = 4.0 * 2.220446049250313e-16;
        // worth a level only when the rows below span several row blocks: a sweep costs ~(row blocks)^2, and up
        // to one pair of blocks the inner Jacobi solver sorts them out in LDS anyway (config 2: T = 92 rows)
        // subspace mode: this level begins within margin x cut of the largest row norm — the cut is placed
        if (O.subspace_cut > 0.0 && ev[0] <= sm2 * ev0[p] * (1.0 + 1e-9)) { placed[p] = 1; continue; }
        if (ns - i <= level_min_rows || i == 0) continue;   // (subspace mode: not placed — the sweeps below do it)
        if (bound <= e4 * e4 * ev0[p]) continue;  // what is left is rounding residue of the largest rows
        if (O.drop_below > 0.0 && ev[i] <= O.drop_below * O.drop_below * ev0[p] * 1e-2) continue;  // nobody wants them
        sub0[p] += i;
        lvl_on[p] = 1;
        more = true;
      }
      if (getenv("DM_DEBUG")) {
        int cnt = 0, lo = 1 << 30, hi = 0;
        for (int p = 0; p < np; ++p)
          if (lvl_on[p]) { ++cnt; lo = std::min(lo, probs[p].nrows - sub0[p]); hi = std::max(hi, probs[p].nrows - sub0[p]); }
        fprintf(stderr, "[jacobi_rows] preconditioner level %d done; %d problems go one level down (%d..%d rows)\n", level,
                cnt, cnt ? lo : 0, hi);
      }
      if (!more) break;
    }
  }

  dbg_mark("level: W Z + rest");
  if (O.subspace_cut > 0.0) {
    // the problems whose cut is placed are done: their rows are unitary mixtures of the input rows, split at the cut to
    // the accuracy a converged SVD would give; the others (too few rows for a level of their own, level cap) are swept
    bool ch = false;
    for (int p = 0; p < np; ++p)
      if (placed[p] && active[p]) { active[p] = 0; ch = true; }
    if (ch) DM_TRY(dm_upload(ctx, d_active, active.data(), sizeof(int) * np));
  }
  if (O.drop_below > 0.0) {
    // rows below the caller's level of interest leave the tournament (they sit at the end: the levels are
    // ordered by scale and sorted inside)
    DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rownorm_kernel, dim3((maxrows + 3) / 4, np), dim3(256), 0, ctx->stream, d_pd, d_key,
                       sigma_stride, maxrows);
    std::vector<double> hk((size_t)np * sigma_stride);
    DM_TRY(dm_download(ctx, hk.data(), d_key, sizeof(double) * hk.size()));
    bool changed = false;
    for (int p = 0; p < np; ++p) {
      const double* kp = &hk[(size_t)p * sigma_stride];
      double mx = 0.0;
      for (int i = 0; i < nrows[p]; ++i) mx = std::max(mx, kp[i]);
      int last = -1;
      for (int i = 0; i < nrows[p]; ++i)
        if (kp[i] >= O.drop_below * mx) last = i;
      const int ne = std::max(std::min(nrows[p], 1), last + 1);
      if (ne != nrows_eff[p]) { nrows_eff[p] = ne; changed = true; }
    }
    if (changed) {
      build_plan();  // never more items per round than the plan the buffers were sized for
      nrounds = (int)plan.round_begin.size() - 1;
      d_items = dm_ws_upload(ctx, plan.items);
      if (!d_items) return DM_ENOMEM;
    }
  }
  const int chunks = (maxcols + APPLY_CHUNK - 1) / APPLY_CHUNK;
  int sweep = 0;
  const int max_sweeps = 40;
  bool any_pairs = false;
  for (int p = 0; p < np; ++p) any_pairs |= active[p] != 0;
  for (; any_pairs && sweep < max_sweeps; ++sweep) {
    DM_HIP(ctx, hipMemsetAsync(d_off, 0, sizeof(unsigned long long) * np, ctx->stream));
    // every block's own Gram is measured from its rows the first time the block is met in a sweep
    DM_HIP(ctx, hipMemsetAsync(d_ok, 0, sizeof(int) * nflags, ctx->stream));
    for (int r = 0; r < nrounds; ++r) {
      const int nb = plan.round_begin[r], ne = plan.round_begin[r + 1];
      const int ni = ne - nb;
      if (ni == 0) continue;
      unsigned long long* fc = ctx->prof_on ? ctx->prof_dev : nullptr;
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_GRAM, 0.0);
        hipLaunchKernelGGL(jac_gram_kernel, dim3(ni), dim3(256), 0, ctx->stream, d_items + nb, d_active, d_G,
                           (const int*)d_ok, allow_cross, fc ? fc + DM_PROF_JAC_GRAM : nullptr);
      }
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_INNER, 0.0);
        hipLaunchKernelGGL(jac_inner_kernel<false>, dim3(ni), dim3(JNT), INNER_LDS, ctx->stream, d_items + nb,
                           d_active, (const double*)d_floor, d_G, d_Q, d_off, d_skip, tol_outer, tol_inner, 0, d_ok);
      }
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_APPLY, 0.0);
        hipLaunchKernelGGL(jac_apply_kernel, dim3(ni, chunks), dim3(APPLY_NT), APPLY_LDS, ctx->stream, d_items + nb,
                           d_active, d_skip, d_Q, fc ? fc + DM_PROF_JAC_APPLY : nullptr);
      }
    }
    DM_HIP(ctx, hipGetLastError());
    DM_TRY(dm_download(ctx, h_off.data(), d_off, sizeof(unsigned long long) * np));
    bool any = false;
    double dbg_max = 0.0;
    for (int p = 0; p < np; ++p) {
      double mo;
      std::memcpy(&mo, &h_off[p], sizeof(double));
      dbg_max = std::max(dbg_max, mo);
      // A sweep that met nothing above 1e-9 leaves nothing above ~n 1e-18 behind (the rotations of a sweep
      // disturb each other only to second order): the problem is done without a sweep that merely confirms it.
      active[p] = (active[p] && mo > std::max(tol_outer, 1e-9)) ? 1 : 0;
      any |= active[p] != 0;
    }
    if (getenv("DM_DEBUG")) {
      int nact = 0, first = -1;
      for (int p = 0; p < np; ++p) if (active[p]) { ++nact; if (first < 0) first = p; }
      fprintf(stderr, "[jacobi_rows] sweep %d offmax %.3e active %d/%d first %d (rows %d gram %d-%d)\n", sweep, dbg_max,
              nact, np, first, first >= 0 ? probs[first].nrows : 0, first >= 0 ? probs[first].gc0 : 0,
              first >= 0 ? probs[first].gc1 : 0);
    }
    if (!any) { ++sweep; break; }
    DM_TRY(dm_upload(ctx, d_active, active.data(), sizeof(int) * np));
  }
  if (sweeps_out) *sweeps_out = sweep;

  dbg_mark("sweeps");
  // sort rows by descending norm over the Gram columns
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rownorm_kernel, dim3((maxrows + 3) / 4, np), dim3(256), 0, ctx->stream, d_pd, d_key,
                     sigma_stride, maxrows);
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rank_kernel, dim3((maxrows + 255) / 256, np), dim3(256), 0, ctx->stream, d_key,
                     sigma_stride, d_nrows, d_rank, 1);
  const int gx = std::max(1, std::min(8, (maxcols + 255) / 256));
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_gather_rows_kernel, dim3(gx, maxrows, np), dim3(256), 0, ctx->stream, d_pd, d_rank,
                     sigma_stride, d_tmp, d_toff, d_key, sigma);
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_scatter_back_kernel, dim3(gx, maxrows, np), dim3(256), 0, ctx->stream, d_pd, d_tmp,
                     d_toff);
  DM_HIP(ctx, hipGetLastError());
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dbg_mark("final sort");
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// ===========================================================================
// two-sided (Hermitian) driver
// ===========================================================================
int dm_jacobi_herm(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* evals, int evals_stride,
                   int* sweeps_out) {
  const int np = (int)probs.size();
  if (sweeps_out) *sweeps_out = 0;
  if (np == 0) return DM_OK;
  DM_TRY(set_attrs(ctx));
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;

  std::vector<int> nrows(np);
  int maxn = 0;
  size_t ttot = 0;
  std::vector<size_t> toff(np);
  for (int p = 0; p < np; ++p) {
    nrows[p] = probs[p].n;
    maxn = std::max(maxn, probs[p].n);
    toff[p] = ttot;
    ttot += (size_t)probs[p].n * probs[p].n;
  }
  DM_ARG(ctx, maxn <= evals_stride);
  cplx* d_T = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(ttot, 1));  // ping-pong partner of C
  if (!d_T) return DM_ENOMEM;

  // Items: for each round three lists share the Q slots —
  //   (a) rows of the current C buffer, (b) rows of W, (c) rows of the transposed buffer.
  // The matrix alternates between C (even rounds) and T (odd rounds): round r
  // reads cur, row-updates it in place, transposes into oth, row-updates oth;
  // oth then holds Q^H C Q and becomes cur.
  round_plan planC, planT, planW;
  auto mk = [&](int which) {
    return [&, which](int p, int ba, int bb, int slot) {
      const dm_jac_herm_problem& P = probs[p];
      jac_item it;
      if (which == 0) { it.Z = P.C; it.ld = P.ldc; }
      else if (which == 1) { it.Z = d_T + toff[p]; it.ld = P.n; }
      else { it.Z = P.W; it.ld = P.ldw; }
      it.ra = ba * JB; it.na = std::min(JB, P.n - ba * JB);
      if (bb >= 0) { it.rb = bb * JB; it.nb = std::min(JB, P.n - bb * JB); } else { it.rb = 0; it.nb = 0; }
      it.c0 = 0; it.c1 = P.n; it.g0 = 0; it.g1 = P.n;
      it.prob = p; it.q = slot;
      it.fa = it.fb = 0;
      return it;
    };
  };
  plan_rounds(nrows, mk(0), planC);
  plan_rounds(nrows, mk(1), planT);
  plan_rounds(nrows, mk(2), planW);

  std::vector<jac_tdesc> tdC(np), tdT(np);
  for (int p = 0; p < np; ++p) {
    tdC[p] = jac_tdesc{probs[p].C, probs[p].ldc, d_T + toff[p], probs[p].n, probs[p].n};
    tdT[p] = jac_tdesc{d_T + toff[p], probs[p].n, probs[p].C, probs[p].ldc, probs[p].n};
  }

  jac_item* d_iC = dm_ws_upload(ctx, planC.items);
  jac_item* d_iT = dm_ws_upload(ctx, planT.items);
  jac_item* d_iW = dm_ws_upload(ctx, planW.items);
  jac_tdesc* d_tdC = dm_ws_upload(ctx, tdC);
  jac_tdesc* d_tdT = dm_ws_upload(ctx, tdT);
  std::vector<int> active(np, 1);
  int* d_active = dm_ws_upload(ctx, active);
  unsigned long long* d_off = dm_ws_alloc_t<unsigned long long>(ctx, np);
  double* d_floor = dm_ws_alloc_t<double>(ctx, np);
  const size_t nslots = (size_t)std::max(planC.max_items_per_round, 1);
  cplx* d_Q = dm_ws_alloc_t<cplx>(ctx, nslots * JP * JP);
  int* d_skip = dm_ws_alloc_t<int>(ctx, nslots);
  if (!d_iC || !d_iT || !d_iW || !d_tdC || !d_tdT || !d_active || !d_off || !d_floor || !d_Q || !d_skip)
    return DM_ENOMEM;

  // absolute rotation floor: 32 eps * max|C_ij| (a backward-stable solver cannot
  // resolve off-diagonals below this; kltransform's LAPACK path is no different)
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_absmax_kernel, dim3(np), dim3(256), 0, ctx->stream, d_tdC, d_floor,
                     32.0 * 2.220446049250313e-16);

  const double tol_outer = 1e-13, tol_inner = 1e-15;
  const int nrounds = (int)planC.round_begin.size() - 1;
  const int chunks = (maxn + APPLY_CHUNK - 1) / APPLY_CHUNK;
  const int tb = (maxn + 31) / 32;
  std::vector<unsigned long long> h_off(np);
  int sweep = 0;
  const int max_sweeps = 40;
  int cur = 0;  // 0: matrix lives in C, 1: in T
  bool any_pairs = false;
  for (int p = 0; p < np; ++p) any_pairs |= nrows[p] > 1;
  for (; any_pairs && sweep < max_sweeps; ++sweep) {
    DM_HIP(ctx, hipMemsetAsync(d_off, 0, sizeof(unsigned long long) * np, ctx->stream));
    for (int r = 0; r < nrounds; ++r) {
      const int nb = planC.round_begin[r], ne = planC.round_begin[r + 1];
      const int ni = ne - nb;
      if (ni == 0) continue;
      jac_item* icur = (cur == 0 ? d_iC : d_iT) + nb;
      jac_item* ioth = (cur == 0 ? d_iT : d_iC) + nb;
      unsigned long long* fc = ctx->prof_on ? ctx->prof_dev + DM_PROF_JAC_APPLY : nullptr;
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_INNER, 0.0);
        hipLaunchKernelGGL(jac_inner_kernel<true>, dim3(ni), dim3(JNT), INNER_LDS, ctx->stream, icur, d_active,
                           d_floor, (const cplx*)nullptr, d_Q, d_off, d_skip, tol_outer, tol_inner, 0, (int*)nullptr);
      }
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_APPLY, 0.0);
        hipLaunchKernelGGL(jac_apply_kernel, dim3(ni, chunks), dim3(APPLY_NT), APPLY_LDS, ctx->stream, icur, d_active,
                           d_skip, d_Q, fc);
      }
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_APPLY, 0.0);
        hipLaunchKernelGGL(jac_apply_kernel, dim3(ni, chunks), dim3(APPLY_NT), APPLY_LDS, ctx->stream, d_iW + nb,
                           d_active, d_skip, d_Q, fc);
      }
      DM_PLAUNCH(ctx, DM_PROF_UTIL, jac_ctrans_kernel, dim3(tb, tb, np), dim3(256), 0, ctx->stream,
                         cur == 0 ? d_tdC : d_tdT, d_active);
      {
        dm_prof_scope ps(ctx, DM_PROF_JAC_APPLY, 0.0);
        hipLaunchKernelGGL(jac_apply_kernel, dim3(ni, chunks), dim3(APPLY_NT), APPLY_LDS, ctx->stream, ioth, d_active,
                           d_skip, d_Q, fc);
      }
      cur ^= 1;
    }
    DM_HIP(ctx, hipGetLastError());
    DM_TRY(dm_download(ctx, h_off.data(), d_off, sizeof(unsigned long long) * np));
    bool any = false;
    std::vector<int> newly_done;
    double dbg_max = 0.0;
    for (int p = 0; p < np; ++p) {
      double mo;
      std::memcpy(&mo, &h_off[p], sizeof(double));
      dbg_max = std::max(dbg_max, mo);
      int was = active[p];
      active[p] = (was && mo > tol_outer) ? 1 : 0;
      any |= active[p] != 0;
      if (was && !active[p] && cur == 1) newly_done.push_back(p);
    }
    if (getenv("DM_DEBUG")) fprintf(stderr, "[jacobi_herm] sweep %d offmax %.3e\n", sweep, dbg_max);
    // a problem that converged while its matrix lives in T: bring it home to C
    // (the transpose of a Hermitian matrix's conjugate is itself)
    if (!newly_done.empty()) {
      std::vector<int> only(np, 0);
      for (int p : newly_done) only[p] = 1;
      int* d_only = dm_ws_upload(ctx, only);
      if (!d_only) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_UTIL, jac_ctrans_kernel, dim3(tb, tb, np), dim3(256), 0, ctx->stream, d_tdT, d_only);
    }
    if (!any) { ++sweep; break; }
    DM_TRY(dm_upload(ctx, d_active, active.data(), sizeof(int) * np));
  }
  // problems still active at the sweep cap and living in T: copy home as well
  if (cur == 1) {
    bool any_left = false;
    for (int p = 0; p < np; ++p) any_left |= active[p] != 0;
    if (any_left) {
      int* d_act2 = dm_ws_upload(ctx, active);
      if (!d_act2) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_UTIL, jac_ctrans_kernel, dim3(tb, tb, np), dim3(256), 0, ctx->stream, d_tdT, d_act2);
    }
  }
  if (sweeps_out) *sweeps_out = sweep;
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_diag_kernel, dim3((maxn + 255) / 256, np), dim3(256), 0, ctx->stream, d_tdC, evals,
                     evals_stride);
  DM_HIP(ctx, hipGetLastError());
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// ===========================================================================
// sort the rows of a batch of matrices by a per-row key (device), stable
// ===========================================================================
int dm_sort_rows_by_key(dm_ctx* ctx, const std::vector<dm_jac_problem>& probs, double* key, int key_stride,
                        bool descending) {
  const int np = (int)probs.size();
  if (np == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  std::vector<jac_pdesc> pd(np);
  std::vector<size_t> toff(np);
  std::vector<int> nrows(np);
  size_t ttot = 0;
  int maxrows = 0, maxcols = 0;
  for (int p = 0; p < np; ++p) {
    pd[p] = jac_pdesc{probs[p].Z, probs[p].ld, probs[p].row0, probs[p].nrows, 0, probs[p].ncols, 0, 0};
    toff[p] = ttot;
    ttot += (size_t)probs[p].nrows * probs[p].ncols;
    nrows[p] = probs[p].nrows;
    maxrows = std::max(maxrows, probs[p].nrows);
    maxcols = std::max(maxcols, probs[p].ncols);
  }
  if (maxrows == 0) return DM_OK;
  jac_pdesc* d_pd = dm_ws_upload(ctx, pd);
  size_t* d_toff = dm_ws_upload(ctx, toff);
  int* d_nrows = dm_ws_upload(ctx, nrows);
  int* d_rank = dm_ws_alloc_t<int>(ctx, (size_t)np * key_stride);
  double* d_ks = dm_ws_alloc_t<double>(ctx, (size_t)np * key_stride);
  cplx* d_tmp = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(ttot, 1));
  if (!d_pd || !d_toff || !d_nrows || !d_rank || !d_ks || !d_tmp) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_rank_kernel, dim3((maxrows + 255) / 256, np), dim3(256), 0, ctx->stream, key, key_stride,
                     d_nrows, d_rank, descending ? 1 : 0);
  const int gx = std::max(1, std::min(8, (maxcols + 255) / 256));
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_gather_rows_kernel, dim3(gx, maxrows, np), dim3(256), 0, ctx->stream, d_pd, d_rank,
                     key_stride, d_tmp, d_toff, key, d_ks);
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, jac_scatter_back_kernel, dim3(gx, maxrows, np), dim3(256), 0, ctx->stream, d_pd, d_tmp, d_toff);
  // sorted keys back into `key` (only the first nrows entries of each problem)
  {
    std::vector<dm_cdesc> cp;
    for (int p = 0; p < np; ++p)
      if (nrows[p] > 0)
        cp.push_back(dm_cdesc{d_ks + (size_t)p * key_stride, key + (size_t)p * key_stride, sizeof(double) * nrows[p]});
    DM_TRY(dm_copy_batched(ctx, cp));
  }
  DM_HIP(ctx, hipGetLastError());
  dm_ws_release(ctx, mark);
  return DM_OK;
}
