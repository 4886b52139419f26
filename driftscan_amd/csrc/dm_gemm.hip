// dm_gemm.hip — grouped complex128 GEMM on the fp64 matrix cores (gfx950).
//
//   C[M x N] = alpha * opA(A)[M x K] * diag(kscale) * opB(B)[K x N] + beta * C
//
// One launch serves an arbitrary list of problems ("grouped GEMM"): the host
// flattens every problem into 64x64 output tiles, so a single dispatch covers
// e.g. all (f,f') covariance blocks of all m at once — the per-m blocks of
// driftscan are far too small to fill 256 CUs one at a time.
//
// Replaces the numpy GEMM call sites of the reference:
//   beamtransfer.py:1186  (B_f * C_l) B_f'^H          project_matrix_sky_to_svd
//   beamtransfer.py:1226  (U_f * N) U_f^H             project_matrix_diagonal_telescope_to_svd
//   doublekl.py:73-74,80  E C E^H, E2^H E
//   and serves the trailing updates of the blocked Cholesky / triangular solves.
//
// Kernel shape: 256 threads = 4 waves, each wave owns a 32x32 block of the tile
// as 2x2 MFMA tiles of v_mfma_f64_16x16x4_f64.  A complex product is four real
// MFMAs (re*re, -im*im, re*im, im*re) — the 3M trick is avoided on purpose: its
// cancellation would cost the 1e-10 parity on small eigenvalues.  Operand panels
// are staged through LDS as separate re/im planes, [64][16] with a 17-double row
// pitch so that both the staging writes and the fragment reads are bank-conflict
// free for ds_{read,write}_b64.  Global loads for panel k+1 are issued before the
// MFMAs of panel k (register prefetch).
#include "dm_common.h"
#include "dm_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = 17;
#ifndef ZG_CH
#define ZG_CH 2
#endif
#ifndef ZG_OCC
#define ZG_OCC 2
#endif

template <bool B_REAL, bool B_GATHER>
__global__ __launch_bounds__(256, ZG_OCC) void zgemm_grouped_kernel(const dm_gemm_desc* __restrict__ descs,
                                                            const dm_gemm_tile* __restrict__ tiles,
                                                            int ntiles) {
  __shared__ double As_re[BM * LDP], As_im[BM * LDP];
  __shared__ double Bs_re[BN * LDP], Bs_im[B_REAL ? 1 : BN * LDP];

  const int bid = dm_xcd_remap(blockIdx.x, ntiles);
  const dm_gemm_tile t = tiles[bid];
  const dm_gemm_desc d = descs[t.desc];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = t.tm * BM, n0 = t.tn * BN;
  // 16x16 MFMA blocks of this wave that lie inside the matrix: ragged edges (the per-frequency
  // blocks of the covariance projections are <= 92 wide) skip the MFMAs of the padding
  bool vi[2], vj[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    vi[i] = m0 + wm * 32 + i * 16 < d.M;
    vj[i] = n0 + wn * 32 + i * 16 < d.N;
  }

  const cplx* __restrict__ A = reinterpret_cast<const cplx*>(d.A);
  const cplx* __restrict__ Bc = reinterpret_cast<const cplx*>(d.B);
  const double* __restrict__ Br = reinterpret_cast<const double*>(d.B);
  const bool conjA = d.flags & DM_GEMM_CONJ_A;
  const bool conjB = d.flags & DM_GEMM_CONJ_B;
  constexpr bool gatherB = B_GATHER;  // separate instantiation: the common kernel keeps its register budget
  // loader mapping: walk the contiguous direction with consecutive lanes
  const bool a_kfast = d.csA <= d.rsA;
  const bool b_kfast = d.rsB <= d.csB;

  dm_f64x4 acc_re[2][2], acc_im[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc_re[i][j] = dm_f64x4{0, 0, 0, 0};
      acc_im[i][j] = dm_f64x4{0, 0, 0, 0};
    }

  cplx ra[4], rb[4];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      int gm = m0 + m, gk = k0 + k;
      cplx v = make_double2(0.0, 0.0);
      if (gm < d.M && gk < d.K) {
        v = dm_ldg(A, (size_t)gm * d.rsA + (size_t)gk * d.csA);
        if (conjA) v.y = -v.y;
        if (d.kscale && !gatherB) { double s = dm_ldg(d.kscale, gk); v.x *= s; v.y *= s; }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int n, k;
      if (b_kfast) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
      int gn = n0 + n, gk = k0 + k;
      cplx v = make_double2(0.0, 0.0);
      if (gn < d.N && gk < d.K) {
        size_t off = (size_t)gk * d.rsB + (size_t)gn * d.csB;
        if (B_REAL) {
          v.x = dm_ldg(Br, off);
        } else if (gatherB) {
          const int2 g = d.bgather[gn];
          v = dm_ldg(Bc, (size_t)g.x + (size_t)gk * d.rsB);
          if (conjB) v.y = -v.y;
          const double s = dm_ldg(d.kscale, (size_t)g.y + gk);
          v.x *= s;
          v.y *= s;
        } else {
          v = dm_ldg(Bc, off);
          if (conjB) v.y = -v.y;
        }
      }
      rb[i] = v;
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      As_re[m * LDP + k] = ra[i].x;
      As_im[m * LDP + k] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int n, k;
      if (b_kfast) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
      Bs_re[n * LDP + k] = rb[i].x;
      if (!B_REAL) Bs_im[n * LDP + k] = rb[i].y;
    }
  };

  const int fi = lane & 15, fk = lane >> 4;
  // MFMAs of one staged panel, ONE ACCUMULATOR AT A TIME: the eight products of a panel that feed an accumulator (four
  // k-sub-steps x two real products) are issued back to back.  v_mfma_f64_16x16x4_f64 pays ~40 cycles whenever the
  // accumulator changes (its 8 VGPRs of C go through the register file) and none when SrcC is the previous MFMA's
  // result: scratch/mfma_peak3.hip measures 105 cycles per MFMA with the accumulators taken in rotation (47 TFLOP/s,
  // what rounds 1-4 took for the instruction's ceiling), 83 in chains of 2, 75 in chains of 4, 69.5 in chains of 8, 66
  // in chains of 16+ (75 TFLOP/s, the datasheet rate).  (Round 1 issued the two products of an accumulator eight MFMAs
  // apart "to avoid the stall of a dependent MFMA" — the opposite of what the hardware wants.)
  auto compute_panel = [&]() {
    // ZG_CH k-sub-steps per chain group: 2 x ZG_CH chained MFMAs per accumulator (ZG_CH = 4: the whole panel, 64 operand VGPRs)
#pragma unroll
    for (int k0 = 0; k0 < BK / 4; k0 += ZG_CH) {
      double a_re[2][ZG_CH], a_im[2][ZG_CH], b_re[2][ZG_CH], b_im[2][ZG_CH];
#pragma unroll
      for (int kk = 0; kk < ZG_CH; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int r = (wm * 32 + i * 16 + fi) * LDP + (k0 + kk) * 4 + fk;
          a_re[i][kk] = As_re[r];
          a_im[i][kk] = As_im[r];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int r = (wn * 32 + j * 16 + fi) * LDP + (k0 + kk) * 4 + fk;
          b_re[j][kk] = Bs_re[r];
          b_im[j][kk] = B_REAL ? 0.0 : Bs_im[r];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (vi[i] && vj[j]) {
#pragma unroll
            for (int kk = 0; kk < ZG_CH; ++kk) {
              acc_re[i][j] = dm_mfma(a_re[i][kk], b_re[j][kk], acc_re[i][j]);
              if (!B_REAL) acc_re[i][j] = dm_mfma(-a_im[i][kk], b_im[j][kk], acc_re[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the scheduler would interleave the chains again (it models a dependent MFMA as a stall)
#pragma unroll
            for (int kk = 0; kk < ZG_CH; ++kk) {
              acc_im[i][j] = dm_mfma(a_im[i][kk], b_re[j][kk], acc_im[i][j]);
              if (!B_REAL) acc_im[i][j] = dm_mfma(a_re[i][kk], b_im[j][kk], acc_im[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    }
  };

  cplx* __restrict__ C = reinterpret_cast<cplx*>(d.C);
  const int crow = lane >> 4, ccol = lane & 15;
  const bool rmw = d.beta != 0.0;
  cplx cold[2][2][4];
  // For beta != 0 the sixteen C values are fetched back to back from clamped (always valid)
  // addresses, and BEFORE the MFMAs of the last panel, so that their latency is covered.
  auto load_c = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = min(m0 + wm * 32 + i * 16 + crow + 4 * r, d.M - 1);
          const int gn = min(n0 + wn * 32 + j * 16 + ccol, d.N - 1);
          cold[i][j][r] = dm_ldg(C, (size_t)gm * d.ldc + (size_t)gn * d.csc);
        }
  };

  const int nk = (d.K + BK - 1) / BK;
  if (nk > 0) load_tiles(0);
  for (int kt = 0; kt < nk - 1; ++kt) {
    __syncthreads();  // previous panel fully consumed
    store_tiles();
    __syncthreads();
    load_tiles((kt + 1) * BK);  // in flight during the MFMAs below
    compute_panel();
  }
  if (nk > 0) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    compute_panel();
  }
  // the old C values of a read-modify-write tile are requested together AFTER the last panel: held across it they cost 64
  // VGPRs (spills at three workgroups per CU); the other workgroups of the CU cover the one memory latency this exposes
  if (rmw) load_c();

  // epilogue: lane l, reg r -> row (l>>4) + 4r, col l&15 of each 16x16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int gm = m0 + wm * 32 + i * 16 + crow + 4 * r;
        int gn = n0 + wn * 32 + j * 16 + ccol;
        const double are = acc_re[i][j][r], aim = acc_im[i][j][r];
        cplx v = make_double2(d.alpha * are - d.alpha_im * aim, d.alpha * aim + d.alpha_im * are);
        if (rmw) {
          v.x += d.beta * cold[i][j][r].x;
          v.y += d.beta * cold[i][j][r].y;
        }
        if (gm < d.M && gn < d.N) dm_stg(C, (size_t)gm * d.ldc + (size_t)gn * d.csc, v);
      }
}


// ---- register-only variant on v_mfma_f64_4x4x4_4b_f64 ------------------------------------------------------
// The 16x16x4 form reaches the datasheet rate only in long chains on one accumulator (64 + ~41 / R cycles for R chained
// MFMAs, scratch/mfma_peak3.hip; rounds 1-4 read its 47 TFLOP/s with rotating accumulators as a ceiling), the 4x4x4 form
// (four independent 4x4x4 products per instruction) issues at the full rate in any order; it needs four times the
// operand values per flop, which an LDS-staged loop pays in barriers and ds_read latency.  (Round 5 re-measured the
// LDS-staged 16x16x4 kernel above WITH per-accumulator chains: 47 TFLOP/s at 2048^3 against 39.6 before and 52.5 here
// — that kernel is bound by its barriers, not by the instruction.)  Here there is no LDS and no barrier at all: every lane loads the one
// A element and the one B element the instruction wants from it straight from L1 / L2 (the panels of a 64x64 tile
// are shared by the four waves of the workgroup through the vector L1), prefetched one K-step of 4 ahead.
//   lane l = 16 k + 4 g + t:  A operand = A[row 4 g + t][k],  B operand = B[k][col 4 g' + t]
//   rotation s (s = 0..3) pairs row block g with column block g' = (g + s) & 3, so four instructions cover a
//   16x16 tile; D of rotation s lands in lane 16 i + 4 g + j = C[4 g + i][4 ((g + s) & 3) + j].
// The rotated B operands are the rotation-0 registers moved by DPP row_ror inside each 16-lane row (same k).
// NJ = 16-column sub-tiles of B per wave: 2 (a 32 x 32 block per wave, 64 x 64 per workgroup) or 4 (32 x 64 per wave, 64 x 128
// per workgroup — for the real-B products, whose two MFMAs per output block and K-step would otherwise sit behind the same four
// loads as the four of a complex product: twice the B columns per wave restore the MFMA-to-load ratio of the complex kernel)
template <bool B_REAL, bool B_GATHER, int PD, int NJ = 2>
__global__ __launch_bounds__(256, 2) void zgemm4_grouped_kernel(const dm_gemm_desc* __restrict__ descs,
                                                                const dm_gemm_tile* __restrict__ tiles, int ntiles,
                                                                int xchunk) {
  const int bid = xchunk > 0 ? dm_xcd_remap_chunked(blockIdx.x, ntiles, xchunk) : dm_xcd_remap(blockIdx.x, ntiles);
  const dm_gemm_tile tl = tiles[bid];
  const dm_gemm_desc d = descs[tl.desc];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int k = lane >> 4, g = (lane >> 2) & 3, t = lane & 3;
  // square tile: 2 x 2 waves of 32 x 32; flat tile (tm < 0, unit -tm - 1): 32 rows x 128 columns, one wave per 32
  // columns — ragged row counts (the per-frequency blocks of 70-92 modes) then waste at most 15 rows instead of 63
  const bool flat = tl.tm < 0;
  const int mrow = flat ? (-tl.tm - 1) * 32 : tl.tm * BM + wm * 32;
  const int ncol = flat ? tl.tn * (64 * NJ) + wave * (16 * NJ) : tl.tn * (32 * NJ) + wn * (16 * NJ);
  if (mrow >= d.M || ncol >= d.N) return;  // no LDS, no barrier: a wave without outputs just leaves
  const cplx* __restrict__ A = reinterpret_cast<const cplx*>(d.A);
  const cplx* __restrict__ Bc = reinterpret_cast<const cplx*>(d.B);
  const double* __restrict__ Br = reinterpret_cast<const double*>(d.B);
  const double sa = (d.flags & DM_GEMM_CONJ_A) ? -1.0 : 1.0, sb = (d.flags & DM_GEMM_CONJ_B) ? -1.0 : 1.0;
  // this lane's rows / columns of the two 16-wide sub-tiles (clamped; masked by zeroing the operand)
  size_t aoff[2], boff[NJ], ksoff[NJ];
  bool aok[2], bok[NJ], tile_i[2], tile_j[NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = mrow + i * 16 + 4 * g + t;
    aok[i] = r < d.M;
    tile_i[i] = mrow + i * 16 < d.M;
    aoff[i] = (size_t)min(r, d.M - 1) * d.rsA;
  }
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int c = ncol + i * 16 + 4 * g + t;
    bok[i] = c < d.N;
    tile_j[i] = ncol + i * 16 < d.N;
    const int cc = min(c, d.N - 1);
    ksoff[i] = 0;
    if (B_GATHER) {
      const int2 gg = d.bgather[cc];
      boff[i] = (size_t)gg.x;
      ksoff[i] = (size_t)gg.y;
    } else {
      boff[i] = (size_t)cc * d.csB;
    }
  }
  double acc_re[2][NJ][4], acc_im[2][NJ][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc_re[i][j][s] = acc_im[i][j][s] = 0.0;

  // Operands travel as RAW loads: no arithmetic touches a fragment before its compute step, so the loads of step
  // t + 1 stay in flight under the 64 MFMAs of step t (an `s_waitcnt` lands wherever the first use is — conjugation
  // signs, masks and scale factors are therefore applied at compute time).
  struct raw { cplx a[2], b[NJ]; double ks, kb[NJ]; };
  // tiles that lie inside the matrix with nothing to scale take the lean steps: no masks, no scale loads
  const bool plain = !d.kscale && !B_GATHER && mrow + 32 <= d.M && ncol + 16 * NJ <= d.N;
  auto load_lean = [&](int k0) {
    raw f;
    const size_t kk = (size_t)(k0 + k);
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = dm_ldg(A, aoff[i] + kk * d.csA);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (B_REAL) f.b[j] = make_double2(dm_ldg(Br, kk * d.rsB + boff[j]), 0.0);
      else f.b[j] = dm_ldg(Bc, boff[j] + kk * d.rsB);
    }
    f.ks = 1.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) f.kb[j] = 1.0;
    return f;
  };
  auto load_gen = [&](int k0) {
    raw f;
    const size_t kc = (size_t)min(k0 + k, d.K - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = dm_ldg(A, aoff[i] + kc * d.csA);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (B_REAL) f.b[j] = make_double2(dm_ldg(Br, kc * d.rsB + boff[j]), 0.0);
      else f.b[j] = dm_ldg(Bc, boff[j] + kc * d.rsB);
      f.kb[j] = B_GATHER ? dm_ldg(d.kscale, ksoff[j] + kc) : 1.0;
    }
    f.ks = (d.kscale && !B_GATHER) ? dm_ldg(d.kscale, kc) : 1.0;
    return f;
  };
  auto mfmas = [&](const cplx (&fa)[2], const cplx (&fb)[NJ]) {
    double bre[NJ][4], bim[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bre[j][0] = fb[j].x;
      bim[j][0] = fb[j].y;
      // rotation s needs the operand of the lane 4 s further along its 16-lane row: row_ror:n hands lane i the
      // value of lane (i - n) mod 16, so "4 s further" is a rotation by 16 - 4 s (checked against the LDS kernel)
      bre[j][1] = dm_dpp_f64<0x12C>(fb[j].x); bim[j][1] = dm_dpp_f64<0x12C>(fb[j].y);
      bre[j][2] = dm_dpp_f64<0x128>(fb[j].x); bim[j][2] = dm_dpp_f64<0x128>(fb[j].y);
      bre[j][3] = dm_dpp_f64<0x124>(fb[j].x); bim[j][3] = dm_dpp_f64<0x124>(fb[j].y);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (!(tile_i[i] && tile_j[j])) continue;
        if (!B_REAL) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc_re[i][j][s] = dm_mfma4(fa[i].x, bre[j][s], acc_re[i][j][s]);
            acc_im[i][j][s] = dm_mfma4(fa[i].x, bim[j][s], acc_im[i][j][s]);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc_re[i][j][s] = dm_mfma4(-fa[i].y, bim[j][s], acc_re[i][j][s]);
            acc_im[i][j][s] = dm_mfma4(fa[i].y, bre[j][s], acc_im[i][j][s]);
          }
        } else {  // real B: two products per output block, not four
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc_re[i][j][s] = dm_mfma4(fa[i].x, bre[j][s], acc_re[i][j][s]);
            acc_im[i][j][s] = dm_mfma4(fa[i].y, bre[j][s], acc_im[i][j][s]);
          }
        }
      }
  };
  auto compute_lean = [&](const raw& f) {
    cplx fa[2], fb[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = make_double2(f.a[i].x, f.a[i].y * sa);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j] = make_double2(f.b[j].x, f.b[j].y * sb);
    mfmas(fa, fb);
  };
  auto compute_gen = [&](const raw& f, int k0) {
    const bool kv = k0 + k < d.K;
    cplx fa[2], fb[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool on = kv && aok[i];
      fa[i] = on ? make_double2(f.a[i].x * f.ks, f.a[i].y * f.ks * sa) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool on = kv && bok[j];
      fb[j] = on ? make_double2(f.b[j].x * f.kb[j], f.b[j].y * f.kb[j] * sb) : make_double2(0.0, 0.0);
    }
    mfmas(fa, fb);
  };
  if (plain) {
    const int nfull = d.K / 4;
    if (PD == 1) {
      // one step ahead: 144 VGPRs, three waves per SIMD — the better trade for short K and ragged tiles
      if (nfull > 0) {
        raw cur = load_lean(0);
        for (int kt = 0; kt + 1 < nfull; ++kt) {
          const raw nxt = load_lean((kt + 1) * 4);  // in flight under the MFMAs of this step
          compute_lean(cur);
          cur = nxt;
        }
        compute_lean(cur);
      }
    } else if (nfull == 1) {
      compute_lean(load_lean(0));
    } else if (nfull > 1) {
      // two steps of operands in flight, three buffers rotating through an unrolled-by-three loop (no copies):
      // 184 VGPRs, two waves per SIMD — +3-4 % on K >= 512, -5-8 % on K = 64..128
      raw b0 = load_lean(0), b1 = load_lean(4), b2;
      int t = 0;
      for (; t + 5 <= nfull; t += 3) {
        b2 = load_lean((t + 2) * 4);
        compute_lean(b0);
        b0 = load_lean((t + 3) * 4);
        compute_lean(b1);
        b1 = load_lean((t + 4) * 4);
        compute_lean(b2);
      }
      const int rem = nfull - t;  // 2, 3 or 4 steps left; b0, b1 hold the first two
      if (rem >= 3) b2 = load_lean((t + 2) * 4);
      compute_lean(b0);
      if (rem >= 4) b0 = load_lean((t + 3) * 4);
      compute_lean(b1);
      if (rem >= 3) compute_lean(b2);
      if (rem >= 4) compute_lean(b0);
    }
    if (nfull * 4 < d.K) compute_gen(load_gen(nfull * 4), nfull * 4);
  } else {
    const int nk = (d.K + 3) / 4;
    if (nk > 0) {
      raw cur = load_gen(0);
      for (int kt = 0; kt + 1 < nk; ++kt) {
        const raw nxt = load_gen((kt + 1) * 4);
        compute_gen(cur, kt * 4);
        cur = nxt;
      }
      compute_gen(cur, (nk - 1) * 4);
    }
  }
  // epilogue: lane 16 i' + 4 g + j' of rotation s holds C[4 g + i'][4 ((g + s) & 3) + j']
  // For beta != 0 the old values of EIGHT outputs (one 16-row half of the wave's block) are requested together from
  // clamped, always valid addresses and waited for once: element by element (load, wait, combine, store) a wave paid
  // sixteen memory latencies per tile — more than the K loop of a rank-64 update takes.
  cplx* __restrict__ C = reinterpret_cast<cplx*>(d.C);
  const int ip = lane >> 4, jp = lane & 3;
  const bool rmw = d.beta != 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    size_t off[NJ][4];
    bool ok[NJ][4];
    cplx old[NJ][4];
    const int gm = mrow + i * 16 + 4 * g + ip;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int gn = ncol + j * 16 + 4 * ((g + s) & 3) + jp;
        ok[j][s] = gm < d.M && gn < d.N;
        off[j][s] = (size_t)min(gm, d.M - 1) * d.ldc + (size_t)min(gn, d.N - 1) * d.csc;
        old[j][s] = make_double2(0.0, 0.0);
      }
    if (rmw) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) old[j][s] = dm_ldg(C, off[j][s]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double are = acc_re[i][j][s], aim = acc_im[i][j][s];
        cplx v = make_double2(d.alpha * are - d.alpha_im * aim, d.alpha * aim + d.alpha_im * are);
        if (rmw) {
          v.x += d.beta * old[j][s].x;
          v.y += d.beta * old[j][s].y;
        }
        if (ok[j][s]) dm_stg(C, off[j][s], v);
      }
  }
}

// ---- all-real variant: C[M x N] (double) = alpha * A * B + beta * C, same tiling, one MFMA per
// (tile, k-step).  Used by the divide & conquer eigenvector updates (real orthogonal matrices).
__global__ __launch_bounds__(256) void dgemm_grouped_kernel(const dm_gemm_desc* __restrict__ descs,
                                                            const dm_gemm_tile* __restrict__ tiles, int ntiles, int xchunk) {
  __shared__ double As[BM * LDP], Bs[BN * LDP];
  const int bid = xchunk > 0 ? dm_xcd_remap_chunked(blockIdx.x, ntiles, xchunk) : dm_xcd_remap(blockIdx.x, ntiles);
  const dm_gemm_tile t = tiles[bid];
  const dm_gemm_desc d = descs[t.desc];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = t.tm * BM, n0 = t.tn * BN;
  const double* __restrict__ A = reinterpret_cast<const double*>(d.A);
  const double* __restrict__ B = reinterpret_cast<const double*>(d.B);
  const bool a_kfast = d.csA <= d.rsA;
  const bool b_kfast = d.rsB <= d.csB;
  dm_f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = dm_f64x4{0, 0, 0, 0};
  double ra[4], rb[4];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < d.M && gk < d.K) ? dm_ldg(A, (size_t)gm * d.rsA + (size_t)gk * d.csA) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int n, k;
      if (b_kfast) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
      int gn = n0 + n, gk = k0 + k;
      rb[i] = (gn < d.N && gk < d.K) ? dm_ldg(B, (size_t)gk * d.rsB + (size_t)gn * d.csB) : 0.0;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      As[m * LDP + k] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      int n, k;
      if (b_kfast) { k = idx & 15; n = idx >> 4; } else { n = idx & 63; k = idx >> 6; }
      Bs[n * LDP + k] = rb[i];
    }
  };
  const int fi = lane & 15, fk = lane >> 4;
  const int nk = (d.K + BK - 1) / BK;
  if (nk > 0) load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    // the four MFMAs of a panel that feed an accumulator back to back (chains on one accumulator: 75 cycles per
    // v_mfma_f64_16x16x4_f64 against 105 with the four accumulators in rotation, scratch/mfma_peak3.hip)
    double a[2][BK / 4], b[2][BK / 4];
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i][kk] = As[(wm * 32 + i * 16 + fi) * LDP + kk * 4 + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j][kk] = Bs[(wn * 32 + j * 16 + fi) * LDP + kk * 4 + fk];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) acc[i][j] = dm_mfma(a[i][kk], b[j][kk], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  double* __restrict__ C = reinterpret_cast<double*>(d.C);
  const int crow = lane >> 4, ccol = lane & 15;
  const bool rmw = d.beta != 0.0;
  double cold[2][2][4];
  if (rmw) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = min(m0 + wm * 32 + i * 16 + crow + 4 * r, d.M - 1);
          const int gn = min(n0 + wn * 32 + j * 16 + ccol, d.N - 1);
          cold[i][j][r] = dm_ldg(C, (size_t)gm * d.ldc + gn);
        }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int gm = m0 + wm * 32 + i * 16 + crow + 4 * r;
        int gn = n0 + wn * 32 + j * 16 + ccol;
        double v = d.alpha * acc[i][j][r];
        if (rmw) v += d.beta * cold[i][j][r];
        if (gm < d.M && gn < d.N) dm_stg(C, (size_t)gm * d.ldc + gn, v);
      }
}

}  // namespace

// Host side: flatten descriptors into tiles (plan), upload, launch (run).  A plan is the host image of one grouped
// launch — descriptors and tile lists in ONE blob; a caller with a chain of launches (a panel of the two-stage
// tridiagonalisation: a dozen of them) builds all its plans, sends the blobs in one copy and runs them in turn: every
// staged upload is a copy kernel of its own in the stream, in front of the launch that needs it.
int dm_gemm_plan_build(const std::vector<dm_gemm_desc>& descs, dm_gemm_plan& plan) {
  plan = dm_gemm_plan();
  if (descs.empty()) return DM_OK;
  // DM_GEMM4 = 1 / 0 selects the register-only 4x4x4 kernel (default) / the LDS-staged 16x16x4 kernel
  static const int use4_env = getenv("DM_GEMM4") ? atoi(getenv("DM_GEMM4")) : -1;
  static const bool noflat = getenv("DM_GEMM4_NOFLAT") != nullptr;
  const bool use4 = use4_env >= 0 ? use4_env != 0 : true;
  std::vector<dm_gemm_tile> tiles;
  std::vector<dm_gemm_tile> tiles_real;
  std::vector<dm_gemm_tile> tiles_dd;
  std::vector<dm_gemm_tile> tiles_gat;
  double fl_g = 0.0;
  double fl_c = 0.0, fl_r = 0.0, fl_d = 0.0;
  // The real-B class of the register-only kernel runs 64 x 128 tiles (32 x 64 per wave, zgemm4_grouped_kernel<.., NJ = 4>)
  // when most of the launch's work sits in products wide enough for them — decided per launch (measured: a configs[2] BT-gen rank call 5.16 -> 4.83 s
  // with N = 170 .. 513, configs[1] with N <= 129 loses 5 % to the padding of the wide tiles)
  double w_all = 0.0, w_wide = 0.0;   // work of the real-B products, and of those at least 160 columns wide
  for (const auto& d : descs)
    if (d.M > 0 && d.N > 0 && (d.flags & DM_GEMM_B_REAL) && !(d.flags & DM_GEMM_ALL_REAL)) {
      const double w = (double)d.M * d.N * d.K;
      w_all += w;
      if (d.N >= 160) w_wide += w;
    }
  const bool wide_r = use4 && w_all > 0.0 && w_wide >= 0.6 * w_all;
  for (size_t i = 0; i < descs.size(); ++i) {
    const dm_gemm_desc& d = descs[i];
    if (d.M <= 0 || d.N <= 0) continue;
    if ((d.flags & DM_GEMM_B_REAL) && (d.flags & (DM_GEMM_LOWER | DM_GEMM_UPPER))) return DM_EARG;  // no triangular real-B products
    const bool wide = wide_r && (d.flags & DM_GEMM_B_REAL) && !(d.flags & DM_GEMM_ALL_REAL);
    const int BNd = wide ? 2 * BN : BN, FNd = wide ? 256 : 128;
    int tm = (d.M + BM - 1) / BM, tn = (d.N + BNd - 1) / BNd;
    // flat 32 x 128 tiles of the register-only kernel where the 32-row granularity saves work and the problem is small
    if (use4 && !noflat && !(d.flags & (DM_GEMM_ALL_REAL | DM_GEMM_LOWER | DM_GEMM_UPPER)) && ((d.M + 31) / 32) * 32 < tm * BM &&
        (d.M <= 32 || d.N <= 256)) {  // measured: 32-row panels 36 -> 50 TFLOP/s, 92 x 92 x 129 26 -> 33.5; wide ragged ones lose 5 %
      const int um = (d.M + 31) / 32, un = (d.N + FNd - 1) / FNd;
      for (int a = 0; a < um; ++a)
        for (int b = 0; b < un; ++b) {
          dm_gemm_tile t{(int)i, -a - 1, b};
          const double rows = std::min(32, d.M - a * 32), cols = std::min(FNd, d.N - b * FNd);
          if (d.flags & DM_GEMM_B_REAL) { tiles_real.push_back(t); fl_r += 4.0 * rows * cols * d.K; }
          else if (d.flags & DM_GEMM_B_GATHER) { tiles_gat.push_back(t); fl_g += 8.0 * rows * cols * d.K; }
          else { tiles.push_back(t); fl_c += 8.0 * rows * cols * d.K; }
        }
      continue;
    }
    // tile order inside a problem: groups of GM tile rows, column by column within a group — 64 consecutive tiles (what an
    // XCD's 32 CUs hold at two workgroups each, and one chunk of the XCD remap) are then an 8 x 8 block of tiles that shares
    // 8 + 8 operand panels instead of the 1 + 64 of a row of tiles
    static const int GM = getenv("DM_GEMM_GROUPM") ? std::max(1, atoi(getenv("DM_GEMM_GROUPM"))) : 8;
    // (not for the gathered-B products: their two tile rows side by side cost the covariance projections 10 %, measured)
    const int gm = (d.flags & DM_GEMM_B_GATHER) ? 1 : GM;
    for (int a0 = 0; a0 < tm; a0 += gm)
     for (int b = 0; b < tn; ++b)
      for (int a = a0; a < std::min(tm, a0 + gm); ++a) {
        if ((d.flags & DM_GEMM_LOWER) && b > a) continue;
        if ((d.flags & DM_GEMM_UPPER) && b < ((d.flags & DM_GEMM_UPPER128) ? (a & ~1) : a)) continue;
        dm_gemm_tile t{(int)i, a, b};
        const double rows = std::min(BM, d.M - a * BM), cols = std::min(BNd, d.N - b * BNd);
        if (d.flags & DM_GEMM_ALL_REAL) { tiles_dd.push_back(t); fl_d += 2.0 * rows * cols * d.K; }
        else if (d.flags & DM_GEMM_B_REAL) { tiles_real.push_back(t); fl_r += 4.0 * rows * cols * d.K; }
        else if (d.flags & DM_GEMM_B_GATHER) { tiles_gat.push_back(t); fl_g += 8.0 * rows * cols * d.K; }
        else { tiles.push_back(t); fl_c += 8.0 * rows * cols * d.K; }
      }
  }
  // longest tiles first: a launch mixes problems of very different K, and a long tile that
  // starts last sets the duration of the launch (stable, so tiles of one problem stay together)
  auto by_k = [&](const dm_gemm_tile& a, const dm_gemm_tile& b) { return descs[a.desc].K > descs[b.desc].K; };
  for (auto* tl : {&tiles, &tiles_real, &tiles_dd, &tiles_gat})
    if (!std::is_sorted(tl->begin(), tl->end(), by_k)) std::stable_sort(tl->begin(), tl->end(), by_k);
  // descriptors and all tile lists travel in ONE host-to-device copy: on the chains of short
  // dependent products (triangular solves, Cholesky) the API calls per launch are what the GPU waits for
  const size_t nd = descs.size(), n0 = tiles.size(), n1 = tiles_real.size(), n2 = tiles_dd.size(), n3 = tiles_gat.size();
  const size_t desc_bytes = (nd * sizeof(dm_gemm_desc) + 15) & ~size_t(15);
  const size_t tile_bytes = (n0 + n1 + n2 + n3) * sizeof(dm_gemm_tile);
  plan.blob.resize(desc_bytes + tile_bytes);
  std::memcpy(plan.blob.data(), descs.data(), nd * sizeof(dm_gemm_desc));
  {
    char* tp = plan.blob.data() + desc_bytes;
    for (const auto* tl : {&tiles, &tiles_real, &tiles_dd, &tiles_gat}) {
      if (!tl->empty()) std::memcpy(tp, tl->data(), tl->size() * sizeof(dm_gemm_tile));
      tp += tl->size() * sizeof(dm_gemm_tile);
    }
  }
  plan.desc_bytes = desc_bytes;
  plan.n0 = n0; plan.n1 = n1; plan.n2 = n2; plan.n3 = n3;
  plan.fl_c = fl_c; plan.fl_r = fl_r; plan.fl_d = fl_d; plan.fl_g = fl_g;
  plan.use4 = use4;
  plan.wide_r = wide_r;
  // deep operand prefetch (two waves per SIMD) pays on long inner dimensions only
  static const int deep_env = getenv("DM_GEMM4_DEEP") ? atoi(getenv("DM_GEMM4_DEEP")) : -1;
  int kmin_c = 1 << 30;
  for (const auto& d : descs)
    if (d.M > 0 && d.N > 0 && !(d.flags & (DM_GEMM_ALL_REAL | DM_GEMM_B_REAL | DM_GEMM_B_GATHER))) kmin_c = std::min(kmin_c, d.K);
  plan.deep = deep_env >= 0 ? deep_env != 0 : kmin_c >= 512;
  // figures for DM_GEMM_LOG
  plan.log_kmin = 1 << 30; plan.log_kmax = 0; plan.log_mmax = 0; plan.log_nmax = 0; plan.log_rmw = 0; plan.log_full = 0.0;
  for (const auto& d : descs) {
    if (d.M <= 0 || d.N <= 0 || (d.flags & (DM_GEMM_ALL_REAL | DM_GEMM_B_REAL))) continue;
    plan.log_kmin = std::min(plan.log_kmin, d.K); plan.log_kmax = std::max(plan.log_kmax, d.K);
    plan.log_mmax = std::max(plan.log_mmax, d.M); plan.log_nmax = std::max(plan.log_nmax, d.N);
    plan.log_rmw |= d.beta != 0.0;
  }
  for (const auto& t : tiles) plan.log_full += 8.0 * BM * BN * descs[t.desc].K;
  plan.log_ndesc = nd;
  return DM_OK;
}

// Upload the blobs of several plans in one staged copy; dev[i] = device address of plan i (nullptr for an empty plan).
int dm_gemm_plans_upload(dm_ctx* ctx, const std::vector<const dm_gemm_plan*>& plans, std::vector<const char*>& dev) {
  dev.assign(plans.size(), nullptr);
  size_t tot = 0;
  std::vector<size_t> off(plans.size(), 0);
  for (size_t i = 0; i < plans.size(); ++i) {
    off[i] = tot;
    tot += (plans[i]->blob.size() + 15) & ~size_t(15);
  }
  if (tot == 0) return DM_OK;
  std::vector<char> all(tot);
  for (size_t i = 0; i < plans.size(); ++i)
    if (!plans[i]->blob.empty()) std::memcpy(all.data() + off[i], plans[i]->blob.data(), plans[i]->blob.size());
  char* d = dm_ws_upload(ctx, all);
  if (!d) return DM_ENOMEM;
  for (size_t i = 0; i < plans.size(); ++i)
    if (!plans[i]->blob.empty()) dev[i] = d + off[i];
  return DM_OK;
}

int dm_gemm_plan_run(dm_ctx* ctx, const dm_gemm_plan& plan, const char* dblob) {
  if (plan.blob.empty()) return DM_OK;
  const bool use4 = plan.use4;
  const size_t n0 = plan.n0, n1 = plan.n1, n2 = plan.n2, n3 = plan.n3;
  const double fl_c = plan.fl_c, fl_r = plan.fl_r, fl_d = plan.fl_d, fl_g = plan.fl_g;
  const dm_gemm_desc* dd = reinterpret_cast<const dm_gemm_desc*>(dblob);
  const dm_gemm_tile* dtiles = reinterpret_cast<const dm_gemm_tile*>(dblob + plan.desc_bytes);
  const dm_gemm_tile* const dt_c = dtiles;
  const dm_gemm_tile* const dt_r = dtiles + n0;
  const dm_gemm_tile* const dt_d = dtiles + n0 + n1;
  const dm_gemm_tile* const dt_g = dtiles + n0 + n1 + n2;
  if (n0) {
    const dm_gemm_tile* dt = dt_c;
    const bool deep = plan.deep;
    static const bool log = getenv("DM_GEMM_LOG") != nullptr;  // debugging aid: per-launch shape and rate
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (log) {
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, ctx->stream);
    }
    static const int xchunk_c = getenv("DM_GEMM_XCHUNK") ? atoi(getenv("DM_GEMM_XCHUNK")) : 256;
    {
      dm_prof_scope ps(ctx, DM_PROF_GEMM, fl_c);
      if (use4)
        if (deep)
          hipLaunchKernelGGL((zgemm4_grouped_kernel<false, false, 2>), dim3((unsigned)n0), dim3(256), 0, ctx->stream,
                             dd, dt, (int)n0, xchunk_c);
        else
          hipLaunchKernelGGL((zgemm4_grouped_kernel<false, false, 1>), dim3((unsigned)n0), dim3(256), 0, ctx->stream,
                             dd, dt, (int)n0, xchunk_c);
      else
        hipLaunchKernelGGL((zgemm_grouped_kernel<false, false>), dim3((unsigned)n0), dim3(256), 0, ctx->stream, dd,
                           dt, (int)n0);
    }
    if (log) {
      (void)hipEventRecord(e1, ctx->stream);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      fprintf(stderr, "GEMMLOG tiles %6zu descs %5zu Mmax %5d Nmax %5d K %4d..%4d rmw %d  %8.3f ms  %6.2f TF (padded %6.2f)\n",
              n0, plan.log_ndesc, plan.log_mmax, plan.log_nmax, plan.log_kmin, plan.log_kmax, plan.log_rmw, ms, fl_c / ms / 1e9,
              plan.log_full / ms / 1e9);
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
  }
  if (n3) {
    const dm_gemm_tile* dt = dt_g;
    dm_prof_scope ps(ctx, DM_PROF_GEMM_COV, fl_g);
    // DM_COV_XCHUNK (and DM_GEMM_XCHUNK, DM_GEMMR_XCHUNK, DM_DGEMM_XCHUNK for the other classes): tiles per XCD chunk of the
    // launch (dm_xcd_remap_chunked); 0 = one contiguous run of the tile list per XCD, the rule of rounds 1-3.  Measured
    // (round 4, scratch/cov_chunk.sh, scratch/gemm_chunk.sh): covariance projections of a configs[2] share 40.0 -> 45.0
    // TFLOP/s, its grouped ZGEMM 11.0 -> 10.3 s, configs[1] step 129.3 -> 126 ms; flat between 4 and 1024 tiles per chunk.
    static const int xchunk = getenv("DM_COV_XCHUNK") ? atoi(getenv("DM_COV_XCHUNK")) : 64;
    if (use4)
      hipLaunchKernelGGL((zgemm4_grouped_kernel<false, true, 1>), dim3((unsigned)n3), dim3(256), 0, ctx->stream,
                         dd, dt, (int)n3, xchunk);
    else
      hipLaunchKernelGGL((zgemm_grouped_kernel<false, true>), dim3((unsigned)n3), dim3(256), 0, ctx->stream,
                         dd, dt, (int)n3);
  }
  if (n1) {
    const dm_gemm_tile* dt = dt_r;
    dm_prof_scope ps(ctx, DM_PROF_GEMM_REAL, fl_r);
    static const int xchunk_r = getenv("DM_GEMMR_XCHUNK") ? atoi(getenv("DM_GEMMR_XCHUNK")) : 64;
    if (use4 && plan.wide_r)
      hipLaunchKernelGGL((zgemm4_grouped_kernel<true, false, 1, 4>), dim3((unsigned)n1), dim3(256), 0, ctx->stream,
                         dd, dt, (int)n1, xchunk_r);
    else if (use4)
      hipLaunchKernelGGL((zgemm4_grouped_kernel<true, false, 1>), dim3((unsigned)n1), dim3(256), 0, ctx->stream,
                         dd, dt, (int)n1, xchunk_r);
    else
      hipLaunchKernelGGL((zgemm_grouped_kernel<true, false>), dim3((unsigned)n1), dim3(256), 0, ctx->stream,
                         dd, dt, (int)n1);
  }
  if (n2) {
    const dm_gemm_tile* dt = dt_d;
    dm_prof_scope ps(ctx, DM_PROF_DGEMM, fl_d);
    static const int xchunk_d = getenv("DM_DGEMM_XCHUNK") ? atoi(getenv("DM_DGEMM_XCHUNK")) : 64;
    hipLaunchKernelGGL(dgemm_grouped_kernel, dim3((unsigned)n2), dim3(256), 0, ctx->stream, dd, dt, (int)n2, xchunk_d);
  }
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_gemm_grouped_launch(dm_ctx* ctx, const std::vector<dm_gemm_desc>& descs) {
  if (descs.empty()) return DM_OK;
  dm_gemm_plan plan;
  DM_TRY(dm_gemm_plan_build(descs, plan));
  if (plan.blob.empty()) return DM_OK;
  // descriptors live in the bump arena until the caller's enclosing mark is released; kernels on the stream read them
  // asynchronously, so the arena is only rewound when the caller synchronises (see dm_ws_release contract)
  char* dblob = dm_ws_upload(ctx, plan.blob);
  if (!dblob) return DM_ENOMEM;
  return dm_gemm_plan_run(ctx, plan, dblob);
}
