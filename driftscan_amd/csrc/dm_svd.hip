// dm_svd.hip — the three-stage SVD compression of every (m, frequency) beam
// block, batched over all blocks handed in (drift/core/beamtransfer.py:802-924,
// BeamTransfer._generate_svdfile_m).
//
// The reference runs SVD1 -> project -> SVD2 (null space) -> project -> SVD3 ->
// project -> pinv, each a LAPACK call plus numpy GEMMs.  Here the whole chain is
// three phases of the one-sided block-Jacobi engine on ONE augmented matrix per
// (m, f):
//
//        Z = [ noisew * beam_m(f)  |  I_T ]        (T rows, P*L + T columns)
//
//   phase 1  rows 0..T      orthogonalised over all P*L sky columns   (SVD1, rtol 1e-10)
//   phase 2  rows 0..r1     orthogonalised over the polarised columns (SVD2, null space: rows cut2..r1)
//   phase 3  rows cut2..r1  orthogonalised over the T (pol 0) columns (SVD3, rtol 0)
//
// Because the unitary row mixing is applied to every column, after phase 3 the
// sky part of the surviving rows IS `beam = ut3 . bfr` (beamtransfer.py:877) and
// the identity part IS `ut3` (:866) — the projection GEMMs of the reference
// (:831, :850-851, :866, :877) disappear.  For unpolarised telescopes only
// phase 3 runs (:821-823).
//
// The pseudo-inverse (:887-921, scipy.linalg.pinv) is one more one-sided pass on
// [beam | I]: W beam = S V^H, so pinv(beam) = V S^-1 W = (S V^H)^H S^-2 W, a
// single grouped ZGEMM with the 1/s^2 weights on the contraction index.
#include "dm_common.h"
#include "dm_kernels.h"
#include "../../include/driftmi.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

// Per-chain geometry of the augmented matrices.  The blocks of an m carry exact zeros in the columns l < m (beam_m is
// stored with l padded from 0, beamtransfer.py:257-308): those columns add nothing to any inner product and stay zero
// under any row mixing, so a chain works on the COMPACT sky columns (p, l >= lmin) only — Lc = L - lmin per
// polarisation — and the products are scattered back into the padded layout at the end.
struct svd_geom {
  size_t zoff;   // element offset of the chain's Z (T rows x ldz)
  int lmin;      // first l kept
  int Lc;        // L - lmin
  int ldz;       // P * Lc + T
};

// Z[c] = [ nw[f] .* beam[c][:, (p, l >= lmin)] | I ]
__global__ void svd_build_z_kernel(const cplx* __restrict__ beam, const double* __restrict__ noisew,
                                   cplx* __restrict__ Z, const svd_geom* __restrict__ geo, int F, int T, int P, int L) {
  const int c = blockIdx.z;        // chain = blk * F + f
  const int f = c % F;
  const svd_geom g = geo[c];
  const int row = blockIdx.y;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= g.ldz) return;
  const int PLc = P * g.Lc;
  cplx v;
  if (col < PLc) {
    const int p = col / g.Lc, l = g.lmin + (col - p * g.Lc);
    const double w = noisew[(size_t)f * T + row];
    cplx b = beam[((size_t)c * T + row) * ((size_t)P * L) + (size_t)p * L + l];
    v = make_double2(b.x * w, b.y * w);
  } else {
    v = make_double2((col - PLc) == row ? 1.0 : 0.0, 0.0);
  }
  Z[g.zoff + (size_t)row * g.ldz + col] = v;
}

// scatter the surviving rows into the (zero-initialised) output products
__global__ void svd_extract_kernel(const cplx* __restrict__ Z, const svd_geom* __restrict__ geo,
                                   const int* __restrict__ row0, const int* __restrict__ nmodes,
                                   const double* __restrict__ noisew, const double* __restrict__ sig3,
                                   cplx* __restrict__ beam_svd, cplx* __restrict__ beam_ut, double* __restrict__ sigma,
                                   int F, int T, int P, int L, int K) {
  const int c = blockIdx.z;
  const int f = c % F;
  const int i = blockIdx.y;  // mode index
  if (i >= nmodes[c]) return;
  const svd_geom g = geo[c];
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int PLc = P * g.Lc;
  const cplx* z = Z + g.zoff + (size_t)(row0[c] + i) * g.ldz;
  if (col < PLc) {
    const int p = col / g.Lc, l = g.lmin + (col - p * g.Lc);   // the columns l < lmin of the (zero-filled) output stay zero
    beam_svd[((size_t)c * K + i) * ((size_t)P * L) + (size_t)p * L + l] = z[col];
  } else if (col < PLc + T) {
    const int t = col - PLc;
    const double w = noisew[(size_t)f * T + t];
    cplx u = z[col];
    beam_ut[((size_t)c * K + i) * T + t] = make_double2(u.x * w, u.y * w);
  }
  if (col == 0) sigma[(size_t)c * K + i] = sig3[(size_t)c * T + i];
}

// TALL chains (P (L - lmin) < T: more rows than sky columns — every m-block above m = L - T / P), SVD1: the T x T Gram
// eigenproblem of the row-side preconditioner has rank <= P Lc, and rounds 1-4 paid for all of it (864^3 against 452^3 at
// m = 400 of configs[2]).  The same singular triplets come out of the TRANSPOSED matrix: Yt = (w B)^H is P Lc x T, its
// rows are mixed until orthogonal over all T columns — Yt' = Sigma U^H: row i is sigma_i u_i^H — and the rows of the
// chain's Z follow as  [ u_i^H (w B) | u_i^H ]  (one product per polarisation); only r1 <= P Lc rows ever exist.
// Yt[c][k = p Lc + j][t] = conj( nw[f][t] beam[c][t][p][lmin + j] ), through a 32 x 32 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void svd_build_yt_kernel(const cplx* __restrict__ beam, const double* __restrict__ noisew,
                                                           cplx* __restrict__ Yt, const svd_geom* __restrict__ geo,
                                                           const size_t* __restrict__ yoff, const int* __restrict__ tall,
                                                           int F, int T, int P, int L) {
  __shared__ cplx tile[32][33];
  const int c = blockIdx.z;
  if (!tall[c]) return;
  const svd_geom g = geo[c];
  const int f = c % F;
  const int Kc = P * g.Lc;
  const int k0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  if (k0 >= Kc || t0 >= T) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int t = t0 + j, k = k0 + tx;
    cplx v = make_double2(0.0, 0.0);
    if (t < T && k < Kc) {
      const int p = k / g.Lc, l = g.lmin + (k - p * g.Lc);
      const double w = noisew[(size_t)f * T + t];
      const cplx b = beam[((size_t)c * T + t) * ((size_t)P * L) + (size_t)p * L + l];
      v = make_double2(b.x * w, -b.y * w);
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int k = k0 + j, t = t0 + tx;
    if (k < Kc && t < T) Yt[yoff[c] + (size_t)k * T + t] = tile[tx][j];
  }
}

// identity part of the rows of a tall chain: Z[c][i][P Lc + t] = Yt'[c][i][t] / sigma_i = u_i^H   (i < r1)
__global__ void svd_tall_rows_kernel(const cplx* __restrict__ Yt, const size_t* __restrict__ yoff,
                                     const int* __restrict__ tall, const int* __restrict__ r1,
                                     const double* __restrict__ sigt, cplx* __restrict__ Z,
                                     const svd_geom* __restrict__ geo, int T, int P) {
  const int c = blockIdx.z;
  if (!tall[c]) return;
  const int i = blockIdx.y;
  if (i >= r1[c]) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const svd_geom g = geo[c];
  const double s = sigt[(size_t)c * T + i];
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  const cplx y = Yt[yoff[c] + (size_t)i * T + t];
  Z[g.zoff + (size_t)i * g.ldz + (size_t)P * g.Lc + t] = make_double2(y.x * inv, y.y * inv);
}

// unpolarised tall chains (SVD3 through the transposed matrix): rows of Yt' = Sigma U^H -> u_i^H in place, beam_ut, sigma
__global__ void svd_tall3_products_kernel(cplx* __restrict__ Yt, const size_t* __restrict__ yoff,
                                          const int* __restrict__ tall, const int* __restrict__ nmodes,
                                          const double* __restrict__ sigt, const double* __restrict__ noisew,
                                          cplx* __restrict__ beam_ut, double* __restrict__ sigma, int F, int T, int K) {
  const int c = blockIdx.z;
  if (!tall[c]) return;
  const int i = blockIdx.y;
  if (i >= nmodes[c]) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int f = c % F;
  const double s = sigt[(size_t)c * T + i];
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  cplx y = Yt[yoff[c] + (size_t)i * T + t];
  y = make_double2(y.x * inv, y.y * inv);
  Yt[yoff[c] + (size_t)i * T + t] = y;
  const double w = noisew[(size_t)f * T + t];
  beam_ut[((size_t)c * K + i) * T + t] = make_double2(y.x * w, y.y * w);
  if (t == 0) sigma[(size_t)c * K + i] = s;
}

// Polarised telescopes, round 5: SVD3 runs on a matrix of its own, Z3[c] = [ U^H (w B_T) | U^H ] — the rows cut2 .. r1 of
// the accumulated row mixing (the identity part of Z after SVD2) and their total-intensity columns, recomputed from the
// input block by one product — instead of dragging the 3 (L - lmin) polarised passenger columns through every level
// product and rotation of the phase; the polarised part of `beam` is one product at the end, `ut3 . bfr` as the
// reference writes it (beamtransfer.py:877).  geo3: r3 rows x (Lc + T) columns per chain.
__global__ void svd_build_z3_kernel(const cplx* __restrict__ Z, const svd_geom* __restrict__ geo,
                                    const svd_geom* __restrict__ geo3, const int* __restrict__ row0,
                                    const int* __restrict__ nrow3, cplx* __restrict__ Z3, int T, int P) {
  const int c = blockIdx.z;
  const int i = blockIdx.y;
  if (i >= nrow3[c]) return;
  const svd_geom g = geo[c], g3 = geo3[c];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  Z3[g3.zoff + (size_t)i * g3.ldz + g.Lc + t] = Z[g.zoff + (size_t)(row0[c] + i) * g.ldz + (size_t)P * g.Lc + t];
}

// products of a polarised chain from Z3 (rows sorted by descending sigma): total-intensity part of beam_svd, beam_ut, sigma
__global__ void svd_extract3_kernel(const cplx* __restrict__ Z3, const svd_geom* __restrict__ geo3,
                                    const int* __restrict__ nmodes, const double* __restrict__ noisew,
                                    const double* __restrict__ sig3, cplx* __restrict__ beam_svd,
                                    cplx* __restrict__ beam_ut, double* __restrict__ sigma, int F, int T, int P, int L,
                                    int K) {
  const int c = blockIdx.z;
  const int f = c % F;
  const int i = blockIdx.y;
  if (i >= nmodes[c]) return;
  const svd_geom g = geo3[c];
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const cplx* z = Z3 + g.zoff + (size_t)i * g.ldz;
  if (col < g.Lc) {
    beam_svd[((size_t)c * K + i) * ((size_t)P * L) + g.lmin + col] = z[col];
  } else if (col < g.Lc + T) {
    const int t = col - g.Lc;
    const double w = noisew[(size_t)f * T + t];
    const cplx u = z[col];
    beam_ut[((size_t)c * K + i) * T + t] = make_double2(u.x * w, u.y * w);
  }
  if (col == 0) sigma[(size_t)c * K + i] = sig3[(size_t)c * T + i];
}

// Z2[c] = [ beam_svd[c][:nm][:, (p, l >= lmin)] | I_nm ]   (geo2: K rows allocated per chain, ld2 = P * Lc + K)
__global__ void svd_build_pinv_kernel(const cplx* __restrict__ beam_svd, const int* __restrict__ nmodes,
                                      cplx* __restrict__ Z2, const svd_geom* __restrict__ geo2, int K, int P, int L) {
  const int c = blockIdx.z;
  const int i = blockIdx.y;
  if (i >= nmodes[c]) return;
  const svd_geom g = geo2[c];
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= g.ldz) return;
  const int PLc = P * g.Lc;
  cplx v;
  if (col < PLc) {
    const int p = col / g.Lc, l = g.lmin + (col - p * g.Lc);
    v = beam_svd[((size_t)c * K + i) * ((size_t)P * L) + (size_t)p * L + l];
  } else {
    v = make_double2((col - PLc) == i ? 1.0 : 0.0, 0.0);
  }
  Z2[g.zoff + (size_t)i * g.ldz + col] = v;
}

// w[c][i] = 1/s^2 if s > rtol * s_max else 0   (scipy.linalg.pinv: rtol = max(M,N) eps)
__global__ void svd_pinv_weights_kernel(const double* __restrict__ s4, const int* __restrict__ nmodes,
                                        double* __restrict__ w, int K, double rtol) {
  const int c = blockIdx.x;
  const int nm = nmodes[c];
  const double smax = nm > 0 ? s4[(size_t)c * K] : 0.0;  // sorted descending
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    double s = i < nm ? s4[(size_t)c * K + i] : 0.0;
    w[(size_t)c * K + i] = (i < nm && s > rtol * smax && s > 0.0) ? 1.0 / (s * s) : 0.0;
  }
}

}  // namespace

extern "C" int dm_svd_chain(dm_ctx* ctx, int nblk, int F, int T, int P, int L, const void* beam_m_dev,
                            const double* noisew_dev, double polsvcut, void* beam_svd_dev, void* invbeam_svd_dev,
                            void* beam_ut_dev, double* sigma_dev, int* nmodes_host, int* sweeps_host) {
  return dm_svd_chain_lmin(ctx, nblk, F, T, P, L, nullptr, beam_m_dev, noisew_dev, polsvcut, beam_svd_dev,
                           invbeam_svd_dev, beam_ut_dev, sigma_dev, nmodes_host, sweeps_host);
}

extern "C" int dm_svd_chain_lmin(dm_ctx* ctx, int nblk, int F, int T, int P, int L, const int* lmin_host,
                                 const void* beam_m_dev, const double* noisew_dev, double polsvcut,
                                 void* beam_svd_dev, void* invbeam_svd_dev, void* beam_ut_dev, double* sigma_dev,
                                 int* nmodes_host, int* sweeps_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nblk >= 0 && F > 0 && T > 0 && P > 0 && L > 0 && beam_m_dev && noisew_dev && beam_svd_dev &&
                  beam_ut_dev && sigma_dev && nmodes_host);
  const int nch = nblk * F;
  if (sweeps_host) { sweeps_host[0] = sweeps_host[1] = sweeps_host[2] = sweeps_host[3] = 0; }
  if (nch == 0) return DM_OK;
  const int PL = P * L;
  const int K = std::min(L, T);
  if (lmin_host)
    for (int b = 0; b < nblk; ++b) DM_ARG(ctx, lmin_host[b] >= 0 && lmin_host[b] < L);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  // geometry: chain c = blk * F + f works on the columns l >= lmin[blk] of every polarisation
  const bool no_compact = getenv("DM_SVD_NO_COMPACT") != nullptr;   // (the switches of this call are read per call: the tests flip them)
  std::vector<svd_geom> geo(nch);
  size_t ztot = 0;
  int ldz_max = 0;
  for (int c = 0; c < nch; ++c) {
    const int lm = (lmin_host && !no_compact) ? lmin_host[c / F] : 0;
    geo[c].zoff = ztot;
    geo[c].lmin = lm;
    geo[c].Lc = L - lm;
    geo[c].ldz = P * (L - lm) + T;
    ztot += (size_t)T * geo[c].ldz;
    ldz_max = std::max(ldz_max, geo[c].ldz);
  }

  const cplx* beam = reinterpret_cast<const cplx*>(beam_m_dev);
  cplx* beam_svd = reinterpret_cast<cplx*>(beam_svd_dev);
  cplx* beam_ut = reinterpret_cast<cplx*>(beam_ut_dev);
  cplx* ibeam = reinterpret_cast<cplx*>(invbeam_svd_dev);

  // DM_SVD_NARROW=0: SVD2 / SVD3 on all columns of Z, as rounds 1-4 (the passengers of a phase ride through it)
  const bool narrow_env = !getenv("DM_SVD_NARROW") || atoi(getenv("DM_SVD_NARROW")) != 0;
  const bool narrow = narrow_env && P > 1;
  cplx* Z = dm_ws_alloc_t<cplx>(ctx, ztot);
  double* sig = dm_ws_alloc_t<double>(ctx, (size_t)nch * T);
  svd_geom* d_geo = dm_ws_upload(ctx, geo);
  if (!Z || !sig || !d_geo) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_build_z_kernel, dim3((ldz_max + 255) / 256, T, nch), dim3(256), 0, ctx->stream, beam,
                     noisew_dev, Z, d_geo, F, T, P, L);
  DM_HIP(ctx, hipGetLastError());

  std::vector<double> hs((size_t)nch * T);
  std::vector<int> r1(nch, T), cut2(nch, 0), alive(nch, 1);
  int sw = 0;

  if (P > 1) {
    // ---- phase 1: SVD1, image with rtol 1e-10 (beamtransfer.py:826, :98)
    // tall chains (P Lc <= 0.95 T) go through the transposed matrix (svd_build_yt_kernel); DM_SVD_TALL=0: all chains as they lie
    const bool tall_env = !getenv("DM_SVD_TALL") || atoi(getenv("DM_SVD_TALL")) != 0;
    static const int tall_pct = getenv("DM_SVD_TALL_PCT") ? std::min(100, atoi(getenv("DM_SVD_TALL_PCT"))) : 95;   // tall: P Lc <= tall_pct % of T (m = 320 of configs[2], P Lc = 0.89 T: 2.01 -> 1.64 s per 11 blocks; at P Lc = T even)
    std::vector<int> tall(nch, 0);
    std::vector<size_t> yoff(nch, 0);
    size_t ytot = 0;
    int ntall = 0, kc_max = 0;
    for (int c = 0; c < nch; ++c) {
      const int Kc = P * geo[c].Lc;
      tall[c] = (tall_env && Kc * 100 <= T * tall_pct) ? 1 : 0;
      if (tall[c]) { yoff[c] = ytot; ytot += (size_t)Kc * T; ++ntall; kc_max = std::max(kc_max, Kc); }
    }
    std::vector<dm_jac_problem> pr(nch);
    for (int c = 0; c < nch; ++c)
      pr[c] = dm_jac_problem{Z + geo[c].zoff, geo[c].ldz, 0, tall[c] ? 0 : T, geo[c].ldz, 0, P * geo[c].Lc};
    // SVD1 keeps s > 1e-10 s_0 (beamtransfer.py:826): rows two decades further down are left out of the sweeps
    dm_jac_rows_opts o1;
    o1.unconverged = true;
    o1.drop_below = 1e-12;
    // SVD1 hands its IMAGE to SVD2 (the rows above the cut, as a subspace): DM_SVD_SUBSPACE=0 converges it as an SVD
    const bool subspace = !getenv("DM_SVD_SUBSPACE") || atoi(getenv("DM_SVD_SUBSPACE")) != 0;   // (read per call: the tests flip it)
    if (subspace) o1.subspace_cut = 1e-10;
    if (ntall < nch) DM_TRY(dm_jacobi_rows(ctx, pr, sig, T, &sw, &o1));
    if (sweeps_host) sweeps_host[0] = sw;
    DM_TRY(dm_download(ctx, hs.data(), sig, sizeof(double) * hs.size()));
    cplx* Yt = nullptr;
    double* sigt = nullptr;
    int* d_tall = nullptr;
    size_t* d_yoff = nullptr;
    if (ntall > 0) {
      Yt = dm_ws_alloc_t<cplx>(ctx, ytot);
      sigt = dm_ws_alloc_t<double>(ctx, (size_t)nch * T);
      d_tall = dm_ws_upload(ctx, tall);
      d_yoff = dm_ws_upload(ctx, yoff);
      if (!Yt || !sigt || !d_tall || !d_yoff) return DM_ENOMEM;
      DM_TRY(dm_fill_zero(ctx, sigt, sizeof(double) * (size_t)nch * T));
      DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_build_yt_kernel, dim3((kc_max + 31) / 32, (T + 31) / 32, nch), dim3(256), 0, ctx->stream,
                 beam, noisew_dev, Yt, d_geo, d_yoff, d_tall, F, T, P, L);
      DM_HIP(ctx, hipGetLastError());
      std::vector<dm_jac_problem> pt(nch);
      for (int c = 0; c < nch; ++c)
        pt[c] = dm_jac_problem{Yt + yoff[c], T, 0, tall[c] ? P * geo[c].Lc : 0, T, 0, T};
      int swt = 0;
      // The Gram matrices of these problems have exactly zero rows and columns (sky columns beyond a frequency's band
      // limit) — what exposed the underflow in the Householder scalars of the band chase (DM_REFL_TINY, dm_kernels.h).
      // Both reductions are right now; the one-stage one is as fast at n ~ 450 (2.64 against 2.63 s on 14 blocks at
      // m = 300) and stays the default of this call, DM_SVD_TALL_TWOSTAGE=1 leaves the choice to the size policy.
      dm_jac_rows_opts ot = o1;
      ot.subspace_cut = 0.0;   // the rows of Yt become sigma_i u_i^H only when they are ORTHOGONAL: a converged SVD, not a split
      static const bool tall_two_stage = getenv("DM_SVD_TALL_TWOSTAGE") != nullptr;
      ot.one_stage_eig = !tall_two_stage;
      DM_TRY(dm_jacobi_rows(ctx, pt, sigt, T, &swt, &ot));
      sw = std::max(sw, swt);
      if (sweeps_host) sweeps_host[0] = sw;
      std::vector<double> hst((size_t)nch * T);
      DM_TRY(dm_download(ctx, hst.data(), sigt, sizeof(double) * hst.size()));
      for (int c = 0; c < nch; ++c)
        if (tall[c]) {
          const int Kc = P * geo[c].Lc;
          for (int i = 0; i < T; ++i) hs[(size_t)c * T + i] = i < Kc ? hst[(size_t)c * T + i] : 0.0;
        }
    }
    for (int c = 0; c < nch; ++c) {
      const double* s = &hs[(size_t)c * T];
      int cnt = 0;
      for (int i = 0; i < T; ++i) cnt += (s[i] > s[0] * 1e-10) ? 1 : 0;
      r1[c] = cnt;
      // the reference's guard `(s1 > 0.0).any()` (beamtransfer.py:855-857)
      alive[c] = (s[0] > 0.0) ? 1 : 0;
    }
    if (ntall > 0) {
      // rows of the tall chains: [ u_i^H (w B) | u_i^H ], i < r1
      int maxr1 = 0;
      for (int c = 0; c < nch; ++c) if (tall[c]) maxr1 = std::max(maxr1, r1[c]);
      int* d_r1 = dm_ws_upload(ctx, r1);
      if (!d_r1) return DM_ENOMEM;
      if (maxr1 > 0) {
        DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_tall_rows_kernel, dim3((T + 255) / 256, maxr1, nch), dim3(256), 0, ctx->stream, Yt, d_yoff,
                   d_tall, d_r1, sigt, Z, d_geo, T, P);
        DM_HIP(ctx, hipGetLastError());
        std::vector<dm_gemm_desc> g;
        g.reserve((size_t)ntall * P);
        for (int c = 0; c < nch; ++c) {
          if (!tall[c] || r1[c] == 0) continue;
          cplx* z = Z + geo[c].zoff;
          const cplx* u = z + (size_t)P * geo[c].Lc;
          for (int pp = 0; pp < P; ++pp)
            g.push_back(dm_gemm_make(u, geo[c].ldz, 1, false, beam + (size_t)c * T * PL + (size_t)pp * L + geo[c].lmin, PL, 1, false,
                                     z + (size_t)pp * geo[c].Lc, geo[c].ldz, r1[c], geo[c].Lc, T, 1.0, 0.0,
                                     noisew_dev + (size_t)(c % F) * T));
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
    }
    if (getenv("DM_DEBUG")) {
      // decades of the SVD1 spectrum of the first chain, and the rank range over the batch
      const double* s = &hs[0];
      int dec[20] = {0};
      for (int i = 0; i < T; ++i) {
        const double r = s[i] > 0.0 ? -std::log10(s[i] / s[0]) : 19.0;
        dec[std::min(19, std::max(0, (int)r))]++;
      }
      fprintf(stderr, "[svd_chain] SVD1 sweeps %d, r1 %d..%d, chain 0 per-decade counts:", sw,
              *std::min_element(r1.begin(), r1.end()), *std::max_element(r1.begin(), r1.end()));
      for (int d = 0; d < 20; ++d) fprintf(stderr, " %d", dec[d]);
      fprintf(stderr, "\n");
    }
    // ---- phase 2: SVD2, left null space of the polarised columns, `>=` cut (:844-848, :137)
    // (narrow: the phase works on the columns [pol | I] — the view starts behind the total-intensity block, which is
    // stale from here on and rebuilt for SVD3 from the identity part)
    for (int c = 0; c < nch; ++c)
      pr[c] = narrow ? dm_jac_problem{Z + geo[c].zoff + geo[c].Lc, geo[c].ldz, 0, r1[c], geo[c].ldz - geo[c].Lc, 0, (P - 1) * geo[c].Lc}
                     : dm_jac_problem{Z + geo[c].zoff, geo[c].ldz, 0, r1[c], geo[c].ldz, geo[c].Lc, P * geo[c].Lc};
    dm_jac_rows_opts o2;
    o2.unconverged = true;
    if (subspace && polsvcut > 0.0 && polsvcut <= 1e-3) {   // SVD2 hands its null space (the rows below the cut) to SVD3
      o2.subspace_cut = polsvcut;
      o2.subspace_margin = 100.0;
    }
    DM_TRY(dm_jacobi_rows(ctx, pr, sig, T, &sw, &o2));
    if (sweeps_host) sweeps_host[1] = sw;
    DM_TRY(dm_download(ctx, hs.data(), sig, sizeof(double) * hs.size()));
    for (int c = 0; c < nch; ++c) {
      const double* s = &hs[(size_t)c * T];
      int cnt = 0;
      for (int i = 0; i < r1[c]; ++i) cnt += (s[i] >= s[0] * polsvcut) ? 1 : 0;
      cut2[c] = cnt;
    }
    if (getenv("DM_DEBUG"))
      fprintf(stderr, "[svd_chain] SVD2 sweeps %d, cut2 %d..%d\n", sw, *std::min_element(cut2.begin(), cut2.end()),
              *std::max_element(cut2.begin(), cut2.end()));
  }

  // ---- phase 3: SVD3 on the total-intensity columns, rtol 0 (:859-865)
  std::vector<int> row0(nch), nrow3(nch);
  std::vector<svd_geom> geo3;
  cplx* Z3 = nullptr;
  svd_geom* d_geo3 = nullptr;
  std::vector<int> tall3(nch, 0);       // unpolarised chains whose SVD3 runs on the transposed matrix
  std::vector<size_t> yoff3(nch, 0);
  int ntall3 = 0, kc3_max = 0;
  cplx* Yt3 = nullptr;
  double* sigt3 = nullptr;
  int* d_tall3 = nullptr;
  size_t* d_yoff3 = nullptr;
  {
    std::vector<dm_jac_problem> pr(nch);
    for (int c = 0; c < nch; ++c) {
      row0[c] = cut2[c];
      nrow3[c] = alive[c] ? std::max(0, r1[c] - cut2[c]) : 0;
      pr[c] = dm_jac_problem{Z + geo[c].zoff, geo[c].ldz, row0[c], nrow3[c], geo[c].ldz, 0, geo[c].Lc};
    }
    if (narrow) {
      geo3.resize(nch);
      size_t z3tot = 0;
      int maxr3 = 0;
      for (int c = 0; c < nch; ++c) {
        geo3[c] = geo[c];
        geo3[c].zoff = z3tot;
        geo3[c].ldz = geo[c].Lc + T;
        z3tot += (size_t)nrow3[c] * geo3[c].ldz;
        maxr3 = std::max(maxr3, nrow3[c]);
      }
      Z3 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(z3tot, 1));
      d_geo3 = dm_ws_upload(ctx, geo3);
      int* d_r0 = dm_ws_upload(ctx, row0);
      int* d_n3 = dm_ws_upload(ctx, nrow3);
      if (!Z3 || !d_geo3 || !d_r0 || !d_n3) return DM_ENOMEM;
      if (maxr3 > 0) {
        DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_build_z3_kernel, dim3((T + 255) / 256, maxr3, nch), dim3(256), 0, ctx->stream, Z,
                   d_geo, d_geo3, d_r0, d_n3, Z3, T, P);
        DM_HIP(ctx, hipGetLastError());
        // total-intensity part: U^H diag(noisew) B_T — one product per chain out of the input block
        std::vector<dm_gemm_desc> g;
        g.reserve(nch);
        for (int c = 0; c < nch; ++c) {
          if (nrow3[c] == 0) continue;
          cplx* z3 = Z3 + geo3[c].zoff;
          g.push_back(dm_gemm_make(z3 + geo[c].Lc, geo3[c].ldz, 1, false, beam + (size_t)c * T * PL + geo[c].lmin, PL, 1, false,
                                   z3, geo3[c].ldz, nrow3[c], geo[c].Lc, T, 1.0, 0.0, noisew_dev + (size_t)(c % F) * T));
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
      for (int c = 0; c < nch; ++c)
        pr[c] = dm_jac_problem{Z3 + geo3[c].zoff, geo3[c].ldz, 0, nrow3[c], geo3[c].ldz, 0, geo[c].Lc};
    }
    // Unpolarised telescopes: SVD3 is the whole chain, and a block with Lc <= 0.95 T sky columns goes through the
    // transposed matrix Yt = (w B)^H (Lc x T) — Lc rows to orthogonalise instead of T (configs[1]: T = 92, Lc = 129 - m)
    if (P == 1) {
      static const bool tall_env3 = !getenv("DM_SVD_TALL") || atoi(getenv("DM_SVD_TALL")) != 0;
      static const int tall_pct3 = getenv("DM_SVD_TALL_PCT") ? std::min(100, atoi(getenv("DM_SVD_TALL_PCT"))) : 95;
      size_t ytot = 0;
      for (int c = 0; c < nch; ++c) {
        tall3[c] = (tall_env3 && geo[c].Lc * 100 <= T * tall_pct3) ? 1 : 0;
        if (tall3[c]) {
          yoff3[c] = ytot; ytot += (size_t)geo[c].Lc * T; ++ntall3; kc3_max = std::max(kc3_max, geo[c].Lc);
          pr[c].nrows = 0;
        }
      }
      if (ntall3 > 0) {
        Yt3 = dm_ws_alloc_t<cplx>(ctx, ytot);
        sigt3 = dm_ws_alloc_t<double>(ctx, (size_t)nch * T);
        d_tall3 = dm_ws_upload(ctx, tall3);
        d_yoff3 = dm_ws_upload(ctx, yoff3);
        if (!Yt3 || !sigt3 || !d_tall3 || !d_yoff3) return DM_ENOMEM;
        DM_TRY(dm_fill_zero(ctx, sigt3, sizeof(double) * (size_t)nch * T));
        DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_build_yt_kernel, dim3((kc3_max + 31) / 32, (T + 31) / 32, nch), dim3(256), 0,
                   ctx->stream, beam, noisew_dev, Yt3, d_geo, d_yoff3, d_tall3, F, T, P, L);
        DM_HIP(ctx, hipGetLastError());
      }
    }
    // polarised: certainly not orthogonal yet.  Unpolarised: the measuring pass is kept, it retires the
    // all-zero and trivially orthogonal blocks of the high m (a fifth of config 2) before the eigensolver.
    dm_jac_rows_opts o3;
    o3.unconverged = P > 1;
    if (ntall3 < nch) DM_TRY(dm_jacobi_rows(ctx, pr, sig, T, &sw, &o3));
    if (sweeps_host) sweeps_host[2] = sw;
    DM_TRY(dm_download(ctx, hs.data(), sig, sizeof(double) * hs.size()));
    if (ntall3 > 0) {
      std::vector<dm_jac_problem> pt(nch);
      for (int c = 0; c < nch; ++c) pt[c] = dm_jac_problem{Yt3 + yoff3[c], T, 0, tall3[c] ? geo[c].Lc : 0, T, 0, T};
      int swt = 0;
      DM_TRY(dm_jacobi_rows(ctx, pt, sigt3, T, &swt, &o3));
      sw = std::max(sw, swt);
      if (sweeps_host) sweeps_host[2] = sw;
      std::vector<double> hst((size_t)nch * T);
      DM_TRY(dm_download(ctx, hst.data(), sigt3, sizeof(double) * hst.size()));
      for (int c = 0; c < nch; ++c)
        if (tall3[c])
          for (int i = 0; i < T; ++i) hs[(size_t)c * T + i] = i < geo[c].Lc ? hst[(size_t)c * T + i] : 0.0;
    }
  }
  std::vector<int> nmodes(nch, 0);
  int maxnm = 0;
  for (int c = 0; c < nch; ++c) {
    const double* s = &hs[(size_t)c * T];
    int cnt = 0;
    const int lim = std::min(tall3[c] ? geo[c].Lc : nrow3[c], K);
    for (int i = 0; i < lim; ++i) cnt += (s[i] > 0.0) ? 1 : 0;  // rtol = 0.0: strictly positive
    nmodes[c] = cnt;
    nmodes_host[c] = cnt;
    maxnm = std::max(maxnm, cnt);
  }
  if (getenv("DM_DEBUG"))
    fprintf(stderr, "[svd_chain] SVD3 sweeps %d, nmodes %d..%d\n", sw, *std::min_element(nmodes.begin(), nmodes.end()),
            maxnm);

  // ---- products
  int* d_row0 = dm_ws_upload(ctx, row0);
  int* d_nm = dm_ws_upload(ctx, nmodes);
  if (!d_row0 || !d_nm) return DM_ENOMEM;
  DM_TRY(dm_fill_zero(ctx, beam_svd, sizeof(cplx) * (size_t)nch * K * PL));
  DM_TRY(dm_fill_zero(ctx, beam_ut, sizeof(cplx) * (size_t)nch * K * T));
  DM_TRY(dm_fill_zero(ctx, sigma_dev, sizeof(double) * (size_t)nch * K));
  if (maxnm > 0 && !narrow) {
    const int* d_nm_z = d_nm;
    if (ntall3 > 0) {   // the rows of those chains are not in Z
      std::vector<int> nmz(nmodes);
      for (int c = 0; c < nch; ++c) if (tall3[c]) nmz[c] = 0;
      d_nm_z = dm_ws_upload(ctx, nmz);
      if (!d_nm_z) return DM_ENOMEM;
    }
    DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_extract_kernel, dim3((ldz_max + 255) / 256, maxnm, nch), dim3(256), 0, ctx->stream, Z, d_geo,
                       d_row0, d_nm_z, noisew_dev, sig, beam_svd, beam_ut, sigma_dev, F, T, P, L, K);
    DM_HIP(ctx, hipGetLastError());
    if (ntall3 > 0) {
      // u_i^H = Yt'[i] / sigma_i (in place), beam_ut = u_i^H diag(noisew), sigma; beam = u_i^H (w B): one product per chain
      DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_tall3_products_kernel, dim3((T + 255) / 256, maxnm, nch), dim3(256), 0, ctx->stream, Yt3,
                 d_yoff3, d_tall3, d_nm, sigt3, noisew_dev, beam_ut, sigma_dev, F, T, K);
      DM_HIP(ctx, hipGetLastError());
      std::vector<dm_gemm_desc> g;
      g.reserve(ntall3);
      for (int c = 0; c < nch; ++c) {
        if (!tall3[c] || nmodes[c] == 0) continue;
        g.push_back(dm_gemm_make(Yt3 + yoff3[c], T, 1, false, beam + (size_t)c * T * PL + geo[c].lmin, PL, 1, false,
                                 beam_svd + (size_t)c * K * PL + geo[c].lmin, PL, nmodes[c], geo[c].Lc, T, 1.0, 0.0,
                                 noisew_dev + (size_t)(c % F) * T));
      }
      DM_TRY(dm_gemm_grouped_launch(ctx, g));
    }
  }
  if (maxnm > 0 && narrow) {
    DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_extract3_kernel, dim3((L + T + 255) / 256, maxnm, nch), dim3(256), 0, ctx->stream, Z3,
                       d_geo3, d_nm, noisew_dev, sig, beam_svd, beam_ut, sigma_dev, F, T, P, L, K);
    DM_HIP(ctx, hipGetLastError());
    // the polarised part of `beam = ut3 . bfr` (beamtransfer.py:877): rows of U^H (the identity part of Z3) times the
    // noise-weighted input block, one product per polarisation into the columns l >= lmin of the (zero-filled) output
    std::vector<dm_gemm_desc> g;
    g.reserve((size_t)nch * (P - 1));
    for (int c = 0; c < nch; ++c) {
      const int nm = nmodes[c];
      if (nm == 0) continue;
      const cplx* u = Z3 + geo3[c].zoff + geo[c].Lc;
      for (int pp = 1; pp < P; ++pp)
        g.push_back(dm_gemm_make(u, geo3[c].ldz, 1, false, beam + (size_t)c * T * PL + (size_t)pp * L + geo[c].lmin, PL, 1, false,
                                 beam_svd + (size_t)c * K * PL + (size_t)pp * L + geo[c].lmin, PL, nm, geo[c].Lc, T, 1.0, 0.0,
                                 noisew_dev + (size_t)(c % F) * T));
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g));
  }

  // ---- pseudo-inverse of `beam` (:887-921)
  if (ibeam) {
    DM_TRY(dm_fill_zero(ctx, ibeam, sizeof(cplx) * (size_t)nch * PL * K));
    if (maxnm > 0) {
      // [beam | I] per chain: K rows of P * Lc + K columns
      std::vector<svd_geom> geo2(nch);
      size_t z2tot = 0;
      int ld2_max = 0;
      for (int c = 0; c < nch; ++c) {
        geo2[c] = geo[c];
        geo2[c].zoff = z2tot;
        geo2[c].ldz = P * geo[c].Lc + K;
        z2tot += (size_t)K * geo2[c].ldz;
        ld2_max = std::max(ld2_max, geo2[c].ldz);
      }
      // Z is no longer needed: reuse its storage when it is large enough
      cplx* Z2 = (z2tot <= ztot) ? Z : dm_ws_alloc_t<cplx>(ctx, z2tot);
      double* s4 = dm_ws_alloc_t<double>(ctx, (size_t)nch * K);
      double* w4 = dm_ws_alloc_t<double>(ctx, (size_t)nch * K);
      svd_geom* d_geo2 = dm_ws_upload(ctx, geo2);
      if (!Z2 || !s4 || !w4 || !d_geo2) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_build_pinv_kernel, dim3((ld2_max + 255) / 256, maxnm, nch), dim3(256), 0, ctx->stream,
                         beam_svd, d_nm, Z2, d_geo2, K, P, L);
      std::vector<dm_jac_problem> pr(nch);
      for (int c = 0; c < nch; ++c)
        pr[c] = dm_jac_problem{Z2 + geo2[c].zoff, geo2[c].ldz, 0, nmodes[c], P * geo[c].Lc + nmodes[c], 0, P * geo[c].Lc};
      // unpolarised: these are exactly the rows SVD3 left orthogonal over the same columns (the measuring pass
      // sees that and skips everything); polarised: orthogonal over the T columns only
      dm_jac_rows_opts o4;
      o4.unconverged = P > 1;
      DM_TRY(dm_jacobi_rows(ctx, pr, s4, K, &sw, &o4));
      if (sweeps_host) sweeps_host[3] = sw;
      // scipy.linalg.pinv: rtol = max(M, N) eps of the matrix it is GIVEN — the padded (nm x P L) beam (beamtransfer.py:891)
      const double rtol = (double)std::max(PL, maxnm) * 2.220446049250313e-16;
      DM_PLAUNCH(ctx, DM_PROF_SVD_OTHER, svd_pinv_weights_kernel, dim3(nch), dim3(256), 0, ctx->stream, s4, d_nm, w4, K, rtol);
      std::vector<dm_gemm_desc> g;
      g.reserve(nch);
      for (int c = 0; c < nch; ++c) {
        const int nm = nmodes[c];
        if (nm == 0) continue;
        const int ld2 = geo2[c].ldz, Lc = geo[c].Lc;
        const cplx* Y = Z2 + geo2[c].zoff;                   // (nm x P Lc): rows = s_k v_k^H
        const cplx* W = Y + P * Lc;                          // (nm x nm): rows of U_b^H
        // ibeam (PL x nm) = Y^H diag(w) W ; destination is (P, L, K) with K the fastest axis: one product per
        // polarisation, into the rows l >= lmin of the (zero-filled) output
        for (int pp = 0; pp < P; ++pp)
          g.push_back(dm_gemm_make(Y + (size_t)pp * Lc, 1, ld2, true, W, ld2, 1, false,
                                   ibeam + (size_t)c * PL * K + ((size_t)pp * L + geo[c].lmin) * K, K, Lc, nm, nm, 1.0,
                                   0.0, w4 + (size_t)c * K));
      }
      DM_TRY(dm_gemm_grouped_launch(ctx, g));
    }
  }
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}
