/* driftmi.h — C ABI of libdriftmi, the MI355X (gfx950) implementation of
 * driftscan's per-m hot path (beam-transfer generation, SVD compression,
 * KL / DoubleKL).  This is the drop-in boundary: plain pointers and sizes, no
 * torch / numpy types, `extern "C"`.  The reference has no FFI of its own (it is
 * Python + LAPACK); each entry point therefore cites the reference *call site*
 * whose arithmetic it replaces (paths relative to radiocosmology/driftscan).
 *
 * Conventions
 *   - complex128 = interleaved (double re, double im); matrices are row-major.
 *   - every pointer named *_dev is DEVICE memory owned by the caller (e.g. a
 *     torch tensor's data_ptr()); pointers named *_host are host memory.
 *   - every function returns int: 0 ok, <0 argument/runtime error (see
 *     dm_last_error), >0 numerical status mirroring LAPACK `info`.  0 is the ONLY
 *     success value: no entry point reports a count or a warning through its
 *     return value (the Python binding raises on every non-zero status, positive
 *     ones included — products behind a failed eigen-iteration are not usable).
 *   - a dm_ctx owns a device id, a HIP stream and a growable device workspace;
 *     one context per GPU, one host thread per context.  All work is enqueued on
 *     the context's stream; functions documented as "synchronises" block the
 *     host until their results are complete.
 *   - there is NO CPU fallback: without a GPU every compute entry point fails.
 */
#ifndef DRIFTMI_H
#define DRIFTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_ctx dm_ctx;

/* ---- context ----------------------------------------------------------- */
/* stream: an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)
 * or NULL to let the context create its own. */
int dm_ctx_create(int device, size_t workspace_bytes, void* stream, dm_ctx** out);
int dm_ctx_destroy(dm_ctx* ctx);
int dm_ctx_sync(dm_ctx* ctx);
size_t dm_ctx_workspace_bytes(dm_ctx* ctx);
/* Replace the (idle) workspace arena by one of exactly `bytes` (0: just give the memory back).  The arena
 * only ever grows on its own; a stage that needs most of the card in one piece (dm_eigh_gen at n = 32 576:
 * ~140 GB) calls this first so that the previous stage's arena is not in the way.  Synchronises.
 * No reference counterpart (numpy allocates per call). */
int dm_ctx_workspace_reset(dm_ctx* ctx, size_t bytes);
const char* dm_last_error(dm_ctx* ctx);
int dm_version(void);

/* dm_prof_trd_stride: the live measurement samples — every dm_prof_trd_stride()-th column of the tridiagonalisation
 * is bracketed by events, every launch of >= 2e9 work units of the other classes, and one in four of the smaller
 * ones; dm_prof_report weights the samples (by the stride / by four), so its figures estimate the totals over all
 * launches and `launches` the number of launches.  Event records cost a marker packet each. */
int dm_prof_trd_stride(void);

/* Per-kernel-class instrumentation for bench.py: when enabled (1) every launch of the MFMA / HBM-bound kernel
 * classes is bracketed by HIP events on the context's stream (sampled as dm_prof_trd_stride describes); with
 * enable = 2 every remaining kernel of the path is bracketed as well (the "extended" classes, time only).
 * dm_prof_report fills 24-entry arrays with summed event time [ms], algorithmic flops actually executed and launch
 * counts since the last reset:
 *   0 grouped ZGEMM, 1 real-B GEMM, 2 Jacobi Gram, 3 Jacobi inner solver, 4 Jacobi apply, 5 real GEMM of the D&C merges,
 *   6 / 7 the one-stage tridiagonalisation kernels [algorithmic BYTES], 8 panel QR, 9 bulge chase, 10 second-stage
 *   back-transformation of the two-stage tridiagonalisation, 11 gathered-B ZGEMM (the covariance projections);
 *   extended: 12 fused ring transform of BT-gen, 13 other BT-gen kernels, 14 LDS-resident small tridiagonalisation,
 *   15 divide & conquer (without its real GEMMs), 16 Cholesky panels + triangular diagonal solves, 17 transposes /
 *   copies / scans, 18 T factors / slice sums / gathers of the eigensolver, 19 Jacobi-engine and SVD-chain helpers,
 *   20-23 unused. */
int dm_prof_reset(dm_ctx* ctx, int enable);
int dm_prof_report(dm_ctx* ctx, double* ms, double* flops, long long* launches);

/* ---- dense building blocks (exposed for the parity tests) --------------- */
/* C = alpha * op(A) diag(kscale) op(B) + beta C on the fp64 matrix cores.
 * A is viewed as (M x K) through element strides (rsA, csA), B as (K x N)
 * through (rsB, csB); conj* conjugate the elements.  `batch` problems at the
 * given element strides between consecutive A/B/C/kscale.
 * Replaces: np.dot at drift/core/beamtransfer.py:1186, :1226; doublekl.py:73-85. */
int dm_zgemm_strided_batched(dm_ctx* ctx, int M, int N, int K, double alpha, const void* A_dev, int rsA, int csA,
                             int conjA, int64_t strideA, const void* B_dev, int rsB, int csB, int conjB,
                             int64_t strideB, double beta, void* C_dev, int ldc, int64_t strideC,
                             const double* kscale_dev, int64_t stride_kscale, int batch);

/* The same product for a ragged group: `nprob` independent problems of different shapes in ONE launch
 * (every 64x64 output tile of every problem is a workgroup).  Used where the reference loops over m or
 * frequency with one np.dot per iteration.
 * Replaces: np.dot at drift/core/doublekl.py:73-74, :80, :85; drift/core/kltransform.py:124-143 (inv_gen,
 * formed here as N E^H from E N E^H = I). */
typedef struct dm_zgemm_problem {
  const void* A;  /* device c128, viewed (M x K) through element strides (rsA, csA) */
  const void* B;  /* device c128, viewed (K x N) through (rsB, csB) */
  void* C;        /* device c128, row-major, leading dimension ldc */
  int M, N, K;
  int rsA, csA, rsB, csB, ldc;
  int conjA, conjB;
  double alpha, beta;
} dm_zgemm_problem;
int dm_zgemm_grouped(dm_ctx* ctx, int nprob, const dm_zgemm_problem* probs_host);

/* Batched lower Cholesky A = L L^H in place (n x n, row-major, ld), `batch`
 * matrices `stride` elements apart.  info_host[i] = 0 or the order of the first
 * non-positive-definite leading minor.  Synchronises.
 * Replaces: the zpotrf inside scipy.linalg.eigh(A, B) at drift/core/kltransform.py:89. */
int dm_zpotrf_batched(dm_ctx* ctx, int n, void* A_dev, int ld, int64_t stride, int batch, int* info_host);

/* Batched triangular solve with the lower factor: L X = B (conjtrans = 0) or
 * L^H X = B (conjtrans = 1), X overwrites B (n x nrhs row-major).
 * Replaces: zhegst / ztrsm inside scipy.linalg.eigh(A, B), kltransform.py:89. */
int dm_ztrsm_left_lower_batched(dm_ctx* ctx, int n, int nrhs, const void* L_dev, int ldl, int64_t strideL,
                                void* B_dev, int ldb, int64_t strideB, int conjtrans, int batch);

/* One-sided block-Jacobi SVD of `batch` row-major matrices (rows x cols, ld,
 * `stride` elements apart): the rows of each matrix are unitarily mixed in place
 * until mutually orthogonal with respect to the columns [gc0, gc1) and sorted by
 * descending norm over those columns.  On return rows = diag(sigma) V^H padded
 * with the passenger columns; sigma_dev gets `rows` doubles per matrix.
 * Synchronises.  sweeps_host (optional) receives the sweep count.
 * Replaces: scipy.linalg.svd at drift/core/beamtransfer.py:74, :113. */
int dm_jacobi_rows_batched(dm_ctx* ctx, int rows, int cols, int gc0, int gc1, void* Z_dev, int ld, int64_t stride,
                           int batch, double* sigma_dev, int* sweeps_host);

/* Two-sided block-Jacobi eigendecomposition of `batch` Hermitian n x n matrices:
 * on return C is diagonal (to working accuracy), W_dev (n x n per matrix, rows
 * are eigenvectors^H) holds the accumulated unitary, evals_dev the unsorted
 * diagonal.  Synchronises.
 * Replaces: the zheevd inside scipy.linalg.eigh, kltransform.py:89, :107. */
int dm_jacobi_herm_batched(dm_ctx* ctx, int n, void* C_dev, int ldc, int64_t strideC, void* W_dev, int ldw,
                           int64_t strideW, int batch, double* evals_dev, int* sweeps_host);

/* Hermitian eigendecomposition of `batch` n x n matrices by Householder
 * tridiagonalisation + implicit QL + blocked back-transformation (the production
 * path of dm_eigh_gen).  C is destroyed; W_dev rows are eigenvectors^H; evals_dev
 * (n per matrix) is NOT sorted.  Returns > 0 if the QL iteration fails.  Synchronises.
 * Replaces: the zheevd inside scipy.linalg.eigh, drift/core/kltransform.py:89, :107. */
int dm_herm_eig_batched(dm_ctx* ctx, int n, void* C_dev, int ldc, int64_t strideC, void* W_dev, int ldw,
                        int64_t strideW, int batch, double* evals_dev);

/* ---- SVD compression of beam-transfer blocks -------------------------------- */
/* Three-stage SVD chain + pseudo-inverse for nblk m-blocks x F frequencies.
 *   beam_m_dev   (nblk, F, T, P, L) c128   noise-unweighted beam_m blocks, T = 2*nbase
 *                                         (the reference's (2, nbase) axes flattened), zero for l < m
 *   noisew_dev   (F, T) f64               noisepower(b, f)^-1/2, duplicated for the two m signs
 *   beam_svd_dev (nblk, F, K, P, L) c128  out, K = min(L, T); rows >= nmodes are zero
 *   invbeam_svd_dev (nblk, F, P, L, K)    out or NULL (skip_svd_inv)
 *   beam_ut_dev  (nblk, F, K, T) c128     out
 *   sigma_dev    (nblk, F, K) f64         out
 *   nmodes_host  (nblk*F) int             out
 *   sweeps_host  optional int[4]: Jacobi sweeps used by SVD1, SVD2, SVD3, pinv
 * SVD1 hands its image and SVD2 its null space to the next SVD as orthonormal bases (:826, :844-848): a unitary change of
 * basis inside either subspace is undone by the next SVD, so on polarised blocks these two phases stop when the cut
 * (1e-10, polsvcut) is placed to the accuracy of a converged SVD and do not order the rows inside each side
 * (sweeps_host[0..1] = 0 then; DM_SVD_SUBSPACE=0 converges them).  sigma, nmodes and every product are those of the
 * reference chain (same spectra to 1e-13 sigma_max).
 * Synchronises.
 * Replaces: BeamTransfer._generate_svdfile_m, drift/core/beamtransfer.py:802-924
 * (matrix_image :68-104, matrix_nullspace :107-143, scipy.linalg.pinv :891). */
int dm_svd_chain(dm_ctx* ctx, int nblk, int F, int T, int P, int L, const void* beam_m_dev,
                 const double* noisew_dev, double polsvcut, void* beam_svd_dev, void* invbeam_svd_dev,
                 void* beam_ut_dev, double* sigma_dev, int* nmodes_host, int* sweeps_host);
/* The same chain for blocks whose columns l < lmin_host[blk] are exactly zero — block blk = the m-block of
 * m = lmin_host[blk], as `BeamTransfer.beam_m` returns it (beamtransfer.py:257-308: l padded from 0).  Zero columns add
 * nothing to an inner product and stay zero under row mixing, so every chain of block blk works on the
 * P * (L - lmin) columns that can be non-zero; products come back in the padded layout, zero for l < lmin.
 *   lmin_host    (nblk) int, 0 <= lmin < L; NULL = all zero (dm_svd_chain)
 * The caller guarantees the zeros (a non-zero entry at l < lmin would be dropped).  Same products as dm_svd_chain up to
 * the order of floating-point sums.  Replaces the same reference lines. */
int dm_svd_chain_lmin(dm_ctx* ctx, int nblk, int F, int T, int P, int L, const int* lmin_host,
                      const void* beam_m_dev, const double* noisew_dev, double polsvcut, void* beam_svd_dev,
                      void* invbeam_svd_dev, void* beam_ut_dev, double* sigma_dev, int* nmodes_host, int* sweeps_host);

/* ---- KL: covariance projection and generalised eigenproblem ---------------- */
/* Project a sky covariance into the SVD basis for nblk m-blocks.
 *   svnum_host   (nblk*F) int   kept modes per frequency (BeamTransfer._svd_num)
 *   l0_host      (nblk) int or NULL: first non-zero l of each block (= m), columns below are skipped
 *   cl_pfl_dev   (P, P, F, F, L) f64: the reference's (P,P,L,F,F) C_l array with l moved last
 *   npol         pol pairs looped: P, or 1 for `temponly`
 *   polmask_host (P*P) int or NULL: non-zero where cl[pi,pj] is not identically zero
 *   out_dev      c128, block b is (ndof_b x ndof_b) row-major at element offset out_off_host[b]
 *   zero_first   flag bits: 1 = clear the outputs before accumulating; 2 = the caller has checked that
 *                cl[pi,pj,f,f',l] == cl[pi,pj,f',f,l] for every pair it passes (the usual sky models): together
 *                with bit 1 and a pol mask without off-diagonal pairs the result is Hermitian block by block and
 *                only the frequency blocks f' >= f are formed, the others mirrored.  Without bit 2 every block
 *                is formed, as the reference does for an arbitrary array.
 * Replaces: BeamTransfer.project_matrix_sky_to_svd, drift/core/beamtransfer.py:1135-1188. */
int dm_project_cov(dm_ctx* ctx, int nblk, int F, int K, int P, int L, const void* beam_svd_dev,
                   const int* svnum_host, const int* l0_host, const double* cl_pfl_dev, int npol,
                   const int* polmask_host, void* out_dev, const int64_t* out_off_host, int zero_first);

/* Block-diagonal projection of a diagonal telescope-basis matrix dmat (F, T):
 * block f (+)= alpha (U_f * d_f) U_f^H.
 * Replaces: project_matrix_diagonal_telescope_to_svd, drift/core/beamtransfer.py:1190-1231. */
int dm_project_diag(dm_ctx* ctx, int nblk, int F, int K, int T, const void* beam_ut_dev, const int* svnum_host,
                    const double* dmat_dev, double alpha, void* out_dev, const int64_t* out_off_host,
                    int accumulate);

/* diag(N_b) += reg * max(N_b), numpy's lexicographic complex max.
 * Replaces: drift/core/kltransform.py:289-290. */
int dm_regularise(dm_ctx* ctx, int nblk, const int* n_host, void* mats_dev, const int64_t* off_host, double reg);

/* Generalised Hermitian-definite eigenproblems A v = lambda B v (A, B destroyed).
 * evals (ALL of them) ascending at evals_dev + evoff_host[b]; evecs_dev + off_host[b] holds an
 * (n x n) matrix whose ROWS are the modes (the reference's evecs.T.conj()), row i belonging to
 * eigenvalue i.  add_const_host[b] = diagonal shift applied by the non-positive-definite rescue.
 * cut_mode selects which modes are formed at all (the back-transformation and the final
 * triangular solve are 2/3 of the vector work):
 *   0  all rows;
 *   1  only rows i >= i_ev, i_ev = first index with evals[i] >= cut_value (numpy searchsorted) —
 *      what KLTransform keeps with `subset` (kltransform.py:388-398);
 *   2  only rows i < i_ev — the foreground-clean modes DoubleKL keeps in its first stage (doublekl.py:54-60).
 * The other rows of the block are zero.  nkeep_host[b] (may be NULL) receives the number of rows formed.
 * Returns > 0 if a B stays indefinite after the rescue.  Synchronises.
 * Replaces: eigh_gen, drift/core/kltransform.py:55-121 (+ the threshold cuts cited above). */
int dm_eigh_gen(dm_ctx* ctx, int nblk, const int* n_host, void* A_dev, void* B_dev, const int64_t* off_host,
                double* evals_dev, const int64_t* evoff_host, void* evecs_dev, double* add_const_host,
                int* sweeps_host, int cut_mode, double cut_value, int* nkeep_host);

/* KLTransform._transform_m for nblk m-blocks in one call: S = B C_sg B^H, N = B C_fg B^H (cl_fg_dev NULL: zero) +
 * regulariser * max(N) on the diagonal + noise_scale * U diag(npower) U^H, then dm_eigh_gen with the given cut.
 *   beam_svd_dev (nblk, F, K, P, L), beam_ut_dev (nblk, F, K, T), svnum_host (nblk * F), l0_host (nblk) or NULL
 *   cl_*_dev (P, P, F, F, L) f64 with masks / symmetry flags as for dm_project_cov; npower_dev (F, T) f64
 *   noise_scale  1 with `use_thermal`, (1e-3 / Tsys)^2 without (kltransform.py:292-296)
 *   outputs as dm_eigh_gen (block b of the modes at evecs_dev + off_host[b], (ndof_b x ndof_b), rows = modes)
 * Synchronises.  Replaces: drift/core/kltransform.py:258-355 (sn_covariance + _transform_m). */
int dm_kl_m(dm_ctx* ctx, int nblk, int F, int K, int P, int L, int T, const void* beam_svd_dev, const void* beam_ut_dev,
            const int* svnum_host, const int* l0_host, const double* cl_sg_dev, const int* sg_mask_host, int sg_symmetric,
            const double* cl_fg_dev, const int* fg_mask_host, int fg_symmetric, const double* npower_dev, double noise_scale,
            double regulariser, int cut_mode, double cut_value, double* evals_dev, const int64_t* evoff_host, void* evecs_dev,
            const int64_t* off_host, double* add_const_host, int* nkeep_host);

/* DoubleKL._transform_m for nblk m-blocks in one call: stage 1 diagonalises S against N with the thermal term at
 * floor_scale = (1e-3 / Tsys)^2 and keeps the modes with eigenvalue > foreground_threshold; stage 2 diagonalises the full
 * S, N (thermal term at 1) inside that subspace; the composed modes E2 . E1[kept] are returned.
 *   f_evals_dev  ndof_b stage-1 eigenvalues at evoff_host[b]
 *   evals_dev    the nmodes_host[b] stage-2 eigenvalues at evoff_host[b]
 *   modes_dev    (nmodes_b x ndof_b) at off_host[b]; with cut_mode 1 the rows below the cut are zero and
 *                nkeep_host[b] (may be NULL) counts the rows formed
 *   add_const_host  the stage-1 rescue shift, the one the reference stores (doublekl.py:58)
 * Synchronises.  Replaces: drift/core/doublekl.py:30-87 (without `inverse`). */
int dm_doublekl_m(dm_ctx* ctx, int nblk, int F, int K, int P, int L, int T, const void* beam_svd_dev, const void* beam_ut_dev,
                  const int* svnum_host, const int* l0_host, const double* cl_sg_dev, const int* sg_mask_host, int sg_symmetric,
                  const double* cl_fg_dev, const int* fg_mask_host, int fg_symmetric, const double* npower_dev,
                  double floor_scale, double regulariser, double foreground_threshold, int cut_mode, double cut_value,
                  double* f_evals_dev, double* evals_dev, const int64_t* evoff_host, void* modes_dev, const int64_t* off_host,
                  int* nmodes_host, int* nkeep_host, double* add_const_host);

/* Exact per-m Fisher matrices of the band powers for nblk m-blocks:
 *   C_a = E (B C_l^a B^H) E^H,  F[a][b] = sum_ij C_a[i][j] C_b[j][i] / ((lam_i + 1)(lam_j + 1)).
 *   beam_svd_dev, svnum_host, l0_host   as for dm_project_cov
 *   cl_bands_dev (nbands, F, F, L) f64  band angular power spectra (temperature block), an INPUT
 *   evecs_dev + evecs_off_host[b]       (nmodes_host[b] x ndof_b) c128, rows = KL modes above the threshold
 *   evals_dev + evals_off_host[b]       their eigenvalues (f64)
 *   fisher_dev (nblk, nbands, nbands) c128  out (zero for blocks without modes)
 *   cl_symmetric  non-zero when cl_bands[a,f,f',l] == cl_bands[a,f',f,l] was checked by the caller (see dm_project_cov)
 * Synchronises.
 * Replaces: PSExact.makeproj + _work_fisher_bias_m, drift/core/psestimation.py:672-699, :775-815
 * (the bias of PSExact is identically zero, :797). */
int dm_fisher(dm_ctx* ctx, int nblk, int F, int K, int P, int L, const void* beam_svd_dev, const int* svnum_host,
              const int* l0_host, int nbands, const double* cl_bands_dev, const void* evecs_dev,
              const int64_t* evecs_off_host, const int* nmodes_host, const double* evals_dev,
              const int64_t* evals_off_host, void* fisher_dev, int cl_symmetric);

/* ---- beam-transfer generation (cylinder telescopes) --------------------------- */
/* Host geometry shared by the three calls below: ring_cth_host / ring_sth_host hold
 * cos / sin of the colatitude of the 4*nside-1 HEALPix rings; frame_host (9 doubles)
 * holds xhat (East), yhat (North), zhat (zenith) in sky cartesian coordinates.
 *
 * dm_bt_beam_cyl: field pattern of one (frequency, beam class) on all pixel centres.
 *   kind 0: unpolarised amplitude, out (npix) f64; 1 / 2: X / Y dipole, out (npix, 2) f64
 *   tab_*_host (ntab): knots (x, y, y'') of the natural cubic spline of the E-W
 *   Fraunhofer pattern; fwhm_ns: FWHM of the exptan N-S pattern.
 * Replaces: cylbeam.beam_amp / beam_x / beam_y, drift/telescope/cylbeam.py:101-212,
 *           _fast_tools.beam_exptan, drift/util/_fast_tools.pyx:248-282. */
int dm_bt_beam_cyl(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                   const double* frame_host, int kind, const double* tab_x_host, const double* tab_y_host,
                   const double* tab_y2_host, int ntab, double fwhm_ns, double* out_dev);

/* dm_bt_beams_cyl: nbeam patterns of dm_bt_beam_cyl in one call (one upload of the ring geometry and one of all spline
 * tables instead of eight staged copies per beam).  tab_off_host (nbeam + 1) delimits the knots of each beam in the
 * concatenated tab_x / tab_y / tab_y2 arrays; row b of the output starts at out_dev + b * out_stride (doubles). */
int dm_bt_beams_cyl(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                    const double* frame_host, int nbeam, const int* kind_host, const double* tab_x_host,
                    const double* tab_y_host, const double* tab_y2_host, const int* tab_off_host,
                    const double* fwhm_ns_host, double* out_dev, size_t out_stride);

/* dm_bt_maps: complex visibility response maps for ncol (frequency, baseline) columns:
 * h * fringe * (b_i x b_j) / sqrt(Omega_i Omega_j) -> (ncol, P, npix) c128 with P = 4
 * (I, Q, U, V) if polarised else 1.  beams_dev: nbeam maps from dm_bt_beam_cyl;
 * uv_host (ncol, 2) = baseline / wavelength; bi_host / bj_host (ncol) beam indices.
 * Replaces: _fast_tools.fringe, _construct_pol_real (drift/util/_fast_tools.pyx:18-164),
 *           Unpolarised/PolarisedTelescope._beam_map_single (drift/core/telescope.py:1156-1176, :1268-1283). */
int dm_bt_maps(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
               const double* frame_host, int polarised, int nbeam, const double* beams_dev, int ncol,
               const double* uv_host, const int* bi_host, const int* bj_host, void* maps_dev);

/* dm_bt_maps_c: dm_bt_maps for COMPLEX field patterns — beams_dev holds nbeam maps of npix * ncomp complex128 (zero
 * below the horizon); the second beam of a pair enters conjugated and the solid angles are sums of |b|^2.
 * Replaces drift/util/_fast_tools.pyx:167-242 (_construct_pol_complex, reached from telescope.py:1278-1279) and the
 * complex case of UnpolarisedTelescope._beam_map_single (telescope.py:1156-1176). */
int dm_bt_maps_c(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                 const double* frame_host, int polarised, int nbeam, const void* beams_dev, int ncol,
                 const double* uv_host, const int* bi_host, const int* bj_host, void* maps_dev);

/* dm_bt_sht: spherical-harmonic transform of the maps straight into the m-ordered
 * blocks beam_m_dev (mmax+1, F, 2, B, P, lside+1) c128 (the reference's beam_m layout
 * with l padded from 0): rows (f, :, b) of the given columns are overwritten, zero for
 * l < m and l > col_lmax.  lmax_grp = max(col_lmax) <= lside.  Synchronises.
 * Replaces: _transfer_single / transfer_matrices (drift/core/telescope.py:755-830,
 *           :1178-1193, :1287-1316, i.e. cora.util.hputil.sphtrans_complex[_pol]) and the
 *           +/-m fold of BeamTransfer._generate_mfiles (drift/core/beamtransfer.py:620-624). */
int dm_bt_sht(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
              int lside, int mmax, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
              const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev);

/* dm_bt_sht_range: the same for the m-blocks m_lo .. m_hi only; beam_m_dev is then
 * (m_hi - m_lo + 1, F, 2, B, P, L).  A rank that owns a range of m synthesises the maps
 * (replicated, a few per cent of the per-m cost) but transforms and stores only its own blocks —
 * this replaces the reference's (f, b) -> m all-to-all (drift/core/beamtransfer.py:626-640). */
int dm_bt_sht_range(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                    int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                    const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev);

/* dm_bt_sht_opts: dm_bt_sht_range with the two knobs of healpy.map2alm that cora's sphtrans_* may set
 * (neither healpy nor cora is available here, so which values the reference ends up with is not verifiable;
 * the defaults of every other entry point are niter = 0 and equal weights):
 *   niter        Jacobi refinements: coefficients += analysis(map - synthesis(coefficients)), niter times
 *                (healpy's `iter`, default 3 there).  The synthesis uses every (l, m) of a column, whatever m-range
 *                is requested: with niter > 0 the call transforms all m <= lmax_grp into a private buffer first.
 *   ring_w_host  (4 nside - 1) factors on the equal-area quadrature weight of each ring, or NULL
 *                (healpy's `use_weights` multiplies by 1 + w_ring from its data files).
 * Replaces: the same call sites as dm_bt_sht (drift/core/telescope.py:1179-1191, :1288-1312). */
int dm_bt_sht_opts(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                   int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                   const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev, int niter,
                   const double* ring_w_host);

/* dm_bt_columns: dm_bt_maps + dm_bt_sht_range in one call — the visibility response of every pixel is formed in
 * registers inside the ring transform (v_mfma_f64_4x4x4 over the pixels of a ring) and the (ncol, P, npix) Stokes
 * maps are never written to memory; pixels below the horizon are skipped.  Arguments as for those two calls.
 * Returns when the work is queued on the context's stream (host arrays are copied before it returns): call dm_ctx_sync
 * before reading beam_m_dev from the host or from another stream.
 * Replaces: _beam_map_single + _transfer_single + the +/-m fold (drift/core/telescope.py:1156-1193, :1268-1316;
 * drift/util/_fast_tools.pyx:18-164; drift/core/beamtransfer.py:620-624). */
int dm_bt_columns(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                  int polarised, int nbeam, const double* beams_dev, int ncol, const double* uv_host, const int* bi_host,
                  const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, const int* col_f_host,
                  const int* col_b_host, const int* col_lmax_host, void* beam_m_dev, const double* ring_w_host);

/* dm_bt_columns_c: the same for COMPLEX field patterns — beams_dev holds nbeam maps of npix * ncomp complex128 (zero
 * below the horizon), the map values are formed as drift/util/_fast_tools.pyx:169-242 (_construct_pol_complex) and
 * drift/core/telescope.py:1156-1176 form them, inside the ring transform. */
int dm_bt_columns_c(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                    int polarised, int nbeam, const void* beams_dev, int ncol, const double* uv_host, const int* bi_host,
                    const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, const int* col_f_host,
                    const int* col_b_host, const int* col_lmax_host, void* beam_m_dev, const double* ring_w_host);

/* dm_bt_columns_iter: dm_bt_columns / dm_bt_columns_c (complex_beams != 0) with healpy.map2alm's `iter`: after the plain
 * quadrature the coefficients are refined niter times,  a <- a_0 + a - (A o S) a  with A = map2alm(iter = 0), S = alm2map —
 * what healpy does with the residual map, carried out in harmonic space: per m one real Gram matrix of the ring functions
 * under the quadrature (the same for every column) plus the explicit alias terms of the few dozen polar rings with fewer
 * than 2 lmax + 1 pixels.  No Stokes map and no residual map is ever formed; blocks are bit-identical under any partition
 * of m.  healpy's documented default is iter = 3; what cora passes is not verifiable here.  niter = 0 is
 * dm_bt_columns.  Returns when the work is queued on the context's stream.
 * Replaces: the same call sites as dm_bt_columns (cora.util.hputil.sphtrans_complex[_pol] -> healpy.map2alm behind
 * drift/core/telescope.py:1179-1191, :1288-1312). */
int dm_bt_columns_iter(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                       int polarised, int nbeam, const void* beams_dev, int complex_beams, int ncol, const double* uv_host,
                       const int* bi_host, const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B,
                       const int* col_f_host, const int* col_b_host, const int* col_lmax_host, void* beam_m_dev,
                       const double* ring_w_host, int niter);

/* dm_bt_alias_info: which m the refinement of an nside group couples through the polar rings (host only, no GPU work):
 * n_alias_rings per cap and mcut (-1: none).  A refined call whose m-range starts at or below mcut transforms
 * m = 0 .. max(m_hi, mcut) internally; callers size their column chunks with it. */
int dm_bt_alias_info(int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised, int lmax_grp,
                     int* n_alias_rings, int* mcut);

/* ---- bit truncation of beam-transfer blocks before they are written ------------------------------- */
/* In place on `nrows` rows of `ncols` complex128 values (`ld` elements between rows): every real and
 * imaginary part is rounded to the coarsest multiple of a power of two that keeps its error below
 *   err = max(prec * |z|, prec_max_row * max_j |z_row,j|),
 * ties to even — trailing mantissa bits become zero and the byte planes compress.  Rows are the runs
 * over l of the m-ordered blocks (the reference truncates `m_array.reshape(-1, nl)`).
 * Replaces: caput.truncate.bit_truncate_max_complex at drift/core/beamtransfer.py:641-646 (caput is not
 * vendored with the reference; restated from its documented contract, see oracle/truncate.py). */
int dm_bit_truncate_max_complex(dm_ctx* ctx, void* data_dev, int64_t nrows, int ncols, int64_t ld, double prec,
                                double prec_max_row);

#ifdef __cplusplus
}
#endif
#endif /* DRIFTMI_H */
