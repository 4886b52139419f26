// dm_btgen.hip — beam-transfer matrix generation on the GPU (gfx950).
//
// Replaces, for cylinder telescopes:
//   drift/telescope/cylbeam.py:101-212          beam_amp / beam_x / beam_y on HEALPix pixels
//   drift/util/_fast_tools.pyx:18-164           fringe, _construct_pol_real (the OpenMP pixel kernels)
//   drift/core/telescope.py:1156-1193, 1268-1316  _beam_map_single, _transfer_single (SHT via cora->healpy)
//   drift/core/telescope.py:755-830             transfer_matrices
//   drift/core/beamtransfer.py:620-624, :663    +/-m fold and compact m-ordered layout
//
// Pipeline for one group of (frequency, baseline) columns sharing a HEALPix nside:
//   1. bt_beam     field pattern of every (frequency, beam class) on the pixel centres   (HBM-bound, sincos+exp)
//   2. bt_omega    beam solid angles  (4 pi / n) sum h |b|^2                              (reduction)
//   3. bt_maps     h * fringe * (b_i x b_j) / sqrt(O_i O_j) -> 1 or 4 Stokes maps         (HBM-bound)
//   4. ring DFT    G[m, ring, col] = sum_j map[col, pix(ring, j)] e^{+i m phi_j}, -mmax <= m <= mmax
//                  one grouped ZGEMM per ring against a twiddle table                      (MFMA)
//   5. Legendre    beam_m[m][f, +/-, b, p, l] = sum_ring lambda_lm(ring) G[+/-m, ring, col]
//                  real x complex grouped GEMM per (m, f, sign, Stokes term)               (MFMA)
//      The reference's `conj(SHT(conj(map)))` (telescope.py:1189-1191) turns e^{-im phi} into
//      e^{+im phi}; the (-1)^m conj fold of the -m half (beamtransfer.py:622) turns into
//      "conjugate G[-m]" with the SAME real Legendre matrices as +m, so step 5 writes the
//      reference's (F, 2, B, P, L) beam_m layout directly — no fold pass, no transpose.
//   6. bt_mask     zero l > lmax(baseline, frequency) (telescope.py:792-802)
//
// The SHT itself is restated from the published HEALPix algorithm (equal weights,
// no iterations): see oracle/btgen.py for the pinning status of that boundary.
#include "dm_common.h"
#include <cstring>
#include "dm_kernels.h"
#include "../../include/driftmi.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>

namespace {

constexpr double kPi = 3.14159265358979323846;

struct ring_geo {
  const double* cth;   // cos(theta) per ring
  const double* sth;   // sin(theta) per ring
  const double* phi0;  // phi of first pixel
  const int* nphi;     // pixels in ring
  const int* start;    // first pixel index
  int nring;
  int npix;
};

__device__ __forceinline__ int ring_of_pixel(const ring_geo& g, int pix) {
  // binary search over ring starts
  int lo = 0, hi = g.nring - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (g.start[mid] <= pix) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct frame3 {
  double x[3], y[3], z[3];  // East, North, up (telescope frame in sky cartesian coordinates)
};

// natural cubic spline evaluation with searchsorted(side="left") interval choice
__device__ __forceinline__ double spline_eval(const double* __restrict__ x, const double* __restrict__ y,
                                              const double* __restrict__ y2, int n, double xv) {
  int lo = 0, hi = n;  // first index with x[idx] >= xv
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (x[mid] < xv) lo = mid + 1; else hi = mid;
  }
  int khi = min(max(lo, 1), n - 1);
  int klo = khi - 1;
  double h = x[khi] - x[klo];
  double a = (x[khi] - xv) / h;
  double b = (xv - x[klo]) / h;
  return a * y[klo] + b * y[khi] + ((a * a * a - a) * y2[klo] + (b * b * b - b) * y2[khi]) * (h * h) / 6.0;
}

// kind 0: unpolarised amplitude (1 component); 1: X dipole; 2: Y dipole (2 components theta, phi)
__device__ __forceinline__ void bt_beam_pixel(const ring_geo& g, const frame3& fr, int kind, const double* __restrict__ tx,
                                              const double* __restrict__ ty, const double* __restrict__ ty2, int ntab,
                                              double alpha_ns, double* __restrict__ out, int pix) {
  const int r = ring_of_pixel(g, pix);
  const int j = pix - g.start[r];
  const double phi = g.phi0[r] + 2.0 * kPi * (double)j / (double)g.nphi[r];
  double sp, cp;
  sincos(phi, &sp, &cp);
  const double st = g.sth[r], ct = g.cth[r];
  const double n0 = st * cp, n1 = st * sp, n2 = ct;
  const double dz = n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2];
  const double hz = dz > 0.0 ? 1.0 : 0.0;
  const double dx = n0 * fr.x[0] + n1 * fr.x[1] + n2 * fr.x[2];
  const double dy = n0 * fr.y[0] + n1 * fr.y[1] + n2 * fr.y[2];
  const double ew = spline_eval(tx, ty, ty2, ntab, dx);
  const double s2 = dy * dy;
  const double ns = exp(-alpha_ns * s2 / (1.0 - s2 + 1e-100));
  const double amp = ew * ns * hz;
  if (kind == 0) {
    out[pix] = amp;
    return;
  }
  // dipole projected on (thetahat, phihat), normalised to unit length (cylbeam.polpattern)
  const double* dip = (kind == 1) ? fr.x : fr.y;
  const double t0 = ct * cp, t1 = ct * sp, t2 = -st;
  const double p0 = -sp, p1 = cp;
  double pt = t0 * dip[0] + t1 * dip[1] + t2 * dip[2];
  double pp = p0 * dip[0] + p1 * dip[1];
  double nrm = hypot(pt, pp);
  if (nrm == 0.0) nrm = 1.0;
  out[2 * (size_t)pix] = amp * pt / nrm;
  out[2 * (size_t)pix + 1] = amp * pp / nrm;
}
__global__ void bt_beam_kernel(ring_geo g, frame3 fr, int kind, const double* __restrict__ tx,
                               const double* __restrict__ ty, const double* __restrict__ ty2, int ntab,
                               double alpha_ns, double* __restrict__ out) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= g.npix) return;
  bt_beam_pixel(g, fr, kind, tx, ty, ty2, ntab, alpha_ns, out, pix);
}
// several patterns in one launch: blockIdx.y = beam
struct bt_beam_desc { int kind, ntab; const double* tx; const double* ty; const double* ty2; double alpha_ns; double* out; };
__global__ void bt_beams_kernel(ring_geo g, frame3 fr, const bt_beam_desc* __restrict__ descs) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= g.npix) return;
  const bt_beam_desc d = descs[blockIdx.y];
  bt_beam_pixel(g, fr, d.kind, d.tx, d.ty, d.ty2, d.ntab, d.alpha_ns, d.out, pix);
}

// the solid angle in two steps that keep the chip busy: NOMEGA partial sums per beam, then their sum in a fixed order
constexpr int NOMEGA = 64;
__global__ __launch_bounds__(256) void bt_omega_part_kernel(ring_geo g, const double* __restrict__ beams, int ncomp, size_t bstride,
                                                            double* __restrict__ part) {
  __shared__ double red[4];
  const double* b = beams + (size_t)blockIdx.x * bstride;
  const int per = (g.npix + NOMEGA - 1) / NOMEGA;
  const int p0 = blockIdx.y * per, p1 = min(p0 + per, g.npix);
  double s = 0.0;
  for (int pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    double v = 0.0;
    for (int c = 0; c < ncomp; ++c) {
      const double x = b[(size_t)pix * ncomp + c];
      v += x * x;
    }
    s += v;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)blockIdx.x * NOMEGA + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
}
__global__ void bt_omega_fin_kernel(const double* __restrict__ part, int nbeam, double scale, double* __restrict__ omega) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbeam) return;
  double s = 0.0;
  for (int i = 0; i < NOMEGA; ++i) s += part[(size_t)b * NOMEGA + i];
  omega[b] = s * scale;
}

// omega[b] = (4 pi / npix) sum_pix h |beam_b|^2 ; one block per beam (beams already carry h)
__global__ void bt_omega_kernel(ring_geo g, frame3 fr, const double* __restrict__ beams, int ncomp, size_t bstride,
                                double* __restrict__ omega) {
  __shared__ double red[4];
  const double* b = beams + (size_t)blockIdx.x * bstride;
  double s = 0.0;
  for (int pix = threadIdx.x; pix < g.npix; pix += blockDim.x) {
    // the reference multiplies by the horizon again; beams are zero below it already, and h in {0,1}
    double v = 0.0;
    for (int c = 0; c < ncomp; ++c) {
      double x = b[(size_t)pix * ncomp + c];
      v += x * x;
    }
    s += v;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) omega[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * 4.0 * kPi / (double)g.npix;
}

// maps[col][p][pix]
__global__ void bt_maps_kernel(ring_geo g, frame3 fr, int polarised, int ncol, const double* __restrict__ uv,
                               const int* __restrict__ bi, const int* __restrict__ bj,
                               const double* __restrict__ beams, size_t bstride, const double* __restrict__ omega,
                               cplx* __restrict__ maps) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int col = blockIdx.y;
  if (pix >= g.npix) return;
  const int r = ring_of_pixel(g, pix);
  const int j = pix - g.start[r];
  const double phi = g.phi0[r] + 2.0 * kPi * (double)j / (double)g.nphi[r];
  double sp, cp;
  sincos(phi, &sp, &cp);
  const double st = g.sth[r], ct = g.cth[r];
  const double n0 = st * cp, n1 = st * sp, n2 = ct;
  const double hz = (n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2]) > 0.0 ? 1.0 : 0.0;
  // uv3 = u * East + v * North
  const double u = uv[2 * col], v = uv[2 * col + 1];
  const double du = n0 * (u * fr.x[0] + v * fr.y[0]) + n1 * (u * fr.x[1] + v * fr.y[1]) + n2 * (u * fr.x[2] + v * fr.y[2]);
  double sf, cf;
  sincos(2.0 * kPi * du, &sf, &cf);
  const int ib = bi[col], jb = bj[col];
  const double pre = hz / sqrt(omega[ib] * omega[jb]);
  const double tre = pre * cf, tim = pre * sf;
  if (!polarised) {
    const double bb = beams[(size_t)ib * bstride + pix] * beams[(size_t)jb * bstride + pix];
    maps[(size_t)col * g.npix + pix] = make_double2(tre * bb, tim * bb);
  } else {
    const double* a = beams + (size_t)ib * bstride + 2 * (size_t)pix;
    const double* b = beams + (size_t)jb * bstride + 2 * (size_t)pix;
    const double a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    const double sI = a0 * b0 + a1 * b1, sQ = a0 * b0 - a1 * b1, sU = a0 * b1 + a1 * b0, sV = a0 * b1 - a1 * b0;
    cplx* m = maps + (size_t)col * 4 * g.npix + pix;
    m[0] = make_double2(tre * sI, tim * sI);
    m[(size_t)g.npix] = make_double2(tre * sQ, tim * sQ);
    m[2 * (size_t)g.npix] = make_double2(tre * sU, tim * sU);
    m[3 * (size_t)g.npix] = make_double2(-tim * sV, tre * sV);  // 1j * tc * sV
  }
}


// ---- complex field patterns (_fast_tools.pyx:167-242, telescope.py:1156-1176): beams are (npix * ncomp) complex128,
// the second beam enters conjugated, the solid angles are sums of |b|^2 ------------------------------------
__global__ void bt_omega_c_kernel(ring_geo g, const cplx* __restrict__ beams, int ncomp, size_t bstride,
                                  double* __restrict__ omega) {
  __shared__ double red[4];
  const cplx* b = beams + (size_t)blockIdx.x * bstride;
  double s = 0.0;
  for (int pix = threadIdx.x; pix < g.npix; pix += blockDim.x) {
    double v = 0.0;
    for (int c = 0; c < ncomp; ++c) v += cabs2(b[(size_t)pix * ncomp + c]);  // beams already carry the horizon
    s += v;
  }
  s = dm_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) omega[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * 4.0 * kPi / (double)g.npix;
}

__global__ void bt_maps_c_kernel(ring_geo g, frame3 fr, int polarised, int ncol, const double* __restrict__ uv,
                                 const int* __restrict__ bi, const int* __restrict__ bj,
                                 const cplx* __restrict__ beams, size_t bstride, const double* __restrict__ omega,
                                 cplx* __restrict__ maps) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int col = blockIdx.y;
  if (pix >= g.npix) return;
  const int r = ring_of_pixel(g, pix);
  const int j = pix - g.start[r];
  const double phi = g.phi0[r] + 2.0 * kPi * (double)j / (double)g.nphi[r];
  double sp, cp;
  sincos(phi, &sp, &cp);
  const double st = g.sth[r], ct = g.cth[r];
  const double n0 = st * cp, n1 = st * sp, n2 = ct;
  const double hz = (n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2]) > 0.0 ? 1.0 : 0.0;
  const double u = uv[2 * col], v = uv[2 * col + 1];
  const double du = n0 * (u * fr.x[0] + v * fr.y[0]) + n1 * (u * fr.x[1] + v * fr.y[1]) + n2 * (u * fr.x[2] + v * fr.y[2]);
  double sf, cf;
  sincos(2.0 * kPi * du, &sf, &cf);
  const int ib = bi[col], jb = bj[col];
  const double pre = hz / sqrt(omega[ib] * omega[jb]);
  const cplx tc = make_double2(pre * cf, pre * sf);
  if (!polarised) {
    const cplx bb = cmulc(beams[(size_t)ib * bstride + pix], beams[(size_t)jb * bstride + pix]);  // b_i conj(b_j)
    maps[(size_t)col * g.npix + pix] = cmul(tc, bb);
  } else {
    const cplx* a = beams + (size_t)ib * bstride + 2 * (size_t)pix;
    const cplx* b = beams + (size_t)jb * bstride + 2 * (size_t)pix;
    const cplx a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    const cplx p00 = cmulc(a0, b0), p11 = cmulc(a1, b1), p01 = cmulc(a0, b1), p10 = cmulc(a1, b0);
    cplx* m = maps + (size_t)col * 4 * g.npix + pix;
    m[0] = cmul(tc, cadd(p00, p11));
    m[(size_t)g.npix] = cmul(tc, csub(p00, p11));
    m[2 * (size_t)g.npix] = cmul(tc, cadd(p01, p10));
    const cplx sv = cmul(tc, csub(p01, p10));
    m[3 * (size_t)g.npix] = make_double2(-sv.y, sv.x);  // 1j * tc * (...)
  }
}

// ---- fused map synthesis + ring DFT (no Stokes maps in HBM) ---------------------------------------------
// G[mm][ring][col * P + p] = w_ring * sum_j exp(i m_mm phi_j) * map_p(col, ring, j) with the map value
// h * fringe * (b_i x b_j) / sqrt(O_i O_j) formed in registers right before it is used.  The sum over the pixels
// of a ring is the K dimension of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction, the
// one fp64 MFMA shape that issues at the datasheet rate on gfx950):
//   lane l = 16 k + 4 g + t supplies   A_g[i = t][k] = twiddle of m-value t at pixel k of the current quad (same for all g)
//                                      B_g[k][j = t] = map value of pixel k, column 16 cg + 4 g + t
//   and receives                       D_g[i][j] in lane 16 i + 4 g + j = partial G of m-value i, column 16 cg + (l & 15).
// A wave owns one ring and NCG groups of 16 columns; per pixel quad the geometry (n.x, n.y, horizon) and the NMG x 4
// twiddles are computed once and shared by all its column groups, per (quad, group) a lane does one sincos (the
// fringe of ITS pixel and column), loads the two beams and issues NMG * P * 4 MFMAs.  m-values beyond 4 NMG take
// further passes (blockIdx.z), which repeat the synthesis: with P * NMG * 4 >= 16 MFMAs behind every sincos the
// matrix pipe, not the VALU, bounds the kernel.
// Algorithmic bytes per (pixel, column): 2 beams x ncomp x 8 B from L1/L2 (beams are shared by all baselines of a
// frequency), 16 P nm / npix-th of the G row written — compute-bound, see DESIGN.md section 4.5.
struct fdft_col { double u, v, pre; int bi, bj; };   // pre = 1 / sqrt(Omega_i Omega_j); bi < 0: padding
// Map values of one (pixel, column) from COMPLEX field patterns (drift/util/_fast_tools.pyx:169-242, _construct_pol_complex,
// and telescope.py:1156-1176 with complex beams): a, b point at the NCOMP complex components of the two beams at this
// pixel (interleaved re, im), (tre, tim) = pre * fringe.  The real-pattern kernels keep their own (cheaper) expressions.
template <int P>
__device__ __forceinline__ void bt_synth_complex(const double* __restrict__ a, const double* __restrict__ b, double tre, double tim,
                                                 double (&m_re)[P], double (&m_im)[P]) {
  const cplx tc = make_double2(tre, tim);
  if constexpr (P == 1) {
    const cplx bb = cmulc(make_double2(dm_ldg(a), dm_ldg(a, 1)), make_double2(dm_ldg(b), dm_ldg(b, 1)));  // b_i conj(b_j)
    const cplx v = cmul(tc, bb);
    m_re[0] = v.x; m_im[0] = v.y;
  } else {
    const cplx a0 = make_double2(dm_ldg(a), dm_ldg(a, 1)), a1 = make_double2(dm_ldg(a, 2), dm_ldg(a, 3));
    const cplx b0 = make_double2(dm_ldg(b), dm_ldg(b, 1)), b1 = make_double2(dm_ldg(b, 2), dm_ldg(b, 3));
    const cplx p00 = cmulc(a0, b0), p11 = cmulc(a1, b1), p01 = cmulc(a0, b1), p10 = cmulc(a1, b0);
    const cplx vI = cmul(tc, cadd(p00, p11)), vQ = cmul(tc, csub(p00, p11)), vU = cmul(tc, cadd(p01, p10));
    const cplx sv = cmul(tc, csub(p01, p10));
    m_re[0] = vI.x; m_im[0] = vI.y;
    m_re[1] = vQ.x; m_im[1] = vQ.y;
    m_re[2] = vU.x; m_im[2] = vU.y;
    m_re[3] = -sv.y; m_im[3] = sv.x;   // 1j * tc * (p01 - p10)
  }
}
// smallest |m| among the rows lo .. hi of a ring-transform pass (rows [0, cnt) are +m_lo + row, rows [cnt, 2 cnt) the -m)
__device__ __forceinline__ int bt_pass_min_m(int lo, int hi, int m_lo, int cnt) {
  return hi < cnt ? m_lo + lo : (lo >= cnt ? m_lo + lo - cnt : m_lo);
}
__global__ __launch_bounds__(256) void bt_fdft_pre_kernel(fdft_col* __restrict__ cols, int ncol, const double* __restrict__ omega) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncol) return;
  cols[c].pre = 1.0 / sqrt(omega[cols[c].bi] * omega[cols[c].bj]);
}
template <int P, int NMG, int NCG, bool CB = false>
__global__ __launch_bounds__(256) void bt_fused_dft_kernel(ring_geo g, frame3 fr, const double* __restrict__ beams, size_t bstride,
                                                           const fdft_col* __restrict__ cols, int ncol16, int m_lo, int cnt,
                                                           const double* __restrict__ ring_w, cplx* __restrict__ G, int ncp,
                                                           const int* __restrict__ ring_list, const int* __restrict__ mskip) {
  constexpr int NCOMP = P == 4 ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = ring_list ? ring_list[blockIdx.y] : (int)blockIdx.y;   // (the polar caps only, when the belt goes by FFT)
  const int cg0 = (blockIdx.x * 4 + wave) * NCG;          // first column group of this wave
  if (cg0 >= ncol16) return;                              // (wave-uniform; no workgroup barrier follows)
  const int mg0 = blockIdx.z * NMG;                       // first group of four m-values of this pass
  const int nm = 2 * cnt;
  const int k = lane >> 4, t = lane & 3;
  const int nphi = g.nphi[r];
  const double phi0 = g.phi0[r], st = g.sth[r], ct = g.cth[r];
  const int pix0 = g.start[r];
  // rows whose |m| has reached mskip[r] are exact zeros (bt_ring_skip_lookup); a pass that holds only such rows does no work
  const int msk = mskip ? mskip[r] : 0x7fffffff;
  const int nloop = bt_pass_min_m(mg0 * 4, min(nm, (mg0 + NMG) * 4) - 1, m_lo, cnt) >= msk ? 0 : nphi;
  // m-values of this lane's twiddle rows: rows [0, cnt) are +m, rows [cnt, 2 cnt) are -m
  int mval[NMG];
  bool mok[NMG];
#pragma unroll
  for (int a = 0; a < NMG; ++a) {
    const int mm = (mg0 + a) * 4 + t;
    mok[a] = mm < nm;
    mval[a] = mm < cnt ? m_lo + mm : -(m_lo + mm - cnt);
  }
  // per-column constants of this wave's groups live in LDS (read once per (quad, group); in registers they would cost
  // 8 VGPRs per group and the kernel one of its two waves per SIMD)
  __shared__ fdft_col s_cd[4][NCG * 16];
  for (int c = lane; c < NCG * 16; c += 64) {
    const int cg = cg0 + (c >> 4);
    s_cd[wave][c] = cg < ncol16 ? cols[(size_t)cg * 16 + (c & 15)] : fdft_col{0.0, 0.0, 0.0, -1, -1};
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave reads back only what it wrote itself
  __builtin_amdgcn_wave_barrier();
  double acc_re[NCG][NMG][P], acc_im[NCG][NMG][P];
#pragma unroll
  for (int c = 0; c < NCG; ++c)
#pragma unroll
    for (int a = 0; a < NMG; ++a)
#pragma unroll
      for (int p = 0; p < P; ++p) acc_re[c][a][p] = acc_im[c][a][p] = 0.0;

  // The twiddles exp(i m phi_j) advance by four pixels per step: one complex rotation each, re-seeded exactly every
  // FDFT_RS steps (a rotation costs 6 flops against the ~60 of a sincos plus a 64-bit modulo; the drift over 15
  // rotations stays below 1e-14, the exact seed bounds it).
  constexpr int FDFT_RS = 16;
  double rot_re[NMG], rot_im[NMG];
#pragma unroll
  for (int a = 0; a < NMG; ++a) {
    const long long m4 = (4ll * mval[a]) % nphi;
    sincos(2.0 * kPi * (double)m4 / (double)nphi, &rot_im[a], &rot_re[a]);
  }
  double sp = 0.0, cp = 1.0;
  double tw_re[NMG], tw_im[NMG];
#pragma unroll
  for (int a = 0; a < NMG; ++a) tw_re[a] = tw_im[a] = 0.0;
  int step = 0;
  for (int q = 0; q < nloop; q += 4, ++step) {
    const int j = q + k;
    const bool pv = j < nphi;
    const int jj = pv ? j : nphi - 1;
    // the pixel direction is taken exactly for every quad: it enters the map values, which have to come out with
    // the same bits from every kernel that synthesises them (bt_fused_dft2_kernel visits the quads in another order)
    sincos(phi0 + 2.0 * kPi * (double)j / (double)nphi, &sp, &cp);
    if ((step & (FDFT_RS - 1)) == 0) {
#pragma unroll
      for (int a = 0; a < NMG; ++a) {
        // reduce the argument exactly: m * j mod nphi keeps the phase in [0, 2 pi) (as bt_twiddle_kernel)
        const long long mj = ((long long)mval[a] * j) % nphi;
        double s, c;
        sincos((double)mval[a] * phi0 + 2.0 * kPi * (double)mj / (double)nphi, &s, &c);
        tw_re[a] = mok[a] ? c : 0.0;
        tw_im[a] = mok[a] ? s : 0.0;
      }
    } else {
      // explicit rounding order: every instantiation (and therefore every partition of m over ranks) forms the
      // same bits for a given (m, pixel)
#pragma unroll
      for (int a = 0; a < NMG; ++a) {
        const double r2 = __fma_rn(tw_re[a], rot_re[a], -__dmul_rn(tw_im[a], rot_im[a]));
        tw_im[a] = __fma_rn(tw_re[a], rot_im[a], __dmul_rn(tw_im[a], rot_re[a]));
        tw_re[a] = r2;
      }
    }
    const double n0 = st * cp, n1 = st * sp, n2 = ct;
    const double hz = (pv && (n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2]) > 0.0) ? 1.0 : 0.0;
    if (__ballot(hz != 0.0) == 0ull) continue;  // the whole quad lies below the horizon: every map value is zero
    const double nx = n0 * fr.x[0] + n1 * fr.x[1] + n2 * fr.x[2];
    const double ny = n0 * fr.y[0] + n1 * fr.y[1] + n2 * fr.y[2];
    const size_t pix = (size_t)pix0 + jj;
#pragma unroll
    for (int c = 0; c < NCG; ++c) {
      double m_re[P], m_im[P];
      const fdft_col cdc = s_cd[wave][c * 16 + (lane & 15)];
      // branch-free: padding columns and pixels below the horizon read beam 0 and are zeroed through `pre`, so the
      // loads and the sincos of all NCG groups of a quad can be in flight together
      const bool on = cdc.bi >= 0 && hz != 0.0;
      constexpr int BW = CB ? 2 : 1;   // doubles per beam component (complex patterns: interleaved re, im)
      const double* a = beams + (size_t)max(cdc.bi, 0) * bstride + BW * NCOMP * pix;
      const double* b = beams + (size_t)max(cdc.bj, 0) * bstride + BW * NCOMP * pix;
      // the fringe phase in turns: sincospi reduces 2 t exactly (no large-argument path, no branches), which is
      // also closer to the true phase than sin(fl(2 pi t)) once |u| reaches hundreds of wavelengths
      double sf, cf;
      sincospi(2.0 * (cdc.u * nx + cdc.v * ny), &sf, &cf);
      const double pre = on ? cdc.pre : 0.0;
      const double tre = pre * cf, tim = pre * sf;
      if constexpr (CB) {
        bt_synth_complex<P>(a, b, tre, tim, m_re, m_im);
      } else if constexpr (P == 1) {
        const double bb = dm_ldg(a) * dm_ldg(b);
        m_re[0] = tre * bb;
        m_im[0] = tim * bb;
      } else {
        const double a0 = dm_ldg(a), a1 = dm_ldg(a, 1), b0 = dm_ldg(b), b1 = dm_ldg(b, 1);
        const double sI = a0 * b0 + a1 * b1, sQ = a0 * b0 - a1 * b1, sU = a0 * b1 + a1 * b0, sV = a0 * b1 - a1 * b0;
        m_re[0] = tre * sI; m_im[0] = tim * sI;
        m_re[1] = tre * sQ; m_im[1] = tim * sQ;
        m_re[2] = tre * sU; m_im[2] = tim * sU;
        m_re[3] = -tim * sV; m_im[3] = tre * sV;  // 1j * fringe * sV
      }
#pragma unroll
      for (int a = 0; a < NMG; ++a)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          acc_re[c][a][p] = dm_mfma4(tw_re[a], m_re[p], acc_re[c][a][p]);
          acc_im[c][a][p] = dm_mfma4(tw_re[a], m_im[p], acc_im[c][a][p]);
        }
#pragma unroll
      for (int a = 0; a < NMG; ++a)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          acc_re[c][a][p] = dm_mfma4(-tw_im[a], m_im[p], acc_re[c][a][p]);
          acc_im[c][a][p] = dm_mfma4(tw_im[a], m_re[p], acc_im[c][a][p]);
        }
    }
  }
  // lane 16 i + (lane & 15) holds m-row i of the group: G[mm][ring][p * ncol + col] (Stokes-major rows: the Legendre
  // products read ONE Stokes parameter of consecutive columns — 16 lanes x 16 B contiguous instead of every fourth element)
  const double w = ring_w ? ring_w[r] : 1.0;
  const int i = lane >> 4;
#pragma unroll
  for (int c = 0; c < NCG; ++c) {
    const int col = (cg0 + c) * 16 + (lane & 15);
    if (cg0 + c >= ncol16 || col * P >= ncp) continue;
#pragma unroll
    for (int a = 0; a < NMG; ++a) {
      const int mm = (mg0 + a) * 4 + i;
      if (mm >= nm) continue;
      const double wz = (mm < cnt ? m_lo + mm : m_lo + mm - cnt) >= msk ? 0.0 : w;
      cplx* out = G + ((size_t)mm * g.nring + r) * ncp + (size_t)col;
#pragma unroll
      for (int p = 0; p < P; ++p) dm_stg(out, (size_t)p * (ncp / P), make_double2(wz * acc_re[c][a][p], wz * acc_im[c][a][p]));
    }
  }
}

// ---- the same transform with the synthesis SHARED by the four waves of a workgroup ---------------------------
// bt_fused_dft_kernel repeats the map synthesis (a sincospi, two beam loads and the Stokes products per pixel and
// column) for every pass over m: 17 times for a rank's 130 m-values at configs[2], and its MFMA pipe sits idle
// behind that VALU work.  Here the four waves of a workgroup own the SAME ring and the same NCG column groups but
// different m-values (4 NMG each, 16 NMG per pass): per chunk of four pixel quads every wave synthesises ONE quad
// (for all NCG groups) into LDS, and after a barrier every wave runs its own twiddles over all four quads.  The
// synthesis per MFMA drops fourfold, the passes over m fourfold: the kernel is bound by the matrix pipe.
// LDS: two buffers x 4 quads x NCG groups x P x (re, im) x 64 lanes x 8 B (32 KB for <4, *, 1> and <1, *, 4>); one
// barrier per chunk (the buffers alternate: a wave can only reach the second write of a buffer through the barrier
// that every reader of its previous content has passed).
template <int P, int NMG, int NCG, bool CB = false>
__global__ __launch_bounds__(256) void bt_fused_dft2_kernel(ring_geo g, frame3 fr, const double* __restrict__ beams, size_t bstride,
                                                            const fdft_col* __restrict__ cols, int ncol16, int m_lo, int cnt,
                                                            const double* __restrict__ ring_w, cplx* __restrict__ G, int ncp,
                                                            const int* __restrict__ ring_list, const int* __restrict__ mskip) {
  constexpr int NCOMP = P == 4 ? 2 : 1;
  constexpr int QC = 4;  // quads per chunk = waves per workgroup
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = ring_list ? ring_list[blockIdx.y] : (int)blockIdx.y;
  const int cg0 = blockIdx.x * NCG;
  if (cg0 >= ncol16) return;  // uniform over the workgroup
  const int nm = 2 * cnt;
  const int mg0 = (blockIdx.z * 4 + wave) * NMG;  // first group of four m-values of this wave
  const bool has_m = mg0 * 4 < nm;                // (a wave without m-values still synthesises its quads)
  // rows whose |m| has reached mskip[r] are exact zeros (bt_ring_skip_lookup): a workgroup whose pass holds only such rows
  // does no work (uniform: the barriers below are skipped by all four waves), a wave whose own rows are all of that kind
  // still synthesises its quads for the others but issues no MFMAs
  const int msk = mskip ? mskip[r] : 0x7fffffff;
  const int wg_row0 = blockIdx.z * 16 * NMG;
  const bool wg_skip = bt_pass_min_m(wg_row0, min(nm, wg_row0 + 16 * NMG) - 1, m_lo, cnt) >= msk;
  const bool use_m = has_m && bt_pass_min_m(mg0 * 4, min(nm, (mg0 + NMG) * 4) - 1, m_lo, cnt) < msk;
  const int k = lane >> 4, t = lane & 3;
  const int nphi = g.nphi[r];
  const double phi0 = g.phi0[r], st = g.sth[r], ct = g.cth[r];
  const int pix0 = g.start[r];
  int mval[NMG];
  bool mok[NMG];
#pragma unroll
  for (int a = 0; a < NMG; ++a) {
    const int mm = (mg0 + a) * 4 + t;
    mok[a] = mm < nm;
    mval[a] = mm < cnt ? m_lo + mm : -(m_lo + mm - cnt);
  }
  __shared__ fdft_col s_cd[NCG * 16];
  __shared__ double s_map[2][QC][NCG][P][2][64];
  __shared__ int s_on[2][QC];
  for (int c = threadIdx.x; c < NCG * 16; c += 256) {
    const int cg = cg0 + (c >> 4);
    s_cd[c] = cg < ncol16 ? cols[(size_t)cg * 16 + (c & 15)] : fdft_col{0.0, 0.0, 0.0, -1, -1};
  }
  __syncthreads();
  double acc_re[NCG][NMG][P], acc_im[NCG][NMG][P];
#pragma unroll
  for (int c = 0; c < NCG; ++c)
#pragma unroll
    for (int a = 0; a < NMG; ++a)
#pragma unroll
      for (int p = 0; p < P; ++p) acc_re[c][a][p] = acc_im[c][a][p] = 0.0;

  // twiddle state: advances by 4 pixels per quad; exact re-seed every 16 quads (as bt_fused_dft_kernel)
  constexpr int TW_RS = 16;
  double rot_re[NMG], rot_im[NMG], tw_re[NMG], tw_im[NMG];
#pragma unroll
  for (int a = 0; a < NMG; ++a) {
    const long long m4 = (4ll * mval[a]) % nphi;
    sincos(2.0 * kPi * (double)m4 / (double)nphi, &rot_im[a], &rot_re[a]);
    tw_re[a] = tw_im[a] = 0.0;
  }
  const int nchunk = wg_skip ? 0 : (nphi + 4 * QC - 1) / (4 * QC);
  for (int ch = 0; ch < nchunk; ++ch) {
    const int buf = ch & 1;
    // ---- synthesis of quad `wave` of this chunk, all NCG groups
    {
      const int j = 16 * ch + 4 * wave + k;
      const bool pv = j < nphi;
      const int jj = pv ? j : nphi - 1;
      double sp, cp;
      sincos(phi0 + 2.0 * kPi * (double)j / (double)nphi, &sp, &cp);  // exact, as in bt_fused_dft_kernel: same map bits
      const double n0 = st * cp, n1 = st * sp, n2 = ct;
      const double hz = (pv && (n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2]) > 0.0) ? 1.0 : 0.0;
      const bool any = __ballot(hz != 0.0) != 0ull;
      if (lane == 0) s_on[buf][wave] = any ? 1 : 0;
      if (any) {  // (wave-uniform) a quad wholly below the horizon is skipped by every reader
        const double nx = n0 * fr.x[0] + n1 * fr.x[1] + n2 * fr.x[2];
        const double ny = n0 * fr.y[0] + n1 * fr.y[1] + n2 * fr.y[2];
        const size_t pix = (size_t)pix0 + jj;
#pragma unroll
        for (int c = 0; c < NCG; ++c) {
          const fdft_col cdc = s_cd[c * 16 + (lane & 15)];
          const bool on = cdc.bi >= 0 && hz != 0.0;
          constexpr int BW = CB ? 2 : 1;
          const double* a = beams + (size_t)max(cdc.bi, 0) * bstride + BW * NCOMP * pix;
          const double* b = beams + (size_t)max(cdc.bj, 0) * bstride + BW * NCOMP * pix;
          double sf, cf;
          sincospi(2.0 * (cdc.u * nx + cdc.v * ny), &sf, &cf);
          const double pre = on ? cdc.pre : 0.0;
          const double tre = pre * cf, tim = pre * sf;
          if constexpr (CB) {
            double m_re[P], m_im[P];
            bt_synth_complex<P>(a, b, tre, tim, m_re, m_im);
#pragma unroll
            for (int p = 0; p < P; ++p) {
              s_map[buf][wave][c][p][0][lane] = m_re[p];
              s_map[buf][wave][c][p][1][lane] = m_im[p];
            }
          } else if constexpr (P == 1) {
            const double bb = dm_ldg(a) * dm_ldg(b);
            s_map[buf][wave][c][0][0][lane] = tre * bb;
            s_map[buf][wave][c][0][1][lane] = tim * bb;
          } else {
            const double a0 = dm_ldg(a), a1 = dm_ldg(a, 1), b0 = dm_ldg(b), b1 = dm_ldg(b, 1);
            const double sI = a0 * b0 + a1 * b1, sQ = a0 * b0 - a1 * b1, sU = a0 * b1 + a1 * b0, sV = a0 * b1 - a1 * b0;
            s_map[buf][wave][c][0][0][lane] = tre * sI; s_map[buf][wave][c][0][1][lane] = tim * sI;
            s_map[buf][wave][c][1][0][lane] = tre * sQ; s_map[buf][wave][c][1][1][lane] = tim * sQ;
            s_map[buf][wave][c][2][0][lane] = tre * sU; s_map[buf][wave][c][2][1][lane] = tim * sU;
            s_map[buf][wave][c][3][0][lane] = -tim * sV; s_map[buf][wave][c][3][1][lane] = tre * sV;  // 1j * fringe * sV
          }
        }
      }
    }
    __syncthreads();
    if (!use_m) continue;  // (wave-uniform; the barrier above is still reached every chunk)
    // ---- this wave's m-values over the four quads of the chunk
#pragma unroll
    for (int qi = 0; qi < QC; ++qi) {
      const int quad = ch * QC + qi;
      const int j = 4 * quad + k;
      if (4 * quad >= nphi) break;
      if ((quad & (TW_RS - 1)) == 0) {
#pragma unroll
        for (int a = 0; a < NMG; ++a) {
          const long long mj = ((long long)mval[a] * j) % nphi;
          double s_, c_;
          sincos((double)mval[a] * phi0 + 2.0 * kPi * (double)mj / (double)nphi, &s_, &c_);
          tw_re[a] = mok[a] ? c_ : 0.0;
          tw_im[a] = mok[a] ? s_ : 0.0;
        }
      } else {
#pragma unroll
        for (int a = 0; a < NMG; ++a) {
          const double r2 = __fma_rn(tw_re[a], rot_re[a], -__dmul_rn(tw_im[a], rot_im[a]));
          tw_im[a] = __fma_rn(tw_re[a], rot_im[a], __dmul_rn(tw_im[a], rot_re[a]));
          tw_re[a] = r2;
        }
      }
      if (!s_on[buf][qi]) continue;
#pragma unroll
      for (int c = 0; c < NCG; ++c) {
        double m_re[P], m_im[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          m_re[p] = s_map[buf][qi][c][p][0][lane];
          m_im[p] = s_map[buf][qi][c][p][1][lane];
        }
#pragma unroll
        for (int a = 0; a < NMG; ++a)
#pragma unroll
          for (int p = 0; p < P; ++p) {
            acc_re[c][a][p] = dm_mfma4(tw_re[a], m_re[p], acc_re[c][a][p]);
            acc_im[c][a][p] = dm_mfma4(tw_re[a], m_im[p], acc_im[c][a][p]);
          }
#pragma unroll
        for (int a = 0; a < NMG; ++a)
#pragma unroll
          for (int p = 0; p < P; ++p) {
            acc_re[c][a][p] = dm_mfma4(-tw_im[a], m_im[p], acc_re[c][a][p]);
            acc_im[c][a][p] = dm_mfma4(tw_im[a], m_re[p], acc_im[c][a][p]);
          }
      }
    }
  }
  if (!has_m) return;
  const double w = ring_w ? ring_w[r] : 1.0;
  const int i = lane >> 4;
#pragma unroll
  for (int c = 0; c < NCG; ++c) {
    const int col = (cg0 + c) * 16 + (lane & 15);
    if (cg0 + c >= ncol16 || col * P >= ncp) continue;
#pragma unroll
    for (int a = 0; a < NMG; ++a) {
      const int mm = (mg0 + a) * 4 + i;
      if (mm >= nm) continue;
      const double wz = (mm < cnt ? m_lo + mm : m_lo + mm - cnt) >= msk ? 0.0 : w;
      cplx* out = G + ((size_t)mm * g.nring + r) * ncp + (size_t)col;
#pragma unroll
      for (int p = 0; p < P; ++p) dm_stg(out, (size_t)p * (ncp / P), make_double2(wz * acc_re[c][a][p], wz * acc_im[c][a][p]));
    }
  }
}

// ---- the equatorial belt by FFT -----------------------------------------------------------------------------
// The 2 nside + 1 rings of the belt all have N = 4 nside pixels (a power of two): 2/3 of the sphere.  There the sum over
// the pixels of a ring is a length-N FFT per (column, Stokes map) — 5 N log2 N flops for EVERY m at once against
// 8 N flops per m-value of the matrix form (19 x fewer for a rank's 130 m-values at nside 512).  One workgroup owns a
// ring and walks over `cpw` columns: the pixel geometry of its NPT pixels per thread is computed once, per column the
// map values go to LDS in bit-reversed order, a radix-8 pass and radix-4 passes (one barrier each) transform them in
// place, and the wanted rows  G[+-m] = w exp(i m phi_0) X[(+-m) mod N]  are written out.  The map values are formed
// by the same expressions as in the two kernels above (the caps still go through those).
// LDS: P N complex values + N / 2 twiddles (144 KB at nside 512 with four Stokes maps; beyond 160 KB the caller
// keeps the matrix form for the belt too).  The result of a ring does not depend on the m-range asked for.
template <int P, int NPT, int TPB, bool CB = false>
__global__ __launch_bounds__(TPB) void bt_fused_fft_kernel(ring_geo g, frame3 fr, const double* __restrict__ beams, size_t bstride,
                                                           const fdft_col* __restrict__ cols, int ncol, int m_lo, int cnt,
                                                           const double* __restrict__ ring_w, cplx* __restrict__ G, int ncp,
                                                           int ring0, int cpw) {
  constexpr int NCOMP = P == 4 ? 2 : 1;
  extern __shared__ __align__(16) unsigned char fft_smem[];
  const int tid = threadIdx.x;
  const int r = ring0 + blockIdx.y;
  const int N = g.nphi[r];
  const int logn = 31 - __clz(N);
  // Bit-reversed and strided accesses would put all lanes of a wave on one LDS bank: element a lives at a + (a >> sh)
  // (one empty slot per 2^sh values), which spreads them (the synthesis writes went 64-way conflicted without it).
  const int sh = max(5, logn - 6);
  auto ph = [&](int a2) { return a2 + (a2 >> sh); };
  const int Np = N + (N >> sh) + 1;                        // padded length of one map
  cplx* X = reinterpret_cast<cplx*>(fft_smem);            // [P][Np]
  cplx* TW = X + (size_t)P * Np;                           // [N / 2] (padded likewise): exp(+2 pi i k / N)
  const double phi0 = g.phi0[r], st = g.sth[r], ct = g.cth[r];
  const int pix0 = g.start[r];
  for (int k = tid; k < N / 2; k += TPB) {
    double s_, c_;
    sincospi(2.0 * (double)k / (double)N, &s_, &c_);
    TW[ph(k)] = make_double2(c_, s_);
  }
  // geometry of this thread's pixels j = tid + TPB q
  double nxv[NPT], nyv[NPT], hzv[NPT];
#pragma unroll
  for (int q = 0; q < NPT; ++q) {
    const int j = tid + TPB * q;
    double sp, cp;
    sincos(phi0 + 2.0 * kPi * (double)j / (double)N, &sp, &cp);  // as in bt_fused_dft_kernel: same map bits
    const double n0 = st * cp, n1 = st * sp, n2 = ct;
    hzv[q] = (j < N && (n0 * fr.z[0] + n1 * fr.z[1] + n2 * fr.z[2]) > 0.0) ? 1.0 : 0.0;
    nxv[q] = n0 * fr.x[0] + n1 * fr.x[1] + n2 * fr.x[2];
    nyv[q] = n0 * fr.y[0] + n1 * fr.y[1] + n2 * fr.y[2];
  }
  const int nm = 2 * cnt;
  const double w = ring_w ? ring_w[r] : 1.0;
  const int c_lo = blockIdx.x * cpw, c_hi = min(c_lo + cpw, ncol);
  for (int col = c_lo; col < c_hi; ++col) {
    const fdft_col cdc = cols[col];
    __syncthreads();   // the previous column's outputs have been read; the twiddles are there
    // ---- synthesis into bit-reversed positions
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int j = tid + TPB * q;
      if (j >= N) continue;
      const int jr = (int)(__brev((unsigned)j) >> (32 - logn));
      const size_t pix = (size_t)pix0 + j;
      const bool on = cdc.bi >= 0 && hzv[q] != 0.0;
      constexpr int BW = CB ? 2 : 1;
      const double* a = beams + (size_t)max(cdc.bi, 0) * bstride + BW * NCOMP * pix;
      const double* b = beams + (size_t)max(cdc.bj, 0) * bstride + BW * NCOMP * pix;
      double sf, cf;
      sincospi(2.0 * (cdc.u * nxv[q] + cdc.v * nyv[q]), &sf, &cf);
      const double pre = on ? cdc.pre : 0.0;
      const double tre = pre * cf, tim = pre * sf;
      if constexpr (CB) {
        double m_re[P], m_im[P];
        bt_synth_complex<P>(a, b, tre, tim, m_re, m_im);
        const int jp = ph(jr);
#pragma unroll
        for (int p = 0; p < P; ++p) X[(size_t)p * Np + jp] = make_double2(m_re[p], m_im[p]);
      } else if constexpr (P == 1) {
        const double bb = dm_ldg(a) * dm_ldg(b);
        X[ph(jr)] = make_double2(tre * bb, tim * bb);
      } else {
        const double a0 = dm_ldg(a), a1 = dm_ldg(a, 1), b0 = dm_ldg(b), b1 = dm_ldg(b, 1);
        const double sI = a0 * b0 + a1 * b1, sQ = a0 * b0 - a1 * b1, sU = a0 * b1 + a1 * b0, sV = a0 * b1 - a1 * b0;
        const int jp = ph(jr);
        X[jp] = make_double2(tre * sI, tim * sI);
        X[(size_t)Np + jp] = make_double2(tre * sQ, tim * sQ);
        X[2 * (size_t)Np + jp] = make_double2(tre * sU, tim * sU);
        X[3 * (size_t)Np + jp] = make_double2(-tim * sV, tre * sV);  // 1j * fringe * sV
      }
    }
    __syncthreads();
    // ---- stages 1..3: eight consecutive values per unit, in registers
    for (int U = tid; U < P * N / 8; U += TPB) {
      const int p = U / (N / 8), u = U - p * (N / 8);
      cplx* x = X + (size_t)p * Np + ph(u * 8);   // eight values never straddle a padding slot (2^sh >= 32)
      cplx e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = x[i];
#pragma unroll
      for (int i = 0; i < 8; i += 2) { const cplx t = e[i + 1]; e[i + 1] = csub(e[i], t); e[i] = cadd(e[i], t); }
#pragma unroll
      for (int i = 0; i < 8; i += 4) {
        cplx t = e[i + 2]; e[i + 2] = csub(e[i], t); e[i] = cadd(e[i], t);
        t = make_double2(-e[i + 3].y, e[i + 3].x);   // times exp(2 pi i / 4) = +i
        e[i + 3] = csub(e[i + 1], t); e[i + 1] = cadd(e[i + 1], t);
      }
      {
        const double h = 0.70710678118654752440;
        cplx t = e[4]; e[4] = csub(e[0], t); e[0] = cadd(e[0], t);
        t = make_double2(h * (e[5].x - e[5].y), h * (e[5].x + e[5].y));        // times exp(i pi / 4)
        e[5] = csub(e[1], t); e[1] = cadd(e[1], t);
        t = make_double2(-e[6].y, e[6].x);                                       // times i
        e[6] = csub(e[2], t); e[2] = cadd(e[2], t);
        t = make_double2(-h * (e[7].x + e[7].y), h * (e[7].x - e[7].y));       // times exp(3 i pi / 4)
        e[7] = csub(e[3], t); e[3] = cadd(e[3], t);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = e[i];
    }
    __syncthreads();
    // ---- the remaining stages two at a time (h = half size of the first of the two), a single one at the end if odd
    int sdone = 3;
    for (; sdone + 2 <= logn; sdone += 2) {
      const int h = 1 << sdone;
      const int s1 = N / (2 * h), s2 = N / (4 * h);
      for (int U = tid; U < P * N / 4; U += TPB) {
        const int p = U / (N / 4), u = U - p * (N / 4);
        const int blk = u / h, pos = u - blk * h;
        cplx* x = X + (size_t)p * Np;
        const int a0 = blk * 4 * h + pos;
        const int i0 = ph(a0), i1 = ph(a0 + h), i2 = ph(a0 + 2 * h), i3 = ph(a0 + 3 * h);
        cplx e0 = x[i0], e1 = x[i1], e2 = x[i2], e3 = x[i3];
        const cplx w1 = TW[ph(pos * s1)], w2 = TW[ph(pos * s2)], w3 = TW[ph((pos + h) * s2)];
        cplx t = cmul(w1, e1); e1 = csub(e0, t); e0 = cadd(e0, t);
        t = cmul(w1, e3); e3 = csub(e2, t); e2 = cadd(e2, t);
        t = cmul(w2, e2); e2 = csub(e0, t); e0 = cadd(e0, t);
        t = cmul(w3, e3); e3 = csub(e1, t); e1 = cadd(e1, t);
        x[i0] = e0; x[i1] = e1; x[i2] = e2; x[i3] = e3;
      }
      __syncthreads();
    }
    if (sdone < logn) {
      const int h = N / 2;
      for (int U = tid; U < P * N / 2; U += TPB) {
        const int p = U / h, u = U - p * h;
        cplx* x = X + (size_t)p * Np;
        const int i0 = ph(u), i1 = ph(u + h);
        const cplx t = cmul(TW[ph(u)], x[i1]);
        const cplx e0 = x[i0];
        x[i0] = cadd(e0, t);
        x[i1] = csub(e0, t);
      }
      __syncthreads();
    }
    // ---- the wanted rows: G[mm][ring][p ncol + col] = w exp(i m phi_0) X_p[m mod N]
    for (int mm = tid; mm < nm; mm += TPB) {
      const int m = mm < cnt ? m_lo + mm : -(m_lo + mm - cnt);
      const int idx = ((m % N) + N) % N;
      double s_, c_;
      sincos((double)m * phi0, &s_, &c_);
      const cplx phs = make_double2(w * c_, w * s_);
      cplx* out = G + ((size_t)mm * g.nring + r) * ncp + (size_t)col;
#pragma unroll
      for (int p = 0; p < P; ++p) dm_stg(out, (size_t)p * (ncp / P), cmul(phs, X[(size_t)p * Np + ph(idx)]));
    }
  }
}

// ---- north / south fold of the ring transform --------------------------------------------------------------
// HEALPix rings come in mirror pairs (r, nring - 1 - r) with z -> -z, and lambda_lm, W_lm are even / odd under z -> -z
// with l + m (X_lm the other way round): the sum over rings of the Legendre stage needs only the northern half once the
// pairs are combined — half the flops of that stage.  In place:  G[r] <- G[r] + G[r'],  G[r'] <- G[r'] - G[r]  (r < mid);
// symmetric functions then run over rings 0 .. mid, antisymmetric ones over mid + 1 .. nring - 1 (where the table value
// f(z_r') = -f(z_r) restores the sign).
__global__ __launch_bounds__(256) void bt_fold_kernel(cplx* __restrict__ G, int nm, int nring, size_t ncp) {
  const int mid = nring / 2;                       // nring = 4 nside - 1 is odd: ring `mid` is the equator
  const size_t per = (size_t)mid * ncp;            // elements of the northern rows of one m
  const size_t tot = (size_t)nm * per;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * 256) {
    const size_t mm = idx / per, rem = idx - mm * per;
    const size_t r = rem / ncp, c = rem - r * ncp;
    cplx* gn = G + (mm * nring + r) * ncp + c;
    cplx* gs = G + (mm * nring + (nring - 1 - r)) * ncp + c;
    const cplx a = *gn, b = *gs;
    *gn = cadd(a, b);
    *gs = csub(b, a);
  }
}

// tw[pix][mm] laid out per ring as (2*mmax+1) x nphi row-major: tw[off_r + mm*nphi + j] = exp(i (mm - mmax) phi_j)
__global__ void bt_twiddle_kernel(ring_geo g, int m_lo, int cnt, const size_t* __restrict__ toff, cplx* __restrict__ tw) {
  // rows [0, cnt): m = +m_lo .. +(m_lo + cnt - 1);  rows [cnt, 2 cnt): the same with a minus sign
  const int r = blockIdx.y;
  const int nphi = g.nphi[r];
  const int nm = 2 * cnt;
  const size_t tot = (size_t)nm * nphi;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
    const int mm = (int)(idx / nphi), j = (int)(idx % nphi);
    const int m = mm < cnt ? m_lo + mm : -(m_lo + mm - cnt);
    // reduce the argument exactly: m*j mod nphi keeps the phase in [0, 2 pi)
    const long long mj = ((long long)m * j) % nphi;
    const double ph = (double)m * g.phi0[r] + 2.0 * kPi * (double)mj / (double)nphi;
    double s, c;
    sincos(ph, &s, &c);
    tw[toff[r] + idx] = make_double2(c, s);
  }
}

// Legendre tables: lam[loff[m] + (l-m)*nring + r] = w * lambda_lm(theta_r), same for W and X (polarised)
__global__ void bt_legendre_kernel(ring_geo g, int lmax, int m_lo, int mmax, double w, const size_t* __restrict__ loff_,
                                   double* __restrict__ lam, double* __restrict__ Wt, double* __restrict__ Xt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = m_lo + blockIdx.y;
  const size_t* loff = loff_ - m_lo;  // tables are stored for m_lo .. mmax
  if (r >= g.nring || m > mmax || m > lmax) return;
  const double z = g.cth[r], st = g.sth[r];
  const double s2 = st * st;
  double logpre = 0.5 * (log(2.0 * m + 1.0) - log(4.0 * kPi));
  for (int k = 1; k <= m; ++k) logpre += 0.5 * log((2.0 * k - 1.0) / (2.0 * k));
  double lmm = (m > 0) ? exp(logpre + (double)m * log(st)) : exp(logpre);
  if (m & 1) lmm = -lmm;
  double* out = lam + loff[m];
  const size_t nr = g.nring;
  double pm2 = 0.0, pm1 = lmm;  // lambda_{l-2}, lambda_{l-1} as l advances
  out[r] = w * lmm;
  if (Wt) {
    double* wo = Wt + loff[m];
    double* xo = Xt + loff[m];
    // l = m term (needs lambda_{m-1,m} = 0)
    if (m >= 2) {
      const double l = m;
      const double nl = 2.0 * sqrt(1.0 / ((l - 1.0) * l * (l + 1.0) * (l + 2.0)));
      wo[r] = -w * nl * (-((l - l * l) / s2 + 0.5 * l * (l - 1.0)) * lmm);
      xo[r] = w * nl * (l / s2) * ((l - 1.0) * z * lmm);
    } else {
      wo[r] = 0.0;
      xo[r] = 0.0;
    }
  }
  for (int l = m + 1; l <= lmax; ++l) {
    double cur;
    if (l == m + 1) {
      cur = sqrt(2.0 * m + 3.0) * z * pm1;
    } else {
      const double a = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
      const double b = sqrt(((l - 1.0) * (l - 1.0) - (double)m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
      cur = a * (z * pm1 - b * pm2);
    }
    out[(size_t)(l - m) * nr + r] = w * cur;
    if (Wt) {
      double wv = 0.0, xv = 0.0;
      if (l >= 2) {
        const double dl = l, dm = m;
        const double nl = 2.0 * sqrt(1.0 / ((dl - 1.0) * dl * (dl + 1.0) * (dl + 2.0)));
        const double c = sqrt((2.0 * dl + 1.0) / (2.0 * dl - 1.0) * (dl * dl - dm * dm));
        wv = -nl * (-((dl - dm * dm) / s2 + 0.5 * dl * (dl - 1.0)) * cur + c * z / s2 * pm1);
        xv = nl * (dm / s2) * ((dl - 1.0) * z * cur - c * pm1);
      }
      (Wt + loff[m])[(size_t)(l - m) * nr + r] = w * wv;
      (Xt + loff[m])[(size_t)(l - m) * nr + r] = w * xv;
    }
    pm2 = pm1;
    pm1 = cur;
  }
}

// zero l > lmax_col for the rows written by this group
__global__ void bt_mask_kernel(cplx* __restrict__ beam_m, int F, int B, int P, int L, int mmax, int ncol,
                               const int* __restrict__ colf, const int* __restrict__ colb,
                               const int* __restrict__ collmax) {
  const int col = blockIdx.y;
  const int m = blockIdx.z;
  const int lm = collmax[col];
  const int f = colf[col], b = colb[col];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (s, p, l)
  const int tot = 2 * P * L;
  if (idx >= tot) return;
  const int l = idx % L, p = (idx / L) % P, s = idx / (L * P);
  if (l <= lm) return;
  beam_m[((((size_t)m * F + f) * 2 + s) * B + b) * P * L + (size_t)p * L + l] = make_double2(0.0, 0.0);
}

// rows of the private coefficient buffer of the refinement path -> the caller's beam_m blocks
// src (msrc, 1, 2, ncol, P, L), dst (m_hi - m_lo + 1, F, 2, B, P, L); blocks beyond msrc - 1 are zero
__global__ void bt_scatter_kernel(const cplx* __restrict__ src, int msrc, cplx* __restrict__ dst, int m_lo, int F, int B, int P,
                                  int L, int ncol, const int* __restrict__ colf, const int* __restrict__ colb) {
  const int col = blockIdx.y;
  const int mo = blockIdx.z;  // output block index, m = m_lo + mo
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (s, p, l)
  if (idx >= 2 * P * L) return;
  const int s = idx / (P * L), pl = idx % (P * L);
  const int m = m_lo + mo;
  cplx v = make_double2(0.0, 0.0);
  if (m < msrc) v = src[(((size_t)m * 2 + s) * ncol + col) * P * L + pl];
  dst[((((size_t)mo * F + colf[col]) * 2 + s) * B + colb[col]) * P * L + pl] = v;
}


// ---- harmonic-space Jacobi refinement (healpy.map2alm `iter`) ------------------------------------------------------
// healpy refines the quadrature as  a <- a + A(map - S a)  (A: map2alm with iter = 0, S: alm2map), i.e.
//   a_{k+1} = a_0 + a_k - (A o S) a_k,   a_0 = A(map).
// A o S never needs the map: the synthesis on ring r is F_m[r] = sum_l lambda_lm(z_r) a_lm and the ring DFT of the
// synthesised ring is  G_m'[r] = N_r sum_{m = m' (mod N_r)} e^{i (m' - m) phi0_r} F_m[r].  On a ring with N_r > 2 lmax only
// m = m' survives and A o S is, per m, the real (L - m) x (L - m) Gram matrix of the ring functions under the quadrature,
//   K_m[l][l'] = sum_r w_r N_r lambda_lm(z_r) lambda_l'm(z_r)
// (for the spin-2 pair the Hermitian block [[K_P, -i K_X], [i K_X, K_P]], K_P = W D W^T + X D X^T, K_X = W D X^T + X D W^T)
// — the same for every column, so an iteration is ONE real-B grouped product per (m, Stokes term) against K_m instead of
// a synthesis and an analysis over all rings.  The polar rings with N_r = 4 i <= 2 lmax alias m' with m' - 4 i k
// (e^{i k N phi0} = (-1)^k there); lambda_lm(z_i) falls off super-exponentially once m exceeds l sin(theta_i), so only the
// rings next to the poles couple anything above rounding: ring i is treated as aliasing when some m >= 2 i still has
// max_l |lambda_lm(z_i)| >= 1e-13 (mlim(i) = the last such m; the terms left out are below 1e-15 of the coefficients), which bounds both the ring set (a few dozen per cap) and the
// m involved (m <= mcut, around 50 - 160).  Those terms are formed explicitly: synthesis on the alias rings, the fold
// over k, analysis.  Everything is carried in the beam_m convention b0 = a_{l,+m}, b1 = (-1)^m conj(a_{l,-m}), in which
// analysis and synthesis use the same real tables for both slots; F_{-m} = conj(h1_m).
struct bt_alias_info {
  int ia = 0;              // rings 0 .. ia - 1 of the north cap (and their mirrors) carry alias terms
  int mcut = -1;           // largest m coupled to another m by them
  std::vector<int> mlim;   // per north alias ring: m above this are negligible on it
};

// max_l |lambda_lm(z)| (and |W|, |X|) over l = m .. lmax, unweighted; the recurrences of bt_legendre_kernel
static double bt_table_peak(int lmax, int m, double z, double st, bool pol) {
  const double s2 = st * st;
  double logpre = 0.5 * (std::log(2.0 * m + 1.0) - std::log(4.0 * kPi));
  for (int k = 1; k <= m; ++k) logpre += 0.5 * std::log((2.0 * k - 1.0) / (2.0 * k));
  double lmm = (m > 0) ? std::exp(logpre + (double)m * std::log(st)) : std::exp(logpre);
  double peak = std::fabs(lmm);
  if (pol && m >= 2) {
    const double l = m;
    const double nl = 2.0 * std::sqrt(1.0 / ((l - 1.0) * l * (l + 1.0) * (l + 2.0)));
    peak = std::max(peak, std::fabs(nl * (-((l - l * l) / s2 + 0.5 * l * (l - 1.0)) * lmm)));
    peak = std::max(peak, std::fabs(nl * (l / s2) * ((l - 1.0) * z * lmm)));
  }
  double pm2 = 0.0, pm1 = lmm;
  for (int l = m + 1; l <= lmax; ++l) {
    double cur;
    if (l == m + 1) {
      cur = std::sqrt(2.0 * m + 3.0) * z * pm1;
    } else {
      const double a = std::sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
      const double b = std::sqrt(((l - 1.0) * (l - 1.0) - (double)m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
      cur = a * (z * pm1 - b * pm2);
    }
    peak = std::max(peak, std::fabs(cur));
    if (pol && l >= 2) {
      const double dl = l, dm = m;
      const double nl = 2.0 * std::sqrt(1.0 / ((dl - 1.0) * dl * (dl + 1.0) * (dl + 2.0)));
      const double c = std::sqrt((2.0 * dl + 1.0) / (2.0 * dl - 1.0) * (dl * dl - dm * dm));
      peak = std::max(peak, std::fabs(nl * (-((dl - dm * dm) / s2 + 0.5 * dl * (dl - 1.0)) * cur + c * z / s2 * pm1)));
      peak = std::max(peak, std::fabs(nl * (dm / s2) * ((dl - 1.0) * z * cur - c * pm1)));
    }
    pm2 = pm1;
    pm1 = cur;
  }
  return peak;
}

constexpr double kAliasEps = 1e-13;

// Alias rings and their m limits for (nside, lmax, polarised); a function of these three alone (every rank, every column
// chunk and every m-range sees the same sets, so the refined blocks do not depend on how m is partitioned).  Cached.
static const bt_alias_info& bt_alias_lookup(int nside, int lmax, bool pol, const double* cth, const double* sth) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, bool>, bt_alias_info> cache;
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_tuple(nside, lmax, pol);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  bt_alias_info info;
  int quiet = 0;
  for (int i = 1; i < nside && 4 * i <= 2 * lmax; ++i) {
    // beyond m = 2 i every l <= lmax is past its turning point on this ring: the tables fall monotonically with m
    const double z = cth[i - 1], st = sth[i - 1];
    int ml = -1, below = 0;
    for (int m = 2 * i; m <= lmax && below < 3; ++m) {
      if (bt_table_peak(lmax, m, z, st, pol) >= kAliasEps) { ml = m; below = 0; } else ++below;
    }
    if (ml >= 2 * i) {
      info.mlim.resize(i, -1);
      info.mlim[i - 1] = ml;
      info.ia = i;
      info.mcut = std::max(info.mcut, ml);
      quiet = 0;
    } else if (++quiet >= 8) {
      break;
    }
  }
  info.mlim.resize(info.ia, -1);
  return cache.emplace(key, std::move(info)).first->second;
}

// ---- rings that carry nothing at high m ------------------------------------------------------------------------------
// lambda_lm(z_r) (and the spin-2 W, X) fall off super-exponentially once m exceeds l sin(theta_r): on the rings next to the
// poles every l <= lmax is past its turning point from m = lmax sin(theta) on, and from some m on the whole table column
// of the ring is below rounding.  mskip[r] = the first m from which max_l |table| < kRingSkipEps for good: the ring
// transform G_m[r] of such (m, ring) pairs multiplies table entries that cannot move a coefficient by a tenth of an ulp
// (the sum over at most 4 nside rings of 1e-18 x the scale of the kept terms), so it is neither computed nor stored as
// anything but zero — libsharp's `mlim` rule of healpy.map2alm, here with the bound computed from the tables themselves.
// The decision is per (m, ring), a function of (nside, lmax, polarised) alone: any partition of m over ranks and any
// pass structure of the kernels gives the same bits.  A rank that owns m = 315 .. 512 of configs[2] (nside 256) skips
// 3/4 of the cap work of its ring transform.
constexpr double kRingSkipEps = 1e-18;

static const std::vector<int>& bt_ring_skip_lookup(int nside, int lmax, bool pol, const double* cth, const double* sth) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, bool>, std::vector<int>> cache;
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_tuple(nside, lmax, pol);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const int nring = 4 * nside - 1;
  std::vector<int> ms(nring, lmax + 1);
  for (int r = 0; r <= nring / 2; ++r) {
    const double z = cth[r], st = sth[r];
    int lo = std::min(lmax + 1, (int)std::ceil((double)lmax * st) + 1);   // candidates: lo .. lmax + 1 (= never skipped)
    int hi = lmax + 1;
    if (lo <= lmax && bt_table_peak(lmax, lmax, z, st, pol) < kRingSkipEps) {
      // beyond the turning point the peak falls monotonically with m: bisect for the first m below the threshold
      int a = lo, b = lmax;   // invariant: peak(b) < eps
      while (a < b) {
        const int mid = (a + b) / 2;
        if (bt_table_peak(lmax, mid, z, st, pol) < kRingSkipEps) b = mid; else a = mid + 1;
      }
      hi = b;
      // (guard against a non-monotonic stretch right at the boundary: step up while any of the next few m is above)
      for (int m = hi; m <= std::min(lmax, hi + 3); ++m)
        if (bt_table_peak(lmax, m, z, st, pol) >= kRingSkipEps) hi = m + 1;
    }
    ms[r] = hi;
    ms[nring - 1 - r] = hi;
  }
  return cache.emplace(key, std::move(ms)).first->second;
}

// out[idx] = in[idx] * sc[idx % nring]   (tables scaled by the per-ring factor of A o S)
__global__ __launch_bounds__(256) void bt_scale_table_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n,
                                                             int nring, const double* __restrict__ sc) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256)
    out[idx] = in[idx] * sc[idx % (size_t)nring];
}

// Alias fold on the alias rings.  Ha is (2 (Mc + 1), ncp, nra), block mm = s (Mc + 1) + m, holding h0_m = F_m and
// h1_m = conj(F_-m).  g[s' = 0][m'] = sc sum_{k != 0} (-1)^k F_{m' - k N},  g[s' = 1][m'] = sc sum_{k != 0} (-1)^k conj(F_{-m' - k N}),
// over the |m' - k N| <= mlim of the ring (sc = ring weight x N), written into the alias slots of the increment rows:
// d[(m', s', col, p)][Lg + ja] — the K dimension of the next product runs over the coefficients AND these slots.
__global__ __launch_bounds__(256) void bt_alias_fold_kernel(const cplx* __restrict__ Ha, cplx* __restrict__ d, int Mc, int nra,
                                                            size_t ncp, int L, int Lg, const int* __restrict__ nphiA,
                                                            const int* __restrict__ mlimA, const double* __restrict__ scA) {
  const int ja = blockIdx.y * 64 + threadIdx.x, mmp = blockIdx.z;
  const size_t c = (size_t)blockIdx.x * 4 + threadIdx.y;
  if (ja >= nra || c >= ncp) return;
  const int sp = mmp / (Mc + 1), mp = mmp - sp * (Mc + 1);
  const int N = nphiA[ja], ml = min(mlimA[ja], Mc);
  double are = 0.0, aim = 0.0;
  if (mp <= ml && !(sp == 1 && mp == 0)) {
    const int base = sp == 0 ? mp : -mp;
    // mu = base - k N in [-ml, ml]
    int klo = base - ml, khi = base + ml;
    klo = klo >= 0 ? (klo + N - 1) / N : -((-klo) / N);
    khi = khi >= 0 ? khi / N : -((-khi + N - 1) / N);
    for (int k = klo; k <= khi; ++k) {
      if (k == 0) continue;
      const int mu = base - k * N;
      cplx f;
      if (mu >= 0) f = Ha[((size_t)mu * ncp + c) * nra + ja];
      else { f = Ha[((size_t)(Mc + 1 - mu) * ncp + c) * nra + ja]; f.y = -f.y; }
      if (sp == 1) f.y = -f.y;
      if (k & 1) { are -= f.x; aim -= f.y; } else { are += f.x; aim += f.y; }
    }
  }
  const double sc = scA[ja];
  d[(((size_t)mp * 2 + sp) * ncp + c) * L + Lg + ja] = make_double2(sc * are, sc * aim);
}

// The alias rows of the extended Gram matrices: element (k = Lm + ja, n = l) of the (Lm + nra) x Lm operand of block m is
// the alias-ring table value tabA_m[l][ja] (the analysis of the folded rings rides in the K dimension of the product)
__global__ __launch_bounds__(256) void bt_kext_fill_kernel(const double* __restrict__ tabA, const size_t* __restrict__ loffA,
                                                           double* __restrict__ Kx, const size_t* __restrict__ kxoff, int lmax_grp,
                                                           int nra, int kfast) {
  const int m = blockIdx.y;
  const int Lm = lmax_grp + 1 - m, Kd = Lm + nra;
  const size_t tot = (size_t)Lm * nra;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * 256) {
    const size_t l = idx / nra, ja = idx - l * nra;
    const double v = tabA[loffA[m] + idx];
    if (kfast) Kx[kxoff[m] + l * Kd + Lm + ja] = v;
    else Kx[kxoff[m] + (Lm + ja) * Lm + l] = v;
  }
}

// d <- mask(d - t), acc += d on the private coefficient buffers (nm, 2, ncol, P, L) (L = Lg coefficients + alias slots);
// entries l < m stay zero, the -m slot of m = 0 stays zero, l > lmax of the column is cut (the column's own band limit,
// telescope.py:792-802)
__global__ __launch_bounds__(256) void bt_refine_update_kernel(cplx* __restrict__ acc, cplx* __restrict__ d, const cplx* __restrict__ t,
                                                               int m_lo, int ncol, int P, int L, int Lg,
                                                               const int* __restrict__ collmax) {
  const int col = blockIdx.y;
  const int mi = blockIdx.z >> 1, s = blockIdx.z & 1;
  const int m = m_lo + mi;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // over (p, l)
  if (idx >= P * L) return;
  const int l = idx % L;
  if (l < m || l >= Lg || (m == 0 && s == 1)) return;
  const size_t o = ((((size_t)mi * 2 + s) * ncol + col) * P) * L + idx;
  cplx dn = make_double2(0.0, 0.0);
  if (l <= collmax[col]) dn = csub(d[o], t[o]);
  d[o] = dn;
  acc[o] = cadd(acc[o], dn);
}

// private coefficient buffer (blocks m = src_mlo .. src_mlo + src_nm - 1, layout (src_nm, 2, ncol, P, Lrow) with Ls
// coefficients per row) -> the caller's beam_m blocks (m_hi - m_lo + 1, F, 2, B, P, L): l >= Ls and blocks outside the
// source are zero
__global__ void bt_scatter2_kernel(const cplx* __restrict__ src, int src_mlo, int src_nm, int Ls, int Lrow, cplx* __restrict__ dst,
                                   int m_lo, int F, int B, int P, int L, int ncol, const int* __restrict__ colf,
                                   const int* __restrict__ colb) {
  const int col = blockIdx.y;
  const int mo = blockIdx.z;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over (s, p, l)
  if (idx >= 2 * P * L) return;
  const int s = idx / (P * L), pl = idx % (P * L), p = pl / L, l = pl % L;
  const int ms = m_lo + mo - src_mlo;
  cplx v = make_double2(0.0, 0.0);
  if (ms >= 0 && ms < src_nm && l < Ls) v = src[((((size_t)ms * 2 + s) * ncol + col) * P + p) * Lrow + l];
  dst[((((size_t)mo * F + colf[col]) * 2 + s) * B + colb[col]) * P * L + pl] = v;
}

struct geo_host {
  ring_geo g;
  std::vector<double> cth, sth, phi0;
  std::vector<int> nphi, start;
};

int upload_geo(dm_ctx* ctx, int nside, const double* cth, const double* sth, geo_host& gh) {
  const int nring = 4 * nside - 1;
  gh.cth.assign(cth, cth + nring);
  gh.sth.assign(sth, sth + nring);
  gh.phi0.resize(nring);
  gh.nphi.resize(nring);
  gh.start.resize(nring);
  int acc = 0;
  for (int r = 0; r < nring; ++r) {
    const int i = r + 1;
    int np_;
    double p0;
    if (i < nside) { np_ = 4 * i; p0 = kPi / (4.0 * i); }
    else if (i <= 3 * nside) { np_ = 4 * nside; p0 = (((i + nside) & 1) == 0) ? kPi / (4.0 * nside) : 0.0; }
    else { const int j = 4 * nside - i; np_ = 4 * j; p0 = kPi / (4.0 * j); }
    gh.nphi[r] = np_;
    gh.phi0[r] = p0;
    gh.start[r] = acc;
    acc += np_;
  }
  gh.g.nring = nring;
  gh.g.npix = acc;
  // one staged copy for the five arrays (every descriptor upload is a copy kernel of its own in the stream)
  std::vector<double> blob(4 * (size_t)nring);
  std::memcpy(blob.data(), gh.cth.data(), sizeof(double) * nring);
  std::memcpy(blob.data() + nring, gh.sth.data(), sizeof(double) * nring);
  std::memcpy(blob.data() + 2 * (size_t)nring, gh.phi0.data(), sizeof(double) * nring);
  int* ib = reinterpret_cast<int*>(blob.data() + 3 * (size_t)nring);
  std::memcpy(ib, gh.nphi.data(), sizeof(int) * nring);
  std::memcpy(ib + nring, gh.start.data(), sizeof(int) * nring);
  double* d = dm_ws_upload(ctx, blob);
  if (!d) return DM_ENOMEM;
  gh.g.cth = d;
  gh.g.sth = d + nring;
  gh.g.phi0 = d + 2 * (size_t)nring;
  gh.g.nphi = reinterpret_cast<const int*>(d + 3 * (size_t)nring);
  gh.g.start = gh.g.nphi + nring;
  return DM_OK;
}

frame3 make_frame(const double* xhat, const double* yhat, const double* zhat) {
  frame3 f;
  for (int i = 0; i < 3; ++i) { f.x[i] = xhat[i]; f.y[i] = yhat[i]; f.z[i] = zhat[i]; }
  return f;
}

}  // namespace

extern "C" {

// Cylinder field pattern on the pixel centres of a HEALPix map.
//   ring_cth/ring_sth (4 nside - 1) host: cos/sin of the ring colatitudes (host geometry)
//   frame_host (9): xhat (East), yhat (North), zhat (zenith) in sky cartesian coordinates
//   kind 0: amplitude only (npix doubles out); 1 / 2: X / Y dipole (npix x 2 doubles out)
//   tab_* (ntab) host: knots of the E-W Fraunhofer pattern spline (x, y, y'')
//   fwhm_ns: FWHM of the exptan N-S pattern
int dm_bt_beam_cyl(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                   const double* frame_host, int kind, const double* tab_x_host, const double* tab_y_host,
                   const double* tab_y2_host, int ntab, double fwhm_ns, double* out_dev) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && frame_host && kind >= 0 && kind <= 2 && tab_x_host &&
                  tab_y_host && tab_y2_host && ntab >= 2 && out_dev);
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  geo_host gh;
  DM_TRY(upload_geo(ctx, nside, ring_cth_host, ring_sth_host, gh));
  std::vector<double> tx(tab_x_host, tab_x_host + ntab), ty(tab_y_host, tab_y_host + ntab),
      ty2(tab_y2_host, tab_y2_host + ntab);
  double* dx = dm_ws_upload(ctx, tx);
  double* dy = dm_ws_upload(ctx, ty);
  double* dy2 = dm_ws_upload(ctx, ty2);
  if (!dx || !dy || !dy2) return DM_ENOMEM;
  const double th = tan(fwhm_ns / 2.0);
  const double alpha = log(2.0) / (2.0 * th * th);
  frame3 fr = make_frame(frame_host, frame_host + 3, frame_host + 6);
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_beam_kernel, dim3((gh.g.npix + 255) / 256), dim3(256), 0, ctx->stream, gh.g, fr, kind, dx, dy,
                     dy2, ntab, alpha, out_dev);
  DM_HIP(ctx, hipGetLastError());
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// nbeam cylinder patterns in one call: geometry and spline tables go up in two staged copies instead of eight per beam.
//   kind_host, fwhm_ns_host (nbeam); tab_off_host (nbeam + 1): beam b uses knots [tab_off[b], tab_off[b + 1]) of the
//   concatenated tab_x / tab_y / tab_y2 arrays; out_dev row b starts at out_dev + b * out_stride (doubles)
int dm_bt_beams_cyl(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                    const double* frame_host, int nbeam, const int* kind_host, const double* tab_x_host,
                    const double* tab_y_host, const double* tab_y2_host, const int* tab_off_host,
                    const double* fwhm_ns_host, double* out_dev, size_t out_stride) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && frame_host && nbeam >= 0 && kind_host && tab_x_host &&
                  tab_y_host && tab_y2_host && tab_off_host && fwhm_ns_host && out_dev);
  if (nbeam == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  geo_host gh;
  DM_TRY(upload_geo(ctx, nside, ring_cth_host, ring_sth_host, gh));
  const int ntot = tab_off_host[nbeam];
  for (int b = 0; b < nbeam; ++b) {
    DM_ARG(ctx, kind_host[b] >= 0 && kind_host[b] <= 2 && tab_off_host[b + 1] - tab_off_host[b] >= 2);
    DM_ARG(ctx, out_stride >= (size_t)gh.g.npix * (kind_host[b] == 0 ? 1 : 2));
  }
  std::vector<double> tabs(3 * (size_t)ntot);
  std::memcpy(tabs.data(), tab_x_host, sizeof(double) * ntot);
  std::memcpy(tabs.data() + ntot, tab_y_host, sizeof(double) * ntot);
  std::memcpy(tabs.data() + 2 * (size_t)ntot, tab_y2_host, sizeof(double) * ntot);
  double* dt = dm_ws_upload(ctx, tabs);
  if (!dt) return DM_ENOMEM;
  frame3 fr = make_frame(frame_host, frame_host + 3, frame_host + 6);
  std::vector<bt_beam_desc> bd(nbeam);
  for (int b = 0; b < nbeam; ++b) {
    const int o = tab_off_host[b], nt = tab_off_host[b + 1] - o;
    const double th = tan(fwhm_ns_host[b] / 2.0);
    const double alpha = log(2.0) / (2.0 * th * th);
    bd[b] = bt_beam_desc{kind_host[b], nt, dt + o, dt + ntot + o, dt + 2 * (size_t)ntot + o, alpha, out_dev + (size_t)b * out_stride};
  }
  bt_beam_desc* d_bd = dm_ws_upload(ctx, bd);
  if (!d_bd) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_beams_kernel, dim3((gh.g.npix + 255) / 256, nbeam), dim3(256), 0, ctx->stream, gh.g, fr, d_bd);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

// Visibility response maps of `ncol` (frequency, baseline) columns.
//   beams_dev: nbeam maps of npix*ncomp doubles (ncomp = 2 if polarised else 1), as written by dm_bt_beam_cyl
//   uv_host (ncol x 2): baseline / wavelength;  bi_host, bj_host (ncol): beam index of the two feeds
//   maps_dev out: (ncol, P, npix) complex128, P = 4 (I,Q,U,V) if polarised else 1
int dm_bt_maps(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
               const double* frame_host, int polarised, int nbeam, const double* beams_dev, int ncol,
               const double* uv_host, const int* bi_host, const int* bj_host, void* maps_dev) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && frame_host && nbeam > 0 && beams_dev && ncol >= 0 &&
                  uv_host && bi_host && bj_host && maps_dev);
  if (ncol == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  geo_host gh;
  DM_TRY(upload_geo(ctx, nside, ring_cth_host, ring_sth_host, gh));
  const int ncomp = polarised ? 2 : 1;
  const size_t bstride = (size_t)gh.g.npix * ncomp;
  frame3 fr = make_frame(frame_host, frame_host + 3, frame_host + 6);
  double* omega = dm_ws_alloc_t<double>(ctx, nbeam);
  std::vector<double> uv(uv_host, uv_host + 2 * (size_t)ncol);
  std::vector<int> bi(bi_host, bi_host + ncol), bj(bj_host, bj_host + ncol);
  for (int c = 0; c < ncol; ++c) DM_ARG(ctx, bi[c] >= 0 && bi[c] < nbeam && bj[c] >= 0 && bj[c] < nbeam);
  double* duv = dm_ws_upload(ctx, uv);
  int* dbi = dm_ws_upload(ctx, bi);
  int* dbj = dm_ws_upload(ctx, bj);
  if (!omega || !duv || !dbi || !dbj) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_omega_kernel, dim3(nbeam), dim3(256), 0, ctx->stream, gh.g, fr, beams_dev, ncomp, bstride,
                     omega);
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_maps_kernel, dim3((gh.g.npix + 255) / 256, ncol), dim3(256), 0, ctx->stream, gh.g, fr,
                     polarised, ncol, duv, dbi, dbj, beams_dev, bstride, omega, reinterpret_cast<cplx*>(maps_dev));
  DM_HIP(ctx, hipGetLastError());
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// dm_bt_maps for COMPLEX field patterns: beams_dev holds nbeam maps of npix * ncomp complex128 (zero below the horizon).
int dm_bt_maps_c(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host,
                 const double* frame_host, int polarised, int nbeam, const void* beams_dev, int ncol,
                 const double* uv_host, const int* bi_host, const int* bj_host, void* maps_dev) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && frame_host && nbeam > 0 && beams_dev && ncol >= 0 &&
                  uv_host && bi_host && bj_host && maps_dev);
  if (ncol == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  geo_host gh;
  DM_TRY(upload_geo(ctx, nside, ring_cth_host, ring_sth_host, gh));
  const int ncomp = polarised ? 2 : 1;
  const size_t bstride = (size_t)gh.g.npix * ncomp;
  frame3 fr = make_frame(frame_host, frame_host + 3, frame_host + 6);
  double* omega = dm_ws_alloc_t<double>(ctx, nbeam);
  std::vector<double> uv(uv_host, uv_host + 2 * (size_t)ncol);
  std::vector<int> bi(bi_host, bi_host + ncol), bj(bj_host, bj_host + ncol);
  for (int c = 0; c < ncol; ++c) DM_ARG(ctx, bi[c] >= 0 && bi[c] < nbeam && bj[c] >= 0 && bj[c] < nbeam);
  double* duv = dm_ws_upload(ctx, uv);
  int* dbi = dm_ws_upload(ctx, bi);
  int* dbj = dm_ws_upload(ctx, bj);
  if (!omega || !duv || !dbi || !dbj) return DM_ENOMEM;
  const cplx* bc = reinterpret_cast<const cplx*>(beams_dev);
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_omega_c_kernel, dim3(nbeam), dim3(256), 0, ctx->stream, gh.g, bc, ncomp, bstride, omega);
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_maps_c_kernel, dim3((gh.g.npix + 255) / 256, ncol), dim3(256), 0, ctx->stream, gh.g, fr,
                     polarised, ncol, duv, dbi, dbj, bc, bstride, omega, reinterpret_cast<cplx*>(maps_dev));
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

// Spherical-harmonic transform of the maps of one nside group straight into the m-ordered
// beam_m blocks: beam_m_dev is (mmax+1, F, 2, B, P, L) complex128, L = lside + 1; the rows
// (f, :, b, :, :) of the group's columns are overwritten (zero for l < m and l > lmax_col).
//   lmax_grp: largest per-column lmax in the group (<= lside) — tables are built up to it
//   m_lo .. m_hi: only these m-blocks are produced; beam_m_dev is then (m_hi - m_lo + 1, F, 2, B, P, L)
//   (a rank that owns a range of m synthesises the maps but transforms and stores its own m only)
//   niter > 0: healpy-style Jacobi refinement of the quadrature (map2alm's `iter`): after the first analysis the
//   map is re-synthesised from the coefficients, the residual map analysed and added, niter times.  It needs every
//   (l, m) of a column, i.e. m_lo = 0 and m_hi >= lmax_grp (dm_bt_sht_opts arranges that), and a scratch copy of the maps.
//   ring_w_host (nring) or NULL: per-ring quadrature weights multiplying the equal-area weight 4 pi / npix
//   (healpy's `use_weights` ring weights are 1 + w_ring; the tables themselves are data files of healpy).
// optional inputs of the fused path: with `syn` given (and maps_dev == NULL) the ring DFT synthesises the map values itself
struct bt_synth_in {
  const double* frame_host;
  int nbeam;
  const double* beams_dev;
  const double* uv_host;
  const int* bi_host;
  const int* bj_host;
  int complex_beams;   // beams_dev holds complex patterns (interleaved re, im): _construct_pol_complex inside the kernels
};

static int bt_sht_impl(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                       int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                       const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev, int niter,
                       const double* ring_w_host, const bt_synth_in* syn = nullptr, bool harmonic = false, int row_pad = 0) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, maps_dev || (syn && (niter == 0 || harmonic)));
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && lside >= 0 && m_lo >= 0 && m_hi >= m_lo && lmax_grp >= 0 &&
                  lmax_grp <= lside && F > 0 && B > 0 && ncol >= 0 && col_f_host && col_b_host && col_lmax_host &&
                  beam_m_dev && niter >= 0);
  DM_ARG(ctx, niter == 0 || harmonic || (m_lo == 0 && m_hi >= lmax_grp));  // the residual map needs every m of a column
  // the harmonic-space refinement runs on a private coefficient buffer laid out (m, 2, col, P, L) (bt_sht_refined)
  DM_ARG(ctx, !(niter > 0 && harmonic) || (F == 1 && B == ncol && lside == lmax_grp));
  DM_ARG(ctx, row_pad == 0 || (niter > 0 && harmonic));
  if (ncol == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  geo_host gh;
  DM_TRY(upload_geo(ctx, nside, ring_cth_host, ring_sth_host, gh));
  const int P = polarised ? 4 : 1;
  const int L = lside + 1 + row_pad;   // row length of the destination (row_pad: alias slots of the private buffers)
  const int nring = gh.g.nring, npix = gh.g.npix;
  const int mtop = std::min(m_hi, lmax_grp);  // no (l, m) content above the group's band limit
  const int cnt = std::max(mtop - m_lo + 1, 0);  // m values with content in this range
  const int nm = 2 * std::max(cnt, 1);
  const int ncp = ncol * P;  // map columns
  const int nmblk = m_hi - m_lo + 1;

  // ---- twiddles and ring DFT: G[mm][ring][colp] — colp = col * P + p from materialised maps, p * ncol + col from the fused kernels
  const bool fused = maps_dev == nullptr;
  std::vector<size_t> toff(nring);
  size_t ttot = 0;
  for (int r = 0; r < nring; ++r) { toff[r] = ttot; ttot += (size_t)nm * gh.nphi[r]; }
  size_t* d_toff = dm_ws_upload(ctx, toff);
  cplx* tw = fused ? nullptr : dm_ws_alloc_t<cplx>(ctx, ttot);
  cplx* G = dm_ws_alloc_t<cplx>(ctx, (size_t)nm * nring * ncp);
  if (!d_toff || (!fused && !tw) || !G) return DM_ENOMEM;
  if (!fused)
    DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_twiddle_kernel, dim3(8, nring), dim3(256), 0, ctx->stream, gh.g, m_lo, std::max(cnt, 1), d_toff,
                       tw);
  if (fused && cnt > 0) {
    // beam solid angles, per-column constants, then synthesis + DFT in one kernel (no Stokes maps in HBM)
    const bool cbm = syn->complex_beams != 0;
    const int ncomp = (polarised ? 2 : 1) * (cbm ? 2 : 1);   // doubles per pixel of a beam (complex patterns: re, im)
    const size_t bstride = (size_t)npix * ncomp;
    frame3 fr = make_frame(syn->frame_host, syn->frame_host + 3, syn->frame_host + 6);
    double* omega = dm_ws_alloc_t<double>(ctx, syn->nbeam);
    double* opart = dm_ws_alloc_t<double>(ctx, (size_t)syn->nbeam * NOMEGA);
    if (!omega || !opart) return DM_ENOMEM;
    DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_omega_part_kernel, dim3(syn->nbeam, NOMEGA), dim3(256), 0, ctx->stream, gh.g, syn->beams_dev, ncomp,
                       bstride, opart);
    DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_omega_fin_kernel, dim3((syn->nbeam + 63) / 64), dim3(64), 0, ctx->stream, opart, syn->nbeam,
                       4.0 * kPi / (double)npix, omega);
    // (the solid angles stay on the device: the per-column factor 1 / sqrt(Omega_i Omega_j) is filled in by a small
    // kernel, so the host never waits inside the call)
    const int ncol16 = (ncol + 15) / 16;
    std::vector<fdft_col> fc((size_t)ncol16 * 16, fdft_col{0.0, 0.0, 0.0, -1, -1});
    for (int c = 0; c < ncol; ++c) {
      const int bi = syn->bi_host[c], bj = syn->bj_host[c];
      DM_ARG(ctx, bi >= 0 && bi < syn->nbeam && bj >= 0 && bj < syn->nbeam);
      fc[c] = fdft_col{syn->uv_host[2 * c], syn->uv_host[2 * c + 1], 0.0, bi, bj};
    }
    fdft_col* d_fc = dm_ws_upload(ctx, fc);
    if (d_fc) DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_fdft_pre_kernel, dim3((ncol + 255) / 256), dim3(256), 0, ctx->stream, d_fc, ncol, omega);
    double* d_rw = nullptr;
    if (ring_w_host) {
      std::vector<double> rw(ring_w_host, ring_w_host + nring);
      d_rw = dm_ws_upload(ctx, rw);
      if (!d_rw) return DM_ENOMEM;
    }
    if (!d_fc) return DM_ENOMEM;
    const int nmg = (nm + 3) / 4;  // groups of four m-values
    // The belt (rings nside - 1 .. 3 nside - 1, all with N = 4 nside pixels) goes by FFT when N is a power of two and the
    // P maps of a column fit the LDS; the matrix-form kernels below then see the rings of the two caps only.  The choice
    // depends on nside, P and on whether the call is narrow (below) — not on WHICH m it asks for.
    static const bool fft_off = getenv("DM_BT_FFT") && atoi(getenv("DM_BT_FFT")) == 0;
    const int N = 4 * nside;
    int fft_logn = 0;
    while ((1 << fft_logn) < N) ++fft_logn;
    const int fft_sh = std::max(5, fft_logn - 6);   // padding of the LDS arrays, as in the kernel
    const size_t fft_lds = sizeof(cplx) * ((size_t)P * (N + (N >> fft_sh) + 1) + (N / 2 + ((N / 2) >> fft_sh) + 1));
    // A NARROW call (at most DM_BT_NARROW = 8 m-values: one pass of the shared-synthesis kernel) keeps the matrix form on
    // the belt too: the FFT computes every m of a ring whether wanted or not — 14 CU-cycles per (pixel, column) at nside
    // 512 with four maps (150 KB of LDS: one workgroup per CU) against ~1 per group of four m-rows on the matrix pipe.
    // One m-block of configs[4] (59.6 GB: a rank holds one or two at a time) 24.8 -> 4.6 s of ring transform
    // (scratch/btgen_narrow_probe.py: 784 -> 146 ms on 8 of the 256 frequencies; 4 m-values 833 -> 229, 8: 1039 -> 555).
    // The two forms round differently (2e-13 of the largest coefficient): blocks are the same BITS for any partition
    // of m whose calls are all wide, and equal to rounding when narrow calls are involved.
    static const int narrow_max = getenv("DM_BT_NARROW") ? atoi(getenv("DM_BT_NARROW")) : 8;
    const bool narrow = cnt <= narrow_max;
    const bool use_fft = !fft_off && !narrow && nside >= 2 && (N & (N - 1)) == 0 && N <= 4096 && fft_lds <= 160u * 1024u - 256u;
    int nring_dft = nring;
    const int* d_caps = nullptr;
    if (use_fft) {
      std::vector<int> caps;
      for (int r = 0; r < nring; ++r)
        if (r < nside - 1 || r > 3 * nside - 1) caps.push_back(r);
      nring_dft = (int)caps.size();
      if (nring_dft > 0) {
        d_caps = dm_ws_upload(ctx, caps);
        if (!d_caps) return DM_ENOMEM;
      }
      const int nring_eq = 2 * nside + 1;
      const int cpw = std::max(4, std::min(32, (int)(((size_t)ncol * nring_eq) / 4096)));
      const dim3 grid((unsigned)((ncol + cpw - 1) / cpw), (unsigned)nring_eq);
      static const int wide_env = getenv("DM_BT_FFT_WIDE") ? atoi(getenv("DM_BT_FFT_WIDE")) : -1;
      // (measured at nside 512 with four maps: 256 threads 3.6 s per rank call, 1024 threads 6.4 s — the barriers of
      // sixteen waves cost more than their latency hiding brings; the wide variant stays behind DM_BT_FFT_WIDE=1)
      const bool wide = wide_env > 0 && (polarised ? N >= 1024 : N == 4096);
      auto fft_launch = [&](auto kern, int tpb) -> int {
        DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)fft_lds));
        DM_PLAUNCH(ctx, DM_PROF_BT_RING, kern, grid, dim3(tpb), fft_lds, ctx->stream, gh.g, fr, syn->beams_dev, bstride, d_fc, ncol, m_lo,
                           cnt, d_rw, G, ncp, nside - 1, cpw);
        return DM_OK;
      };
      if (wide) {
        const int npt = std::max(1, N / 1024);
        if (polarised) {
          if (npt == 1) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 1, 1024, true> : bt_fused_fft_kernel<4, 1, 1024>), 1024));
          else DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 2, 1024, true> : bt_fused_fft_kernel<4, 2, 1024>), 1024));   // N = 2048 (4096 x 4 maps does not fit)
        } else {
          DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 4, 1024, true> : bt_fused_fft_kernel<1, 4, 1024>), 1024));         // N = 4096
        }
      } else {
        const int npt = std::max(1, N / 256);
        if (polarised) {
          if (npt == 1) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 1, 256, true> : bt_fused_fft_kernel<4, 1, 256>), 256));
          else if (npt == 2) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 2, 256, true> : bt_fused_fft_kernel<4, 2, 256>), 256));      // N = 512
          else if (npt == 4) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 4, 256, true> : bt_fused_fft_kernel<4, 4, 256>), 256));
          else DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<4, 8, 256, true> : bt_fused_fft_kernel<4, 8, 256>), 256));
        } else {
          if (npt == 1) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 1, 256, true> : bt_fused_fft_kernel<1, 1, 256>), 256));
          else if (npt == 2) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 2, 256, true> : bt_fused_fft_kernel<1, 2, 256>), 256));
          else if (npt == 4) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 4, 256, true> : bt_fused_fft_kernel<1, 4, 256>), 256));
          else if (npt == 8) DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 8, 256, true> : bt_fused_fft_kernel<1, 8, 256>), 256));      // N = 2048
          else DM_TRY(fft_launch((cbm ? bt_fused_fft_kernel<1, 16, 256, true> : bt_fused_fft_kernel<1, 16, 256>), 256));
        }
      }
    }
    // (m, ring) pairs below rounding: exact zeros, no work (bt_ring_skip_lookup; DM_BT_RING_SKIP=0 computes them all)
    const bool ring_skip_off = getenv("DM_BT_RING_SKIP") && atoi(getenv("DM_BT_RING_SKIP")) == 0;   // (read per call: the tests flip it)
    const int* d_mskip = nullptr;
    if (!ring_skip_off && nring_dft > 0) {
      d_mskip = dm_ws_upload(ctx, bt_ring_skip_lookup(nside, lmax_grp, polarised != 0, ring_cth_host, ring_sth_host));
      if (!d_mskip) return DM_ENOMEM;
    }
    auto launch = [&](auto kern, int NMG, int NCG) {
      if (nring_dft == 0) return;
      const dim3 grid((unsigned)((ncol16 + 4 * NCG - 1) / (4 * NCG)), (unsigned)nring_dft, (unsigned)((nmg + NMG - 1) / NMG));
      DM_PLAUNCH(ctx, DM_PROF_BT_RING, kern, grid, dim3(256), 0, ctx->stream, gh.g, fr, syn->beams_dev, bstride, d_fc, ncol16, m_lo, cnt, d_rw, G,
                         ncp, d_caps, d_mskip);
    };
    // 32 complex accumulators per lane (64 AGPRs) keep two waves per SIMD: the sincos of one wave runs under
    // the MFMAs of the other
    // measured: twice / four times the m-values per pass (<4,4,1>, <4,8,1>, <1,8,2>, <1,16,1>) halve the repeated
    // synthesis but cost a wave per SIMD — no faster on configs[1] or configs[2]
    // shared-synthesis kernel: the four waves of a workgroup take different m-values (16 NMG per pass)
    auto launch2 = [&](auto kern, int NMG, int NCG) {
      if (nring_dft == 0) return;
      const dim3 grid((unsigned)((ncol16 + NCG - 1) / NCG), (unsigned)nring_dft, (unsigned)((nmg + 4 * NMG - 1) / (4 * NMG)));
      DM_PLAUNCH(ctx, DM_PROF_BT_RING, kern, grid, dim3(256), 0, ctx->stream, gh.g, fr, syn->beams_dev, bstride, d_fc, ncol16, m_lo, cnt, d_rw, G,
                         ncp, d_caps, d_mskip);
    };
    static const int shared_env = getenv("DM_FDFT_SHARED") ? atoi(getenv("DM_FDFT_SHARED")) : 1;
    if (polarised) {
      if (nmg <= 1) launch((cbm ? bt_fused_dft_kernel<4, 1, 4, true> : bt_fused_dft_kernel<4, 1, 4>), 1, 4);
      else if (shared_env == 2 && nmg > 8) launch2((cbm ? bt_fused_dft2_kernel<4, 4, 1, true> : bt_fused_dft2_kernel<4, 4, 1>), 4, 1);
      else if (shared_env >= 1 && nmg > 2) launch2((cbm ? bt_fused_dft2_kernel<4, 2, 1, true> : bt_fused_dft2_kernel<4, 2, 1>), 2, 1);
      else launch((cbm ? bt_fused_dft_kernel<4, 2, 2, true> : bt_fused_dft_kernel<4, 2, 2>), 2, 2);
    } else {
      if (nmg <= 1) launch((cbm ? bt_fused_dft_kernel<1, 1, 8, true> : bt_fused_dft_kernel<1, 1, 8>), 1, 8);
      else if (shared_env == 2 && nmg > 8) launch2((cbm ? bt_fused_dft2_kernel<1, 4, 4, true> : bt_fused_dft2_kernel<1, 4, 4>), 4, 4);
      else if (shared_env >= 1 && nmg > 4) launch2((cbm ? bt_fused_dft2_kernel<1, 2, 4, true> : bt_fused_dft2_kernel<1, 2, 4>), 2, 4);
      else launch((cbm ? bt_fused_dft_kernel<1, 4, 4, true> : bt_fused_dft_kernel<1, 4, 4>), 4, 4);
    }
    DM_HIP(ctx, hipGetLastError());
  }
  auto ring_dft = [&](const cplx* maps) -> int {
    std::vector<dm_gemm_desc> g;
    g.reserve(nring);
    for (int r = 0; r < nring; ++r) {
      // C[mm][colp] (ld = nring*ncp, origin at ring r) = w_r * tw_r[mm][j] * maps[colp][start_r + j]
      g.push_back(dm_gemm_make(tw + toff[r], gh.nphi[r], 1, false, maps + gh.start[r], 1, npix, false,
                               G + (size_t)r * ncp, nring * ncp, nm, ncp, gh.nphi[r], ring_w_host ? ring_w_host[r] : 1.0));
    }
    return dm_gemm_grouped_launch(ctx, g);
  };
  if (cnt > 0 && !fused) DM_TRY(ring_dft(reinterpret_cast<const cplx*>(maps_dev)));

  // ---- Legendre tables up to lmax_grp
  std::vector<size_t> loff(std::max(cnt, 1), 0);  // loff[m - m_lo]
  size_t ltot = 0;
  for (int m = m_lo; m <= mtop; ++m) { loff[m - m_lo] = ltot; ltot += (size_t)(lmax_grp + 1 - m) * nring; }
  size_t* d_loff = dm_ws_upload(ctx, loff);
  double* lam = dm_ws_alloc_t<double>(ctx, std::max<size_t>(ltot, 1));
  double* Wt = polarised ? dm_ws_alloc_t<double>(ctx, std::max<size_t>(ltot, 1)) : nullptr;
  double* Xt = polarised ? dm_ws_alloc_t<double>(ctx, std::max<size_t>(ltot, 1)) : nullptr;
  if (!d_loff || !lam || (polarised && (!Wt || !Xt))) return DM_ENOMEM;
  if (cnt > 0)
    DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_legendre_kernel, dim3((nring + 63) / 64, cnt), dim3(64), 0, ctx->stream, gh.g, lmax_grp, m_lo,
                       mtop, 4.0 * kPi / (double)npix, d_loff, lam, Wt, Xt);
  DM_HIP(ctx, hipGetLastError());

  // ---- clear the destination rows of this group for every m (then GEMMs fill l in [m, lmax_grp])
  cplx* bm = reinterpret_cast<cplx*>(beam_m_dev);
  std::vector<int> cf(col_f_host, col_f_host + ncol), cb(col_b_host, col_b_host + ncol),
      cl(col_lmax_host, col_lmax_host + ncol);
  for (int c = 0; c < ncol; ++c) DM_ARG(ctx, cf[c] >= 0 && cf[c] < F && cb[c] >= 0 && cb[c] < B && cl[c] <= lmax_grp);
  int* d_cf = dm_ws_upload(ctx, cf);
  int* d_cb = dm_ws_upload(ctx, cb);
  int* d_cl = dm_ws_upload(ctx, cl);
  std::vector<int> neg1(ncol, -1);
  int* d_neg = dm_ws_upload(ctx, neg1);
  if (!d_cf || !d_cb || !d_cl || !d_neg) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_mask_kernel, dim3((2 * P * L + 255) / 256, ncol, nmblk), dim3(256), 0, ctx->stream, bm, F, B,
                     P, L, m_hi, ncol, d_cf, d_cb, d_neg);

  // ---- Legendre products.  Columns of a group are arbitrary (f, b) pairs, so one GEMM row per
  // column would be wasteful; instead consecutive columns with the same f and consecutive b are
  // merged into runs (the usual case: all baselines of a frequency in order).
  struct run { int c0, n, f, b0; };
  std::vector<run> runs;
  for (int c = 0; c < ncol;) {
    int e = c + 1;
    while (e < ncol && cf[e] == cf[c] && cb[e] == cb[e - 1] + 1) ++e;
    runs.push_back(run{c, e - c, cf[c], cb[c]});
    c = e;
  }
  // Terms whose outputs accumulate (E and B each take two products) go in separate launches
  // so that no two tiles of one launch touch the same C entries.
  // (the refinement path synthesises from G-shaped buffers and keeps the plain sum; DM_BT_FOLD=0 switches the fold off)
  static const bool fold_off = getenv("DM_BT_FOLD") && atoi(getenv("DM_BT_FOLD")) == 0;
  const bool folded = (niter == 0 || harmonic) && !fold_off && cnt > 0 && nring >= 3;
  if (folded) {
    const size_t tot = (size_t)nm * (nring / 2) * ncp;
    const unsigned nb = (unsigned)std::min<size_t>((tot + 255) / 256, 65536);
    DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_fold_kernel, dim3(nb), dim3(256), 0, ctx->stream, G, nm, nring, (size_t)ncp);
  }
  auto legendre_analysis = [&](bool accumulate) -> int {
  for (int pass = 0; pass < (polarised ? 2 : 1); ++pass) {
    std::vector<dm_gemm_desc> g;
    for (int m = m_lo; m <= mtop; ++m) {
      const int Lm = lmax_grp + 1 - m;
      for (int s = 0; s < 2; ++s) {
        if (m == 0 && s == 1) continue;  // the -m slot of m = 0 stays zero (beamtransfer.py:624)
        const int mm = (s == 0) ? (m - m_lo) : (cnt + m - m_lo);
        const cplx* Gm = G + (size_t)mm * nring * ncp;
        for (const run& rn : runs) {
          cplx* out = bm + ((((size_t)(m - m_lo) * F + rn.f) * 2 + s) * B + rn.b0) * P * L + m;
          auto add = [&](int pa, const double* tab, int pout, double are, double aim, double beta) {
            if (!folded) {
              dm_gemm_desc d = dm_gemm_make(Gm + (fused ? (size_t)pa * ncol + rn.c0 : (size_t)rn.c0 * P + pa), fused ? 1 : P, ncp, s == 1,
                                            tab + loff[m - m_lo], 1, nring, false,
                                            out + (size_t)pout * L, P * L, rn.n, Lm, nring, are, beta, nullptr,
                                            DM_GEMM_B_REAL);
              d.alpha_im = aim;
              g.push_back(d);
              return;
            }
            // folded rings: the outputs l = m + 2 j (+ 1) are interleaved (column stride 2 of C), each parity sums over one
            // half of the rings — the northern one (with the equator) where the function is even in z, the southern one
            // where it is odd; lambda and W are even for even l - m, X for odd l - m
            const int mid = nring / 2;
            for (int par = 0; par < 2; ++par) {
              const int nl = (Lm - par + 1) / 2;       // l' = par, par + 2, ... < Lm
              if (nl <= 0) continue;
              const bool even_fn = (tab == Xt) ? par == 1 : par == 0;
              const int r0 = even_fn ? 0 : mid + 1, nr = even_fn ? mid + 1 : mid;
              if (nr <= 0) continue;
              dm_gemm_desc d = dm_gemm_make(Gm + (size_t)r0 * ncp + (fused ? (size_t)pa * ncol + rn.c0 : (size_t)rn.c0 * P + pa),
                                            fused ? 1 : P, ncp, s == 1,
                                            tab + loff[m - m_lo] + (size_t)par * nring + r0, 1, 2 * nring, false,
                                            out + (size_t)pout * L + par, P * L, rn.n, nl, nr, are, beta, nullptr,
                                            DM_GEMM_B_REAL);
              d.csc = 2;
              d.alpha_im = aim;
              g.push_back(d);
            }
          };
          const double b0 = accumulate ? 1.0 : 0.0;
          if (!polarised) {
            add(0, lam, 0, 1.0, 0.0, b0);
          } else if (pass == 0) {
            add(0, lam, 0, 1.0, 0.0, b0);  // T = lam . G^I
            add(3, lam, 3, 1.0, 0.0, b0);  // V = lam . G^V
            add(1, Wt, 1, 1.0, 0.0, b0);   // E  = W . G^Q ...
            add(2, Wt, 2, 1.0, 0.0, b0);   // B  = W . G^U ...
          } else {
            add(2, Xt, 1, 0.0, -1.0, 1.0);  // E -= i X . G^U
            add(1, Xt, 2, 0.0, 1.0, 1.0);   // B += i X . G^Q
          }
        }
      }
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g));
  }
  // ---- per-column band limit
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_mask_kernel, dim3((2 * P * L + 255) / 256, ncol, nmblk), dim3(256), 0, ctx->stream, bm, F, B,
                     P, L, m_hi, ncol, d_cf, d_cb, d_cl);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
  };
  {
    static const bool host_times = getenv("DM_TIME_HOST") != nullptr;  // debugging aid
    const auto t0 = std::chrono::steady_clock::now();
    DM_TRY(legendre_analysis(false));
    if (host_times)
      fprintf(stderr, "HOSTTIME bt_sht_impl: Legendre descriptors + launch %.3f ms (ncol %d, m %d..%d)\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ncol, m_lo, m_hi);
  }

  // ---- Jacobi refinement (healpy map2alm `iter`): coefficients += analysis(map - synthesis(coefficients)).
  // With c_lm = sum_pix w f Y_lm (the reference's conj(SHT(conj f))) the synthesis is f = sum_lm c_lm conj(Y_lm):
  //   H[+m][col][ring] = sum_l lambda_lm(ring) b0_lm,   H[-m] = sum_l lambda_lm conj(b1_lm)   (the fold undone)
  //   f[col][ring, j]  = sum_mm conj(tw[mm][j]) H[mm][col][ring]
  // and for the spin-2 pair the Hermitian 2x2 block [[W, -iX], [iX, W]] of the analysis applied once more.
  // The tables carry the quadrature weight w = 4 pi / npix: the synthesis divides it out again.
  if (niter > 0 && harmonic && cnt > 0) {
    for (int c = 0; c < ncol; ++c) DM_ARG(ctx, cf[c] == 0 && cb[c] == c);
    const int Lg = lmax_grp + 1;
    const double wq = 4.0 * kPi / (double)npix, iw = 1.0 / wq;
    const bt_alias_info& al = bt_alias_lookup(nside, lmax_grp, polarised != 0, ring_cth_host, ring_sth_host);
    // alias terms couple the m <= mcut among themselves: a call that touches them must hold all of them (bt_sht_refined)
    const int Mc = (al.ia > 0 && m_lo == 0) ? std::min(al.mcut, mtop) : -1;
    DM_ARG(ctx, al.ia == 0 || m_lo == 0 || m_lo > al.mcut);
    DM_ARG(ctx, Mc < 0 || mtop >= std::min(al.mcut, lmax_grp));
    const int nra = Mc >= 0 ? 2 * al.ia : 0;
    DM_ARG(ctx, row_pad == nra);
    static const bool kfast = !(getenv("DM_SHT_KNFAST") && atoi(getenv("DM_SHT_KNFAST")) == 1);
    // ---- per-ring factor of A o S and the Gram matrices K_m, extended by the alias-ring tables for m <= Mc:
    //      block m is a (Lm + xa) x Lm operand, xa = nra alias rows
    std::vector<double> sc(nring);
    for (int r = 0; r < nring; ++r) sc[r] = (ring_w_host ? ring_w_host[r] : 1.0) * (double)gh.nphi[r] * iw;
    double* d_sc = dm_ws_upload(ctx, sc);
    std::vector<size_t> koff(cnt);
    size_t ktot = 0;
    auto xa = [&](int m) { return m <= Mc ? nra : 0; };
    for (int m = m_lo; m <= mtop; ++m) { koff[m - m_lo] = ktot; const size_t Lm = lmax_grp + 1 - m; ktot += Lm * (Lm + xa(m)); }
    double* Kl = dm_ws_alloc_t<double>(ctx, ktot);
    double* Kp = polarised ? dm_ws_alloc_t<double>(ctx, ktot) : nullptr;
    double* Kx = polarised ? dm_ws_alloc_t<double>(ctx, ktot) : nullptr;
    if (!d_sc || !Kl || (polarised && (!Kp || !Kx))) return DM_ENOMEM;
    {
      dm_ws_scope tmp_scope(ctx);   // scaled tables: released (stream-ordered) once the K products are queued
      double* lamS = dm_ws_alloc_t<double>(ctx, ltot);
      double* WS = polarised ? dm_ws_alloc_t<double>(ctx, ltot) : nullptr;
      double* XS = polarised ? dm_ws_alloc_t<double>(ctx, ltot) : nullptr;
      if (!lamS || (polarised && (!WS || !XS))) return DM_ENOMEM;
      const unsigned nb = (unsigned)std::min<size_t>((ltot + 255) / 256, 65536);
      DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_scale_table_kernel, dim3(nb), dim3(256), 0, ctx->stream, lam, lamS, ltot, nring, d_sc);
      if (polarised) {
        DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_scale_table_kernel, dim3(nb), dim3(256), 0, ctx->stream, Wt, WS, ltot, nring, d_sc);
        DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_scale_table_kernel, dim3(nb), dim3(256), 0, ctx->stream, Xt, XS, ltot, nring, d_sc);
      }
      for (int pass = 0; pass < (polarised ? 2 : 1); ++pass) {
        std::vector<dm_gemm_desc> g;
        for (int m = m_lo; m <= mtop; ++m) {
          const int Lm = lmax_grp + 1 - m;
          const size_t lo = loff[m - m_lo], ko = koff[m - m_lo];
          // K is symmetric: row i of the product lands where the chosen layout keeps (k = i, n) resp. (k, n = i)
          const int ldk = kfast ? Lm + xa(m) : Lm;
          auto kadd = [&](const double* a, const double* b, double* c, double beta) {
            g.push_back(dm_gemm_make(reinterpret_cast<const cplx*>(a + lo), nring, 1, false, b + lo, 1, nring, false,
                                     reinterpret_cast<cplx*>(c + ko), ldk, Lm, Lm, nring, 1.0, beta, nullptr, DM_GEMM_ALL_REAL));
          };
          if (pass == 0) {
            kadd(lam, lamS, Kl, 0.0);
            if (polarised) { kadd(Wt, WS, Kp, 0.0); kadd(Wt, XS, Kx, 0.0); }
          } else {
            kadd(Xt, XS, Kp, 1.0);
            kadd(Xt, WS, Kx, 1.0);
          }
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
    }
    // ---- alias rings: their own small tables (same recurrences, same bits as the full tables)
    const int nmmA = 2 * (Mc + 1);
    double *lamA = nullptr, *WA = nullptr, *XA = nullptr, *d_scA = nullptr;
    int *d_nphiA = nullptr, *d_mlimA = nullptr;
    cplx* Ha = nullptr;
    std::vector<size_t> loffA(std::max(Mc + 1, 1), 0);
    if (Mc >= 0) {
      std::vector<double> geoA(2 * (size_t)nra), scA(nra);
      std::vector<int> nphiA(nra), mlimA(nra);
      for (int ja = 0; ja < nra; ++ja) {
        const int r = ja < al.ia ? ja : nring - nra + ja;
        geoA[ja] = gh.cth[r];
        geoA[nra + ja] = gh.sth[r];
        nphiA[ja] = gh.nphi[r];
        mlimA[ja] = al.mlim[ja < al.ia ? ja : nra - 1 - ja];
        scA[ja] = (ring_w_host ? ring_w_host[r] : 1.0) * (double)gh.nphi[r];
      }
      double* d_geoA = dm_ws_upload(ctx, geoA);
      d_scA = dm_ws_upload(ctx, scA);
      d_nphiA = dm_ws_upload(ctx, nphiA);
      d_mlimA = dm_ws_upload(ctx, mlimA);
      size_t ltotA = 0;
      for (int m = 0; m <= Mc; ++m) { loffA[m] = ltotA; ltotA += (size_t)(lmax_grp + 1 - m) * nra; }
      size_t* d_loffA = dm_ws_upload(ctx, loffA);
      size_t* d_koff = dm_ws_upload(ctx, koff);
      lamA = dm_ws_alloc_t<double>(ctx, ltotA);
      if (polarised) { WA = dm_ws_alloc_t<double>(ctx, ltotA); XA = dm_ws_alloc_t<double>(ctx, ltotA); }
      Ha = dm_ws_alloc_t<cplx>(ctx, (size_t)nmmA * nra * ncp);
      if (!d_geoA || !d_scA || !d_nphiA || !d_mlimA || !d_loffA || !d_koff || !lamA || (polarised && (!WA || !XA)) || !Ha)
        return DM_ENOMEM;
      ring_geo ga = gh.g;
      ga.cth = d_geoA;
      ga.sth = d_geoA + nra;
      ga.nring = nra;
      DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_legendre_kernel, dim3((nra + 63) / 64, Mc + 1), dim3(64), 0, ctx->stream, ga, lmax_grp, 0, Mc, wq,
                         d_loffA, lamA, WA, XA);
      const dim3 fg(64, Mc + 1);
      DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_kext_fill_kernel, fg, dim3(256), 0, ctx->stream, lamA, d_loffA, Kl, d_koff, lmax_grp, nra, kfast ? 1 : 0);
      if (polarised) {
        DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_kext_fill_kernel, fg, dim3(256), 0, ctx->stream, WA, d_loffA, Kp, d_koff, lmax_grp, nra, kfast ? 1 : 0);
        DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_kext_fill_kernel, fg, dim3(256), 0, ctx->stream, XA, d_loffA, Kx, d_koff, lmax_grp, nra, kfast ? 1 : 0);
      }
    }
    // ---- iteration on the increments:  d_0 = a_0,  d_{k+1} = d_k - mask((A o S) d_k),  a_n = sum_k d_k
    const size_t nacc = (size_t)nmblk * 2 * ncol * P * L;
    cplx* dbuf = dm_ws_alloc_t<cplx>(ctx, nacc);
    cplx* tbuf = dm_ws_alloc_t<cplx>(ctx, nacc);
    if (!dbuf || !tbuf) return DM_ENOMEM;
    DM_HIP(ctx, hipMemcpyAsync(dbuf, bm, sizeof(cplx) * nacc, hipMemcpyDeviceToDevice, ctx->stream));
    auto blk = [&](cplx* base, int m, int s2) { return base + ((size_t)(m - m_lo) * 2 + s2) * ncol * P * L + m; };
    for (int it = 0; it < niter; ++it) {
      if (Mc >= 0) {
        // synthesis on the alias rings: Ha[mm][col p][ja] = (1 / w) sum_l tabA[l][ja] d[col][p][l]
        for (int pass = 0; pass < (polarised ? 2 : 1); ++pass) {
          std::vector<dm_gemm_desc> g;
          for (int m = 0; m <= Mc; ++m) {
            const int Lm = lmax_grp + 1 - m;
            for (int s2 = 0; s2 < 2; ++s2) {
              cplx* Hm = Ha + (size_t)(s2 * (Mc + 1) + m) * nra * ncp;
              if (m == 0 && s2 == 1) continue;   // never read by the fold
              const cplx* in = blk(dbuf, m, s2);
              auto add = [&](int pin, const double* tab, int pout, double are, double aim, double beta) {
                dm_gemm_desc dsc = dm_gemm_make(in + (size_t)pin * L, P * L, 1, false, tab + loffA[m], nra, 1, false,
                                                Hm + (size_t)pout * nra, P * nra, ncol, nra, Lm, are * iw, beta, nullptr,
                                                DM_GEMM_B_REAL);
                dsc.alpha_im = aim * iw;
                g.push_back(dsc);
              };
              if (!polarised) {
                add(0, lamA, 0, 1.0, 0.0, 0.0);
              } else if (pass == 0) {
                add(0, lamA, 0, 1.0, 0.0, 0.0);
                add(3, lamA, 3, 1.0, 0.0, 0.0);
                add(1, WA, 1, 1.0, 0.0, 0.0);   // Q = W E - i X B
                add(2, WA, 2, 1.0, 0.0, 0.0);   // U = W B + i X E
              } else {
                add(2, XA, 1, 0.0, -1.0, 1.0);
                add(1, XA, 2, 0.0, 1.0, 1.0);
              }
            }
          }
          DM_TRY(dm_gemm_grouped_launch(ctx, g));
        }
        DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_alias_fold_kernel, dim3((unsigned)((ncp + 3) / 4), (nra + 63) / 64, nmmA), dim3(64, 4), 0, ctx->stream,
                           Ha, dbuf, Mc, nra, (size_t)ncp, L, Lg, d_nphiA, d_mlimA, d_scA);
      }
      // t = [d | folded rings] [K ; alias tables]: the Gram product and the analysis of the folded rings in one K dimension
      for (int pass = 0; pass < (polarised ? 2 : 1); ++pass) {
        std::vector<dm_gemm_desc> g;
        for (int m = m_lo; m <= mtop; ++m) {
          const int Lm = lmax_grp + 1 - m, Kd = Lm + xa(m);
          const size_t ko = koff[m - m_lo];
          for (int s2 = 0; s2 < 2; ++s2) {
            if (m == 0 && s2 == 1) continue;
            const cplx* in = blk(dbuf, m, s2);
            cplx* out = blk(tbuf, m, s2);
            auto add = [&](int pin, const double* Km, int pout, double are, double aim, double beta) {
              dm_gemm_desc dsc = dm_gemm_make(in + (size_t)pin * L, P * L, 1, false, Km + ko, kfast ? 1 : Lm, kfast ? Kd : 1, false,
                                              out + (size_t)pout * L, P * L, ncol, Lm, Kd, are, beta, nullptr, DM_GEMM_B_REAL);
              dsc.alpha_im = aim;
              g.push_back(dsc);
            };
            if (!polarised) {
              add(0, Kl, 0, 1.0, 0.0, 0.0);
            } else if (pass == 0) {
              add(0, Kl, 0, 1.0, 0.0, 0.0);
              add(3, Kl, 3, 1.0, 0.0, 0.0);
              add(1, Kp, 1, 1.0, 0.0, 0.0);   // E' = K_P E - i K_X B  (+ W g_Q - i X g_U from the alias slots)
              add(2, Kp, 2, 1.0, 0.0, 0.0);   // B' = K_P B + i K_X E  (+ W g_U + i X g_Q)
            } else {
              add(2, Kx, 1, 0.0, -1.0, 1.0);
              add(1, Kx, 2, 0.0, 1.0, 1.0);
            }
          }
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
      DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_refine_update_kernel, dim3((P * L + 255) / 256, ncol, 2 * cnt), dim3(256), 0, ctx->stream, bm, dbuf,
                         tbuf, m_lo, ncol, P, L, Lg, d_cl);
      DM_HIP(ctx, hipGetLastError());
    }
  } else if (niter > 0 && cnt > 0) {
    const double iw = (double)npix / (4.0 * kPi);
    cplx* res = dm_ws_alloc_t<cplx>(ctx, (size_t)ncp * npix);
    if (!res) return DM_ENOMEM;
    cplx* H = G;  // same size: (nm, ncp, nring) here against (nm, nring, ncp) there
    for (int it = 0; it < niter; ++it) {
      DM_HIP(ctx, hipMemcpyAsync(res, maps_dev, sizeof(cplx) * (size_t)ncp * npix, hipMemcpyDeviceToDevice, ctx->stream));
      for (int pass = 0; pass < (polarised ? 2 : 1); ++pass) {
        std::vector<dm_gemm_desc> g;
        for (int m = m_lo; m <= mtop; ++m) {
          const int Lm = lmax_grp + 1 - m;
          for (int s = 0; s < 2; ++s) {
            const int mm = (s == 0) ? (m - m_lo) : (cnt + m - m_lo);
            cplx* Hm = H + (size_t)mm * ncp * nring;
            for (const run& rn : runs) {
              const cplx* in = bm + ((((size_t)(m - m_lo) * F + rn.f) * 2 + s) * B + rn.b0) * P * L + m;
              // C[col][ring] (row stride P * nring: the P maps of a column are adjacent rows) = A[col][l] * tab[l][ring]
              auto add = [&](int pin, const double* tab, int pout, double are, double aim, double beta) {
                dm_gemm_desc d = dm_gemm_make(in + (size_t)pin * L, P * L, 1, s == 1, tab + loff[m - m_lo], nring, 1, false,
                                              Hm + ((size_t)rn.c0 * P + pout) * nring, P * nring, rn.n, nring, Lm,
                                              are * iw, beta, nullptr, DM_GEMM_B_REAL);
                d.alpha_im = (s == 1 ? -aim : aim) * iw;  // H[-m] = conj(M b1) = conj(M) conj(b1)
                g.push_back(d);
              };
              if (!polarised) {
                add(0, lam, 0, 1.0, 0.0, 0.0);
              } else if (pass == 0) {
                add(0, lam, 0, 1.0, 0.0, 0.0);
                add(3, lam, 3, 1.0, 0.0, 0.0);
                add(1, Wt, 1, 1.0, 0.0, 0.0);   // Q  = W . E ...
                add(2, Wt, 2, 1.0, 0.0, 0.0);   // U  = W . B ...
              } else {
                add(2, Xt, 1, 0.0, -1.0, 1.0);  // Q -= i X . B
                add(1, Xt, 2, 0.0, 1.0, 1.0);   // U += i X . E
              }
            }
          }
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
      {
        std::vector<dm_gemm_desc> g;
        g.reserve(nring);
        for (int r = 0; r < nring; ++r)  // res[colp][start_r + j] -= sum_mm H[mm][colp][r] conj(tw_r[mm][j])
          g.push_back(dm_gemm_make(H + r, nring, ncp * nring, false, tw + toff[r], gh.nphi[r], 1, true, res + gh.start[r],
                                   npix, ncp, gh.nphi[r], nm, -1.0, 1.0));
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
      }
      DM_TRY(ring_dft(res));
      DM_TRY(legendre_analysis(true));
    }
  }
  // The fused path returns once everything is queued (stream-ordered with the caller's later work on the context's stream,
  // workspace re-use included: dm_ws_release); the refinement path keeps its wait.
  if ((niter > 0 && !harmonic) || !fused) DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}

// healpy's `iter` without the maps: the first analysis and the harmonic-space refinement run on a private coefficient
// buffer that holds the blocks the refinement of [m_lo, m_hi] depends on — the range itself, plus every m <= mcut when the
// range reaches into the m the polar rings alias (those couple among themselves only) — and the requested blocks are
// copied out.  The private buffer is (nm, 2, ncol, P, lmax_grp + 1): compact in l, columns adjacent.
static int bt_sht_refined(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                          int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                          const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev, int niter,
                          const double* ring_w_host, const bt_synth_in* syn) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, nside > 0 && ring_cth_host && ring_sth_host && lside >= 0 && m_lo >= 0 && m_hi >= m_lo && lmax_grp >= 0 &&
                  lmax_grp <= lside && F > 0 && B > 0 && ncol >= 0 && col_f_host && col_b_host && col_lmax_host && beam_m_dev &&
                  niter > 0);
  if (ncol == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);
  const int P = polarised ? 4 : 1, L = lside + 1, Ls = lmax_grp + 1;
  for (int c = 0; c < ncol; ++c) DM_ARG(ctx, col_f_host[c] >= 0 && col_f_host[c] < F && col_b_host[c] >= 0 && col_b_host[c] < B);
  const bt_alias_info& al = bt_alias_lookup(nside, lmax_grp, polarised != 0, ring_cth_host, ring_sth_host);
  int e_lo = m_lo, e_hi = m_hi;
  if (al.ia > 0 && m_lo <= al.mcut) { e_lo = 0; e_hi = std::max(m_hi, std::min(al.mcut, lmax_grp)); }
  const int enm = e_hi - e_lo + 1;
  // rows of the private buffers: Ls coefficients + one slot per alias ring (the folded rings ride in the K dimension of
  // the Gram products)
  const int pad = (al.ia > 0 && e_lo == 0 && std::min(e_hi, lmax_grp) >= 0) ? 2 * al.ia : 0;
  const int Lrow = Ls + pad;
  cplx* acc = dm_ws_alloc_t<cplx>(ctx, (size_t)enm * 2 * ncol * P * Lrow);
  if (!acc) return DM_ENOMEM;
  std::vector<int> zf(ncol, 0), ib(ncol);
  for (int c = 0; c < ncol; ++c) ib[c] = c;
  DM_TRY(bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lmax_grp, e_lo, e_hi, lmax_grp, 1, ncol, ncol,
                     zf.data(), ib.data(), col_lmax_host, maps_dev, acc, niter, ring_w_host, syn, true, pad));
  std::vector<int> cfv(col_f_host, col_f_host + ncol), cbv(col_b_host, col_b_host + ncol);
  int* d_cf = dm_ws_upload(ctx, cfv);
  int* d_cb = dm_ws_upload(ctx, cbv);
  if (!d_cf || !d_cb) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_scatter2_kernel, dim3((2 * P * L + 255) / 256, ncol, m_hi - m_lo + 1), dim3(256), 0, ctx->stream, acc, e_lo,
                     enm, Ls, Lrow, reinterpret_cast<cplx*>(beam_m_dev), m_lo, F, B, P, L, ncol, d_cf, d_cb);
  DM_HIP(ctx, hipGetLastError());
  return DM_OK;
}

int dm_bt_sht_opts(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                   int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                   const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev, int niter,
                   const double* ring_w_host) {
  if (!ctx) return DM_EARG;
  if (niter <= 0)
    return bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol,
                       col_f_host, col_b_host, col_lmax_host, maps_dev, beam_m_dev, 0, ring_w_host);
  // default: the refinement in harmonic space (no residual maps; bt_sht_refined).  DM_SHT_PIXEL_REFINE=1 keeps the
  // map-space form (synthesis, inverse ring DFT, residual map, re-analysis) for cross-checks.
  static const bool pixel_refine = getenv("DM_SHT_PIXEL_REFINE") && atoi(getenv("DM_SHT_PIXEL_REFINE")) == 1;
  if (!pixel_refine) {
    DM_ARG(ctx, maps_dev != nullptr);
    DM_TRY(bt_sht_refined(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol,
                          col_f_host, col_b_host, col_lmax_host, maps_dev, beam_m_dev, niter, ring_w_host, nullptr));
    DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DM_OK;
  }
  // The refinement synthesises the map from EVERY (l, m) of a column, whatever range of m the caller keeps (and the
  // telescope's mmax may lie below a column's lmax): it runs on a private coefficient buffer holding m = 0 .. lmax_grp
  // of this group's columns, laid out like beam_m with F = 1, B = ncol; the requested blocks are copied out at the end.
  DM_ARG(ctx, nside > 0 && lside >= 0 && m_lo >= 0 && m_hi >= m_lo && lmax_grp >= 0 && lmax_grp <= lside && F > 0 && B > 0 &&
                  ncol >= 0 && col_f_host && col_b_host && col_lmax_host && maps_dev && beam_m_dev);
  if (ncol == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);
  const int P = polarised ? 4 : 1, L = lside + 1, msrc = lmax_grp + 1;
  for (int c = 0; c < ncol; ++c) DM_ARG(ctx, col_f_host[c] >= 0 && col_f_host[c] < F && col_b_host[c] >= 0 && col_b_host[c] < B);
  cplx* cf = dm_ws_alloc_t<cplx>(ctx, (size_t)msrc * 2 * ncol * P * L);
  if (!cf) return DM_ENOMEM;
  std::vector<int> zf(ncol, 0), ib(ncol);
  for (int c = 0; c < ncol; ++c) ib[c] = c;
  DM_TRY(bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, 0, lmax_grp, lmax_grp, 1, ncol, ncol,
                     zf.data(), ib.data(), col_lmax_host, maps_dev, cf, niter, ring_w_host));
  std::vector<int> cfv(col_f_host, col_f_host + ncol), cbv(col_b_host, col_b_host + ncol);
  int* d_cf = dm_ws_upload(ctx, cfv);
  int* d_cb = dm_ws_upload(ctx, cbv);
  if (!d_cf || !d_cb) return DM_ENOMEM;
  DM_PLAUNCH(ctx, DM_PROF_BT_OTHER, bt_scatter_kernel, dim3((2 * P * L + 255) / 256, ncol, m_hi - m_lo + 1), dim3(256), 0, ctx->stream, cf, msrc,
                     reinterpret_cast<cplx*>(beam_m_dev), m_lo, F, B, P, L, ncol, d_cf, d_cb);
  DM_HIP(ctx, hipGetLastError());
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DM_OK;
}

// Beams -> beam_m blocks in one call: the Stokes maps are never written to memory (dm_bt_maps + dm_bt_sht_range fused).
int dm_bt_columns(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                  int polarised, int nbeam, const double* beams_dev, int ncol, const double* uv_host, const int* bi_host,
                  const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, const int* col_f_host,
                  const int* col_b_host, const int* col_lmax_host, void* beam_m_dev, const double* ring_w_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, frame_host && nbeam > 0 && beams_dev && uv_host && bi_host && bj_host);
  bt_synth_in syn{frame_host, nbeam, beams_dev, uv_host, bi_host, bj_host, 0};
  return bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol, col_f_host,
                     col_b_host, col_lmax_host, nullptr, beam_m_dev, 0, ring_w_host, &syn);
}

// dm_bt_columns for COMPLEX field patterns (beams_dev: nbeam maps of npix * ncomp complex128, zero below the horizon)
int dm_bt_columns_c(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                    int polarised, int nbeam, const void* beams_dev, int ncol, const double* uv_host, const int* bi_host,
                    const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, const int* col_f_host,
                    const int* col_b_host, const int* col_lmax_host, void* beam_m_dev, const double* ring_w_host) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, frame_host && nbeam > 0 && beams_dev && uv_host && bi_host && bj_host);
  bt_synth_in syn{frame_host, nbeam, reinterpret_cast<const double*>(beams_dev), uv_host, bi_host, bj_host, 1};
  return bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol, col_f_host,
                     col_b_host, col_lmax_host, nullptr, beam_m_dev, 0, ring_w_host, &syn);
}

// dm_bt_columns / dm_bt_columns_c with healpy's `iter`: niter Jacobi refinements of the quadrature carried out in harmonic
// space (no Stokes maps, no residual maps); complex_beams selects the complex-pattern kernels.
int dm_bt_columns_iter(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, const double* frame_host,
                       int polarised, int nbeam, const void* beams_dev, int complex_beams, int ncol, const double* uv_host,
                       const int* bi_host, const int* bj_host, int lside, int m_lo, int m_hi, int lmax_grp, int F, int B,
                       const int* col_f_host, const int* col_b_host, const int* col_lmax_host, void* beam_m_dev,
                       const double* ring_w_host, int niter) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, frame_host && nbeam > 0 && beams_dev && uv_host && bi_host && bj_host && niter >= 0);
  bt_synth_in syn{frame_host, nbeam, reinterpret_cast<const double*>(beams_dev), uv_host, bi_host, bj_host, complex_beams ? 1 : 0};
  if (niter == 0)
    return bt_sht_impl(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol, col_f_host,
                       col_b_host, col_lmax_host, nullptr, beam_m_dev, 0, ring_w_host, &syn);
  return bt_sht_refined(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol, col_f_host,
                        col_b_host, col_lmax_host, nullptr, beam_m_dev, niter, ring_w_host, &syn);
}

// The m the refinement of one nside group couples through the polar rings: n_alias_rings per cap, and mcut — a call for a
// range of m that starts at or below mcut transforms m = 0 .. max(m_hi, mcut) internally (-1: no coupling).  A function of
// (nside, lmax_grp, polarised) alone; callers use it to size their column chunks.
int dm_bt_alias_info(int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised, int lmax_grp,
                     int* n_alias_rings, int* mcut) {
  if (nside <= 0 || !ring_cth_host || !ring_sth_host || lmax_grp < 0 || !n_alias_rings || !mcut) return DM_EARG;
  const bt_alias_info& al = bt_alias_lookup(nside, lmax_grp, polarised != 0, ring_cth_host, ring_sth_host);
  *n_alias_rings = al.ia;
  *mcut = al.ia > 0 ? std::min(al.mcut, lmax_grp) : -1;
  return DM_OK;
}

int dm_bt_sht_range(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
                    int lside, int m_lo, int m_hi, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
                    const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev) {
  return dm_bt_sht_opts(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, m_lo, m_hi, lmax_grp, F, B, ncol,
                        col_f_host, col_b_host, col_lmax_host, maps_dev, beam_m_dev, 0, nullptr);
}

// all m-blocks 0 .. mmax
int dm_bt_sht(dm_ctx* ctx, int nside, const double* ring_cth_host, const double* ring_sth_host, int polarised,
              int lside, int mmax, int lmax_grp, int F, int B, int ncol, const int* col_f_host,
              const int* col_b_host, const int* col_lmax_host, const void* maps_dev, void* beam_m_dev) {
  if (!ctx) return DM_EARG;
  DM_ARG(ctx, mmax >= 0);
  return dm_bt_sht_range(ctx, nside, ring_cth_host, ring_sth_host, polarised, lside, 0, mmax, lmax_grp, F, B, ncol,
                         col_f_host, col_b_host, col_lmax_host, maps_dev, beam_m_dev);
}

}  // extern "C"
