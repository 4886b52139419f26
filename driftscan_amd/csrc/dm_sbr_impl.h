// dm_sbr_impl.h — two-stage tridiagonalisation (dense -> band on MFMA, band -> tridiagonal by bulge chasing)
// and the back-transformation of the second stage.  Included by dm_tridiag_impl.h inside the namespace of
// the 32-wide instantiation (the bandwidth is the panel width TNB = 32).  No include guard on purpose.
//
// Replaces the zhetrd inside scipy.linalg.eigh (drift/core/kltransform.py:89, :107) by the LAPACK 3.7
// two-stage scheme (zhetrd_he2hb + zhetrd_hb2st):
//
//   S1  dense -> band, bandwidth SB = 32.  Per panel of 32 columns: Householder QR of the block below the
//       band (sb_qr_* kernels, or the one-workgroup-per-matrix kernel for small matrices), T factor, then the
//       two-sided update  A22 <- Q^H A22 Q  as level-3 products through the grouped ZGEMM:
//         Y = A22 (V T),  W = Y - V (T^H V^H Y) / 2,  A22 -= V W^H + W V^H      (16 n^3 / 3 flop, MFMA-bound)
//       The trailing matrix is read 3 times per 32 columns instead of once per column: 5-10 x fewer HBM
//       bytes than the one-stage reduction (8 n^3 / 3 bytes), which is HBM-bound at 0.5 of the roofline.
//   S2  band -> real tridiagonal: sweep s annihilates column s below the sub-diagonal and chases the bulge
//       down the band with reflectors of length <= 32 (sb_chase_kernel).  One WAVE runs one sweep; a 32 x 32
//       block lives in the registers of the wave as 4 x 4 sub-blocks per lane (row and column reductions are
//       three butterfly steps each); sweep s + 1 follows sweep s two blocks behind, synchronised through
//       progress words (LDS within a workgroup, agent-scope flags across workgroups for one large matrix).
//   B2  X = Q2 Z: the n^2 / 64 short reflectors are applied by sb_q2_apply_kernel to column slabs of X that
//       stay in registers while a group of sweeps passes over them (VALU fp64 — as fast as the fp64 MFMA on
//       this part — with no T factors and no padding flops); slabs are independent, no inter-workgroup sync.
//   B1  X = Q1 (Q2 Z): the compact-WY path of the one-stage solver with the T factors of S1.

constexpr int SB = TNB;       // bandwidth = reflector length of the chase
constexpr int SLD = 2 * SB;   // band storage: column c holds A[c .. c + 2 SB - 1][c] (band + bulge)
constexpr int SQR = 256;      // indices per workgroup in the launched panel-QR kernels
constexpr int SBG = 32;       // sweeps per group in the second-stage back-transformation
constexpr int SBW = SBG + SB; // window rows of a diamond block (SBG + SB - 1, padded)

// complex multiply-adds spelled as four FMAs (the compiler turns a*b + c*d + e into mul + fma + add)
__device__ __forceinline__ void sb_cfma(cplx& acc, cplx a, cplx b) {        // acc += a b
  acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
__device__ __forceinline__ void sb_cfma_ca(cplx& acc, cplx a, cplx b) {     // acc += conj(a) b
  acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}
__device__ __forceinline__ void sb_cfms(cplx& x, cplx a, cplx b) {          // x -= a b
  x.x = fma(-a.x, b.x, x.x); x.x = fma(a.y, b.y, x.x);
  x.y = fma(-a.x, b.y, x.y); x.y = fma(-a.y, b.x, x.y);
}
__device__ __forceinline__ void sb_cfms_cb(cplx& x, cplx a, cplx b) {       // x -= a conj(b)
  x.x = fma(-a.x, b.x, x.x); x.x = fma(-a.y, b.y, x.x);
  x.y = fma(-a.y, b.x, x.y); x.y = fma(a.x, b.y, x.y);
}

struct sb_mat {
  cplx* A; int lda; int n;
  cplx* Vt;     // n x n: row k = reflector of column k (first stage): zero below index k + SB, 1 at k + SB
  cplx* Vp;     // SB x n panel of V, then W, then V again (her2k layout, see trd_mat)
  cplx* Wp;
  cplx* Vp2;
  cplx* Pw;     // SB x n: working copy of the panel columns, row q = column k0 + q of the Hermitian matrix
  cplx* tau;    // n
  cplx* Yp;     // (n / SQR + 1) x SB: partial dot products v_q^H P[:, c]
  double* Np;   // 2 x (n / SQR + 1): partial norms (double-buffered over q)
  int npstride; // n / SQR + 1
  cplx* Rb;     // look-ahead only (else NULL): n x SB, row c holds the part of R of column c that lies beyond its diagonal
                // block, A[c][i0 ..] with i0 = (c / SB + 1) SB — the trailing update of the previous panel is still
                // writing the rows of A this panel would put it in
};

// ---- S1: launched panel QR (any size; two launches per column, all matrices in lock-step) ----------------------

// copy the panel columns into the work array (math orientation) and clear the panel rows above the panel
__global__ __launch_bounds__(256) void sb_panel_load_kernel(const sb_mat* __restrict__ ms, int k0, int a0) {
  const sb_mat M = ms[blockIdx.y];
  const int n = M.n;
  if (n - k0 - SB < 2) return;
  const int i = a0 + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int i0 = k0 + SB;
  const cplx z = make_double2(0.0, 0.0);
  if (i >= i0) {
#pragma unroll 4
    for (int q = 0; q < SB; ++q) dm_stg(M.Pw, (size_t)q * n + i, cconj(dm_ldg(M.A, (size_t)(k0 + q) * M.lda + i)));
  } else {
    for (int q = 0; q < SB; ++q) {
      dm_stg(M.Vp, (size_t)q * n + i, z);
      dm_stg(M.Vp2, (size_t)q * n + i, z);
      dm_stg(M.Wp, (size_t)q * n + i, z);
    }
  }
}

__device__ __forceinline__ double sb_block_sum(double v, double* red) {
  v = dm_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// K_a(q): finish reflector q - 1 (apply it to the columns c >= q, write v, R and tau), partial norms of column q
__global__ __launch_bounds__(256) void sb_qr_update_kernel(const sb_mat* __restrict__ ms, int k0, int q) {
  const sb_mat M = ms[blockIdx.y];
  const int n = M.n;
  const int m = n - k0 - SB;
  if (m < 2) return;
  const int nch = (m + SQR - 1) / SQR;
  if ((int)blockIdx.x >= nch) return;
  const int nrf = min(SB, m - 1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int i = k0 + SB + blockIdx.x * SQR + tid;
  const bool valid = i < n;
  __shared__ cplx ys[SB];
  __shared__ double red[4];
  if (q >= 1) {
    const int r = q - 1;
    const int lead = k0 + SB + r;
    if (r < nrf) {
      double np = 0.0;
      const double* Npr = M.Np + (size_t)(r & 1) * M.npstride;
      for (int t = lane; t < nch; t += 64) np += dm_ldg(Npr, t);
      const cplx alpha = dm_ldg(M.Pw, (size_t)r * n + lead);
      const trd_refl R = trd_reflector_from(np, alpha);
      if (tid < SB) {
        cplx acc = make_double2(0.0, 0.0);
        if (tid >= q)
          for (int t = 0; t < nch; ++t) acc = cadd(acc, dm_ldg(M.Yp, (size_t)t * SB + tid));
        ys[tid] = acc;
      }
      __syncthreads();
      if (valid) {
        const cplx x = dm_ldg(M.Pw, (size_t)r * n + i);
        cplx v = cmul(x, R.scal);
        if (i == lead) v = make_double2(1.0, 0.0);
        if (i < lead) v = make_double2(0.0, 0.0);
        if (i >= lead) {
          const cplx tv = cmul(cconj(R.tau), v);  // H^H a = a - conj(tau) v (v^H a)
          for (int c = q; c < SB; ++c) {
            const cplx a = dm_ldg(M.Pw, (size_t)c * n + i);
            dm_stg(M.Pw, (size_t)c * n + i, csub(a, cmul(tv, ys[c])));
          }
        }
        dm_stg(M.Vp, (size_t)r * n + i, v);
        dm_stg(M.Vp2, (size_t)r * n + i, v);
        dm_stg(M.Vt, (size_t)(k0 + r) * n + i, v);
        if (i <= lead) dm_stg(M.A, (size_t)(k0 + r) * M.lda + i, i < lead ? cconj(x) : make_double2(R.beta, 0.0));
      }
      if (blockIdx.x == 0 && tid == 0) M.tau[k0 + r] = R.tau;
    } else {
      if (valid) {
        const cplx z = make_double2(0.0, 0.0);
        dm_stg(M.Vp, (size_t)r * n + i, z);
        dm_stg(M.Vp2, (size_t)r * n + i, z);
        dm_stg(M.Vt, (size_t)(k0 + r) * n + i, z);
        dm_stg(M.A, (size_t)(k0 + r) * M.lda + i, cconj(dm_ldg(M.Pw, (size_t)r * n + i)));
      }
      if (blockIdx.x == 0 && tid == 0) M.tau[k0 + r] = make_double2(0.0, 0.0);
    }
  }
  if (q < nrf) {
    double s = 0.0;
    if (valid && i > k0 + SB + q) s = cabs2(dm_ldg(M.Pw, (size_t)q * n + i));
    s = sb_block_sum(s, red);
    if (tid == 0) M.Np[(size_t)(q & 1) * M.npstride + blockIdx.x] = s;
  }
}

// K_b(q): partial dot products v_q^H P[:, c], c > q, of this workgroup's rows
__global__ __launch_bounds__(256) void sb_qr_dots_kernel(const sb_mat* __restrict__ ms, int k0, int q) {
  const sb_mat M = ms[blockIdx.y];
  const int n = M.n;
  const int m = n - k0 - SB;
  if (m < 2) return;
  const int nch = (m + SQR - 1) / SQR;
  if ((int)blockIdx.x >= nch) return;
  const int nrf = min(SB, m - 1);
  if (q >= nrf) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = k0 + SB + blockIdx.x * SQR + tid;
  const int lead = k0 + SB + q;
  double np = 0.0;
  const double* Npr = M.Np + (size_t)(q & 1) * M.npstride;
  for (int t = lane; t < nch; t += 64) np += dm_ldg(Npr, t);
  const cplx alpha = dm_ldg(M.Pw, (size_t)q * n + lead);
  const trd_refl R = trd_reflector_from(np, alpha);
  cplx v = make_double2(0.0, 0.0);
  if (i < n && i >= lead) {
    v = cmul(dm_ldg(M.Pw, (size_t)q * n + i), R.scal);
    if (i == lead) v = make_double2(1.0, 0.0);
  }
  __shared__ cplx part[4][SB];
  for (int c = q + 1; c < SB; ++c) {
    cplx t = make_double2(0.0, 0.0);
    if (i < n && i >= lead) {
      const cplx a = dm_ldg(M.Pw, (size_t)c * n + i);
      t = make_double2(v.x * a.x + v.y * a.y, v.x * a.y - v.y * a.x);  // conj(v) * a
    }
    t.x = dm_wave_sum(t.x);
    t.y = dm_wave_sum(t.y);
    if (lane == 0) part[wave][c] = t;
  }
  __syncthreads();
  if (tid > q && tid < SB) {
    const cplx s = cadd(cadd(part[0][tid], part[1][tid]), cadd(part[2][tid], part[3][tid]));
    M.Yp[(size_t)blockIdx.x * SB + tid] = s;
  }
}

// Butterfly sums without the LDS crossbar: DPP moves inside a 16-lane row (quad_perm for xor 1 and 2, row_half_mirror
// once the quads are uniform, row_ror:8 for xor 8) and the gfx950 lane swaps across rows: v_permlane16_swap(x, x) leaves
// {row0, row0, row2, row2} and {row1, row1, row3, row3}, v_permlane32_swap(x, x) the two halves, each in every lane.
__device__ __forceinline__ double sb_swap16_sum(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
}
__device__ __forceinline__ double sb_swap32_sum(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)(b & 0xffffffffll), hi = (unsigned)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
}
__device__ __forceinline__ double sb_quad_sum(double v) {  // sum over the four lanes of a quad
  v += dm_dpp_f64<0xB1>(v);  // quad_perm [1, 0, 3, 2]
  v += dm_dpp_f64<0x4E>(v);  // quad_perm [2, 3, 0, 1]
  return v;
}
// sum over the lanes that share the row set (bits 0..2 of the lane differ)
__device__ __forceinline__ cplx sb_sum_bc(cplx v) {
  v.x = sb_quad_sum(v.x); v.y = sb_quad_sum(v.y);
  v.x += dm_dpp_f64<0x141>(v.x); v.y += dm_dpp_f64<0x141>(v.y);  // row_half_mirror: the other quad of the 8 lanes
  return v;
}
// sum over the lanes that share the column set (bits 3..5 differ)
__device__ __forceinline__ cplx sb_sum_br(cplx v) {
  v.x += dm_dpp_f64<0x128>(v.x); v.y += dm_dpp_f64<0x128>(v.y);  // row_ror:8
  v.x = sb_swap16_sum(v.x); v.y = sb_swap16_sum(v.y);
  v.x = sb_swap32_sum(v.x); v.y = sb_swap32_sum(v.y);
  return v;
}
__device__ __forceinline__ cplx sb_from_lane(cplx v, int src) {
  return make_double2(__shfl(v.x, src, 64), __shfl(v.y, src, 64));
}

// Householder scalars from alpha and the squared norm of the rest (zlarfg), every lane the same
__device__ __forceinline__ trd_refl sb_reflector(double xnorm2, cplx alpha) {
  trd_refl R;
  if ((xnorm2 == 0.0 && alpha.y == 0.0) || alpha.x * alpha.x + alpha.y * alpha.y + xnorm2 < DM_REFL_TINY) {
    R.tau = make_double2(0.0, 0.0);
    R.beta = alpha.x;
    R.scal = make_double2(0.0, 0.0);
  } else {
    R.beta = -copysign(sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2), alpha.x);
    R.tau = make_double2((R.beta - alpha.x) / R.beta, -alpha.y / R.beta);
    const double dr = alpha.x - R.beta, di = alpha.y;
    const double den = dr * dr + di * di;
    R.scal = make_double2(dr / den, -di / den);
  }
  return R;
}

// ---- S1: fused panel QR, one workgroup per matrix (panels of at most SFR * SFT rows) ---------------------------------
//
// The 32 columns are factorised as four sub-panels of 8 that live in the registers of the workgroup (thread t holds
// the rows i0 + t + 512 r): the reflectors of the earlier sub-panels are applied to the sub-panel first (one workgroup
// reduction each), then 8 right-looking column steps.  A column step costs two workgroup reductions (norm + alpha, then
// the dot products with the remaining columns of the sub-panel) and no memory traffic; one launch replaces the 65
// launches per panel of the launched kernels above.
constexpr int SFT = 512;  // threads
constexpr int SFR = 3;    // rows per thread
constexpr int SFH = 8;    // columns per sub-panel

// Transposing butterfly: 32 doubles per lane -> lane L ends with the wave total of value (L >> 1).  Each stage pairs
// two values and halves their number: the gfx950 lane swaps exchange the halves (rows) of TWO registers at once, the
// stages inside a 16-lane row send one value of the pair and keep the other (DPP moves, no LDS crossbar).  ~140
// instructions for 32 values against ~800 for 32 separate all-reduces.
__device__ __forceinline__ void sb_swap32_pair(double& a, double b) {  // lower half: a summed, upper half: b summed
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)(ba & 0xffffffffll), (unsigned)(bb & 0xffffffffll), false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  a = __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
}
__device__ __forceinline__ void sb_swap16_pair(double& a, double b) {  // even rows: a summed, odd rows: b summed
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)(ba & 0xffffffffll), (unsigned)(bb & 0xffffffffll), false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
  a = __longlong_as_double(((long long)rh[0] << 32) | rl[0]) + __longlong_as_double(((long long)rh[1] << 32) | rl[1]);
}
__device__ __forceinline__ double sb_wave_reduce32(double (&v)[32]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 16; ++k) sb_swap32_pair(v[k], v[k + 16]);
#pragma unroll
  for (int k = 0; k < 8; ++k) sb_swap16_pair(v[k], v[k + 8]);
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double send = b3 ? v[k] : v[k + 4], keep = b3 ? v[k + 4] : v[k];
    v[k] = keep + dm_dpp_f64<0x128>(send);  // row_ror:8
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double send = b2 ? v[k] : v[k + 2], keep = b2 ? v[k + 2] : v[k];
    const double up = dm_dpp_f64<0x114>(send), dn = dm_dpp_f64<0x104>(send);  // row_shr:4 (from lane - 4), row_shl:4 (from lane + 4)
    v[k] = keep + (b2 ? up : dn);
  }
  {
    const double send = b1 ? v[0] : v[1], keep = b1 ? v[1] : v[0];
    v[0] = keep + dm_dpp_f64<0x4E>(send);  // quad_perm [2, 3, 0, 1]
  }
  return v[0] + dm_dpp_f64<0xB1>(v[0]);    // quad_perm [1, 0, 3, 2]
}

// sums of K <= 32 doubles over the workgroup (8 waves): the butterfly inside each wave, then through LDS
template <int K>
__device__ __forceinline__ void sb_wg_reduce(double (&v)[K], double* red /* [2][8 * 32 + 32] */, int& phase) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* r = red + (size_t)phase * (8 * 32 + 32);
  if constexpr (K <= 4) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double t = sb_quad_sum(v[k]);
      t += dm_dpp_f64<0x141>(t);
      t += dm_dpp_f64<0x128>(t);
      t = sb_swap16_sum(t);
      t = sb_swap32_sum(t);
      if (lane == 0) r[wave * 32 + k] = t;
    }
  } else {
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < K ? v[k] : 0.0;
    const double t = sb_wave_reduce32(w);
    if (!(lane & 1) && (lane >> 1) < K) r[wave * 32 + (lane >> 1)] = t;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) t += r[wv * 32 + threadIdx.x];
    r[8 * 32 + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = r[8 * 32 + k];
  phase ^= 1;  // the next reduction uses the other buffer
}

// Right-looking Householder QR of the SFH columns [cb, cb + SFH) of the panel held in a[][] (nrf = reflectors of the
// panel).  The column being factorised is always a[.][0]: the array is shifted by one column after every step, so the
// loop body is the same code for every column (columns past the end are zero and cost nothing but their flops).
// gpair[j] = v_{2j+1}^H v_{2j} is left for the later sub-panels, which apply the reflectors two at a time.
__device__ __forceinline__ void sb_fused_half(const sb_mat& M, cplx (&a)[SFR][SFH], int k0, int cb, int nrf, double* red, int& phase,
                                              cplx* gpair) {
  const int n = M.n;
  const int tid = threadIdx.x;
  const int i0 = k0 + SB;
  cplx vprev[SFR];
#pragma unroll
  for (int r = 0; r < SFR; ++r) vprev[r] = make_double2(0.0, 0.0);
  // The norm of the NEXT column below its diagonal and its diagonal element after this column's reflector follow from
  // sums over the rows i > lead + 1 of quantities known BEFORE the update (|a_i|^2, conj(v_i) a_i, |v_i|^2) and from two
  // single elements — all of which ride along with the dot products of this column:
  //   sigma'^2 = P1 - 2 Re(conj(f) P2) + |f|^2 P3,   alpha' = a_{lead+1} - v_{lead+1} f,   f = conj(tau) v^H a.
  // The downdated norm is taken only when it keeps at least half of P1 (then it is as accurate as a direct sum: the
  // rounding errors of the three terms are ~eps P1); otherwise the column gets its own reduction as before.
  bool pre_ok = false;
  double pre_sig = 0.0;
  cplx pre_alpha = make_double2(0.0, 0.0);
  static_assert(2 * SFH + 8 <= 32, "the workgroup reduction carries at most 32 values");
#pragma unroll 1
  for (int q = 0; q < SFH; ++q) {
    const int qq = cb + q;          // column of the panel
    const int lead = i0 + qq;
    const bool has = qq < nrf;
    trd_refl R;
    R.tau = make_double2(0.0, 0.0); R.scal = make_double2(0.0, 0.0); R.beta = 0.0;
    cplx v[SFR];
    if (has && pre_ok) {
      R = sb_reflector(pre_sig, pre_alpha);
    } else if (has) {
      double t3[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < SFR; ++r) {
        const int i = i0 + tid + SFT * r;
        if (i < n && i > lead) t3[0] += cabs2(a[r][0]);
        if (i == lead) { t3[1] = a[r][0].x; t3[2] = a[r][0].y; }
      }
      sb_wg_reduce<3>(t3, red, phase);
      R = sb_reflector(t3[0], make_double2(t3[1], t3[2]));
    }
#pragma unroll
    for (int r = 0; r < SFR; ++r) {
      const int i = i0 + tid + SFT * r;
      cplx x = cmul(a[r][0], R.scal);
      if (i == lead) x = make_double2(1.0, 0.0);
      if (i < lead || i >= n || !has) x = make_double2(0.0, 0.0);
      v[r] = x;
    }
    // the coupling of an odd reflector with its predecessor rides along with the dot products (or takes a small
    // reduction of its own at the last column of a sub-panel, where there are none)
    const bool want_g = (qq & 1) && cb + SFH < SB;
    cplx gacc = make_double2(0.0, 0.0);
    if (want_g) {
#pragma unroll
      for (int r = 0; r < SFR; ++r) sb_cfma_ca(gacc, v[r], vprev[r]);   // conj(v_qq) * v_{qq-1}
    }
    pre_ok = false;
    if (has && q + 1 < SFH) {
      double y[2 * SFH + 8];
#pragma unroll
      for (int c = 1; c < SFH; ++c) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < SFR; ++r) {
          sb_cfma_ca(acc, v[r], a[r][c]);  // conj(v) * a
        }
        y[2 * (c - 1)] = acc.x;
        y[2 * (c - 1) + 1] = acc.y;
      }
      y[2 * (SFH - 1)] = gacc.x;
      y[2 * (SFH - 1) + 1] = gacc.y;
      {
        // the look at the next column (c = 1): sums over the rows below ITS diagonal, and the two single elements
        double p1 = 0.0, p3 = 0.0;
        cplx p2 = make_double2(0.0, 0.0), an = make_double2(0.0, 0.0), vn = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < SFR; ++r) {
          const int i = i0 + tid + SFT * r;
          if (i < n && i > lead + 1) {
            p1 += cabs2(a[r][1]);
            p3 += cabs2(v[r]);
            sb_cfma_ca(p2, v[r], a[r][1]);
          }
          if (i == lead + 1) { an = a[r][1]; vn = v[r]; }
        }
        y[2 * SFH] = p1; y[2 * SFH + 1] = p3;
        y[2 * SFH + 2] = p2.x; y[2 * SFH + 3] = p2.y;
        y[2 * SFH + 4] = an.x; y[2 * SFH + 5] = an.y;
        y[2 * SFH + 6] = vn.x; y[2 * SFH + 7] = vn.y;
      }
      sb_wg_reduce<2 * SFH + 8>(y, red, phase);
      if (want_g && tid == 0) gpair[qq >> 1] = make_double2(y[2 * (SFH - 1)], y[2 * (SFH - 1) + 1]);
      const cplx ct = cconj(R.tau);
      if (qq + 1 < nrf) {
        const cplx f1 = cmul(ct, make_double2(y[0], y[1]));
        const double p1 = y[2 * SFH], p3 = y[2 * SFH + 1];
        const cplx p2 = make_double2(y[2 * SFH + 2], y[2 * SFH + 3]);
        const double sig = p1 - 2.0 * (f1.x * p2.x + f1.y * p2.y) + cabs2(f1) * p3;
        pre_alpha = csub(make_double2(y[2 * SFH + 4], y[2 * SFH + 5]), cmul(make_double2(y[2 * SFH + 6], y[2 * SFH + 7]), f1));
        pre_sig = sig > 0.0 ? sig : 0.0;
        pre_ok = p1 > 0.0 && sig >= 0.5 * p1;   // (uniform over the workgroup: every thread holds the same sums)
      }
#pragma unroll
      for (int c = 1; c < SFH; ++c) {
        const cplx f = cmul(ct, make_double2(y[2 * (c - 1)], y[2 * (c - 1) + 1]));
#pragma unroll
        for (int r = 0; r < SFR; ++r) sb_cfms(a[r][c], v[r], f);
      }
    }
    else if (want_g) {   // (uniform over the workgroup)
      double g2[2] = {gacc.x, gacc.y};
      sb_wg_reduce<2>(g2, red, phase);
      if (tid == 0) gpair[qq >> 1] = make_double2(g2[0], g2[1]);
    }
#pragma unroll
    for (int r = 0; r < SFR; ++r) vprev[r] = v[r];
    // outputs: v into the panel buffers and Vt, the column of R into the band part of A, tau
#pragma unroll
    for (int r = 0; r < SFR; ++r) {
      const int i = i0 + tid + SFT * r;
      if (i < n) {
        dm_stg(M.Vp, (size_t)qq * n + i, v[r]);
        dm_stg(M.Vp2, (size_t)qq * n + i, v[r]);
        dm_stg(M.Vt, (size_t)(k0 + qq) * n + i, v[r]);
        if (M.Rb) {
          if (has ? i <= lead : i - i0 < SB)
            dm_stg(M.Rb, (size_t)(k0 + qq) * SB + (i - i0), (!has || i < lead) ? cconj(a[r][0]) : make_double2(R.beta, 0.0));
        } else if (has) {
          if (i <= lead) dm_stg(M.A, (size_t)(k0 + qq) * M.lda + i, i < lead ? cconj(a[r][0]) : make_double2(R.beta, 0.0));
        } else {
          dm_stg(M.A, (size_t)(k0 + qq) * M.lda + i, cconj(a[r][0]));
        }
      }
    }
    if (tid == 0) M.tau[k0 + qq] = R.tau;
    // next column to the front
#pragma unroll
    for (int r = 0; r < SFR; ++r) {
#pragma unroll
      for (int c = 0; c + 1 < SFH; ++c) a[r][c] = a[r][c + 1];
      a[r][SFH - 1] = make_double2(0.0, 0.0);
    }
  }
}

// snap != 0: the panel is read from the snapshot M.Pw (row q = column k0 + q, already updated by the previous panel's
// reflectors: the look-ahead copy made before that panel's trailing update started) instead of from A.
__global__ __launch_bounds__(SFT) void sb_panel_fused_kernel(const sb_mat* __restrict__ ms, int k0, int a0, int snap) {
  const sb_mat M = ms[blockIdx.x];
  const int n = M.n;
  const int m = n - k0 - SB;
  if (m < 2) return;
  const int nrf = min(SB, m - 1);
  const int tid = threadIdx.x;
  const int i0 = k0 + SB;
  __shared__ double red[2 * (8 * 32 + 32)];
  __shared__ cplx gpair[SB / 2];
  int phase = 0;
  // the panel rows above the trailing matrix inside its first 64-aligned tile are zero in V and W
  for (int i = a0 + tid; i < i0; i += SFT) {
    const cplx z = make_double2(0.0, 0.0);
    for (int q = 0; q < SB; ++q) {
      dm_stg(M.Vp, (size_t)q * n + i, z);
      dm_stg(M.Vp2, (size_t)q * n + i, z);
      dm_stg(M.Wp, (size_t)q * n + i, z);
    }
  }
  cplx a[SFR][SFH];
#pragma unroll 1
  for (int cb = 0; cb < SB; cb += SFH) {
    // ---- sub-panel [cb, cb + SFH) into registers
#pragma unroll
    for (int r = 0; r < SFR; ++r) {
      const int i = i0 + tid + SFT * r;
#pragma unroll
      for (int c = 0; c < SFH; ++c)
        a[r][c] = (i < n) ? cconj(snap ? dm_ldg(M.Pw, (size_t)(cb + c) * n + i) : dm_ldg(M.A, (size_t)(k0 + cb + c) * M.lda + i))
                          : make_double2(0.0, 0.0);
    }
    // ---- the reflectors of the earlier sub-panels: a_c <- a_c - conj(tau_q) v_q (v_q^H a_c), q = 0 .. cb - 1 in turn
    __syncthreads();  // their vectors are in memory (this workgroup wrote them: L1 holds no older copy)
    // two reflectors per workgroup reduction:  H_{q+1}^H H_q^H a = a - v_q f_q - v_{q+1} f_{q+1},  f_q = conj(tau_q) v_q^H a,
    // f_{q+1} = conj(tau_{q+1}) (v_{q+1}^H a - (v_{q+1}^H v_q) f_q)  — the coupling v_{q+1}^H v_q was left in gpair
    int q = 0;
#pragma unroll 1
    for (; q + 1 < cb && q + 1 < nrf; q += 2) {
      cplx v1[SFR], v2[SFR];
#pragma unroll
      for (int r = 0; r < SFR; ++r) {
        const int i = i0 + tid + SFT * r;
        v1[r] = (i < n) ? dm_ldg(M.Vp, (size_t)q * n + i) : make_double2(0.0, 0.0);
        v2[r] = (i < n) ? dm_ldg(M.Vp, (size_t)(q + 1) * n + i) : make_double2(0.0, 0.0);
      }
      double y[4 * SFH];
#pragma unroll
      for (int c = 0; c < SFH; ++c) {
        cplx a1 = make_double2(0.0, 0.0), a2 = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < SFR; ++r) {
          sb_cfma_ca(a1, v1[r], a[r][c]);
          sb_cfma_ca(a2, v2[r], a[r][c]);
        }
        y[2 * c] = a1.x; y[2 * c + 1] = a1.y;
        y[2 * SFH + 2 * c] = a2.x; y[2 * SFH + 2 * c + 1] = a2.y;
      }
      sb_wg_reduce<4 * SFH>(y, red, phase);
      const cplx t1 = cconj(M.tau[k0 + q]), t2 = cconj(M.tau[k0 + q + 1]);
      const cplx g = gpair[q >> 1];
#pragma unroll
      for (int c = 0; c < SFH; ++c) {
        const cplx f1 = cmul(t1, make_double2(y[2 * c], y[2 * c + 1]));
        const cplx f2 = cmul(t2, csub(make_double2(y[2 * SFH + 2 * c], y[2 * SFH + 2 * c + 1]), cmul(g, f1)));
#pragma unroll
        for (int r = 0; r < SFR; ++r) {
          sb_cfms(a[r][c], v1[r], f1);
          sb_cfms(a[r][c], v2[r], f2);
        }
      }
    }
#pragma unroll 1
    for (; q < cb && q < nrf; ++q) {
      cplx v[SFR];
#pragma unroll
      for (int r = 0; r < SFR; ++r) {
        const int i = i0 + tid + SFT * r;
        v[r] = (i < n) ? dm_ldg(M.Vp, (size_t)q * n + i) : make_double2(0.0, 0.0);
      }
      const cplx tq = M.tau[k0 + q];
      double y[2 * SFH];
#pragma unroll
      for (int c = 0; c < SFH; ++c) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < SFR; ++r) {
          sb_cfma_ca(acc, v[r], a[r][c]);
        }
        y[2 * c] = acc.x;
        y[2 * c + 1] = acc.y;
      }
      sb_wg_reduce<2 * SFH>(y, red, phase);
      const cplx ct = cconj(tq);
#pragma unroll
      for (int c = 0; c < SFH; ++c) {
        const cplx f = cmul(ct, make_double2(y[2 * c], y[2 * c + 1]));
#pragma unroll
        for (int r = 0; r < SFR; ++r) sb_cfms(a[r][c], v[r], f);
      }
    }
    sb_fused_half(M, a, k0, cb, nrf, red, phase, gpair);
  }
}

// Split-K for the skinny products of the first stage (32 x 32 Gram matrices and 32 x 128 blocks of Y with K = the whole
// trailing size): the host cuts K into slices that are separate problems of the grouped ZGEMM writing partial results,
// this kernel adds the slices up:  dst = alpha * sum_s src[s] + beta * dst.
struct sb_sum_desc { cplx* dst; const cplx* src; int nslice; int rows, cols, ldd; double alpha, beta; };  // src: nslice x rows x cols
__global__ __launch_bounds__(256) void sb_sum_partials_kernel(const sb_sum_desc* __restrict__ ds) {
  const sb_sum_desc D = ds[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= D.rows * D.cols) return;
  const int r = idx / D.cols, c = idx % D.cols;
  const size_t ss = (size_t)D.rows * D.cols;
  cplx acc = make_double2(0.0, 0.0);
  int t = 0;
  for (; t + 4 <= D.nslice; t += 4) {   // four slices in flight (the sum keeps its order)
    const cplx v0 = dm_ldg(D.src, (size_t)t * ss + idx), v1 = dm_ldg(D.src, (size_t)(t + 1) * ss + idx);
    const cplx v2 = dm_ldg(D.src, (size_t)(t + 2) * ss + idx), v3 = dm_ldg(D.src, (size_t)(t + 3) * ss + idx);
    acc = cadd(cadd(cadd(cadd(acc, v0), v1), v2), v3);
  }
  for (; t < D.nslice; ++t) acc = cadd(acc, dm_ldg(D.src, (size_t)t * ss + idx));
  cplx out = cscale(acc, D.alpha);
  if (D.beta != 0.0) out = cadd(out, cscale(dm_ldg(D.dst, (size_t)r * D.ldd + c), D.beta));
  dm_stg(D.dst, (size_t)r * D.ldd + c, out);
}

// make the 128-aligned diagonal blocks complete (lower part <- conj of the upper part): the products
// Y = A22 X of the first stage read whole diagonal tiles, the her2k updates keep them complete
struct sb_dmat { cplx* A; int lda; int n; };
__global__ __launch_bounds__(256) void sb_diag_tiles_kernel(const sb_dmat* __restrict__ ms) {
  const sb_dmat M = ms[blockIdx.y];
  const int t0 = blockIdx.x * 128;
  if (t0 >= M.n) return;
  for (int idx = threadIdx.x; idx < 128 * 128; idx += 256) {
    const int r = t0 + idx / 128, c = t0 + idx % 128;
    if (r < M.n && c < r) dm_stg(M.A, (size_t)r * M.lda + c, cconj(dm_ldg(M.A, (size_t)c * M.lda + r)));
  }
}

// S = T^H M1 for one matrix per workgroup, M1 given as `nslice` split-K slices (SB x SB each; nslice = 0: M1 itself):
// the slice sum and the 32 x 32 x 32 product were two launches of their own in every panel.
struct sb_s_desc { const cplx* T; int ldt; const cplx* M1; int nslice; cplx* S; };
__global__ __launch_bounds__(256) void sb_s_kernel(const sb_s_desc* __restrict__ ds) {
  const sb_s_desc D = ds[blockIdx.x];
  __shared__ cplx sT[SB][SB + 1];
  __shared__ cplx sM[SB][SB + 1];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < SB * SB; idx += 256) {
    const int r = idx / SB, c = idx % SB;
    sT[r][c] = dm_ldg(D.T, (size_t)r * D.ldt + c);
    cplx m = dm_ldg(D.M1, (size_t)idx);
    int sl = 1;
    for (; sl + 4 <= D.nslice; sl += 4) {   // four slices in flight (the sum keeps its order)
      const cplx v0 = dm_ldg(D.M1, (size_t)sl * SB * SB + idx), v1 = dm_ldg(D.M1, (size_t)(sl + 1) * SB * SB + idx);
      const cplx v2 = dm_ldg(D.M1, (size_t)(sl + 2) * SB * SB + idx), v3 = dm_ldg(D.M1, (size_t)(sl + 3) * SB * SB + idx);
      m = cadd(cadd(cadd(cadd(m, v0), v1), v2), v3);
    }
    for (; sl < D.nslice; ++sl) m = cadd(m, dm_ldg(D.M1, (size_t)sl * SB * SB + idx));
    sM[r][c] = m;
  }
  __syncthreads();
  for (int idx = tid; idx < SB * SB; idx += 256) {
    const int i = idx / SB, j = idx % SB;
    cplx acc = make_double2(0.0, 0.0);
    for (int k = 0; k <= i; ++k) sb_cfma_ca(acc, sT[k][i], sM[k][j]);   // T upper triangular: (T^H)[i][k] = conj(T[k][i]), k <= i
    dm_stg(D.S, (size_t)idx, acc);
  }
}

// look-ahead: snapshot of the rows [i0, i0 + SB) of A at the columns >= i0 + SB (the next panel, as the trailing update of
// the previous panel left it) into Pw; the grouped product that follows applies the current panel's update to the copy
__global__ __launch_bounds__(256) void sb_strip_copy_kernel(const sb_mat* __restrict__ ms, int i0) {
  const sb_mat M = ms[blockIdx.z];
  const int q = blockIdx.y;
  const int c = i0 + SB + blockIdx.x * 256 + threadIdx.x;
  if (c >= M.n || i0 + q >= M.n) return;
  dm_stg(M.Pw, (size_t)q * M.n + c, dm_ldg(M.A, (size_t)(i0 + q) * M.lda + c));
}

// band extraction: AB[c][i] = A_math[c + i][c] = conj(C[c][c + i]), i <= SB; the bulge rows start at zero
// (rows c < nrb: the entries beyond the diagonal block of c come from the look-ahead's R buffer, see sb_mat::Rb)
struct sb_bmat { const cplx* A; int lda; int n; cplx* AB; const cplx* Rb; int nrb; };
__global__ __launch_bounds__(256) void sb_band_extract_kernel(const sb_bmat* __restrict__ ms) {
  const sb_bmat M = ms[blockIdx.y];
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M.n * SLD) return;
  const int c = (int)(idx / SLD), i = (int)(idx % SLD);
  cplx v = make_double2(0.0, 0.0);
  if (i <= SB && c + i < M.n) {
    const int i0 = (c / SB + 1) * SB;
    if (M.Rb && c < M.nrb && c + i >= i0) v = cconj(dm_ldg(M.Rb, (size_t)c * SB + (c + i - i0)));
    else v = cconj(dm_ldg(M.A, (size_t)c * M.lda + c + i));
    if (i == 0) v.y = 0.0;
  }
  dm_stg(M.AB, idx, v);
}

// The back-transformation reads every diamond block (G, j), j < nb(G) = (n - 2 - G SBG) / SB + 1, whole.  The paired
// chase writes complete rows, and a sweep of a full group reaches every block j <= nb(G) - 3 with a full-length vector;
// what it may leave untouched lies in the last two blocks of a group (short or missing vectors at the end of the matrix)
// and in the last group (missing sweeps): those are cleared here, 64 KB per group instead of the whole n^2 array.
__global__ __launch_bounds__(256) void sb_vd_tail_zero_kernel(const struct sb_chase_mat* __restrict__ ms);

// ---- S2: bulge chasing ---------------------------------------------------------------------------------------------
//
// Sweep s, task 0:   x = A[s+1 : s+1+SB, s]  ->  H_0 (zlarfg);  A[s+1, s] = beta = e[s];  D_0 <- H_0^H D_0 H_0
//          task j>0: E_j = A[R_j, R_{j-1}]   (R_j = s + 1 + j SB + [0, SB))
//                    E_j <- E_j H_{j-1};  H_j from the first column of E_j;  E_j <- H_j^H E_j;  D_j <- H_j^H D_j H_j
// Task j of sweep s may start once sweep s - 1 has finished its task j + 1 (the two share one row / column).
//
// Reflector (s, j) is stored where the back-transformation wants it: group G = s / SBG, member i = s % SBG of the
// diamond block (G, j) — a SBG x SBW array whose row i holds the vector at columns [i, i + SB) and zeros elsewhere
// (the window of block (G, j) starts at row G SBG + 1 + j SB of X).
struct sb_chase_mat {
  cplx* AB; int n;
  cplx* Vd;        // diamond blocks: ((G * jb + j) * SBG + i) * SBW
  cplx* tau2;      // (G * jb + j) * SBG + i
  double* d; double* e;
  int jb;          // blocks per sweep group: (n - 2) / SB + 1
  unsigned* prog;  // one progress word per sweep (zero-initialised): tasks finished, SB_DONE at the end
  int* next;       // sweep counter (zero-initialised)
  int* owner;      // XCD that works on this matrix (-1 initially)
};
// Work queues, one per XCD: entries are matrix ids (a large matrix has several entries = several workgroups).
// All workgroups that work on one matrix sit on ONE XCD — its L2 is their coherence point: band data are written
// with plain stores (write-through to L2) and read with sc1 loads (served by L2, never by a stale L1 line), so a
// hand-off between CUs costs an L2 round trip and no cache maintenance.  Nothing is assumed about placement: a
// workgroup reads its XCC_ID, takes entries from the queue of its XCD first and then from the others, and the
// first workgroup to touch a matrix claims it for its XCD (entries of a matrix claimed by another XCD are skipped).
struct sb_chase_ctl { int* qhead; const int* qent; int* err; unsigned long long* dbg; int qoff[9]; };

constexpr unsigned SB_DONE = 0xffffu;
// A wave that polls a progress word gives up after this much WALL time (ticks of the constant 100 MHz counter behind
// wall_clock64: 30 s) — far beyond any chase (the n = 32 576 chase takes 1.5 s in all), so that slow clocks or a
// time-sliced card cannot turn into a failed batch, while a genuine deadlock still drains the grid.
constexpr unsigned long long SB_WAIT_TICKS = 30ull * 100000000ull;
// The hand-off protocol of the chase — band data written with plain stores, read back by another workgroup of the SAME XCD
// with sc1 loads, progress words as agent-scope atomics — relies on the gfx942 / gfx950 cache behaviour: vector stores
// write through to the XCD's L2, sc1 loads are served by that L2 and never by a stale L1 line.  The queue hands a matrix
// to one XCD only (the first workgroup to touch it claims it for its XCC_ID).  Built for gfx950 only:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "dm_sbr_impl.h: the bulge-chase hand-offs assume the gfx942 / gfx950 L2 write-through + sc1 semantics"
#endif

typedef unsigned int sb_u4 __attribute__((ext_vector_type(4)));
// sc1 loads: served by the XCD's L2, bypassing the vector L1 of this CU (element index in units of cplx / dwords)
__device__ __forceinline__ cplx sb_ld(__amdgpu_buffer_rsrc_t rs, unsigned idx) {
  const sb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16u, 0, 16);
  cplx w;
  __builtin_memcpy(&w, &v, 16);
  return w;
}
__device__ __forceinline__ unsigned sb_ld_u32(__amdgpu_buffer_rsrc_t rs, unsigned idx) {
  return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4u, 0, 16);
}

// One sweep of the chase by one wave.  A 32 x 32 block lives in the wave as 4 x 4 values per lane with INTERLEAVED
// ownership: lane (br, bc) = (lane >> 3, lane & 7) holds the rows br + 8 a and the columns bc + 8 b — eight lanes
// with consecutive br read 128 contiguous bytes of a band column, so every load instruction moves whole lines.
__device__ __forceinline__ void sb_chase_sweep(const sb_chase_mat& M, __amdgpu_buffer_rsrc_t rsAB, int* err, int s, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  cplx* AB = M.AB;
  const int G = s / SBG, gi = s % SBG;
  unsigned seen = (s == 0) ? SB_DONE : 0u;  // tasks of sweep s - 1 known to be finished
  cplx vrow[4], vcol[4];  // the current reflector by the rows / the columns of this lane
  cplx tau = make_double2(0.0, 0.0);
  for (int j = 0;; ++j) {
    const int r0 = s + 1 + j * SB;  // first row of R_j
    if (r0 >= n) break;
    const int nr = min(SB, n - r0);
    // ---- wait for sweep s - 1 to have finished task j + 1
    if (seen < (unsigned)(j + 2)) {
      // (an atomic load: a plain one would be hoisted out of the loop; agent scope = sc1 = served by L2)
      int spins = 0;
      bool bail = false;
      const unsigned long long t_wait0 = wall_clock64();
      for (;;) {
        const unsigned v = __hip_atomic_load(M.prog + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= (unsigned)(j + 2)) { seen = v; break; }
        __builtin_amdgcn_s_sleep(1);
        // never in a correct run: every wave leaves instead of hanging the GPU (the host reports the failure).  The
        // limit is WALL time (SB_WAIT_TICKS of the 100 MHz counter), not a poll count: a predecessor that was
        // descheduled for a while (another process on the card, clock throttling) is not a failure
        ++spins;
        if ((spins & 1023) == 0 && wall_clock64() - t_wait0 > SB_WAIT_TICKS) {
          if (lane == 0) atomicCAS(err, 0, 1 + s + (j << 12) + ((int)(v & 0xff) << 20));  // first failure wins
          bail = true;
        }
        if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bail = true;
        if (bail) {
          if (lane == 0) __hip_atomic_store(M.prog + s, 0x40000000u | ((unsigned)j << 8) | (v & 0xffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // keeps the band loads below behind the poll
    }
    bool reflect = true;
    double beta = 0.0;
    if (j == 0) {
      // ---- task 0: reflector from column s
      cplx x[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) x[a] = sb_ld(rsAB, (unsigned)s * SLD + 1u + (unsigned)(br + 8 * a));
      if (lane == 0) M.d[s] = sb_ld(rsAB, (unsigned)s * SLD).x;
      double sq = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (br + 8 * a > 0) sq += cabs2(x[a]);
      cplx t = sb_sum_br(make_double2(sq, 0.0));
      const double xn2 = __shfl(t.x, 0, 64);
      const cplx alpha = sb_from_lane(x[0], 0);
      const trd_refl R = sb_reflector(xn2, alpha);
      tau = R.tau;
      beta = R.beta;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int rr = br + 8 * a;
        cplx v = cmul(x[a], R.scal);
        if (rr == 0) v = make_double2(1.0, 0.0);
        if (rr >= nr) v = make_double2(0.0, 0.0);
        vrow[a] = v;
      }
      if (bc == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int rr = br + 8 * a;
          if (rr < nr) dm_stg(AB, (size_t)s * SLD + 1 + rr, rr == 0 ? make_double2(beta, 0.0) : make_double2(0.0, 0.0));
        }
      }
      if (lane == 0) M.e[s] = beta;
    } else {
      // ---- E <- E H_{j-1}, new reflector from its first column, E <- H_j^H E
      const int c0 = r0 - SB;
      cplx e[4][4];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = c0 + bc + 8 * bb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r = r0 + br + 8 * a;
          e[a][bb] = sb_ld(rsAB, (unsigned)c * SLD + (unsigned)(r - c));
        }
      }
      cplx w[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sb_cfma(acc, e[a][bb], vcol[bb]);
        w[a] = cmul(tau, sb_sum_bc(acc));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sb_cfms_cb(e[a][bb], w[a], vcol[bb]);
      reflect = nr >= 2;
      cplx tauj = make_double2(0.0, 0.0);
      cplx vnew[4];
      if (reflect) {
        double sq = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (br + 8 * a > 0 && br + 8 * a < nr) sq += cabs2(e[a][0]);
        if (bc != 0) sq = 0.0;
        cplx t = sb_sum_br(make_double2(sq, 0.0));
        const double xn2 = __shfl(t.x, 0, 64);
        const cplx alpha = sb_from_lane(e[0][0], 0);
        const trd_refl R = sb_reflector(xn2, alpha);
        tauj = R.tau;
        beta = R.beta;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int rr = br + 8 * a;
          cplx x = sb_from_lane(e[a][0], lane & ~7);  // column 0 of the block lives in the lanes bc == 0
          cplx v = cmul(x, R.scal);
          if (rr == 0) v = make_double2(1.0, 0.0);
          if (rr >= nr) v = make_double2(0.0, 0.0);
          vnew[a] = v;
        }
        cplx y[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          cplx acc = make_double2(0.0, 0.0);
#pragma unroll
          for (int a = 0; a < 4; ++a) sb_cfma_ca(acc, vnew[a], e[a][bb]);
          y[bb] = cmul(cconj(tauj), sb_sum_br(acc));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) sb_cfms(e[a][bb], vnew[a], y[bb]);
        if (bc == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a) e[a][0] = (br + 8 * a == 0) ? make_double2(beta, 0.0) : make_double2(0.0, 0.0);
        }
      }
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = c0 + bc + 8 * bb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r = r0 + br + 8 * a;
          if (r < n) dm_stg(AB, (size_t)c * SLD + (r - c), e[a][bb]);
        }
      }
      tau = tauj;
#pragma unroll
      for (int a = 0; a < 4; ++a) vrow[a] = reflect ? vnew[a] : make_double2(0.0, 0.0);
    }
    if (reflect) {
      // the Hermitian diagonal block D_j (loaded once E_j is on its way out: both blocks at once do not fit the registers
      // of two waves per SIMD)
      cplx d[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = r0 + br + 8 * a;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int c = r0 + bc + 8 * bb;
          d[a][bb] = (r >= c) ? sb_ld(rsAB, (unsigned)c * SLD + (unsigned)(r - c)) : sb_ld(rsAB, (unsigned)r * SLD + (unsigned)(c - r));
        }
      }
      // ---- store the reflector (lanes bc == 0 hold it by rows)
      const size_t blk = (size_t)G * M.jb + j;
      if (bc == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int rr = br + 8 * a;
          if (rr < nr) dm_stg(M.Vd, (blk * SBG + gi) * SBW + gi + rr, vrow[a]);
        }
      }
      if (lane == 0) M.tau2[blk * SBG + gi] = tau;
      // the same vector by columns: lane (br, bc) wants v[bc + 8 b] = vrow[b] of the lanes br' = bc
#pragma unroll
      for (int a = 0; a < 4; ++a) vcol[a] = sb_from_lane(vrow[a], bc * 8);
      // ---- D <- H^H D H on the Hermitian diagonal block (zhetd2's x, w recurrences)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = br + 8 * a;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int c = bc + 8 * bb;
          if (r < c) d[a][bb] = cconj(d[a][bb]);
          if (r == c) d[a][bb].y = 0.0;
        }
      }
      cplx x[4];
      cplx xv = make_double2(0.0, 0.0);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sb_cfma(acc, d[a][bb], vcol[bb]);
        x[a] = cmul(tau, sb_sum_bc(acc));
        // x^H v over the rows of this lane (the same in all lanes that share the rows)
        sb_cfma_ca(xv, x[a], vrow[a]);
      }
      xv = sb_sum_br(xv);
      const cplx al = cmul(make_double2(-0.5 * tau.x, -0.5 * tau.y), xv);
      cplx wv[4], wc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wv[a] = cadd(x[a], cmul(al, vrow[a]));
#pragma unroll
      for (int a = 0; a < 4; ++a) wc[a] = sb_from_lane(wv[a], bc * 8);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = r0 + br + 8 * a;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int c = r0 + bc + 8 * bb;
          if (r < n && c <= r) {
            cplx v = d[a][bb];
            sb_cfms_cb(v, vrow[a], wc[bb]);
            sb_cfms_cb(v, wv[a], vcol[bb]);
            if (r == c) v.y = 0.0;
            dm_stg(AB, (size_t)c * SLD + (r - c), v);
          }
        }
      }
    }
    // ---- publish: task j of sweep s is finished (the stores have reached L2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(M.prog + s, (unsigned)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!reflect) break;
  }
  // ---- sweep finished
  if (s == n - 2 && lane == 0) M.d[n - 1] = sb_ld(rsAB, (unsigned)(n - 1) * SLD).x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(M.prog + s, SB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void sb_chase_kernel(const sb_chase_mat* __restrict__ ms, const sb_chase_ctl ctl) {
  const int lane = threadIdx.x & 63;
  const int xcd = __builtin_amdgcn_s_getreg(6164) & 7;  // hwreg(HW_REG_XCC_ID, 0, 4)
  __shared__ int s_mat;
  for (int qq = 0; qq < 8; ++qq) {
    const int q = (xcd + qq) & 7;
    const int qlen = ctl.qoff[q + 1] - ctl.qoff[q];
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) {
        int m = -1;
        const int ent = atomicAdd(ctl.qhead + q, 1);
        if (ent < qlen) {
          m = ctl.qent[ctl.qoff[q] + ent];
          const int old = atomicCAS(ms[m].owner, -1, xcd);
          if (old != -1 && old != xcd) m = -2;  // claimed by another XCD
        }
        s_mat = m;
      }
      __syncthreads();
      const int m = s_mat;
      if (m == -1) break;
      if (m == -2) continue;
      const sb_chase_mat M = ms[m];
      if (M.n == 1) {
        if (threadIdx.x == 0) M.d[0] = dm_ldg(M.AB, 0).x;
        continue;
      }
      const __amdgpu_buffer_rsrc_t rsAB =
          __builtin_amdgcn_make_buffer_rsrc((void*)M.AB, 0, (int)min((size_t)M.n * SLD * sizeof(cplx), (size_t)0x7fffffff), 0x00020000);
      for (;;) {
        int s = 0;
        if (lane == 0) s = atomicAdd(M.next, 1);
        s = __builtin_amdgcn_readfirstlane(s);
        if (s >= M.n - 1) break;
        if (__hip_atomic_load(ctl.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        const unsigned long long t0 = wall_clock64();
        sb_chase_sweep(M, rsAB, ctl.err, s, lane);
        if (ctl.dbg && lane == 0) { ctl.dbg[2 * s] = t0; ctl.dbg[2 * s + 1] = wall_clock64(); }
      }
    }
  }
}

// ---- S2, two waves per sweep -------------------------------------------------------------------------------------------
//
// Inside a sweep the reflectors form a chain  E_j -> v_j -> E_{j+1}  that never reads the diagonal blocks; D_j only needs
// v_j.  So a sweep is run by a PAIR of waves of one workgroup: the E wave walks the chain and posts each reflector in an
// LDS mailbox, the D wave applies it to D_j (and has loaded D_j while the E wave was still working).  Sweep s + 1 waits for
//   E_j(s+1):  E_{j+1}(s) and D_j(s) finished          D_j(s+1):  E_{j+1}(s) and D_{j+1}(s) finished
// (two progress words per sweep), which makes the lag between consecutive sweeps 2 max(T_E, T_D) instead of 2 (T_E + T_D).
struct sb_pair_box {
  int sweep;     // sweep posted by the E wave (-2: no more work)
  int gen;       // incremented with every post
  int done_gen;  // last generation the D wave has finished
  int seq;       // reflectors posted in the current sweep
  int ack;       // reflectors taken by the D wave
  int pad[3];
  cplx tau[4];
  cplx v[4][SB];
};

__device__ __forceinline__ int sb_lds_ld(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sb_lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// poll a global progress word until it reaches `need` (cached in `seen`); false = gave up (error flag set)
__device__ __forceinline__ bool sb_wait_prog(const unsigned* word, unsigned need, unsigned& seen, int* err, int code, int lane) {
  if (seen >= need) return true;
  int spins = 0;
  const unsigned long long t_wait0 = wall_clock64();
  for (;;) {
    const unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= need) { seen = v; break; }
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 1023) == 0 && wall_clock64() - t_wait0 > SB_WAIT_TICKS) {   // wall time, not a poll count
      if (lane == 0) atomicCAS(err, 0, code);
      return false;
    }
    if ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}
// spin on an LDS word of the pair until pred(value); false = gave up
template <typename P>
__device__ __forceinline__ bool sb_wait_lds(int* word, P pred, int* err, int code, int lane) {
  int spins = 0;
  const unsigned long long t_wait0 = wall_clock64();
  for (;;) {
    if (pred(sb_lds_ld(word))) break;
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 4095) == 0 && wall_clock64() - t_wait0 > SB_WAIT_TICKS) {
      if (lane == 0) atomicCAS(err, 0, code);
      return false;
    }
    if ((spins & 4095) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

// progress words of sweep s: prog[2 s] = E tasks finished, prog[2 s + 1] = D tasks finished
__device__ __forceinline__ bool sb_chase_E(const sb_chase_mat& M, __amdgpu_buffer_rsrc_t rsAB, sb_pair_box* box, int* err, int s, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  cplx* AB = M.AB;
  const int G = s / SBG, gi = s % SBG;
  unsigned seenE = (s == 0) ? SB_DONE : 0u, seenD = seenE;
  cplx vcol[4];
  cplx tau = make_double2(0.0, 0.0);
  for (int j = 0;; ++j) {
    const int r0 = s + 1 + j * SB;
    if (r0 >= n) break;
    const int nr = min(SB, n - r0);
    if (!sb_wait_prog(M.prog + 2 * (s - 1), (unsigned)(j + 2), seenE, err, 1 + s, lane)) return false;
    if (!sb_wait_prog(M.prog + 2 * (s - 1) + 1, (unsigned)(j + 1), seenD, err, 1 + s, lane)) return false;
    bool reflect = true;
    double beta = 0.0;
    cplx vrow[4];
    if (j == 0) {
      cplx x[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) x[a] = sb_ld(rsAB, (unsigned)s * SLD + 1u + (unsigned)(br + 8 * a));
      if (lane == 0) M.d[s] = sb_ld(rsAB, (unsigned)s * SLD).x;
      double sq = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (br + 8 * a > 0) sq += cabs2(x[a]);
      cplx t = sb_sum_br(make_double2(sq, 0.0));
      const double xn2 = __shfl(t.x, 0, 64);
      const cplx alpha = sb_from_lane(x[0], 0);
      const trd_refl R = sb_reflector(xn2, alpha);
      tau = R.tau;
      beta = R.beta;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int rr = br + 8 * a;
        cplx v = cmul(x[a], R.scal);
        if (rr == 0) v = make_double2(1.0, 0.0);
        if (rr >= nr) v = make_double2(0.0, 0.0);
        vrow[a] = v;
      }
      if (bc == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int rr = br + 8 * a;
          if (rr < nr) dm_stg(AB, (size_t)s * SLD + 1 + rr, rr == 0 ? make_double2(beta, 0.0) : make_double2(0.0, 0.0));
        }
      }
      if (lane == 0) M.e[s] = beta;
    } else {
      const int c0 = r0 - SB;
      cplx e[4][4];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = c0 + bc + 8 * bb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r = r0 + br + 8 * a;
          e[a][bb] = sb_ld(rsAB, (unsigned)c * SLD + (unsigned)(r - c));
        }
      }
      cplx w[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sb_cfma(acc, e[a][bb], vcol[bb]);
        w[a] = cmul(tau, sb_sum_bc(acc));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) sb_cfms_cb(e[a][bb], w[a], vcol[bb]);
      reflect = nr >= 2;
      cplx tauj = make_double2(0.0, 0.0);
      if (reflect) {
        double sq = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (br + 8 * a > 0 && br + 8 * a < nr) sq += cabs2(e[a][0]);
        if (bc != 0) sq = 0.0;
        cplx t = sb_sum_br(make_double2(sq, 0.0));
        const double xn2 = __shfl(t.x, 0, 64);
        const cplx alpha = sb_from_lane(e[0][0], 0);
        const trd_refl R = sb_reflector(xn2, alpha);
        tauj = R.tau;
        beta = R.beta;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int rr = br + 8 * a;
          cplx x = sb_from_lane(e[a][0], lane & ~7);
          cplx v = cmul(x, R.scal);
          if (rr == 0) v = make_double2(1.0, 0.0);
          if (rr >= nr) v = make_double2(0.0, 0.0);
          vrow[a] = v;
        }
        cplx y[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          cplx acc = make_double2(0.0, 0.0);
#pragma unroll
          for (int a = 0; a < 4; ++a) sb_cfma_ca(acc, vrow[a], e[a][bb]);
          y[bb] = cmul(cconj(tauj), sb_sum_br(acc));
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) sb_cfms(e[a][bb], vrow[a], y[bb]);
        if (bc == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a) e[a][0] = (br + 8 * a == 0) ? make_double2(beta, 0.0) : make_double2(0.0, 0.0);
        }
      }
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = c0 + bc + 8 * bb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int r = r0 + br + 8 * a;
          if (r < n) dm_stg(AB, (size_t)c * SLD + (r - c), e[a][bb]);
        }
      }
      tau = tauj;
    }
    if (reflect) {
      // ---- post the reflector for the D wave (ring of four), keep it for the back-transformation
      const int slot = j & 3;
      if (j >= 4 && !sb_wait_lds(&box->ack, [&](int a) { return a > j - 4; }, err, 1 + s, lane)) return false;
      if (bc == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) box->v[slot][br + 8 * a] = vrow[a];
      }
      if (lane == 0) box->tau[slot] = tau;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) sb_lds_st(&box->seq, j + 1);
      // (the whole SBW-wide row of the diamond block in one store, zeros outside the vector: nothing has to clear
      // the reflector array beforehand — see sb_vd_tail_zero_kernel for the slots no sweep reaches)
      const size_t blk = (size_t)G * M.jb + j;
      {
        const int o = lane - gi;
        const cplx vv = (o >= 0 && o < SB) ? box->v[slot][o] : make_double2(0.0, 0.0);
        dm_stg(M.Vd, (blk * SBG + gi) * SBW + lane, vv);
      }
      if (lane == 0) M.tau2[blk * SBG + gi] = tau;
      // the vector by columns for the next block: straight from the mailbox (this wave wrote it)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) vcol[bb] = box->v[slot][bc + 8 * bb];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(M.prog + 2 * s, (unsigned)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!reflect) break;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(M.prog + 2 * s, SB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

__device__ __forceinline__ bool sb_chase_D(const sb_chase_mat& M, __amdgpu_buffer_rsrc_t rsAB, sb_pair_box* box, int* err, int s, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  cplx* AB = M.AB;
  unsigned seenE = (s == 0) ? SB_DONE : 0u, seenD = seenE;
  for (int j = 0;; ++j) {
    const int r0 = s + 1 + j * SB;
    if (r0 >= n) break;
    const int nr = min(SB, n - r0);
    if (j > 0 && nr < 2) break;  // the E wave ends the sweep without a reflector
    if (!sb_wait_prog(M.prog + 2 * (s - 1), (unsigned)(j + 2), seenE, err, 1 + s, lane)) return false;
    if (!sb_wait_prog(M.prog + 2 * (s - 1) + 1, (unsigned)(j + 2), seenD, err, 1 + s, lane)) return false;
    // D_j is final as far as sweep s - 1 goes: fetch it while the E wave is still working on v_j
    cplx d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = r0 + br + 8 * a;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = r0 + bc + 8 * bb;
        d[a][bb] = (r >= c) ? sb_ld(rsAB, (unsigned)c * SLD + (unsigned)(r - c)) : sb_ld(rsAB, (unsigned)r * SLD + (unsigned)(c - r));
      }
    }
    if (!sb_wait_lds(&box->seq, [&](int q) { return q > j; }, err, 1 + s, lane)) return false;
    const int slot = j & 3;
    cplx vrow[4], vcol[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      vrow[a] = box->v[slot][br + 8 * a];
      vcol[a] = box->v[slot][bc + 8 * a];
    }
    const cplx tau = box->tau[slot];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the reads above are done before the slot is handed back
    if (lane == 0) sb_lds_st(&box->ack, j + 1);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = br + 8 * a;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = bc + 8 * bb;
        if (r < c) d[a][bb] = cconj(d[a][bb]);
        if (r == c) d[a][bb].y = 0.0;
      }
    }
    cplx x[4];
    cplx xv = make_double2(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      cplx acc = make_double2(0.0, 0.0);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) sb_cfma(acc, d[a][bb], vcol[bb]);
      x[a] = cmul(tau, sb_sum_bc(acc));
      sb_cfma_ca(xv, x[a], vrow[a]);
    }
    xv = sb_sum_br(xv);
    const cplx al = cmul(make_double2(-0.5 * tau.x, -0.5 * tau.y), xv);
    cplx wv[4], wc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) wv[a] = cadd(x[a], cmul(al, vrow[a]));
#pragma unroll
    for (int a = 0; a < 4; ++a) wc[a] = sb_from_lane(wv[a], bc * 8);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = r0 + br + 8 * a;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const int c = r0 + bc + 8 * bb;
        if (r < n && c <= r) {
          cplx v = d[a][bb];
          sb_cfms_cb(v, vrow[a], wc[bb]);
          sb_cfms_cb(v, wv[a], vcol[bb]);
          if (r == c) v.y = 0.0;
          dm_stg(AB, (size_t)c * SLD + (r - c), v);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(M.prog + 2 * s + 1, (unsigned)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s == n - 2 && lane == 0) M.d[n - 1] = sb_ld(rsAB, (unsigned)(n - 1) * SLD).x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_store(M.prog + 2 * s + 1, SB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// NP pairs of waves per workgroup (wave 2 p = E, wave 2 p + 1 = D); queues and ownership as in sb_chase_kernel
template <int NP>
__global__ __launch_bounds__(128 * NP) void sb_chase2_kernel(const sb_chase_mat* __restrict__ ms, const sb_chase_ctl ctl) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave >> 1;
  const bool isE = (wave & 1) == 0;
  const int xcd = __builtin_amdgcn_s_getreg(6164) & 7;  // hwreg(HW_REG_XCC_ID, 0, 4)
  __shared__ int s_mat;
  __shared__ sb_pair_box boxes[NP];
  sb_pair_box* box = &boxes[pair];
  if (threadIdx.x < NP) {
    boxes[threadIdx.x].sweep = -1; boxes[threadIdx.x].gen = 0; boxes[threadIdx.x].done_gen = 0;
    boxes[threadIdx.x].seq = 0; boxes[threadIdx.x].ack = 0;
  }
  int my_gen = 0;
  for (int qq = 0; qq < 8; ++qq) {
    const int q = (xcd + qq) & 7;
    const int qlen = ctl.qoff[q + 1] - ctl.qoff[q];
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) {
        int m = -1;
        const int ent = atomicAdd(ctl.qhead + q, 1);
        if (ent < qlen) {
          m = ctl.qent[ctl.qoff[q] + ent];
          const int old = atomicCAS(ms[m].owner, -1, xcd);
          if (old != -1 && old != xcd) m = -2;
        }
        s_mat = m;
      }
      __syncthreads();
      const int m = s_mat;
      if (m == -1) break;
      if (m == -2) continue;
      const sb_chase_mat M = ms[m];
      if (M.n == 1) {
        if (threadIdx.x == 0) M.d[0] = dm_ldg(M.AB, 0).x;
        continue;
      }
      const __amdgpu_buffer_rsrc_t rsAB =
          __builtin_amdgcn_make_buffer_rsrc((void*)M.AB, 0, (int)min((size_t)M.n * SLD * sizeof(cplx), (size_t)0x7fffffff), 0x00020000);
      if (isE) {
        for (;;) {
          int s = 0;
          if (lane == 0) s = atomicAdd(M.next, 1);
          s = __builtin_amdgcn_readfirstlane(s);
          bool stop = s >= M.n - 1 || __hip_atomic_load(ctl.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
          // the D wave must be through with the previous sweep before the mailbox is reused
          if (!sb_wait_lds(&box->done_gen, [&](int g) { return g == my_gen; }, ctl.err, 9, lane)) stop = true;
          ++my_gen;
          if (lane == 0) {
            box->sweep = stop ? -2 : s;
            sb_lds_st(&box->seq, 0);
            sb_lds_st(&box->ack, 0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) sb_lds_st(&box->gen, my_gen);
          if (stop) break;
          const unsigned long long t0 = ctl.dbg ? wall_clock64() : 0ull;
          const bool okE = sb_chase_E(M, rsAB, box, ctl.err, s, lane);
          if (ctl.dbg && lane == 0) { ctl.dbg[2 * s] = t0; ctl.dbg[2 * s + 1] = wall_clock64(); }
          if (!okE) {
            // failed: make sure the partner is released as well
            ++my_gen;
            sb_wait_lds(&box->done_gen, [&](int g) { return g == my_gen - 1; }, ctl.err, 9, lane);
            if (lane == 0) { box->sweep = -2; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) sb_lds_st(&box->gen, my_gen);
            break;
          }
        }
      } else {
        for (;;) {
          ++my_gen;
          if (!sb_wait_lds(&box->gen, [&](int g) { return g >= my_gen; }, ctl.err, 10, lane)) break;
          const int s = __builtin_amdgcn_readfirstlane(box->sweep);
          if (s == -2) {
            if (lane == 0) sb_lds_st(&box->done_gen, my_gen);
            break;
          }
          const bool ok = sb_chase_D(M, rsAB, box, ctl.err, s, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) sb_lds_st(&box->done_gen, my_gen);
          if (!ok) {
            // take the stop post (if the partner still makes one) so that generations stay in step
            ++my_gen;
            sb_wait_lds(&box->gen, [&](int g) { return g >= my_gen; }, ctl.err, 10, lane);
            if (lane == 0) sb_lds_st(&box->done_gen, my_gen);
            break;
          }
        }
      }
    }
  }
}

// ---- S2 by band position -----------------------------------------------------------------------------------------------
//
// sb_chase2_kernel lets a wave pair own a SWEEP and follow its bulge down the band: every task takes its two 32 x 32 blocks
// from the pair that wrote them one task earlier (store -> L2 -> progress word -> poll -> load: 32 KB per hand-off, 43 GB per
// launch at configs[1] against 1 GB of bands, more than half of the ~5 us per task).  Here a wave pair owns a POSITION j of
// the band for ALL sweeps: E_j = A[R_j, R_{j-1}] lives in the registers of the E wave, D_j = A[R_j, R_j] in those of the D
// wave (R_j(s) = s + 1 + j SB + [0, SB)), and the windows slide by one row and one column per sweep.  The blocks are kept in
// SLOT coordinates — element (r, c) sits in lane (r & 7, c & 7), register [(r >> 3) & 3][(c >> 3) & 3] — so sliding moves
// nothing: the slot of the dropped first row / column (r0_old & 31) is where the new last row / column belongs.  What
// crosses between positions per task (scratch/proto_chase_pos.py is the index-level model of this):
//   left -> right   the reflector v_{j-1}(s) and its tau                           (E_{j-1} -> E_j; E_j -> D_j inside the pair)
//   right -> left   E_{j+1}(s - 1)[0, 0] (beta of its reflector: the corner of E_j(s); the rest of E's new last row lies outside
//                   the band), the first row of E_{j+1}(s - 1) after its task and D_{j+1}(s - 1)[0, 0]: the new last row of D_j(s)
//   D_j -> E_j      the first column of D_j(s - 1): the new last column of E_j(s)
// — about 1 KB per task instead of 32 KB, and nothing is written back to the band array at all (AB is read once, at sweep 0;
// the outputs are d, e and the reflectors).  Everything a neighbour waits for is posted AS SOON AS IT EXISTS: the reflector
// and beta right after the pivot column is known (before the block itself is updated), the first column and corner of D
// right after its w vector — the dependent cycle  E_j(s) -> v -> E_{j+1}(s) -> beta -> E_j(s + 1)  only holds a matrix-vector
// product and the Householder scalars per hop, the rank-1 / rank-2 updates run beside it.
// NP positions share a workgroup and hand over through LDS mailboxes; at a workgroup boundary the same packets travel as
// DATA-TAGGED granules {4 bytes of payload, 4 bytes of tag = sweep + 1} written by sc1 stores and polled by sc1 loads
// (MI355X_MICROARCH.md "handoff-1to1": ~1 us per hop, valid for ANY placement of the two workgroups — no XCD affinity is
// needed, and none is assumed).  Mailbox depths follow from the dependencies (task (s + 1, j) needs (s, j + 1) and
// (s + 1, j - 1)): v, beta and the D column are consumed before their writer can come round again (depth 1); the row packet
// and the D corner can be overtaken by one sweep (depth 2, slot s & 1).
//
// Scheduling: one workgroup per entry (matrix, group of NP positions), entries taken by TICKET (atomic counter) in table
// order — largest matrix first, its groups left to right — so that whatever the dispatch order, the matrix of the lowest
// unfinished ticket has all its groups resident and makes progress (the host keeps matrices that would not fit the chip
// in one piece on sb_chase2_kernel).  Every wait gives up after SB_POS_WAIT_TICKS of wall time and raises the error flag,
// which every other wait polls: the grid always drains.
struct sb_pos_ctl { int* ticket; const int2* ent; int nent; int* err; char* mail; };
constexpr int SB_POS_NP = 4;                   // positions (wave pairs) per workgroup
// bytes of global mailbox per entry: V 1 KB | ROW[2] 2 x 1 KB | DCORN[2] 2 x 64 B | ECORN[2] 2 x 64 B
constexpr size_t SB_POS_MAIL = 4096;
constexpr unsigned SB_GM_V = 0u, SB_GM_ROW = 1024u, SB_GM_DCORN = 3072u, SB_GM_ECORN = 3200u;
constexpr unsigned long long SB_POS_WAIT_TICKS = 10ull * 100000000ull;   // 10 s of the 100 MHz counter

struct sb_pos_box {           // what ONE position publishes inside its workgroup
  cplx v[SB];                 // reflector of the current sweep by row slot (E wave)
  cplx tau;
  cplx ecorn[2];              // E[0, 0] after the task (beta when a reflector was made) (E wave); [s & 1]
  cplx row[2][SB];            // first row of E after the task, by column slot (E wave); [s & 1]
  cplx dcol[SB];              // first column of D after the task, by row slot (D wave)
  double dcorn[2];            // D[0, 0] after the task (D wave); [s & 1]
  int seqV, seqEc[2], seqRow[2], seqDcol, seqDc[2];   // sweep + 1 of the last post
};

// one double per lane as two tagged granules in one 16-byte sc1 store; the reader polls until all 64 lanes carry the tag
__device__ __forceinline__ void sb_g_post(__amdgpu_buffer_rsrc_t rs, unsigned byteoff, double val, unsigned tag) {
  const long long b = __double_as_longlong(val);
  sb_u4 w;
  w.x = (unsigned)(b & 0xffffffffll); w.y = tag; w.z = (unsigned)(b >> 32); w.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(w, rs, byteoff, 0, 16);
}
__device__ __forceinline__ bool sb_g_wait(__amdgpu_buffer_rsrc_t rs, unsigned byteoff, unsigned tag, double& val, int* err, int code, int lane) {
  int spins = 0;
  const unsigned long long t_wait0 = wall_clock64();
  for (;;) {
    const sb_u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, byteoff, 0, 16);
    const bool ok = (w.y == tag) && (w.w == tag);
    if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
      val = __longlong_as_double(((long long)w.z << 32) | (long long)w.x);
      return true;
    }
    ++spins;
    if ((spins & 255) == 0) {
      if (wall_clock64() - t_wait0 > SB_POS_WAIT_TICKS) {
        if (lane == 0) atomicCAS(err, 0, code);
        return false;
      }
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
  }
}
// wait for an LDS sequence word of the workgroup to reach `need`
__device__ __forceinline__ bool sb_pos_wait(int* word, int need, int* err, int code, int lane) {
  int spins = 0;
  const unsigned long long t_wait0 = wall_clock64();
  for (;;) {
    if (sb_lds_ld(word) >= need) break;
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 4095) == 0) {
      if (wall_clock64() - t_wait0 > SB_POS_WAIT_TICKS) {
        if (lane == 0) atomicCAS(err, 0, code);
        return false;
      }
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}
__device__ __forceinline__ void sb_pos_publish(int* word, int v, int lane) {   // after the payload writes of this wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) sb_lds_st(word, v);
}
// Register choice by a WAVE-UNIFORM index: one scalar branch into code with static register numbers (a select chain would
// cost a dozen v_cndmask per value — or, left visible to the optimiser, become ONE load with a dynamic index that sends
// the whole 32 x 32 block of the wave to scratch memory).
__device__ __forceinline__ cplx sb_sel4v(cplx x0, cplx x1, cplx x2, cplx x3, int k) {   // (for the few single values)
  asm("" : "+v"(x0.x), "+v"(x0.y), "+v"(x1.x), "+v"(x1.y), "+v"(x2.x), "+v"(x2.y), "+v"(x3.x), "+v"(x3.y));
  cplx r = x0;
  r.x = (k == 1) ? x1.x : r.x; r.y = (k == 1) ? x1.y : r.y;
  r.x = (k == 2) ? x2.x : r.x; r.y = (k == 2) ? x2.y : r.y;
  r.x = (k == 3) ? x3.x : r.x; r.y = (k == 3) ? x3.y : r.y;
  return r;
}
#define SB_DISPATCH4(k, CALL)            \
  do {                                   \
    switch (k) {                         \
      case 0: { constexpr int K = 0; CALL; } break; \
      case 1: { constexpr int K = 1; CALL; } break; \
      case 2: { constexpr int K = 2; CALL; } break; \
      default: { constexpr int K = 3; CALL; } break; \
    }                                    \
  } while (0)

// Householder scalars and the vector from x (by rows br + 8 a, replicated over the lanes that share the rows), pivot slot o
__device__ __forceinline__ trd_refl sb_pos_reflector(const cplx (&x)[4], int o, int br, int bc, cplx (&vrow)[4]) {
  const int oa = o >> 3, ob = o & 7;
  double sq = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const bool piv = (a == oa) && (br == ob);
    sq += piv ? 0.0 : cabs2(x[a]);
  }
  // x is replicated over the lanes that share the rows: the sum over the eight row groups of a column of lanes IS the sum
  // over all 32 rows, in every lane — the norm and the pivot element (held by the row group br == ob alone) come out of one
  // butterfly, no broadcast through the LDS crossbar
  cplx xo = sb_sel4v(x[0], x[1], x[2], x[3], oa);
  if (br != ob) xo = make_double2(0.0, 0.0);
  const double xn2 = sb_sum_br(make_double2(sq, 0.0)).x;
  const cplx alpha = sb_sum_br(xo);
  const trd_refl R = sb_reflector(xn2, alpha);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    cplx v = cmul(x[a], R.scal);
    if ((a == oa) && (br == ob)) v = make_double2(1.0, 0.0);
    vrow[a] = v;
  }
  return R;
}

// post the reflector of sweep s (LDS: the D wave of this position and the E wave of the next one; the wire when that one
// lives in another workgroup — tau travels in the place of the leading 1) and keep it for the back-transformation
__device__ __forceinline__ void sb_pos_post_v(const sb_chase_mat& M, sb_pos_box* me, const cplx (&vrow)[4], cplx tau, int s, int j, int r0,
                                              bool wire, __amdgpu_buffer_rsrc_t rsM, int lane) {
  const int br = lane >> 3, bc = lane & 7, o = r0 & 31;
  if (bc == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) me->v[br + 8 * a] = vrow[a];
  }
  if (lane == 0) me->tau = tau;
  sb_pos_publish(&me->seqV, s + 1, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (wire) {
    double val = reinterpret_cast<const double*>(me->v)[lane];
    if ((lane >> 1) == o) val = (lane & 1) ? tau.y : tau.x;
    sb_g_post(rsM, SB_GM_V + 16u * lane, val, (unsigned)(s + 1));
  }
}
__device__ __forceinline__ void sb_pos_store_v(const sb_chase_mat& M, const sb_pos_box* me, cplx tau, int s, int j, int r0, int lane) {
  // row gi of the diamond block (G, j), zeros outside the vector
  const int G = s / SBG, gi = s % SBG;
  const size_t blk = (size_t)G * M.jb + j;
  const int k = lane - gi;
  const cplx vv = (k >= 0 && k < SB) ? me->v[(r0 + k) & 31] : make_double2(0.0, 0.0);
  dm_stg(M.Vd, (blk * SBG + gi) * SBW + lane, vv);
  if (lane == 0) M.tau2[blk * SBG + gi] = tau;
}

// corner of the window of position j at sweep s (> 0): E_{j+1}(s - 1)[0, 0], zero when that position did not run
__device__ __forceinline__ bool sb_pos_get_corner(sb_pos_box* right, bool remoteR, __amdgpu_buffer_rsrc_t rsR, int n, int s, int j, cplx& corner,
                                                  int* err, int lane) {
  corner = make_double2(0.0, 0.0);
  if (s + (j + 1) * SB >= n) return true;   // position j + 1 did not run sweep s - 1
  const int sl = (s - 1) & 1;
  if (remoteR) {
    double val;
    if (!sb_g_wait(rsR, SB_GM_ECORN + 64u * sl + 16u * (lane & 1), (unsigned)s, val, err, 100 + s, lane)) return false;
    corner = make_double2(__shfl(val, 0, 64), __shfl(val, 1, 64));
  } else {
    if (!sb_pos_wait(&right->seqEc[sl], s, err, 100 + s, lane)) return false;
    corner = right->ecorn[sl];
  }
  return true;
}
__device__ __forceinline__ void sb_pos_post_corner(sb_pos_box* me, cplx corner, int s, bool remoteL, __amdgpu_buffer_rsrc_t rsM, int lane) {
  const int sl = s & 1;
  if (lane == 0) me->ecorn[sl] = corner;
  sb_pos_publish(&me->seqEc[sl], s + 1, lane);
  if (remoteL && lane < 2) sb_g_post(rsM, SB_GM_ECORN + 64u * sl + 16u * lane, lane ? corner.y : corner.x, (unsigned)(s + 1));
}

// The E wave of position 0: no block to its left — its "E" is column s of the matrix, a vector
__device__ __forceinline__ void sb_pos_E0(const sb_chase_mat& M, sb_pos_box* boxes, bool remoteR, __amdgpu_buffer_rsrc_t rsM, __amdgpu_buffer_rsrc_t rsR,
                                          int* err, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  sb_pos_box* me = boxes + 1;
  sb_pos_box* right = boxes + 2;
  for (int s = 0; s <= n - 2; ++s) {
    const int r0 = s + 1, o = r0 & 31, oo = s & 31;
    cplx x[4];
    if (s == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = 1 + ((br + 8 * a - 1) & 31);
        x[a] = (r < n) ? dm_ldg(M.AB, (size_t)r) : make_double2(0.0, 0.0);
      }
    } else {
      cplx corner;
      if (!sb_pos_get_corner(right, remoteR, rsR, n, s, 0, corner, err, lane)) return;
      if (!sb_pos_wait(&me->seqDcol, s, err, 200 + s, lane)) return;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        x[a] = me->dcol[br + 8 * a];
        if (br + 8 * a == oo) x[a] = corner;
      }
    }
    cplx vrow[4];
    const trd_refl R = sb_pos_reflector(x, o, br, bc, vrow);
    sb_pos_post_v(M, me, vrow, R.tau, s, 0, r0, remoteR && s + 1 + SB < n, rsM, lane);
    if (lane == 0) M.e[s] = R.beta;
    sb_pos_store_v(M, me, R.tau, s, 0, r0, lane);
  }
}

template <int K>
__device__ __forceinline__ void sb_pos_e_insert(cplx (&e)[4][4], const cplx (&dc)[4], cplx corner, bool rowl, bool coll) {
  // slot oo = 8 K + oob: the new last row is zero but for the corner, the new last column is the first column of D
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (rowl) e[K][b] = make_double2(0.0, 0.0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (coll) e[a][K] = dc[a];
  if (rowl && coll) e[K][K] = corner;
}
template <int K>
__device__ __forceinline__ void sb_pos_col(const cplx (&e)[4][4], cplx (&xs)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) xs[a] = e[a][K];
}
template <int K>
__device__ __forceinline__ void sb_pos_row(const cplx (&e)[4][4], cplx (&xs)[4]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) xs[b] = e[K][b];
}

// The E wave of position j >= 1 (local index i in its workgroup).  rsL / rsR: the global mailboxes of the entries to the
// left / right (read side), rsM: of this entry (write side); remoteL / remoteR: that neighbour lives in another workgroup.
__device__ __forceinline__ void sb_pos_E(const sb_chase_mat& M, sb_pos_box* boxes, int i, int j, bool remoteL, bool remoteR,
                                         __amdgpu_buffer_rsrc_t rsL, __amdgpu_buffer_rsrc_t rsM, __amdgpu_buffer_rsrc_t rsR,
                                         int* err, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  sb_pos_box* me = boxes + (i + 1);
  sb_pos_box* left = boxes + i;          // virtual box 0 = the remote left neighbour
  sb_pos_box* right = boxes + (i + 2);   // virtual box NP + 1 = the remote right neighbour
  const int s_last = n - 2 - j * SB;
  cplx e[4][4];
  // ---- sweep 0: the block from the band (rows >= n are zero)
  {
    const int r0 = 1 + j * SB, o = r0 & 31;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = r0 + ((br + 8 * a - o) & 31);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int c = r0 - SB + ((bc + 8 * b - o) & 31);
        e[a][b] = (r < n) ? dm_ldg(M.AB, (size_t)c * SLD + (r - c)) : make_double2(0.0, 0.0);
      }
    }
  }
  for (int s = 0; s <= s_last; ++s) {
    const int r0 = s + 1 + j * SB;
    const int nr = min(SB, n - r0);
    const int o = r0 & 31, oa = o >> 3, ob = o & 7;
    const int oo = (r0 - 1) & 31, ooa = oo >> 3, oob = oo & 7;
    if (s > 0) {
      // ---- slide: the slot oo takes the new last row (zeros + corner) and the new last column (first column of D_j(s - 1))
      cplx corner;
      if (!sb_pos_get_corner(right, remoteR, rsR, n, s, j, corner, err, lane)) return;
      if (!sb_pos_wait(&me->seqDcol, s, err, 200 + s, lane)) return;
      cplx dc[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) dc[a] = me->dcol[br + 8 * a];
      SB_DISPATCH4(ooa, sb_pos_e_insert<K>(e, dc, corner, br == oob, bc == oob));
    }
    // ---- v_{j-1}(s)
    if (remoteL) {
      double val;
      if (!sb_g_wait(rsL, SB_GM_V + 16u * lane, (unsigned)(s + 1), val, err, 300 + s, lane)) return;
      // tau travels in the place of the reflector's leading 1 (slot of the first row of R_{j-1}: o again, SB = 32)
      const double tx = __shfl(val, 2 * o, 64), ty = __shfl(val, 2 * o + 1, 64);
      if ((lane >> 1) == o) val = (lane & 1) ? 0.0 : 1.0;
      reinterpret_cast<double*>(left->v)[lane] = val;
      if (lane == 0) left->tau = make_double2(tx, ty);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      if (!sb_pos_wait(&left->seqV, s + 1, err, 300 + s, lane)) return;
    }
    cplx vcol[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) vcol[b] = left->v[bc + 8 * b];
    const cplx taup = left->tau;
    // ---- w = tau E v_{j-1}: with it the first column of E H_{j-1} is known (v_{j-1} has its 1 in that column's slot)
    cplx w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      cplx acc = make_double2(0.0, 0.0);
#pragma unroll
      for (int b = 0; b < 4; ++b) sb_cfma(acc, e[a][b], vcol[b]);
      w[a] = cmul(taup, sb_sum_bc(acc));
    }
    cplx x[4];
    {
      cplx xs[4];
      SB_DISPATCH4(oa, sb_pos_col<K>(e, xs));
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const cplx t = sb_from_lane(xs[a], (lane & ~7) | ob);
        x[a] = make_double2(t.x - w[a].x, t.y - w[a].y);
      }
    }
    const bool reflect = nr >= 2;
    cplx vrow[4];
    cplx tau = make_double2(0.0, 0.0);
    double beta = 0.0;
    if (reflect) {
      const trd_refl R = sb_pos_reflector(x, o, br, bc, vrow);
      tau = R.tau;
      beta = R.beta;
      // ---- what the neighbours wait for leaves now: v_j to the right (and to the D wave), beta to the left
      sb_pos_post_v(M, me, vrow, tau, s, j, r0, remoteR && s + 1 + (j + 1) * SB < n, rsM, lane);
      sb_pos_post_corner(me, make_double2(beta, 0.0), s, remoteL, rsM, lane);
    } else {
      // one row left: no reflector; the corner is the row's first entry after the right-multiplication
      sb_pos_post_corner(me, sb_from_lane(sb_sel4v(x[0], x[1], x[2], x[3], oa), ob * 8), s, remoteL, rsM, lane);
    }
    // ---- E <- E H_{j-1}
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) sb_cfms_cb(e[a][b], w[a], vcol[b]);
    if (reflect) {
      // ---- E <- H_j^H E; the annihilated column keeps beta on top (its other entries leave the window at the next slide)
      cplx y[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        cplx acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < 4; ++a) sb_cfma_ca(acc, vrow[a], e[a][b]);
        y[b] = cmul(cconj(tau), sb_sum_br(acc));
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) sb_cfms(e[a][b], vrow[a], y[b]);
    }
    // ---- the first row of the block goes left: the new last row of D_{j-1}(s + 1)
    {
      const int sl = s & 1;
      cplx rw[4];
      SB_DISPATCH4(oa, sb_pos_row<K>(e, rw));
      if (br == ob) {
#pragma unroll
        for (int b = 0; b < 4; ++b) me->row[sl][bc + 8 * b] = rw[b];
      }
      sb_pos_publish(&me->seqRow[sl], s + 1, lane);
      if (remoteL) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        sb_g_post(rsM, SB_GM_ROW + 1024u * sl + 16u * lane, reinterpret_cast<const double*>(me->row[sl])[lane], (unsigned)(s + 1));
      }
    }
    if (reflect) sb_pos_store_v(M, me, tau, s, j, r0, lane);
  }
}

template <int K>
__device__ __forceinline__ void sb_pos_d_insert(cplx (&d)[4][4], const cplx (&rw)[4], const cplx (&cl)[4], double dcn, bool rowl, bool coll) {
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (rowl) d[K][b] = rw[b];
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (coll) d[a][K] = cl[a];
  if (rowl && coll) d[K][K] = make_double2(dcn, 0.0);
}
// the first column of D after the rank-2 update, from the lanes that own it (bc == ob): d - v conj(w_o) - w conj(v_o), v_o = 1
template <int K>
__device__ __forceinline__ void sb_pos_d_firstcol(const cplx (&d)[4][4], const cplx (&vrow)[4], const cplx (&wv)[4], cplx wo, cplx (&col)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    cplx t = d[a][K];
    sb_cfms_cb(t, vrow[a], wo);
    t.x -= wv[a].x; t.y -= wv[a].y;
    col[a] = t;
  }
}

// D[0, 0] after the task: d[s + 1] of the tridiagonal for position 0, the new corner of D_{j-1}(s + 1) otherwise
__device__ __forceinline__ void sb_pos_post_dcorn(const sb_chase_mat& M, sb_pos_box* me, double dc, int s, int j, bool remoteL,
                                                  __amdgpu_buffer_rsrc_t rsM, int lane) {
  if (j == 0) {
    if (lane == 0) M.d[s + 1] = dc;
  } else {
    const int sl = s & 1;
    if (lane == 0) me->dcorn[sl] = dc;
    sb_pos_publish(&me->seqDc[sl], s + 1, lane);
    if (remoteL) sb_g_post(rsM, SB_GM_DCORN + 64u * sl, dc, (unsigned)(s + 1));   // (all lanes write the same 16 bytes)
  }
}

// The D wave of position j
__device__ __forceinline__ void sb_pos_D(const sb_chase_mat& M, sb_pos_box* boxes, int i, int j, bool remoteL, bool remoteR,
                                         __amdgpu_buffer_rsrc_t rsM, __amdgpu_buffer_rsrc_t rsR, int* err, int lane) {
  const int n = M.n;
  const int br = lane >> 3, bc = lane & 7;
  sb_pos_box* me = boxes + (i + 1);
  sb_pos_box* right = boxes + (i + 2);
  const int s_last = n - 2 - j * SB;
  cplx d[4][4];
  {
    const int r0 = 1 + j * SB, o = r0 & 31;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int r = r0 + ((br + 8 * a - o) & 31);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int c = r0 + ((bc + 8 * b - o) & 31);
        cplx v = make_double2(0.0, 0.0);
        if (r < n && c < n) {
          v = (r >= c) ? dm_ldg(M.AB, (size_t)c * SLD + (r - c)) : cconj(dm_ldg(M.AB, (size_t)r * SLD + (c - r)));
          if (r == c) v.y = 0.0;
        }
        d[a][b] = v;
      }
    }
    if (j == 0 && lane == 0) M.d[0] = dm_ldg(M.AB, 0).x;
  }
  for (int s = 0; s <= s_last; ++s) {
    const int r0 = s + 1 + j * SB;
    const int nr = min(SB, n - r0);
    const int o = r0 & 31, oa = o >> 3, ob = o & 7;
    const int oo = (r0 - 1) & 31, ooa = oo >> 3, oob = oo & 7;
    if (s > 0) {
      // ---- slide: row / column oo <- first row of E_{j+1}(s - 1) (its corner entry belongs to E_j) and D_{j+1}(s - 1)[0, 0]
      const bool rightOn = s + (j + 1) * SB < n;
      const int sl = (s - 1) & 1;
      double dcn = 0.0;
      cplx rw[4], cl[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) rw[a] = cl[a] = make_double2(0.0, 0.0);
      if (rightOn) {
        if (remoteR) {
          // the row packet off the wire, into the place a local neighbour would have written
          double val;
          if (!sb_g_wait(rsR, SB_GM_ROW + 1024u * sl + 16u * lane, (unsigned)s, val, err, 400 + s, lane)) return;
          reinterpret_cast<double*>(right->row[sl])[lane] = val;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (!sb_g_wait(rsR, SB_GM_DCORN + 64u * sl, (unsigned)s, val, err, 500 + s, lane)) return;
          dcn = val;
        } else {
          if (!sb_pos_wait(&right->seqRow[sl], s, err, 400 + s, lane)) return;
          if (!sb_pos_wait(&right->seqDc[sl], s, err, 500 + s, lane)) return;
          dcn = right->dcorn[sl];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          rw[a] = right->row[sl][bc + 8 * a];          // row oo by columns
          cl[a] = cconj(right->row[sl][br + 8 * a]);   // column oo by rows
        }
      }
      SB_DISPATCH4(ooa, sb_pos_d_insert<K>(d, rw, cl, dcn, br == oob, bc == oob));
    }
    const bool reflect = (j == 0) || nr >= 2;
    if (reflect) {
      if (!sb_pos_wait(&me->seqV, s + 1, err, 600 + s, lane)) return;
      cplx vrow[4], vcol[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        vrow[a] = me->v[br + 8 * a];
        vcol[a] = me->v[bc + 8 * a];
      }
      const cplx tau = me->tau;
      // ---- D <- H^H D H (zhetd2's x, w recurrences on the full Hermitian block)
      cplx wv[4], wc[4];
      {
        cplx x[4];
        cplx xv = make_double2(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          cplx acc = make_double2(0.0, 0.0);
#pragma unroll
          for (int b = 0; b < 4; ++b) sb_cfma(acc, d[a][b], vcol[b]);
          x[a] = cmul(tau, sb_sum_bc(acc));
          sb_cfma_ca(xv, x[a], vrow[a]);
        }
        xv = sb_sum_br(xv);
        const cplx al = cmul(make_double2(-0.5 * tau.x, -0.5 * tau.y), xv);
#pragma unroll
        for (int a = 0; a < 4; ++a) wv[a] = cadd(x[a], cmul(al, vrow[a]));
      }
      // ---- the first column (slot o) and the corner leave now: the new last column of E_j(s + 1), the corner of D_{j-1}(s + 1)
      {
        // w at slot o alone (one exchange instead of four before the posts): the row group br == ob holds it
        cplx wo = sb_sel4v(wv[0], wv[1], wv[2], wv[3], oa);
        if (br != ob) wo = make_double2(0.0, 0.0);
        wo = sb_sum_br(wo);
        cplx col[4];
        SB_DISPATCH4(oa, sb_pos_d_firstcol<K>(d, vrow, wv, wo, col));
        if (bc == ob) {
#pragma unroll
          for (int a = 0; a < 4; ++a) me->dcol[br + 8 * a] = col[a];
        }
        sb_pos_publish(&me->seqDcol, s + 1, lane);
        const double dc = __shfl(sb_sel4v(col[0], col[1], col[2], col[3], oa).x, ob * 9, 64);   // lane (br, bc) = (ob, ob)
        sb_pos_post_dcorn(M, me, dc, s, j, remoteL, rsM, lane);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) wc[a] = sb_from_lane(wv[a], bc * 8);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          sb_cfms_cb(d[a][b], vrow[a], wc[b]);
          sb_cfms_cb(d[a][b], wv[a], vcol[b]);
          if (a == b && br == bc) d[a][b].y = 0.0;
        }
    } else {
      cplx rw[4];
      SB_DISPATCH4(oa, sb_pos_row<K>(d, rw));
      const double dc = __shfl(sb_sel4v(rw[0], rw[1], rw[2], rw[3], oa).x, ob * 9, 64);
      sb_pos_post_dcorn(M, me, dc, s, j, remoteL, rsM, lane);
    }
  }
}

__global__ __launch_bounds__(128 * SB_POS_NP) void sb_chase_pos_kernel(const sb_chase_mat* __restrict__ ms, const sb_pos_ctl ctl) {
  constexpr int NP = SB_POS_NP;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ int s_ticket;
  __shared__ sb_pos_box boxes[NP + 2];
  if (threadIdx.x == 0) s_ticket = atomicAdd(ctl.ticket, 1);
  for (int k = threadIdx.x; k < (int)(sizeof(boxes) / (sizeof(int))); k += blockDim.x) reinterpret_cast<int*>(boxes)[k] = 0;
  __syncthreads();
  const int t = s_ticket;
  if (t >= ctl.nent) return;
  const int2 ent = ctl.ent[t];
  const sb_chase_mat M = ms[ent.x];
  if (M.n == 1) {
    if (threadIdx.x == 0) M.d[0] = dm_ldg(M.AB, 0).x;
    return;
  }
  const int i = wave >> 1, j = ent.y * NP + i;
  if (j >= M.jb) return;
  const bool remoteL = (i == 0) && (j > 0);
  const bool remoteR = (i == NP - 1) && (j + 1 < M.jb);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)(ctl.mail + (size_t)t * SB_POS_MAIL), 0, (int)SB_POS_MAIL, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL =
      __builtin_amdgcn_make_buffer_rsrc((void*)(ctl.mail + (size_t)(remoteL ? t - 1 : t) * SB_POS_MAIL), 0, (int)SB_POS_MAIL, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR =
      __builtin_amdgcn_make_buffer_rsrc((void*)(ctl.mail + (size_t)(remoteR ? t + 1 : t) * SB_POS_MAIL), 0, (int)SB_POS_MAIL, 0x00020000);
  if ((wave & 1) == 0) {
    if (j == 0) sb_pos_E0(M, boxes, remoteR, rsM, rsR, ctl.err, lane);
    else sb_pos_E(M, boxes, i, j, remoteL, remoteR, rsL, rsM, rsR, ctl.err, lane);
  } else {
    sb_pos_D(M, boxes, i, j, remoteL, remoteR, rsM, rsR, ctl.err, lane);
  }
}

__global__ __launch_bounds__(256) void sb_vd_tail_zero_kernel(const sb_chase_mat* __restrict__ ms) {
  const sb_chase_mat M = ms[blockIdx.y];
  const int n = M.n, G = blockIdx.x;
  if (n < 2 || G * SBG > n - 2) return;
  const int nb = (n - 2 - G * SBG) / SB + 1;
  for (int j = max(0, nb - 2); j < nb; ++j) {
    cplx* dst = M.Vd + ((size_t)G * M.jb + j) * SBG * SBW;
    for (int idx = threadIdx.x; idx < SBG * SBW; idx += 256) dm_stg(dst, idx, make_double2(0.0, 0.0));
  }
}

// ---- B2: X <- Q2 X, column slabs in registers ------------------------------------------------------------------------
//
// A wave owns 16 columns of X; the four lanes of a quad share a column and take the window rows w = 4 u + part (any 32 consecutive
// rows hold exactly eight — at most nine counting both ends — rows of each part).  Groups of SBG sweeps are applied last
// to first; inside a group the diamond blocks j = 0, 1, ... in turn (block (G, j) touches the rows [G SBG + 1 + j SB,
// + SBG + SB - 1) of X), inside a block the sweeps last to first.  The window slides down by SB rows per block: the rows
// that leave are final for this group and are written back, SB new rows are loaded.  The reflectors of a block (SBG x SBW
// values, zero outside each vector) are staged in LDS by the workgroup, whose waves all work on the same matrix.
struct sb_q2_mat {
  const cplx* Vd; const cplx* tau2; int jb;
  cplx* X; int ldx; int n; int ncol;   // X is n x ncol (row-major, leading dimension ldx)
  int slab0;                           // first slab (of 16 columns) id of this matrix in the launch
};

// LPC = lanes per column of X: 4 (16 columns per wave, 9 rows of a reflector per lane) or 8 (8 columns per wave, 5 rows per
// lane: twice the waves for the same X — the choice when the columns alone cannot fill the chip)
template <int NW, int LPC>
__global__ __launch_bounds__(64 * NW) void sb_q2_apply_kernel(const sb_q2_mat* __restrict__ ms, const int2* __restrict__ wgs) {
  // wgs[blockIdx.x] = (matrix, first slab of this workgroup)
  const int2 wg = wgs[blockIdx.x];
  const sb_q2_mat M = ms[wg.x];
  const int n = M.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NCW = 64 / LPC;          // columns per wave
  constexpr int NRL = SB / LPC + 1;      // rows of one reflector a lane can meet
  const int col = (wg.y + wave) * NCW + lane / LPC;
  const int part = lane % LPC;
  const bool cvalid = col < M.ncol;
  extern __shared__ __align__(16) unsigned char sb_q2_smem[];
  cplx (*sv)[SBG * SBW] = reinterpret_cast<cplx (*)[SBG * SBW]>(sb_q2_smem);
  cplx (*st)[SBG] = reinterpret_cast<cplx (*)[SBG]>(sb_q2_smem + sizeof(cplx) * 2 * SBG * SBW);
  constexpr int NU = SBW / LPC;  // window rows per lane
  const int nsweep = n - 1;
  if (nsweep <= 0) return;
  const int ngroup = (nsweep + SBG - 1) / SBG;
  for (int G = ngroup - 1; G >= 0; --G) {
    const int s0 = G * SBG;
    const int row0 = s0 + 1;                       // window start of block 0
    const int nb = (n - 1 - row0) / SB + 1;        // blocks with at least one row: row0 + j SB <= n - 1
    cplx xw[NU];
    // stage block 0 and load the first window
    __syncthreads();
    {
      const cplx* src = M.Vd + ((size_t)G * M.jb) * SBG * SBW;
      for (int idx = tid; idx < SBG * SBW; idx += 64 * NW) sv[0][idx] = dm_ldg(src, idx);
      if (tid < SBG) st[0][tid] = dm_ldg(M.tau2, ((size_t)G * M.jb) * SBG + tid);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int r = row0 + LPC * u + part;
      xw[u] = (cvalid && r < n) ? dm_ldg(M.X, (size_t)r * M.ldx + col) : make_double2(0.0, 0.0);
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
      const int buf = j & 1;
      // The next block of reflectors goes straight from global memory into the other LDS buffer (global_load_lds:
      // no registers, nothing waits for it until the end of this block), and the 32 rows of X that enter the window at
      // the slide are requested now as well: a block's worth of arithmetic hides both latencies.
      const int wr0 = row0 + j * SB;
      cplx xn[SB / LPC];
      if (j + 1 < nb) {
        const cplx* src = M.Vd + ((size_t)G * M.jb + j + 1) * SBG * SBW;
#pragma unroll
        for (int k = 0; k < SBG * SBW / (64 * NW); ++k) {
          const int i0 = (k * NW + wave) * 64;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i0 + lane),
                                           (__attribute__((address_space(3))) void*)(&sv[buf ^ 1][i0]), 16, 0, 0);
        }
        if (tid < SBG)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(M.tau2 + ((size_t)G * M.jb + j + 1) * SBG + tid),
                                           (__attribute__((address_space(3))) void*)(&st[buf ^ 1][0]), 16, 0, 0);
#pragma unroll
        for (int u = 0; u < SB / LPC; ++u) {
          const int r = wr0 + SB + LPC * (u + NU - SB / LPC) + part;
          xn[u] = (cvalid && r < n) ? dm_ldg(M.X, (size_t)r * M.ldx + col) : make_double2(0.0, 0.0);
        }
      }
      // Sweeps last to first.  A sweep the group does not have (the last group, or a block below the end of the matrix)
      // has tau = 0 and a zero vector: applying it changes nothing, so the loop is straight-line code and the reflector of
      // the next step is fetched from LDS while the current one is applied.
      // (two register sets for the reflector, used in turn: no copies between the fetch and the use)
      auto fetch = [&](cplx (&v)[NRL], int i) {
#pragma unroll
        for (int t = 0; t < NRL; ++t) v[t] = sv[buf][i * SBW + part + LPC * (i / LPC + t)];
      };
      auto apply = [&](const cplx (&v)[NRL], int i) {
        const int iq = i / LPC;
        const cplx tq = st[buf][i];
        // rows w = LPC u + part, u = iq .. iq + NRL - 1 cover [i, i + SB) (zeros outside the vector); three partial sums,
        // each started by a product (no zeroing of accumulators)
        cplx a[3];
#pragma unroll
        for (int t = 0; t < NRL; ++t) {
          if (t < 3) {
            a[t].x = v[t].x * xw[iq + t].x; a[t].x = fma(v[t].y, xw[iq + t].y, a[t].x);      // conj(v) * x
            a[t].y = v[t].x * xw[iq + t].y; a[t].y = fma(-v[t].y, xw[iq + t].x, a[t].y);
          } else {
            sb_cfma_ca(a[t % 3], v[t], xw[iq + t]);
          }
        }
        cplx acc = cadd(cadd(a[0], a[1]), a[2]);
        acc.x = sb_quad_sum(acc.x);
        acc.y = sb_quad_sum(acc.y);
        if (LPC == 8) {
          acc.x += dm_dpp_f64<0x141>(acc.x);   // row_half_mirror: the other quad of the eight lanes
          acc.y += dm_dpp_f64<0x141>(acc.y);
        }
        const cplx f = cmul(tq, acc);  // H x = x - tau v (v^H x)
#pragma unroll
        for (int t = 0; t < NRL; ++t) sb_cfms(xw[iq + t], v[t], f);
      };
      cplx va[NRL], vb[NRL];
      fetch(va, SBG - 1);
#pragma unroll
      for (int i = SBG - 1; i >= 1; i -= 2) {
        fetch(vb, i - 1);
        apply(va, i);
        if (i >= 2) fetch(va, i - 2);
        apply(vb, i - 1);
      }
      // slide: the first SB rows of the window are final for this group
#pragma unroll
      for (int u = 0; u < SB / LPC; ++u) {
        const int r = wr0 + LPC * u + part;
        if (cvalid && r < n) dm_stg(M.X, (size_t)r * M.ldx + col, xw[u]);
      }
      if (j + 1 < nb) {
#pragma unroll
        for (int u = 0; u + SB / LPC < NU; ++u) xw[u] = xw[u + SB / LPC];
#pragma unroll
        for (int u = NU - SB / LPC; u < NU; ++u) xw[u] = xn[u - (NU - SB / LPC)];
      } else {
        // last block: the rest of the window is final too
#pragma unroll
        for (int u = SB / LPC; u < NU; ++u) {
          const int r = wr0 + LPC * u + part;
          if (cvalid && r < n) dm_stg(M.X, (size_t)r * M.ldx + col, xw[u]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the reflectors of the next block have landed in LDS
      __syncthreads();
    }
  }
}
