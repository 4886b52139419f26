// dm_tridiag_impl.h — body of the batched Hermitian eigensolver, compiled once per panel width
// (DM_TNB) inside the namespace DM_TRD_NS by dm_tridiag.hip.  No include guard on purpose.
namespace DM_TRD_NS {

namespace {

constexpr int TNB = DM_TNB;  // reflectors per panel (her2k runs at K = 2 TNB)
constexpr int KS = 16;   // QL sweeps pipelined per pass in rot_apply
constexpr int PF = 8;    // columns prefetched ahead of the window in rot_apply

struct trd_mat {
  cplx* A; int lda; int n;
  cplx* Vt;      // n x n: row k = Householder vector k (zero for index <= k, 1 at k+1)
  cplx* Vp;      // TNB x n panel of V (row j = vector of panel column j)
  cplx* Wp;      // TNB x n panel of W, stored right behind Vp ...
  cplx* Vp2;     // ... and a second copy of V behind W: [V; W] and [W; V] are both contiguous (her2k at K = 2 TNB)
  cplx* x;       // n: unnormalised Householder column of the current step
  cplx* p;       // n: row part of A v
  cplx* Pc;      // (n / SYG + 1) x n: mirrored (column) parts of A v, one row per SYG-row group
  double* Sp;    // n / SYG + 1: partial sums of v^H A v
  double* Np;    // n / WXR + 1: partial sums of |x|^2
  cplx* ab;      // 2*TNB scratch: panel dot products W^H v, V^H v
  double* d;     // n
  double* e;     // n
  cplx* tau;     // n
};

// ---- T1: one column of the reduction = two launches, both spread over (row tiles x matrices) ----
//
// Only the UPPER triangle of the trailing matrix is kept up to date (her2k writes tiles on or
// above the block diagonal) and the matrix-vector product reads each stored element once:
//
//   trd_symv(k)  every wave derives the Householder scalars (beta, tau, 1/(alpha-beta)) of column k
//                from the partial norms left by trd_wx and forms v on the fly from the unnormalised
//                column x.  A wave owns SYR = 4 consecutive rows r and streams them in 64-column
//                chunks c >= r: the row part  sum_c A[r][c] v[c]  is accumulated per row, the
//                mirrored part  conj(A[r][c]) v[r]  per column, and written as one partial row
//                Pc[group][c] (no atomics: the consumer adds the partial rows in a fixed order).
//                It also writes v, e[k], tau[k], the panel dot products a = W^H v, b = V^H v and the
//                partial sums of v^H A v (which give p^H v without another pass over p).
//   trd_wx(k)    finishes w_k = tau (A v - V a - W b) - (tau/2)(p^H v) v  for its 64 rows and, in
//                the same pass over the panel rows V[:, i], W[:, i], forms the next column
//                x_{k+1} = conj(A[k+1][i]) - V conj(W[k+1]) - W conj(V[k+1])  and its partial norms.
//
// HBM traffic per column: (n-k)^2/2 matrix elements + one pass over the panel (the zlatrd scheme
// reads the full square and the panel twice).
#ifndef DM_SYR
#define DM_SYR 4
#endif
#ifndef DM_SYC
#define DM_SYC 2
#endif
constexpr int SYR = DM_SYR;     // rows per wave in trd_symv
#ifndef DM_SYW
#define DM_SYW 4
#endif
constexpr int SYW = DM_SYW;     // waves per workgroup in trd_symv
constexpr int SYG = SYW * SYR;  // rows per workgroup = rows behind one partial row of Pc
constexpr int SYC = DM_SYC;        // 64-column chunks per loop iteration of trd_symv
constexpr int WXR = 64;    // rows per workgroup in trd_wx

struct trd_refl { cplx tau, scal; double beta; };

// Householder scalars of column k from this lane's share of the partial norms and alpha = x[k+1]
__device__ __forceinline__ trd_refl trd_reflector_from(double npart_lane, cplx alpha) {
  const double xnorm2 = dm_wave_sum(npart_lane);
  trd_refl R;
  if ((xnorm2 == 0.0 && alpha.y == 0.0) || alpha.x * alpha.x + alpha.y * alpha.y + xnorm2 < DM_REFL_TINY) {
    R.tau = make_double2(0.0, 0.0);
    R.beta = alpha.x;
    R.scal = make_double2(0.0, 0.0);
  } else {
    R.beta = -copysign(sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2), alpha.x);
    R.tau = make_double2((R.beta - alpha.x) / R.beta, -alpha.y / R.beta);
    const double dr = alpha.x - R.beta, di = alpha.y;  // 1 / (alpha - beta)
    const double den = dr * dr + di * di;
    R.scal = make_double2(dr / den, -di / den);
  }
  return R;
}

// the partial norms are fetched one per lane and folded with shuffles: a serial scalar loop
// over up to n/64 partials would put that many dependent load latencies in front of every wave
__device__ __forceinline__ double trd_npart_lane(const trd_mat& M, int k) {
  const int np = (M.n - k + WXR - 1) / WXR;
  double s = 0.0;
  for (int t = threadIdx.x & 63; t < np; t += 64) s += dm_ldg(M.Np, t);
  return s;
}

__device__ __forceinline__ trd_refl trd_reflector(const trd_mat& M, int k) {
  const double npart = trd_npart_lane(M, k);
  const cplx alpha = dm_ldg(M.x, k + 1);
  return trd_reflector_from(npart, alpha);
}

// v[c] for c > k (v[k+1] = 1, the rest is the scaled column)
__device__ __forceinline__ cplx trd_v_at(const trd_mat& M, const trd_refl& R, int k, int c) {
  const cplx t = cmul(dm_ldg(M.x, c), R.scal);
  return c == k + 1 ? make_double2(1.0, 0.0) : t;
}

__global__ __launch_bounds__(64 * SYW) void trd_symv_kernel(const trd_mat* __restrict__ ms, int k, int j) {
  const trd_mat M = ms[blockIdx.y];
  const int n = M.n;
  if (k >= n - 1) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: row bases stay in SGPRs
  // the panel dot products take SLV vectors per wave: v is formed once per element and reused
  constexpr int SLV = 4;
  const int nslot = (2 * j + SLV - 1) / SLV;
  const int nslotblk = (nslot + SYW - 1) / SYW;
  if ((int)blockIdx.x >= nslotblk && k + 1 + SYG * ((int)blockIdx.x - nslotblk) >= n) return;  // no rows left
  if ((int)blockIdx.x < nslotblk && (int)blockIdx.x * SYW + wave >= nslot) return;
  if ((int)blockIdx.x < nslotblk) {
    const trd_refl R = trd_reflector(M, k);
    // a[q] = W_q^H v, b[q] = V_q^H v (needed by trd_wx); vector index q < j: W_q, else V_{q-j}
    const int q0 = (blockIdx.x * SYW + wave) * SLV;
    const cplx* xs[SLV];
#pragma unroll
    for (int u = 0; u < SLV; ++u) {
      const int q = min(q0 + u, 2 * j - 1);
      xs[u] = (q < j ? M.Wp + (size_t)q * n : M.Vp + (size_t)(q - j) * n);
    }
    double sr[SLV], si[SLV];
#pragma unroll
    for (int u = 0; u < SLV; ++u) sr[u] = si[u] = 0.0;
    for (int i = k + 1 + lane; i < n; i += 64) {
      const cplx vv = trd_v_at(M, R, k, i);
#pragma unroll
      for (int u = 0; u < SLV; ++u) {
        const cplx xx = dm_ldg(xs[u], i);  // conj(x) * v
        sr[u] += xx.x * vv.x + xx.y * vv.y;
        si[u] += xx.x * vv.y - xx.y * vv.x;
      }
    }
#pragma unroll
    for (int u = 0; u < SLV; ++u) {
      const double tr = dm_wave_sum(sr[u]), ti = dm_wave_sum(si[u]);
      if (lane == 0 && q0 + u < 2 * j) M.ab[q0 + u] = make_double2(tr, ti);
    }
    return;
  }
  // ---- a workgroup owns SYG = 4 SYR consecutive rows; its four waves walk the same 64-column
  // chunks (starting at the group's first row) so that the mirrored column sums of the whole
  // group can be folded through LDS into ONE partial row Pc[g][:]
  const int g = blockIdx.x - nslotblk;
  const int R0 = k + 1 + SYG * g;
  if (R0 >= n) return;
  __shared__ cplx colbuf[2][SYC][SYW][64];
  __shared__ double sbuf[SYW];
  const int rstart = R0 + SYR * wave;
  const int dl = SYR * wave;  // lane of this wave's first diagonal element in chunk 0
  const cplx* __restrict__ A = M.A;
  const size_t lda = M.lda;
  cplx* __restrict__ pc = M.Pc + (size_t)g * n;
  const cplx zero = make_double2(0.0, 0.0);

  // Loads always hit a valid address (indices clamped) and are masked afterwards, so the row
  // loads of a chunk are issued back to back instead of one branch each.
  int c = R0 + lane;
  bool valid = c < n;
  int cc = min(c, n - 1);
  // Every load of the prologue is issued before anything waits: the partial norms, alpha, the
  // column values under this chunk and the SYR matrix rows are independent of each other, only
  // their USE needs the Householder scalars — one exposed memory latency instead of three.
  const double npart = trd_npart_lane(M, k);
  const cplx alpha = dm_ldg(M.x, k + 1);
  const cplx xraw0 = dm_ldg(M.x, cc);
  cplx araw[SYR];
#pragma unroll
  for (int rr = 0; rr < SYR; ++rr) araw[rr] = dm_ldg(A, (size_t)min(rstart + rr, n - 1) * lda + cc);
  const trd_refl R = trd_reflector_from(npart, alpha);
  const cplx vc0t = (cc == k + 1) ? make_double2(1.0, 0.0) : cmul(xraw0, R.scal);
  const cplx vc0 = valid ? vc0t : zero;
  if (wave == 0) {
    if (lane < SYG && valid) {
      M.Vp[(size_t)j * n + c] = vc0;
      M.Vp2[(size_t)j * n + c] = vc0;
      M.Vt[(size_t)k * n + c] = vc0;
    }
    if (g == 0 && lane == 0) {
      M.e[k] = R.beta;
      M.tau[k] = R.tau;
    }
  }
  cplx vr[SYR];
#pragma unroll
  for (int rr = 0; rr < SYR; ++rr) vr[rr] = make_double2(__shfl(vc0.x, dl + rr, 64), __shfl(vc0.y, dl + rr, 64));
  double acc[2 * SYR];
#pragma unroll
  for (int q = 0; q < 2 * SYR; ++q) acc[q] = 0.0;
  double adiag = 0.0;
  int t = 0;
  {
    cplx a[SYR];
#pragma unroll
    for (int rr = 0; rr < SYR; ++rr) {
      const int r = rstart + rr;
      a[rr] = (valid && r < n && c >= r) ? araw[rr] : zero;
    }
    cplx col = zero;
#pragma unroll
    for (int rr = 0; rr < SYR; ++rr) {
      if (lane == dl + rr) adiag = a[rr].x;
      acc[2 * rr] += a[rr].x * vc0.x - a[rr].y * vc0.y;
      acc[2 * rr + 1] += a[rr].x * vc0.y + a[rr].y * vc0.x;
      if (lane > dl + rr) {  // strictly above the diagonal: mirrored contribution conj(a) * v[r]
        col.x += a[rr].x * vr[rr].x + a[rr].y * vr[rr].y;
        col.y += a[rr].x * vr[rr].y - a[rr].y * vr[rr].x;
      }
    }
    colbuf[0][0][wave][lane] = col;
    __syncthreads();
    if (wave == 0 && valid) {
      cplx tsum = colbuf[0][0][0][lane];
#pragma unroll
      for (int w = 1; w < SYW; ++w) tsum = cadd(tsum, colbuf[0][0][w][lane]);
      dm_stg(pc, c, tsum);
    }
    t = 1;
  }
  // main loop: SYC chunks (64 SYC columns) per iteration -> SYC * SYR row loads in flight per lane;
  // HBM latency under load is ~5 us, so the bytes in flight per CU set the streaming rate
#pragma unroll 1
  for (int c0 = R0 + 64; c0 < n; c0 += 64 * SYC, ++t) {
    cplx a[SYC][SYR], vcu[SYC];
    bool vld[SYC];
#pragma unroll
    for (int u = 0; u < SYC; ++u) {
      const int cu = c0 + 64 * u + lane;
      vld[u] = cu < n;
      const int ccu = min(cu, n - 1);
      const cplx vct = trd_v_at(M, R, k, ccu);
      vcu[u] = vld[u] ? vct : zero;
#pragma unroll
      for (int rr = 0; rr < SYR; ++rr) {
        const int r = rstart + rr;
        const cplx v = dm_ldg(A, (size_t)min(r, n - 1) * lda + ccu);
        a[u][rr] = (vld[u] && r < n) ? v : zero;
      }
    }
    const int pb = t & 1;  // double buffer: the reader of iteration t-1 may still be summing
#pragma unroll
    for (int u = 0; u < SYC; ++u) {
      cplx col = zero;
#pragma unroll
      for (int rr = 0; rr < SYR; ++rr) {
        acc[2 * rr] += a[u][rr].x * vcu[u].x - a[u][rr].y * vcu[u].y;
        acc[2 * rr + 1] += a[u][rr].x * vcu[u].y + a[u][rr].y * vcu[u].x;
        col.x += a[u][rr].x * vr[rr].x + a[u][rr].y * vr[rr].y;
        col.y += a[u][rr].x * vr[rr].y - a[u][rr].y * vr[rr].x;
      }
      colbuf[pb][u][wave][lane] = col;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SYC; ++u)
      if (wave == ((t * SYC + u) & (SYW - 1)) && vld[u]) {
        cplx tsum = colbuf[pb][u][0][lane];
#pragma unroll
        for (int w = 1; w < SYW; ++w) tsum = cadd(tsum, colbuf[pb][u][w][lane]);
        dm_stg(pc, c0 + 64 * u + lane, tsum);
      }
  }
  // Transposing butterfly: the 2 SYR per-lane partial sums are folded so that lane L ends up with
  // the wave total of entry L / PER (2 SYR + log2(PER) shuffles instead of 2 SYR full reductions).
  constexpr int PER = 64 / (2 * SYR);
#pragma unroll
  for (int o = 32, cnt = 2 * SYR; cnt > 1; o >>= 1, cnt >>= 1) {
    const bool lo = (lane & o) == 0;
    const int h = cnt >> 1;
#pragma unroll
    for (int q = 0; q < h; ++q) {
      const double send = lo ? acc[q + h] : acc[q];
      const double keep = lo ? acc[q] : acc[q + h];
      acc[q] = keep + __shfl_xor(send, o, 64);
    }
  }
  double tot = acc[0];
#pragma unroll
  for (int o = PER / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  const double tim = __shfl_down(tot, PER, 64);  // lane 2 PER rr: tot = re, tim = im of row rr
  const int rr = lane / (2 * PER);
  const double vrx = __shfl(vc0.x, dl + rr, 64), vry = __shfl(vc0.y, dl + rr, 64), ad = __shfl(adiag, dl + rr, 64);
  double s = 0.0;
  if (lane % (2 * PER) == 0 && rstart + rr < n) {
    M.p[rstart + rr] = make_double2(tot, tim);
    s = 2.0 * (vrx * tot + vry * tim) - ad * (vrx * vrx + vry * vry);  // v^H A v, upper storage
  }
  s = dm_wave_sum(s);
  if (lane == 0) sbuf[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tsum = 0.0;
#pragma unroll
    for (int w = 0; w < SYW; ++w) tsum += sbuf[w];
    M.Sp[g] = tsum;
  }
}

__global__ __launch_bounds__(256) void trd_wx_kernel(const trd_mat* __restrict__ ms, int k, int j, int do_w, int do_x) {
  const trd_mat M = ms[blockIdx.y];
  const int n = M.n;
  const bool w_on = do_w && k < n - 1;
  const int kx = do_w ? k + 1 : k;  // column whose x is formed; also the first row handled
  const bool x_on = do_x && kx < n;
  if (!w_on && !x_on) return;
  if (kx + (int)blockIdx.x * WXR >= n) return;
  // 64 rows per workgroup, four waves per row block: wave q takes the panel vectors and the
  // partial rows of A v with index = q (mod 4); wave 0 folds the four partial results and finishes.
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = kx + blockIdx.x * WXR + lane;
  const int npan = do_w ? j : 0;  // finished panel vectors
  __shared__ cplx sa[TNB], sb[TNB], swk[TNB], svk[TNB];
  __shared__ cplx qbuf[3][WXR], xbuf[3][WXR];
  // wave 0 finishes the rows: everything it will need from global memory that does not depend on
  // the other waves is requested now, so that its tail pays no further memory latency
  cplx e_tau = make_double2(0.0, 0.0), e_vv = make_double2(0.0, 0.0), e_a = make_double2(0.0, 0.0),
       e_pkx = make_double2(0.0, 0.0);
  double e_s = 0.0;
  if (wave == 0) {
    if (w_on) {
      e_tau = dm_ldg(M.tau, k);
      const int ng = (n - k - 1 + SYG - 1) / SYG;
      for (int t = lane; t < ng; t += 64) e_s += dm_ldg(M.Sp, t);
      if (i < n) e_vv = dm_ldg(M.Vp, (size_t)j * n + i);
      if (x_on) e_pkx = dm_ldg(M.p, kx);
    }
    if (x_on && i < n) e_a = dm_ldg(M.A, (size_t)kx * M.lda + i);
  }
  if (tid < npan) {
    sa[tid] = M.ab[tid];
    sb[tid] = M.ab[j + tid];
    if (x_on) {
      swk[tid] = M.Wp[(size_t)tid * n + kx];
      svk[tid] = M.Vp[(size_t)tid * n + kx];
    }
  }
  __syncthreads();
  cplx q = make_double2(0.0, 0.0), xacc = make_double2(0.0, 0.0);
  if (i < n) {
    if (w_on) {
      if (wave == 0) q = M.p[i];
      const int gi = (i - k - 1) / SYG;
      const cplx* __restrict__ pc = M.Pc + i;
#pragma unroll 4
      for (int g = wave; g <= gi; g += 4) q = cadd(q, dm_ldg(pc, (size_t)g * n));
    }
    const cplx* __restrict__ vp = M.Vp + i;
    const cplx* __restrict__ wp = M.Wp + i;
#pragma unroll 4
    for (int jj = wave; jj < npan; jj += 4) {
      const cplx vji = dm_ldg(vp, (size_t)jj * n), wji = dm_ldg(wp, (size_t)jj * n);
      q = csub(q, cadd(cmul(vji, sa[jj]), cmul(wji, sb[jj])));
      if (x_on) xacc = csub(xacc, cadd(cmulc(vji, swk[jj]), cmulc(wji, svk[jj])));
    }
  }
  if (wave > 0) {
    qbuf[wave - 1][lane] = q;
    xbuf[wave - 1][lane] = xacc;
  }
  __syncthreads();
  if (wave > 0) return;
  q = cadd(cadd(q, qbuf[0][lane]), cadd(qbuf[1][lane], qbuf[2][lane]));
  xacc = cadd(cadd(xacc, xbuf[0][lane]), cadd(xbuf[1][lane], xbuf[2][lane]));
  cplx tau = make_double2(0.0, 0.0), wk1 = make_double2(0.0, 0.0);
  double coef = 0.0;
  if (w_on) {
    tau = e_tau;
    double s = e_s, ab = 0.0, cr = 0.0, ci = 0.0;
    for (int t = lane; t < npan; t += 64) {
      ab += sa[t].x * sb[t].x + sa[t].y * sb[t].y;  // Re(conj(a) b)
      if (x_on) {
        const cplx u = cadd(cmul(svk[t], sa[t]), cmul(swk[t], sb[t]));
        cr += u.x;
        ci += u.y;
      }
    }
    s = dm_wave_sum(s);
    ab = dm_wave_sum(ab);
    cr = dm_wave_sum(cr);
    ci = dm_wave_sum(ci);
    // p^H v = conj(tau) (v^H A v - a^H b - b^H a)  (real);  coef = (tau/2) p^H v
    coef = 0.5 * (tau.x * tau.x + tau.y * tau.y) * (s - 2.0 * ab);
    if (x_on) {  // w_k[k+1]: the mirrored part of p vanishes on the first trailing row
      const cplx q1 = csub(e_pkx, make_double2(cr, ci));
      wk1 = cmul(tau, q1);
      wk1.x -= coef;
    }
  }
  double part = 0.0;
  if (i < n) {
    cplx wv = make_double2(0.0, 0.0), vv = make_double2(0.0, 0.0);
    if (w_on) {
      vv = e_vv;
      wv = cmul(tau, q);
      wv.x -= coef * vv.x;
      wv.y -= coef * vv.y;
      M.Wp[(size_t)j * n + i] = wv;
    }
    if (x_on) {
      const cplx a = e_a;
      cplx xx = make_double2(a.x + xacc.x, -a.y + xacc.y);
      if (w_on) xx = csub(xx, cadd(cmulc(vv, wk1), wv));  // panel vector j: V[j][kx] = 1
      M.x[i] = xx;
      if (i == kx) M.d[kx] = xx.x;
      if (i > kx + 1) part = cabs2(xx);
    }
  }
  if (x_on) {
    part = dm_wave_sum(part);
    if (lane == 0) M.Np[blockIdx.x] = part;
  }
}

// ---- T1 + Q for small matrices (n <= TSM): one launch, one workgroup per matrix, the matrix
// resident in LDS (96 x 97 complex = 146 KB of the 160 KB).  Same recurrences and conventions as
// the panel path with a panel of one vector (zhetd2); the reflectors stay in the dead columns of
// the LDS copy and the unitary Q = H_0 ... H_{n-2} is then accumulated in place (zung2r order)
// and written out, so the back-transformation of these problems is a single product X = Q Z.
// The Gram-matrix eigenproblems of the SVD preconditioner (n <= ntel, thousands per launch) would
// otherwise pay 2 n latency-bound launches plus the whole compact-WY machinery for a few hundred
// KB of work each.
constexpr int TSM = 96;
constexpr int TSP = TSM + 1;  // row pitch in complex elements: conflict-free column walks
constexpr int TST = 512;      // threads

struct trs_mat { const cplx* A; int lda; int n; cplx* Q; int ldq; double* d; double* e; };

__global__ __launch_bounds__(TST) void trd_small_kernel(const trs_mat* __restrict__ ms) {
  const trs_mat M = ms[blockIdx.x];
  const int n = M.n;
  if (n <= 0) return;
  extern __shared__ __align__(16) unsigned char trd_smem[];
  cplx* As = reinterpret_cast<cplx*>(trd_smem);          // TSM x TSP
  cplx* vs = As + TSM * TSP;                              // TSM
  cplx* ws = vs + TSM;                                    // TSM
  cplx* ph = ws + TSM;                                    // 4 x TSM partial matvec
  cplx* taus = ph + 4 * TSM;                              // TSM
  double* red = reinterpret_cast<double*>(taus + TSM);    // 3 x NW
  constexpr int NW = TST / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the upper triangle is the reference (as in the panel path); mirror it
  for (int idx = tid; idx < n * n; idx += TST) {
    const int r = idx / n, c = idx - r * n;
    if (c >= r) {
      const cplx a = M.A[(size_t)r * M.lda + c];
      As[r * TSP + c] = (c == r) ? make_double2(a.x, 0.0) : a;
      if (c > r) As[c * TSP + r] = make_double2(a.x, -a.y);
    }
  }
  if (tid < n) taus[tid] = make_double2(0.0, 0.0);
  __syncthreads();
  const int r2 = tid % TSM, part4 = tid / TSM;  // matvec: four threads per row (tid < 4 TSM)
  for (int k = 0; k < n - 1; ++k) {
    // --- Householder vector of column k: x_i = conj(A[k][i]), i > k
    cplx xi = make_double2(0.0, 0.0);
    double part = 0.0;
    if (tid < n && tid > k) {
      const cplx a = As[k * TSP + tid];
      xi = make_double2(a.x, -a.y);
      if (tid > k + 1) part = cabs2(xi);
    }
    part = dm_wave_sum(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    double xnorm2 = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) xnorm2 += red[w];
    const cplx al = As[k * TSP + k + 1];
    const cplx alpha = make_double2(al.x, -al.y);
    double beta;
    cplx tau, scal;
    if ((xnorm2 == 0.0 && alpha.y == 0.0) || alpha.x * alpha.x + alpha.y * alpha.y + xnorm2 < DM_REFL_TINY) {
      tau = make_double2(0.0, 0.0);
      beta = alpha.x;
      scal = make_double2(0.0, 0.0);
    } else {
      beta = -copysign(sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2), alpha.x);
      tau = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
      const double dr = alpha.x - beta, di = alpha.y;
      const double den = dr * dr + di * di;
      scal = make_double2(dr / den, -di / den);
    }
    cplx vi = make_double2(0.0, 0.0);
    if (tid < n) {
      if (tid == k + 1) vi = make_double2(1.0, 0.0);
      else if (tid > k + 1) vi = cmul(xi, scal);
      vs[tid] = vi;
    }
    if (tid == 0) {
      M.d[k] = As[k * TSP + k].x;
      M.e[k] = beta;
      taus[k] = tau;
    }
    __syncthreads();
    // --- p = A v over the trailing block (four quarter-rows per row)
    if (tid < 4 * TSM && r2 < n && r2 > k) {
      const int len = n - (k + 1);
      const int h0 = k + 1 + (len * part4) / 4, h1 = k + 1 + (len * (part4 + 1)) / 4;
      double pr = 0.0, pi = 0.0;
      const cplx* arow = As + r2 * TSP;
      for (int c = h0; c < h1; ++c) {
        const cplx a = arow[c], v = vs[c];
        pr += a.x * v.x - a.y * v.y;
        pi += a.x * v.y + a.y * v.x;
      }
      ph[part4 * TSM + r2] = make_double2(pr, pi);
    }
    __syncthreads();
    cplx pt = make_double2(0.0, 0.0);
    double dr = 0.0, di = 0.0;
    if (tid < n && tid > k) {
      pt = cmul(tau, cadd(cadd(ph[tid], ph[TSM + tid]), cadd(ph[2 * TSM + tid], ph[3 * TSM + tid])));
      dr = pt.x * vi.x + pt.y * vi.y;  // conj(p) * v
      di = pt.x * vi.y - pt.y * vi.x;
    }
    dr = dm_wave_sum(dr);
    di = dm_wave_sum(di);
    if (lane == 0) { red[NW + wave] = dr; red[2 * NW + wave] = di; }
    __syncthreads();
    double dre = 0.0, dim = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { dre += red[NW + w]; dim += red[2 * NW + w]; }
    const cplx coef = cscale(cmul(tau, make_double2(dre, dim)), 0.5);
    if (tid < n) ws[tid] = tid > k ? csub(pt, cmul(coef, vi)) : make_double2(0.0, 0.0);
    __syncthreads();
    // --- A -= v w^H + w v^H on the trailing block (full storage keeps the matvec simple);
    //     column k below the subdiagonal is dead from here on and keeps v_k for the Q accumulation
    {
      const int tx = tid & 31, ty = tid >> 5;
      for (int i = k + 1 + ty; i < n; i += TST / 32) {
        const cplx v_i = vs[i], w_i = ws[i];
        cplx* arow = As + i * TSP;
        for (int c = k + 1 + tx; c < n; c += 32) {
          const cplx u = cadd(cmulc(v_i, ws[c]), cmulc(w_i, vs[c]));
          cplx a = arow[c];
          a.x -= u.x;
          a.y -= u.y;
          arow[c] = a;
        }
      }
      if (tid < n && tid > k + 1) As[tid * TSP + k] = vi;
    }
    __syncthreads();
  }
  if (tid == 0) M.d[n - 1] = As[(n - 1) * TSP + n - 1].x;
  // ---- Q = H_0 ... H_{n-2} in place (reflector i: 1 at row i+1, As[r][i] for r >= i+2)
  // step i (descending): apply H_i to the finished columns c >= i+2 (rows >= i+1), then form column i+1
  const int csub4 = tid & 3, ccol = tid >> 2;  // four threads per column
  for (int i = n - 2; i >= 0; --i) {
    const cplx tau = taus[i];
    const int c = i + 2 + ccol;
    if (c < n) {
      double sr = 0.0, si = 0.0;
      for (int r = i + 1 + csub4; r < n; r += 4) {
        const cplx v = (r == i + 1) ? make_double2(1.0, 0.0) : As[r * TSP + i];
        const cplx a = As[r * TSP + c];  // conj(v) * a
        sr += v.x * a.x + v.y * a.y;
        si += v.x * a.y - v.y * a.x;
      }
      sr += __shfl_xor(sr, 1, 64); si += __shfl_xor(si, 1, 64);
      sr += __shfl_xor(sr, 2, 64); si += __shfl_xor(si, 2, 64);
      const cplx ts = cmul(tau, make_double2(sr, si));
      for (int r = i + 1 + csub4; r < n; r += 4) {
        const cplx v = (r == i + 1) ? make_double2(1.0, 0.0) : As[r * TSP + i];
        As[r * TSP + c] = csub(As[r * TSP + c], cmul(v, ts));
      }
    }
    __syncthreads();
    if (tid < n) {
      cplx q;
      if (tid <= i) q = make_double2(0.0, 0.0);
      else if (tid == i + 1) q = make_double2(1.0 - tau.x, -tau.y);
      else { const cplx v = As[tid * TSP + i]; q = cmul(make_double2(-tau.x, -tau.y), v); }
      As[tid * TSP + i + 1] = q;
    }
    __syncthreads();
  }
  if (tid < n) As[tid * TSP] = make_double2(tid == 0 ? 1.0 : 0.0, 0.0);
  __syncthreads();
  for (int idx = tid; idx < n * n; idx += TST) {
    const int r = idx / n, c = idx - r * n;
    M.Q[(size_t)r * M.ldq + c] = As[r * TSP + c];
  }
}

// ---- T2: implicit QL/QR on the tridiagonal (LAPACK dsteqr scheme), recording rotations ------
// A recorded sweep is a run of plane rotations on the columns of Z with dlasr semantics
//     t = z[j+1];  z[j+1] = c t - s z[j];  z[j] = s t + c z[j]
// applied for j descending from lo+cnt-1 to lo (dir 0, QL) or ascending (dir 1, QR).
struct ql_mat {
  double* d; double* e; int n;
  int* sw_dir; int* sw_lo; int* sw_cnt;
  long long* sw_off;  // offset of plane `lo` in rot
  double2* rot;       // (c, s)
  int max_sweeps; long long max_rot;
  int* nsweeps;       // out
  int* status;        // out: 0 ok, 1 no convergence, 2 storage exhausted
  double* Zt = nullptr;  // APPLY instantiation: eigenvectors out, Zt[col * ldz + row]
  int ldz = 0;
};

__device__ __forceinline__ void dev_lartg(double f, double g, double& c, double& s, double& r) {
  if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
  else if (f == 0.0) { c = 0.0; s = 1.0; r = g; }
  else {
    const double h = f * f + g * g;
    // The matrix is scaled to unit max-norm, so h cannot overflow; when the squares underflow
    // fall back to the safe path.  1/sqrt(h) from the hardware estimate plus two Newton steps
    // (error ~ 1 ulp) replaces a sqrt and a division on the serial critical path.
    double dnorm, inv;
    if (h > 1e-290) {
      double y = __builtin_amdgcn_rsq(h);
      y = y * (1.5 - 0.5 * h * y * y);
      y = y * (1.5 - 0.5 * h * y * y);
      inv = y;
      dnorm = h * y;
      // one correction step on dnorm so that dnorm^2 = h to working accuracy
      dnorm = dnorm + 0.5 * y * (h - dnorm * dnorm);
    } else {
      dnorm = hypot(f, g);
      inv = 1.0 / dnorm;
    }
    c = fabs(f) * inv;
    r = copysign(dnorm, f);
    s = g * copysign(inv, f);
  }
}

// eigen-decomposition of [[a, b], [b, c]] (LAPACK dlaev2)
__device__ void dev_laev2(double a, double b, double c, double& rt1, double& rt2, double& cs1, double& sn1) {
  const double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
  double acmx, acmn;
  if (fabs(a) > fabs(c)) { acmx = a; acmn = c; } else { acmx = c; acmn = a; }
  double rt;
  if (adf > ab) { const double q = ab / adf; rt = adf * sqrt(1.0 + q * q); }
  else if (adf < ab) { const double q = adf / ab; rt = ab * sqrt(1.0 + q * q); }
  else rt = ab * sqrt(2.0);
  int sgn1;
  if (sm < 0.0) { rt1 = 0.5 * (sm - rt); sgn1 = -1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
  else if (sm > 0.0) { rt1 = 0.5 * (sm + rt); sgn1 = 1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
  else { rt1 = 0.5 * rt; rt2 = -0.5 * rt; sgn1 = 1; }
  int sgn2;
  double cs;
  if (df >= 0.0) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
  if (fabs(cs) > ab) { const double ct = -tb / cs; sn1 = 1.0 / sqrt(1.0 + ct * ct); cs1 = ct * sn1; }
  else if (ab == 0.0) { cs1 = 1.0; sn1 = 0.0; }
  else { const double tn = -cs / tb; cs1 = 1.0 / sqrt(1.0 + tn * tn); sn1 = tn * cs1; }
  if (sgn1 == sgn2) { const double tn = cs1; cs1 = -sn1; sn1 = tn; }
}

// APPLY (the leaves of the divide & conquer, n <= 64): the rotations are not recorded but applied at once to Z = I
// held in LDS — lane r owns row r of Z, all lanes run the (uniform) scalar recurrences, and a rotation costs two LDS
// reads and writes per lane off the critical path of the next lartg.  This replaces the record / zt_identity /
// rot_apply sequence, whose rot_apply ran one thread per ROW of a 23..32-row leaf.
template <bool USE_LDS, bool APPLY = false>
__global__ __launch_bounds__(64) void ql_kernel(const ql_mat* __restrict__ qs) {
  extern __shared__ __align__(16) unsigned char ql_smem[];
  const ql_mat Q = qs[blockIdx.x];
  const int n = Q.n;
  double* d = Q.d;
  double* e = Q.e;
  double* Zs = nullptr;
  const int lane = threadIdx.x;
  if (USE_LDS) {
    // the serial chain below touches d and e at every rotation: keep them in LDS
    double* ld = reinterpret_cast<double*>(ql_smem);
    double* le = ld + n;
    for (int i = threadIdx.x; i < n; i += 64) { ld[i] = Q.d[i]; le[i] = (i + 1 < n) ? Q.e[i] : 0.0; }
    if (APPLY) {
      Zs = le + n;  // column-major: Zs[c * n + r]
      if (lane < n)
        for (int c = 0; c < n; ++c) Zs[c * n + lane] = (c == lane) ? 1.0 : 0.0;
    }
    __syncthreads();
    d = ld;
    e = le;
  }
  if (!APPLY && threadIdx.x != 0) return;
  int ns = 0;
  long long nr = 0;
  int status = 0;
  const double eps = 1.1102230246251565e-16;  // dlamch('E')
  const double eps2 = eps * eps;
  const double safmin = 2.2250738585072014e-308;
  auto record = [&](int dir, int lo, int cnt) -> bool {
    if (APPLY) { ++ns; return true; }
    if (ns >= Q.max_sweeps || nr + cnt > Q.max_rot) { status = 2; return false; }
    Q.sw_dir[ns] = dir; Q.sw_lo[ns] = lo; Q.sw_cnt[ns] = cnt; Q.sw_off[ns] = nr;
    ++ns;
    nr += cnt;
    return true;
  };
  // plane rotation of the columns (j, j + 1) of Z:  t = z[j+1];  z[j+1] = c t - s z[j];  z[j] = s t + c z[j]
  auto rotate = [&](int j, double c, double s) {
    if (lane < n) {
      const double zj = Zs[j * n + lane], zj1 = Zs[(j + 1) * n + lane];
      Zs[(j + 1) * n + lane] = c * zj1 - s * zj;
      Zs[j * n + lane] = s * zj1 + c * zj;
    }
  };
  if (n > 1) {
    // global scaling to unit max-norm (dsteqr scales each block; one scaling suffices within fp64 range)
    double anorm = 0.0;
    for (int i = 0; i < n; ++i) anorm = fmax(anorm, fabs(d[i]));
    for (int i = 0; i + 1 < n; ++i) anorm = fmax(anorm, fabs(e[i]));
    const double sc = anorm > 0.0 ? 1.0 / anorm : 1.0;
    for (int i = 0; i < n; ++i) d[i] *= sc;
    for (int i = 0; i + 1 < n; ++i) e[i] *= sc;
    const long long nmaxit = 30LL * n;
    long long jtot = 0;
    int l1 = 0;  // 0-based throughout
    while (l1 < n && status == 0) {
      if (l1 > 0) e[l1 - 1] = 0.0;
      int m = n - 1;
      for (int mm = l1; mm < n - 1; ++mm) {
        const double tst = fabs(e[mm]);
        if (tst == 0.0) { m = mm; break; }
        if (tst <= sqrt(fabs(d[mm])) * sqrt(fabs(d[mm + 1])) * eps) { e[mm] = 0.0; m = mm; break; }
      }
      int l = l1, lend = m;
      const int lsv = l, lendsv = lend;
      l1 = m + 1;
      if (lend == l) continue;
      if (fabs(d[lend]) < fabs(d[l])) { lend = lsv; l = lendsv; }
      if (lend > l) {
        // ---------------- QL iteration
        while (l <= lend && status == 0) {
          int mq = lend;
          for (int mm = l; mm < lend; ++mm) {
            const double tst = e[mm] * e[mm];
            if (tst <= (eps2 * fabs(d[mm])) * fabs(d[mm + 1]) + safmin) { mq = mm; break; }
          }
          if (mq < lend) e[mq] = 0.0;
          double p = d[l];
          if (mq == l) { ++l; continue; }  // eigenvalue found (d[l] already p)
          if (mq == l + 1) {
            double rt1, rt2, c, s;
            dev_laev2(d[l], e[l], d[l + 1], rt1, rt2, c, s);
            if (!record(0, l, 1)) break;
            if (APPLY) rotate(l, c, s);
            else Q.rot[nr - 1] = make_double2(c, s);
            d[l] = rt1; d[l + 1] = rt2; e[l] = 0.0;
            l += 2;
            continue;
          }
          if (jtot == nmaxit) { status = 1; break; }
          ++jtot;
          double g = (d[l + 1] - p) / (2.0 * e[l]);
          double r = hypot(g, 1.0);
          g = d[mq] - p + (e[l] / (g + copysign(r, g)));
          double s = 1.0, c = 1.0;
          p = 0.0;
          if (!record(0, l, mq - l)) break;
          double2* rot = APPLY ? nullptr : Q.rot + (nr - (mq - l));
          double dup = d[mq];            // d[i+1], carried in a register
          double ei = e[mq - 1], di = d[mq - 1];
          for (int i = mq - 1; i >= l; --i) {
            // prefetch the next plane's entries: independent of the dependency chain below
            const double en = (i > l) ? e[i - 1] : 0.0, dn = (i > l) ? d[i - 1] : 0.0;
            const double f = s * ei, b = c * ei;
            dev_lartg(g, f, c, s, r);
            if (i != mq - 1) e[i + 1] = r;
            g = dup - p;
            r = (di - g) * s + 2.0 * c * b;
            p = s * r;
            d[i + 1] = g + p;
            g = c * r - b;
            if (APPLY) rotate(i, c, -s);
            else rot[i - l] = make_double2(c, -s);
            dup = di;
            ei = en;
            di = dn;
          }
          d[l] -= p;
          e[l] = g;
        }
      } else {
        // ---------------- QR iteration (mirror image)
        while (l >= lend && status == 0) {
          int mq = lend;
          for (int mm = l; mm > lend; --mm) {
            const double tst = e[mm - 1] * e[mm - 1];
            if (tst <= (eps2 * fabs(d[mm])) * fabs(d[mm - 1]) + safmin) { mq = mm; break; }
          }
          if (mq > lend) e[mq - 1] = 0.0;
          double p = d[l];
          if (mq == l) { --l; continue; }
          if (mq == l - 1) {
            double rt1, rt2, c, s;
            dev_laev2(d[l - 1], e[l - 1], d[l], rt1, rt2, c, s);
            if (!record(1, l - 1, 1)) break;
            if (APPLY) rotate(l - 1, c, s);
            else Q.rot[nr - 1] = make_double2(c, s);
            d[l - 1] = rt1; d[l] = rt2; e[l - 1] = 0.0;
            l -= 2;
            continue;
          }
          if (jtot == nmaxit) { status = 1; break; }
          ++jtot;
          double g = (d[l - 1] - p) / (2.0 * e[l - 1]);
          double r = hypot(g, 1.0);
          g = d[mq] - p + (e[l - 1] / (g + copysign(r, g)));
          double s = 1.0, c = 1.0;
          p = 0.0;
          if (!record(1, mq, l - mq)) break;
          double2* rot = APPLY ? nullptr : Q.rot + (nr - (l - mq));
          double dlo = d[mq];            // d[i], carried in a register
          double ei = e[mq], di1 = d[mq + 1];
          for (int i = mq; i <= l - 1; ++i) {
            const double en = (i < l - 1) ? e[i + 1] : 0.0, dn = (i < l - 1) ? d[i + 2] : 0.0;
            const double f = s * ei, b = c * ei;
            dev_lartg(g, f, c, s, r);
            if (i != mq) e[i - 1] = r;
            g = dlo - p;
            r = (di1 - g) * s + 2.0 * c * b;
            p = s * r;
            d[i] = g + p;
            g = c * r - b;
            if (APPLY) rotate(i, c, s);
            else rot[i - mq] = make_double2(c, s);
            dlo = di1;
            ei = en;
            di1 = dn;
          }
          d[l] -= p;
          e[l - 1] = g;
        }
      }
    }
    if (lane == 0)
      for (int i = 0; i < n; ++i) Q.d[i] = d[i] * (anorm > 0.0 ? anorm : 1.0);
  }
  if (APPLY && lane < n) {
    if (n == 1) Q.Zt[0] = 1.0;
    else
      for (int c = 0; c < n; ++c) Q.Zt[(size_t)c * Q.ldz + lane] = Zs[c * n + lane];
  }
  if (lane == 0) {
    *Q.nsweeps = ns;
    *Q.status = status;
  }
}

// ---- T3: apply the recorded rotations to the rows of Z (stored column-major: Zt[col*n + row]) ----
struct rot_mat {
  double* Zt; int n; int ldz;  // Zt[col * ldz + row]
  const int* sw_dir; const int* sw_lo; const int* sw_cnt; const long long* sw_off; const double2* rot;
  const int* nsweeps;
};

// Up to KS consecutive sweeps of the same direction are pipelined: in "logical" coordinates
// (physical for QL sweeps, reflected c -> n-1-c for QR sweeps) every sweep runs over planes in
// descending order, sweep s+1 trails sweep s by two planes, and a thread keeps the 2*KS columns
// in flight in registers — so one pass over a row of Z does the work of KS sweeps.
__global__ __launch_bounds__(256) void rot_apply_kernel(const rot_mat* __restrict__ rs) {
  const rot_mat R = rs[blockIdx.y];
  const int n = R.n;
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (n < 2) return;
  const bool live = row < n;
  const int ns = *R.nsweeps;
  double* __restrict__ z = R.Zt + (live ? row : 0);
  const size_t ldz = (size_t)R.ldz;
  int s0 = 0;
  while (s0 < ns) {
    const int dir = R.sw_dir[s0];
    int cnt = 1;
    while (cnt < KS && s0 + cnt < ns && R.sw_dir[s0 + cnt] == dir) ++cnt;
    // logical plane range [glo, ghi] of each sweep in the group, and rotation lookup
    int glo[KS], ghi[KS], plo[KS];
    long long goff[KS];
    int cmin = n, cmax = -1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < cnt) {
        const int lo = R.sw_lo[s0 + s], c = R.sw_cnt[s0 + s];
        plo[s] = lo;
        goff[s] = R.sw_off[s0 + s];
        if (dir == 0) { glo[s] = lo; ghi[s] = lo + c - 1; }
        else { glo[s] = n - 2 - (lo + c - 1); ghi[s] = n - 2 - lo; }
        cmin = min(cmin, glo[s]);
        cmax = max(cmax, ghi[s] + 1);
      } else {
        glo[s] = 1; ghi[s] = 0; plo[s] = 0; goff[s] = 0;  // empty
      }
    }
    s0 += cnt;
    if (cmax < 0) continue;
    auto phys = [&](int c) { return dir == 0 ? c : n - 1 - c; };
    const int top = cmax - 1;
    double w[2 * KS];
#pragma unroll
    for (int j = 0; j < 2 * KS; ++j) {
      const int col = top + j;
      w[j] = (live && col <= cmax && col >= cmin) ? z[(size_t)phys(col) * ldz] : 0.0;
    }
    const int tend = top - cmin + 2 * (KS - 1);
    // prefetch queue: pre[q] = logical column (top - 1 - q), i.e. the next PF columns below the window
    double pre[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int col = top - 1 - q;
      pre[q] = (live && col >= cmin) ? z[(size_t)phys(col) * ldz] : 0.0;
    }
    for (int t = 0; t <= tend; ++t) {
      const int base = top - t;  // logical column of w[0]
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int i = base + 2 * s;  // logical plane of sweep s at this step
        if (i >= glo[s] && i <= ghi[s]) {
          const int pj = dir == 0 ? i : n - 2 - i;  // physical plane
          const double2 cs = R.rot[goff[s] + (pj - plo[s])];
          const double a0 = w[2 * s], a1 = w[2 * s + 1];
          if (dir == 0) {  // a0 = z[j], a1 = z[j+1]
            w[2 * s + 1] = cs.x * a1 - cs.y * a0;
            w[2 * s] = cs.y * a1 + cs.x * a0;
          } else {         // a0 = z[j+1], a1 = z[j]
            w[2 * s] = cs.x * a0 - cs.y * a1;
            w[2 * s + 1] = cs.y * a0 + cs.x * a1;
          }
        }
      }
      const int ctop = base + 2 * KS - 1;
      if (live && ctop <= cmax && ctop >= cmin) z[(size_t)phys(ctop) * ldz] = w[2 * KS - 1];
#pragma unroll
      for (int j = 2 * KS - 1; j > 0; --j) w[j] = w[j - 1];
      w[0] = pre[0];  // column base - 1
#pragma unroll
      for (int q = 0; q + 1 < PF; ++q) pre[q] = pre[q + 1];
      const int cpre = base - 1 - PF;  // keeps the queue PF columns ahead
      pre[PF - 1] = (live && cpre >= cmin) ? z[(size_t)phys(cpre) * ldz] : 0.0;
    }
    {
      const int base = top - tend - 1;
#pragma unroll
      for (int j = 0; j < 2 * KS; ++j) {
        const int col = base + j;
        if (live && col >= cmin && col <= cmax) z[(size_t)phys(col) * ldz] = w[j];
      }
    }
  }
}

// Zt (column-major real) identity
__global__ void zt_identity_kernel(const rot_mat* __restrict__ rs) {
  const rot_mat R = rs[blockIdx.z];
  const int col = blockIdx.y, row = blockIdx.x * blockDim.x + threadIdx.x;
  if (col < R.n && row < R.n) R.Zt[(size_t)col * R.ldz + row] = (row == col) ? 1.0 : 0.0;
}

// Zsel[c'][:] = Z[idx[c']][:]  (eigenvector-major: one vector = n contiguous doubles); grid (vector tiles, problems)
struct zsel_mat { const double* Z; double* Zsel; const int* idx; int n; int nsel; };
__global__ __launch_bounds__(256) void zsel_gather_kernel(const zsel_mat* __restrict__ zs) {
  const zsel_mat S = zs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= S.nsel) return;
  const double* src = S.Z + (size_t)S.idx[c] * S.n;
  double* dst = S.Zsel + (size_t)c * S.n;
  for (int i = lane; i < S.n; i += 64) dst[i] = src[i];
}

// X[row][col] (complex row-major, ld) = Zt[col*n + row]
struct cvt_mat { const double* Zt; cplx* X; int ldx; int n; int ncol; };  // X is n x ncol
__global__ void zt_to_x_kernel(const cvt_mat* __restrict__ cs) {
  __shared__ double tile[32][33];
  const cvt_mat C = cs[blockIdx.z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: rows of X, by: cols of X
  if (bx >= C.n || by >= C.ncol) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int jj = ty; jj < 32; jj += 8) {
    const int col = by + jj, row = bx + tx;  // read Zt[col][row], row fastest
    tile[jj][tx] = (col < C.ncol && row < C.n) ? C.Zt[(size_t)col * C.n + row] : 0.0;
  }
  __syncthreads();
  for (int jj = ty; jj < 32; jj += 8) {
    const int row = bx + jj, col = by + tx;
    if (row < C.n && col < C.ncol) C.X[(size_t)row * C.ldx + col] = make_double2(tile[tx][jj], 0.0);
  }
}

// ---- T4 helper: T factor of a block of reflectors from its Gram matrix (zlarft, forward/columnwise) ----
struct tf_mat {
  cplx* G; const cplx* tau; cplx* T; int kb; int ldt;  // G: TNB x TNB row-major; T: leading dimension ldt
  const cplx* part; int nslice;                         // nslice > 0: G = sum of nslice (kb x kb) slices at `part` (stored to G)
};
// The Gram matrix goes to LDS first (summed over the split-K slices if there are any) and the recurrence runs from LDS with
// eight threads per row of T sharing each dot product: 32 dependent steps of a few operations each (one thread per row
// walking along it, waiting for a global load per element, took 50 us per call).
__global__ __launch_bounds__(256) void larft_kernel(const tf_mat* __restrict__ ts) {
  const tf_mat F = ts[blockIdx.x];
  extern __shared__ __align__(16) unsigned char larft_smem[];
  cplx (*T)[TNB + 1] = reinterpret_cast<cplx (*)[TNB + 1]>(larft_smem);
  cplx (*Gl)[TNB + 1] = reinterpret_cast<cplx (*)[TNB + 1]>(larft_smem + sizeof(cplx) * TNB * (TNB + 1));
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int idx = tid; idx < TNB * TNB; idx += nth) {
    const int r = idx / TNB, c = idx % TNB;
    T[r][c] = make_double2(0.0, 0.0);
    cplx g = make_double2(0.0, 0.0);
    if (r < F.kb && c < F.kb) {
      if (F.nslice > 0) {
        const size_t ss = (size_t)F.kb * F.kb, o = (size_t)r * F.kb + c;
        int sl = 0;
        for (; sl + 4 <= F.nslice; sl += 4) {   // four slices in flight (the sum keeps its order)
          const cplx v0 = F.part[sl * ss + o], v1 = F.part[(sl + 1) * ss + o], v2 = F.part[(sl + 2) * ss + o], v3 = F.part[(sl + 3) * ss + o];
          g = cadd(cadd(cadd(cadd(g, v0), v1), v2), v3);
        }
        for (; sl < F.nslice; ++sl) g = cadd(g, F.part[sl * ss + o]);
        F.G[r * TNB + c] = g;
      } else {
        g = F.G[r * TNB + c];
      }
    }
    Gl[r][c] = g;
  }
  __syncthreads();
  constexpr int TPR = 256 / TNB >= 8 ? 8 : 4;   // threads per row of T (8 for the 32-wide panels, 4 for the 64-wide ones)
  const int row = tid / TPR, part = tid % TPR;
  for (int j = 0; j < F.kb; ++j) {
    const cplx tj = F.tau[j];
    // T[0:j, j] = -tau_j * T[0:j, 0:j] * G[0:j, j]   (T upper triangular: columns c >= row)
    cplx acc = make_double2(0.0, 0.0);
    if (row < j)
      for (int c = row + part; c < j; c += TPR) acc = cadd(acc, cmul(T[row][c], Gl[c][j]));
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
      acc.x += __shfl_xor(acc.x, o, 64);
      acc.y += __shfl_xor(acc.y, o, 64);
    }
    if (part == 0 && row < j) T[row][j] = cmul(make_double2(-tj.x, -tj.y), acc);
    if (tid == 0) T[j][j] = tj;
    __syncthreads();
  }
  for (int idx = tid; idx < TNB * TNB; idx += nth) F.T[(size_t)(idx / TNB) * F.ldt + idx % TNB] = T[idx / TNB][idx % TNB];
}


// ===========================================================================
// T2/T3 alternative: divide & conquer on the tridiagonal (Cuppen; deflation, secular equation
// and Gu-Eisenstat vectors as in LAPACK dlaed2/3/4).  The tridiagonal is torn into leaves of
// <= DC_LEAF rows, the leaves are solved by the QL kernels above, and the tree is merged level
// by level with every node of a level (all matrices) in the same launches:
//   dc_setup    z vector, sort, deflation (tiny z / close poles via Givens)      1 WG / node
//   dc_permute  rotate + gather the non-deflated eigenvectors, copy the deflated ones
//   dc_secular  one thread per root: safeguarded rational iteration, root kept as (origin, mu)
//   dc_zhat     Loewner formula for z-hat (numerical orthogonality)
//   dc_unorm / dc_ubuild   eigenvectors of the rank-one modified diagonal
//   grouped DGEMM         Z_parent = U^T Z_children                                (MFMA)
// Parallel depth O(log n) instead of the ~1.1 n^2 serial rotations of QL.
// ===========================================================================
constexpr int DC_LEAF = 32;
constexpr int DC_MAXNODE = 4096;  // LDS-resident setup / secular kernels up to here, global-scratch variants beyond

struct dc_mat {
  int n;
  double* lamA; double* lamB;   // eigenvalues of the current / next level (ping-pong)
  double* ZA; double* ZB;       // eigenvector-major: Z[c * n + r]
  double* Zp;                   // gathered non-deflated eigenvectors
  double* dk; double* zk;       // packed poles / weights of each node (at offset lo)
  int* keepcol; int* deflcol;   // local column indices
  double* defld;
  double4* rots;                // (colA, colB, c, s) with the column indices stored as doubles
  int* org; double* mu; double* zhat; double* inv;
  double* U;                    // n x n scratch: node block at U + lo * n, leading dimension n
  double* gs; int* gi;          // 4 n doubles + n ints: setup scratch of nodes too large for LDS (node at 4 lo / lo)
};

struct dc_node {
  int mat, lo, n1, n2;
  const double* pbeta;  // off-diagonal element torn at this node
  int flip;             // 0: current = A buffers, 1: current = B buffers
};

struct dc_nodeout { int k, ndefl, nrot; double rho; };

// BIG = false: the node's work arrays live in LDS (nn <= DC_MAXNODE); BIG = true: in the global scratch
// M.gs / M.gi (any nn), the counting sort then broadcasts 64 keys at a time through lane reads.
template <bool BIG>
__global__ __launch_bounds__(256) void dc_setup_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                       dc_nodeout* __restrict__ outs) {
  extern __shared__ __align__(16) unsigned char dc_smem[];
  const dc_node nd = nodes[blockIdx.x];
  const dc_mat M = ms[nd.mat];
  const int nn = nd.n1 + nd.n2, lo = nd.lo, n = M.n;
  double* sd = BIG ? M.gs + 4 * (size_t)lo : reinterpret_cast<double*>(dc_smem);   // sorted poles
  double* sz = sd + nn;                              // sorted weights
  double* ud = sz + nn;                              // unsorted copies
  double* uz = ud + nn;
  int* sidx = BIG ? M.gi + lo : reinterpret_cast<int*>(uz + nn);       // sorted position -> local column
  __shared__ double red[4];
  __shared__ double s_norm, s_zmax, s_dmax;
  const int tid = threadIdx.x;
  const double* lam = nd.flip ? M.lamB : M.lamA;
  const double* Z = nd.flip ? M.ZB : M.ZA;
  const double beta = *nd.pbeta;
  const double sgn = beta >= 0.0 ? 1.0 : -1.0;
  double part = 0.0;
  for (int i = tid; i < nn; i += 256) {
    ud[i] = lam[lo + i];
    const double zi = (i < nd.n1) ? Z[(size_t)(lo + i) * n + (lo + nd.n1 - 1)] : sgn * Z[(size_t)(lo + i) * n + (lo + nd.n1)];
    uz[i] = zi;
    part += zi * zi;
  }
  part = dm_wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) s_norm = sqrt(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  const double zn = s_norm;
  const double rho = fabs(beta) * zn * zn;
  // rank by counting (stable), scatter into sorted order
  double zmax = 0.0, dmax = 0.0;
  if (BIG) __syncthreads();  // ud / uz of the other waves (global scratch)
  for (int i0 = 0; i0 < nn; i0 += 256) {
    const int i = i0 + tid;  // the loop is wave-uniform: lanes past the end only help with the broadcasts
    const double di = i < nn ? ud[i] : 0.0;
    int r = 0;
    if (BIG) {
      const int lane = tid & 63;
      for (int j0 = 0; j0 < nn; j0 += 64) {
        const double mine = (j0 + lane < nn) ? ud[j0 + lane] : __builtin_inf();
#pragma unroll 16
        for (int t = 0; t < 64; ++t) {
          const double dj = __shfl(mine, t, 64);
          r += (dj < di || (dj == di && j0 + t < i)) ? 1 : 0;
        }
      }
    } else {
      for (int j = 0; j < nn; ++j) {
        const double dj = ud[j];
        r += (dj < di || (dj == di && j < i)) ? 1 : 0;
      }
    }
    if (i < nn) {
      const double zi = zn > 0.0 ? uz[i] / zn : 0.0;
      sd[r] = di;
      sz[r] = zi;
      sidx[r] = i;
      zmax = fmax(zmax, fabs(zi));
      dmax = fmax(dmax, fabs(di));
    }
  }
  zmax = dm_wave_max(zmax);
  dmax = dm_wave_max(dmax);
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = zmax; }
  __syncthreads();
  if (tid == 0) s_zmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = dmax; }
  __syncthreads();
  if (tid == 0) s_dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  if (tid != 0) return;
  // ---- serial deflation scan (dlaed2)
  const double eps = 1.1102230246251565e-16;
  const double tol = 8.0 * eps * fmax(s_dmax, s_zmax);
  int k = 0, ndefl = 0, nrot = 0;
  int* keeppos = reinterpret_cast<int*>(ud);  // reuse: positions (in sorted order) of kept entries
  if (rho * s_zmax <= tol) {
    for (int i = 0; i < nn; ++i) { M.deflcol[lo + ndefl] = sidx[i]; M.defld[lo + ndefl] = sd[i]; ++ndefl; }
  } else {
    int prev = -1;
    for (int i = 0; i < nn; ++i) {
      if (rho * fabs(sz[i]) <= tol) {
        M.deflcol[lo + ndefl] = sidx[i]; M.defld[lo + ndefl] = sd[i]; ++ndefl;
        continue;
      }
      if (prev >= 0) {
        double s = sz[prev], c = sz[i];
        const double tau = hypot(c, s);
        const double t = sd[i] - sd[prev];
        c /= tau;
        s = -s / tau;
        if (fabs(t * c * s) <= tol) {
          sz[i] = tau;
          sz[prev] = 0.0;
          M.rots[lo + nrot] = make_double4((double)sidx[prev], (double)sidx[i], c, s);
          ++nrot;
          const double dp = sd[prev], di = sd[i];
          sd[prev] = dp * c * c + di * s * s;
          sd[i] = dp * s * s + di * c * c;
          M.deflcol[lo + ndefl] = sidx[prev]; M.defld[lo + ndefl] = sd[prev]; ++ndefl;
          keeppos[k - 1] = i;
          prev = i;
          continue;
        }
      }
      keeppos[k++] = i;
      prev = i;
    }
    // poles must increase: the rotations can perturb the order by a few ulp -> insertion sort
    for (int a = 1; a < k; ++a) {
      const int pa = keeppos[a];
      const double da = sd[pa];
      int b = a - 1;
      while (b >= 0 && sd[keeppos[b]] > da) { keeppos[b + 1] = keeppos[b]; --b; }
      keeppos[b + 1] = pa;
    }
    for (int j = 0; j < k; ++j) {
      const int pos = keeppos[j];
      M.dk[lo + j] = sd[pos];
      M.zk[lo + j] = sz[pos];
      M.keepcol[lo + j] = sidx[pos];
    }
  }
  outs[blockIdx.x] = dc_nodeout{k, ndefl, nrot, rho};
}

__global__ __launch_bounds__(256) void dc_permute_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                         const dc_nodeout* __restrict__ outs) {
  const dc_node nd = nodes[blockIdx.x];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.x];
  const int nn = nd.n1 + nd.n2, lo = nd.lo, n = M.n;
  double* Zc = nd.flip ? M.ZB : M.ZA;
  const int tid = threadIdx.x;
  // chained Givens rotations on pairs of eigenvectors (in place)
  for (int r = 0; r < o.nrot; ++r) {
    const double4 rt = M.rots[lo + r];
    double* qa = Zc + (size_t)(lo + (int)rt.x) * n + lo;
    double* qb = Zc + (size_t)(lo + (int)rt.y) * n + lo;
    for (int i = tid; i < nn; i += 256) {
      const double a = qa[i], b = qb[i];
      qa[i] = rt.z * a + rt.w * b;
      qb[i] = -rt.w * a + rt.z * b;
    }
    __syncthreads();
  }
}

// gather the non-deflated vectors for the GEMM, copy the deflated ones to their final place;
// grid = (vector tiles of DCG, nodes), one wave per vector
constexpr int DCG = 16;
__global__ __launch_bounds__(256) void dc_gather_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                        const dc_nodeout* __restrict__ outs) {
  const dc_node nd = nodes[blockIdx.y];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.y];
  const int nn = nd.n1 + nd.n2, lo = nd.lo, n = M.n;
  if ((int)blockIdx.x * DCG >= nn) return;
  const double* Zc = nd.flip ? M.ZB : M.ZA;
  double* Zn = nd.flip ? M.ZA : M.ZB;
  double* lamn = nd.flip ? M.lamA : M.lamB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int u = wave; u < DCG; u += 4) {
    const int j = blockIdx.x * DCG + u;
    if (j >= nn) break;
    const double* src;
    double* dst;
    if (j < o.k) {
      src = Zc + (size_t)(lo + M.keepcol[lo + j]) * n + lo;
      dst = M.Zp + (size_t)(lo + j) * n + lo;
    } else {
      const int t = j - o.k;
      if (t >= o.ndefl) break;
      src = Zc + (size_t)(lo + M.deflcol[lo + t]) * n + lo;
      dst = Zn + (size_t)(lo + o.k + t) * n + lo;
      if (lane == 0) lamn[lo + o.k + t] = M.defld[lo + t];
    }
    for (int i = lane; i < nn; i += 64) dst[i] = src[i];
  }
}

// grid = (root tiles of 256, nodes); dynamic LDS: 2 * kmax doubles (BIG: poles and weights are read
// from global memory instead -- every lane of a wave reads the same element, one request per load)
template <bool BIG>
__global__ __launch_bounds__(256) void dc_secular_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                         const dc_nodeout* __restrict__ outs) {
  extern __shared__ __align__(16) unsigned char dc_smem[];
  const dc_node nd = nodes[blockIdx.y];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.y];
  const int k = o.k, lo = nd.lo;
  if ((int)(blockIdx.x * 256) >= k) return;
  const double* d = BIG ? M.dk + lo : reinterpret_cast<double*>(dc_smem);
  const double* zsrc = BIG ? M.zk + lo : d + k;
  auto Z2 = [&](int i) -> double {
    const double z = zsrc[i];
    return BIG ? z * z : z;
  };
  if (!BIG) {
    double* d = reinterpret_cast<double*>(dc_smem);
    double* z2 = d + k;
    for (int i = threadIdx.x; i < k; i += 256) {
      d[i] = M.dk[lo + i];
      const double z = M.zk[lo + i];
      z2[i] = z * z;
    }
    __syncthreads();
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k) return;
  const double rho = o.rho;
  const double eps = 2.220446049250313e-16;
  double* lamn = nd.flip ? M.lamA : M.lamB;
  if (k == 1) {
    M.org[lo] = 0;
    M.mu[lo] = rho * Z2(0);
    lamn[lo] = d[0] + rho * Z2(0);
    return;
  }
  const bool last = (j == k - 1);
  int og;
  double lo_b, hi_b;
  if (!last) {
    const double mid = 0.5 * (d[j + 1] - d[j]);
    double f = 1.0;
    for (int i = 0; i < k; ++i) f += rho * Z2(i) / ((d[i] - d[j]) - mid);
    if (f > 0.0) { og = j; lo_b = 0.0; hi_b = mid; } else { og = j + 1; lo_b = -mid; hi_b = 0.0; }
  } else {
    og = j;
    double sz = 0.0;
    for (int i = 0; i < k; ++i) sz += Z2(i);
    lo_b = 0.0;
    hi_b = rho * sz;
  }
  const double dorg = d[og];
  double mu = 0.5 * (lo_b + hi_b);
  for (int it = 0; it < 100; ++it) {
    double psi = 0.0, phi = 0.0, dpsi = 0.0, dphi = 0.0;
    for (int i = 0; i <= j; ++i) {
      const double t = 1.0 / ((d[i] - dorg) - mu);
      const double term = rho * Z2(i) * t;
      psi += term;
      dpsi += term * t;
    }
    for (int i = j + 1; i < k; ++i) {
      const double t = 1.0 / ((d[i] - dorg) - mu);
      const double term = rho * Z2(i) * t;
      phi += term;
      dphi += term * t;
    }
    const double fv = 1.0 + psi + phi;
    const double erretm = 8.0 * (fabs(psi) + fabs(phi)) + 1.0 + fabs(mu) * (dpsi + dphi);
    if (fabs(fv) <= eps * erretm) break;
    if (fv > 0.0) hi_b = mu; else lo_b = mu;
    double eta;
    if (!last) {
      const double dj = (d[j] - dorg) - mu, dj1 = (d[j + 1] - dorg) - mu;
      const double a = (dj + dj1) * fv - dj * dj1 * (dpsi + dphi);
      const double b = dj * dj1 * fv;
      const double c = fv - dj * dpsi - dj1 * dphi;
      if (c == 0.0) {
        eta = a != 0.0 ? b / a : 0.0;
      } else {
        const double disc = sqrt(fmax(a * a - 4.0 * b * c, 0.0));
        eta = (a <= 0.0) ? (a - disc) / (2.0 * c) : 2.0 * b / (a + disc);
      }
    } else {
      const double tq = (d[j] - dorg) - mu, tp = (d[j - 1] - dorg) - mu;
      const double dphil = rho * Z2(j) / (tq * tq);
      const double dpsil = dpsi + dphi - dphil;
      double c = fv - tp * dpsil - tq * dphil;
      const double a = (tp + tq) * fv - tp * tq * (dpsil + dphil);
      const double b = tp * tq * fv;
      if (c < 0.0) c = -c;
      if (c == 0.0) eta = hi_b - mu;
      else if (a >= 0.0) eta = (a + sqrt(fabs(a * a - 4.0 * b * c))) / (2.0 * c);
      else eta = 2.0 * b / (a - sqrt(fabs(a * a - 4.0 * b * c)));
      if (fv * eta > 0.0) eta = -fv / (dpsi + dphi);
    }
    double nw = mu + eta;
    if (!(nw > lo_b && nw < hi_b) || !isfinite(nw)) nw = 0.5 * (lo_b + hi_b);
    if (nw == mu || (hi_b - lo_b) <= 2.0 * eps * fabs(nw)) { mu = nw; break; }
    mu = nw;
  }
  M.org[lo + j] = og;
  M.mu[lo + j] = mu;
  lamn[lo + j] = dorg + mu;
}

// zhat_i = sign(z_i) sqrt( prod_j (lam_j - d_i) / (rho prod_{j != i} (d_j - d_i)) )
__global__ __launch_bounds__(256) void dc_zhat_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                      const dc_nodeout* __restrict__ outs) {
  const dc_node nd = nodes[blockIdx.y];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.y];
  const int k = o.k, lo = nd.lo;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const double di = M.dk[lo + i];
  double prod = 1.0;
  for (int j = 0; j < k; ++j) {
    const double num = M.mu[lo + j] - (di - M.dk[lo + M.org[lo + j]]);  // lam_j - d_i
    if (j == i) prod *= num;
    else prod *= num / (M.dk[lo + j] - di);
  }
  const double zh = sqrt(fabs(prod) / o.rho);
  M.zhat[lo + i] = M.zk[lo + i] >= 0.0 ? zh : -zh;
}

// inv[j] = 1 / || zhat_i / (d_i - lam_j) ||_i
__global__ __launch_bounds__(256) void dc_unorm_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                       const dc_nodeout* __restrict__ outs) {
  const dc_node nd = nodes[blockIdx.y];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.y];
  const int k = o.k, lo = nd.lo;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k) return;
  const double dor = M.dk[lo + M.org[lo + j]], mu = M.mu[lo + j];
  double s = 0.0;
  for (int i = 0; i < k; ++i) {
    const double u = M.zhat[lo + i] / ((M.dk[lo + i] - dor) - mu);
    s += u * u;
  }
  M.inv[lo + j] = 1.0 / sqrt(s);
}

// Ut[j][i] = zhat_i / (d_i - lam_j) * inv_j   (row j = eigenvector j of the rank-one problem)
__global__ __launch_bounds__(256) void dc_ubuild_kernel(const dc_mat* __restrict__ ms, const dc_node* __restrict__ nodes,
                                                        const dc_nodeout* __restrict__ outs) {
  const dc_node nd = nodes[blockIdx.z];
  const dc_mat M = ms[nd.mat];
  const dc_nodeout o = outs[blockIdx.z];
  const int k = o.k, lo = nd.lo;
  const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (i >= k || j >= k) return;
  const double dor = M.dk[lo + M.org[lo + j]], mu = M.mu[lo + j];
  M.U[(size_t)lo * M.n + (size_t)j * M.n + i] = M.zhat[lo + i] / ((M.dk[lo + i] - dor) - mu) * M.inv[lo + j];
}

// LAPACK's dstedc scales the tridiagonal to unit max-norm before the divide & conquer (DLASCL with ORGNRM) and scales
// the eigenvalues back: dlaed2's deflation tolerance 8 eps max(|d|, |z|) compares poles — which carry the scale of the
// matrix — with components of unit vectors.  Without the scaling a matrix of norm 1e-9 had its eigenvalues computed to
// 1e-13 .. 1e-9 of its norm instead of 1e-15 (rounds 1-4; found by the full-size spectrum parity of bench.py on the
// configs[1] blocks m = 101 .. 104, whose S/N pencils have lambda_max ~ 1e-10).
struct dc_scale_mat { double* d; double* e; int n; double* scale; };
__global__ __launch_bounds__(256) void dc_scale_kernel(const dc_scale_mat* __restrict__ ms) {
  const dc_scale_mat M = ms[blockIdx.x];
  __shared__ double red[4];
  double mx = 0.0;
  for (int i = threadIdx.x; i < M.n; i += 256) {
    mx = fmax(mx, fabs(M.d[i]));
    if (i + 1 < M.n) mx = fmax(mx, fabs(M.e[i]));
  }
  mx = dm_wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const bool ok = mx > 0.0 && mx < 1e300;   // (an all-zero or non-finite tridiagonal is left as it is)
  if (threadIdx.x == 0) *M.scale = ok ? mx : 1.0;
  if (!ok) return;
  const double inv = 1.0 / mx;
  for (int i = threadIdx.x; i < M.n; i += 256) {
    M.d[i] *= inv;
    if (i + 1 < M.n) M.e[i] *= inv;
  }
}
__global__ __launch_bounds__(256) void dc_unscale_kernel(const dc_scale_mat* __restrict__ ms) {
  const dc_scale_mat M = ms[blockIdx.x];
  const double sc = *M.scale;
  for (int i = threadIdx.x; i < M.n; i += 256) M.d[i] *= sc;
}

struct dc_tear { double* d; const double* e; int b; };
__global__ void dc_tear_kernel(const dc_tear* __restrict__ ts, int nt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nt) return;
  const dc_tear t = ts[i];
  const double ab = fabs(t.e[t.b - 1]);
  t.d[t.b - 1] -= ab;
  t.d[t.b] -= ab;
}

#if DM_TNB == 32
#include "dm_sbr_impl.h"
#endif

}  // namespace


// ---- D&C driver: on entry dd/ee hold the tridiagonals (offsets offn); on return dd holds the
// eigenvalues (unsorted) and zfinal[p] points at the eigenvector-major n x n eigenvector array.
static int dc_solve(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* dd, double* ee,
                    const std::vector<size_t>& offn, const std::vector<size_t>& off, size_t tot, size_t totn,
                    std::vector<double*>& zfinal, double* scratch2 = nullptr) {
  const int np = (int)probs.size();
  double* ZA = dm_ws_alloc_t<double>(ctx, std::max<size_t>(tot, 1));
  double* ZB = dm_ws_alloc_t<double>(ctx, std::max<size_t>(tot, 1));
  // gathered vectors and rank-one eigenvector blocks only live inside this function: the caller may lend
  // 2 tot doubles it does not need yet (the T V^H buffer of the back-transformation)
  double* Zp = scratch2 ? scratch2 : dm_ws_alloc_t<double>(ctx, std::max<size_t>(tot, 1));
  double* Uw = scratch2 ? scratch2 + tot : dm_ws_alloc_t<double>(ctx, std::max<size_t>(tot, 1));
  double* lamB = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* dk = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* zk = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* defld = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* muv = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* zhat = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* inv = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  int* keepcol = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totn, 1));
  int* deflcol = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totn, 1));
  int* org = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totn, 1));
  double4* rots = dm_ws_alloc_t<double4>(ctx, std::max<size_t>(totn, 1));
  double* gsc = dm_ws_alloc_t<double>(ctx, std::max<size_t>(4 * totn, 1));
  int* gic = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totn, 1));
  if (!gsc || !gic) return DM_ENOMEM;
  if (!ZA || !ZB || !Zp || !Uw || !lamB || !dk || !zk || !defld || !muv || !zhat || !inv || !keepcol || !deflcol ||
      !org || !rots)
    return DM_ENOMEM;
  DM_TRY(dm_fill_zero(ctx, ZA, sizeof(double) * tot));
  DM_TRY(dm_fill_zero(ctx, ZB, sizeof(double) * tot));

  std::vector<dc_mat> dm(np);
  std::vector<int> depth(np, 0);
  int dmax = 0;
  for (int p = 0; p < np; ++p) {
    const int n = probs[p].n;
    int D = 0;
    while (((n + (1 << D) - 1) >> D) > DC_LEAF) ++D;
    depth[p] = D;
    dmax = std::max(dmax, D);
    dm[p] = dc_mat{n, dd + offn[p], lamB + offn[p], ZA + off[p], ZB + off[p], Zp + off[p], dk + offn[p], zk + offn[p],
                   keepcol + offn[p], deflcol + offn[p], defld + offn[p], rots + offn[p], org + offn[p],
                   muv + offn[p], zhat + offn[p], inv + offn[p], Uw + off[p], gsc + 4 * offn[p], gic + offn[p]};
  }
  dc_mat* d_dm = dm_ws_upload(ctx, dm);
  if (!d_dm) return DM_ENOMEM;
  auto bound = [&](int p, int D, int i) { return (int)(((long long)i * probs[p].n) >> D); };

  // ---- unit max-norm tridiagonals (dstedc's DLASCL); the eigenvalues are scaled back at the end
  double* dscale = dm_ws_alloc_t<double>(ctx, std::max(np, 1));
  if (!dscale) return DM_ENOMEM;
  std::vector<dc_scale_mat> scm(np);
  for (int p = 0; p < np; ++p) scm[p] = dc_scale_mat{dd + offn[p], ee + offn[p], probs[p].n, dscale + p};
  dc_scale_mat* d_scm = dm_ws_upload(ctx, scm);
  if (!d_scm) return DM_ENOMEM;
  static const bool dc_noscale = getenv("DM_DC_NOSCALE") != nullptr;
  if (!dc_noscale) DM_PLAUNCH(ctx, DM_PROF_DC, dc_scale_kernel, dim3(np), dim3(256), 0, ctx->stream, d_scm);

  // ---- tear at every leaf boundary, then solve the leaves with the QL kernels
  {
    std::vector<dc_tear> tears;
    std::vector<ql_mat> qm;
    std::vector<rot_mat> rm;
    size_t totsw = 0, totrot = 0;
    int maxleaf = 0;
    for (int p = 0; p < np; ++p) {
      const int n = probs[p].n, D = depth[p];
      if (n == 0) continue;
      for (int i = 0; i < (1 << D); ++i) {
        const int lo = bound(p, D, i), hi = bound(p, D, i + 1);
        if (i > 0) tears.push_back(dc_tear{dd + offn[p], ee + offn[p], lo});
        const int nl = hi - lo;
        maxleaf = std::max(maxleaf, nl);
        totsw += 4 * (size_t)nl + 8;
        totrot += 2 * (size_t)nl * nl + 8;
      }
    }
    int* sw_dir = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totsw, 1));
    int* sw_lo = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totsw, 1));
    int* sw_cnt = dm_ws_alloc_t<int>(ctx, std::max<size_t>(totsw, 1));
    long long* sw_off = dm_ws_alloc_t<long long>(ctx, std::max<size_t>(totsw, 1));
    double2* rot = dm_ws_alloc_t<double2>(ctx, std::max<size_t>(totrot, 1));
    size_t so = 0, ro = 0;
    std::vector<int> leafmat;
    for (int p = 0; p < np; ++p) {
      const int n = probs[p].n, D = depth[p];
      if (n == 0) continue;
      for (int i = 0; i < (1 << D); ++i) {
        const int lo = bound(p, D, i), hi = bound(p, D, i + 1), nl = hi - lo;
        leafmat.push_back(p);
        qm.push_back(ql_mat{dd + offn[p] + lo, ee + offn[p] + lo, nl, sw_dir + so, sw_lo + so, sw_cnt + so, sw_off + so,
                            rot + ro, 4 * nl + 8, 2LL * nl * nl + 8, nullptr, nullptr, ZA + off[p] + (size_t)lo * n + lo, n});
        rm.push_back(rot_mat{ZA + off[p] + (size_t)lo * n + lo, nl, n, sw_dir + so, sw_lo + so, sw_cnt + so,
                             sw_off + so, rot + ro, nullptr});
        so += 4 * (size_t)nl + 8;
        ro += 2 * (size_t)nl * nl + 8;
      }
    }
    const int nleaf = (int)qm.size();
    int* nsw = dm_ws_alloc_t<int>(ctx, std::max(nleaf, 1));
    int* stat = dm_ws_alloc_t<int>(ctx, std::max(nleaf, 1));
    if (!sw_dir || !sw_lo || !sw_cnt || !sw_off || !rot || !nsw || !stat) return DM_ENOMEM;
    for (int i = 0; i < nleaf; ++i) {
      qm[i].nsweeps = nsw + i; qm[i].status = stat + i;
      rm[i].nsweeps = nsw + i;
    }
    if (!tears.empty()) {
      dc_tear* d_t = dm_ws_upload(ctx, tears);
      if (!d_t) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_DC, dc_tear_kernel, dim3(((unsigned)tears.size() + 255) / 256), dim3(256), 0, ctx->stream, d_t,
                         (int)tears.size());
    }
    if (nleaf > 0) {
      ql_mat* d_qm = dm_ws_upload(ctx, qm);
      rot_mat* d_rm = dm_ws_upload(ctx, rm);
      if (!d_qm || !d_rm) return DM_ENOMEM;
      static bool attr = false;
      if (!attr) {
        DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ql_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr = true;
      }
      static const bool leaf_apply = !getenv("DM_QL_LEAF_RECORD");
      if (leaf_apply && maxleaf <= 64) {
        // rotations applied in LDS as they are generated (d, e and the n x n Z of a leaf: 16 n + 8 n^2 bytes)
        DM_PLAUNCH(ctx, DM_PROF_DC, (ql_kernel<true, true>), dim3(nleaf), dim3(64), (size_t)maxleaf * 16 + (size_t)maxleaf * maxleaf * 8,
                           ctx->stream, d_qm);
      } else {
        DM_PLAUNCH(ctx, DM_PROF_DC, ql_kernel<true>, dim3(nleaf), dim3(64), (size_t)maxleaf * 16, ctx->stream, d_qm);
        DM_PLAUNCH(ctx, DM_PROF_DC, zt_identity_kernel, dim3((maxleaf + 255) / 256, maxleaf, nleaf), dim3(256), 0, ctx->stream,
                           d_rm);
        DM_PLAUNCH(ctx, DM_PROF_DC, rot_apply_kernel, dim3((maxleaf + 255) / 256, nleaf), dim3(256), 0, ctx->stream, d_rm);
      }
      DM_HIP(ctx, hipGetLastError());
      std::vector<int> hs(nleaf);
      DM_TRY(dm_download(ctx, hs.data(), stat, sizeof(int) * nleaf));
      for (int i = 0; i < nleaf; ++i)
        if (hs[i] != 0) {
          ctx->err = "tridiagonal QL iteration (D&C leaf) did not converge";
          return 1000 + leafmat[i];
        }
    }
  }

  // ---- merge level by level
  static bool attr2 = false;
  if (!attr2) {
    DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dc_setup_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 36 * DC_MAXNODE + 64));
    DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dc_secular_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16 * DC_MAXNODE + 64));
    attr2 = true;
  }
  for (int l = dmax - 1; l >= 0; --l) {
    std::vector<dc_node> nodes;
    int maxnn = 0;
    for (int p = 0; p < np; ++p) {
      const int D = depth[p];
      if (D <= l || probs[p].n == 0) continue;
      for (int j = 0; j < (1 << l); ++j) {
        const int lo = bound(p, l, j), hi = bound(p, l, j + 1), mid = bound(p, l + 1, 2 * j + 1);
        nodes.push_back(dc_node{p, lo, mid - lo, hi - mid, ee + offn[p] + mid - 1, (D - 1 - l) & 1});
        maxnn = std::max(maxnn, hi - lo);
      }
    }
    if (nodes.empty()) continue;
    const int nn_nodes = (int)nodes.size();
    dc_node* d_nodes = dm_ws_upload(ctx, nodes);
    dc_nodeout* d_out = dm_ws_alloc_t<dc_nodeout>(ctx, nn_nodes);
    if (!d_nodes || !d_out) return DM_ENOMEM;
    // levels with a node beyond the LDS capacity take the global-scratch variants (a handful of nodes)
    const bool big = maxnn > DC_MAXNODE;
    if (big)
      DM_PLAUNCH(ctx, DM_PROF_DC, dc_setup_kernel<true>, dim3(nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    else
      DM_PLAUNCH(ctx, DM_PROF_DC, dc_setup_kernel<false>, dim3(nn_nodes), dim3(256), (size_t)36 * maxnn + 64, ctx->stream, d_dm,
                         d_nodes, d_out);
    DM_PLAUNCH(ctx, DM_PROF_DC, dc_permute_kernel, dim3(nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    DM_PLAUNCH(ctx, DM_PROF_DC, dc_gather_kernel, dim3((maxnn + DCG - 1) / DCG, nn_nodes), dim3(256), 0, ctx->stream, d_dm,
                       d_nodes, d_out);
    DM_HIP(ctx, hipGetLastError());
    std::vector<dc_nodeout> ho(nn_nodes);
    DM_TRY(dm_download(ctx, ho.data(), d_out, sizeof(dc_nodeout) * nn_nodes));
    int kmax = 0;
    for (auto& o : ho) kmax = std::max(kmax, o.k);
    if (kmax == 0) continue;
    const int kt = (kmax + 255) / 256;
    if (big)
      DM_PLAUNCH(ctx, DM_PROF_DC, dc_secular_kernel<true>, dim3(kt, nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    else
      DM_PLAUNCH(ctx, DM_PROF_DC, dc_secular_kernel<false>, dim3(kt, nn_nodes), dim3(256), (size_t)16 * kmax + 64, ctx->stream,
                         d_dm, d_nodes, d_out);
    DM_PLAUNCH(ctx, DM_PROF_DC, dc_zhat_kernel, dim3(kt, nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    DM_PLAUNCH(ctx, DM_PROF_DC, dc_unorm_kernel, dim3(kt, nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    DM_PLAUNCH(ctx, DM_PROF_DC, dc_ubuild_kernel, dim3(kt, kmax, nn_nodes), dim3(256), 0, ctx->stream, d_dm, d_nodes, d_out);
    DM_HIP(ctx, hipGetLastError());
    std::vector<dm_gemm_desc> g;
    for (int i = 0; i < nn_nodes; ++i) {
      const dc_node& nd = nodes[i];
      const int k = ho[i].k;
      if (k == 0) continue;
      const int n = probs[nd.mat].n, nn = nd.n1 + nd.n2;
      double* Zn = (nd.flip ? ZA : ZB) + off[nd.mat];
      dm_gemm_desc d = dm_gemm_make(reinterpret_cast<const cplx*>(Uw + off[nd.mat] + (size_t)nd.lo * n), n, 1, false,
                                    Zp + off[nd.mat] + (size_t)nd.lo * n + nd.lo, n, 1, false,
                                    reinterpret_cast<cplx*>(Zn + (size_t)nd.lo * n + nd.lo), n, k, nn, k, 1.0, 0.0,
                                    nullptr, DM_GEMM_ALL_REAL);
      g.push_back(d);
    }
    DM_TRY(dm_gemm_grouped_launch(ctx, g));
  }
  // ---- results: eigenvalues back into dd, eigenvector buffer per matrix
  zfinal.assign(np, nullptr);
  std::vector<dm_cdesc> cp;
  for (int p = 0; p < np; ++p) {
    const bool inB = depth[p] > 0 && (depth[p] & 1);
    zfinal[p] = (inB ? ZB : ZA) + off[p];
    if (inB && probs[p].n > 0) cp.push_back(dm_cdesc{lamB + offn[p], dd + offn[p], sizeof(double) * probs[p].n});
  }
  DM_TRY(dm_copy_batched(ctx, cp));
  if (!dc_noscale) DM_PLAUNCH(ctx, DM_PROF_DC, dc_unscale_kernel, dim3(np), dim3(256), 0, ctx->stream, d_scm);
  return DM_OK;
}

// ===========================================================================
// driver: C (destroyed) -> evals (unsorted), W rows = eigenvectors^H
// ===========================================================================
// The batch is cut into up to four chunks (largest matrices first) that move through
// T1 -> T2 -> T3/T4 as a software pipeline: the serial QL recurrence of chunk c runs on a
// side stream while the main stream tridiagonalises chunk c+1 and back-transforms chunk c-1,
// so the only latency-bound kernel of the solver is hidden behind HBM- and MFMA-bound work.
namespace {
struct tri_side {
  hipStream_t s = nullptr;
  std::vector<hipEvent_t> ev;
};
tri_side g_side;
hipStream_t g_la_stream = nullptr;   // high-priority stream of the first-stage look-ahead
hipEvent_t side_event(size_t i) {
  while (g_side.ev.size() <= i) {
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    g_side.ev.push_back(e);
  }
  return g_side.ev[i];
}
}  // namespace

int herm_eig_tridiag(dm_ctx* ctx, const std::vector<dm_jac_herm_problem>& probs, double* evals, int evals_stride,
                     dm_eig_select* sel) {
  const int np = (int)probs.size();
  if (np == 0) return DM_OK;
  dm_ws_scope ws_scope__(ctx);  // releases on every return path
  const size_t mark = ws_scope__.mark;
  int maxn = 0;
  size_t tot = 0, totn = 0;
  std::vector<size_t> off(np), offn(np);
  for (int p = 0; p < np; ++p) {
    const size_t n = probs[p].n;
    maxn = std::max(maxn, probs[p].n);
    off[p] = tot; tot += n * n;
    offn[p] = totn; totn += n;
  }
  DM_ARG(ctx, maxn <= evals_stride);
  if (maxn == 0) return DM_OK;

  // ---- storage for every problem (all chunks are in flight at once)
  cplx* Vt = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));
  // panels: per problem 3 TNB rows of n: V, W, V again (see trd_mat)
  cplx* PP = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * 3 * TNB, 1));
  cplx* pv = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn, 1));
  cplx* xv = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn, 1));
  size_t totpc = 0, totsp = 0, totnp = 0;
  std::vector<size_t> offpc(np), offsp(np), offnp(np);
  for (int p = 0; p < np; ++p) {
    const size_t n = probs[p].n;
    offpc[p] = totpc; totpc += (n / SYG + 1) * n;
    offsp[p] = totsp; totsp += n / SYG + 1;
    offnp[p] = totnp; totnp += n / WXR + 1;
  }
  cplx* Pcv = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totpc, 1));
  double* Spv = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totsp, 1));
  double* Npv = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totnp, 1));
  cplx* abv = dm_ws_alloc_t<cplx>(ctx, (size_t)np * 2 * TNB);
  double* dd = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  double* ee = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totn, 1));
  cplx* tau = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn, 1));
  // One chunk by default: T1 is a latency-bound chain of ~3 n small launches whose cost hardly
  // depends on the batch size, so splitting the batch multiplies it (measured: 4 chunks = +40 %).
  // DM_TRIDIAG_CHUNKS > 1 enables the side-stream pipeline for experiments.
  int nch = 1;
  if (const char* e = getenv("DM_TRIDIAG_CHUNKS")) nch = std::max(1, std::min(8, atoi(e)));
  if (maxn < 256 || np < 2 * nch) nch = 1;
  const bool use_dc = nch == 1 && maxn > DC_LEAF && !getenv("DM_EIG_QL");
  // the recorded rotations (2 n^2 double2) and the QL eigenvector array only exist on the QL path
  size_t totsw = 0, totrot = 0;
  std::vector<size_t> swoff(np), rotoff(np);
  for (int p = 0; p < np; ++p) {
    const size_t n = use_dc ? 0 : probs[p].n;
    swoff[p] = totsw; totsw += 4 * n + 8;
    rotoff[p] = totrot; totrot += 2 * n * n + 8;
  }
  int* sw_dir = dm_ws_alloc_t<int>(ctx, totsw);
  int* sw_lo = dm_ws_alloc_t<int>(ctx, totsw);
  int* sw_cnt = dm_ws_alloc_t<int>(ctx, totsw);
  long long* sw_off = dm_ws_alloc_t<long long>(ctx, totsw);
  double2* rot = dm_ws_alloc_t<double2>(ctx, totrot);
  int* nsw = dm_ws_alloc_t<int>(ctx, np);
  int* stat = dm_ws_alloc_t<int>(ctx, np);
  double* Zt = dm_ws_alloc_t<double>(ctx, std::max<size_t>(use_dc ? 1 : tot, 1));
  // back-transformation in compact-WY blocks of NBB reflectors (merged from the TNB-wide panels)
  int NBB = 128;  // merged from the TNB-wide panels, whatever their width
  if (const char* e = getenv("DM_WY_BLOCK")) {
    const int v = atoi(e);
    if (v == TNB || v == 2 * TNB || v == 4 * TNB) NBB = v;
  }
  size_t tottb = 0;
  std::vector<size_t> offtb(np);
  for (int p = 0; p < np; ++p) {
    offtb[p] = tottb;
    tottb += (size_t)((probs[p].n + NBB - 1) / NBB) * NBB * NBB;
  }
  cplx* Tbig = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tottb, 1));       // T factors, NBB x NBB per block
  size_t totg = 0;
  std::vector<size_t> offg(np);
  for (int p = 0; p < np; ++p) {
    offg[p] = totg;
    totg += (size_t)(probs[p].n / TNB + 1) * TNB * TNB + (size_t)NBB * NBB;  // enough for every merge level
  }
  cplx* Gs = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totg, 1));          // Gram scratch
  cplx* Gt = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totg, 1));          // T_left * Gram scratch
  cplx* Ut = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tot, 1));           // T V^H, same layout as Vt
  cplx* W1 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * NBB, 1));
  if (!Vt || !PP || !pv || !xv || !Pcv || !Spv || !Npv || !abv || !dd || !ee || !tau || !sw_dir || !sw_lo || !sw_cnt || !sw_off || !rot || !nsw ||
      !stat || !Zt || !Tbig || !Gs || !Gt || !Ut || !W1)
    return DM_ENOMEM;
  DM_TRY(dm_fill_zero(ctx, PP, sizeof(cplx) * totn * 3 * TNB));
  DM_TRY(dm_fill_zero(ctx, Vt, sizeof(cplx) * tot));  // trd_symv only writes the non-zero part of each vector
  DM_TRY(dm_fill_zero(ctx, tau, sizeof(cplx) * totn));
  DM_TRY(dm_fill_zero(ctx, stat, sizeof(int) * np));

  // ---- chunks: by decreasing size, balanced in n^3
  std::vector<int> order(np);
  for (int p = 0; p < np; ++p) order[p] = p;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return probs[a].n > probs[b].n; });
  std::vector<std::vector<int>> chunks(nch);
  {
    double total = 0.0;
    for (int p = 0; p < np; ++p) total += std::pow((double)probs[p].n, 3);
    double acc = 0.0;
    int c = 0;
    for (int idx : order) {
      chunks[c].push_back(idx);
      acc += std::pow((double)probs[idx].n, 3);
      if (c + 1 < nch && acc >= total * (c + 1) / nch) ++c;
    }
  }
  if (nch > 1 && !g_side.s) DM_HIP(ctx, hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking));

  // ---- two-stage reduction (dm_sbr_impl.h): dense -> band (MFMA) -> tridiagonal (bulge chasing)
#if DM_TNB == 32
  bool two_stage = false;
  {
    // DM_TRD_TWOSTAGE = 1 / 0 forces / forbids it.  By default it is taken where it was measured faster than the
    // one-stage reduction with ALL eigenvectors wanted (scratch/twostage_sweep.py; with a selection it gains more): the
    // bulge chase needs n / 64 sweeps in flight per matrix to be busy, so either many matrices of a few hundred rows or
    // a few large ones — 111 x <= 1218 (configs[1]) 1.08 x, 512 x 864 1.15 x, 8 x 4000 1.17 x, 8 x 6000 1.19 x, 1 x 16384
    // 1.03 x; 32 x 1200 0.89 x, 8 x 2000 0.81 x, 1 x 8192 0.75 x stay on the one-stage path.
    int mode = -1;
    if (const char* e = getenv("DM_TRD_TWOSTAGE")) mode = atoi(e);
    if (ctx->trd_mode_override >= 0) mode = ctx->trd_mode_override;
    // (later sweep, after the launch chains were planned once per panel: 512 x 300 1.05 x, 512 x 432 1.12 x, 256 x 600
    // 1.14 x, 256 x 700 1.17 x, 512 x 864 1.19 x; 64 x 432 0.94 x, 64 x 700 0.96 x, 32 x 600 0.84 x, 16 x 1000 0.86 x)
    // (round 5: the levels of the SVD preconditioner of a configs[4] slice are 23 matrices of n = 2500 .. 3552 — below the
    // n_max >= 3500 rule, on the one-stage path at 0.45 of the HBM roofline for 13 of the 44 s of that stage: n_max >= 2400
    // with sum n >= 48 000 joins; DM_TRD_TS_MID=0 takes it out)
    static const bool ts_mid = !getenv("DM_TRD_TS_MID") || atoi(getenv("DM_TRD_TS_MID")) != 0;
    // (round 6, after the chase by band position: 200 x 128 1.15 x, 300 x 64 1.04, 432 x 64 1.08, 700 x 64 1.05, 700 x 16 1.02,
    // 1000 x 16 1.02, 1200 x 32 1.06, 2000 x 8 1.04, 16384 x 1 1.26; 300 x 16 0.93, 432 x 16 0.95, 600 x 8 0.90, 1200 x 8 0.96,
    // 2000 x 2 0.83, 3000 x 4 0.97, 4000 x 2 0.88, 8192 x 1 0.96 — profiles/r06e_twostage_sweep.txt)
    const bool pays = (maxn >= 700 && np >= 16) || (maxn >= 200 && np >= 64) || (maxn >= 2000 && np >= 8) ||
                      (maxn >= 300 && totn >= 120000) || (maxn >= 3500 && totn >= 24000) ||
                      (ts_mid && maxn >= 2400 && totn >= 48000) || maxn >= 14000;
    two_stage = use_dc && maxn > TSM && maxn > SB + 2 && (mode == 1 || (mode != 0 && pays));
  }
#else
  const bool two_stage = false;
#endif
  const int shift = two_stage ? TNB : 1;  // reflector k has its leading 1 at row k + shift
  auto nrefl_of = [&](int n) { return two_stage ? std::max(0, n - TNB - 1) : std::max(0, n - 1); };
#if DM_TNB == 32
  cplx* sbPart = nullptr;
  cplx *sbPP2 = nullptr, *sbRb = nullptr;   // look-ahead of the first stage: second set of panel buffers, R blocks
  cplx *sbPw = nullptr, *sbXt = nullptr, *sbYp = nullptr, *sbAB = nullptr, *sbVd = nullptr, *sbTau2 = nullptr, *sbM1 = nullptr,
       *sbS = nullptr;
  double* sbNp = nullptr;
  unsigned* sbProg = nullptr;
  int* sbNext = nullptr;
  std::vector<size_t> offyp(np), offvd(np), offt2(np);
  std::vector<int> sb_jb(np, 0);
  size_t totyp = 0, totvd = 0, tott2 = 0;
  if (two_stage) {
    for (int p = 0; p < np; ++p) {
      const size_t n = probs[p].n;
      offyp[p] = totyp; totyp += n / SQR + 1;
      const size_t ng = n > 1 ? (n - 1 + SBG - 1) / SBG : 0;
      sb_jb[p] = n > 1 ? (int)((n - 2) / SB + 1) : 0;
      offvd[p] = totvd; totvd += ng * sb_jb[p] * SBG * SBW;
      offt2[p] = tott2; tott2 += ng * sb_jb[p] * SBG;
    }
    sbPw = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * SB, 1));
    sbXt = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * SB, 1));
    sbYp = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totyp * SB, 1));
    sbNp = dm_ws_alloc_t<double>(ctx, std::max<size_t>(totyp * 2, 1));
    sbAB = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * SLD, 1));
    sbVd = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totvd, 1));
    sbTau2 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(tott2, 1));
    sbM1 = dm_ws_alloc_t<cplx>(ctx, (size_t)np * SB * SB);
    sbS = dm_ws_alloc_t<cplx>(ctx, (size_t)np * SB * SB);
    sbProg = dm_ws_alloc_t<unsigned>(ctx, std::max<size_t>(2 * totn, 1));  // two progress words per sweep
    // split-K partials: at most 32 slices of the 32 x n block of Y per matrix (the Gram matrices need far less)
    sbPart = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * SB * 32 + (size_t)np * SB * SB * 32, 1));
    if (!sbPart) return DM_ENOMEM;
    sbNext = dm_ws_alloc_t<int>(ctx, 2 * (size_t)np + 9);  // sweep counters, owners, queue heads, error flag
    if (maxn - SB <= SFR * SFT && getenv("DM_SB_LOOKAHEAD") && atoi(getenv("DM_SB_LOOKAHEAD")) != 0) {   // (look-ahead experiment)
      sbPP2 = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * 3 * TNB, 1));
      sbRb = dm_ws_alloc_t<cplx>(ctx, std::max<size_t>(totn * SB, 1));
      if (!sbPP2 || !sbRb) return DM_ENOMEM;
      DM_TRY(dm_fill_zero(ctx, sbPP2, sizeof(cplx) * totn * 3 * TNB));
    }
    if (!sbPw || !sbXt || !sbYp || !sbNp || !sbAB || !sbVd || !sbTau2 || !sbM1 || !sbS || !sbProg || !sbNext) return DM_ENOMEM;
  }

  auto phase_T1_two = [&](const std::vector<int>& ch) -> int {
    if (ch.empty()) return DM_OK;
    const int nc = (int)ch.size();
    int cmax = 0;
    std::vector<sb_mat> sm(nc);
    std::vector<sb_dmat> dmv(nc);
    std::vector<sb_bmat> bm(nc);
    std::vector<sb_chase_mat> cm(nc);
    for (int i = 0; i < nc; ++i) {
      const int p = ch[i];
      const size_t n = probs[p].n;
      cmax = std::max(cmax, probs[p].n);
      cplx* pp = PP + offn[p] * 3 * TNB;
      sm[i] = sb_mat{probs[p].C, probs[p].ldc, probs[p].n, Vt + off[p], pp, pp + n * TNB, pp + 2 * n * TNB,
                     sbPw + offn[p] * SB, tau + offn[p], sbYp + offyp[p] * SB, sbNp + offyp[p] * 2, (int)(n / SQR + 1), nullptr};
      dmv[i] = sb_dmat{probs[p].C, probs[p].ldc, probs[p].n};
      bm[i] = sb_bmat{probs[p].C, probs[p].ldc, probs[p].n, sbAB + offn[p] * SLD, nullptr, 0};
      cm[i] = sb_chase_mat{sbAB + offn[p] * SLD, probs[p].n, sbVd + offvd[p], sbTau2 + offt2[p], dd + offn[p], ee + offn[p],
                           sb_jb[p], sbProg + 2 * offn[p], sbNext + p, sbNext + np + p};
    }
    sb_mat* d_sm = dm_ws_upload(ctx, sm);
    sb_dmat* d_dm = dm_ws_upload(ctx, dmv);
    sb_bmat* d_bm = nullptr;   // uploaded before the band extraction (the look-ahead decides where R lives)
    sb_chase_mat* d_cmat = dm_ws_upload(ctx, cm);
    if (!d_sm || !d_dm || !d_cmat) return DM_ENOMEM;
    DM_TRY(dm_fill_zero(ctx, Tbig, sizeof(cplx) * tottb));
    {
      // the paired chase writes whole rows of the reflector array: only the slots no sweep reaches are cleared
      static const bool pairs = !getenv("DM_SB_NOPAIRS") && !getenv("DM_SB_FULLFILL");
      if (!pairs) DM_TRY(dm_fill_zero(ctx, sbVd, sizeof(cplx) * totvd));
      else DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_vd_tail_zero_kernel, dim3((cmax + SBG - 1) / SBG, nc), dim3(256), 0, ctx->stream, d_cmat);
    }
    DM_TRY(dm_fill_zero(ctx, sbTau2, sizeof(cplx) * tott2));
    DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_diag_tiles_kernel, dim3((cmax + 127) / 128, nc), dim3(256), 0, ctx->stream, d_dm);
    // ---- S1: dense -> band, one panel of SB columns at a time, all matrices in lock-step
    //
    // A panel is two chains of launches.  The "side" chain needs nothing but the panel itself: QR, Gram slices, T factor,
    // X^T = T^T V^H.  The "main" chain needs the trailing matrix: Y (slices + sum), M = V^H Y (slices), S, W, and the
    // rank-64 update.  LOOK-AHEAD (all panels of the chunk small enough for the one-workgroup QR): before the update of
    // panel k starts, the rows of the NEXT panel are copied and updated on their own (32 rows: a small product), and the
    // side chain of panel k + 1 runs on a second stream from that snapshot WHILE the main stream updates the trailing
    // matrix — the latency-bound QR hides behind the HBM-bound update.  What the concurrent chains share is kept apart:
    // the panel buffers (V, W, V) alternate between two sets, the QR puts its R block into a buffer of its own instead of
    // into rows of A that the running update still writes (the band extraction collects it from there).
    static const bool nofuse = getenv("DM_SB_NOFUSE") != nullptr;
    // MEASURED, OFF BY DEFAULT (DM_SB_LOOKAHEAD=1 turns it on): correct (scratch/twostage_check.py, the GPU suite), but the
    // QR does not actually overlap — its 512-thread workgroups at 180 VGPRs need two of a CU's three update workgroups to
    // leave at the same time, and every slot an update tile frees is taken by the next tile first, stream priority or
    // not: the QR ran 755 us instead of 205, finishing as late as without look-ahead (configs[1]: 135.7-136.3 ms per step
    // against 133.9-134.2, the snapshot product being extra work).
    static const bool la_on = getenv("DM_SB_LOOKAHEAD") && atoi(getenv("DM_SB_LOOKAHEAD")) != 0;
    const bool la = la_on && !nofuse && sbPP2 && sbRb && cmax - SB <= SFR * SFT && cmax - 2 * SB >= 2;
    sb_mat* d_sm2 = nullptr;   // the descriptors with the second set of panel buffers (odd panels)
    if (la) {
      std::vector<sb_mat> sm_a(sm), sm_b(sm);
      for (int i = 0; i < nc; ++i) {
        const int p = ch[i];
        const size_t n = probs[p].n;
        cplx* pp2 = sbPP2 + offn[p] * 3 * TNB;
        sm_a[i].Rb = sbRb + offn[p] * SB;
        sm_b[i].Rb = sm_a[i].Rb;
        sm_b[i].Vp = pp2; sm_b[i].Wp = pp2 + n * TNB; sm_b[i].Vp2 = pp2 + 2 * n * TNB;
      }
      d_sm = dm_ws_upload(ctx, sm_a);
      d_sm2 = dm_ws_upload(ctx, sm_b);
      if (!d_sm || !d_sm2) return DM_ENOMEM;
      // HIGH priority: the one-workgroup-per-matrix QR must get CUs while the update's ten thousand tiles keep arriving —
      // on an ordinary stream its workgroups were only placed as the update drained (755 us instead of 205)
      if (!g_la_stream) {
        int lo = 0, hi = 0;
        DM_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        DM_HIP(ctx, hipStreamCreateWithPriority(&g_la_stream, hipStreamNonBlocking, hi));
      }
    }
    auto pp_of = [&](int p, int k0) { return ((la && ((k0 / SB) & 1)) ? sbPP2 : PP) + offn[p] * 3 * TNB; };
    auto sm_of = [&](int k0) { return (la && ((k0 / SB) & 1)) ? d_sm2 : d_sm; };
    static const bool nosplit = getenv("DM_SB_NOSPLITK") != nullptr;
    cplx* part_y = sbPart;                               // per matrix: SY x (32 x n) at offn * SB * 32
    cplx* part_g = sbPart + totn * SB * 32;              // per matrix: 32 x (32 x 32)
    struct panel_split { int nact, SG, SY; };
    auto split_of = [&](int k0) {
      const int i0 = k0 + SB, a0 = i0 & ~127;
      int nact = 0, ytiles = 0;
      for (int p : ch) {
        if (probs[p].n - i0 < 2) continue;
        ++nact;
        ytiles += (probs[p].n - a0 + 127) / 128;
      }
      const int kmax = cmax - i0;
      // split-K: the products with K = trailing size have few output tiles (one 32 x 32 tile per matrix for the Gram
      // matrices, one 32 x 128 tile per 128 columns of Y): cut K so that a launch carries ~1000 tiles
      auto slices_for = [&](int tiles) { return std::max(1, std::min(30, std::min((1024 + tiles - 1) / tiles, (kmax + 127) / 128))); };
      return panel_split{nact, (nosplit || nact == 0) ? 1 : slices_for(nact), (nosplit || nact == 0) ? 1 : slices_for(std::max(ytiles, 1))};
    };
    // side chain of panel k0: Gram slices (summed by the T-factor kernel), T, X^T = T^T V^H
    struct side_set { dm_gemm_plan pg, px; std::vector<tf_mat> tf; };
    auto build_side = [&](int k0, side_set& S_) -> int {
      const int i0 = k0 + SB;
      const panel_split sp = split_of(k0);
      std::vector<dm_gemm_desc> gg, gx;
      for (int p : ch) {
        const int n = probs[p].n;
        const int m = n - i0;
        if (m < 2) continue;
        const int kb = std::min(SB, m - 1);
        cplx* Vp = pp_of(p, k0);
        cplx* Xt = sbXt + offn[p] * SB;
        const cplx* Vb = Vt + off[p] + (size_t)k0 * n + i0;
        cplx* G = Gs + offg[p] + (size_t)(k0 / TNB) * TNB * TNB;
        cplx* T = Tbig + offtb[p] + (size_t)(k0 / NBB) * NBB * NBB + (size_t)(k0 % NBB) * NBB + (k0 % NBB);
        cplx* pg = part_g + (size_t)p * SB * SB * 32;
        int gram_slices = 0;
        if (sp.SG == 1) {
          gg.push_back(dm_gemm_make(Vb, n, 1, true, Vb, 1, n, false, G, TNB, kb, kb, m));
        } else {
          const int kc = (m + sp.SG - 1) / sp.SG;
          int ns = 0;
          for (int kk = 0; kk < m; kk += kc, ++ns)
            gg.push_back(dm_gemm_make(Vb + kk, n, 1, true, Vb + kk, 1, n, false, pg + (size_t)ns * kb * kb, kb, kb, kb, std::min(kc, m - kk)));
          gram_slices = ns;   // summed by the T-factor kernel
        }
        S_.tf.push_back(tf_mat{G, tau + offn[p] + k0, T, kb, NBB, gram_slices ? pg : nullptr, gram_slices});
        gx.push_back(dm_gemm_make(T, 1, NBB, false, Vp + i0, n, 1, false, Xt + i0, n, SB, m, SB));   // Xt = T^T Vp (SB x m)
      }
      DM_TRY(dm_gemm_plan_build(gg, S_.pg));
      DM_TRY(dm_gemm_plan_build(gx, S_.px));
      return DM_OK;
    };
    // main chain of panel k0
    struct main_set { dm_gemm_plan py1, py2, pm, pw, ph; std::vector<sb_sum_desc> sy; std::vector<sb_s_desc> ssv; };
    auto build_main = [&](int k0, main_set& M_) -> int {
      const int i0 = k0 + SB, a0 = i0 & ~127;
      const panel_split sp = split_of(k0);
      std::vector<dm_gemm_desc> gy1, gy2, gm, gw, gh;
      for (int p : ch) {
        const int n = probs[p].n;
        const int m = n - i0;
        if (m < 2) continue;
        const int lda = probs[p].ldc;
        cplx* C = probs[p].C;
        cplx* pp = pp_of(p, k0);
        cplx* Vp = pp;
        cplx* Wp = pp + (size_t)n * TNB;
        cplx* Xt = sbXt + offn[p] * SB;
        cplx* T = Tbig + offtb[p] + (size_t)(k0 / NBB) * NBB * NBB + (size_t)(k0 % NBB) * NBB + (k0 % NBB);
        cplx* pg = part_g + (size_t)p * SB * SB * 32;
        // Yt = Xt A22 by 128-column blocks: stored part (rows >= block start, whole diagonal block) + mirrored part
        cplx* py = part_y + offn[p] * SB * 32;
        size_t pyoff = 0;
        // BLOCK-PAIR order (split products only, at most 30 blocks per side): K is cut at the 128-boundaries of the matrix,
        // so every piece reads ONE 128 x 128 block of the stored triangle — and the two pieces that read the same block
        // (the stored part of column block R over the columns of block C, the mirrored part of column block C over the
        // rows of block R) are emitted next to each other: they run on the same XCD at the same time (the tile list is
        // dealt to the XCDs in contiguous runs) and the second one finds the block in L2.  Uniform K = 128 tiles instead of
        // ragged slices; the trailing matrix comes from HBM once per panel for this product instead of twice.
        static const bool ypair_off = getenv("DM_SB_YPAIR") && atoi(getenv("DM_SB_YPAIR")) == 0;
        const int nblk = (n - a0 + 127) / 128;
        if (sp.SY > 1 && !ypair_off && nblk <= 30) {
          // column block b covers [lo(b), hi(b)); slots of block b: stored pieces over the blocks c >= b (slot c - b),
          // then mirrored pieces over the blocks r < b (slot (nblk - b) + r)
          auto lo = [&](int b) { return std::max(a0 + b * 128, i0); };
          auto hi = [&](int b) { return std::min(a0 + (b + 1) * 128, n); };
          std::vector<size_t> base(nblk);
          for (int b = 0; b < nblk; ++b) {
            base[b] = pyoff;
            pyoff += (size_t)nblk * SB * (hi(b) - lo(b));
          }
          for (int r = 0; r < nblk; ++r)
            for (int c = r; c < nblk; ++c) {
              const int wr = hi(r) - lo(r), wc = hi(c) - lo(c);
              // stored: Y[:, block r] += Xt[:, block c] . C[rows of r, columns of c]^T
              gy1.push_back(dm_gemm_make(Xt + lo(c), n, 1, false, C + (size_t)lo(r) * lda + lo(c), 1, lda, false,
                                         py + base[r] + (size_t)(c - r) * SB * wr, wr, SB, wr, wc));
              // mirrored: Y[:, block c] += Xt[:, block r] . conj(C[rows of r, columns of c])
              if (c > r)
                gy1.push_back(dm_gemm_make(Xt + lo(r), n, 1, false, C + (size_t)lo(r) * lda + lo(c), lda, 1, true,
                                           py + base[c] + (size_t)((nblk - c) + r) * SB * wc, wc, SB, wc, wr));
            }
          for (int b = 0; b < nblk; ++b)
            M_.sy.push_back(sb_sum_desc{Wp + lo(b), py + base[b], nblk, SB, hi(b) - lo(b), n, 1.0, 0.0});
        } else
        for (int cb = a0; cb < n; cb += 128) {
          const int c_lo = std::max(cb, i0), c_hi = std::min(cb + 128, n);
          if (c_hi <= c_lo) continue;
          const int wN = c_hi - c_lo;
          if (sp.SY == 1) {
            gy1.push_back(dm_gemm_make(Xt + c_lo, n, 1, false, C + (size_t)c_lo * lda + c_lo, 1, lda, false, Wp + c_lo, n, SB, wN,
                                       n - c_lo));
            if (c_lo > i0)
              gy2.push_back(dm_gemm_make(Xt + i0, n, 1, false, C + (size_t)i0 * lda + c_lo, lda, 1, true, Wp + c_lo, n, SB, wN,
                                         c_lo - i0, 1.0, 1.0));
          } else {
            const int kc = std::max(128, ((m + sp.SY - 1) / sp.SY + 127) & ~127);
            cplx* pb = py + pyoff;
            int ns = 0;
            for (int kk = c_lo; kk < n; kk += kc, ++ns)   // stored part: rows kk .. of the columns [c_lo, c_hi)
              gy1.push_back(dm_gemm_make(Xt + kk, n, 1, false, C + (size_t)c_lo * lda + kk, 1, lda, false, pb + (size_t)ns * SB * wN, wN,
                                         SB, wN, std::min(kc, n - kk)));
            for (int kk = i0; kk < c_lo; kk += kc, ++ns)  // mirrored part: rows i0 .. c_lo of the transposed block
              gy1.push_back(dm_gemm_make(Xt + kk, n, 1, false, C + (size_t)kk * lda + c_lo, lda, 1, true, pb + (size_t)ns * SB * wN, wN,
                                         SB, wN, std::min(kc, c_lo - kk)));
            M_.sy.push_back(sb_sum_desc{Wp + c_lo, pb, ns, SB, wN, n, 1.0, 0.0});
            pyoff += (size_t)ns * SB * wN;
          }
        }
        cplx* M1 = sbM1 + (size_t)p * SB * SB;
        cplx* S = sbS + (size_t)p * SB * SB;
        // M1 = V^H Y, S = T^H M1, W = Y - V S / 2  (row-stored: Wp += -1/2 S^T Vp)
        int m1_slices = 0;
        if (sp.SG == 1) {
          gm.push_back(dm_gemm_make(Vp + i0, n, 1, true, Wp + i0, 1, n, false, M1, SB, SB, SB, m));
        } else {
          const int kc = (m + sp.SG - 1) / sp.SG;
          int ns = 0;
          for (int kk = 0; kk < m; kk += kc, ++ns)
            gm.push_back(dm_gemm_make(Vp + i0 + kk, n, 1, true, Wp + i0 + kk, 1, n, false, pg + (size_t)ns * SB * SB, SB, SB, SB,
                                      std::min(kc, m - kk)));
          m1_slices = ns;
        }
        M_.ssv.push_back(sb_s_desc{T, NBB, m1_slices ? pg : M1, m1_slices, S});   // S = T^H (sum of the slices of M1)
        gw.push_back(dm_gemm_make(S, 1, SB, false, Vp + i0, n, 1, false, Wp + i0, n, SB, m, SB, -0.5, 1.0));
        // A22 -= V W^H + W V^H on the blocks on or above the diagonal (128-aligned origin a0)
        gh.push_back(dm_gemm_make(pp + a0, 1, n, false, pp + (size_t)n * TNB + a0, n, 1, true, C + (size_t)a0 * lda + a0, lda,
                                  n - a0, n - a0, 2 * TNB, -1.0, 1.0, nullptr, DM_GEMM_UPPER | DM_GEMM_UPPER128));
      }
      DM_TRY(dm_gemm_plan_build(gy1, M_.py1));
      DM_TRY(dm_gemm_plan_build(gy2, M_.py2));
      DM_TRY(dm_gemm_plan_build(gm, M_.pm));
      DM_TRY(dm_gemm_plan_build(gw, M_.pw));
      DM_TRY(dm_gemm_plan_build(gh, M_.ph));
      return DM_OK;
    };
    // look-ahead: the update of panel k0 applied to the snapshot of the rows of panel k0 + SB
    //   P[q][c] -= sum_j V[i0 + q][j] conj(W[c][j]) + W[i0 + q][j] conj(V[c][j]),   c >= i0 + SB
    auto build_strip = [&](int k0, dm_gemm_plan& pl) -> int {
      const int i0 = k0 + SB, i1 = i0 + SB;
      std::vector<dm_gemm_desc> g;
      for (int p : ch) {
        const int n = probs[p].n;
        if (n - i1 < 2) continue;
        cplx* pp = pp_of(p, k0);
        cplx* Pw = sbPw + offn[p] * SB;
        g.push_back(dm_gemm_make(pp + i0, 1, n, false, pp + (size_t)n * TNB + i1, n, 1, true, Pw + i1, n, SB, n - i1, 2 * TNB, -1.0,
                                 1.0));
      }
      return dm_gemm_plan_build(g, pl);
    };
    auto qr_flops = [&](int k0) {
      double fl = 0.0;  // Householder QR of an m x SB panel: 2 SB^2 (m - SB / 3) complex multiply-adds
      for (int p : ch) {
        const double m = probs[p].n - k0 - SB;
        if (m >= 2) fl += 8.0 * 2.0 * SB * SB * std::max(m - SB / 3.0, 1.0);
      }
      return fl;
    };
    struct stream_swap {   // grouped launches go to ctx->stream: point it at the side stream for a scope
      dm_ctx* c; hipStream_t keep;
      stream_swap(dm_ctx* c_, hipStream_t s) : c(c_), keep(c_->stream) { if (s) c->stream = s; }
      ~stream_swap() { c->stream = keep; }
    };
    static bool larft_attr = false;
    const size_t larft_lds = 2 * sizeof(cplx) * TNB * (TNB + 1);
    if (!larft_attr) {
      DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(larft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)larft_lds));
      larft_attr = true;
    }
    // QR (one-workgroup kernel, or the launched one) + side chain of panel k0 on stream `st` (nullptr: the main stream)
    auto run_side = [&](int k0, const side_set& S_, const char* d_pg, const char* d_px, const tf_mat* d_tf, hipStream_t st,
                        int snap) -> int {
      const int i0 = k0 + SB, a0 = i0 & ~127;
      stream_swap sw(ctx, st);
      if (cmax - i0 <= SFR * SFT && !nofuse) {
        // panels that fit the registers of one workgroup per matrix: the whole QR in one launch
        dm_prof_scope ps(ctx, DM_PROF_SB_PANEL, qr_flops(k0));
        hipLaunchKernelGGL(sb_panel_fused_kernel, dim3(nc), dim3(SFT), 0, ctx->stream, sm_of(k0), k0, a0, snap);
      } else {
        hipLaunchKernelGGL(sb_panel_load_kernel, dim3((cmax - a0 + 255) / 256, nc), dim3(256), 0, ctx->stream, d_sm, k0, a0);
        const int nchmax = (cmax - i0 + SQR - 1) / SQR;
        for (int q = 0; q <= SB; ++q) {
          hipLaunchKernelGGL(sb_qr_update_kernel, dim3(nchmax, nc), dim3(256), 0, ctx->stream, d_sm, k0, q);
          if (q < SB) hipLaunchKernelGGL(sb_qr_dots_kernel, dim3(nchmax, nc), dim3(256), 0, ctx->stream, d_sm, k0, q);
        }
      }
      if (S_.tf.empty()) return DM_OK;
      DM_TRY(dm_gemm_plan_run(ctx, S_.pg, d_pg));
      // T factor of the panel (zlarft from the Gram matrix), straight into the slot the back-transformation reads
      DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, larft_kernel, dim3((unsigned)S_.tf.size()), dim3(256), larft_lds, ctx->stream, d_tf);
      DM_TRY(dm_gemm_plan_run(ctx, S_.px, d_px));
      return DM_OK;
    };
    side_set side_next;        // look-ahead: the side chain of the panel after the current one (built one iteration early)
    bool side_next_ready = false;
    for (int k0 = 0; cmax - k0 - SB >= 2; k0 += SB) {
      const int i0 = k0 + SB;
      if (split_of(k0).nact == 0) continue;
      const bool have_next = la && cmax - (k0 + SB) - SB >= 2 && split_of(k0 + SB).nact > 0;
      side_set side_cur;
      const bool run_cur_side = !(la && side_next_ready);   // else it already ran on the side stream
      if (run_cur_side) DM_TRY(build_side(k0, side_cur));
      main_set mn;
      DM_TRY(build_main(k0, mn));
      side_set side_new;
      dm_gemm_plan pstrip;
      if (have_next) {
        DM_TRY(build_side(k0 + SB, side_new));
        DM_TRY(build_strip(k0, pstrip));
      }
      // every descriptor of the iteration travels in ONE staged copy: the grouped products as plans, the lists of slice
      // sums, the S and the T-factor descriptors behind them
      dm_gemm_plan extra;   // not a product: raw arrays carried by the same upload
      size_t o_sy, o_ss, o_tf, o_tf2;
      {
        auto put = [&](const void* src, size_t bytes) {
          const size_t o = (extra.blob.size() + 15) & ~size_t(15);
          extra.blob.resize(o + bytes);
          if (bytes) std::memcpy(extra.blob.data() + o, src, bytes);
          return o;
        };
        o_sy = put(mn.sy.data(), mn.sy.size() * sizeof(sb_sum_desc));
        o_ss = put(mn.ssv.data(), mn.ssv.size() * sizeof(sb_s_desc));
        o_tf = put(side_cur.tf.data(), side_cur.tf.size() * sizeof(tf_mat));
        o_tf2 = put(side_new.tf.data(), side_new.tf.size() * sizeof(tf_mat));
        if (extra.blob.empty()) extra.blob.resize(16);
      }
      std::vector<const char*> dv;
      DM_TRY(dm_gemm_plans_upload(ctx, {&side_cur.pg, &side_cur.px, &mn.py1, &mn.py2, &mn.pm, &mn.pw, &mn.ph, &side_new.pg,
                                        &side_new.px, &pstrip, &extra}, dv));
      const char* d_extra = dv[10];
      auto launch_sums = [&](const std::vector<sb_sum_desc>& v, size_t o) -> int {
        if (v.empty()) return DM_OK;
        int mx = 0;
        for (const auto& d : v) mx = std::max(mx, d.rows * d.cols);
        DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_sum_partials_kernel, dim3((mx + 255) / 256, (unsigned)v.size()), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const sb_sum_desc*>(d_extra + o));
        return DM_OK;
      };
      if (run_cur_side) {
        DM_TRY(run_side(k0, side_cur, dv[0], dv[1], reinterpret_cast<const tf_mat*>(d_extra + o_tf), nullptr, 0));
      } else {
        DM_HIP(ctx, hipStreamWaitEvent(ctx->stream, side_event(18 + ((k0 / SB) & 1)), 0));   // the side chain of this panel
      }
      DM_TRY(dm_gemm_plan_run(ctx, mn.py1, dv[2]));
      DM_TRY(dm_gemm_plan_run(ctx, mn.py2, dv[3]));
      DM_TRY(launch_sums(mn.sy, o_sy));
      DM_TRY(dm_gemm_plan_run(ctx, mn.pm, dv[4]));
      if (!mn.ssv.empty())
        DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_s_kernel, dim3((unsigned)mn.ssv.size()), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const sb_s_desc*>(d_extra + o_ss));
      DM_TRY(dm_gemm_plan_run(ctx, mn.pw, dv[5]));
      side_next_ready = false;
      if (have_next) {
        // snapshot of the next panel's rows + this panel's update of them, then its side chain on the second stream
        DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_strip_copy_kernel, dim3((cmax - i0 - SB + 255) / 256, SB, nc), dim3(256), 0, ctx->stream, d_sm, i0);
        DM_TRY(dm_gemm_plan_run(ctx, pstrip, dv[9]));
        hipEvent_t e_go = side_event(16 + ((k0 / SB) & 1));
        DM_HIP(ctx, hipEventRecord(e_go, ctx->stream));
        DM_HIP(ctx, hipStreamWaitEvent(g_la_stream, e_go, 0));
        DM_TRY(run_side(k0 + SB, side_new, dv[7], dv[8], reinterpret_cast<const tf_mat*>(d_extra + o_tf2), g_la_stream, 1));
        DM_HIP(ctx, hipEventRecord(side_event(18 + (((k0 + SB) / SB) & 1)), g_la_stream));
        side_next_ready = true;
      }
      DM_TRY(dm_gemm_plan_run(ctx, mn.ph, dv[6]));
    }
    // ---- S2: band -> tridiagonal
    const char* dump = getenv("DM_SB_DUMP");  // debugging aid: the band and the tridiagonal of every matrix to files
    auto dump_arr = [&](const char* suffix, const void* src, size_t bytes) -> int {
      std::vector<char> h(bytes);
      DM_TRY(dm_download(ctx, h.data(), src, bytes));
      const std::string fn = std::string(dump) + suffix;
      if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
      return DM_OK;
    };
    {
      size_t maxel = (size_t)cmax * SLD;
      if (la)
        for (int i = 0; i < nc; ++i) {
          const int p = ch[i], n = probs[p].n;
          bm[i].Rb = sbRb + offn[p] * SB;
          bm[i].nrb = n - SB - 2 >= 0 ? ((n - SB - 2) / SB) * SB + SB : 0;   // rows of the panels this matrix went through
        }
      d_bm = dm_ws_upload(ctx, bm);
      if (!d_bm) return DM_ENOMEM;
      DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, sb_band_extract_kernel, dim3((unsigned)((maxel + 255) / 256), nc), dim3(256), 0, ctx->stream, d_bm);
      if (dump) DM_TRY(dump_arr(".band", sbAB, sizeof(cplx) * totn * SLD));
      // ---- the chase BY BAND POSITION (sb_chase_pos_kernel; DM_SB_CHASE=pairs keeps the sweep-owning pairs below): one
      // workgroup per (matrix, group of SB_POS_NP positions), largest matrix first, groups left to right; a matrix whose
      // groups could not all be resident at once stays on the sweep-owning kernel (its hand-offs need no co-residency)
      static const bool by_pos = !(getenv("DM_SB_CHASE") && strcmp(getenv("DM_SB_CHASE"), "pairs") == 0) && !getenv("DM_SB_NOPAIRS");
      int pos_cap = 224;   // workgroups of one matrix that may have to be resident together (one per CU, some CUs left to others)
      if (const char* e = getenv("DM_SB_POS_CAP")) pos_cap = std::max(1, atoi(e));
      bool pos_fits = by_pos;
      for (int i = 0; i < nc && pos_fits; ++i)
        if (cm[i].n >= 2 && (cm[i].jb + SB_POS_NP - 1) / SB_POS_NP > pos_cap) pos_fits = false;
      if (pos_fits) {
        std::vector<int> order(nc);
        for (int i = 0; i < nc; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cm[a].n > cm[b].n; });
        std::vector<int2> ent;
        double fl = 0.0;
        for (int i : order) {
          const int n = cm[i].n;
          if (n < 1) continue;
          const int ng = n >= 2 ? (cm[i].jb + SB_POS_NP - 1) / SB_POS_NP : 1;
          for (int g = 0; g < ng; ++g) ent.push_back(make_int2(i, g));
          fl += 8.0 * 6.0 * SB * SB * ((double)n * n / (2.0 * SB));
        }
        if (!ent.empty()) {
          sb_pos_ctl pc;
          int2* d_ent = dm_ws_upload(ctx, ent);
          char* mail = dm_ws_alloc_t<char>(ctx, ent.size() * SB_POS_MAIL);
          if (!d_ent || !mail) return DM_ENOMEM;
          DM_TRY(dm_fill_zero(ctx, mail, ent.size() * SB_POS_MAIL));            // tags 0: no sweep has posted
          DM_TRY(dm_fill_zero(ctx, sbNext + 2 * (size_t)np, sizeof(int) * 9));  // ticket counter, error flag
          pc.ticket = sbNext + 2 * (size_t)np;
          pc.err = sbNext + 2 * (size_t)np + 8;
          pc.ent = d_ent;
          pc.nent = (int)ent.size();
          pc.mail = mail;
          {
            dm_prof_scope ps(ctx, DM_PROF_SB_CHASE, fl);
            hipLaunchKernelGGL(sb_chase_pos_kernel, dim3((unsigned)ent.size()), dim3(128 * SB_POS_NP), 0, ctx->stream, d_cmat, pc);
          }
          int herr = 0;
          DM_TRY(dm_download(ctx, &herr, pc.err, sizeof(int)));
          if (herr) {
            ctx->err = "bulge chase (by position): a wave waited for its neighbour for too long";
            return 2000 + herr;
          }
        }
        if (dump) {
          DM_TRY(dump_arr(".d", dd, sizeof(double) * totn));
          DM_TRY(dump_arr(".e", ee, sizeof(double) * totn));
        }
        DM_HIP(ctx, hipGetLastError());
        return DM_OK;
      }
      // One persistent launch: per-XCD queues of matrix ids; a matrix gets as many entries (= workgroups) as its
      // pipeline of sweeps can keep busy (sweep s + 1 trails sweep s by two blocks: n / (2 SB) sweeps in flight).
      constexpr int NW = 8, NP = 4;
      static const bool pairs = !getenv("DM_SB_NOPAIRS");  // two waves per sweep (E chain + D updates) or one
      const int per_wg = pairs ? NP : NW;                  // sweeps a workgroup runs at a time
      std::vector<int> order(nc);
      for (int i = 0; i < nc; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cm[a].n > cm[b].n; });
      int wgmax = 32;
      if (const char* e = getenv("DM_SB_WGPM")) wgmax = std::max(1, atoi(e));
      std::vector<std::vector<int>> qs(8);
      double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // workgroups are resident for the whole launch and stay with a matrix until its sweeps are taken: hand out at most
      // as many entries as there are workgroups (one per CU), in proportion to the sweeps each matrix can keep in flight
      double want = 0.0;
      for (int i : order)
        if (cm[i].n >= 1) want += std::max(1.0, (double)cm[i].n / (2 * SB * per_wg));
      const double scale = want > 256.0 ? 256.0 / want : 1.0;
      for (int i : order) {
        const int n = cm[i].n;
        if (n < 1) continue;
        int k = std::max(1, std::min(wgmax, (int)(scale * n / (2 * SB * per_wg) + 0.999)));
        if (scale < 1.0) k = std::max(1, (int)(scale * n / (2 * SB * per_wg) + 0.5));
        int q = 0;
        for (int t = 1; t < 8; ++t)
          if (load[t] < load[q]) q = t;
        load[q] += (double)n * n;
        for (int t = 0; t < k; ++t) qs[q].push_back(i);
      }
      sb_chase_ctl ctl;
      std::vector<int> qent;
      for (int q = 0; q < 8; ++q) {
        ctl.qoff[q] = (int)qent.size();
        qent.insert(qent.end(), qs[q].begin(), qs[q].end());
      }
      ctl.qoff[8] = (int)qent.size();
      int* d_qent = dm_ws_upload(ctx, qent);
      if (!d_qent) return DM_ENOMEM;
      ctl.qent = d_qent;
      ctl.qhead = sbNext + 2 * (size_t)np;
      DM_TRY(dm_fill_zero(ctx, sbProg, sizeof(unsigned) * 2 * totn));
      DM_TRY(dm_fill_zero(ctx, sbNext, sizeof(int) * np));
      DM_HIP(ctx, hipMemsetAsync(sbNext + np, 0xff, sizeof(int) * np, ctx->stream));
      DM_TRY(dm_fill_zero(ctx, sbNext + 2 * (size_t)np, sizeof(int) * 9));
      ctl.err = sbNext + 2 * (size_t)np + 8;
      ctl.dbg = nullptr;
      if (dump) {
        ctl.dbg = dm_ws_alloc_t<unsigned long long>(ctx, 2 * (size_t)cmax + 2);
        if (!ctl.dbg) return DM_ENOMEM;
        DM_TRY(dm_fill_zero(ctx, ctl.dbg, sizeof(unsigned long long) * (2 * (size_t)cmax + 2)));
      }
      const int nwg = 256;  // one per CU: a matrix is served by the workgroups of ONE XCD, whichever claims it first
      {
        // one task = E <- H^H (E H) and the two-sided update of the Hermitian D: ~6 SB^2 complex multiply-adds
        double fl = 0.0;
        for (int i = 0; i < nc; ++i) {
          const double n = cm[i].n;
          fl += 8.0 * 6.0 * SB * SB * (n * n / (2.0 * SB));
        }
        dm_prof_scope ps(ctx, DM_PROF_SB_CHASE, fl);
        if (pairs) hipLaunchKernelGGL((sb_chase2_kernel<NP>), dim3(nwg), dim3(128 * NP), 0, ctx->stream, d_cmat, ctl);
        else hipLaunchKernelGGL((sb_chase_kernel<NW>), dim3(nwg), dim3(64 * NW), 0, ctx->stream, d_cmat, ctl);
      }
      {
        int herr = 0;  // (the eigenvalue selection synchronises right after this stage anyway)
        DM_TRY(dm_download(ctx, &herr, sbNext + 2 * (size_t)np + 8, sizeof(int)));
        if (herr) {
          if (dump) {
            DM_TRY(dump_arr(".dbg", ctl.dbg, sizeof(unsigned long long) * (2 * (size_t)cmax + 2)));
            DM_TRY(dump_arr(".prog", sbProg, sizeof(unsigned) * 2 * totn));
            DM_TRY(dump_arr(".next", sbNext, sizeof(int) * (2 * (size_t)np + 9)));
          }
          ctx->err = "bulge chase: a sweep waited for its predecessor for too long";
          return 2000;
        }
      }
      if (dump) {
        DM_TRY(dump_arr(".dbg", ctl.dbg, sizeof(unsigned long long) * (2 * (size_t)cmax + 2)));
        DM_TRY(dump_arr(".d", dd, sizeof(double) * totn));
        DM_TRY(dump_arr(".e", ee, sizeof(double) * totn));
      }
    }
    DM_HIP(ctx, hipGetLastError());
    return DM_OK;
  };

  // X <- Q2 X (X = probs[p].C, n x ncolv[p]): workgroups of NW column slabs, large matrices first
  auto apply_q2 = [&](const std::vector<int>& ch, const std::vector<int>& ncolv) -> int {
    constexpr int NW = 4;
    std::vector<int> order(ch);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return probs[a].n > probs[b].n; });
    // 16 columns per wave (four lanes per column).  Eight lanes per column (twice the waves, five instead of nine rows
    // of a reflector per lane) were measured slower in both regimes — configs[1] batch 11.3 against 9.2 ms, one matrix of
    // 32 576 rows 8.8 against 7.6 s: the fixed cost per reflector (fetch, fold, scale) weighs more than the extra waves
    // hide — and stay behind DM_SB_Q2_LPC=8.
    int lpc = 4;
    if (const char* e = getenv("DM_SB_Q2_LPC")) lpc = atoi(e) == 8 ? 8 : 4;
    const int ncw = 64 / lpc;
    std::vector<sb_q2_mat> qm;
    std::vector<int2> wgs;
    for (int p : order) {
      if (probs[p].n < 2 || ncolv[p] <= 0) continue;
      const int mi = (int)qm.size();
      qm.push_back(sb_q2_mat{sbVd + offvd[p], sbTau2 + offt2[p], sb_jb[p], probs[p].C, probs[p].ldc, probs[p].n, ncolv[p], 0});
      const int nslab = (ncolv[p] + ncw - 1) / ncw;
      for (int s0 = 0; s0 < nslab; s0 += NW) wgs.push_back(make_int2(mi, s0));
    }
    if (wgs.empty()) return DM_OK;
    if (getenv("DM_TRD_SIZES")) {
      fprintf(stderr, "[apply_q2] n:ncol");
      for (const auto& q : qm) fprintf(stderr, " %d:%d", q.n, q.ncol);
      fprintf(stderr, " -> %zu workgroups, %d lanes per column\n", wgs.size(), lpc);
    }
    sb_q2_mat* d_qm = dm_ws_upload(ctx, qm);
    int2* d_wgs = dm_ws_upload(ctx, wgs);
    if (!d_qm || !d_wgs) return DM_ENOMEM;
    const size_t lds = sizeof(cplx) * (2 * SBG * SBW + 2 * SBG);
    static bool attr = false;
    if (!attr) {
      DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sb_q2_apply_kernel<NW, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sb_q2_apply_kernel<NW, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    double fl = 0.0;  // n^2 / (2 SB) reflectors of length SB on ncol columns: 2 SB complex multiply-adds per column each
    for (const auto& q : qm) fl += 8.0 * (double)q.n * q.n * q.ncol;
    dm_prof_scope ps(ctx, DM_PROF_SB_Q2, fl);
    if (lpc == 8)
      hipLaunchKernelGGL((sb_q2_apply_kernel<NW, 8>), dim3((unsigned)wgs.size()), dim3(64 * NW), lds, ctx->stream, d_qm, d_wgs);
    else
      hipLaunchKernelGGL((sb_q2_apply_kernel<NW, 4>), dim3((unsigned)wgs.size()), dim3(64 * NW), lds, ctx->stream, d_qm, d_wgs);
    DM_HIP(ctx, hipGetLastError());
    return DM_OK;
  };
#else
  auto apply_q2 = [&](const std::vector<int>&, const std::vector<int>&) -> int { return DM_OK; };
#endif

  bool small_path = false;  // set by phase_T1 when the chunk went through trd_small (explicit Q in Ut)
  auto phase_T1 = [&](const std::vector<int>& ch) -> int {
    if (ch.empty()) return DM_OK;
    const int nc = (int)ch.size();
    int cmax = 0;
    std::vector<trd_mat> tm(nc);
    for (int i = 0; i < nc; ++i) {
      const int p = ch[i];
      cmax = std::max(cmax, probs[p].n);
      cplx* pp = PP + offn[p] * 3 * TNB;
      const size_t n = probs[p].n;
      tm[i] = trd_mat{probs[p].C, probs[p].ldc, probs[p].n, Vt + off[p], pp, pp + n * TNB, pp + 2 * n * TNB,
                      xv + offn[p], pv + offn[p], Pcv + offpc[p], Spv + offsp[p], Npv + offnp[p],
                      abv + (size_t)p * 2 * TNB, dd + offn[p], ee + offn[p], tau + offn[p]};
    }
    trd_mat* d_tm = dm_ws_upload(ctx, tm);
    if (!d_tm) return DM_ENOMEM;
    if (cmax <= TSM && !getenv("DM_TRD_NOSMALL")) {
      // small matrices: tridiagonal form and the explicit Q in one launch (Q into Ut, leading dimension n)
      std::vector<trs_mat> sm(nc);
      for (int i = 0; i < nc; ++i) {
        const int p = ch[i];
        sm[i] = trs_mat{probs[p].C, probs[p].ldc, probs[p].n, Ut + off[p], probs[p].n, dd + offn[p], ee + offn[p]};
      }
      trs_mat* d_sm = dm_ws_upload(ctx, sm);
      if (!d_sm) return DM_ENOMEM;
      const size_t lds = sizeof(cplx) * (TSM * TSP + 7 * TSM) + sizeof(double) * 3 * (TST / 64);
      static bool attr = false;
      if (!attr) {
        DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trd_small_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
      }
      DM_PLAUNCH(ctx, DM_PROF_TRD_SMALL, trd_small_kernel, dim3(nc), dim3(TST), lds, ctx->stream, d_sm);
      DM_HIP(ctx, hipGetLastError());
      small_path = true;
      return DM_OK;
    }
    for (int k0 = 0; k0 < cmax; k0 += TNB) {
      const int k1 = std::min(k0 + TNB, cmax);
      // first column of the panel: plain row of the (just updated) matrix
      hipLaunchKernelGGL(trd_wx_kernel, dim3((cmax - k0 + WXR - 1) / WXR, nc), dim3(256), 0, ctx->stream, d_tm, k0, 0,
                         0, 1);
      for (int k = k0; k < k1; ++k) {
        const int j = k - k0;
        if (k < cmax - 1) {
          const int ng = (cmax - k - 1 + SYG - 1) / SYG;
          const int nslotblk = ((2 * j + 3) / 4 + SYW - 1) / SYW;  // 4 vectors per wave, SYW waves per workgroup
          // algorithmic HBM bytes of this column: half of every trailing matrix (symv), one pass over
          // the panel rows of V and W (wx)
          // Timed with events on every DM_PROF_TRD_STRIDE-th column only (event records on a chain of
          // ~2400 short launches are not free: all of them cost 7 % of the step); columns are sampled
          // uniformly, so the ratio bytes / time of the sample estimates the average of the kernel.
          // (the sampled position walks through the panel: every column index j of a panel is drawn equally often)
          const bool timed = ctx->prof_on && (k % DM_PROF_TRD_STRIDE) == ((k / DM_PROF_TRD_STRIDE) * 13) % DM_PROF_TRD_STRIDE;
          double by_symv = 0.0, by_wx = 0.0;
          if (timed)
            for (int p : ch) {
              const double r = probs[p].n - k - 1;
              if (r > 0) { by_symv += 8.0 * r * r; by_wx += 32.0 * r * j; }
            }
          hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
          if (timed) { e0 = dm_prof_event(ctx); (void)hipEventRecord(e0, ctx->stream); }
          hipLaunchKernelGGL(trd_symv_kernel, dim3(nslotblk + ng, nc), dim3(64 * SYW), 0, ctx->stream, d_tm, k, j);
          hipEvent_t e1b = nullptr;  // a record owns both of its events: end of symv and start of wx are two events
          if (timed) {
            e1 = dm_prof_event(ctx);
            (void)hipEventRecord(e1, ctx->stream);
            e1b = dm_prof_event(ctx);
            (void)hipEventRecord(e1b, ctx->stream);
          }
          hipLaunchKernelGGL(trd_wx_kernel, dim3((cmax - k - 1 + WXR - 1) / WXR, nc), dim3(256), 0, ctx->stream, d_tm,
                             k, j, 1, k + 1 < k1 ? 1 : 0);
          if (timed) {
            e2 = dm_prof_event(ctx);
            (void)hipEventRecord(e2, ctx->stream);
            // weight = stride: dm_prof_report returns estimates of the totals over ALL columns
            ctx->prof.push_back(dm_ctx::prof_rec{DM_PROF_TRD_SYMV, e0, e1, by_symv, (double)DM_PROF_TRD_STRIDE});
            ctx->prof.push_back(dm_ctx::prof_rec{DM_PROF_TRD_WX, e1b, e2, by_wx, (double)DM_PROF_TRD_STRIDE});
          }
        }
      }
      if (k1 < cmax) {
        // her2k on the upper triangle in one pass: C -= [V W] [W V]^H  (K = 2 TNB; unused panel rows are zero)
        std::vector<dm_gemm_desc> g;
        for (int p : ch) {
          const int n = probs[p].n;
          const int rem = n - k1;
          if (rem <= 0) continue;
          const cplx* pp = PP + offn[p] * 3 * TNB;
          g.push_back(dm_gemm_make(pp + k1, 1, n, false, pp + (size_t)n * TNB + k1, n, 1, true,
                                   probs[p].C + (size_t)k1 * probs[p].ldc + k1, probs[p].ldc, rem, rem, 2 * TNB, -1.0,
                                   1.0, nullptr, DM_GEMM_UPPER));
        }
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
        // The panel buffers are NOT cleared between panels: every entry a kernel reads has been written inside the
        // current panel — trd_symv / trd_wx read the vectors q < j at rows > k only (v_q and w_q are written for all
        // rows > k0 + q), the her2k above reads rows >= k1 of all TNB vectors of a FULL panel (a matrix that ends
        // inside the panel has n - k1 <= 0 and takes no part).  (170 MB of memset per panel at configs[1].)
        static const bool clear_panels = getenv("DM_TRD_CLEAR_PANELS") != nullptr;
        if (clear_panels) DM_TRY(dm_fill_zero(ctx, PP, sizeof(cplx) * totn * 3 * TNB));
      }
    }
    DM_HIP(ctx, hipGetLastError());
    return DM_OK;
  };

  std::vector<rot_mat*> d_rm_of(nch, nullptr);
  auto phase_T2 = [&](int c, hipStream_t st) -> int {
    const std::vector<int>& ch = chunks[c];
    if (ch.empty()) return DM_OK;
    const int nc = (int)ch.size();
    int cmax = 0;
    std::vector<ql_mat> qm(nc);
    std::vector<rot_mat> rm(nc);
    for (int i = 0; i < nc; ++i) {
      const int p = ch[i];
      const int n = probs[p].n;
      cmax = std::max(cmax, n);
      qm[i] = ql_mat{dd + offn[p], ee + offn[p], n, sw_dir + swoff[p], sw_lo + swoff[p], sw_cnt + swoff[p],
                     sw_off + swoff[p], rot + rotoff[p], 4 * n + 8, 2LL * n * n + 8, nsw + p, stat + p};
      rm[i] = rot_mat{Zt + off[p], n, n, sw_dir + swoff[p], sw_lo + swoff[p], sw_cnt + swoff[p], sw_off + swoff[p],
                      rot + rotoff[p], nsw + p};
    }
    // descriptors are uploaded on the main stream; the caller orders `st` after them with an event
    ql_mat* d_qm = dm_ws_upload(ctx, qm);
    d_rm_of[c] = dm_ws_upload(ctx, rm);
    if (!d_qm || !d_rm_of[c]) return DM_ENOMEM;
    if (st != ctx->stream) {
      hipEvent_t e = side_event(2 * c);
      DM_HIP(ctx, hipEventRecord(e, ctx->stream));
      DM_HIP(ctx, hipStreamWaitEvent(st, e, 0));
    }
    if ((size_t)cmax * 16 <= 120u * 1024u) {
      static bool attr = false;
      if (!attr) {
        DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ql_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr = true;
      }
      DM_PLAUNCH(ctx, DM_PROF_DC, ql_kernel<true>, dim3(nc), dim3(64), (size_t)cmax * 16, st, d_qm);
    } else {
      DM_PLAUNCH(ctx, DM_PROF_DC, ql_kernel<false>, dim3(nc), dim3(64), 0, st, d_qm);
    }
    if (st != ctx->stream) DM_HIP(ctx, hipEventRecord(side_event(2 * c + 1), st));
    DM_HIP(ctx, hipGetLastError());
    return DM_OK;
  };

  std::vector<double*> zfinal;  // set by the divide & conquer path
  auto phase_T34 = [&](int c, bool waited_on_side) -> int {
    const std::vector<int>& ch = chunks[c];
    if (ch.empty()) return DM_OK;
    const int nc = (int)ch.size();
    if (waited_on_side) DM_HIP(ctx, hipStreamWaitEvent(ctx->stream, side_event(2 * c + 1), 0));
    int cmax = 0;
    for (int p : ch) cmax = std::max(cmax, probs[p].n);
    // T3 (QL path only: D&C delivers the eigenvectors directly)
    if (zfinal.empty()) {
      DM_PLAUNCH(ctx, DM_PROF_DC, zt_identity_kernel, dim3((cmax + 255) / 256, cmax, nc), dim3(256), 0, ctx->stream,
                         d_rm_of[c]);
      DM_PLAUNCH(ctx, DM_PROF_DC, rot_apply_kernel, dim3((cmax + 255) / 256, nc), dim3(256), 0, ctx->stream, d_rm_of[c]);
    }
    {
      std::vector<dm_cdesc> cp;
      for (int p : ch)
        if (probs[p].n > 0)
          cp.push_back(dm_cdesc{dd + offn[p], evals + (size_t)p * evals_stride, sizeof(double) * probs[p].n});
      DM_TRY(dm_copy_batched(ctx, cp));
    }
    // optional selection of the eigenvectors that are back-transformed at all
    std::vector<const double*> zsrc(np, nullptr);
    std::vector<int> ncolv(np, 0);
    for (int p : ch) {
      zsrc[p] = zfinal.empty() ? Zt + off[p] : zfinal[p];
      ncolv[p] = probs[p].n;
    }
    if (sel) {
      std::vector<double> hev(totn);
      DM_TRY(dm_download(ctx, hev.data(), dd, sizeof(double) * totn));
      if ((int)sel->nsel.size() != np) sel->nsel.assign(np, 0);
      std::vector<int> hidx;
      std::vector<size_t> ioff(np, 0), zoff(np, 0);
      size_t ztot = 0;
      // The callback sorts the spectrum of a matrix: a millisecond of host time for a batch of 10^2 matrices, during
      // which the GPU has nothing queued — the matrices are independent, so a few host threads share them
      // (the callback writes per-matrix state only; see dm_eig_select).
      std::vector<std::vector<int>> colsv(ch.size());
      {
        size_t work = 0;
        for (int p : ch) work += (size_t)probs[p].n;
        const unsigned hw = std::thread::hardware_concurrency();
        const int nth = (work >= 16384 && ch.size() >= 8) ? (int)std::min<size_t>(std::min<unsigned>(8u, std::max(1u, hw / 2)), ch.size()) : 1;
        auto run = [&](int t) {
          for (size_t i = t; i < ch.size(); i += nth) {
            const int p = ch[i];
            if (probs[p].n > 0) sel->pick(p, hev.data() + offn[p], probs[p].n, colsv[i]);
          }
        };
        if (nth == 1) {
          run(0);
        } else {
          std::vector<std::thread> th;
          for (int t = 1; t < nth; ++t) th.emplace_back(run, t);
          run(0);
          for (auto& t : th) t.join();
        }
      }
      for (size_t ci = 0; ci < ch.size(); ++ci) {
        const int p = ch[ci];
        const int n = probs[p].n;
        const std::vector<int>& cols = colsv[ci];
        for (int c : cols) DM_ARG(ctx, c >= 0 && c < n);
        ioff[p] = hidx.size();
        hidx.insert(hidx.end(), cols.begin(), cols.end());
        sel->nsel[p] = (int)cols.size();
        zoff[p] = ztot;
        ztot += cols.size() * (size_t)n;
      }
      int* d_idx = dm_ws_upload(ctx, hidx);
      double* Zsel = dm_ws_alloc_t<double>(ctx, std::max<size_t>(ztot, 1));
      if (!d_idx || !Zsel) return DM_ENOMEM;
      std::vector<zsel_mat> zm;
      int maxsel = 0;
      for (int p : ch) {
        if (sel->nsel[p] > 0) zm.push_back(zsel_mat{zsrc[p], Zsel + zoff[p], d_idx + ioff[p], probs[p].n, sel->nsel[p]});
        maxsel = std::max(maxsel, sel->nsel[p]);
        zsrc[p] = Zsel + zoff[p];
        ncolv[p] = sel->nsel[p];
      }
      if (!zm.empty()) {
        zsel_mat* d_zm = dm_ws_upload(ctx, zm);
        if (!d_zm) return DM_ENOMEM;
        DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, zsel_gather_kernel, dim3((maxsel + 3) / 4, (unsigned)zm.size()), dim3(256), 0, ctx->stream, d_zm);
      }
    }
    // T4: X = Q Z into the (now free) storage of C, block reflectors applied last to first
    std::vector<cvt_mat> cm(nc);
    for (int i = 0; i < nc; ++i)
      cm[i] = cvt_mat{zsrc[ch[i]], probs[ch[i]].C, probs[ch[i]].ldc, probs[ch[i]].n, ncolv[ch[i]]};
    cvt_mat* d_cm = dm_ws_upload(ctx, cm);
    if (!d_cm) return DM_ENOMEM;
    if (small_path) {
      // X = Q Z with the explicit Q of trd_small and the real eigenvectors Z of the tridiagonal
      // (Z[c * n + r], eigenvector-major) as one complex x real product per matrix
      std::vector<dm_gemm_desc> g;
      for (int i = 0; i < nc; ++i) {
        const int p = ch[i];
        const int n = probs[p].n;
        if (n <= 0) continue;
        if (ncolv[p] <= 0) continue;
        g.push_back(dm_gemm_make(Ut + off[p], n, 1, false, cm[i].Zt, 1, n, false, probs[p].C, probs[p].ldc, n, ncolv[p], n,
                                 1.0, 0.0, nullptr, DM_GEMM_B_REAL));
      }
      DM_TRY(dm_gemm_grouped_launch(ctx, g));
      std::vector<dm_tdesc> tr;
      for (int p : ch) tr.push_back(dm_tdesc{probs[p].C, probs[p].ldc, probs[p].W, probs[p].ldw, probs[p].n, ncolv[p]});
      DM_TRY(dm_conj_transpose_batched(ctx, tr));
      DM_HIP(ctx, hipGetLastError());
      return DM_OK;
    }
    const int tb = (cmax + 31) / 32;
    DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, zt_to_x_kernel, dim3(tb, tb, nc), dim3(256), 0, ctx->stream, d_cm);
    // two-stage reduction: X <- Q2 X first (the reflectors of the bulge chase), then the blocks of the first stage
    if (two_stage) DM_TRY(apply_q2(ch, ncolv));
    // ---- T factors of all blocks up front (they depend on V only), batched over blocks and matrices:
    //   level 0: T of every TNB-wide panel from its Gram matrix (zlarft)
    //   merge:   [T_l, -T_l (V_l^H V_r) T_r; 0, T_r] for neighbouring blocks until NBB is reached
    //   U^H = T V^H per block, so that applying a block is two products: W = U^H X, X -= V W
    if (!two_stage) DM_TRY(dm_fill_zero(ctx, Tbig, sizeof(cplx) * tottb));
    if (!two_stage) {  // (the first stage of the two-stage reduction has left the T factors of its panels in Tbig)
      std::vector<dm_gemm_desc> g;
      std::vector<tf_mat> tf;
      for (int p : ch) {
        const int n = probs[p].n;
        for (int k0 = 0; k0 < n - 1; k0 += TNB) {
          const int kb = std::min(k0 + TNB, n - 1) - k0;
          const int r0 = k0 + 1, nr = n - r0;
          const cplx* Vb = Vt + off[p] + (size_t)k0 * n + r0;
          cplx* G = Gs + offg[p] + (size_t)(k0 / TNB) * TNB * TNB;
          cplx* T = Tbig + offtb[p] + (size_t)(k0 / NBB) * NBB * NBB + (size_t)(k0 % NBB) * NBB + (k0 % NBB);
          g.push_back(dm_gemm_make(Vb, n, 1, true, Vb, 1, n, false, G, TNB, kb, kb, nr));
          tf.push_back(tf_mat{G, tau + offn[p] + k0, T, kb, NBB, nullptr, 0});
        }
      }
      if (!g.empty()) {
        DM_TRY(dm_gemm_grouped_launch(ctx, g));
        tf_mat* d_tf = dm_ws_upload(ctx, tf);
        if (!d_tf) return DM_ENOMEM;
        static bool attr = false;
        const size_t lds = 2 * sizeof(cplx) * TNB * (TNB + 1);
        if (!attr) {
          DM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(larft_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          attr = true;
        }
        DM_PLAUNCH(ctx, DM_PROF_EIG_OTHER, larft_kernel, dim3((unsigned)tf.size()), dim3(256), lds, ctx->stream, d_tf);
      }
    }
    std::deque<dm_gemm_plan> plans;
    for (int sz = TNB; sz < NBB; sz *= 2) {
      std::vector<dm_gemm_desc> ga, gb, gc;
      for (int p : ch) {
        const int n = probs[p].n;
        const int nrefl = nrefl_of(n);
        for (int k0 = 0; k0 + sz < nrefl; k0 += 2 * sz) {  // left block [k0, k0+sz), right block [k0+sz, ...)
          const int kr0 = k0 + sz;
          const int kl = sz, kr = std::min(kr0 + sz, nrefl) - kr0;
          const int r0 = kr0 + shift, nr = n - r0;         // rows where the right block is non-zero
          const cplx* Vl = Vt + off[p] + (size_t)k0 * n + r0;
          const cplx* Vr = Vt + off[p] + (size_t)kr0 * n + r0;
          cplx* G = Gs + offg[p] + (size_t)(k0 / (2 * sz)) * sz * sz;
          cplx* H = Gt + offg[p] + (size_t)(k0 / (2 * sz)) * sz * sz;
          cplx* Tblk = Tbig + offtb[p] + (size_t)(k0 / NBB) * NBB * NBB;
          const int o = k0 % NBB;
          cplx* Tl = Tblk + (size_t)o * NBB + o;
          cplx* Tr = Tblk + (size_t)(o + sz) * NBB + (o + sz);
          cplx* T12 = Tblk + (size_t)o * NBB + (o + sz);
          ga.push_back(dm_gemm_make(Vl, n, 1, true, Vr, 1, n, false, G, sz, kl, kr, nr));
          gb.push_back(dm_gemm_make(Tl, NBB, 1, false, G, sz, 1, false, H, sz, kl, kr, kl));
          gc.push_back(dm_gemm_make(H, sz, 1, false, Tr, NBB, 1, false, T12, NBB, kl, kr, kr, -1.0, 0.0));
        }
      }
      for (const auto* gv : {&ga, &gb, &gc}) {
        plans.emplace_back();
        DM_TRY(dm_gemm_plan_build(*gv, plans.back()));
      }
    }
    {
      std::vector<dm_gemm_desc> g;
      for (int p : ch) {
        const int n = probs[p].n;
        const int nrefl = nrefl_of(n);
        for (int k0 = 0; k0 < nrefl; k0 += NBB) {
          const int kb = std::min(k0 + NBB, nrefl) - k0;
          const int r0 = k0 + shift, nr = n - r0;
          const cplx* T = Tbig + offtb[p] + (size_t)(k0 / NBB) * NBB * NBB;
          g.push_back(dm_gemm_make(T, NBB, 1, false, Vt + off[p] + (size_t)k0 * n + r0, n, 1, true,
                                   Ut + off[p] + (size_t)k0 * n + r0, n, kb, nr, kb));
        }
      }
      plans.emplace_back();
      DM_TRY(dm_gemm_plan_build(g, plans.back()));
    }
    // ---- apply the blocks, last to first
    const int nblk = (nrefl_of(cmax) + NBB - 1) / NBB;
    for (int b = nblk - 1; b >= 0; --b) {
      const int k0 = b * NBB;
      std::vector<dm_gemm_desc> g2, g4;
      for (int p : ch) {
        const int n = probs[p].n;
        const int kb = std::min(k0 + NBB, nrefl_of(n)) - k0;
        if (kb <= 0) continue;
        // reflectors k >= k0 vanish on rows < k0 + shift: only rows r0.. of X take part
        const int r0 = k0 + shift, nr = n - r0;
        cplx* Xr = probs[p].C + (size_t)r0 * probs[p].ldc;
        cplx* w1 = W1 + offn[p] * NBB;
        const int nx = ncolv[p];  // columns of X = eigenvectors being back-transformed
        if (nx <= 0) continue;
        g2.push_back(dm_gemm_make(Ut + off[p] + (size_t)k0 * n + r0, n, 1, false, Xr, probs[p].ldc, 1, false, w1, n, kb,
                                  nx, nr));
        g4.push_back(dm_gemm_make(Vt + off[p] + (size_t)k0 * n + r0, 1, n, false, w1, n, 1, false, Xr, probs[p].ldc, nr,
                                  nx, kb, -1.0, 1.0));
      }
      if (g2.empty()) continue;
      for (const auto* gv : {&g2, &g4}) {
        plans.emplace_back();
        DM_TRY(dm_gemm_plan_build(*gv, plans.back()));
      }
    }
    // the merges of the T factors, U^H = T V^H and the two products per block are a chain of ~25 dependent launches:
    // their descriptors travel in one staged copy, then the launches follow each other without a copy in between
    {
      std::vector<const dm_gemm_plan*> pp;
      for (const auto& pl : plans) pp.push_back(&pl);
      std::vector<const char*> dv;
      DM_TRY(dm_gemm_plans_upload(ctx, pp, dv));
      for (size_t i = 0; i < plans.size(); ++i) DM_TRY(dm_gemm_plan_run(ctx, plans[i], dv[i]));
    }
    {
      std::vector<dm_tdesc> tr;
      for (int p : ch) tr.push_back(dm_tdesc{probs[p].C, probs[p].ldc, probs[p].W, probs[p].ldw, probs[p].n, ncolv[p]});
      DM_TRY(dm_conj_transpose_batched(ctx, tr));
    }
    DM_HIP(ctx, hipGetLastError());
    return DM_OK;
  };

  if (use_dc) {
#if DM_TNB == 32
    if (two_stage) DM_TRY(phase_T1_two(chunks[0]));
    else
#endif
    DM_TRY(phase_T1(chunks[0]));
    // Ut is first written by the back-transformation (unless the LDS-resident small path put Q there)
    DM_TRY(dc_solve(ctx, probs, dd, ee, offn, off, tot, totn, zfinal,
                    maxn > TSM ? reinterpret_cast<double*>(Ut) : nullptr));
    DM_TRY(phase_T34(0, false));
  } else if (nch == 1) {
    DM_TRY(phase_T1(chunks[0]));
    DM_TRY(phase_T2(0, ctx->stream));
    DM_TRY(phase_T34(0, false));
  } else {
    for (int c = 0; c < nch; ++c) {
      DM_TRY(phase_T1(chunks[c]));
      DM_TRY(phase_T2(c, g_side.s));
      if (c > 0) DM_TRY(phase_T34(c - 1, true));
    }
    DM_TRY(phase_T34(nch - 1, true));
  }

  std::vector<int> hstat(np);
  DM_TRY(dm_download(ctx, hstat.data(), stat, sizeof(int) * np));
  if (nch > 1) DM_HIP(ctx, hipStreamSynchronize(g_side.s));
  for (int p = 0; p < np; ++p)
    if (hstat[p] != 0) {
      ctx->err = hstat[p] == 1 ? "tridiagonal QL iteration did not converge" : "QL rotation storage exhausted";
      dm_ws_release(ctx, mark);
      return 1000 + p;  // > 0: numerical failure
    }
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dm_ws_release(ctx, mark);
  return DM_OK;
}

}  // namespace DM_TRD_NS
