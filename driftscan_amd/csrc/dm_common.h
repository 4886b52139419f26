// dm_common.h — shared host/device definitions for libdriftmi (gfx950 only).
//
// Everything in csrc/ is written for CDNA4 (MI355X): 64-wide wavefronts, the
// fp64 MFMA v_mfma_f64_16x16x4_f64, 160 KiB LDS per CU, 8 XCDs.  There is no
// other back-end and no CPU fallback.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef double2 cplx;  // interleaved (re, im) complex128, matches numpy/HDF5 {r,i}

// ---------------------------------------------------------------------------
// context: one per GPU, one host thread per context (include/driftmi.h)
// ---------------------------------------------------------------------------
struct dm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // bump-allocated device workspace; grows on demand (growth synchronises)
  char* ws = nullptr;
  size_t ws_cap = 0;
  size_t ws_used = 0;
  std::vector<void*> retired;  // old arenas kept alive until the next reset
  // pinned host staging for small descriptor uploads / flag read-backs
  char* hpin = nullptr;
  size_t hpin_cap = 0;
  size_t hpin_used = 0;  // ring offset (per context: contexts may be driven from different threads)
  char* hpin_dl = nullptr;  // page-locked landing buffer of dm_download (allocated on first use)
  std::string err;
  // optional per-kernel-class timing (HIP events on ctx->stream) and flop accounting
  bool prof_on = false;
  int prof_level = 1;          // 2: the extended classes (every kernel of the path) are bracketed too
  struct prof_rec { int cls; hipEvent_t a, b; double flops; double weight = 1.0; };
  unsigned prof_seq[32] = {0};  // per class: launches seen (small launches are timed one in DM_PROF_SMALL_STRIDE)
  std::vector<prof_rec> prof;
  std::vector<hipEvent_t> ev_pool;
  unsigned long long* prof_dev = nullptr;  // device flop counters, one per class
  // eigensolver policy override of the caller in flight (-1: the library's policy; 0: one-stage tridiagonalisation only):
  // set and restored around dm_herm_eig_tridiag by a caller that knows its batch (dm_jacobi_rows, `one_stage_eig`)
  int trd_mode_override = -1;
};

// Event records are not free on a chain of thousands of short launches (each is a marker packet that breaks the
// back-to-back dispatch: ~1200 of them cost 4-9 ms of a 170 ms step), so the live measurement SAMPLES: every
// DM_PROF_TRD_STRIDE-th column of the tridiagonalisation, every launch of at least DM_PROF_BIG work units, and
// one in DM_PROF_SMALL_STRIDE of the smaller launches of a class (weighted accordingly in dm_prof_report).
constexpr int DM_PROF_TRD_STRIDE = 32;
constexpr int DM_PROF_SMALL_STRIDE = 4;
constexpr double DM_PROF_BIG = 2.0e9;
// classes 0-5 carry algorithmic FLOPs, 6-7 (the HBM-bound tridiagonalisation kernels) algorithmic BYTES
enum { DM_PROF_GEMM = 0, DM_PROF_GEMM_REAL = 1, DM_PROF_JAC_GRAM = 2, DM_PROF_JAC_INNER = 3, DM_PROF_JAC_APPLY = 4,
       DM_PROF_DGEMM = 5, DM_PROF_TRD_SYMV = 6, DM_PROF_TRD_WX = 7,
       // the two-stage tridiagonalisation (dm_sbr_impl.h): fp64 VALU kernels, algorithmic FLOPs
       DM_PROF_SB_PANEL = 8, DM_PROF_SB_CHASE = 9, DM_PROF_SB_Q2 = 10,
       // gathered-B ZGEMM: the covariance projections (B_f o C_l) B_f'^H of the KL stage, algorithmic FLOPs
       DM_PROF_GEMM_COV = 11,
       // "extended" classes: every remaining kernel of the path, time only (bracketed at profiling level 2:
       // dm_prof_reset(ctx, 2); the timed passes of bench.py run at level 1 and do not pay for these events)
       DM_PROF_EXT0 = 12,
       DM_PROF_BT_RING = 12,     // fused map synthesis + ring transform (bt_fused_dft / dft2 / fft)
       DM_PROF_BT_OTHER = 13,    // beams, solid angles, fold, tables, masks, refinement helpers
       DM_PROF_TRD_SMALL = 14,   // LDS-resident tridiagonalisation of the small Gram problems
       DM_PROF_DC = 15,          // divide & conquer on the tridiagonal (everything but its real GEMMs) + QL leaves
       DM_PROF_CHOL = 16,        // potf2, panel substitutions, diagonal-block solves
       DM_PROF_UTIL = 17,        // transposes, copies, identities, regularisation / all-zero scans, Fisher helpers
       DM_PROF_EIG_OTHER = 18,   // T factors, slice sums, band extraction, eigenvector gathers
       DM_PROF_SVD_OTHER = 19,   // Jacobi engine helpers (norms, ranks, gathers, cleaning) and SVD chain assembly
       DM_PROF_NCLASS = 24 };

hipEvent_t dm_prof_event(dm_ctx* ctx);
// bracket one launch: DM_PROF(ctx, cls, flops) { launch; }
struct dm_prof_scope {
  dm_ctx* c; int cls; double flops; hipEvent_t a = nullptr; double weight = 1.0;
  dm_prof_scope(dm_ctx* ctx, int cls_, double fl) : c(ctx), cls(cls_), flops(fl) {
    if (!c->prof_on) return;
    if (cls >= DM_PROF_EXT0 && c->prof_level < 2) return;
    const bool gemm_class = cls == DM_PROF_GEMM || cls == DM_PROF_GEMM_REAL || cls == DM_PROF_DGEMM;
    if (gemm_class && fl < DM_PROF_BIG) {  // (the Jacobi classes are a few dozen launches per step: all timed)
      if (c->prof_seq[cls & 31]++ % DM_PROF_SMALL_STRIDE != 0) return;
      weight = DM_PROF_SMALL_STRIDE;
    }
    a = dm_prof_event(c);
    (void)hipEventRecord(a, c->stream);
  }
  ~dm_prof_scope() {
    if (c->prof_on && a) {
      hipEvent_t b = dm_prof_event(c);
      (void)hipEventRecord(b, c->stream);
      c->prof.push_back(dm_ctx::prof_rec{cls, a, b, flops, weight});
    }
  }
};

// a launch of one of the extended classes: bracketed when the context profiles at level 2, a plain launch otherwise
#define DM_PLAUNCH(ctx_, cls_, ...)                      \
  do {                                                    \
    dm_prof_scope dm_ps__((ctx_), (cls_), 0.0);           \
    hipLaunchKernelGGL(__VA_ARGS__);                      \
  } while (0)


#define DM_OK 0
#define DM_EARG (-1)
#define DM_EHIP (-2)
#define DM_ENOMEM (-3)

#define DM_HIP(ctx, call)                                                        \
  do {                                                                           \
    hipError_t e__ = (call);                                                     \
    if (e__ != hipSuccess) {                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);           \
      return DM_EHIP;                                                            \
    }                                                                            \
  } while (0)

#define DM_ARG(ctx, cond)                                         \
  do {                                                            \
    if (!(cond)) {                                                \
      (ctx)->err = std::string("bad argument: ") + #cond;         \
      return DM_EARG;                                             \
    }                                                             \
  } while (0)

#define DM_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != DM_OK) return rc__; \
  } while (0)

// workspace helpers (dm_ctx.cpp)
int dm_ws_reserve(dm_ctx* ctx, size_t bytes);
void* dm_ws_alloc(dm_ctx* ctx, size_t bytes);  // nullptr on failure (err set)
size_t dm_ws_mark(dm_ctx* ctx);
void dm_ws_release(dm_ctx* ctx, size_t mark);
int dm_upload(dm_ctx* ctx, void* dst, const void* src, size_t bytes);    // async H2D via pinned staging
int dm_download(dm_ctx* ctx, void* dst, const void* src, size_t bytes);  // D2H + stream sync

// RAII guard for the bump arena: every extern "C" entry point (and every internal driver that
// allocates) opens one, so an early `return` after a failed launch or allocation cannot leave
// ws_used above its entry value (dm_ctx_workspace_reset insists on an empty arena).
struct dm_ws_scope {
  dm_ctx* c;
  size_t mark;
  explicit dm_ws_scope(dm_ctx* ctx) : c(ctx), mark(dm_ws_mark(ctx)) {}
  ~dm_ws_scope() { dm_ws_release(c, mark); }
  dm_ws_scope(const dm_ws_scope&) = delete;
  dm_ws_scope& operator=(const dm_ws_scope&) = delete;
};

template <typename T>
static inline T* dm_ws_alloc_t(dm_ctx* ctx, size_t n) {
  return reinterpret_cast<T*>(dm_ws_alloc(ctx, n * sizeof(T)));
}

// upload a std::vector into freshly bump-allocated workspace
template <typename T>
static inline T* dm_ws_upload(dm_ctx* ctx, const std::vector<T>& v) {
  if (v.empty()) return dm_ws_alloc_t<T>(ctx, 1);
  T* d = dm_ws_alloc_t<T>(ctx, v.size());
  if (!d) return nullptr;
  if (dm_upload(ctx, d, v.data(), v.size() * sizeof(T)) != DM_OK) return nullptr;
  return d;
}

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#if defined(__HIPCC__)

typedef double dm_f64x4 __attribute__((ext_vector_type(4)));

// Pointers that come out of descriptor structs are generic to the compiler, which then emits
// flat_load/flat_store; those count on LGKM as well as VM, so every LDS wait (lgkmcnt) also
// drains the global loads in flight and prefetching stops overlapping.  Going through the
// global address space gives global_load/global_store (VM counter only).
typedef double dm_d2 __attribute__((ext_vector_type(2)));
typedef const dm_d2 __attribute__((address_space(1)))* dm_gd2_c;
typedef dm_d2 __attribute__((address_space(1)))* dm_gd2;
typedef const double __attribute__((address_space(1)))* dm_gd_c;
typedef double __attribute__((address_space(1)))* dm_gd;
__device__ __forceinline__ cplx dm_ldg(const cplx* p, size_t i = 0) {
  const dm_d2 v = ((dm_gd2_c)p)[i];
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ double dm_ldg(const double* p, size_t i = 0) { return ((dm_gd_c)p)[i]; }
__device__ __forceinline__ void dm_stg(cplx* p, size_t i, cplx v) {
  dm_d2 t;
  t.x = v.x;
  t.y = v.y;
  ((dm_gd2)p)[i] = t;
}
__device__ __forceinline__ void dm_stg(double* p, size_t i, double v) { ((dm_gd)p)[i] = v; }

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one f64 per lane for A and B.
//   A: lane l holds A[i = l & 15][k = l >> 4]
//   B: lane l holds B[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// (cdna_hip_programming.md §3 "f64 MFMA does NOT use these maps")
__device__ __forceinline__ dm_f64x4 dm_mfma(double a, double b, dm_f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (decoded with
// indicator inputs, scratch/mfma4_layout.hip; cbsz/abid have no effect for f64).  With
// k = l >> 4, g = (l >> 2) & 3, t = l & 3:
//   A: lane l holds A_g[i = t][k];  B: lane l holds B_g[k][j = t];  D_g[i][j] lands in lane 16 i + 4 g + j.
// It issues at 16 cycles per instruction (73-77 TFLOP/s on MI355X) in ANY order of accumulators; the 16x16x4 form costs
// 64 + ~41 / R cycles with R consecutive MFMAs on ONE accumulator (105 cycles = 47 TFLOP/s with the accumulators in
// rotation — what rounds 1-4 took for that instruction's ceiling — 83 in chains of 2, 69.5 in chains of 8, 66 = 75
// TFLOP/s in chains of 16+: scratch/mfma_peak3.hip).  A 16x16 tile is built from the 4x4x4 form
// by putting row block g of A in slot g and column block (g + s) & 3 of B in slot g for the four rotations s (the
// rotated B fragments are four LDS reads with different addresses, no shuffles).
__device__ __forceinline__ double dm_mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// DPP move of a double inside each 16-lane row (two v_mov_b32 with a dpp modifier, no LDS crossbar).
// row_ror:n (CTRL = 0x120 + n) hands lane i the value of lane (i - n) mod 16 of its row.  The 4x4x4 MFMA pairs
// row block g with column block (g + s) & 3 in its rotation s: the operand of the lane 4 s further along the row,
// i.e. a rotation by 16 - 4 s: dm_rot4<1> = row_ror:12, <2> = row_ror:8, <3> = row_ror:4.
template <int CTRL>
__device__ __forceinline__ double dm_dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  // bound_ctrl = true: lanes without a source read 0 and the destination needs no initial value — with `false` the
  // compiler writes the `old` operand first, one more v_mov per half (every control used here has a source for all lanes)
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int S>
__device__ __forceinline__ double dm_rot4(double v) {
  if constexpr (S == 0) return v;
  else if constexpr (S == 1) return dm_dpp_f64<0x12C>(v);
  else if constexpr (S == 2) return dm_dpp_f64<0x128>(v);
  else return dm_dpp_f64<0x124>(v);
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }

// XCD-aware remap of a 1-D block id: consecutive logical tiles land on the same
// XCD (blocks b and b+8 share an XCD under the observed round-robin dispatch;
// speed only, never correctness — MI355X_MICROARCH.md §Workgroup dispatch).
__device__ __forceinline__ int dm_xcd_remap(int bid, int nblocks) {
  const int nx = 8;
  int per = nblocks / nx;
  int full = per * nx;
  if (bid >= full) return bid;  // ragged tail keeps its id
  return (bid % nx) * per + bid / nx;
}

// The same with the XCDs kept NEAR each other in the tile list: the list is cut into super-blocks of 8 * chunk tiles and
// XCD x takes the x-th chunk of each — for launches whose problems share an operand that fits the Infinity Cache but not
// an L2 (the covariance projections: every row-frequency product of an m-block reads all of that block's beam_svd rows,
// 190 MB at configs[2]): with one contiguous run per XCD eight different m-blocks are in flight and the shared operand
// comes from HBM every time.
__device__ __forceinline__ int dm_xcd_remap_chunked(int bid, int nblocks, int chunk) {
  const int nx = 8;
  const int S = nx * chunk;
  const int sb = bid / S;
  if ((sb + 1) * S > nblocks) return bid;  // ragged last super-block keeps its ids
  const int r = bid - sb * S;
  return sb * S + (r % nx) * chunk + r / nx;
}

__device__ __forceinline__ double dm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double dm_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

#endif  // __HIPCC__
