/* dm_h5io.c — libdriftio: HDF5 product files for the per-m hot path (see include/driftio.h).
 *
 * Plain C on the HDF5 1.10 C API.  What is specific to this library:
 *   - its own LZF codec (the byte format of liblzf 3.x, which h5py's filter 32000 wraps), registered as an
 *     HDF5 filter so that H5Dread decodes files written by h5py and h5py decodes ours;
 *   - chunked writes go through H5Dwrite_chunk: the caller's thread gathers and compresses each chunk with
 *     no lock held, and only the hand-over of the finished bytes to HDF5 is serialised.  The product files
 *     of one m are independent, so the writer pool of driftscan_amd/storage.py compresses as many datasets
 *     at once as it has threads although libhdf5 itself is single-threaded.
 */
#include "../../include/driftio.h"

#include <hdf5.h>
#include <emmintrin.h>
#include <immintrin.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DIO_LZF_FILTER 32000
#define DIO_LZF_VERSION 0x0105 /* liblzf 3.5, as h5py writes into cd_values[1] */
#define DIO_LZF_REVISION 4     /* h5py's H5PY_FILTER_LZF_VERSION */

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static __thread char g_err[512];

#define LOCK() pthread_mutex_lock(&g_lock)
#define UNLOCK() pthread_mutex_unlock(&g_lock)

static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}

const char* dio_last_error(void) { return g_err; }
int dio_version(void) { return 100; }

/* ------------------------------------------------------------------------------------------------
 * LZF: control byte c < 32: c + 1 literal bytes follow; else a back reference of length (c >> 5) + 2
 * (a length field of 7 is extended by the next byte) at distance ((c & 31) << 8 | next byte) + 1.
 * ------------------------------------------------------------------------------------------------ */
#define HLOG 14
#define MAX_LIT 32
#define MAX_OFF 8192
#define MAX_REF 264

/* pending literals in[from, to) as runs of at most MAX_LIT; returns the new output position, 0 if they do not fit */
static inline size_t lzf_put_literals(const unsigned char* in, size_t from, size_t to, unsigned char* out, size_t op, size_t out_len) {
  while (from < to) {
    const size_t run = to - from > MAX_LIT ? MAX_LIT : to - from;
    if (op + 1 + run > out_len) return 0;
    out[op++] = (unsigned char)(run - 1);
    memcpy(out + op, in + from, run);
    op += run;
    from += run;
  }
  return op;
}
size_t dio_lzf_compress(const void* in_, size_t n, void* out_, size_t out_len) {
  const unsigned char* in = (const unsigned char*)in_;
  unsigned char* out = (unsigned char*)out_;
  if (n == 0 || out_len == 0 || n >= 0x7fffffffu) return 0;
  /* The hash table is per thread and never cleared between calls (chunks are a few KB to a few MB, thousands per
   * file: clearing would cost more than compressing): an entry holds base + position + 1 and is valid only while it
   * exceeds the base of the current call.  Literals are gathered and copied in runs; a run of misses widens the step
   * (the mantissas of full-precision products are noise: 0.23 -> 1 GB/s per thread through such stretches, while the zero
   * rows and padding that do compress are found as before). */
  static __thread uint32_t htab[(size_t)1 << HLOG];
  static __thread uint32_t hbase = 0;
  if (hbase > 0xffffffffu - (uint32_t)n - 2u) {
    memset(htab, 0, sizeof(htab));
    hbase = 0;
  }
  const uint32_t base = hbase;
  hbase += (uint32_t)n + 1u;
  size_t ip = 0, op = 0, anchor = 0;
  unsigned misses = 1u << 5;
  while (ip + 2 < n) {
    const uint32_t v = ((uint32_t)in[ip] << 16) | ((uint32_t)in[ip + 1] << 8) | in[ip + 2];
    const uint32_t h = ((v * 2654435761u) >> (32 - HLOG)) & ((1u << HLOG) - 1);
    const uint32_t cand = htab[h];
    htab[h] = base + (uint32_t)ip + 1u;
    const size_t ref = cand > base ? (size_t)(cand - base) - 1 : (size_t)-1;
    if (ref == (size_t)-1 || ip - ref > MAX_OFF || in[ref] != in[ip] || in[ref + 1] != in[ip + 1] || in[ref + 2] != in[ip + 2]) {
      ip += misses++ >> 5;
      continue;
    }
    misses = 1u << 5;
    size_t maxlen = n - ip;
    if (maxlen > MAX_REF) maxlen = MAX_REF;
    size_t len = 3;
    while (len + 8 <= maxlen) {
      uint64_t x, y;
      memcpy(&x, in + ref + len, 8);
      memcpy(&y, in + ip + len, 8);
      if (x != y) {
        len += (size_t)(__builtin_ctzll(x ^ y) >> 3);
        goto extended;
      }
      len += 8;
    }
    while (len < maxlen && in[ref + len] == in[ip + len]) ++len;
  extended:
    if (anchor < ip) {
      op = lzf_put_literals(in, anchor, ip, out, op, out_len);
      if (op == 0) return 0;
    }
    if (op + 3 > out_len) return 0;
    {
      const size_t off = ip - ref - 1, l = len - 2;
      if (l < 7) {
        out[op++] = (unsigned char)((l << 5) | (off >> 8));
      } else {
        out[op++] = (unsigned char)((7u << 5) | (off >> 8));
        out[op++] = (unsigned char)(l - 7);
      }
      out[op++] = (unsigned char)(off & 0xff);
    }
    /* index the last positions covered by the match so that later data can refer to them */
    if (len > 3 && ip + len + 2 < n) {
      const size_t q = ip + len - 2;
      const uint32_t w = ((uint32_t)in[q] << 16) | ((uint32_t)in[q + 1] << 8) | in[q + 2];
      htab[((w * 2654435761u) >> (32 - HLOG)) & ((1u << HLOG) - 1)] = base + (uint32_t)q + 1u;
    }
    ip += len;
    anchor = ip;
  }
  if (anchor < n) {
    op = lzf_put_literals(in, anchor, n, out, op, out_len);
    if (op == 0) return 0;
  }
  return op;
}

size_t dio_lzf_decompress(const void* in_, size_t n, void* out_, size_t out_len) {
  const unsigned char* ip = (const unsigned char*)in_;
  const unsigned char* const iend = ip + n;
  unsigned char* out = (unsigned char*)out_;
  size_t op = 0;
  while (ip < iend) {
    unsigned c = *ip++;
    if (c < 32) {
      ++c;
      if (op + c > out_len || ip + c > iend) return 0;
      memcpy(out + op, ip, c);
      op += c;
      ip += c;
    } else {
      size_t len = c >> 5;
      if (len == 7) {
        if (ip >= iend) return 0;
        len += *ip++;
      }
      if (ip >= iend) return 0;
      const size_t dist = (((size_t)(c & 31)) << 8 | *ip++) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return 0;
      const unsigned char* src = out + op - dist;
      for (size_t i = 0; i < len; ++i) out[op + i] = src[i]; /* may overlap: byte by byte */
      op += len;
    }
  }
  return op;
}

/* ---- the HDF5 filter around it (same contract as h5py's lzf_filter.c) --------------------------- */
static size_t lzf_h5_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size,
                            void** buf) {
  if (!(flags & H5Z_FLAG_REVERSE)) { /* compress; "does not shrink" = failure of an optional filter */
    void* out = malloc(nbytes ? nbytes : 1);
    if (!out) return 0;
    const size_t got = dio_lzf_compress(*buf, nbytes, out, nbytes > 0 ? nbytes - 1 : 0);
    if (got == 0) { free(out); return 0; }
    free(*buf);
    *buf = out;
    *buf_size = nbytes;
    return got;
  }
  size_t cap = (cd_nelmts >= 3 && cd_values[2] != 0) ? cd_values[2] : *buf_size;
  for (int attempt = 0; attempt < 32; ++attempt) {
    void* out = malloc(cap ? cap : 1);
    if (!out) return 0;
    const size_t got = dio_lzf_decompress(*buf, nbytes, out, cap);
    if (got != 0) {
      free(*buf);
      *buf = out;
      *buf_size = cap;
      return got;
    }
    free(out);
    cap = cap ? cap * 2 : 4096;
  }
  return 0;
}

static herr_t lzf_h5_set_local(hid_t dcpl, hid_t type, hid_t space) {
  unsigned flags;
  size_t nelem = 8;
  unsigned values[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hsize_t chunk[32];
  (void)space;
  if (H5Pget_filter_by_id2(dcpl, DIO_LZF_FILTER, &flags, &nelem, values, 0, NULL, NULL) < 0) return -1;
  if (nelem < 3) nelem = 3;
  if (values[0] == 0) values[0] = DIO_LZF_REVISION;
  if (values[1] == 0) values[1] = DIO_LZF_VERSION;
  const int nd = H5Pget_chunk(dcpl, 32, chunk);
  if (nd < 0) return -1;
  size_t bytes = H5Tget_size(type);
  if (bytes == 0) return -1;
  for (int i = 0; i < nd; ++i) bytes *= (size_t)chunk[i];
  values[2] = (unsigned)bytes;
  return H5Pmodify_filter(dcpl, DIO_LZF_FILTER, flags, nelem, values) < 0 ? -1 : 0;
}

static const H5Z_class2_t g_lzf_class = {H5Z_CLASS_T_VERS, (H5Z_filter_t)DIO_LZF_FILTER, 1, 1, "lzf", NULL,
                                         (H5Z_set_local_func_t)lzf_h5_set_local, (H5Z_func_t)lzf_h5_filter};


/* ---- bitshuffle + LZ4 (HDF5 filter 32008) --------------------------------------------------------------------
 * The codec the reference asks h5py for when `truncate` is on (drift/core/beamtransfer.py:548-555:
 * bitshuffle.h5.H5FILTER with H5_COMPRESS_LZ4).  Restated from the published formats — the bitshuffle library
 * (K. Masui, https://github.com/kiyo-masui/bitshuffle: src/bitshuffle_core.c, src/bshuf_h5filter.c, 0.3/0.4 series) and
 * the LZ4 block format (lz4_Block_format.md); neither library is linked.
 *   bit transpose of a block of n elements (n a multiple of 8) of `es` bytes:
 *     out[(b * 8 + k) * (n / 8) + i], bit j  =  bit k of byte b of element 8 i + j        (LSB first)
 *   blocks: `blk` elements each (default 8192 / es rounded down to a multiple of 8, at least 8... the chunk header names
 *     it), then one shorter block of the remaining elements rounded down to a multiple of 8, then the last (< 8)
 *     elements copied as they are
 *   LZ4 chunk: 8-byte big-endian uncompressed size, 4-byte big-endian block size in BYTES, then per block a 4-byte
 *     big-endian compressed length and that many bytes of LZ4 block data; the leftover bytes follow uncompressed
 *   without LZ4 (cd_values[4] == 0): the transposed blocks, no header.
 * Pinned in tests/test_storage_hdf5.py on the transform and the LZ4 blocks of the real libraries where the image has
 * them (imagecodecs under /opt/conda wraps both); the chunk framing follows bshuf_h5filter.c and is unpinned (no
 * bitshuffle HDF5 plugin in this image). */
#define DIO_BSHUF_FILTER 32008
#define DIO_BSHUF_LZ4 2
#define DIO_BSHUF_TARGET 8192

/* 8 x 8 bit-matrix transpose of eight bytes packed little-endian in a 64-bit word (three masked swaps; its own inverse):
 * byte k of the result holds bit k of the eight input bytes, bit j from byte j */
static inline uint64_t bshuf_t8x8(uint64_t x) {
  uint64_t t;
  t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAULL;  x = x ^ t ^ (t << 7);
  t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCULL; x = x ^ t ^ (t << 14);
  t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ULL; x = x ^ t ^ (t << 28);
  return x;
}
/* ---- forward transform, vectorised: (1) bytes b of all elements into rows (for 16-byte elements — complex128, every large
 * dataset of the products — 16 x 16 byte transposes out of unpacks), (2) each row of n bytes into its 8 bit rows with
 * movemask: bit 7 of 16 (AVX2: 32) bytes at once, then shift the bytes left by one.  Same output as the scalar form
 * (bshuf_trans_scalar, kept for the tails and as the statement of the layout); 0.8 -> 3+ GB/s per thread. */
static void bshuf_trans_scalar(const unsigned char* in, unsigned char* out, size_t n, size_t es) {
  const size_t nr = n / 8;
  for (size_t b = 0; b < es; ++b)
    for (size_t i = 0; i < nr; ++i) {
      uint64_t x = 0;
      for (int j = 0; j < 8; ++j) x |= (uint64_t)in[(8 * i + j) * es + b] << (8 * j);
      x = bshuf_t8x8(x);
      for (int k = 0; k < 8; ++k) out[(b * 8 + k) * nr + i] = (unsigned char)(x >> (8 * k));
    }
}
/* rows[b * n + i] = in[i * 16 + b] */
static void bshuf_rows16(const unsigned char* in, unsigned char* rows, size_t n) {
  size_t i = 0;
  for (; i + 16 <= n; i += 16) {
    __m128i a[16], t[16];
    for (int r = 0; r < 16; ++r) a[r] = _mm_loadu_si128((const __m128i*)(in + (i + (size_t)r) * 16));
    /* each stage interleaves register r with register r + d (d = 8, 4, 2, 1): one bit of the element index moves into
     * the byte position, the top bit of the byte index comes out as the register index */
    for (int r = 0; r < 8; ++r) { t[r] = _mm_unpacklo_epi8(a[r], a[r + 8]); t[r + 8] = _mm_unpackhi_epi8(a[r], a[r + 8]); }
    for (int r = 0; r < 16; ++r) if (!(r & 4)) { a[r] = _mm_unpacklo_epi8(t[r], t[r + 4]); a[r + 4] = _mm_unpackhi_epi8(t[r], t[r + 4]); }
    for (int r = 0; r < 16; ++r) if (!(r & 2)) { t[r] = _mm_unpacklo_epi8(a[r], a[r + 2]); t[r + 2] = _mm_unpackhi_epi8(a[r], a[r + 2]); }
    for (int r = 0; r < 16; ++r) if (!(r & 1)) { a[r] = _mm_unpacklo_epi8(t[r], t[r + 1]); a[r + 1] = _mm_unpackhi_epi8(t[r], t[r + 1]); }
    for (int c = 0; c < 16; ++c) _mm_storeu_si128((__m128i*)(rows + (size_t)c * n + i), a[c]);
  }
  for (; i < n; ++i)
    for (int c = 0; c < 16; ++c) rows[(size_t)c * n + i] = in[i * 16 + (size_t)c];
}
static void bshuf_bits_sse2(const unsigned char* rows, unsigned char* out, size_t n, size_t es) {
  const size_t nr = n / 8;
  for (size_t b = 0; b < es; ++b) {
    const unsigned char* row = rows + b * n;
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
      __m128i x = _mm_loadu_si128((const __m128i*)(row + i));
      for (int k = 7; k >= 0; --k) {
        const uint16_t m = (uint16_t)_mm_movemask_epi8(x);
        memcpy(out + (b * 8 + (size_t)k) * nr + i / 8, &m, 2);
        x = _mm_add_epi8(x, x);
      }
    }
    for (; i + 8 <= n; i += 8) {
      uint64_t x;
      memcpy(&x, row + i, 8);
      x = bshuf_t8x8(x);
      for (int k = 0; k < 8; ++k) out[(b * 8 + (size_t)k) * nr + i / 8] = (unsigned char)(x >> (8 * k));
    }
  }
}
__attribute__((target("avx2"))) static void bshuf_bits_avx2(const unsigned char* rows, unsigned char* out, size_t n, size_t es) {
  const size_t nr = n / 8;
  for (size_t b = 0; b < es; ++b) {
    const unsigned char* row = rows + b * n;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
      __m256i x = _mm256_loadu_si256((const __m256i*)(row + i));
      for (int k = 7; k >= 0; --k) {
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(x);
        memcpy(out + (b * 8 + (size_t)k) * nr + i / 8, &m, 4);
        x = _mm256_add_epi8(x, x);
      }
    }
    for (; i + 8 <= n; i += 8) {
      uint64_t x;
      memcpy(&x, row + i, 8);
      x = bshuf_t8x8(x);
      for (int k = 0; k < 8; ++k) out[(b * 8 + (size_t)k) * nr + i / 8] = (unsigned char)(x >> (8 * k));
    }
  }
}
static void bshuf_trans(const unsigned char* in, unsigned char* out, size_t n, size_t es) {
  static __thread unsigned char rows_small[DIO_BSHUF_TARGET + 256];
  static int have_avx2 = -1;
  if (have_avx2 < 0) have_avx2 = __builtin_cpu_supports("avx2") ? 1 : 0;
  if (n < 16 || n % 8) {
    bshuf_trans_scalar(in, out, n, es);
    return;
  }
  unsigned char* rows = n * es <= sizeof(rows_small) ? rows_small : (unsigned char*)malloc(n * es);
  if (!rows) {
    bshuf_trans_scalar(in, out, n, es);
    return;
  }
  if (es == 16) {
    bshuf_rows16(in, rows, n);
  } else {
    for (size_t i = 0; i < n; ++i)
      for (size_t b = 0; b < es; ++b) rows[b * n + i] = in[i * es + b];
  }
  if (have_avx2) bshuf_bits_avx2(rows, out, n, es);
  else bshuf_bits_sse2(rows, out, n, es);
  if (rows != rows_small) free(rows);
}
static void bshuf_untrans(const unsigned char* in, unsigned char* out, size_t n, size_t es) {
  const size_t nr = n / 8;
  for (size_t b = 0; b < es; ++b)
    for (size_t i = 0; i < nr; ++i) {
      uint64_t x = 0;
      for (int k = 0; k < 8; ++k) x |= (uint64_t)in[(b * 8 + k) * nr + i] << (8 * k);
      x = bshuf_t8x8(x);
      for (int j = 0; j < 8; ++j) out[(8 * i + j) * es + b] = (unsigned char)(x >> (8 * j));
    }
}
size_t dio_bitshuffle(const void* in, void* out, size_t nelem, size_t elem_size, int inverse) {
  if (!in || !out || elem_size == 0 || nelem % 8) return 0;
  if (inverse) bshuf_untrans((const unsigned char*)in, (unsigned char*)out, nelem, elem_size);
  else bshuf_trans((const unsigned char*)in, (unsigned char*)out, nelem, elem_size);
  return nelem * elem_size;
}

/* LZ4 block decoder: sequences of (token, literals, 2-byte little-endian offset, match); returns 0 on malformed input */
size_t dio_lz4_decompress(const void* in_, size_t in_len, void* out_, size_t out_cap) {
  const unsigned char* ip = (const unsigned char*)in_;
  const unsigned char* const iend = ip + in_len;
  unsigned char* op = (unsigned char*)out_;
  unsigned char* const oend = op + out_cap;
  if (in_len == 0) return 0;
  for (;;) {
    if (ip >= iend) return 0;
    const unsigned token = *ip++;
    size_t ll = token >> 4;
    if (ll == 15) {
      unsigned char c;
      do {
        if (ip >= iend) return 0;
        c = *ip++;
        ll += c;
      } while (c == 255);
    }
    if ((size_t)(iend - ip) < ll || (size_t)(oend - op) < ll) return 0;
    memcpy(op, ip, ll);
    op += ll;
    ip += ll;
    if (ip >= iend) break; /* the last sequence has literals only */
    if (iend - ip < 2) return 0;
    const size_t off = (size_t)ip[0] | ((size_t)ip[1] << 8);
    ip += 2;
    if (off == 0 || off > (size_t)(op - (unsigned char*)out_)) return 0;
    size_t ml = token & 15u;
    if (ml == 15) {
      unsigned char c;
      do {
        if (ip >= iend) return 0;
        c = *ip++;
        ml += c;
      } while (c == 255);
    }
    ml += 4;
    if ((size_t)(oend - op) < ml) return 0;
    const unsigned char* m = op - off;
    for (size_t i = 0; i < ml; ++i) op[i] = m[i]; /* byte by byte: the match may overlap its own output */
    op += ml;
  }
  return (size_t)(op - (unsigned char*)out_);
}

/* LZ4 block encoder (greedy, one hash probe per position).  End-of-block rules of the format: the last match starts at
 * least 12 bytes before the end, the last 5 bytes are literals.  Returns 0 if the output does not fit. */
static size_t lz4_put_len(unsigned char* op, unsigned char* oend, size_t v, unsigned char** out) {
  while (v >= 255) {
    if (op >= oend) return 0;
    *op++ = 255;
    v -= 255;
  }
  if (op >= oend) return 0;
  *op++ = (unsigned char)v;
  *out = op;
  return 1;
}
size_t dio_lz4_compress(const void* in_, size_t in_len, void* out_, size_t out_cap) {
  const unsigned char* const base = (const unsigned char*)in_;
  const unsigned char* ip = base;
  const unsigned char* anchor = base;
  const unsigned char* const iend = base + in_len;
  unsigned char* op = (unsigned char*)out_;
  unsigned char* const oend = op + out_cap;
  /* The hash table is per thread and never cleared between calls (the bitshuffle filter compresses 8 KB blocks, the
   * table is 16 KB): an entry holds tbase + position + 1 and counts only while it exceeds the tbase of this call.  A run
   * of misses widens the step (as the LZ4 library does): the high bit rows of shuffled doubles are noise. */
  enum { LZ4_HLOG = 12 };
  static __thread uint32_t table[1 << LZ4_HLOG];
  static __thread uint32_t tbase = 0;
  if (in_len >= 0x7fffffffu) return 0;
  if (tbase > 0xffffffffu - (uint32_t)in_len - 2u) {
    memset(table, 0, sizeof(table));
    tbase = 0;
  }
  const uint32_t tb = tbase;
  tbase += (uint32_t)in_len + 1u;
  if (in_len >= 13) {
    const unsigned char* const mflimit = iend - 12; /* no match may start beyond this */
    const unsigned char* const matchlimit = iend - 5;
    unsigned misses = 1u << 6;
    while (ip <= mflimit) {
      uint32_t seq;
      memcpy(&seq, ip, 4);
      const uint32_t h = (seq * 2654435761u) >> (32 - LZ4_HLOG);
      const uint32_t e = table[h];
      table[h] = tb + (uint32_t)(ip - base) + 1u;
      uint32_t cseq = ~seq;
      const unsigned char* m = base;
      if (e > tb) {
        m = base + (e - tb - 1u);
        memcpy(&cseq, m, 4);
      }
      if (cseq != seq || (size_t)(ip - m) > 65535) {
        ip += misses++ >> 6;
        continue;
      }
      misses = 1u << 6;
      size_t ml = 4;
      while (ip + ml + 8 <= matchlimit) {
        uint64_t x, y;
        memcpy(&x, ip + ml, 8);
        memcpy(&y, m + ml, 8);
        if (x != y) {
          ml += (size_t)(__builtin_ctzll(x ^ y) >> 3);
          goto extended;
        }
        ml += 8;
      }
      while (ip + ml < matchlimit && ip[ml] == m[ml]) ++ml;
    extended:;
      const size_t ll = (size_t)(ip - anchor);
      if (op >= oend) return 0;
      unsigned char* tok = op++;
      *tok = (unsigned char)((ll >= 15 ? 15 : ll) << 4);
      if (ll >= 15 && !lz4_put_len(op, oend, ll - 15, &op)) return 0;
      if ((size_t)(oend - op) < ll + 2) return 0;
      memcpy(op, anchor, ll);
      op += ll;
      const size_t off = (size_t)(ip - m);
      *op++ = (unsigned char)(off & 255);
      *op++ = (unsigned char)(off >> 8);
      const size_t mc = ml - 4;
      *tok |= (unsigned char)(mc >= 15 ? 15 : mc);
      if (mc >= 15 && !lz4_put_len(op, oend, mc - 15, &op)) return 0;
      ip += ml;
      anchor = ip;
    }
  }
  const size_t ll = (size_t)(iend - anchor);
  if (op >= oend) return 0;
  unsigned char* tok = op++;
  *tok = (unsigned char)((ll >= 15 ? 15 : ll) << 4);
  if (ll >= 15 && !lz4_put_len(op, oend, ll - 15, &op)) return 0;
  if ((size_t)(oend - op) < ll) return 0;
  memcpy(op, anchor, ll);
  op += ll;
  return (size_t)(op - (unsigned char*)out_);
}

static size_t bshuf_default_block(size_t es) {
  size_t b = DIO_BSHUF_TARGET / es;
  b = (b / 8) * 8;
  return b < 8 ? 8 : b;
}
static void put_be32(unsigned char* p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
static uint32_t get_be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* a whole buffer, block by block as bshuf_bitshuffle / bshuf_bitunshuffle do: full blocks, one block of the remaining
 * elements rounded down to a multiple of eight, the last (< 8) elements copied */
size_t dio_bitshuffle_blocked(const void* in_, void* out_, size_t nelem, size_t es, size_t block_elems, int inverse) {
  if (!in_ || !out_ || es == 0) return 0;
  const size_t b = block_elems ? block_elems : bshuf_default_block(es);
  if (b % 8) return 0;
  size_t done = 0;
  while (done < nelem) {
    const size_t cur = nelem - done >= b ? b : ((nelem - done) / 8) * 8;
    if (cur == 0) break;
    if (inverse) bshuf_untrans((const unsigned char*)in_ + done * es, (unsigned char*)out_ + done * es, cur, es);
    else bshuf_trans((const unsigned char*)in_ + done * es, (unsigned char*)out_ + done * es, cur, es);
    done += cur;
  }
  memcpy((unsigned char*)out_ + done * es, (const unsigned char*)in_ + done * es, (nelem - done) * es);
  return nelem * es;
}

/* one chunk, bitshuffle + LZ4: returns the encoded size, 0 if it does not fit `cap` */
size_t dio_bshuf_lz4_encode(const void* in_, size_t nbytes, size_t es, size_t block_elems, void* out_, size_t cap) {
  if (es == 0 || nbytes % es) return 0;
  const unsigned char* in = (const unsigned char*)in_;
  unsigned char* out = (unsigned char*)out_;
  const size_t n = nbytes / es;
  const size_t blk = block_elems ? block_elems : bshuf_default_block(es);
  if (blk % 8) return 0;
  if (cap < 12) return 0;
  for (int i = 0; i < 8; ++i) out[i] = (unsigned char)((uint64_t)nbytes >> (8 * (7 - i)));
  put_be32(out + 8, (uint32_t)(blk * es));
  size_t op = 12, done = 0;
  unsigned char* tmp = (unsigned char*)malloc(blk * es);
  if (!tmp) return 0;
  while (done < n) {
    size_t cur = n - done >= blk ? blk : ((n - done) / 8) * 8;
    if (cur == 0) break;
    bshuf_trans(in + done * es, tmp, cur, es);
    if (cap - op < 4) { free(tmp); return 0; }
    const size_t got = dio_lz4_compress(tmp, cur * es, out + op + 4, cap - op - 4);
    if (got == 0) { free(tmp); return 0; }
    put_be32(out + op, (uint32_t)got);
    op += 4 + got;
    done += cur;
  }
  free(tmp);
  const size_t left = (n - done) * es;
  if (cap - op < left) return 0;
  memcpy(out + op, in + done * es, left);
  return op + left;
}
/* returns the decoded size (what the header says), 0 on malformed input or if it does not fit */
size_t dio_bshuf_lz4_decode(const void* in_, size_t in_len, size_t es, void* out_, size_t cap) {
  const unsigned char* in = (const unsigned char*)in_;
  unsigned char* out = (unsigned char*)out_;
  if (es == 0 || in_len < 12) return 0;
  uint64_t nbytes = 0;
  for (int i = 0; i < 8; ++i) nbytes = (nbytes << 8) | in[i];
  const size_t blkb = get_be32(in + 8);
  if (nbytes > cap || nbytes % es || blkb == 0 || blkb % (8 * es)) return 0;
  const size_t n = (size_t)nbytes / es, blk = blkb / es;
  size_t ip = 12, done = 0;
  unsigned char* tmp = (unsigned char*)malloc(blkb);
  if (!tmp) return 0;
  while (done < n) {
    size_t cur = n - done >= blk ? blk : ((n - done) / 8) * 8;
    if (cur == 0) break;
    if (in_len - ip < 4) { free(tmp); return 0; }
    const size_t clen = get_be32(in + ip);
    ip += 4;
    if (in_len - ip < clen || dio_lz4_decompress(in + ip, clen, tmp, cur * es) != cur * es) { free(tmp); return 0; }
    ip += clen;
    bshuf_untrans(tmp, out + done * es, cur, es);
    done += cur;
  }
  free(tmp);
  const size_t left = (n - done) * es;
  if (in_len - ip < left) return 0;
  memcpy(out + done * es, in + ip, left);
  return (size_t)nbytes;
}

static size_t bshuf_h5_filter(unsigned flags, size_t cd_nelmts, const unsigned cd_values[], size_t nbytes, size_t* buf_size,
                              void** buf) {
  if (cd_nelmts < 3) return 0;
  const size_t es = cd_values[2];
  const size_t blk = cd_nelmts > 3 ? cd_values[3] : 0;
  const unsigned comp = cd_nelmts > 4 ? cd_values[4] : 0;
  if (es == 0) return 0;
  if (comp == DIO_BSHUF_LZ4) {
    if (flags & H5Z_FLAG_REVERSE) {
      if (nbytes < 12) return 0;
      uint64_t want = 0;
      for (int i = 0; i < 8; ++i) want = (want << 8) | ((const unsigned char*)*buf)[i];
      void* out = malloc(want ? (size_t)want : 1);
      if (!out) return 0;
      const size_t got = dio_bshuf_lz4_decode(*buf, nbytes, es, out, (size_t)want);
      if (got == 0 && want != 0) { free(out); return 0; }
      free(*buf);
      *buf = out;
      *buf_size = (size_t)want;
      return (size_t)want;
    }
    const size_t b = blk ? blk : bshuf_default_block(es);
    const size_t cap = nbytes + 12 + 4 * (nbytes / (b * es) + 2) + nbytes / 255 + 64 * (nbytes / (b * es) + 2);
    void* out = malloc(cap);
    if (!out) return 0;
    const size_t got = dio_bshuf_lz4_encode(*buf, nbytes, es, blk, out, cap);
    if (got == 0) { free(out); return 0; }
    free(*buf);
    *buf = out;
    *buf_size = cap;
    return got;
  }
  if (comp != 0) return 0; /* zstd and anything newer: not restated here */
  /* plain bitshuffle: same size, block by block */
  if (nbytes % es) return 0;
  void* out = malloc(nbytes ? nbytes : 1);
  if (!out) return 0;
  if (dio_bitshuffle_blocked(*buf, out, nbytes / es, es, blk, (flags & H5Z_FLAG_REVERSE) ? 1 : 0) != nbytes) { free(out); return 0; }
  free(*buf);
  *buf = out;
  *buf_size = nbytes;
  return nbytes;
}
static herr_t bshuf_h5_set_local(hid_t dcpl, hid_t type, hid_t space) {
  unsigned flags;
  size_t nelem = 8;
  unsigned values[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)space;
  if (H5Pget_filter_by_id2(dcpl, DIO_BSHUF_FILTER, &flags, &nelem, values, 0, NULL, NULL) < 0) return -1;
  /* the user gives (block size, compression); the library's filter stores (major, minor, element size, block, compression) */
  unsigned user[2] = {nelem > 0 ? values[0] : 0, nelem > 1 ? values[1] : 0};
  if (nelem >= 5) { user[0] = values[3]; user[1] = values[4]; }
  const size_t es = H5Tget_size(type);
  if (es == 0) return -1;
  unsigned out[5] = {0, 4, (unsigned)es, user[0], user[1]};
  return H5Pmodify_filter(dcpl, DIO_BSHUF_FILTER, flags, 5, out) < 0 ? -1 : 0;
}
static const H5Z_class2_t g_bshuf_class = {H5Z_CLASS_T_VERS, (H5Z_filter_t)DIO_BSHUF_FILTER, 1, 1, "bitshuffle (own restatement)", NULL,
                                           (H5Z_set_local_func_t)bshuf_h5_set_local, (H5Z_func_t)bshuf_h5_filter};

static void init_once(void) {
  H5open();
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL); /* errors are reported through return codes + dio_last_error */
  if (H5Zfilter_avail(DIO_LZF_FILTER) <= 0) H5Zregister(&g_lzf_class);
  if (H5Zfilter_avail(DIO_BSHUF_FILTER) <= 0) H5Zregister(&g_bshuf_class);
}

int dio_hdf5_version(void) {
  unsigned a = 0, b = 0, c = 0;
  pthread_once(&g_once, init_once);
  H5get_libversion(&a, &b, &c);
  return (int)(a * 10000 + b * 100 + c);
}

/* ---- types --------------------------------------------------------------------------------------- */
static size_t dtype_size(int dtype) {
  switch (dtype) {
    case DIO_F64: return 8;
    case DIO_C128: return 16;
    case DIO_I64: return 8;
    case DIO_I32: return 4;
    case DIO_BOOL: return 1;
    case DIO_F32: return 4;
    default: return 0;
  }
}

/* a new type id (caller closes) */
static hid_t make_type(int dtype) {
  switch (dtype) {
    case DIO_F64: return H5Tcopy(H5T_IEEE_F64LE);
    case DIO_F32: return H5Tcopy(H5T_IEEE_F32LE);
    case DIO_I64: return H5Tcopy(H5T_STD_I64LE);
    case DIO_I32: return H5Tcopy(H5T_STD_I32LE);
    case DIO_C128: {
      hid_t t = H5Tcreate(H5T_COMPOUND, 16);
      if (t < 0) return t;
      H5Tinsert(t, "r", 0, H5T_IEEE_F64LE);
      H5Tinsert(t, "i", 8, H5T_IEEE_F64LE);
      return t;
    }
    case DIO_BOOL: {
      hid_t t = H5Tenum_create(H5T_STD_I8LE);
      if (t < 0) return t;
      signed char v = 0;
      H5Tenum_insert(t, "FALSE", &v);
      v = 1;
      H5Tenum_insert(t, "TRUE", &v);
      return t;
    }
    case DIO_STR: {
      hid_t t = H5Tcopy(H5T_C_S1);
      if (t < 0) return t;
      H5Tset_size(t, H5T_VARIABLE);
      H5Tset_cset(t, H5T_CSET_UTF8);
      return t;
    }
    default: return -1;
  }
}

static int classify(hid_t t) {
  const H5T_class_t c = H5Tget_class(t);
  const size_t sz = H5Tget_size(t);
  if (c == H5T_FLOAT) return sz == 8 ? DIO_F64 : (sz == 4 ? DIO_F32 : DIO_OTHER);
  if (c == H5T_INTEGER) return sz == 8 ? DIO_I64 : (sz == 4 ? DIO_I32 : (sz == 1 ? DIO_BOOL : DIO_OTHER));
  if (c == H5T_ENUM) return sz == 1 ? DIO_BOOL : DIO_OTHER;
  if (c == H5T_STRING) return DIO_STR;
  if (c == H5T_COMPOUND && H5Tget_nmembers(t) == 2 && sz == 16) {
    char* a = H5Tget_member_name(t, 0);
    char* b = H5Tget_member_name(t, 1);
    const int ok = a && b && strcmp(a, "r") == 0 && strcmp(b, "i") == 0;
    if (a) H5free_memory(a);
    if (b) H5free_memory(b);
    return ok ? DIO_C128 : DIO_OTHER;
  }
  return DIO_OTHER;
}

/* ---- files ----------------------------------------------------------------------------------------- */
int dio_open(const char* path, const char* mode, int64_t* file_out) {
  pthread_once(&g_once, init_once);
  if (!path || !mode || !file_out) return fail("dio_open: bad argument");
  hid_t f = -1;
  LOCK();
  if (strcmp(mode, "w") == 0) {
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  } else if (strcmp(mode, "r") == 0) {
    f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  } else if (strcmp(mode, "r+") == 0) {
    f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
  }
  UNLOCK();
  if (f < 0) return fail("dio_open: cannot open '%s' with mode '%s'", path, mode);
  *file_out = (int64_t)f;
  return 0;
}

/* Create a file whose final size is roughly known.  (HDF5's in-memory "core" driver was tried for the small
 * per-m files — one write at close instead of one per chunk — and measured 10x SLOWER than the default driver
 * on create + close: it zero-fills and copies its image.  The hint is kept in the ABI and unused.) */
int dio_create(const char* path, uint64_t expected_bytes, int64_t* file_out) {
  (void)expected_bytes;
  return dio_open(path, "w", file_out);
}

int dio_close(int64_t file) {
  LOCK();
  const herr_t rc = H5Fclose((hid_t)file);
  UNLOCK();
  return rc < 0 ? fail("dio_close failed") : 0;
}

int dio_exists(int64_t file, const char* name) {
  LOCK();
  const htri_t r = H5Lexists((hid_t)file, name, H5P_DEFAULT);
  UNLOCK();
  return r < 0 ? fail("dio_exists failed") : (r > 0);
}

/* ---- datasets --------------------------------------------------------------------------------------- */
/* gather + compress the chunks c = c_begin + t, c_begin + t + nt, ... of one block into their slots */
typedef struct {
  const unsigned char* src;
  size_t esz, cbytes;
  int ndim, compression;
  hsize_t dims[DIO_MAX_DIMS], cdims[DIO_MAX_DIMS], nchunk[DIO_MAX_DIMS];
  size_t stride[DIO_MAX_DIMS];
  size_t c_begin, c_end;
  unsigned char** slots; /* per chunk of the block: 2 * cbytes, gathered chunk then packed chunk */
  size_t* zbytes;
  uint32_t* zmask;
  hsize_t* offs;         /* per chunk: DIO_MAX_DIMS offsets */
  unsigned char* raw;    /* per chunk: 1 = hand over the gathered bytes (stored raw), 0 = the packed ones */
} chunk_job;
typedef struct { chunk_job* J; int t, nt; } chunk_worker_arg;

static void* chunk_worker(void* argp) {
  chunk_worker_arg* A = (chunk_worker_arg*)argp;
  chunk_job* J = A->J;
  const int ndim = J->ndim;
  const size_t esz = J->esz, cbytes = J->cbytes;
  for (size_t c = J->c_begin + (size_t)A->t; c < J->c_end; c += (size_t)A->nt) {
    const size_t k = c - J->c_begin;
    unsigned char* cbuf = J->slots[k];
    unsigned char* zbuf = cbuf + cbytes;
    hsize_t* off = J->offs + k * DIO_MAX_DIMS;
    size_t ext[DIO_MAX_DIMS];
    int full = 1;
    size_t rem = c;
    for (int i = ndim - 1; i >= 0; --i) {   /* row-major chunk index, last axis fastest (the order of the serial loop) */
      const size_t ci = rem % (size_t)J->nchunk[i];
      rem /= (size_t)J->nchunk[i];
      off[i] = (hsize_t)ci * J->cdims[i];
    }
    for (int i = 0; i < ndim; ++i) {
      ext[i] = (size_t)((off[i] + J->cdims[i] <= J->dims[i]) ? J->cdims[i] : J->dims[i] - off[i]);
      if (ext[i] != J->cdims[i]) full = 0;
    }
    if (!full) memset(cbuf, 0, cbytes);
    /* copy runs along the last axis */
    size_t idx[DIO_MAX_DIMS] = {0};
    const size_t run = ext[ndim - 1] * esz;
    for (;;) {
      size_t so = 0, dof = 0, cs = 1;
      for (int i = ndim - 1; i >= 0; --i) {
        so += ((size_t)off[i] + idx[i]) * J->stride[i];
        dof += idx[i] * cs;
        cs *= (size_t)J->cdims[i];
      }
      memcpy(cbuf + dof * esz, J->src + so * esz, run);
      int d = ndim - 2;
      while (d >= 0) {
        if (++idx[d] < ext[d]) break;
        idx[d] = 0;
        --d;
      }
      if (d < 0) break;
    }
    size_t wbytes = cbytes;
    uint32_t mask = 0;
    int raw = 1;
    if (J->compression == DIO_COMP_LZF) {
      /* dense full-precision doubles do not compress: probe before paying for the whole chunk — three windows (head,
       * middle, tail), so that a chunk whose leading rows are dense and whose rest is zero padding is still packed */
      const size_t probe = cbytes < 2048 ? cbytes : 2048;
      size_t got = dio_lzf_compress(cbuf, probe, zbuf, probe - probe / 16 - 1);
      if (got == 0 && cbytes >= 3 * probe) {
        got = dio_lzf_compress(cbuf + (cbytes / 2 / 16) * 16, probe, zbuf, probe - probe / 16 - 1);
        if (got == 0) got = dio_lzf_compress(cbuf + cbytes - probe, probe, zbuf, probe - probe / 16 - 1);
      }
      if (got > 0 && probe < cbytes) got = dio_lzf_compress(cbuf, cbytes, zbuf, cbytes - 1);
      if (got > 0) {
        raw = 0;
        wbytes = got;
      } else {
        mask = 1; /* filter 0 of the pipeline was skipped for this chunk */
      }
    } else if (J->compression == DIO_COMP_BSHUF_LZ4) {
      const size_t got = dio_bshuf_lz4_encode(cbuf, cbytes, esz, 0, zbuf, cbytes - 1);
      if (got > 0) {
        raw = 0;
        wbytes = got;
      } else {
        mask = 1; /* does not shrink: stored raw, as an optional filter */
      }
    }
    J->zbytes[k] = wbytes;
    J->zmask[k] = mask;
    J->raw[k] = (unsigned char)raw;
  }
  return NULL;
}

int dio_write_dataset(int64_t file, const char* name, int dtype, int ndim, const uint64_t* shape, const uint64_t* chunks,
                      int compression, const void* data) {
  if (!name || ndim < 0 || ndim > DIO_MAX_DIMS || (ndim > 0 && !shape)) return fail("dio_write_dataset: bad argument");
  const size_t esz = dtype_size(dtype);
  if (esz == 0) return fail("dio_write_dataset: unsupported element type %d", dtype);
  if (!chunks && compression != DIO_COMP_NONE) return fail("dio_write_dataset: compression needs chunks");
  hsize_t dims[DIO_MAX_DIMS], cdims[DIO_MAX_DIMS];
  size_t total = 1, chunk_elems = 1;
  for (int i = 0; i < ndim; ++i) {
    dims[i] = shape[i];
    total *= (size_t)shape[i];
  }
  int chunked = chunks != NULL && total > 0 && ndim > 0;
  if (chunked) {
    for (int i = 0; i < ndim; ++i) {
      cdims[i] = chunks[i] < 1 ? 1 : (chunks[i] > shape[i] ? shape[i] : chunks[i]);
      chunk_elems *= (size_t)cdims[i];
    }
    if (chunk_elems * esz > 0xffffffffu) return fail("dio_write_dataset: chunk larger than 4 GiB");
  }
  if (total > 0 && !data) return fail("dio_write_dataset: no data");
  hid_t ftype = -1, space = -1, dcpl = -1, dset = -1;
  int rc = 0;
  LOCK();
  ftype = make_type(dtype);
  space = ndim == 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(ndim, dims, NULL);
  dcpl = H5Pcreate(H5P_DATASET_CREATE);
  if (ftype < 0 || space < 0 || dcpl < 0) rc = fail("dio_write_dataset: HDF5 object creation failed");
  if (rc == 0 && chunked) {
    if (H5Pset_chunk(dcpl, ndim, cdims) < 0) rc = fail("H5Pset_chunk failed");
    if (rc == 0 && compression == DIO_COMP_LZF) {
      const unsigned cd[3] = {DIO_LZF_REVISION, DIO_LZF_VERSION, (unsigned)(chunk_elems * esz)};
      if (H5Pset_filter(dcpl, DIO_LZF_FILTER, H5Z_FLAG_OPTIONAL, 3, cd) < 0) rc = fail("H5Pset_filter(lzf) failed");
    }
    if (rc == 0 && compression == DIO_COMP_BSHUF_LZ4) {
      /* what bitshuffle's own set_local leaves in the file: (major, minor, element size, block size, LZ4) */
      const unsigned cd[5] = {0, 4, (unsigned)esz, 0, DIO_BSHUF_LZ4};
      if (H5Pset_filter(dcpl, DIO_BSHUF_FILTER, H5Z_FLAG_OPTIONAL, 5, cd) < 0) rc = fail("H5Pset_filter(bitshuffle) failed");
    }
  }
  if (rc == 0) {
    dset = H5Dcreate2((hid_t)file, name, ftype, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    if (dset < 0) rc = fail("H5Dcreate2('%s') failed", name);
  }
  if (rc == 0 && total > 0 && !chunked) {
    if (H5Dwrite(dset, ftype, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) rc = fail("H5Dwrite('%s') failed", name);
  }
  UNLOCK();
  if (rc == 0 && total > 0 && chunked) {
    /* Chunk by chunk: gather (zero padded at the edges) and compress WITHOUT the lock, hand over with it.  The chunks of a
     * dataset are independent: blocks of them are gathered and compressed by a few threads at once (DRIFTMI_IO_CHUNK_THREADS,
     * default 4: a configs[2] svd file is 2.6 GB of chunks that really compress, ~13 s on one core) and then handed to HDF5
     * in order by the calling thread — the file layout does not depend on the thread count. */
    chunk_job J;
    memset(&J, 0, sizeof J);
    J.src = (const unsigned char*)data;
    J.esz = esz;
    J.ndim = ndim;
    J.compression = compression;
    J.cbytes = chunk_elems * esz;
    size_t nch = 1;
    for (int i = 0; i < ndim; ++i) {
      J.dims[i] = dims[i];
      J.cdims[i] = cdims[i];
      J.nchunk[i] = (dims[i] + cdims[i] - 1) / cdims[i];
      nch *= (size_t)J.nchunk[i];
    }
    for (int i = ndim - 1; i >= 0; --i) J.stride[i] = i == ndim - 1 ? 1 : J.stride[i + 1] * (size_t)dims[i + 1];
    int nt = 4;
    const char* e = getenv("DRIFTMI_IO_CHUNK_THREADS");
    if (e && atoi(e) > 0) nt = atoi(e);
    if (nt > 16) nt = 16;
    if ((size_t)nt > nch) nt = (int)nch;
    if (J.cbytes * nch < ((size_t)4 << 20)) nt = 1;   /* small datasets: not worth a thread */
    const size_t blk = (size_t)nt * 4;                /* chunks per block */
    J.slots = (unsigned char**)calloc(blk, sizeof(unsigned char*));
    J.zbytes = (size_t*)calloc(blk, sizeof(size_t));
    J.zmask = (uint32_t*)calloc(blk, sizeof(uint32_t));
    J.offs = (hsize_t*)calloc(blk * DIO_MAX_DIMS, sizeof(hsize_t));
    J.raw = (unsigned char*)calloc(blk, 1);
    if (!J.slots || !J.zbytes || !J.zmask || !J.offs || !J.raw) rc = fail("dio_write_dataset: out of memory");
    for (size_t k = 0; k < blk && rc == 0; ++k) {
      J.slots[k] = (unsigned char*)malloc(2 * J.cbytes);   /* [gathered chunk | packed chunk] */
      if (!J.slots[k]) rc = fail("dio_write_dataset: out of memory");
    }
    for (size_t c0 = 0; c0 < nch && rc == 0; c0 += blk) {
      J.c_begin = c0;
      J.c_end = c0 + blk < nch ? c0 + blk : nch;
      if (nt <= 1) {
        chunk_worker_arg a = {&J, 0, 1};
        chunk_worker(&a);
      } else {
        pthread_t th[16];
        chunk_worker_arg args[16];
        int started = 0;
        for (int t = 0; t < nt; ++t) {
          args[t].J = &J;
          args[t].t = t;
          args[t].nt = nt;
          if (pthread_create(&th[t], NULL, chunk_worker, &args[t]) != 0) break;
          ++started;
        }
        if (started < nt) {   /* could not start them all: the caller does the rest */
          for (int t = started; t < nt; ++t) {
            args[t].J = &J; args[t].t = t; args[t].nt = nt;
            chunk_worker(&args[t]);
          }
        }
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
      }
      LOCK();
      for (size_t c = J.c_begin; c < J.c_end && rc == 0; ++c) {
        const size_t k = c - J.c_begin;
        const unsigned char* wbuf = J.raw[k] ? J.slots[k] : J.slots[k] + J.cbytes;
        if (H5Dwrite_chunk(dset, H5P_DEFAULT, J.zmask[k], J.offs + k * DIO_MAX_DIMS, J.zbytes[k], wbuf) < 0)
          rc = fail("H5Dwrite_chunk('%s') failed", name);
      }
      UNLOCK();
    }
    if (J.slots)
      for (size_t k = 0; k < blk; ++k) free(J.slots[k]);
    free(J.slots);
    free(J.zbytes);
    free(J.zmask);
    free(J.offs);
    free(J.raw);
  }
  LOCK();
  if (dset >= 0) H5Dclose(dset);
  if (dcpl >= 0) H5Pclose(dcpl);
  if (space >= 0) H5Sclose(space);
  if (ftype >= 0) H5Tclose(ftype);
  UNLOCK();
  return rc;
}

int dio_dataset_info(int64_t file, const char* name, int* dtype, int* ndim, uint64_t* shape, uint64_t* chunks,
                     int* compression) {
  int rc = 0;
  LOCK();
  hid_t dset = H5Dopen2((hid_t)file, name, H5P_DEFAULT);
  if (dset < 0) {
    UNLOCK();
    return fail("no dataset '%s'", name);
  }
  hid_t t = H5Dget_type(dset), s = H5Dget_space(dset), p = H5Dget_create_plist(dset);
  hsize_t dims[H5S_MAX_RANK];
  const int nd = H5Sget_simple_extent_ndims(s);
  if (nd < 0 || nd > DIO_MAX_DIMS) rc = fail("dataset '%s': unsupported rank %d", name, nd);
  if (rc == 0) {
    H5Sget_simple_extent_dims(s, dims, NULL);
    if (dtype) *dtype = classify(t);
    if (ndim) *ndim = nd;
    for (int i = 0; i < nd; ++i) {
      if (shape) shape[i] = dims[i];
      if (chunks) chunks[i] = 0;
    }
    if (chunks && H5Pget_layout(p) == H5D_CHUNKED) {
      hsize_t cd[H5S_MAX_RANK];
      H5Pget_chunk(p, nd, cd);
      for (int i = 0; i < nd; ++i) chunks[i] = cd[i];
    }
    if (compression) {
      const int nf = H5Pget_nfilters(p);
      *compression = DIO_COMP_NONE;
      if (nf > 0) {
        unsigned fl;
        size_t ne = 0;
        const H5Z_filter_t id = H5Pget_filter2(p, 0, &fl, &ne, NULL, 0, NULL, NULL);
        *compression = (nf == 1 && id == DIO_LZF_FILTER) ? DIO_COMP_LZF : ((nf == 1 && id == DIO_BSHUF_FILTER) ? DIO_COMP_BSHUF_LZ4 : -1);
      }
    }
  }
  H5Pclose(p);
  H5Sclose(s);
  H5Tclose(t);
  H5Dclose(dset);
  UNLOCK();
  return rc;
}

int dio_read_dataset(int64_t file, const char* name, int dtype, const uint64_t* start, const uint64_t* count, void* out) {
  int rc = 0;
  LOCK();
  hid_t dset = H5Dopen2((hid_t)file, name, H5P_DEFAULT);
  if (dset < 0) {
    UNLOCK();
    return fail("no dataset '%s'", name);
  }
  hid_t mt = make_type(dtype), fs = H5Dget_space(dset), ms = H5S_ALL;
  if (mt < 0) rc = fail("dio_read_dataset: unsupported element type %d", dtype);
  const int nd = H5Sget_simple_extent_ndims(fs);
  if (rc == 0 && start && count && nd > 0) {
    hsize_t st[H5S_MAX_RANK], ct[H5S_MAX_RANK];
    size_t total = 1;
    for (int i = 0; i < nd; ++i) {
      st[i] = start[i];
      ct[i] = count[i];
      total *= (size_t)count[i];
    }
    if (total == 0) goto done;
    if (H5Sselect_hyperslab(fs, H5S_SELECT_SET, st, NULL, ct, NULL) < 0) rc = fail("hyperslab selection failed");
    ms = H5Screate_simple(nd, ct, NULL);
  } else if (rc == 0) {
    if (H5Sget_simple_extent_npoints(fs) == 0) goto done;
  }
  if (rc == 0 && H5Dread(dset, mt, ms, ms == H5S_ALL ? H5S_ALL : fs, H5P_DEFAULT, out) < 0)
    rc = fail("H5Dread('%s') failed (unknown filter or type conversion)", name);
done:
  if (ms != H5S_ALL && ms >= 0) H5Sclose(ms);
  if (mt >= 0) H5Tclose(mt);
  H5Sclose(fs);
  H5Dclose(dset);
  UNLOCK();
  return rc;
}

struct namebuf {
  char* buf;
  int64_t cap, used;
};

static void nb_add(struct namebuf* nb, const char* s) {
  const int64_t n = (int64_t)strlen(s);
  if (nb->buf && nb->used + n + 1 <= nb->cap) {
    memcpy(nb->buf + nb->used, s, (size_t)n);
    nb->buf[nb->used + n] = '\n';
  }
  nb->used += n + 1;
}

static herr_t link_cb(hid_t g, const char* name, const H5L_info_t* info, void* data) {
  (void)g;
  (void)info;
  nb_add((struct namebuf*)data, name);
  return 0;
}

static herr_t attr_cb(hid_t loc, const char* name, const H5A_info_t* info, void* data) {
  (void)loc;
  (void)info;
  nb_add((struct namebuf*)data, name);
  return 0;
}

int64_t dio_list(int64_t file, char* buf, int64_t buflen) {
  struct namebuf nb = {buf, buflen, 0};
  LOCK();
  const herr_t rc = H5Literate((hid_t)file, H5_INDEX_NAME, H5_ITER_INC, NULL, link_cb, &nb);
  UNLOCK();
  return rc < 0 ? fail("H5Literate failed") : nb.used;
}

int64_t dio_list_attrs(int64_t file, char* buf, int64_t buflen) {
  struct namebuf nb = {buf, buflen, 0};
  LOCK();
  const herr_t rc = H5Aiterate2((hid_t)file, H5_INDEX_NAME, H5_ITER_INC, NULL, attr_cb, &nb);
  UNLOCK();
  return rc < 0 ? fail("H5Aiterate2 failed") : nb.used;
}

/* ---- attributes ------------------------------------------------------------------------------------- */
int dio_write_attr(int64_t file, const char* name, int dtype, int ndim, const uint64_t* shape, const void* data) {
  if (!name || !data || ndim < 0 || ndim > DIO_MAX_DIMS) return fail("dio_write_attr: bad argument");
  hsize_t dims[DIO_MAX_DIMS];
  for (int i = 0; i < ndim; ++i) dims[i] = shape[i];
  int rc = 0;
  LOCK();
  hid_t t = make_type(dtype);
  hid_t s = ndim == 0 ? H5Screate(H5S_SCALAR) : H5Screate_simple(ndim, dims, NULL);
  if (t < 0 || s < 0) rc = fail("dio_write_attr: unsupported type %d", dtype);
  if (rc == 0) {
    if (H5Aexists((hid_t)file, name) > 0) H5Adelete((hid_t)file, name);
    hid_t a = H5Acreate2((hid_t)file, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    if (a < 0) {
      rc = fail("H5Acreate2('%s') failed", name);
    } else {
      const char* sp = (const char*)data;
      const void* src = dtype == DIO_STR ? (const void*)&sp : data;
      if (H5Awrite(a, t, src) < 0) rc = fail("H5Awrite('%s') failed", name);
      H5Aclose(a);
    }
  }
  if (s >= 0) H5Sclose(s);
  if (t >= 0) H5Tclose(t);
  UNLOCK();
  return rc;
}

int dio_attr_info(int64_t file, const char* name, int* dtype, int* ndim, uint64_t* shape, int64_t* strlen_out) {
  int rc = 0;
  LOCK();
  hid_t a = H5Aopen((hid_t)file, name, H5P_DEFAULT);
  if (a < 0) {
    UNLOCK();
    return fail("no attribute '%s'", name);
  }
  hid_t t = H5Aget_type(a), s = H5Aget_space(a);
  const int cls = classify(t);
  const int nd = H5Sget_simple_extent_ndims(s);
  hsize_t dims[H5S_MAX_RANK];
  if (nd < 0 || nd > DIO_MAX_DIMS) rc = fail("attribute '%s': unsupported rank", name);
  if (rc == 0) {
    if (nd > 0) H5Sget_simple_extent_dims(s, dims, NULL);
    if (dtype) *dtype = cls;
    if (ndim) *ndim = nd;
    for (int i = 0; i < nd && shape; ++i) shape[i] = dims[i];
    if (strlen_out) {
      *strlen_out = 0;
      if (cls == DIO_STR && nd == 0) {
        if (H5Tis_variable_str(t) > 0) {
          char* p = NULL;
          hid_t mt = make_type(DIO_STR);
          if (H5Aread(a, mt, &p) >= 0 && p) {
            *strlen_out = (int64_t)strlen(p);
            H5free_memory(p);
          }
          H5Tclose(mt);
        } else {
          *strlen_out = (int64_t)H5Tget_size(t);
        }
      }
    }
  }
  H5Sclose(s);
  H5Tclose(t);
  H5Aclose(a);
  UNLOCK();
  return rc;
}

int dio_read_attr(int64_t file, const char* name, int dtype, void* out, int64_t outlen) {
  int rc = 0;
  LOCK();
  hid_t a = H5Aopen((hid_t)file, name, H5P_DEFAULT);
  if (a < 0) {
    UNLOCK();
    return fail("no attribute '%s'", name);
  }
  if (dtype == DIO_STR) {
    hid_t t = H5Aget_type(a);
    if (H5Tis_variable_str(t) > 0) {
      char* p = NULL;
      hid_t mt = make_type(DIO_STR);
      if (H5Aread(a, mt, &p) < 0 || !p) {
        rc = fail("H5Aread('%s') failed", name);
      } else {
        snprintf((char*)out, (size_t)outlen, "%s", p);
        H5free_memory(p);
      }
      H5Tclose(mt);
    } else {
      const size_t sz = H5Tget_size(t);
      char* tmp = (char*)calloc(sz + 1, 1);
      hid_t mt = H5Tcopy(H5T_C_S1);
      H5Tset_size(mt, sz);
      if (!tmp || H5Aread(a, mt, tmp) < 0) rc = fail("H5Aread('%s') failed", name);
      else snprintf((char*)out, (size_t)outlen, "%s", tmp);
      H5Tclose(mt);
      free(tmp);
    }
    H5Tclose(t);
  } else {
    hid_t mt = make_type(dtype);
    if (mt < 0 || H5Aread(a, mt, out) < 0) rc = fail("H5Aread('%s') failed", name);
    if (mt >= 0) H5Tclose(mt);
  }
  H5Aclose(a);
  UNLOCK();
  return rc;
}
