/* driftio.h — C ABI of libdriftio, the product-file side of the hot path: real HDF5 files with the
 * reference's dataset names, chunk shapes, compound complex type and LZF compression, written from host
 * buffers by several threads at once.  Plain pointers and sizes; no Python / numpy types.
 *
 * Replaces (paths relative to radiocosmology/driftscan): the h5py calls of
 *   BeamTransfer._generate_mfiles      drift/core/beamtransfer.py:548-577, :649-663   beam.hdf5
 *   BeamTransfer._generate_svdfile_m   drift/core/beamtransfer.py:738-798, :927-929   svd.hdf5
 *   KLTransform.transform_save         drift/core/kltransform.py:377-433              ev_m_<m>.hdf5
 *   _collect / _collect_svd_spectrum   drift/core/kltransform.py:452-478, beamtransfer.py:931-947
 * and reads the files those calls (or this library) wrote.
 *
 * The library links the HDF5 C library (1.10.x) and registers its own implementation of the LZF filter
 * (HDF5 filter id 32000, the one h5py ships): files are readable by plain h5py, and lzf-compressed files
 * written by h5py are readable here.  libhdf5 builds are usually not thread-safe: every HDF5 call is taken
 * under one process-wide mutex INSIDE this library, while the chunk compression — the expensive part of a
 * write — runs outside it, so N writer threads compress N datasets concurrently (H5Dwrite_chunk).
 *
 * Every function returns 0 on success, < 0 on failure (dio_last_error() of the calling thread).
 */
#ifndef DRIFTIO_H
#define DRIFTIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types */
enum {
  DIO_F64 = 0,   /* IEEE double */
  DIO_C128 = 1,  /* compound {r: f64, i: f64} — h5py's complex128 */
  DIO_I64 = 2,
  DIO_I32 = 3,
  DIO_BOOL = 4,  /* h5py's numpy.bool_: enum {FALSE = 0, TRUE = 1} over int8 */
  DIO_STR = 5,   /* attributes only: variable-length UTF-8 string scalar, as h5py writes a Python str */
  DIO_F32 = 6,
  DIO_OTHER = 99
};

enum { DIO_COMP_NONE = 0, DIO_COMP_LZF = 1, DIO_COMP_BSHUF_LZ4 = 2 };

#define DIO_MAX_DIMS 8

const char* dio_last_error(void);
int dio_version(void);
/* version of the HDF5 library in use: major * 10000 + minor * 100 + release */
int dio_hdf5_version(void);

/* mode: "w" (create / truncate), "r", "r+" */
int dio_open(const char* path, const char* mode, int64_t* file_out);
/* "w" with a size hint (reserved: the default file driver is used whatever the size) */
int dio_create(const char* path, uint64_t expected_bytes, int64_t* file_out);
int dio_close(int64_t file);

/* Create dataset `name` (shape[ndim]) and write `data` (C order, contiguous, host).
 *   chunks       NULL: contiguous layout (compression must be DIO_COMP_NONE); else the chunk shape
 *   compression  DIO_COMP_LZF: each chunk is LZF-compressed by the calling thread outside the HDF5 lock
 *                (a chunk that does not shrink is stored raw with its filter-mask bit set, as HDF5's
 *                optional filters do) */
int dio_write_dataset(int64_t file, const char* name, int dtype, int ndim, const uint64_t* shape,
                      const uint64_t* chunks, int compression, const void* data);

/* dtype / shape / layout of an existing dataset; chunks[i] = 0 for a contiguous dataset; compression
 * as above, or -1 for a filter pipeline this library cannot decode */
int dio_dataset_info(int64_t file, const char* name, int* dtype, int* ndim, uint64_t* shape, uint64_t* chunks,
                     int* compression);
/* Read the hyperslab start[] / count[] (NULL, NULL: everything) into `out` as `dtype` (HDF5 converts). */
int dio_read_dataset(int64_t file, const char* name, int dtype, const uint64_t* start, const uint64_t* count,
                     void* out);
int dio_exists(int64_t file, const char* name);   /* 1 / 0 / < 0 */
/* names of the root group's members, '\n'-separated, into buf (returns the byte count needed) */
int64_t dio_list(int64_t file, char* buf, int64_t buflen);

/* attributes of the root group.  ndim = 0: scalar.  DIO_STR: data is a NUL-terminated UTF-8 string. */
int dio_write_attr(int64_t file, const char* name, int dtype, int ndim, const uint64_t* shape, const void* data);
int dio_attr_info(int64_t file, const char* name, int* dtype, int* ndim, uint64_t* shape, int64_t* strlen_out);
int dio_read_attr(int64_t file, const char* name, int dtype, void* out, int64_t outlen);
int64_t dio_list_attrs(int64_t file, char* buf, int64_t buflen);

/* The LZF codec itself (exposed for the parity tests): returns the compressed / decompressed size, 0 if the
 * output does not fit (for compression: "does not shrink"). */
size_t dio_lzf_compress(const void* in, size_t in_len, void* out, size_t out_len);
size_t dio_lzf_decompress(const void* in, size_t in_len, void* out, size_t out_len);

/* bitshuffle + LZ4, HDF5 filter 32008 — what the reference writes when `truncate` is on
 * (drift/core/beamtransfer.py:548-555); own restatement of the published formats (dm_h5io.c), registered for reading
 * and, with DIO_COMP_BSHUF_LZ4, for writing.  The pieces, exposed for the parity tests:
 *   dio_bitshuffle        bit transpose of nelem (a multiple of 8) elements of elem_size bytes; inverse != 0 undoes it
 *   dio_lz4_compress / _decompress   one LZ4 block (returns the size, 0 on failure / if it does not fit)
 *   dio_bshuf_lz4_encode / _decode   one HDF5 chunk: 12-byte header, per-block length + LZ4 data, raw tail */
size_t dio_bitshuffle(const void* in, void* out, size_t nelem, size_t elem_size, int inverse);
/* any nelem, cut into blocks as the bitshuffle library does (block_elems = 0: its default, 8192 bytes worth) */
size_t dio_bitshuffle_blocked(const void* in, void* out, size_t nelem, size_t elem_size, size_t block_elems, int inverse);
size_t dio_lz4_compress(const void* in, size_t in_len, void* out, size_t out_cap);
size_t dio_lz4_decompress(const void* in, size_t in_len, void* out, size_t out_cap);
size_t dio_bshuf_lz4_encode(const void* in, size_t nbytes, size_t elem_size, size_t block_elems, void* out, size_t cap);
size_t dio_bshuf_lz4_decode(const void* in, size_t in_len, size_t elem_size, void* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DRIFTIO_H */
