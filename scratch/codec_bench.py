"""single-thread throughput of the chunk codecs of dm_h5io.c on truncated beam-like data"""
import ctypes, time, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driftscan_amd import storage
lib = ctypes.CDLL(storage._DIO_PATH)
rng = np.random.default_rng(0)
n = 1 << 21
z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.exp(rng.uniform(-6, 0, n))
# truncate to ~1e-7 relative: zero the low mantissa bits
v = z.view(np.float64).copy()
iv = v.view(np.uint64); iv &= np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(30)
raw = v.view(np.uint8)
out = np.empty(raw.size + 4096, np.uint8)
lib.dio_bshuf_lz4_encode.restype = ctypes.c_size_t
lib.dio_bshuf_lz4_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
lib.dio_lzf_compress.restype = ctypes.c_size_t
lib.dio_lzf_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
lib.dio_bitshuffle_blocked.restype = ctypes.c_size_t
lib.dio_bitshuffle_blocked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
for name, fn in (("bshuf+lz4 es16", lambda: lib.dio_bshuf_lz4_encode(raw.ctypes.data, raw.size, 16, 0, out.ctypes.data, out.size)),
                 ("bshuf only es16", lambda: lib.dio_bitshuffle_blocked(raw.ctypes.data, out.ctypes.data, raw.size // 16, 16, 0, 0)),
                 ("lzf", lambda: lib.dio_lzf_compress(raw.ctypes.data, raw.size, out.ctypes.data, raw.size - 1))):
    fn(); t0 = time.perf_counter(); got = fn(); dt = time.perf_counter() - t0
    print("%-16s %.3f s  %.2f GB/s  ratio %.3f" % (name, dt, raw.size / dt / 1e9, got / raw.size))
import hashlib
got = lib.dio_bshuf_lz4_encode(raw.ctypes.data, raw.size, 16, 0, out.ctypes.data, out.size)
print("sha", hashlib.sha256(out[:got].tobytes()).hexdigest()[:16], got)
# full-precision products: noise, and noise followed by zero rows (beam_svd rows >= nmodes)
full = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).view(np.uint8)
half = full.copy(); half[half.size // 2:] = 0
for name, buf in (("lzf full-precision", full), ("lzf half zeros", half)):
    fn = lambda: lib.dio_lzf_compress(buf.ctypes.data, buf.size, out.ctypes.data, buf.size - 1)
    fn(); t0 = time.perf_counter(); got = fn(); dt = time.perf_counter() - t0
    print("%-20s %.3f s  %.2f GB/s  ratio %.3f" % (name, dt, buf.size / dt / 1e9, got / buf.size))
