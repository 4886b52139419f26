"""Product-file throughput of the storage layer alone (host only): configs[2]-shaped beam_m blocks written through the
writer pool, threads only (DRIFTMI_IO_PROCS=0) against writer processes."""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DRIFTMI_STORAGE", "hdf5")
from driftscan_amd import storage

def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    F, B, P, L = 64, 432, 4, 513
    rng = np.random.default_rng(0)
    base = (rng.standard_normal((F, 2, B, P, L // 4 + 1)) + 1j * rng.standard_normal((F, 2, B, P, L // 4 + 1)))
    blk = np.repeat(base, 4, axis=-1)[..., :L].copy()    # 1.8 GB, compressible like smooth beams are not: a hard case
    blk *= np.exp(-np.arange(L) / 80.0)
    print("block %.2f GB" % (blk.nbytes / 1e9), flush=True)
    for procs, threads in ((0, 8), (8, 8), (16, 16), (0, 1)):
        os.environ["DRIFTMI_IO_PROCS"], os.environ["DRIFTMI_IO_THREADS"] = str(procs), str(threads)
        storage.shutdown_writers()
        storage._pool = None
        d = tempfile.mkdtemp(dir=os.environ.get("IO_BENCH_DIR", "/tmp"))
        def write_m(mi, data):
            with storage.File(os.path.join(d, "beam_%d.hdf5" % mi), "w") as f:
                f.create_dataset("beam_m", data=data[..., mi:], **storage.compression_kwargs((1, 2, 10, P, L - mi)))
                f.attrs["m"] = mi
        t0 = time.time()
        for mi in range(nblk):
            storage.submit(write_m, mi, blk)
        storage.flush()
        dt = time.time() - t0
        size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print("procs %2d threads %2d: %d blocks in %.1f s = %.2f GB/s of products (%.1f GB on disk)" % (procs, threads, nblk, dt, nblk * blk.nbytes / 1e9 / dt, size / 1e9), flush=True)
        shutil.rmtree(d)
    storage.shutdown_writers()

if __name__ == "__main__":
    main()
