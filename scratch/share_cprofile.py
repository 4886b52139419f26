"""cProfile of one configs[2] share through generate() (host side: where the GPU-idle seconds go)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
share = sys.argv[1] if len(sys.argv) > 1 else "0/8"
files = len(sys.argv) > 2 and sys.argv[2] == "files"
workload = sys.argv[3] if len(sys.argv) > 3 else "configs2"
pr = cProfile.Profile()
pr.enable()
line = bench.measure_share(workload, share, files=files, truncate=files, outdir="/dev/shm" if files else None)
pr.disable()
print("share_s", line["share_s"], "kernel_s", line["kernel_s"])
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumtime").print_stats(45)
print(st.getvalue()[:9000])
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(25)
print(st.getvalue()[:6000])
