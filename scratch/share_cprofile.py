"""cProfile of one configs[2] share (host side): where the Python process spends time that is not a wait for the GPU."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
share = sys.argv[1] if len(sys.argv) > 1 else "0/8"
pr = cProfile.Profile()
pr.enable()
line = bench.measure_share(sys.argv[2] if len(sys.argv) > 2 else "configs2", share)
pr.disable()
print("share_s", line["share_s"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:9000])
