"""Host-side profile of the bench step (development aid): cProfile over a few hot-path steps."""
import cProfile, pstats, sys, os, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from driftscan_amd import device
tmp = tempfile.mkdtemp()
tel, bt, kl = bench.build_objects(tmp)
ctx = device.get_context()
for _ in range(3):
    bench.hot_path_step(tel, bt, kl, ctx)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    bench.hot_path_step(tel, bt, kl, ctx)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
