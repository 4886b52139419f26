import cProfile, pstats, sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
from driftscan_amd import device
ctx = device.get_context(workspace_bytes=24 << 30)
with tempfile.TemporaryDirectory() as tmp:
    tel, bt, kl = bench.build_objects(tmp)
    bench.hot_path_step(tel, bt, kl, ctx)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    bench.hot_path_step(tel, bt, kl, ctx)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
