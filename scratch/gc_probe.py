import sys, os, tempfile, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, torch
from driftscan_amd import device
ctx = device.get_context(workspace_bytes=24 << 30)
_t = {}
def cb(phase, info):
    if phase == "start": _t["s"] = time.perf_counter()
    else: print("   gc gen%d %.2f ms collected %d" % (info["generation"], 1e3 * (time.perf_counter() - _t["s"]), info["collected"]))
gc.callbacks.append(cb)
with tempfile.TemporaryDirectory() as tmp:
    tel, bt, kl = bench.build_objects(tmp)
    for rep in range(4):
        st = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.hot_path_step(tel, bt, kl, ctx, stage_times=st)
        torch.cuda.synchronize(); print("step %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)), [round(1e3 * x, 1) for x in st[0]])
    gc.collect(); gc.freeze()
    print("frozen")
    for rep in range(4):
        st = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.hot_path_step(tel, bt, kl, ctx, stage_times=st)
        torch.cuda.synchronize(); print("step %d: %.1f ms" % (rep, 1e3 * (time.perf_counter() - t0)), [round(1e3 * x, 1) for x in st[0]])
