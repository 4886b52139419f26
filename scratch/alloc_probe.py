import sys, time, torch
sys.path.insert(0, "/root/repo")
from driftscan_amd import device
t0 = time.perf_counter(); ctx = device.get_context(workspace_bytes=100 << 30); torch.cuda.synchronize(); print("arena 100 GB: %.2f s" % (time.perf_counter() - t0), flush=True)
def alloc(gb, tag):
    t0 = time.perf_counter(); t = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    t.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-28s %5.1f GB: empty %.3f s, zero_ %.3f s" % (tag, gb, t1 - t0, t2 - t1), flush=True)
    return t
a = alloc(53, "beam_m (fresh)")
b = alloc(12, "beam_svd (fresh)")
c = alloc(12, "invbeam (fresh)")
d = alloc(5, "beam_ut (fresh)")
del b, c, d
b = alloc(11, "beam_svd (cached 12)")
c = alloc(11, "invbeam (cached 12)")
e = alloc(6, "S (fresh)"); f = alloc(6, "N (fresh)"); g = alloc(6, "E (fresh)")
import threading
def bg():
    t0 = time.perf_counter(); x = torch.empty(20 << 30, dtype=torch.uint8, device="cuda"); print("  [thread] empty 20 GB %.3f s" % (time.perf_counter() - t0), flush=True); del x
th = threading.Thread(target=bg); 
# a long kernel sequence on the main thread meanwhile
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
torch.cuda.synchronize(); t0 = time.perf_counter(); th.start()
for _ in range(20): y = x @ x
torch.cuda.synchronize(); print("20 dgemm with a concurrent 20 GB allocation: %.3f s" % (time.perf_counter() - t0)); th.join()
t0 = time.perf_counter()
for _ in range(20): y = x @ x
torch.cuda.synchronize(); print("20 dgemm alone: %.3f s" % (time.perf_counter() - t0))
h = alloc(20, "20 GB (cached by thread)")
