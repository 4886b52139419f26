"""CPU-only host logic: config properties, file patterns, storage shim, sky model shapes,
m-partitioning (serial and 2-rank gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from driftscan_amd import config, parallel, skymodel, storage, util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_patterns():
    assert util.natpattern(94) % 7 == "07"
    assert util.natpattern(512) % 3 == "003"
    assert util.natpattern(9) % 3 == "3"
    assert util.intpattern(94) % 7 == "+07"


def test_cache_last():
    calls = []

    @util.cache_last
    def f(a, b=1):
        calls.append((a, b))
        return [a, b]

    r1 = f(1)
    assert f(1) is r1 and len(calls) == 1
    f(2)
    assert len(calls) == 2


def test_config_reader():
    class A(config.Reader):
        x = config.Property(proptype=float, default=2.0)
        flag = config.Property(proptype=config.truthy, default=False, key="my_flag")
        mode = config.enum(["a", "b"], default="a")
        lst = config.list_type(type_=int, default=[])

    a = A.from_config(dict(x="3.5", my_flag="Yes", mode="b", lst=[1, "2"]))
    assert a.x == 3.5 and a.flag is True and a.mode == "b" and a.lst == [1, 2]
    assert A().x == 2.0
    with pytest.raises(ValueError):
        A.from_config(dict(mode="c"))


def test_storage_roundtrip(tmp_path):
    p = str(tmp_path / "x.hdf5")
    with storage.File(p, "w") as f:
        d = f.create_dataset("beam_m", shape=(2, 3), dtype=np.complex128)
        d[0] = [1, 2j, 3]
        f.create_dataset("sv", data=np.arange(4.0))
        f.attrs["m"] = 5
        f.attrs["FLAGS"] = "Normal"
    assert storage.can_open(p)
    with storage.File(p, "r") as f:
        assert "beam_m" in f and f["beam_m"].shape == (2, 3)
        assert np.allclose(f["beam_m"][0], [1, 2j, 3]) and np.allclose(f["sv"][1:3], [1, 2])
        assert int(f.attrs["m"]) == 5 and str(f.attrs["FLAGS"]) == "Normal"
    assert not storage.can_open(str(tmp_path / "missing.hdf5"))


def test_skymodel_shapes():
    nu = np.linspace(400, 450, 5)
    s = skymodel.im21cm_model(10, nu, 4)
    f = skymodel.foreground_model(10, nu, 4)
    assert s.shape == f.shape == (4, 4, 11, 5, 5)
    assert np.abs(s[1:]).max() == 0 and np.abs(f[0, 0]).min() > 0 and np.abs(f[3, 3]).max() == 0
    assert np.allclose(f, f.transpose(0, 1, 2, 4, 3))
    for l in (0, 5, 10):  # covariance in frequency must be positive semi-definite
        assert np.linalg.eigvalsh(f[0, 0, l]).min() > -1e-12 * np.abs(f[0, 0, l]).max()


def test_partition_serial():
    items = list(range(10))
    assert parallel.partition(items) == items
    assert parallel.partition(items, costs=[1.0] * 10) == items


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from driftscan_amd import parallel
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
ms = list(range(11))
costs = [float(12 - m) for m in ms]
mine = parallel.partition(ms, costs)
plain = parallel.partition(ms)
parts = parallel.gather_objects((parallel.rank(), mine, plain))
tot = parallel.allreduce_sum(np.array([float(sum(mine)), 1.0]))
word = parallel.bcast_object("hello" if parallel.rank0() else None)
parallel.barrier()
if parallel.rank0():
    print(json.dumps(dict(parts=parts, tot=tot.tolist(), word=word, coll=parallel.collective_stats())))
dist.destroy_process_group()
'''


def test_two_rank_gloo(tmp_path):
    """world_size = 2 on CPU: m-blocks are sharded without overlap, spectra gather to rank 0,
    the Fisher-style all-reduce sums over ranks."""
    import json
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    by_rank = {r: (mine, plain) for r, mine, plain in res["parts"]}
    assert sorted(by_rank[0][0] + by_rank[1][0]) == list(range(11))      # LPT: disjoint cover
    assert by_rank[0][1] == list(range(6)) and by_rank[1][1] == list(range(6, 11))  # the reference's contiguous split
    load = [sum(12 - m for m in by_rank[r][0]) for r in (0, 1)]
    assert abs(load[0] - load[1]) <= 12                                     # balanced
    assert res["tot"] == [float(sum(range(11))), 2.0] and res["word"] == "hello"
    # the seconds a rank spends inside collectives are counted (bench.py's N-rank job reports them per rank)
    assert res["coll"]["calls"] >= 4 and res["coll"]["allreduce_calls"] == 1 and res["coll"]["seconds"] > 0.0


def test_storage_background_writers_and_lazy_reads(tmp_path, monkeypatch):
    """Product files queued on the writer pool are complete after flush(); the npz mirror loads datasets only
    when they are touched; a failing write surfaces in flush()."""
    from driftscan_amd import storage

    if storage.HAVE_H5PY:
        pytest.skip("the npz mirror is not in use")
    rng = np.random.default_rng(0)
    data = {i: rng.standard_normal((50, 7)) + 1j * rng.standard_normal((50, 7)) for i in range(12)}

    def write(i):
        with storage.File(str(tmp_path / ("f%d.hdf5" % i)), "w") as f:
            f.create_dataset("big", data=data[i])
            f.create_dataset("small", data=np.arange(3) + i)
            f.attrs["m"] = i

    for threads in ("4", "0"):
        monkeypatch.setenv("DRIFTMI_IO_THREADS", threads)
        for i in data:
            storage.submit(write, i)
        storage.flush()
        for i in data:
            with storage.File(str(tmp_path / ("f%d.hdf5" % i)), "r") as f:
                assert f.attrs["m"] == i
                assert f._data["big"] is None            # not read yet
                assert (f["small"][:] == np.arange(3) + i).all()
                assert f._data["big"] is None            # still not
                assert np.array_equal(f["big"][:], data[i])
                assert f["big"].shape == (50, 7)
    # read-modify-write keeps the untouched datasets
    with storage.File(str(tmp_path / "f0.hdf5"), "r+") as f:
        f["small"][0] = 99
    with storage.File(str(tmp_path / "f0.hdf5"), "r") as f:
        assert f["small"][0] == 99 and np.array_equal(f["big"][:], data[0])

    def boom():
        raise IOError("disk full")

    monkeypatch.setenv("DRIFTMI_IO_THREADS", "2")
    storage.submit(boom)
    with pytest.raises(IOError):
        storage.flush()
    storage.flush()  # the queue is empty again


def test_partition_contiguous_balances_cost():
    """Contiguous m-ranges of equal estimated cost (bench.py --mode sharded, BeamTransfer generation ranges)."""
    from driftscan_amd import parallel

    items = list(range(129))
    costs = [130.0 - m for m in items]
    for n in (1, 2, 3, 8):
        parts = [parallel.partition_contiguous(items, costs, n=n, r=r) for r in range(n)]
        assert sum(parts, []) == items                      # a partition, in order
        assert all(p == list(range(p[0], p[-1] + 1)) for p in parts)
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) <= 1.15 * sum(costs) / n
    # more ranks than items: everybody gets at most one, nothing is lost
    parts = [parallel.partition_contiguous([7, 8, 9], [1, 1, 1], n=5, r=r) for r in range(5)]
    assert sorted(sum(parts, [])) == [7, 8, 9] and max(len(p) for p in parts) == 1


def test_virtual_rank_share():
    """`parallel.set_virtual(r, n)`: a single process sees the partitions of rank r of n (what `bench.py --share r/n`
    uses to time one GPU's share of the north-star job through the product classes); collectives keep to this rank."""
    from driftscan_amd import parallel

    items = list(range(129))
    costs = [130.0 - m for m in items]
    try:
        for n in (2, 8):
            parts = []
            for r in range(n):
                parallel.set_virtual(r, n)
                assert (parallel.rank(), parallel.size(), parallel.rank0()) == (r, n, r == 0)
                parts.append(parallel.partition_contiguous(items, costs))
                assert parts[-1] == parallel.partition_contiguous(items, costs, n=n, r=r)
                assert parallel.gather_objects("x") == ["x"]
                assert np.array_equal(parallel.allreduce_sum(np.arange(3.0)), np.arange(3.0))
            assert sum(parts, []) == items
    finally:
        parallel.set_virtual(None)
    assert (parallel.rank(), parallel.size()) == (0, 1)


def _exchange_worker(rank, world, port, q):
    import os

    import numpy as np
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from driftscan_amd import parallel

    # a frequency-partitioned array regrouped by m and back (the two transposes of the timestream simulation)
    nfreq, nm = 5, 7
    f_of = [parallel.partition_for(range(nfreq), r, world) for r in range(world)]
    m_of = [parallel.partition_for(range(nm), r, world) for r in range(world)]
    full = np.arange(nfreq * nm, dtype=np.float64).reshape(nfreq, nm)
    loc = full[f_of[rank]]
    got = parallel.exchange([np.ascontiguousarray(loc[:, m_of[d]]) for d in range(world)])
    by_m = np.zeros((nfreq, len(m_of[rank])))
    for src, part in enumerate(got):
        by_m[f_of[src]] = part
    ok1 = np.array_equal(by_m, full[:, m_of[rank]])
    back = parallel.exchange([np.ascontiguousarray(by_m[f_of[d]]) for d in range(world)])
    again = np.zeros((len(f_of[rank]), nm))
    for src, part in enumerate(back):
        again[:, m_of[src]] = part
    q.put((rank, bool(ok1), bool(np.array_equal(again, loc)), parallel.partition(list(range(nm))) == m_of[rank]))
    dist.destroy_process_group()


def test_exchange_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True, True, True), (1, True, True, True)]


def test_exchange_single_process():
    from driftscan_amd import parallel

    assert parallel.exchange([("a", 1)]) == [("a", 1)]
    assert parallel.partition_for(range(7), 1, 3) == [3, 4] and parallel.partition_for(range(2), 2, 3) == []


def _fisher_worker(rank, world, port, q, nm, resident):
    """One rank of a Fisher assembly with a stand-in per-m estimator (the per-m matrices need the GPU; the partition logic
    and the all-reduce do not)."""
    import os

    import numpy as np
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DRIFTMI_STORAGE="discard")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from driftscan_amd import parallel, psestimation

    class _BT:
        def _my_ms(self):
            return parallel.partition_contiguous(list(range(nm)), [float(nm - m) for m in range(nm)])

        def ndof(self, mi):
            return 4

    class _KL:
        beamtransfer = _BT()
        evdir = "/nonexistent"

        class telescope:
            mmax = nm - 1

    class Fake(psestimation.PSEstimation):
        nbands = 2

        def genbands(self):
            self.clarray = np.zeros((2, 1, 1, 1))

        def fisher_bias_batch(self, ms):
            return [(np.full((2, 2), 10.0 ** mi) + 0j, np.full(2, 10.0 ** mi) + 0j) for mi in ms]

    ps = Fake(_KL())
    if resident:                        # what ProductManager.generate does around BeamTransfer.generate
        ps.accumulate_ms([])
        mine = _BT()._my_ms()
        if mine:
            ps.accumulate_ms(mine)
    ps.generate()
    q.put((rank, float(ps.fisher[0, 0]), float(ps.bias[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("resident", [True, False])
def test_fisher_assembly_with_more_ranks_than_m_blocks(resident):
    """world_size 4 over 3 m-blocks (gloo): `partition_contiguous` leaves a rank without a block; every m must enter the
    all-reduced Fisher matrix exactly once, whether the rank accumulated during the resident pipeline or not."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + (7 if resident else 0)
    nm, world = 3, 4
    ps = [ctx.Process(target=_fisher_worker, args=(r, world, port, q, nm, resident)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    want = sum(10.0 ** m for m in range(nm))    # 111: a block counted twice or dropped shows in its own digit
    assert all(abs(f - want) < 1e-9 and abs(b - want) < 1e-9 for _, f, b in res), res


def test_m_ranges_account_for_the_sht_coupling(tmp_path):
    """`BeamTransfer._my_ms`: contiguous cost-balanced ranges; with healpy's refinement on, the ranks whose range starts among
    the m the polar rings couple (m <= mcut, `dm_bt_alias_info` — host arithmetic, no GPU) carry the BT-gen of all of those m
    and get fewer blocks; every rank computes the same partition."""
    from driftscan_amd import beamtransfer, cylinder, parallel

    cfg = dict(num_freq=4, freq_start=400.0, freq_end=420.0, freq_mode="edge", num_cylinders=2, cylinder_width=8.0,
               num_feeds=4, feed_spacing=0.5, tsys=1.0, force_lmax=200, force_mmax=200)
    parts = {}
    try:
        for it in (0, 3):
            tel = cylinder.PolarisedCylinderTelescope.from_config(dict(cfg, sht_iter=it))
            bt = beamtransfer.BeamTransfer(str(tmp_path / ("bt%d" % it)), telescope=tel)
            mcut = bt._sht_mcut()
            assert (mcut == -1) if it == 0 else (0 < mcut < tel.mmax)
            got = []
            for r in range(6):
                parallel.set_virtual(r, 6)
                got.append(bt._my_ms())
            assert sum(got, []) == list(range(tel.mmax + 1))
            assert all(p == list(range(p[0], p[-1] + 1)) for p in got)
            parts[it] = got
    finally:
        parallel.set_virtual(None)
    # the ranks that start among the coupled m give blocks away (none of them gains any), the last rank takes more
    assert all(len(a) <= len(b) for a, b in zip(parts[3][:3], parts[0][:3]))
    assert sum(len(p) for p in parts[3][:3]) < sum(len(p) for p in parts[0][:3]) and len(parts[3][-1]) > len(parts[0][-1])


def test_north_star_partitions_match_the_committed_share_records(tmp_path):
    """The m-ranges of the eight ranks of the configs[2] and configs[3] jobs, as `ProductManager` partitions them (cost
    model of `BeamTransfer._m_cost` + the downstream weights of `ProductManager`: 0.19 KLTransform, 0.57 DoubleKL, 0.14
    per Fisher estimator), are the ranges the committed all-shares records were measured on — a change of the model
    without a new record would leave `north_star.projected_job_s` describing another partition."""
    import glob
    import json
    import os

    import yaml

    import bench
    from driftscan_amd import manager, parallel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        for wl, pat, w in (("configs2", "*_configs2_shares.json", 0.19), ("configs3", "*_configs3_shares.json", 0.90)):
            recs = sorted(glob.glob(os.path.join(root, "profiles", pat)))
            assert recs, pat
            rec = json.load(open(recs[-1]))
            conf = bench.job_conf(wl)
            conf["config"]["output_directory"] = str(tmp_path / wl)
            cfile = str(tmp_path / (wl + ".yaml"))
            with open(cfile, "w") as fh:
                yaml.dump(conf, fh)
            pm = manager.ProductManager.from_config(cfile)
            assert abs(pm.beamtransfer.kl_cost_weight - w) < 1e-12
            for sh in rec["shares"]:
                r, n = (int(x) for x in sh["share"].split("/"))
                parallel.set_virtual(r, n)
                mine = pm.beamtransfer._my_ms()
                assert [mine[0], mine[-1]] == list(sh["m_range"]), (wl, sh["share"], mine[0], mine[-1], sh["m_range"])
    finally:
        parallel.set_virtual(None)


def test_rebalance_contiguous_levels_measured_times():
    """`parallel.rebalance_contiguous` (the measured load balancing of `bench.py --mode sharded`): with a cost model the static
    partition does not know — a floor per rank plus a steeply falling cost per item, like the lock-step chains of the small
    configs[1] matrices — a few rounds bring the slowest rank close to the mean; every rank computes the same ranges."""
    from driftscan_amd import parallel

    n, nr = 129, 8
    true = np.array([(1.0 - m / 140.0) ** 3 for m in range(n)])

    def measure(rg):
        return [0.3 * true[a] + float(true[a : b + 1].sum()) for a, b in rg]    # chain floor ~ largest item + the work

    static = [parallel.partition_contiguous(list(range(n)), [0.15 + 0.6 * (1 - m / n) + 0.25 * (1 - m / n) ** 3 for m in range(n)],
                                            n=nr, r=r) for r in range(nr)]
    rg = [(p[0], p[-1]) for p in static]
    first = max(measure(rg)) / np.mean(measure(rg))
    for _ in range(4):
        rg = parallel.rebalance_contiguous(rg, measure(rg), r="all")
        assert rg[0][0] == 0 and rg[-1][1] == n - 1 and all(rg[k][1] + 1 == rg[k + 1][0] for k in range(nr - 1))
    last = max(measure(rg)) / np.mean(measure(rg))
    assert first > 1.3 and last < 1.12, (first, last)


def test_deferred_arguments_go_through_the_copy_thread(tmp_path, monkeypatch):
    """storage.submit with `Deferred` arguments (device products on their way into a file): the copy thread resolves them
    in submission order, the write sees numpy arrays, failures of either stage come out of flush(), and the inline mode
    resolves on the spot.  (The device side of `Deferred.host` is covered by tests/test_gpu_pipeline.py.)"""
    import threading
    from driftscan_amd import storage

    order, seen = [], {}

    class Stub(storage.Deferred):
        __slots__ = ("i", "fail")

        def __init__(self, i, fail=False):
            self.i, self.fail, self.nbytes = i, fail, 800
            self.ctx = self.t = self.event = None
            self.resident = False

        def host(self):
            assert threading.current_thread().name.startswith(("driftmi-copy", "MainThread"))
            if self.fail:
                raise RuntimeError("copy failed")
            order.append(self.i)
            return np.full(100, float(self.i))

    def write(i, arr, plain):
        assert isinstance(arr, np.ndarray) and plain == "x"
        seen[i] = float(arr[0])

    for threads in ("3", "0"):
        monkeypatch.setenv("DRIFTMI_IO_THREADS", threads)
        order.clear(); seen.clear()
        for i in range(20):
            storage.submit(write, i, Stub(i), "x")
        storage.flush()
        assert order == list(range(20)) and seen == {i: float(i) for i in range(20)}
    monkeypatch.setenv("DRIFTMI_IO_THREADS", "2")
    storage.submit(write, 0, Stub(0, fail=True), "x")
    with pytest.raises(RuntimeError):
        storage.flush()
    storage.flush()


def test_batches_are_cut_to_equal_sizes(tmp_path):
    """`BeamTransfer._svd_batch_lists` / `KLTransform._batches`: as many batches as the budget asks for, of (nearly) equal
    size — a short last batch would run the same lock-step launch chains for a fraction of the work."""
    from driftscan_amd import beamtransfer, cylinder, kltransform

    tel = cylinder.PolarisedCylinderTelescope.from_config(dict(num_freq=4, freq_start=400.0, freq_end=420.0, freq_mode="edge",
                                                               num_cylinders=2, cylinder_width=8.0, num_feeds=4,
                                                               feed_spacing=0.5, tsys=1.0, force_lmax=60, force_mmax=60))
    bt = beamtransfer.BeamTransfer(str(tmp_path / "bt"), telescope=tel)
    F, T, P, L, K = tel.nfreq, bt.ntel, tel.num_pol_sky, tel.lmax + 1, bt.svd_len
    per_m = F * (T * (P * L + T) * 2 + K * P * L * 2 + K * T) * 16
    bt.svd_chunk_gb = 12.5 * per_m / float(1 << 30)          # room for 12 blocks
    assert [len(b) for b in bt._svd_batch_lists(range(34))] == [12, 11, 11]
    assert [len(b) for b in bt._svd_batch_lists(range(24))] == [12, 12]
    assert [len(b) for b in bt._svd_batch_lists(range(5))] == [5]
    assert sum(bt._svd_batch_lists(range(7, 41)), []) == list(range(7, 41))
    kl = kltransform.KLTransform(bt)
    bt.ndof = lambda mi: 1000                                   # every block the same size: 16 * n^2 * 16 bytes each
    kl.kl_chunk_gb = 10.5 * 16.0 * 1000.0 ** 2 * 16.0 / float(1 << 30)
    assert [len(b) for b in kl._batches(list(range(23)))] == [8, 8, 7]
    assert sum(kl._batches(list(range(23))), []) == list(range(23))
    bt.ndof = lambda mi: 3000 if mi < 3 else 1000              # uneven blocks: the even cut must still respect the budget
    for b in kl._batches(list(range(23))):
        assert sum(16.0 * bt.ndof(mi) ** 2 * 16.0 for mi in b) <= kl.kl_chunk_gb * (1 << 30) or len(b) == 1


def test_nccl_collectives_bind_the_rank_device_before_any_context(monkeypatch):
    """ProductManager.from_config / BeamTransfer.__init__ reach barriers before a driftmi Context has set torch's current
    device: under nccl every rank would then report device 0 ('Duplicate GPU detected').  `parallel` binds the device
    `device.device_index()` names (LOCAL_RANK, DRIFTMI_DEVICE overriding) before each nccl collective."""
    import torch

    from driftscan_amd import device

    calls = []
    state = dict(cur=0)

    class FakeDist(object):
        group = type("g", (), dict(WORLD=object()))

        @staticmethod
        def get_backend():
            return "nccl"

        @staticmethod
        def barrier(device_ids=None):
            calls.append(("barrier", tuple(device_ids), state["cur"]))

        @staticmethod
        def all_reduce(t):
            calls.append(("all_reduce", state["cur"]))

    monkeypatch.setattr(parallel, "_dist", lambda: FakeDist)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["cur"])
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: state.__setitem__("cur", int(d)))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, dev=None: self)
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.delenv("DRIFTMI_DEVICE", raising=False)
    assert device.device_index() == 3
    parallel.barrier()
    assert calls[-1] == ("barrier", (3,), 3)
    state["cur"] = 0
    parallel.allreduce_sum(np.ones(4))
    assert calls[-1] == ("all_reduce", 3)
    monkeypatch.setenv("DRIFTMI_DEVICE", "1")   # several ranks on one card (tests): the override wins
    parallel.barrier()
    assert calls[-1] == ("barrier", (1,), 1)
