"""The driver line: bench.py's LAST stdout line must stay small enough for the driver's parser (round 5 lost its record to a
25 KB line) and carry the contract keys + roofline + cpu_baseline + parity.  The canned full record is round 5's own
(profiles/r05z_bench_default.json, 22 KB), inflated the way a later round would inflate it."""
import copy
import io
import json
import os

import pytest

from benchlib import line as benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "profiles", "r05z_bench_default.json")


@pytest.fixture()
def full():
    rec = json.load(open(CANNED))
    rec["parity"]["ev_kept_max_rel_err"] = 3.2e-9
    rec["parity"]["ev_kept_modes"] = 41234
    return rec


def _check_contract(line):
    for k in benchline.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline", "parity"):
        assert k in line, k
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert len(line["parity"]) <= 9 and all(not isinstance(v, (dict, list)) for v in line["parity"].values())


def test_compact_line_is_small_and_complete(full):
    assert len(json.dumps(full)) > 20000            # the record that broke the parser
    line = benchline.compact(full)
    s = json.dumps(line)
    assert len(s) < benchline.MAX_LINE_BYTES
    assert len(s) < 4096                            # in practice: a few KB
    back = json.loads(s)
    assert back == line
    _check_contract(back)
    assert back["value"] == pytest.approx(full["value"], rel=1e-5)
    assert back["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert "classes" not in back["roofline"] and "valu_busy" not in back["roofline"]
    ns = back["north_star"]
    assert ns["projected_job_s"] == pytest.approx(full["north_star"]["projected_job_s"], rel=1e-5)
    assert set(ns["live_share_s"]) == {k for k, v in full["north_star"]["shares"].items() if v["source"] == "live"}
    assert ns["zgemm_cov"]["frac"] == pytest.approx(full["north_star"]["zgemm_cov"]["frac"], rel=1e-5)
    assert back["parity"]["ev_kept_max_rel_err"] == pytest.approx(3.2e-9)


def test_compact_line_survives_growth(full):
    """Whatever later legs add to the full record, the last line stays under the cap."""
    big = copy.deepcopy(full)
    big["roofline"]["classes"].update({"extra_%d" % i: dict(ms_per_step=1.0, note="x" * 200) for i in range(400)})
    big["north_star"]["shares"].update({"%d/64" % i: dict(share_s=1.0 + i, source="live", stale=False) for i in range(64)})
    big["ranks"]["per_rank"] = [dict(rank=i, step_ms=10.0 + i, m_lo=i, m_hi=i) for i in range(64)]
    big["config"]["workload"] = "w" * 5000
    big["cpu_baseline"]["sample"] = "s" * 5000
    assert len(json.dumps(big)) > 100000
    line = benchline.compact(big)
    assert len(json.dumps(line)) < benchline.MAX_LINE_BYTES
    _check_contract(line)


def test_compact_line_multi_rank_job(full):
    rec = copy.deepcopy(full)
    rec["n_gpus"] = 8
    rec["ranks"]["per_rank"] = [dict(rank=i, step_ms=15.0 + i, m_lo=16 * i, m_hi=16 * i + 15) for i in range(8)]
    rec["north_star"] = dict(job=dict(job_s=21.5, value=23.9, n_gpus=8, ranks_seen_by_rccl=8, imbalance_max_over_mean=1.05,
                                      per_rank=[dict(rank=i, seconds=20.0 + 0.1 * i, collective_s=0.5) for i in range(8)]),
                             leg_wall_s=60.0, what="the REAL 8-rank job", target="...")
    line = benchline.compact(rec)
    assert len(json.dumps(line)) < benchline.MAX_LINE_BYTES
    assert line["rank_step_ms"] == [15.0 + i for i in range(8)]
    assert line["north_star"]["ranks_seen_by_rccl"] == 8 and len(line["north_star"]["rank_seconds"]) == 8
    assert line["north_star"]["job_s"] == 21.5


def test_compact_line_without_optional_legs():
    rec = dict(metric="m-blocks/sec (BT-gen + SVD + KL)", value=1000.0, unit="m-blocks/s", n_gpus=1, steps=2, warmup=1,
               ms_per_step=129.0, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="configs[1]"), roofline=None, cpu_baseline=None, parity=None)
    line = benchline.compact(rec)
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["parity"] is None
    assert json.loads(json.dumps(line)) == line


def test_emit_prints_the_compact_line_last(full, tmp_path):
    buf = io.StringIO()
    path = str(tmp_path / "bench_detail.json")
    benchline.emit(full, detail_path=path, out=buf)
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) == 2
    assert lines[0].startswith("bench_detail {") and json.loads(lines[0][len("bench_detail "):])["metric"] == full["metric"]
    assert [ln for ln in lines if ln.startswith("{")] == [lines[-1]]      # the only JSON-object line of stdout
    last = json.loads(lines[-1])
    assert len(lines[-1]) < benchline.MAX_LINE_BYTES
    _check_contract(last)
    assert last["detail"] == path
    assert json.load(open(path))["roofline"]["classes"] == full["roofline"]["classes"]
