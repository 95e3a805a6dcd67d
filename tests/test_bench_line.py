"""The ONE line bench.py prints must stay machine-readable (round 5's 24 KB line left BENCH_r05.json unparsed):
strict JSON, one line, at most 4096 bytes, flat scalars inside `roofline` and `cpu_baseline`.  Builds the line from round 5's full
record (a committed profile) and from a stub with hostile content."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_common as B  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _strict(s):
    def bad(c):
        raise ValueError(c)
    return json.loads(s, parse_constant=bad)


def _check(s):
    assert "\n" not in s and len(s.encode()) <= B.LINE_MAX, len(s)
    d = _strict(s)
    for k in CONTRACT:
        assert k in d, k
    assert set(d["config"]) >= {"workload", "constraints_unpadded_per_step", "inputs"}
    for sub in ("roofline", "cpu_baseline"):
        assert sub in d
        for k, v in d[sub].items():
            assert not isinstance(v, (dict, list)), (sub, k)
            assert not isinstance(v, str) or len(v) <= 120, (sub, k)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    return d


def test_line_from_round5_full_record_is_compact_and_strict():
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000   # the record that broke the driver's parser
    d = _check(B.dumps_line(B.compact_line(full, "gpurun_out/bench_detail_n1.json")))
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4
    assert d["roofline"]["traffic_how"] == "live_pmc" and d["roofline"]["limiter_how"] == "live_pmc"
    assert d["bytes_ok"] is True and d["verified_ok"] is True
    assert d["detail"] == "gpurun_out/bench_detail_n1.json"
    assert d["reference_span_ms"] == full["reference_span"]["ms_per_trace"]


def test_hostile_record_still_gives_a_valid_line(tmp_path, capsys):
    full = {"metric": "m" * 5000, "value": float("nan"), "unit": "constraints/s", "n_gpus": 1, "steps": 3, "warmup": 1,
            "ms_per_step": float("inf"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256", "data": "synthetic",
            "config": {"workload": "w" * 900, "instances": {str(i): i for i in range(500)}, "constraints_unpadded_per_step": 5,
                       "inputs": "x" * 400, "parallelism": "p" * 4000},
            "roofline": {"kernel": "k" * 3000, "bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": float("nan"),
                         "traffic": None, "limiter_note": "n" * 9000, "secondary": [{"a": "b" * 5000}]},
            "cpu_baseline": {"value": 1.0, "unit": "constraints/s", "cores": 4, "kind": "port", "sample": "s" * 4000,
                             "full_config": {"x": "y" * 3000}},
            "reference_span": {"ms_per_trace": 1.0, "ms": {"a": 1.0}, "dead_work": {"note": "z" * 4000}},
            "strong_error": "e" * 5000, "strong_ms_per_step": 12.5, "strong": {"deep": ["x"] * 1000},
            "spans_ms_last_step": {str(i): {"a": 1.0} for i in range(400)}}
    path = str(tmp_path / "sub" / "detail.json")
    s = B.emit(full, path)
    assert capsys.readouterr().out.strip() == s
    d = _check(s)
    assert d["value"] is None and d["ms_per_step"] is None and d["roofline"]["frac"] is None
    assert len(d["strong_error"]) <= 160 and d["strong_ms_per_step"] == 12.5
    with open(path) as f:
        side = json.load(f)          # the side file holds everything, NaN cleaned to null
    assert side["value"] is None and len(side["roofline"]["limiter_note"]) == 9000


def test_over_long_line_is_refused_by_dumps_line():
    with pytest.raises(ValueError):
        B.dumps_line({"x": "y" * (B.LINE_MAX + 1)})
    with pytest.raises(ValueError):
        B.dumps_line({"x": math.nan})


def test_host_cpu_helpers_return_well_formed_records():
    """bench.py's CFS-throttling readout and the per-thread CPU list (round 6: what found the W = 8 rehearsal's stalls): present or
    empty, never raising, whatever cgroup layout the box has"""
    t = B.host_cpu_throttle()
    assert isinstance(t, dict)
    if t:
        assert set(t) >= {"nr_throttled", "throttled_ms", "quota_cpus"} and t["throttled_ms"] >= 0.0
        assert t["quota_cpus"] is None or t["quota_cpus"] > 0
    th = B.thread_cpu_seconds(4)
    assert isinstance(th, list) and len(th) <= 4 and all(len(x) == 3 and x[2] >= 0.0 for x in th)


def test_compact_line_carries_host_throttling_and_errors():
    full = {"metric": "m", "value": 1.0, "unit": "constraints/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u256", "data": "synthetic",
            "config": {"workload": "w", "constraints_unpadded_per_step": 1, "inputs": "resident in HBM"},
            "roofline": {"kernel": "k", "bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
            "cpu_baseline": {"value": 1.0, "unit": "constraints/s", "cores": 1, "kind": "port", "sample": "s"},
            "host_cpu": {"quota_cpus": 16.0, "throttled_ms_in_timed_region": 12.5},
            "errors": {"reference_span": "VpinError('x')", "parity": "proof bytes differ"}}
    d = _check(B.dumps_line(B.compact_line(full, None)))
    assert d["host_quota_cpus"] == 16.0 and d["host_throttled_ms"] == 12.5
    assert "parity" in d["errors"] and "reference_span" in d["errors"]
