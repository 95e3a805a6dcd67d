"""bench.py end to end on the GPU, small: the printed line must be what the driver can read (BENCH_r05.json parsed = null).
One run of the conv f=3 trace with a tiny CPU sample: ONE line on stdout, strict JSON, at most 4096 bytes, the contract's keys, flat
`roofline` and `cpu_baseline`, bytes equal to the oracle's digests, the side file written."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_py_prints_one_compact_strict_json_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--trace", "3_32", "--steps", "3", "--warmup", "1",
           "--cpu-sample-mult", "2", "--cpu-sample-add", "16", "--no-live-pmc", "--detail-out", detail]
    env = dict(os.environ)
    for k in ("VPIN_GENS_BUDGET_GB", "VPIN_SPARK_GENS_BUDGET_GB", "VPIN_TABLE_SLOT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    line = lines[0]
    assert len(line.encode()) <= 4096

    def bad(c):
        raise ValueError(c)
    d = json.loads(line, parse_constant=bad)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "constraints/s" and d["value"] > 0
    assert abs(d["value"] - d["config"]["constraints_unpadded_per_step"] * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    for sub in ("roofline", "cpu_baseline"):
        assert all(not isinstance(v, (dict, list)) for v in d[sub].values()), sub
    assert d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0 and d["roofline"]["peak"] == 8000.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["bytes_ok"] is True and d["verified_ok"] is True and "errors" not in d
    assert d["config"]["table_slot_bytes"] == 128       # every rank owns its GPU
    assert os.path.isabs(detail) and os.path.exists(detail)
    with open(detail) as f:
        full = json.load(f)
    assert full["value"] == d["value"] and "spans_ms_last_step" in full and "reference_span" in full
