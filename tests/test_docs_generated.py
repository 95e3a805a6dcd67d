"""README.md / DESIGN.md quote measured numbers only through tools/gen_docs.py (typed numbers drift).  The block
between the GENERATED markers of both documents must equal a fresh generation from the committed profiles, and every entry of
profiles/r06_INDEX.json must resolve to the value it states."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_docs  # noqa: E402


def test_documents_and_index_match_the_committed_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_docs.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_every_index_entry_resolves_to_its_file_and_key():
    with open(os.path.join(ROOT, "profiles", f"{gen_docs.TAG}_INDEX.json")) as f:
        idx = json.load(f)["entries"]
    assert len(idx) > 60 and len({e["id"] for e in idx}) == len(idx)
    for e in idx:
        with open(os.path.join(ROOT, e["file"])) as f:
            doc = json.load(f)
        assert gen_docs.dig(doc, e["key"]) == e["value"], e["id"]


def test_the_committed_bench_line_is_the_compact_form_of_the_committed_record():
    """profiles/r06_bench_default.line is what bench.py printed; it must be strict JSON under 4 KB and agree with the full record"""
    import bench_common as B
    line = open(os.path.join(ROOT, "profiles", f"{gen_docs.TAG}_bench_default.line")).read().strip().splitlines()[-1]
    assert len(line.encode()) <= B.LINE_MAX
    d = json.loads(line)
    with open(os.path.join(ROOT, "profiles", f"{gen_docs.TAG}_bench_default.json")) as f:
        full = json.load(f)
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["steps"] == full["steps"]
    assert d["roofline"]["traffic_how"] == "live_pmc" and d["cpu_baseline"]["kind"] == "port"
    assert d["bytes_ok"] is True and d["verified_ok"] is True and "span_warning" not in d
