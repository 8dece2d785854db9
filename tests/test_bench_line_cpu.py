"""The line the driver reads (BENCH_r04.json `parsed: null` - the one stdout line had grown to 22 KB).

`bench.compact_result` is fed the full record of a real run (profiles/r04_bench_default.json: ten configs, every
roofline / cpu_baseline / work-counter object) and the result must be ONE short JSON line carrying the contract's keys;
`bench.emit_result` must leave exactly that line on stdout and the whole record in the sidecar file."""

import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench_default.json")) as fh:
        return json.load(fh)


def test_compact_line_is_short_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the record that broke the driver's capture
    line = bench.compact_result(full, "bench_configs.json")
    assert "\n" not in line and len(line) < 4096 < 8192
    d = json.loads(line)
    for key in CONTRACT:
        assert key in d, key
    assert d["config"]["workload"].startswith("c2(iii)") and "model" not in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch"):
        assert key in d["roofline"], key
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    # the numbers are the full record's, to six figures
    assert abs(d["value"] / full["value"] - 1) < 1e-5 and abs(d["ms_per_step"] / full["ms_per_step"] - 1) < 1e-5
    # at most five scalars per extra config
    assert set(d["configs"]) == set(full["configs"])
    for name, brief in d["configs"].items():
        assert len(brief) <= 5 and all(not isinstance(v, (dict, list)) for v in brief.values())
        assert abs(brief["value"] / full["configs"][name]["value"] - 1) < 1e-5


def test_compact_line_survives_many_and_failed_configs():
    full = _canned()
    for i in range(200):  # far more entries than the bench has: the per-config scalars go, the headline stays whole
        full["configs"][f"extra{i}"] = dict(error="DeviceError: " + "x" * 500)
    d = json.loads(bench.compact_result(full))
    assert "configs" not in d and "roofline" in d and "cpu_baseline" in d


def test_emit_result_prints_one_stdout_line_and_writes_the_sidecar(tmp_path, monkeypatch):
    side = tmp_path / "bench_configs.json"
    monkeypatch.setenv("MICI_AMD_BENCH_SIDECAR", str(side))
    full = _canned()
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit_result(full)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    last = json.loads(lines[-1])
    assert "roofline" in last and "cpu_baseline" in last and last["detail"] == str(side)
    assert json.loads(side.read_text()) == full  # nothing of the detail is lost
    assert all(ln.startswith("# ") for ln in err.getvalue().splitlines())
