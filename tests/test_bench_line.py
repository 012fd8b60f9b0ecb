"""The bench output contract: stdout = ONE compact JSON line (< 4 KB, strict JSON) carrying metric / value /
ms_per_step / config / roofline / cpu_baseline / secondary_ms; the full record goes to bench_detail.json.  CPU-only: the full
record of a real run (profiles/r4/final2/bench_default.json, 22 KB — the line the round-4 driver could not parse) is pushed
through bench_common.compact_line / emit."""
import io
import json
import os
import sys
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_common as C  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "secondary_ms")
ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "kernel_ms_per_step")


def _record():
    with open(os.path.join(ROOT, "profiles", "r4", "final2", "bench_default.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _strict(s):
    def no_const(x):
        raise ValueError("non-finite constant " + x)
    return json.loads(s, parse_constant=no_const)


def test_compact_line_of_a_real_record():
    full = _record()
    assert len(json.dumps(full)) > 20000                       # the record that did not parse
    s = json.dumps(C.compact_line(full), allow_nan=False)
    assert len(s) < C.COMPACT_LIMIT and "\n" not in s
    c = _strict(s)
    for k in REQUIRED:
        assert k in c, k
    for k in ROOF:
        assert k in c["roofline"], k
    assert c["value"] == C._num(full["value"]) and c["ms_per_step"] == C._num(full["ms_per_step"])
    assert c["roofline"]["frac"] == C._num(full["roofline"]["frac"])
    assert c["roofline"]["traffic"] == C._num(full["roofline"]["traffic"])
    assert c["config"]["workload"] == full["config"]["workload"] and c["config"]["arithmetic_kind"] == "fp16x2-pairs"
    assert set(c["secondary_ms"]) == set(full["secondary"])
    assert all(isinstance(v, float) for v in c["secondary_ms"].values())
    for k in ("value", "unit", "kind", "cores"):
        assert k in c["cpu_baseline"]
    # nothing nested deeper than one level below config / roofline / cpu_baseline / secondary_ms
    for k in ("config", "roofline", "cpu_baseline", "secondary_ms"):
        assert all(not isinstance(v, dict) for v in c[k].values()), k


def test_emit_prints_one_line_and_writes_the_detail(tmp_path, monkeypatch):
    full = _record()
    full["config"]["a_paragraph"] = "x" * 5000                   # notes of any length stay out of the line
    full["roofline"]["kernels"] = [{"kernel": "k" * 300, "ms": 1.0}] * 50
    for i in range(12):                                          # more secondary blocks than today: the line still fits
        full["secondary"][f"extra_{i}"] = {"ms_per_step": 1.0 + i, "roofline": {"kernel": "z" * 200, "frac": 0.5}}
    monkeypatch.setattr(C, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        C.emit(full)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < C.COMPACT_LIMIT
    c = _strict(lines[0])
    assert c["detail_file"] == "bench_detail.json" and "a_paragraph" not in c["config"]
    assert len(c["secondary_ms"]) == len(full["secondary"])
    detail = json.loads((tmp_path / "bench_detail.json").read_text())
    assert detail["config"]["a_paragraph"] == "x" * 5000 and len(detail["roofline"]["kernels"]) == 50
    assert err.getvalue().startswith("BENCH_DETAIL {")


def test_non_finite_numbers_never_reach_the_line():
    full = _record()
    full["config"]["rtf"] = float("nan")
    full["config"]["weight_bcast_ms"] = float("inf")
    c = C.compact_line(full)
    assert "rtf" not in c["config"] and "weight_bcast_ms" not in c["config"]
    json.dumps(c, allow_nan=False)
