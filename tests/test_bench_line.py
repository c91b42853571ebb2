"""The line the driver parses: bench.py prints every extra block FIRST, one JSON line each, and the headline LAST and small.
Round 5's single 23 KB line came back as `parsed: null`.  The recorded sample is that very line (profiles/r05_bench.json)."""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SAMPLE = os.path.join(ROOT, "profiles", "r05_bench.json")


def sample():
    return json.load(open(SAMPLE))


def test_headline_is_small_and_complete():
    extras, head = bench.split_output(sample())
    line = json.dumps(head)
    assert len(line) < bench.HEADLINE_LIMIT == 6144
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity"):
        assert k in head, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launches_timed", "algorithmic_bytes_per_launch", "valu_fp64", "floor"):
        assert k in head["roofline"], k
    assert set(head["roofline"]["floor"]) == {"lane_ops_per_pass", "nominal_ms_per_pass_at_full_issue", "measured_ms_per_pass", "issue_frac_profiled", "record"}
    assert head["cpu_baseline"]["kind"] in ("port", "reference") and head["cpu_baseline"]["cores"] >= 1
    assert head["config"]["three_call_ops_per_s"] > 0
    assert "other_configs" not in head


def test_every_extra_line_is_bounded_and_nothing_is_lost():
    src = sample()
    extras, head = bench.split_output(src)
    assert extras
    for e in extras:
        assert len(json.dumps(e)) < bench.EXTRA_LIMIT, e["extra"]
        assert "dropped" not in json.dumps(e)
    names = {e["extra"].split(".")[0] for e in extras}
    assert {"cfg4", "cfg5", "cpp_api", "single_object_latency_us", "sizes", "roofline_detail"} <= names
    assert head["extras"] == sorted(names)
    # the numbers other rounds quote are still on some line
    text = "\n".join(json.dumps(e) for e in extras)
    assert str(src["other_configs"]["cfg4"]["value"]) in text
    assert str(src["other_configs"]["cfg5"]["value"]) in text
    assert "single_threads64_three_calls_ops_per_s" in text


def test_no_extra_run_has_the_same_headline_keys():
    src = sample()
    _, full = bench.split_output(src)
    src.pop("other_configs")
    extras, bare = bench.split_output(src)
    assert list(full) == list(bare)
    assert list(full["roofline"]) == list(bare["roofline"])
    assert {k: v for k, v in full.items() if k != "extras"} == {k: v for k, v in bare.items() if k != "extras"}


def test_emit_prints_the_headline_last_and_refuses_an_oversized_one():
    buf = io.StringIO()
    line = bench.emit(sample(), out=buf)
    lines = buf.getvalue().splitlines()
    assert lines[-1] == line and json.loads(lines[-1])["metric"].startswith("homomorphic mul+relinearize")
    assert all(json.loads(ln).get("extra") for ln in lines[:-1])
    big = sample()
    big["config"]["prose"] = "x" * 7000
    with pytest.raises(RuntimeError):
        bench.emit(big, out=io.StringIO())


def test_pack_splits_large_blocks():
    block = {"a": "x" * 5000, "b": "y" * 5000, "c": {"d": "z" * 9000, "e": 1}}
    lines = bench._pack("t", block)
    assert all(len(json.dumps(ln)) < bench.EXTRA_LIMIT for ln in lines)
    assert any(ln.get("a") for ln in lines) and any(ln.get("b") for ln in lines)
    assert any("dropped" in str(ln.get("d", "")) for ln in lines) and any(ln.get("e") == 1 for ln in lines)


def test_dry_run_last_stdout_line_parses():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < bench.HEADLINE_LIMIT
    assert json.loads(last)["dry_run"] is True
