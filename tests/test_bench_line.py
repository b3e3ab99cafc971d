"""bench.py's stdout contract: the LAST line is one compact JSON object the driver can parse -- under 8 KB (its stdout tail),
one line, depth <= 2, strings <= 110 characters -- built here from a canned full result (the round-5 object, 20 KB, which the driver
could NOT parse: BENCH_r05.json "parsed": null).  SURVEY 8d."""
import copy
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def canned():
    return json.load(open(os.path.join(REPO, "profiles", "r05_bench.json")))


def check(s, n_gpus):
    assert "\n" not in s and len(s) < bench.COMPACT_LIMIT
    assert len(s) < 4500, "target: about 4 KB"
    d = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["unit"] == "Mreads/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    for k, v in d.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                assert not isinstance(v2, (dict, list)), (k, k2)          # nothing below depth 2
                assert not isinstance(v2, str) or len(v2) <= 110, (k, k2)
        else:
            assert not isinstance(v, list), k
    assert len(d["config"]["workload"]) <= 110
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel_ms", "kernel_trace_ms",
              "valu_issue_frac_of_step", "whole_step_frac"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    return d


def test_compact_line_single_gpu():
    full = canned()
    assert len(json.dumps(full)) > 15000          # the object that did not parse
    d = check(bench.compact_line(full), 1)
    cb = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample", "single_core")) <= set(cb) and isinstance(cb["single_core"], float)
    for k in ("memo_mreads", "sub1_nomemo_mreads", "mixed99_mreads", "mixed90_mreads", "host_fed_mreads", "cli_e2e_mreads", "cli_stream_mreads",
              "align_kernel_ms", "sig_kernel_ms", "list_pass_ms", "valu_wave_insts_per_batch"):
        assert k in d["roofline"], k
    # the value survives the rounding to 6 digits
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5


def test_compact_line_multi_gpu_carries_the_ranks_flat():
    full = canned()
    full["n_gpus"] = 8
    for k in ("cpu_baseline", "host_fed", "cli_e2e", "mixed", "memo_tier", "thresholds", "kernel_path", "robustness"):
        full.pop(k, None)
    full["per_rank"] = {"ms_per_step": [4.9 + 0.01 * i for i in range(8)], "allreduce_ms": [0.2 + 0.01 * i for i in range(8)]}
    full["roofline"].update(allreduce_ms_max=0.27, rank_ms_per_step_max=4.97, rank_ms_per_step_min=4.9)
    d = check(bench.compact_line(full), 8)
    rf = d["roofline"]
    assert rf["rank_ms_per_step_min"] == 4.9 and rf["rank_ms_per_step_max"] == 4.97 and rf["allreduce_ms_max"] == 0.27
    assert rf["rank7_ms_per_step"] == pytest.approx(4.97) and rf["rank0_allreduce_ms"] == pytest.approx(0.2)
    assert "cpu_baseline" not in d


def test_a_failed_leg_is_named_and_the_line_still_fits():
    full = canned()
    full["host_fed"] = {"error": "RuntimeError('x' * 500)" + "y" * 500}
    d = check(bench.compact_line(full), 1)
    assert len(d["leg_errors"]["host_fed"]) <= 110


def test_oversized_roofline_is_trimmed_not_lost():
    full = canned()
    for i in range(400):
        full["roofline"]["extra_scalar_number_%03d" % i] = 1.0 / (i + 3)
    s = bench.compact_line(full)
    assert len(s) < bench.COMPACT_LIMIT
    d = json.loads(s)
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]


def test_emit_prints_the_compact_line_last_and_keeps_the_full_object(tmp_path, monkeypatch):
    full = canned()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(copy.deepcopy(full))
    lines = out.getvalue().strip().split("\n")
    assert len(lines) == 1
    check(lines[-1], 1)
    for f in (tmp_path / "bench_full.json", tmp_path / "gpurun_out" / "bench_full.json"):
        assert json.load(open(f))["roofline"]["kernels"]          # the per-kernel blocks live in the full object
    assert "kernels" in err.getvalue()
