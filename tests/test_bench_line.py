"""The ONE line bench.py prints for the driver: short enough to be parsed (round 5's 22 KB line was not), valid JSON, carrying the
contract's keys, `roofline` and `cpu_baseline`.  Fed with the full record of a real run (profiles/r05, 22 KB) and with a record whose
free-text fields have grown out of hand.  The timing contract it mirrors: one number per run,
[REF roman/align/submap_align.py:155-157]."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)

FULL = os.path.join(ROOT, "profiles", "r05", "bench_line_driver_flags_20steps.json")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "timed_region_s", "p50_latency_ms")


def _full():
    with open(FULL) as fh:
        return json.load(fh)


def test_headline_of_a_real_record_is_short_valid_and_complete():
    import bench
    full = _full()
    assert len(json.dumps(full)) > 20000                              # the record that the driver could not parse
    text = bench.headline(full)
    assert len(text) < bench.HEADLINE_MAX_BYTES == 4096 and "\n" not in text
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["config"]["workload"].startswith("config 3") and "model" not in line["config"]
    r = line["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert {"k_count", "k_cos"} <= set(r["kernels"]) and "frac" in r["step"] and "frac" in r["isolated"]
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "identical_to_gpu"):
        assert k in c, k
    assert c["kind"] == "port" and line["vs_baseline"] is None and line["higher_is_better"] is True


def test_headline_stays_short_whatever_the_notes_grow_to():
    import bench
    full = _full()
    full["config"]["workload"] = "w" * 3000
    full["cpu_baseline"]["sample"] = "s" * 3000
    full["p50_note"] = "n" * 5000
    full["side_legs_error"] = "e" * 9000
    full["ranks"] = {"ms_per_step_by_rank": [1.234567] * 64}
    text = bench.headline(full)
    assert len(text) < bench.HEADLINE_MAX_BYTES
    line = json.loads(text)
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and "roofline" in line and "cpu_baseline" in line


def test_emit_prints_the_headline_last_and_alone(tmp_path, capsys, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_full())
    out = capsys.readouterr().out
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096                   # stdout: the one line; everything else went to the side file / stderr
    line = json.loads(lines[0])
    extras = json.load(open(tmp_path / "bench_extras.json"))
    assert line["extras"] == "bench_extras.json" and "caller" in extras and "decision_sensitivity" in extras


def test_live_traffic_reports_why_when_there_is_no_profiler(monkeypatch):
    """roofline.traffic is measured by re-running the main measurement under rocprofv3 --pmc (bench.live_traffic); without the tool the
    bench says so and quotes the committed pass instead — it never loses the line over it."""
    import shutil
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    tb, why = bench.live_traffic()
    assert tb is None and "rocprofv3" in why
