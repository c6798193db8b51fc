"""bench.py's output contract (CPU): the LAST stdout line is a compact headline object that the driver's 8 KB tail keeps
whole -- round 5's single 20 KB line came back as `parsed: null` -- with `roofline` and `cpu_baseline` inside; every other
line is a self-contained JSON object as well; the full record goes to gpurun_out/bench_detail.json."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = os.path.join(ROOT, "profiles", "r05", "bench_default_run.json")     # a real full record of the default run


def _emit(record, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    bench.emit(record, buf)
    return buf.getvalue().splitlines()


def _record():
    rec = json.load(open(RECORDED))
    for sec, tag in zip(rec["secondary"], ("sweep_2q", "sweep_3q", "pgdb_3q_sic", "pgdb_3q_pauli", "pgdb_1q")):
        sec["tag"] = tag                                                          # run_* set these since round 6
    # the workloads added after that run was recorded, shaped like run_mle_state's lines
    for n, (val, ms) in ((2, (5.1e6, 205.0)), (3, (7.3e5, 359.0))):
        rec["secondary"].append({
            "tag": f"mle_state_{n}q", "metric": f"state-tomography iterative-MLE reconstructions/sec ({n}-qubit, maxiter 100)",
            "value": val, "unit": "reconstructions/s", "n_gpus": 1, "steps": 10, "warmup": 2, "ms_per_step": ms, "dtype": "f64",
            "config": {"workload": "x" * 400, "batch_per_gpu": 1 << 20, "iters": 100, "mean_outer_iters": 100.0},
            "roofline": {"bound": "mfma", "achieved": 1.2345678, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.0157, "measured_frac": None,
                         "mfma_frac": 0.0, "traffic": None, "kernel": "mle_state_packed_kernel<2>", "kernel_ms": ms, "note": "y" * 500},
            "cpu_baseline": {"value": 12.5, "unit": "reconstructions/s", "cores": 1, "kind": "port", "sample": "z" * 300}})
    return rec


def test_last_line_is_the_compact_headline(tmp_path, monkeypatch):
    rec = _record()
    lines = _emit(rec, tmp_path, monkeypatch)
    last = lines[-1]
    assert len(last) < bench.HEADLINE_MAX_BYTES == 4096
    head = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in head, key
    assert head["metric"] == "process-tomography MLE reconstructions/sec (2-qubit, 100 iters)"
    assert head["value"] == pytest.approx(rec["value"], rel=1e-5) and head["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-5)
    assert "workload" in head["config"] and "model" not in head["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert key in head["roofline"], key
    assert head["roofline"]["frac"] == pytest.approx(head["roofline"]["achieved"] / head["roofline"]["peak"], rel=1e-4)
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in head["cpu_baseline"], key
    # what the review asked to keep where the driver's record keeps it
    cfg = head["config"]
    assert cfg["transfer_inclusive"]["value"] > 0 and cfg["cpu_reference_faithful_1core"]["value"] > 0 and cfg["cpu_multicore"]["value"] > 0
    fixed = [f for f in cfg["parity_vs_reference_fixtures"] if f["mode"] == "fixed100"][0]
    assert fixed["max_fidelity"] == pytest.approx(3.677e-8, rel=1e-2)            # the 3.7e-8 of the timed mode stays visible
    assert set(cfg["secondary"]) == {"sweep_2q", "sweep_3q", "pgdb_3q_sic", "pgdb_3q_pauli", "pgdb_1q", "mle_state_2q", "mle_state_3q"}


def test_every_line_parses_and_the_tail_holds_the_headline(tmp_path, monkeypatch):
    lines = _emit(_record(), tmp_path, monkeypatch)
    assert len(lines) >= 8
    for ln in lines:
        obj = json.loads(ln)
        assert isinstance(obj, dict)
        assert len(ln) < 2048 or ln is lines[-1], (len(ln), ln[:80])
    tail = "\n".join(lines)[-8192:]
    assert tail.endswith(lines[-1]) and json.loads(tail.splitlines()[-1])["roofline"]["kernel"].startswith("pgdb_kernel")
    detail = json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))
    assert "secondary" in detail and "strong_65536" in detail                    # nothing is lost: the full record is on disk


def test_headline_shrinks_rather_than_overflowing(tmp_path, monkeypatch):
    rec = _record()
    rec["config"]["workload"] = "w" * 5000
    for k in range(40):
        rec["secondary"].append(dict(rec["secondary"][-1], tag=f"extra_{k}"))
    lines = _emit(rec, tmp_path, monkeypatch)
    assert len(lines[-1]) < 4096
    head = json.loads(lines[-1])
    assert "roofline" in head and "cpu_baseline" in head and "secondary" not in head["config"]
