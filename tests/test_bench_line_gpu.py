"""The contract of bench.py's output: ONE JSON line on stdout (whatever libraries print -- RCCL writes a version banner
to stdout when its first communicator comes up), carrying the fields the driver reads, for the plain single-GPU run and
for the data-parallel schedule rehearsed over RCCL with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-ref-batch", *flags]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, f"stdout must hold the JSON line only, got {len(lines)} lines: {r.stdout[:400]!r}"
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("rehearse", [False, True])
def test_one_json_line_with_the_contract_fields(rehearse):
    d, err = _run(*(["--dp-rehearse"] if rehearse else []))
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "utterances/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2      # value = clips / time
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["measured"] == "in_replay"
    if rehearse:
        assert d["dp"]["backend"] == "nccl" and d["dp"]["world_size_reported_by_backend"] == 1
        assert "rehearse" in d["config"]["parallelism"]
    else:
        assert d["dp"] is None and d["config"]["parallelism"] == "dp1"
