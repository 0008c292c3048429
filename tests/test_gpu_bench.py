"""GPU: bench.py end to end — the one-line JSON contract at N = 1, and the multi-rank control flow (barriers, result
gather on the main stream, max-over-ranks timing) with two ranks sharing GPU 0 over gloo (PRAM_BENCH_ONE_DEVICE test
hook; the driver's real N > 1 runs use one GPU per rank over RCCL)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_one_gpu_line(hip_lib):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--batch-per-gpu", "2", "--cpu-queries", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 0.05 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches_per_step"] == 33


def test_bench_two_ranks_control_flow(hip_lib):
    env = dict(os.environ, PRAM_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch-per-gpu", "2", "--cpu-queries", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    # whole-job aggregate: both ranks' queries over the max-over-ranks time
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3e-3)) < 0.05 * d["value"]
