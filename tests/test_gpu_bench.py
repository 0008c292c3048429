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
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["dtype"].startswith("f32") and d["vs_baseline"] is None
    par = d["parity"]
    assert par["ok"] and par["match_indices_identical"] and par["timed_record_identical"] and d["config"]["matches_last_step"] > 0
    assert par["queries_checked"] == [0, 1] and len(par["per_query"]) == 2 and par["keypoint_set_identical"]      # the gate reads the timed batch itself
    assert 0.9 <= par["keypoint_order_frac_min"] <= 1.0 and par["logits_maxdiff_max"] < 1e-3
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 0.05 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches_per_step"] == 33


def test_bench_alt_matcher_pattern_and_upload_in_step(hip_lib):
    """Two of the `alt` configurations on a small batch: the matcher's real call pattern (seg_k ragged pairs per query in one grouped
    call: every reported match is the planted twin) and the frame upload inside the step (uint8 frames from pinned host memory +
    pram_stage_frames_u8)."""
    env = dict(os.environ, PRAM_BENCH_ALTS="matcher_segk5,h2d_in_step")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2", "--cpu-queries", "0",
                        "--no-parity", "--alt", "on"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    alt = _last_json(r.stdout)["alt"]
    assert set(alt) == {"matcher_segk5", "h2d_in_step"} and not any("error" in v for v in alt.values()), alt
    sk = alt["matcher_segk5"]
    assert sk["pairs_per_step"] == 10 and sk["query_side"][2] <= 512 and 600 <= sk["reference_side"][0] <= sk["reference_side"][2] <= 1376
    assert sk["matches"] > 500 and sk["matches_correct"] >= 0.99 * sk["matches"]
    assert alt["h2d_in_step"]["queries_per_s"] > 0


def test_bench_alt_sustained_and_activation_scale(hip_lib):
    """Round 6's `alt` entries on a small batch: `sustained` (a run of seconds over rotating DISTINCT batches with a power / clock
    trace and its ratio to the headline) and `act_scale_1` (every model's planes at scale 1 — where the range guard leaves a hot
    checkpoint — with the fp32 parity gate green on it)."""
    env = dict(os.environ, PRAM_BENCH_ALTS="sustained,act_scale_1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--batch-per-gpu", "2", "--cpu-queries", "0",
                        "--alt", "on"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    alt = d["alt"]
    assert set(alt) == {"sustained", "act_scale_1"} and not any("error" in v for v in alt.values()), alt
    su = alt["sustained"]
    assert su["seconds"] >= 9.0 and su["distinct_batches"] == 4 and 0.8 < su["ratio_to_value"] < 2.0      # (a 4-step headline of 2 queries is not warm: the default run reads 0.999)
    assert su["power"] is None or len(su["power"]["trace"]) == 10
    a1 = alt["act_scale_1"]
    assert a1["parity"]["ok"] and a1["parity"]["match_indices_identical"] and a1["parity"]["logits_maxdiff_max"] < 1e-3
    assert d["alt_parity_failed"] == []


def test_bench_latency_mode_line(hip_lib):
    """`--latency`: one query per step, nothing in flight, the step replayed from captured graphs on two streams, 512-key attention
    chunks (split launches); the parity gate runs on the same arithmetic."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--latency", "--steps", "8", "--warmup", "3", "--cpu-queries", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 1 and c["queries_per_step"] == 1 and c["batches_in_flight_per_gpu"] == 1 and c["hipgraph_replay"] is True
    assert c["attention_chunk_keys"] == 512 and "latency mode" in c["workload"]
    assert d["parity"]["ok"] and d["parity"]["match_indices_identical"] and d["parity"]["timed_record_identical"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.05 * d["value"] and d["ms_per_step"] < 20.0
    assert d["range_guard"]["x3_range_exceeded"] is False


def test_bench_two_ranks_control_flow(hip_lib):
    env = dict(os.environ, PRAM_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch-per-gpu", "2", "--cpu-queries", "0", "--no-parity"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    # whole-job aggregate: both ranks' queries over the max-over-ranks time
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3e-3)) < 0.05 * d["value"]


def test_bench_spawns_its_own_ranks_and_shards_unevenly(hip_lib):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (one-GPU test hook: both on GPU 0 over gloo) and
    reports n_gpus = 2; --batch-total 3 gives the ranks 2 + 1 queries (uneven shards through the padded all-gather)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PRAM_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-total", "3",
                        "--cpu-queries", "0", "--no-parity"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["queries_per_step"] == 3
    assert d["config"]["queries_per_gpu_per_step"] == [2, 1]
    assert abs(d["value"] - 3 * 2 / (d["ms_per_step"] * 2e-3)) < 0.05 * d["value"]


@pytest.mark.parametrize("total", [64, 65])
def test_bench_eight_ranks_on_one_device(hip_lib, total):
    """BASELINE configs[2]'s launch shape with the one-GPU test hook: `bench.py --gpus 8 --batch-total 64 | 65` starts eight ranks
    (all on GPU 0, gloo), shards 8 x 8 / 9 + 7 x 8 queries, and every rank finds its own records at its own query positions of the
    gathered record (config.gather_order_verified); the line carries one time per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PRAM_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch-total", str(total),
                        "--cpu-queries", "0", "--no-parity"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    want = [9] + [8] * 7 if total == 65 else 8
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["queries_per_step"] == total
    assert d["config"]["queries_per_gpu_per_step"] == want and d["config"]["gather_order_verified"] is True
    assert len(d["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in d["per_rank_ms_per_step"])
    # what explains a curve: per-rank host issue time, the all-gather's time on the main stream (the wait for the slowest rank inside), the skew
    assert len(d["per_rank_host_issue_ms_per_step"]) == 8 and len(d["per_rank_gather_ms"]) == 8 and all(g >= 0 for g in d["per_rank_gather_ms"])
    assert abs(d["rank_skew_ms_per_step"] - (max(d["per_rank_ms_per_step"]) - min(d["per_rank_ms_per_step"]))) < 2e-3
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) < 1e-3 * d["ms_per_step"] + 1e-3
    assert abs(d["value"] - total * 2 / (d["ms_per_step"] * 2e-3)) < 0.05 * d["value"]


def test_bench_one_rank_through_the_launcher_equals_the_plain_run(hip_lib):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (how the driver starts N > 1) measures what `python bench.py`
    measures: same line, value within the run-to-run spread."""
    args = ["--gpus", "1", "--steps", "10", "--warmup", "4", "--cpu-queries", "0", "--no-parity", "--alt", "off"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PRAM_BENCH_ONE_DEVICE")}
    a = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", str(ROOT / "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr + b.stderr)[-3000:]
    da, db = _last_json(a.stdout), _last_json(b.stdout)
    assert da["n_gpus"] == db["n_gpus"] == 1 and da["config"]["queries_per_step"] == db["config"]["queries_per_step"] == 16
    assert abs(da["value"] - db["value"]) < 0.03 * da["value"], (da["value"], db["value"])


def test_bench_refuses_more_gpus_than_visible(hip_lib):
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PRAM_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout) and not [l for l in r.stdout.splitlines() if l.startswith("{")]
