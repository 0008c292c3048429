"""CPU: bench.py plumbing that does not need the GPU — CLI contract, core detection, and the refusal to run
without an MI355X (there is no CPU fallback to time by accident)."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_bench_refuses_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("this host has a GPU: bench.py runs here (tests/test_gpu_bench.py covers that)")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)


def test_bench_cli_contract_and_cores():
    sys.path.insert(0, str(ROOT))
    import bench
    assert 1 <= bench.usable_cores() <= 4096
    src = (ROOT / "bench.py").read_text()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"',
                '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key


def test_committed_bench_line_is_well_formed():
    line = json.loads((ROOT / "profiles" / "r01_bench_final.json").read_text())
    assert line["unit"] == "queries/s" and line["n_gpus"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["peak"] == 157.3
    assert abs(rf["traffic"] / rf["algorithmic_bytes_per_launch"] - 1.0) < 0.05          # no over-fetch
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


def test_committed_round2_bench_line_is_well_formed():
    line = json.loads((ROOT / "profiles" / "r02_bench_default.json").read_text())
    assert line["unit"] == "queries/s" and line["n_gpus"] == 1 and line["vs_baseline"] is None and line["scaling"] == "weak"
    assert line["config"]["precision"] == "x3" and line["dtype"].startswith("f32 results")
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and abs(rf["peak"] - 2500.0 / 3) < 0.1
    assert line["parity"]["ok"] and line["parity"]["match"]["indices_identical"] and line["config"]["matches_last_step"] > 0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and set(cb["stage_seconds"]) == {"extract", "recognise", "match"}


def test_committed_round6_bench_line_is_well_formed():
    """The round's final default line as committed: headline fields, the roofline with PMC traffic of the kernel source that ran,
    the parity gate on four timed queries, and round 6's additions (sustained run, parity blocks on the alts, C5 fractions)."""
    line = json.loads((ROOT / "profiles" / "r06_bench_default_final.json").read_text())
    assert line["unit"] == "queries/s" and line["n_gpus"] == 1 and line["vs_baseline"] is None and line["scaling"] == "weak"
    assert line["config"]["precision"] == "x3" and line["config"]["queries_per_step"] == 16
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and abs(rf["peak"] - 2500.0 / 3) < 0.1
    assert abs(line["value"] - 16 * 1e3 / line["ms_per_step"]) < 0.01 * line["value"]
    p = line["parity"]
    assert p["ok"] and len(p["queries_checked"]) == 4 and p["match_indices_identical"] and p["keypoint_set_identical"] and p["timed_record_identical"]
    alt = line["alt"]
    su = alt["sustained"]
    assert su["seconds"] >= 10.0 and su["distinct_batches"] == 4 and 0.97 <= su["ratio_to_value"] <= 1.03 and len(su["power"]["trace"]) == 10
    for name in ("adagml", "c4", "act_scale_1"):
        assert alt[name]["parity"]["ok"] and alt[name]["parity"]["match_indices_identical"], name
    c5 = alt["c5_f16"]["parity"]
    assert c5["ok"] and c5["keypoint_set_overlap_min"] >= 0.9978 and c5["match_index_agreement_min"] >= 0.9992
    assert line["alt_parity_failed"] == []
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
