#!/usr/bin/env python
"""Benchmark of the PRAM per-query hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 16 synthetic query frames per GPU
(SFD2 extract + sample -> SegNetViT nc113 -> GML match with 20 Sinkhorn iterations against a
2048-keypoint reference set), inputs resident in HBM.  Queries shard across ranks with no data-path
collective; each step ends with the single all-gather of the fixed-size result records.
Prints ONE JSON line on rank 0 (contract in the task statement): metric / value (whole-job
queries/s) / roofline (attention kernel, f32 MFMA) / cpu_baseline (oracle on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
H, W_IMG = 480, 640


def build_models(dev, matcher_name, n_class=113):
    from pram_amd import weights as W
    from pram_amd.nets.adagml import AdaGML
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    from pram_amd.nets.sfd2 import ResNet4x
    sfd2, seg = ResNet4x(), load_segnet('segnetvit', n_class, 256, 15, 1024)
    matcher = GML({}) if matcher_name == "gml" else AdaGML({})
    sds = {}
    for name, m in (("sfd2", sfd2), ("segnetvit", seg), (matcher_name, matcher)):
        sds[name] = W.make_state_dict(name, m.state_dict(), seed=7)
        m.load_state_dict(sds[name], strict=True)
        m.to(dev).eval()
    return sfd2, seg, matcher, sds


def make_reference_sets(q_desc, q_kpts, q_scores, seed_base):
    """ref set = permuted copy of the query descriptors + 0.2 N(0,1)/sqrt(128) noise,
    re-normalised, last 25 % replaced by random unit vectors.  Built once, untimed."""
    from pram_amd import weights as W
    B, k, D = q_desc.shape
    dev = q_desc.device
    descs, kps, scs = [], [], []
    for b in range(B):
        seed = seed_base + b
        perm = torch.argsort(W.uniform(seed, "bench/perm", (k,), 0.0, 1.0)).to(dev)
        d = q_desc[b, perm] + W.normal(seed, "bench/noise", (k, D), 0.2 / D ** 0.5).to(dev)
        kp = q_kpts[b, perm].clone()
        n_out = k // 4
        d[k - n_out:] = W.normal(seed, "bench/out", (n_out, D), 1.0).to(dev)
        kp[k - n_out:, 0] = torch.floor(W.uniform(seed, "bench/ox", (n_out,), 4.0, W_IMG - 4.0)).to(dev)
        kp[k - n_out:, 1] = torch.floor(W.uniform(seed, "bench/oy", (n_out,), 4.0, H - 4.0)).to(dev)
        descs.append(torch.nn.functional.normalize(d, dim=-1))
        kps.append(kp)
        scs.append(W.uniform(seed, "bench/sc", (k,), 0.0, 1.0).to(dev))
    return {"descriptors": torch.stack(descs).contiguous(), "keypoints": torch.stack(kps).contiguous(),
            "scores": torch.stack(scs).contiguous()}


def usable_cores() -> int:
    """Cores this process may actually run on: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores even inside a quota-limited container, and an
    oversubscribed OpenMP pool makes the CPU baseline pathologically slow)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(sds, matcher_name, n_queries, kpts, budget_s=40.0):
    """The oracle (CPU restatement pinned to the reference) on the host cores, same synthetic workload."""
    from oracle import ref_cpu as R
    from pram_amd import weights as W
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} threads (os.cpu_count() = {os.cpu_count()})", file=sys.stderr, flush=True)
    times = []
    t_begin = time.perf_counter()
    for i in range(n_queries + 1):          # first one is the warm-up
        if i >= 2 and time.perf_counter() - t_begin > budget_s:
            break
        img = W.synthetic_image(i)[None]
        t0 = time.perf_counter()
        with torch.no_grad():
            o = R.sfd2_extract_local_global(sds["sfd2"], img, max_keypoints=kpts, min_keypoints=128, per_image_fallback=True)
            kp = o["keypoints"][0]
            _, seg = R.sfd2_sample(o["score_map"], o["mid_features"], kp, norm_desc=False)
            R.segnetvit_forward(sds["segnetvit"], seg.t()[None], kp[None], img.shape)
            d0 = o["descriptors"][0].t()[None]
            data = {"descriptors0": d0, "keypoints0": kp[None], "scores0": o["scores"][0][None],
                    "descriptors1": d0.flip(1).contiguous(), "keypoints1": kp.flip(0)[None], "scores1": o["scores"][0].flip(0)[None],
                    "image_shape0": (1, 3, W_IMG, H), "image_shape1": (1, 3, W_IMG, H)}
            if matcher_name == "gml":
                R.gml_produce_matches(sds["gml"], data)
            else:
                R.adagml_produce_matches(sds["adagml"], data)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu query {i}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    per_q = min(times[1:]) if len(times) > 1 else times[0]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": 1.0 / per_q, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{max(1, len(times) - 1)} full queries (SFD2+sample+SegNetViT+{matcher_name.upper()} 2048x2048) after 1 warm-up, min; "
                      f"torch CPU fp32, {cores} threads, {model}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch-per-gpu", type=int, default=16,
                    help="queries per GPU per step (BASELINE configs[1]: batch = 16 on one MI355X; weak scaling keeps it per GPU)")
    ap.add_argument("--kpts", type=int, default=2048)
    ap.add_argument("--n-class", type=int, default=113, help="landmark classes (7Scenes 113, Cambridge 161, Aachen 513)")
    ap.add_argument("--matcher", default="gml", choices=["gml", "adagml"])
    ap.add_argument("--stages", default="erm", help="e=extract r=recognise m=match")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight per GPU: consecutive steps are issued round-robin on this many HIP streams, so the "
                         "HBM-bound kernels of one batch run under the MFMA-bound kernels of the next (throughput mode)")
    ap.add_argument("--cpu-queries", type=int, default=4, help="CPU-baseline sample size (0 = skip)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16"],
                    help="f32 (default, the parity configuration) or f16 = BASELINE C5 'fp16 MFMA path' (fp16 operands, "
                         "fp32 accumulate) for attention, token GEMMs and convolutions; own tolerance, not the headline")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # PRAM_BENCH_ONE_DEVICE=1 (test hook): every rank uses GPU 0 over gloo, to exercise the multi-rank control flow
    # (barriers, result gather, max-over-ranks timing) on a one-GPU box; never set by the driver
    one_device = os.environ.get("PRAM_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # backend 'nccl' is RCCL on ROCm
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from pram_amd import ops, weights as Wt
    from pram_amd.pipeline import QueryPipeline, gather_records
    if args.precision:
        ops.attention_precision = ops.gemm_precision = args.precision
    sfd2, seg, matcher, sds = build_models(dev, args.matcher, args.n_class)
    pipe = QueryPipeline(sfd2, seg, matcher, max_keypoints=args.kpts, min_keypoints=128)

    B = args.batch_per_gpu
    q0 = rank * B                                          # weak scaling: fixed queries per GPU
    images = torch.stack([Wt.synthetic_image(q0 + i) for i in range(B)]).to(dev).contiguous()
    with torch.no_grad():
        ex = sfd2.extract_batched(images, pipe.cfg)
        counts = ex["counts"].tolist()
        ref = make_reference_sets(ex["descriptors"], ex["keypoints"], ex["scores"], 5000 + q0) if "m" in args.stages else None
    del ex

    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)] if args.inflight > 1 else None
    issued = [0]

    def step():
        if lanes is None:
            out = pipe.run(images, ref, stages=args.stages)
            rec = QueryPipeline.pack_record(out)
            return gather_records(rec)
        lane = lanes[issued[0] % len(lanes)]
        issued[0] += 1
        with torch.cuda.stream(lane):
            out = pipe.run(images, ref, stages=args.stages)
            rec = QueryPipeline.pack_record(out)
        if world == 1:
            return rec
        # the (tiny) all-gather stays on the one main stream, in step order on every rank: RCCL never sees
        # collectives of one communicator issued from several streams
        main = torch.cuda.current_stream(dev)
        main.wait_stream(lane)
        rec.record_stream(main)
        return gather_records(rec)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_matches = int((rec[:, :, 4] >= 0).sum().item())

    # ---- roofline of the dominant kernel (attention), one extra instrumented step, HIP events on the launch stream
    lanes = None                      # the instrumented step runs alone on the current stream
    torch.cuda.synchronize()
    ops.attention_probe = []
    step()
    torch.cuda.synchronize()
    probe, ops.attention_probe = ops.attention_probe, None
    attn_ms, attn_flops = 0.0, 0.0
    for ql, kl, mm, nn, hh, bb, e0, e1 in probe:
        attn_ms += e0.elapsed_time(e1)
        qv = ql.double() if ql is not None else torch.full((bb,), float(mm), dtype=torch.float64, device=dev)
        kv = kl.double() if kl is not None else torch.full((bb,), float(nn), dtype=torch.float64, device=dev)
        attn_flops += 4.0 * 64 * hh * float((qv * kv).sum().item())
    achieved = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
    traffic = None
    if (args.kpts, B, args.matcher, args.stages, args.n_class, ops.attention_precision) == (2048, 16, "gml", "erm", 113, "f32") and ops.gemm_precision == "f32":
        try:   # HBM bytes per attention launch of THIS configuration, from the committed rocprofv3 --pmc passes
            traffic = json.load(open(ROOT / "profiles" / "r01_pmc_attention.json"))["attention_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pass
    f16 = ops.attention_precision == "f16"
    peak = 2500.0 if f16 else PEAK_F32_MFMA_TFLOPS      # dense f16/bf16 MFMA peak vs f32 MFMA peak (MI355X_MICROARCH.md)
    roofline = {"bound": "mfma", "kernel": "attention_f16_kernel (C5 fp16 MFMA path)" if f16 else "attention_kernel (f32 MFMA flash attention)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": round(attn_flops / max(1, len(probe)) / 512.0), "launches_per_step": len(probe), "avg_launch_ms": round(attn_ms / max(1, len(probe)), 4),
                "attention_share_of_step": round(attn_ms / (dt / args.steps * 1e3), 3)}

    if rank == 0:
        total_q = world * B * args.steps
        line = {
            "metric": f"query images/sec (640x480, {args.kpts} kpts, {'7Scenes ' if args.n_class == 113 else ''}nc{args.n_class})", "value": round(total_q / dt, 3), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if (ops.attention_precision, ops.gemm_precision) == ("f32", "f32") else
                     f"f16 operands / f32 accumulate (attention {ops.attention_precision}, GEMM+conv {ops.gemm_precision}): "
                     "BASELINE C5 'fp16 MFMA path', NOT the fp32 parity configuration",
            "data": "synthetic",
            "config": {"workload": f"{'7Scenes' if args.n_class == 113 else 'custom'} full hot path: SFD2 extract+sample -> SegNetViT nc{args.n_class} (15 layers) -> "
                                   f"{args.matcher.upper()} match + 20 Sinkhorn iters vs a {args.kpts}-kpt reference set; "
                                   f"stages={args.stages}",
                       "queries_per_gpu_per_step": B, "batches_in_flight_per_gpu": max(1, args.inflight),
                       "frame": f"{W_IMG}x{H}", "keypoints": args.kpts,
                       "keypoints_found": counts[:4], "parallelism": f"query-sharded x{world}, one all-gather of result records",
                       "matches_last_step": n_matches},
            "roofline": roofline,
        }
        if world == 1 and args.cpu_queries > 0:
            line["cpu_baseline"] = cpu_baseline(sds, args.matcher, args.cpu_queries, args.kpts)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
