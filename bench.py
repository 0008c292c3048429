#!/usr/bin/env python
"""Benchmark of the PRAM per-query hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic query frames (16 per GPU by default =
BASELINE configs[1]; ``--batch-total Q`` shards Q queries over the N ranks instead = configs[2] / [4]):
SFD2 extract + sample -> SegNetViT nc113 -> GML match with 20 Sinkhorn iterations against a 2048-keypoint
reference set, inputs resident in HBM.  Queries shard across ranks with no data-path collective; each step ends
with the single all-gather of the fixed-size result records.

With ``--gpus N`` (N > 1) and no launcher in the environment the script starts its own N ranks
(``torch.distributed.run``, one process per GPU over RCCL) and refuses to run when fewer than N GPUs are visible.

Rank 0 prints ONE JSON line (contract in the task statement): metric / value (whole-job queries/s) / roofline
(attention kernel) / parity (one untimed query checked against the oracle, stage by stage) / cpu_baseline (the
oracle on the host cores, per stage, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

# MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz; dense f16 MFMA ~2.5 PFLOP/s
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
H, W_IMG = 480, 640


def build_models(dev, matcher_name, n_class=113, precision=None):
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


def make_reference_sets(q_desc, q_kpts, q_scores, counts, seed_base, noise_frac=0.25, n_ref=0, m_match=0):
    """the reference set of a query = a permuted copy of its descriptors + noise, re-normalised, the
    last 25 % replaced by outliers (fresh keypoint positions, descriptors drawn around the population mean).  The noise
    (norm = noise_frac x) and the outliers are scaled to the DISCRIMINATIVE part of the descriptors (||d - mean|| ~ 0.18
    on the synthetic SFD2 weights), not to their unit norm.  Built once, untimed.  n_ref: size of the reference sets (0 = the
    query size k); m_match: only the first m_match query keypoints enter the matcher (0 = all) — the twins are drawn among
    those.  Returns (ref, gt) with gt[b, i] = index of query keypoint i's twin in the reference set or -1 (i < m_match)."""
    from pram_amd import weights as W
    B, k, D = q_desc.shape
    M = m_match or k
    N = n_ref or k
    dev = q_desc.device
    mu = q_desc.reshape(-1, D).mean(0)
    spread = float((q_desc.reshape(-1, D) - mu).norm(dim=1).mean())
    n_in = min(N - N // 4, M)
    n_out = N - n_in
    descs, kps, scs, gts = [], [], [], []
    for b in range(B):
        seed = seed_base + b
        perm = torch.argsort(W.uniform(seed, "bench/perm", (M,), 0.0, 1.0)).to(dev)
        src = perm[torch.arange(N, device=dev) % M]
        d = q_desc[b, src] + W.normal(seed, "bench/noise", (N, D), noise_frac * spread / D ** 0.5).to(dev)
        kp = q_kpts[b, src].clone()
        d[n_in:] = mu + W.normal(seed, "bench/out", (n_out, D), spread / D ** 0.5).to(dev)
        kp[n_in:, 0] = torch.floor(W.uniform(seed, "bench/ox", (n_out,), 4.0, W_IMG - 4.0)).to(dev)
        kp[n_in:, 1] = torch.floor(W.uniform(seed, "bench/oy", (n_out,), 4.0, H - 4.0)).to(dev)
        descs.append(torch.nn.functional.normalize(d, dim=-1))
        kps.append(kp)
        scs.append(W.uniform(seed, "bench/sc", (N,), 0.0, 1.0).to(dev))
        gt = torch.full((M,), -1, dtype=torch.long, device=dev)
        inl = torch.arange(n_in, device=dev)
        s_in = src[:n_in]
        ok = s_in < counts[b]                       # padded query rows have no twin
        gt[s_in[ok]] = inl[ok]
        gts.append(gt)
    ref = {"descriptors": torch.stack(descs).contiguous(), "keypoints": torch.stack(kps).contiguous(),
           "scores": torch.stack(scs).contiguous()}
    return ref, torch.stack(gts)


class PowerSampler:
    """Socket power and shader clock of one GPU during the timed region, read from the amdgpu hwmon files (power1_input in uW,
    freq1_input in Hz, power1_cap) every 20 ms on a host thread: no tool, no GPU work.  Best effort — any failure gives None.
    The matrix kernels of this path run AT the board's power cap (profiles/r02_power_probe.txt), which is what bounds them."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.dir = None
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        try:
            props = torch.cuda.get_device_properties(device_index)
            want = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
                if os.path.basename(os.path.realpath(os.path.join(d, "device"))) == want and os.path.exists(os.path.join(d, "power1_input")):
                    self.dir = d
                    break
        except Exception:
            self.dir = None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
            except Exception:
                pass
            self._stop.wait(0.02)

    def start(self):
        if self.dir is not None:
            import threading
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self, trace_bins=0):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join(timeout=1.0)
        if not self.samples:
            return None
        w = [a for a, _ in self.samples]
        f = [b for _, b in self.samples]
        out = {"socket_w_mean": round(sum(w) / len(w), 1), "socket_w_max": round(max(w), 1), "sclk_mhz_mean": round(sum(f) / len(f), 1),
               "samples": len(w), "source": "amdgpu hwmon power1_input / freq1_input, 20 ms period, timed region only"}
        if trace_bins and len(w) >= trace_bins:      # the run cut into equal parts: [mean W, mean MHz] of each (does it sag as the package heats up?)
            n = len(w) // trace_bins
            out["trace"] = [[round(sum(w[i * n:(i + 1) * n]) / n), round(sum(f[i * n:(i + 1) * n]) / n)] for i in range(trace_bins)]
        try:
            out["socket_w_cap"] = round(self._read("power1_cap") * 1e-6, 1)
        except Exception:
            pass
        return out


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


def _oracle_matcher(name):
    from oracle import ref_cpu as R
    return R.gml_produce_matches if name == "gml" else R.adagml_produce_matches


MUTUAL_FLIP_GAP = 2e-5      # a sub-threshold mutual flip is excused only if the oracle's own assignment proves a top-2 gap this small
MAX_MUTUAL_FLIPS = 4         # ... and at most this many per query


def compare_matches(m_got, s_got, m_ref, s_ref, thr, assignment=None):
    """The matcher's part of the parity gate, as a pure function (tests/test_parity_gate.py drives it with constructed cases).

    Bars: indices identical, scores within 1e-3.  Two discrete decisions of the reference are recognised — and nothing wider:
    * threshold ties: an index may differ only where the SAME candidate sits on the acceptance threshold itself (`score > thr`,
      nets/gml.py:316, decided on two fp32 values that both lie within 1e-5 of thr and straddle it): one side reports the partner,
      the other -1;
    * sub-threshold mutual flips: `mscores0 = where(mutual0, max0, 0)` (nets/gml.py:310-314) — when the two best entries of a row
      (or of a column) of the assignment agree to the last bits, which partner is named, and with it whether the row is mutual, can
      flip between two fp32 implementations.  Excused ONLY for a candidate both sides report unmatched, whose larger score is below
      thr, and ONLY when `assignment` (the oracle's own [m, n] matrix) proves the near-tie: the row's two best entries agree to
      MUTUAL_FLIP_GAP, or the row is one of the two best of its best column and those agree to MUTUAL_FLIP_GAP.  Without the
      matrix (AdaGML's restatement hands none out) nothing is excused.  At most MAX_MUTUAL_FLIPS per query.
    -> (ok, report)"""
    flip = (m_got == -1) & (m_ref == -1) & ((s_got == 0) ^ (s_ref == 0)) & (torch.maximum(s_got, s_ref) <= thr)
    gaps = {}
    if assignment is None:
        flip[:] = False
    elif bool(flip.any()):
        P = assignment
        for i in torch.nonzero(flip).flatten().tolist():
            rt = P[i].topk(2)
            ct = P[:, int(rt.indices[0])].topk(2)
            row_gap, col_gap = float(rt.values[0] - rt.values[1]), float(ct.values[0] - ct.values[1])
            gaps[i] = {"row_top2_gap": row_gap, "column_top2_gap": col_gap}
            if not (row_gap <= MUTUAL_FLIP_GAP or (i in ct.indices.tolist() and col_gap <= MUTUAL_FLIP_GAP)):
                flip[i] = False
    d_sc = float(((s_got - s_ref).abs() * (~flip)).max()) if s_got.numel() else 0.0
    bad = torch.nonzero(m_got != m_ref).flatten()
    ties = [int(i) for i in bad if abs(float(s_ref[i]) - thr) < 1e-5 and abs(float(s_got[i]) - thr) < 1e-5
            and min(int(m_got[i]), int(m_ref[i])) == -1]
    idx_same = len(ties) == bad.numel()
    rep = {"indices_identical": idx_same, "matches": int((m_ref >= 0).sum()), "scores_maxdiff": float(f"{d_sc:.3e}"),
           "threshold_ties": len(ties), "mutual_flips_below_threshold": int(flip.sum())}
    if bool(flip.any()):
        rep["mutual_flips"] = [{"i": int(i), "score_got": float(s_got[i]), "score_ref": float(s_ref[i]), **(gaps.get(int(i)) or {})}
                               for i in torch.nonzero(flip).flatten()[:MAX_MUTUAL_FLIPS]]
    if bad.numel():
        rep["differing"] = [{"i": int(i), "got": int(m_got[i]), "ref": int(m_ref[i]), "score_got": float(s_got[i]), "score_ref": float(s_ref[i])}
                            for i in bad[:8]]
    if d_sc >= 1e-3:      # where the scores part company (diagnostic: a handful of entries)
        far = torch.nonzero(((s_got - s_ref).abs() >= 1e-3) & ~flip).flatten()
        rep["score_outliers"] = [{"i": int(i), "match_got": int(m_got[i]), "match_ref": int(m_ref[i]),
                                  "score_got": float(s_got[i]), "score_ref": float(s_ref[i])} for i in far[:6]]
    return bool(idx_same and d_sc < 1e-3 and int(flip.sum()) <= MAX_MUTUAL_FLIPS), rep


# C5 'fp16 MFMA path': bars at 1.5 x the measured distance from the fp32 oracle — logits 4.05e-2, arg-max agreement 99.61 %
# (tests/test_gpu_configs.py); keypoint-set overlap 0.99854 and match-index agreement 0.99951 at 4096 keypoints (profiles/r06_bench_*)
F16_BARS = {"logits": 6.1e-2, "argmax": 0.9942, "kp_overlap": 0.9978, "match_agree": 0.9992}


def parity_gate(pipe, sds, matcher_name, images, ref, kpts, out=None, queries=(0,), f16=False):
    """Queries of the TIMED batch, checked stage by stage against the oracle (CPU restatement pinned to the reference).

    `out` is the result of ONE run of the batch through the very path that was timed (all B queries in one call — main() also
    asserts that its record equals the timed steps' record bit for bit); `queries` are the batch elements checked (>= 4 of the
    16 by default: the oracle needs ~2 s per query, outside the timed region).  Per query: the extraction against the oracle's
    own extraction of the same frame (keypoint SET identical; `keypoint_order_frac` = share of top-k positions holding the same
    keypoint — 1e-7 score differences between the dense maps swap neighbours in the sorted list); the recogniser and the
    matcher stage-isolated (the oracle is fed the HIP path's keypoints / descriptors, so one swapped pair in the keypoint order
    cannot cascade into a spurious mismatch).  Bars: fp32 outputs 1e-3, indices exact (north_star); compare_matches() has the
    two recognised discrete decisions of the matcher.
    f16=True: the C5 'fp16 MFMA path' against ITS documented bars (F16_BARS: the recogniser's logits / arg-max, the keypoint-set
    overlap and the match-index agreement as FRACTIONS, each bar at 1.5 x the measured distance from the fp32 oracle: a single fp16
    product per MAC is not an fp32 parity configuration)."""
    from oracle import ref_cpu as R
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        if out is None:
            out = pipe.run(images, ref, stages="erm" if ref is not None else "er")
        torch.cuda.synchronize()
        per, ok_all = [], True
        thr = float(getattr(pipe.matcher, "match_threshold", 0.2))
        for b in queries:
            img = images[b:b + 1]
            n = int(out["counts"][b].item())
            kp = out["keypoints"][b, :n].cpu()
            o = R.sfd2_extract_local_global(sds["sfd2"], img.cpu(), max_keypoints=kpts, min_keypoints=128, per_image_fallback=True)
            okp = o["keypoints"][0]
            same_set = {(int(x), int(y)) for x, y in kp.tolist()} == {(int(x), int(y)) for x, y in okp.tolist()}
            same_pos = float((kp[:min(n, len(okp))] == okp[:min(n, len(okp))]).all(1).float().mean()) if n else 1.0
            res = {"query": int(b), "extract": {"keypoints": n, "keypoint_set_identical": bool(same_set and n == len(okp)),
                                                "keypoint_order_frac": round(same_pos, 5)}}
            ok = res["extract"]["keypoint_set_identical"] or f16
            if f16:
                a, c = {(int(x), int(y)) for x, y in kp.tolist()}, {(int(x), int(y)) for x, y in okp.tolist()}
                res["extract"]["keypoint_set_overlap"] = round(len(a & c) / max(1, len(c)), 5)
                ok = res["extract"]["keypoint_set_overlap"] >= F16_BARS["kp_overlap"]
            # recogniser, stage-isolated
            _, segd = R.sfd2_sample(o["score_map"], o["mid_features"], kp, norm_desc=False)
            ref_logits = R.segnetvit_forward(sds["segnetvit"], segd.t()[None], kp[None], tuple(img.shape))[0]
            got = out["prediction"][b, :n].cpu()
            d_log = float((got - ref_logits).abs().max())
            agree = float((got.argmax(-1) == ref_logits.argmax(-1)).float().mean())
            res["recognise"] = {"logits_maxdiff": float(f"{d_log:.3e}"), "argmax_agreement": round(agree, 6)}
            ok = ok and ((d_log < F16_BARS["logits"] and agree >= F16_BARS["argmax"]) if f16 else (d_log < 1e-3 and agree == 1.0))
            if ref is not None and "matches0" in out:
                nm = min(n, pipe.match_keypoints) if pipe.match_keypoints else n      # the matcher's share of the query's keypoints
                data = {"descriptors0": out["descriptors"][b:b + 1, :nm].cpu(), "keypoints0": kp[None, :nm], "scores0": out["scores"][b:b + 1, :nm].cpu(),
                        "descriptors1": ref["descriptors"][b:b + 1].cpu(), "keypoints1": ref["keypoints"][b:b + 1].cpu(),
                        "scores1": ref["scores"][b:b + 1].cpu(), "image_shape0": (1, 3, W_IMG, H), "image_shape1": (1, 3, W_IMG, H)}
                probes = {} if matcher_name == "gml" else None      # (the GML restatement hands out its assignment matrix)
                r = _oracle_matcher(matcher_name)(sds[matcher_name], data, **({"probes": probes} if probes is not None else {}))
                m_ok, res["match"] = compare_matches(out["matches0"][b, :nm].cpu(), out["matching_scores0"][b, :nm].cpu(),
                                                     r["matches0"][0], r["matching_scores0"][0], thr,
                                                     probes["score"][0, :-1, :-1] if probes else None)
                if f16:
                    mg, mr = out["matches0"][b, :nm].cpu(), r["matches0"][0]
                    res["match"]["index_agreement"] = round(float((mg == mr).float().mean()), 5)
                    ok = ok and res["match"]["index_agreement"] >= F16_BARS["match_agree"]
                else:
                    ok = ok and m_ok
            res["ok"] = bool(ok)
            ok_all = ok_all and ok
            per.append(res)
    worst = {"queries_checked": [int(b) for b in queries],
             "keypoint_set_identical": all(q["extract"]["keypoint_set_identical"] for q in per),
             "keypoint_order_frac_min": min(q["extract"]["keypoint_order_frac"] for q in per),
             "logits_maxdiff_max": max(q["recognise"]["logits_maxdiff"] for q in per),
             "argmax_agreement_min": min(q["recognise"]["argmax_agreement"] for q in per)}
    if f16:      # the fp16 path's discrete outputs as agreement FRACTIONS (a boolean cannot say whether 1 or 1 000 keypoints differ)
        worst["keypoint_set_overlap_min"] = min(q["extract"]["keypoint_set_overlap"] for q in per)
        if all("match" in q for q in per):
            worst["match_index_agreement_min"] = min(q["match"]["index_agreement"] for q in per)
    if all("match" in q for q in per):
        worst.update({"match_indices_identical": all(q["match"]["indices_identical"] for q in per),
                      "match_scores_maxdiff_max": max(q["match"]["scores_maxdiff"] for q in per),
                      "threshold_ties": sum(q["match"]["threshold_ties"] for q in per),
                      "mutual_flips_below_threshold": sum(q["match"]["mutual_flips_below_threshold"] for q in per)})
    return {"ok": bool(ok_all), **worst, "per_query": per,
            "source": "the batch's own outputs (one run of all queries through the timed path; its record equals the timed steps' record bit for bit)",
            "bars": (f"fp16 path's own bars: logits <= {F16_BARS['logits']}, arg-max agreement >= {F16_BARS['argmax']} (recogniser, stage-isolated); "
                     f"keypoint-set overlap >= {F16_BARS['kp_overlap']}, match-index agreement >= {F16_BARS['match_agree']} (fractions, 1.5 x the measured deficit)" if f16 else
                     "fp32 outputs <= 1e-3 abs, indices exact, keypoint SET exact (order: see keypoint_order_frac); recogniser / matcher "
                     "stage-isolated on the HIP path's keypoints")}


def cpu_baseline(sds, matcher_name, n_queries, kpts, ref_cpu_sets, budget_s=40.0):
    """The oracle (CPU restatement pinned to the reference) on the host cores, same synthetic workload, per stage."""
    from oracle import ref_cpu as R
    from pram_amd import weights as W
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} threads (os.cpu_count() = {os.cpu_count()})", file=sys.stderr, flush=True)
    times = []
    t_begin = time.perf_counter()
    match = _oracle_matcher(matcher_name)
    for i in range(n_queries + 1):          # first one is the warm-up
        if i >= 2 and time.perf_counter() - t_begin > budget_s:
            break
        img = W.synthetic_image(i)[None]
        with torch.no_grad():
            t0 = time.perf_counter()
            o = R.sfd2_extract_local_global(sds["sfd2"], img, max_keypoints=kpts, min_keypoints=128, per_image_fallback=True)
            kp = o["keypoints"][0]
            t1 = time.perf_counter()
            _, seg = R.sfd2_sample(o["score_map"], o["mid_features"], kp, norm_desc=False)
            R.segnetvit_forward(sds["segnetvit"], seg.t()[None], kp[None], img.shape)
            t2 = time.perf_counter()
            t3 = t2
            if ref_cpu_sets is not None:
                j = min(i, ref_cpu_sets["descriptors"].shape[0] - 1)
                data = {"descriptors0": o["descriptors"][0].t()[None], "keypoints0": kp[None], "scores0": o["scores"][0][None],
                        "descriptors1": ref_cpu_sets["descriptors"][j:j + 1], "keypoints1": ref_cpu_sets["keypoints"][j:j + 1],
                        "scores1": ref_cpu_sets["scores"][j:j + 1], "image_shape0": (1, 3, W_IMG, H), "image_shape1": (1, 3, W_IMG, H)}
                match(sds[matcher_name], data)
                t3 = time.perf_counter()
        times.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
        print(f"[bench] cpu query {i}: {times[-1][0]:.2f} s (extract {times[-1][1]:.2f}, recognise {times[-1][2]:.2f}, match {times[-1][3]:.2f})",
              file=sys.stderr, flush=True)
    use = times[1:] if len(times) > 1 else times
    best = min(use, key=lambda t: t[0])
    tot = sorted(t[0] for t in use)
    median = tot[len(tot) // 2] if len(tot) % 2 else 0.5 * (tot[len(tot) // 2 - 1] + tot[len(tot) // 2])
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(1.0 / best[0], 4), "value_median": round(1.0 / median, 4), "unit": "queries/s", "cores": cores, "kind": "port",
            "seconds_per_query": {"min": round(best[0], 3), "median": round(median, 3), "max": round(tot[-1], 3)},
            "stage_seconds": {"extract": round(best[1], 3), "recognise": round(best[2], 3), "match": round(best[3], 3)},
            "sample": f"{len(use)} full queries (SFD2+sample+SegNetViT+{matcher_name.upper()} {kpts}x{ref_cpu_sets['descriptors'].shape[1] if ref_cpu_sets is not None else 0}) "
                      f"after 1 warm-up: value = fastest, value_median = median; torch CPU fp32, {cores} threads, {model}"}


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """--gpus N without a launcher: start N ranks of this script (one process per GPU), like the reference's
    mp.spawn -> init_process_group("nccl") (main.py:111-155).  Never silently measures fewer GPUs than asked."""
    one_device = os.environ.get("PRAM_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have == 0:
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    if have < n and not one_device:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node — refusing to report n_gpus={n} "
                         f"from fewer devices")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PRAM_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] --gpus {n}: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def attention_roofline(probe, precision, dev, step_ms):
    """Roofline of the dominant kernel (attention) from the HIP-event probe of one instrumented step.

    Algorithmic FLOPs follow SURVEY.md §8(d): self attention 4 m n 64 per (sequence, head) (= 1024 N^2 per 4-head layer); cross
    attention 3 * 2 m n 64 per (pair, head) and direction-pair, i.e. 1536 M N per layer — the reference multiplies q k^T ONCE for
    both directions (nets/gml.py:175-179) while the kernel, like every flash formulation, multiplies it once per direction: the
    launch executes 4/3 of the algorithmic work and that shows up as a LOWER fraction, not as extra credit.
    The split-fp16 kernel issues `mfma_per_tile` v_mfma_f32_32x32x16_f16 per 64-key tile where a single-product fp16 attention
    needs 16 (the library reports what the launch issues: 48 = 3 per product by default; 40 = 2.5 from 1024 keys on when
    pram_attention_x3_set_p_split(0) carries the probabilities as one fp16): the fp32-class ceiling of THIS kernel is
    the fp16 MFMA peak divided by that multiplier."""
    t_ms, alg, launched, executed = 0.0, 0.0, 0.0, 0.0
    for ql, kl, mm, nn, hh, bb, e0, e1, kind, mfma_tile in probe:
        t_ms += e0.elapsed_time(e1)
        qv = ql.double() if ql is not None else torch.full((bb,), float(mm), dtype=torch.float64, device=dev)
        kv = kl.double() if kl is not None else torch.full((bb,), float(nn), dtype=torch.float64, device=dev)
        f = 4.0 * 64 * hh * float((qv * kv).sum().item())       # q k^T + p v of every (sequence, head) the launch computes
        launched += f
        alg += f * (0.75 if kind == "cross" else 1.0)
        executed += f * (mfma_tile / 16.0)
    n = max(1, len(probe))
    secs = max(t_ms * 1e-3, 1e-12)
    achieved = alg / secs / 1e12
    mult = executed / launched if launched else 1.0              # MFMA flops executed per flop of a one-product kernel
    pipe_peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
    peak = pipe_peak / mult
    kern = {"f32": "attention_kernel (v_mfma_f32_32x32x2_f32 flash attention, exact fp32 products)",
            "x3": "attention_x3_pipe_kernel (split-fp16: 3 x v_mfma_f32_32x32x16_f16 per product; mfma_per_product says what this run issued)",
            "f16": "attention_x3_pipe_kernel<HI> (C5 fp16 MFMA path: one product)"}[precision]
    return {"bound": "mfma", "kernel": kern, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "frac_fp32_class": round(achieved / peak, 4), "frac_fp16_peak": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
            "mfma_pipe_busy": round(executed / secs / 1e12 / pipe_peak, 4),
            "mfma_per_product": round(mult, 3), "mfma_pipe_peak": pipe_peak,
            "flops_convention": "SURVEY 8(d): self 1024 N^2 per layer and sequence, cross 1536 M N per layer and pair (one q.k^T for both directions)",
            "launched_over_algorithmic_flops": round(launched / alg, 4) if alg else None,
            "algorithmic_bytes_per_launch": round(launched / n / 512.0), "launches_per_step": len(probe),
            "avg_launch_ms": round(t_ms / n, 4), "attention_share_of_step": round(t_ms / step_ms, 3) if step_ms else None}


def pmc_traffic(precision):
    """HBM bytes per attention launch from the committed rocprofv3 --pmc passes — only if they were taken on THIS kernel source
    (profiles/pmc_attention.json records the sha256 of the kernel file it measured); None otherwise."""
    import hashlib
    src = {"x3": "attention_x3.hip", "f32": "attention.hip", "f16": "attention_x3.hip"}[precision]
    try:
        ent = json.load(open(ROOT / "profiles" / "pmc_attention.json"))[precision]
        sha = hashlib.sha256((ROOT / "pram_amd" / "csrc" / src).read_bytes()).hexdigest()[:16]
        if ent.get("kernel_source_sha16") != sha:
            return None
        return ent["hbm_bytes_per_launch"]
    except Exception:
        return None


_MASKED = []      # (keeps the HIP handles of masked streams alive for the process)


def masked_stream(dev, mode: str, i: int):
    """A stream that may only use part of the chip.  mode "eo": lane i gets the even (i even) or odd CUs of the mask's bit order;
    "halves": bits 0-127 or 128-255; "w": alternate 32-bit words.  -> torch.cuda.ExternalStream"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = {"eo": [0x55555555, 0xaaaaaaaa], "halves": None, "w": None}[mode]
    if mode == "eo":
        m = [words[i % 2]] * 8
    elif mode == "halves":
        m = ([0xffffffff] * 4 + [0] * 4) if i % 2 == 0 else ([0] * 4 + [0xffffffff] * 4)
    else:
        m = [0xffffffff if (j % 2) == (i % 2) else 0 for j in range(8)]
    arr = (ctypes.c_uint32 * 8)(*m)
    st = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    _MASKED.append(st)
    return torch.cuda.ExternalStream(st.value, device=dev)


class Job:
    """One configuration of the hot path on this rank: models, resident inputs, the lanes (streams, optionally one captured
    hipGraph per lane) and the step function."""

    def __init__(self, dev, rank, world, q0, B, matcher_name, kpts, n_class, stages, inflight, use_graph, precision=None,
                 ref_kpts=0, match_kpts=0, shard_sizes=None, segk=0, h2d=False, rotate=1, act_scale=None):
        from pram_amd import ops, weights as Wt
        from pram_amd.pipeline import GraphedPipeline, QueryPipeline
        self.dev, self.world, self.B, self.stages, self.matcher_name = dev, world, B, stages, matcher_name
        self.shard_sizes = shard_sizes
        self.uneven = shard_sizes is not None and min(shard_sizes) != max(shard_sizes)
        self.sfd2, self.seg, self.matcher, self.sds = build_models(dev, matcher_name, n_class)
        for m in (self.sfd2, self.seg, self.matcher):
            m.set_precision(precision)
            if act_scale is not None:      # the split planes carry value * act_scale instead of value * 16 (what the range guard would settle on for a hot checkpoint)
                m.set_act_scale(act_scale)
        # the bench keeps several steps in flight: the range guard is read once, after the timed region (guard="deferred")
        self.pipe = QueryPipeline(self.sfd2, self.seg, self.matcher, max_keypoints=kpts, min_keypoints=128, guard="deferred",
                                  match_keypoints=match_kpts)
        self.images = torch.stack([Wt.synthetic_image(q0 + i) for i in range(B)]).to(dev).contiguous()
        self.do_match = "m" in stages
        self.ref = self.gt = None
        with torch.no_grad(), ops.precision_scope(precision):
            if self.do_match:
                # synthetic matcher weights calibrated to the extractor's descriptor statistics (weights.calibrate_matcher_input):
                # without it the untrained matcher sees near-identical tokens and the record holds no matches at all.  Calibrated
                # on the SAME frame (global query 0) on every rank, so all ranks — and the oracle of the parity gate — share one
                # set of weights.
                cal = self.sfd2.extract_batched(Wt.synthetic_image(0)[None].to(dev).contiguous(), self.pipe.cfg)
                self.sds[matcher_name] = Wt.calibrate_matcher_input(self.sds[matcher_name], cal["descriptors"][0, :int(cal["counts"][0])])
                self.matcher.load_state_dict(self.sds[matcher_name], strict=True)
                self.matcher.to(dev).eval()
                del cal
            ex = self.sfd2.extract_batched(self.images, self.pipe.cfg)
            self.counts = ex["counts"].tolist()
            if self.do_match:
                self.ref, self.gt = make_reference_sets(ex["descriptors"], ex["keypoints"], ex["scores"], self.counts, 5000 + q0,
                                                        n_ref=ref_kpts, m_match=match_kpts)
            del ex
        # rotate > 1 (alt.sustained): that many DISTINCT batches (frames q0 + 1000 i ..., each with its own reference sets), walked
        # round-robin by the steps — with a lane count that does not divide it every lane meets every batch
        self.rot = [(self.images, self.ref)]
        for i in range(1, int(rotate)):
            im = torch.stack([Wt.synthetic_image(q0 + 1000 * i + j) for j in range(B)]).to(dev).contiguous()
            rf = None
            with torch.no_grad(), ops.precision_scope(precision):
                if self.do_match:
                    ex = self.sfd2.extract_batched(im, self.pipe.cfg)
                    rf, _ = make_reference_sets(ex["descriptors"], ex["keypoints"], ex["scores"], ex["counts"].tolist(), 5000 + q0 + 1000 * i,
                                                n_ref=ref_kpts, m_match=match_kpts)
                    del ex
            self.rot.append((im, rf))
        if len(self.rot) > 1:
            use_graph = False
        self.segk, self.h2d = int(segk), bool(h2d)
        if self.segk:
            self._build_segk(q0, kpts)
            use_graph = False
        if self.h2d:
            # the reference's per-frame preparation inside the step (localization/loc_by_rec_online.py:86-106): uint8 frames in
            # pinned host memory -> one H2D copy per step on the lane's stream -> pram_stage_frames_u8 (/ 255, Normalize, HWC -> CHW)
            from pram_amd.nets.sfd2 import RGB_mean, RGB_std
            m = torch.tensor(RGB_mean).view(1, 3, 1, 1)
            sd_ = torch.tensor(RGB_std).view(1, 3, 1, 1)
            u8 = ((self.images.cpu() * sd_ + m) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            self.host_u8 = u8.pin_memory()
            self.dev_u8 = torch.empty_like(u8, device=dev)
            self.lut = ops.frame_lut(RGB_mean, RGB_std, dev)
            self.images = ops.stage_frames(u8.to(dev), self.lut)      # what the staged frames are (the resident run rounds differently)
            use_graph = False
        self.lanes = [torch.cuda.Stream(device=dev) for _ in range(inflight)] if inflight > 1 else None
        cu_mask = os.environ.get("PRAM_BENCH_CU_MASK", "")
        if self.lanes is not None and cu_mask:
            # experiment (profiles/r06_cu_mask.txt): lanes pinned to complementary halves of the chip (hipExtStreamCreateWithCUMask):
            # two batches then run side by side on 128 CUs each instead of taking turns on 256
            self.lanes = [masked_stream(dev, cu_mask, i) for i in range(inflight)]
        self.graphs = None
        if use_graph:
            # one captured step per lane: the host's share of a step drops from ~3 ms of ctypes launches to one graph launch
            # (what matters with eight ranks on a cgroup-limited host); GPU work is the same kernels in the same order
            self.graphs = [GraphedPipeline(self.pipe, self.images, self.ref, stages, record=True, stream=(self.lanes[i] if self.lanes else None))
                           for i in range(max(1, inflight))]
        self.issued = 0
        self.gather_events = None      # world > 1: main() arms it for the timed steps (HIP events around every all-gather)

    def _build_segk(self, q0, kpts):
        """The matcher's real call pattern (localization/multimap3d.py:112-139 -> singlemap3d.py:143-154): per query, for each of
        the seg_k best-voted landmarks, the keypoints voted to that landmark (all keypoints when they are too few) are matched
        against that landmark's reference frame — seg_k small ragged pairs, each with its own reference set.  Built once, untimed,
        from a first recognition of the batch: pair p = (query b, its i-th most voted landmark): query side = that landmark's
        keypoints (<= 512, >= 32 else the query's 512 best), reference side = a twin set of the query side (make_reference_sets)
        cut to a seeded length in [600, 1376].  The step then runs extract + recognise as usual and ALL pairs of the batch in ONE
        grouped produce_matches call (lens0 / lens1: each pair computes its own B = 1 result)."""
        from pram_amd import weights as Wt
        K, B, dev = self.segk, self.B, self.dev
        with torch.no_grad():
            out = self.pipe.run(self.images, None, stages="er")
            torch.cuda.synchronize()
        lm = out["landmark"].cpu()
        cnt = out["counts"].cpu().tolist()
        MQ, NR = 512, 1376
        idx = torch.zeros(B * K, MQ, dtype=torch.long)
        lens0 = torch.zeros(B * K, dtype=torch.int32)
        for b in range(B):
            ids = lm[b, :cnt[b]]
            fg = ids[ids >= 0]
            order = torch.bincount(fg).argsort(descending=True)[:K].tolist() if fg.numel() else []
            for i in range(K):
                sel = torch.nonzero(ids == order[i]).flatten()[:MQ] if i < len(order) else torch.zeros(0, dtype=torch.long)
                if sel.numel() < 32:                                  # (the reference's fallback: all keypoints; capped like the secondary shape)
                    sel = torch.arange(min(MQ, cnt[b]))
                idx[b * K + i, :sel.numel()] = sel
                lens0[b * K + i] = sel.numel()
        self.sk_q = torch.arange(B).repeat_interleave(K).to(dev)
        self.sk_idx, self.sk_lens0 = idx.to(dev), lens0.to(dev)
        sub = lambda t: t[self.sk_q[:, None], self.sk_idx]
        d0, k0, s0 = sub(out["descriptors"]), sub(out["keypoints"]), sub(out["scores"])
        ref, gt = make_reference_sets(d0, k0, s0, lens0.tolist(), 7000 + q0, n_ref=NR)
        lens1 = torch.floor(Wt.uniform(11 + q0, "bench/segk/lens1", (B * K,), 600.0, NR + 0.999)).to(torch.int32).to(dev)
        gt = torch.where(gt < lens1[:, None].long(), gt, torch.full_like(gt, -1))
        pad0 = torch.arange(MQ, device=dev)[None] >= self.sk_lens0[:, None]
        gt[pad0] = -1
        self.sk_ref, self.sk_lens1, self.sk_gt = ref, lens1, gt
        self.segk_sizes = {"pairs_per_step": B * K, "query_side": [int(lens0.min()), int(lens0.float().mean()), int(lens0.max())],
                           "reference_side": [int(lens1.min()), int(lens1.float().mean()), int(lens1.max())]}

    def _segk_step(self):
        from pram_amd.pipeline import QueryPipeline
        out = self.pipe.run(self.images, None, stages="er")
        sub = lambda t: t[self.sk_q[:, None], self.sk_idx]      # (torch indexing: this mode's gather is a framework kernel)
        data = {"descriptors0": sub(out["descriptors"]), "keypoints0": sub(out["keypoints"]), "scores0": sub(out["scores"]),
                "lens0": self.sk_lens0, "image_shape0": (1, 3, W_IMG, H),
                "descriptors1": self.sk_ref["descriptors"], "keypoints1": self.sk_ref["keypoints"], "scores1": self.sk_ref["scores"],
                "lens1": self.sk_lens1, "image_shape1": (1, 3, W_IMG, H)}
        m = self.matcher.produce_matches(data)
        self.sk_matches = m["matches0"]
        return QueryPipeline.pack_record(out)

    def _upload(self):
        self.dev_u8.copy_(self.host_u8, non_blocking=True)
        from pram_amd import ops
        ops.stage_frames(self.dev_u8, self.lut, out=self.images)

    def step(self):
        from pram_amd.pipeline import QueryPipeline, gather_records
        sizes = self.shard_sizes if self.uneven else None
        i = self.issued
        self.issued += 1
        if self.lanes is None:
            if self.h2d:
                self._upload()
            if self.graphs is not None:
                self.graphs[0].replay()
                rec = self.graphs[0].record
            elif self.segk:
                rec = self._segk_step()
            else:
                im, rf = self.rot[i % len(self.rot)]
                rec = QueryPipeline.pack_record(self.pipe.run(im, rf, stages=self.stages))
            self.last_local = rec
            return gather_records(rec, sizes)
        lane = self.lanes[i % len(self.lanes)]
        with torch.cuda.stream(lane):
            if self.h2d:
                self._upload()
            if self.graphs is not None:
                g = self.graphs[i % len(self.graphs)]
                g.replay()
                rec = g.record
            elif self.segk:
                rec = self._segk_step()
            else:
                im, rf = self.rot[i % len(self.rot)]
                rec = QueryPipeline.pack_record(self.pipe.run(im, rf, stages=self.stages))
        self.last_local = rec          # this rank's own records of the step just issued (main(): checked against its slice of the gather)
        if self.world == 1:
            return rec
        # the (tiny) all-gather stays on the one main stream, in step order on every rank: RCCL never sees
        # collectives of one communicator issued from several streams
        main_s = torch.cuda.current_stream(self.dev)
        main_s.wait_stream(lane)
        rec.record_stream(main_s)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.gather_events is not None else None
        if ev:
            ev[0].record(main_s)
        full = gather_records(rec, sizes)
        if ev:
            ev[1].record(main_s)
            self.gather_events.append(ev)
        lane.wait_stream(main_s)          # a captured record buffer is rewritten by the lane's next replay: not before the gather read it
        return full

    def timed(self, steps, warmup, sync_all, ahead=0, min_seconds=0.0):
        """ahead > 0 (long runs): the host never runs more than `ahead` steps in front of the device (it waits for the event behind
        step i - ahead before issuing step i: no bubble, bounded queue and allocator footprint).  min_seconds: keep issuing steps
        (beyond `steps`) until that much wall time has passed; self.steps_done says how many ran."""
        for _ in range(warmup):
            self.step()
        sync_all()
        t0 = time.perf_counter()
        rec = None
        evs = []
        i = 0
        while i < steps or (min_seconds and time.perf_counter() - t0 < min_seconds):
            i += 1
            rec = self.step()
            if ahead:
                lane = self.lanes[(self.issued - 1) % len(self.lanes)] if self.lanes else torch.cuda.current_stream(self.dev)
                e = torch.cuda.Event()
                e.record(lane)
                evs.append(e)
                if len(evs) > ahead:
                    evs.pop(0).synchronize()
        sync_all()
        self.steps_done = i
        return time.perf_counter() - t0, rec


def pin_rank_thread(local: int, world: int):
    """Give every rank its own slice of the cores this job may use (the launching thread and every helper thread it starts
    later — HIP's, RCCL's proxy — inherit it): eight ranks otherwise migrate over one shared core set and a descheduled rank
    stalls the step's barrier.  A slice, not a single core: the helper threads must not queue behind the launching thread.
    -> the first core of the slice, or None when there are fewer cores than ranks (left alone)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        per = len(allowed) // world
        if per < 1:
            return None
        mine = allowed[local * per:(local + 1) * per]
        os.sched_setaffinity(0, set(mine))
        return mine[0]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 100 with --latency)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 6; 30 with --latency)")
    ap.add_argument("--batch-per-gpu", type=int, default=16,
                    help="queries per GPU per step (BASELINE configs[1]: batch = 16 on one MI355X; weak scaling keeps it per GPU)")
    ap.add_argument("--batch-total", type=int, default=0,
                    help="total queries per step, sharded over the ranks (BASELINE configs[2]: 64 over 8 GPUs; configs[4]: 128 "
                         "over 8); strong scaling; overrides --batch-per-gpu; uneven shards are fine")
    ap.add_argument("--kpts", type=int, default=2048)
    ap.add_argument("--ref-kpts", type=int, default=0, help="size of the reference keypoint sets the matcher sees (0 = --kpts)")
    ap.add_argument("--match-kpts", type=int, default=0,
                    help="only the best M keypoints of a query enter the matcher (0 = all): with --ref-kpts 1024 the secondary "
                         "512 x 1024 matcher shape of SURVEY.md 8(d) (recognition/recmap.py:670-692, multimap3d.py:131-139)")
    ap.add_argument("--n-class", type=int, default=113, help="landmark classes (7Scenes 113, Cambridge 161, Aachen 513)")
    ap.add_argument("--matcher", default="gml", choices=["gml", "adagml"])
    ap.add_argument("--stages", default="erm", help="e=extract r=recognise m=match")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU (default 3; 1 with --latency): consecutive steps are issued round-robin on this "
                         "many HIP streams, so the HBM-bound kernels of one batch run under the MFMA-bound kernels of the next")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay one captured hipGraph per lane instead of ~330 ctypes launches per step (auto: on for --gpus > 1 "
                         "and --latency, where the host thread matters; the GPU work is identical)")
    ap.add_argument("--latency", action="store_true",
                    help="the reference's online loop (localization/loc_by_rec_online.py:109-133): ONE query per step, one step at "
                         "a time, graph replay; ms_per_step is the per-query latency")
    ap.add_argument("--cpu-queries", type=int, default=4, help="CPU-baseline sample size (0 = skip)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle parity gate (kernel profiling runs)")
    ap.add_argument("--alt", default="auto", choices=["auto", "on", "off"],
                    help="also time, in the same run (6 steps each), AdaGML, the exact-fp32 path, the secondary 512 x 1024 matcher "
                         "shape and the one-query latency, and report them as 'alt' (auto: on for the default 1-GPU configuration)")
    ap.add_argument("--attn-chunk-keys", type=int, default=None,
                    help="keys per chunk of the split-fp16 attention (pram_attention_x3_set_chunk_keys; default: the library's 4096 = one chunk for every shipped shape, "
                         "512 with --latency so that one-frame launches split along the keys and fill the chip)")
    ap.add_argument("--precision", default=None, choices=["f32", "x3", "f16"],
                    help="MFMA path of the three matrix families: f32 = v_mfma_f32_32x32x2_f32 (exact fp32 products); "
                         "x3 = split-fp16, three v_mfma_f32_32x32x16_f16 per product (fp32-class accuracy, passes the fp32 "
                         "parity gate); f16 = BASELINE C5 'fp16 MFMA path' (single fp16 product, own tolerance, not the "
                         "headline).  Default: the package default (pram_amd.ops.default_precision)")
    args = ap.parse_args()
    if args.latency:
        args.batch_per_gpu, args.batch_total = 1, 0
    steps = args.steps if args.steps is not None else (100 if args.latency else 20)
    warmup = args.warmup if args.warmup is not None else (30 if args.latency else 6)
    inflight = args.inflight if args.inflight is not None else (1 if args.latency else 3)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # PRAM_BENCH_ONE_DEVICE=1 (test hook): every rank uses GPU 0 over gloo, to exercise the multi-rank control flow
    # (barriers, result gather, max-over-ranks timing) on a one-GPU box; never set by the driver
    one_device = os.environ.get("PRAM_BENCH_ONE_DEVICE") == "1"
    core = pin_rank_thread(local, world) if world > 1 else None
    if one_device:
        local = 0
    elif torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # backend 'nccl' is RCCL on ROCm

    from pram_amd import ops
    from pram_amd.pipeline import shard_range
    if args.precision:
        ops.set_precision(args.precision)
    from pram_amd import _lib as _plib
    chunk_keys = args.attn_chunk_keys if args.attn_chunk_keys is not None else (512 if args.latency else 0)
    chunk_keys = int(_plib.load().pram_attention_x3_set_chunk_keys(int(chunk_keys)))      # 0 = query; process-wide, before the first launch
    # workgroups a split attention launch aims at: the library's 256 (one per CU); 512 in latency mode (four key groups for a
    # one-frame launch: 4.19 -> 4.13 ms per query); PRAM_BENCH_SPLIT_TARGET overrides (profiling).  Never changes a result bit.
    split_target = int(os.environ.get("PRAM_BENCH_SPLIT_TARGET", "512" if args.latency else "-1"))
    _plib.load().pram_attention_x3_set_split_target(split_target)
    if ops.attn_prec() != ops.gemm_prec():
        raise SystemExit(f"bench.py: PRAM_GEMM_PRECISION={ops.gemm_prec()} and PRAM_ATTENTION_PRECISION={ops.attn_prec()} "
                         f"differ — the line reports ONE arithmetic (use --precision / PRAM_PRECISION)")
    precision = ops.gemm_prec()
    use_graph = args.graph == "on" or (args.graph == "auto" and (world > 1 or args.latency))

    if args.batch_total > 0:
        spans = [shard_range(args.batch_total, r, world) for r in range(world)]
        scaling = "strong"
    else:
        spans = [(r * args.batch_per_gpu, (r + 1) * args.batch_per_gpu) for r in range(world)]
        scaling = "weak"
    shard_sizes = [hi - lo for lo, hi in spans]
    q0, q1 = spans[rank]
    B = q1 - q0
    total_per_step = sum(shard_sizes)
    if B == 0:
        raise SystemExit(f"bench.py: rank {rank} has no queries (--batch-total {args.batch_total} over {world} ranks)")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    job = Job(dev, rank, world, q0, B, args.matcher, args.kpts, args.n_class, args.stages, inflight, use_graph,
              ref_kpts=args.ref_kpts, match_kpts=args.match_kpts, shard_sizes=shard_sizes)
    pipe, sds, images, ref, gt, counts, do_match = job.pipe, job.sds, job.images, job.ref, job.gt, job.counts, job.do_match

    for _ in range(warmup):
        job.step()
    sync_all()
    sampler = PowerSampler(dev.index if dev.index is not None else 0) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    if world > 1 and job.lanes is not None:
        job.gather_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        rec = job.step()
    t_issue = time.perf_counter() - t0      # host time to ISSUE the steps (a rank whose host thread is starved shows here first)
    sync_all()
    dt = time.perf_counter() - t0
    power = sampler.stop() if sampler is not None else None
    rank_ms = [dt / steps * 1e3]
    rank_issue_ms = rank_gather_ms = None
    if world > 1:
        # per rank: wall time per step, host issue time per step, mean time of the all-gather on the main stream (the wait for the
        # slowest rank of the step is inside it): the first real 1 -> 8 curve should explain itself
        gev, job.gather_events = job.gather_events, None
        g_ms = sum(a.elapsed_time(b) for a, b in gev) / len(gev) if gev else 0.0
        t = torch.tensor([dt, t_issue, g_ms * 1e-3], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rank_ms = [float(x[0].item()) / steps * 1e3 for x in allt]
        rank_issue_ms = [float(x[1].item()) / steps * 1e3 for x in allt]
        rank_gather_ms = [float(x[2].item()) * 1e3 for x in allt]
        dt = max(float(x[0].item()) for x in allt)
    assert rec.shape[0] == total_per_step, (rec.shape, total_per_step)
    local_rec = rec[q0 - spans[0][0]:q1 - spans[0][0]] if world > 1 else rec
    # the gather keeps query order: on EVERY rank, rows [q0, q1) of the gathered record are the records this rank computed in the
    # last step (uneven shards included); one flag per rank, gathered
    order_ok = bool(torch.equal(local_rec, job.last_local))
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, order_ok)
        order_ok = all(flags)
    n_matches = int((local_rec[:, :, 4] >= 0).sum().item())
    n_correct = 0
    if do_match:
        km = gt.shape[1]
        m0 = local_rec[:, :km, 4].long()
        n_matches = int((m0 >= 0).sum().item())
        n_correct = int(((m0 == gt) & (m0 >= 0)).sum().item())
        n_inliers = int((gt >= 0).sum().item())
        # the workload is built so that a known share of the twins is recovered; an empty record means it degenerated
        if n_matches < 0.1 * n_inliers:
            raise SystemExit(f"bench.py: degenerate matcher workload ({n_matches} matches for {n_inliers} planted twins)")
    # range guard of the split-fp16 path: the timed steps ran "deferred" (several in flight) — read it once, now
    # (armed by what was LAUNCHED, not by the precision setting: the fp16 path runs a few split-fp16 kernels on purpose — GML's
    # matching descriptors and their score matrix — and their range is checked like everybody's)
    x3_ran = ops.x3_launched(dev)
    range_hit = ops.x3_range_exceeded(dev) if x3_ran else False

    # ---- roofline of the dominant kernel (attention), one extra instrumented EAGER step, HIP events on the launch stream
    lanes_saved, graphs_saved = job.lanes, job.graphs
    job.lanes = job.graphs = None                      # the instrumented step runs alone on the current stream
    torch.cuda.synchronize()
    ops.attention_probe = []
    job.step()
    torch.cuda.synchronize()
    probe, ops.attention_probe = ops.attention_probe, None
    job.lanes, job.graphs = lanes_saved, graphs_saved
    roofline = attention_roofline(probe, precision, dev, dt / steps * 1e3)
    default_shape = (args.kpts, B, args.matcher, args.stages, args.n_class, args.ref_kpts, args.match_kpts) == (2048, 16, "gml", "erm", 113, 0, 0)
    roofline["traffic"] = pmc_traffic(precision) if default_shape else None
    roofline["traffic_source"] = ("profiles/pmc_attention.json (rocprofv3 --pmc passes of this kernel source: 2 x FETCH_SIZE + WRITE_SIZE)"
                                  if roofline["traffic"] is not None else None)

    parity = None
    if rank == 0 and not args.no_parity and "e" in args.stages and "r" in args.stages:
        # the gate reads the TIMED batch's own outputs: one more run of all B queries through the path that was timed; its record
        # must equal the last timed step's record bit for bit (the kernels are deterministic), which ties the two together
        with torch.no_grad():
            out_b = pipe.run(images, ref, stages=args.stages)
            rec_b = pipe.pack_record(out_b)
            torch.cuda.synchronize()
        same_rec = bool(torch.equal(rec_b, local_rec))
        nq = min(B, int(os.environ.get("PRAM_BENCH_PARITY_QUERIES", "4")))
        qs = sorted({int(round(i * (B - 1) / max(1, nq - 1))) for i in range(nq)}) if B > 1 else [0]
        parity = parity_gate(pipe, sds, args.matcher, images, ref, args.kpts, out=out_b, queries=qs)
        parity["timed_record_identical"] = same_rec
        parity["ok"] = bool(parity["ok"] and same_rec)
        del out_b, rec_b
        print(f"[bench] parity: {json.dumps(parity)}", file=sys.stderr, flush=True)

    # ---- other configurations of the same path, measured in this run (rank 0, N = 1): what the driver otherwise never sees
    alt = None
    want_alt = args.alt == "on" or (args.alt == "auto" and default_shape and world == 1 and precision == "x3" and not args.latency)
    if rank == 0 and world == 1 and want_alt:
        alt = {}
        alt_parity_failed = []

        only = [x for x in os.environ.get("PRAM_BENCH_ALTS", "").split(",") if x]      # profiling: a subset, in the usual order

        def alt_run(name, note, steps_=6, warm_=4, p_split=None, chunk=None, parity_f16=False, parity_q=0, sustain_s=0.0, **kw):
            if only and name not in only:
                return
            cfg = dict(matcher_name=args.matcher, kpts=args.kpts, n_class=args.n_class, stages=args.stages, inflight=inflight,
                       use_graph=False, precision=None, ref_kpts=0, match_kpts=0)
            cfg.update(kw)
            Bq = cfg.pop("B", B)
            from pram_amd import _lib
            L = _lib.load()
            saved_p = L.pram_attention_x3_set_p_split(-1)
            if p_split is not None:
                L.pram_attention_x3_set_p_split(int(p_split))
            saved_ck = L.pram_attention_x3_set_chunk_keys(0)
            if chunk is not None:
                L.pram_attention_x3_set_chunk_keys(int(chunk))
                L.pram_attention_x3_set_split_target(512)          # as --latency
            try:
                j = Job(dev, 0, 1, 0, Bq, **cfg)
                smp = None
                if sustain_s:
                    # a run of >= sustain_s seconds over rotating batches with the socket power / shader clock sampled across it
                    smp = PowerSampler(dev.index if dev.index is not None else 0)
                    for _ in range(warm_):
                        j.step()
                    sync_all()
                    smp.start()
                    t, _ = j.timed(steps_, 0, sync_all, ahead=8, min_seconds=sustain_s)
                    steps_ = j.steps_done
                else:
                    t, _ = j.timed(steps_, warm_, sync_all)
                hit = ops.x3_range_exceeded(dev) if ops.x3_launched(dev) else False
                alt[name] = {"queries_per_s": round(Bq * steps_ / t, 2), "ms_per_step": round(t / steps_ * 1e3, 3), "steps": steps_, "what": note}
                if smp is not None:
                    alt[name]["seconds"] = round(t, 2)
                    alt[name]["distinct_batches"] = len(j.rot)
                    alt[name]["ratio_to_value"] = round((Bq * steps_ / t) / (total_per_step * steps / dt), 4)
                    alt[name]["power"] = smp.stop(trace_bins=10)
                if j.segk:
                    torch.cuda.synchronize()
                    ok_ = (j.sk_matches >= 0)
                    alt[name].update(j.segk_sizes)
                    alt[name]["matches"] = int(ok_.sum())
                    alt[name]["matches_correct"] = int(((j.sk_matches == j.sk_gt) & ok_).sum())
                if hit:
                    alt[name]["x3_range_exceeded"] = True
                if (parity_f16 or parity_q) and not args.no_parity:
                    # the same gate as the headline's, on this configuration's own batch (outside its timed region): the fp32 bars for
                    # the split-fp16 configurations, the fp16 path's own for C5
                    nq_ = max(1, min(Bq, parity_q or 2))
                    qs_ = sorted({int(round(i * (Bq - 1) / max(1, nq_ - 1))) for i in range(nq_)}) if Bq > 1 else [0]
                    with torch.no_grad(), ops.precision_scope(cfg["precision"]):
                        pq = parity_gate(j.pipe, j.sds, cfg["matcher_name"], j.images, j.ref, cfg["kpts"], queries=qs_, f16=parity_f16)
                    alt[name]["parity"] = {k: v for k, v in pq.items() if k != "per_query"}
                    if not pq["ok"]:
                        alt_parity_failed.append(name)
                del j
            except Exception as e:      # an alternative that fails must not take the headline line with it — but it is reported
                alt[name] = {"error": f"{type(e).__name__}: {e}"[:300], "what": note}
            finally:
                L.pram_attention_x3_set_p_split(saved_p)
                L.pram_attention_x3_set_chunk_keys(saved_ck)
                L.pram_attention_x3_set_split_target(split_target)
            torch.cuda.empty_cache()

        alt_run("attention_p_one_fp16", "same step with the soft-max probabilities entering P.V as ONE fp16 (pram_attention_x3_set_p_split(0): two MFMAs per "
                "product instead of three; logits 7e-4..1.3e-3 instead of 4e-5 from the fp32 oracle on flat synthetic attention: NOT inside the 1e-3 parity bar, not parity-gated, not the default)", p_split=0)
        alt_run("sustained", "the default step for >= 10 s: 4 DISTINCT 16-frame batches (own frames, own reference sets) walked round-robin over the 3 lanes, "
                "host at most 8 steps ahead; power.trace = [W, MHz] means of ten equal parts of the run; ratio_to_value = its q/s over the headline's",
                warm_=6, sustain_s=10.0, rotate=4)
        alt_run("adagml", "same step with the AdaGML matcher (BASELINE configs[2] names it; pruning / early exit are data-dependent)", matcher_name="adagml",
                parity_q=4)
        alt_run("act_scale_1", "same step with every model's activation planes at scale 1 instead of 16 (|x| < 65520 instead of 4094.97): where the range "
                "guard leaves a checkpoint with hot activations — the split kernels at the same speed, not the exact-fp32 kernels (alt.exact_f32 is "
                "what a tripped guard cost before round 6)", act_scale=1.0, parity_q=4)
        alt_run("exact_f32", "same step on the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32)", precision="f32")
        alt_run("gml_512x1024", "secondary matcher shape of SURVEY 8(d): the 512 best keypoints of a query against 1024-keypoint reference sets",
                ref_kpts=1024, match_kpts=512)
        alt_run("matcher_segk5", "the matcher's real call pattern (localization/multimap3d.py:112-139): extract + recognise, then per query the 5 most "
                "voted landmarks' keypoint subsets (<= 512) each against its own ragged reference set (600..1376), all 80 pairs of the batch in "
                "ONE grouped produce_matches call (lens0 / lens1)", segk=5)
        alt_run("h2d_in_step", "the default step with the reference's frame preparation INSIDE it (loc_by_rec_online.py:86-106): 16 uint8 frames from "
                "pinned host memory per step (14.7 MB over PCIe on the lane's stream) -> pram_stage_frames_u8 -> the step; `value` itself keeps "
                "the frames resident, as the benchmark contract asks", h2d=True)
        alt_run("latency_b1", "one query per step, one step at a time, captured hipGraph replayed (the reference's online loop), 512-key attention chunks "
                "(= python bench.py --latency); ms_per_step = per-query latency", steps_=30, warm_=10, B=1, inflight=1, use_graph=True, chunk=512)
        alt_run("c4", "BASELINE configs[3] shape (CambridgeLandmarks-like): 4096 keypoints, nc161, 8 queries per step, default (split-fp16) path",
                kpts=4096, n_class=161, B=8, parity_q=4)
        alt_run("c5_f16", "BASELINE configs[4] per-GPU shape (Aachen-like): 4096 keypoints, nc513, 8 queries per step, the 'fp16 MFMA path' "
                "(--precision f16: one fp16 product per MAC — its own tolerance, see parity.bars)", kpts=4096, n_class=513, B=8, precision="f16",
                parity_f16=True, parity_q=2)

    if rank == 0:
        total_q = total_per_step * steps
        dtype = {"f32": "f32", "x3": "f32 results via split-fp16 MFMA (f16 x3 products, f32 accumulate)",
                 "f16": "f16 operands / f32 accumulate: BASELINE C5 'fp16 MFMA path', NOT the fp32 parity configuration"
                        + ("; the matcher's final descriptor projection and score matrix run split-fp16 (x3) on purpose" if x3_ran else "")}[precision]
        rk = args.ref_kpts or args.kpts
        line = {
            "metric": f"query images/sec (640x480, {args.kpts} kpts, {'7Scenes ' if args.n_class == 113 else ''}nc{args.n_class})",
            "value": round(total_q / dt, 3), "unit": "queries/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{'7Scenes' if args.n_class == 113 else 'custom'} full hot path: SFD2 extract+sample -> SegNetViT nc{args.n_class} (15 layers) -> "
                                   f"{args.matcher.upper()} match + 20 Sinkhorn iters, {args.match_kpts or args.kpts} query kpts vs a {rk}-kpt reference set; "
                                   f"stages={args.stages}" + ("; latency mode: one query at a time" if args.latency else ""),
                       "precision": precision,
                       "queries_per_step": total_per_step, "queries_per_gpu_per_step": shard_sizes if job.uneven else B,
                       "batches_in_flight_per_gpu": max(1, inflight), "hipgraph_replay": bool(use_graph), "attention_chunk_keys": chunk_keys,
                       "frame": f"{W_IMG}x{H}", "keypoints": args.kpts,
                       "keypoints_found": counts[:4], "parallelism": f"query-sharded x{world}, one all-gather of result records",
                       "matches_last_step": n_matches, "matches_correct_last_step": n_correct},
            "roofline": roofline,
            "range_guard": {"x3_range_exceeded": bool(range_hit), "split_fp16_kernels_in_step": bool(x3_ran),
                            "policy": "deferred: the status word is read once after the timed region (steps in flight); set = the line is void"},
        }
        line["config"]["gather_order_verified"] = order_ok
        if world > 1:
            line["per_rank_ms_per_step"] = [round(x, 3) for x in rank_ms]
            line["rank_skew_ms_per_step"] = round(max(rank_ms) - min(rank_ms), 3)
            line["per_rank_host_issue_ms_per_step"] = [round(x, 3) for x in rank_issue_ms]
            line["per_rank_gather_ms"] = [round(x, 4) for x in rank_gather_ms]
            line["config"]["host_core_of_rank0"] = core
        if power is not None:
            line["power"] = power
        if parity is not None:
            line["parity"] = parity
        if alt is not None:
            line["alt"] = alt
            line["alt_parity_failed"] = alt_parity_failed
        if world == 1 and args.cpu_queries > 0:
            ref_cpu_sets = None if ref is None else {k: v.cpu() for k, v in ref.items()}
            line["cpu_baseline"] = cpu_baseline(sds, args.matcher, args.cpu_queries, args.kpts, ref_cpu_sets)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if range_hit:
        raise SystemExit("bench.py: the split-fp16 path met an activation beyond its range (range_guard) — the timed steps are void")
    if not order_ok:
        raise SystemExit("bench.py: the gathered record does not hold every rank's records at its own query positions")
    if parity is not None and not parity["ok"]:
        raise SystemExit("bench.py: PARITY GATE FAILED (see the 'parity' object of the JSON line)")


if __name__ == "__main__":
    main()
