"""Timing of attention_x3 at the bench's launch shapes (warm), optionally under PRAM_ATTN_ABLATE / PRAM_ATTN_MODE (see
attention_x3.hip: MODE 0 = the unchunked kernel, 1 = fused (key chunks folded in registers), 2 = split).
    python profiles/tools/x3_attn_probe.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def planes(t):
    s = t * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()


SHAPES = ((16, 2048), (32, 2048), (8, 4096), (1, 2048), (2, 2048))
if os.environ.get("PRAM_PROBE_SHAPES"):      # e.g. "4x2048,8x2048,12x2048": a batch sweep
    SHAPES = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["PRAM_PROBE_SHAPES"].split(","))
if os.environ.get("PRAM_PROBE_SPLIT_TARGET"):      # workgroups a split launch aims at (bench.py --latency: 512)
    ops._lib.load().pram_attention_x3_set_split_target(int(os.environ["PRAM_PROBE_SPLIT_TARGET"]))
for B, N in SHAPES:
    q = torch.randn(B * N, 256, device=dev)
    k = torch.randn(B * N, 256, device=dev)
    v = torch.randn(B * N, 256, device=dev)
    pq, pk = planes(q), planes(k)
    vt = ops.value_planes_t(planes(v), B, 4, N)
    f = lambda: ops.attention_x3(pq, pk, vt, B, 4, N, N, 0.125)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * 4 * N * N * 64
    per_product = ops._lib.load().pram_attention_x3_mfma_per_tile(N) / 16.0      # 3 (two-part probabilities, the default) or 2.5
    print(f"mode={os.environ.get('PRAM_ATTN_MODE', 'auto')} ablate={os.environ.get('PRAM_ATTN_ABLATE', '0')} B={B:3d} N={N}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s algorithmic ({fl * per_product / us / 1e6:7.1f} executed, {per_product:g} MFMAs per product)")
