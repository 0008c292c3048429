"""Bit checksum of attention_x3 on full grids (the eight-wave kernel's launch shapes): run under different PRAM_ATTN_* settings
and compare the lines — variants that only move work around must print identical sums.
    python profiles/tools/x3_attn_sum.py"""
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


g = torch.Generator(device="cpu").manual_seed(11)
for B, N, ragged in ((16, 2048, False), (16, 2048, True), (8, 4096, False), (32, 1984, True)):
    q = (torch.randn(B * N, 256, generator=g) * 1.2).to(dev)
    k = (torch.randn(B * N, 256, generator=g) * 1.2).to(dev)
    v = torch.randn(B * N, 256, generator=g).to(dev)
    lens = torch.tensor([N - (37 * i) % 700 for i in range(B)], dtype=torch.int32, device=dev) if ragged else None
    vt = ops.value_planes_t(planes(v), B, 4, N, lens)
    o, lse = ops.attention_x3(planes(q), planes(k), vt, B, 4, N, N, 0.125, lens, lens, want_lse=True)
    o = o.view(B, N, 256)
    tot = 0
    for b in range(B):
        n = int(lens[b]) if ragged else N
        ob = o[b, :n]
        assert torch.isfinite(ob).all()
        tot += int(ob.contiguous().view(torch.int32).to(torch.int64).sum().item()) + int(lse[b, :, :n].contiguous().view(torch.int32).to(torch.int64).sum().item())
    print(f"B={B} N={N} ragged={ragged}: {tot}")
