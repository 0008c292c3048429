"""x3 token GEMM at the step's shapes: time and a bit checksum of the output (variants that only move work must print the same sums).
    python profiles/tools/x3_gemm_quick.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
out = []
for m, k0, k1, n in ((32768, 256, 0, 768), (32768, 256, 256, 512), (32768, 512, 0, 256), (65536, 256, 0, 768), (65536, 256, 256, 512), (65536, 512, 0, 256)):
    x = torch.randn(m, k0, generator=g).to(dev)
    x2 = torch.randn(m, k1, generator=g).to(dev) if k1 else None
    w = (torch.randn(n, k0 + k1, generator=g) / (k0 + k1) ** 0.5).to(dev)
    b = (torch.randn(n, generator=g) * 0.1).to(dev)
    if os.environ.get("PRAM_PROBE_ZERO"):      # the same instruction stream on zero operands: what the power cap costs
        x.zero_(); w.zero_(); b.zero_()
        if x2 is not None:
            x2.zero_()
    f = lambda: ops.linear(x, w, b, x2=x2, precision="x3")
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        y = f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    cs = int(y.view(torch.int32).to(torch.int64).sum().item())
    out.append(f"{m}x{k0}+{k1}x{n}: {us:.1f} us {2.0 * m * (k0 + k1) * n / us / 1e6:.0f} TF #{cs & 0xffffff:06x}")
print(" | ".join(out))
# round 6: the two GEMMs of an MLP tail at the step's shapes (first: fp32 out + row statistics; second: LayerNorm + GELU staging, + residual),
# and a plain fp32 GEMM with a residual
out = []
for m in (32768, 65536):
    x = torch.randn(m, 256, generator=g).to(dev)
    c = torch.randn(m, 256, generator=g).to(dev)
    w0, b0 = (t.to(dev) for t in ops.center_linear((torch.randn(512, 512, generator=g) / 512 ** 0.5).to(dev), (torch.randn(512, generator=g) * 0.1).to(dev)))
    gam, bet = (1 + 0.1 * torch.randn(512, generator=g)).to(dev), (0.1 * torch.randn(512, generator=g)).to(dev)
    w3, b3 = (torch.randn(256, 512, generator=g) / 512 ** 0.5).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev)
    w3a = w3[:, :256].contiguous()
    fs = {"tail": lambda: ops.mlp_tail(x, w0, b0, gam, bet, w3, b3, x2=c, residual=x),
          "res": lambda: ops.linear(x, w3a, b3, residual=x, precision="x3")}
    for name, f in fs.items():
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            y = f()
        e1.record()
        torch.cuda.synchronize()
        cs = int(y.view(torch.int32).to(torch.int64).sum().item())
        out.append(f"{name} {m}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us #{cs & 0xffffff:06x}")
print(" | ".join(out))
