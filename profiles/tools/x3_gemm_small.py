"""x3 token GEMMs at ONE query's shapes (2048 rows: 64-row tiles, 64-192 workgroups): time per launch back to back and a bit checksum.
    python profiles/tools/x3_gemm_small.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
out = []
for m, k0, k1, n in ((2048, 256, 0, 768), (2048, 256, 0, 256), (2048, 256, 256, 512), (2048, 512, 0, 256), (4096, 256, 0, 768), (1024, 256, 0, 256)):
    x = torch.randn(m, k0, generator=g).to(dev)
    x2 = torch.randn(m, k1, generator=g).to(dev) if k1 else None
    w = (torch.randn(n, k0 + k1, generator=g) / (k0 + k1) ** 0.5).to(dev)
    b = (torch.randn(n, generator=g) * 0.1).to(dev)
    f = lambda: ops.linear(x, w, b, x2=x2, precision="x3")
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = f()
    e1.record()
    torch.cuda.synchronize()
    cs = int(y.view(torch.int32).to(torch.int64).sum().item())
    out.append(f"{m}x{k0}+{k1}x{n}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us #{cs & 0xffffff:06x}")
print(" | ".join(out))
