"""SFD2's conv1a -> conv1b at the bench shape (16 / 1 frames of 480 x 640): the fused kernel against the two it replaces.
    python profiles/tools/conv1_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
wa = (torch.randn(64, 3, 3, 4, generator=g) * 0.25).to(dev)
wb = (torch.randn(64, 3, 3, 64, generator=g) * 0.06).to(dev)
v = {k: ((torch.rand(64, generator=g) * 0.8 + 0.6) if k[0] == "s" else torch.randn(64, generator=g) * 0.2).to(dev) for k in ("ba", "sa", "ta", "bb", "sb", "tb")}


def two(x4):
    y = ops.conv2d_nhwc(x4, wa, v["ba"], v["sa"], v["ta"], ks=3, stride=1, relu=True)
    return ops.conv2d_nhwc(y, wb, v["bb"], v["sb"], v["tb"], ks=3, stride=2, relu=True, precision="x3")


def fused(x4):
    return ops.sfd2_conv1(x4, wa, v["ba"], v["sa"], v["ta"], wb, v["bb"], v["sb"], v["tb"])


def timeit(f, x, n=20):
    for _ in range(5):
        f(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (16, 8, 1):
    x4 = ops.image_to_nhwc4(torch.randn(B, 3, 480, 640, generator=g).to(dev))
    a, b = two(x4), fused(x4)
    print(f"B={B:2d} 480x640: two kernels {timeit(two, x4):8.1f} us | fused {timeit(fused, x4):8.1f} us | max|d| {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3g}")
