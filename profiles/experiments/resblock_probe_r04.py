"""One SFD2 ResBlock at the bench shape (16 frames of 120 x 160 x 256): the fused kernel against the three it replaces.
    python profiles/tools/resblock_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
P = {"w1": (torch.randn(256, 1, 1, 256, generator=g) * 0.08).to(dev), "w2": (torch.randn(256, 3, 3, 8, generator=g) * 0.15).to(dev),
     "w3": (torch.randn(256, 1, 1, 256, generator=g) * 0.08).to(dev)}
for j in (1, 2, 3):
    P[f"s{j}"] = (torch.rand(256, generator=g) * 0.8 + 0.6).to(dev)
    P[f"t{j}"] = (torch.randn(256, generator=g) * 0.2).to(dev)


def three(x):
    y = ops.conv2d_nhwc(x, P["w1"], None, P["s1"], P["t1"], ks=1, relu=True, precision="x3")
    y = ops.conv3x3_grouped_nhwc(y, P["w2"], P["s2"], P["t2"], groups=32, relu=True)
    return ops.conv2d_nhwc(y, P["w3"], None, P["s3"], P["t3"], residual=x, ks=1, relu=True, precision="x3")


def fused(x):
    return ops.resblock_nhwc(x, P["w1"], P["s1"], P["t1"], P["w2"], P["s2"], P["t2"], P["w3"], P["s3"], P["t3"])


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
    x = torch.relu(torch.randn(B, 120, 160, 256, generator=g)).to(dev)
    a, b = three(x), fused(x)
    print(f"B={B:2d} 120x160x256: three kernels {timeit(three, x):8.1f} us | fused {timeit(fused, x):8.1f} us | equal {torch.equal(a, b)} max|d| {float((a - b).abs().max()):.3e}")
