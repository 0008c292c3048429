"""Per-layer timing of the SFD2 convolution shapes (batch 16, 480x640 frames) on the split-fp16 path.
    python profiles/tools/x3_conv_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 16
for name, h, w_, ci, co, ks, st in (("conv1b", 480, 640, 64, 64, 3, 2), ("conv2a", 240, 320, 64, 128, 3, 1), ("conv2b", 240, 320, 128, 128, 3, 2),
                                    ("conv3a", 120, 160, 128, 256, 3, 1), ("conv3b/Da", 120, 160, 256, 256, 3, 1), ("conv4 1x1", 120, 160, 256, 256, 1, 1),
                                    ("convPa.0", 120, 160, 256, 256, 3, 2), ("convPa.3", 60, 80, 256, 256, 3, 1)):
    x = torch.randn(B, h, w_, ci, device=dev)
    wt = torch.randn(co, ks, ks, ci, device=dev) / (ks * ks * ci) ** 0.5
    bias = torch.randn(co, device=dev) * 0.1
    t = timeit(lambda: ops.conv2d_nhwc(x, wt, bias, None, None, ks=ks, stride=st, relu=True, precision="x3"))
    ho, wo = (h + 2 * (ks // 2) - ks) // st + 1, (w_ + 2 * (ks // 2) - ks) // st + 1
    fl = 2.0 * B * ho * wo * co * ks * ks * ci
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), bias.double(), stride=st, padding=ks // 2).relu()
    got = ops.conv2d_nhwc(x[:1].contiguous(), wt, bias, None, None, ks=ks, stride=st, relu=True, precision="x3").permute(0, 3, 1, 2).double()
    print(f"{name:>10} {h}x{w_} {ci}->{co} k{ks} s{st}: {t:8.1f} us  {fl / t / 1e6:6.0f} TF   max|err| vs fp64 {float((ref - got).abs().max()):.2e}")
