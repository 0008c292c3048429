"""Grouped 3x3 of the ResBlock at the bench's shape (16 x 120 x 160 x 256): the exact-fp32 vector kernel, the matrix-pipe kernel
on split planes, and the 1x1 that feeds it with fp32 / planes output.  PYTHONPATH=. python profiles/tools/gconv_probe.py"""
import torch
from pram_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(16, 120, 160, 256, device=dev)
w = torch.randn(256, 3, 3, 8, device=dev) * 0.1
w1 = torch.randn(256, 1, 1, 256, device=dev) * 0.05
s = torch.rand(256, device=dev) + 0.5
t = torch.randn(256, device=dev) * 0.1
x16 = x * 16.0
hi = x16.half()
lo = (x16 - hi.float()).half()


def timed(name, fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us / launch")


timed("grouped 3x3, vector ALU (fp32 in)", lambda: ops.conv3x3_grouped_nhwc(x, w, s, t, 32, True))
timed("grouped 3x3, matrix pipe (planes in)", lambda: ops.conv3x3_grouped_planes(hi, lo, w, s, t, 32, True))
timed("1x1 256 -> 256, fp32 out", lambda: ops.conv2d_nhwc(x, w1, None, s, t, ks=1, relu=True, precision="x3"))
timed("1x1 256 -> 256, planes out", lambda: ops.conv2d_nhwc_planes(x, w1, None, s, t, ks=1, relu=True))
w3 = torch.randn(256, 1, 1, 256, device=dev) * 0.05
res = torch.randn(16, 120, 160, 256, device=dev)
y2 = torch.randn(16, 120, 160, 256, device=dev)
timed("1x1 256 -> 256 + residual, fp32 out", lambda: ops.conv2d_nhwc(y2, w3, None, s, t, residual=res, ks=1, relu=True, precision="x3"))
