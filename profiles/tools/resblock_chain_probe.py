"""One SFD2 ResBlock as the default path runs it (1x1 -> planes, grouped 3x3 on the matrix pipe, 1x1 + residual) at 120 x 160 x 256,
per frame, against the number of frames per launch: does a chain whose tensors fit the 256 MB memory-side cache run faster per frame?
PYTHONPATH=. python profiles/tools/resblock_chain_probe.py"""
import torch
from pram_amd import ops

dev = torch.device("cuda:0")
w1 = torch.randn(256, 1, 1, 256, device=dev) * 0.05
w2 = torch.randn(256, 3, 3, 8, device=dev) * 0.1
w3 = torch.randn(256, 1, 1, 256, device=dev) * 0.05
s = torch.rand(256, device=dev) + 0.5
t = torch.randn(256, device=dev) * 0.1


def block(x):
    yh, yl = ops.conv2d_nhwc_planes(x, w1, None, s, t, ks=1, relu=True)
    y = ops.conv3x3_grouped_planes(yh, yl, w2, s, t, 32, True)
    return ops.conv2d_nhwc(y, w3, None, s, t, residual=x, ks=1, relu=True, precision="x3")


total = 16
xs_all = torch.randn(total, 120, 160, 256, device=dev)
for B in (1, 2, 4, 8, 16):
    chunks = [xs_all[i:i + B] for i in range(0, total, B)]
    g = torch.cuda.CUDAGraph()
    for c in chunks:
        o = block(c)
        o = block(o)
        o = block(o)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for c in chunks:
            o = block(c)
            o = block(o)
            o = block(o)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{total} frames through three ResBlocks in chunks of {B:2d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
