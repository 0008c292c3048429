"""Timing of the MLP tail (Linear -> LayerNorm -> GELU -> Linear + residual) at the bench's shapes: the GEMM pair of ops.mlp_tail
(centred first GEMM + row sums of squares; LayerNorm + GELU in the second GEMM's staging) next to the three-kernel form.
    python profiles/tools/x3_mlp_tail_probe.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in (65536, 32768, 4096, 2048):
    x, ctx = torch.randn(m, 256, device=dev), torch.randn(m, 256, device=dev)
    w0 = torch.randn(512, 512, device=dev) / 512 ** 0.5; b0 = torch.randn(512, device=dev) * 0.1
    g, bt = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    w3 = torch.randn(256, 512, device=dev) / 512 ** 0.5; b3 = torch.randn(256, device=dev) * 0.1
    w0c, b0c = [t.to(dev) for t in ops.center_linear(w0, b0)]
    t_pair = timeit(lambda: ops.mlp_tail(x, w0c, b0c, g, bt, w3, b3, x2=ctx, residual=x))
    t0 = timeit(lambda: ops.linear(x, w0, b0, x2=ctx, precision="x3"))
    h = ops.linear(x, w0, b0, x2=ctx, precision="x3")
    tl = timeit(lambda: ops.layernorm_gelu_(h, g, bt))
    t3 = timeit(lambda: ops.linear(h, w3, b3, residual=x, precision="x3"))
    print(f"rows {m:>6}: GEMM pair {t_pair:7.1f} us | three kernels {t0 + tl + t3:7.1f} us = mlp.0 {t0:6.1f} + LayerNorm+GELU {tl:5.1f} + mlp.3 {t3:6.1f}")

# the two kernels of the pair on their own
L = ops._lib.load()
for m in (65536, 32768):
    x, ctx = torch.randn(m, 256, device=dev), torch.randn(m, 256, device=dev)
    w0 = torch.randn(512, 512, device=dev) / 512 ** 0.5; b0 = torch.randn(512, device=dev) * 0.1
    g, bt = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    w3 = torch.randn(256, 512, device=dev) / 512 ** 0.5; b3 = torch.randn(256, device=dev) * 0.1
    h = torch.empty(m, 512, device=dev); ssq = torch.empty(8, m, device=dev); out = torch.empty(m, 256, device=dev)
    wh, wl, ws = ops.split_weight(w0); w3h, w3l, w3s = ops.split_weight(w3)
    p = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    f1 = lambda: L.pram_linear_x3_ssq_f32(p(x), 256, 256, p(ctx), 256, 256, p(wh), p(wl), ws, p(b0), p(h), 512, p(ssq), m, 512, None, 0, st)
    f2 = lambda: L.pram_linear_x3_lngelu_f32(p(h), 512, 512, p(w3h), p(w3l), w3s, p(b3), p(x), 256, p(out), 256, m, 256, p(ssq), 8, p(g), p(bt), 1e-5, None, 0, st)
    f1(); f2()
    print(f"rows {m:>6}: first GEMM + row sums {timeit(f1):7.1f} us, second GEMM with LayerNorm+GELU staging {timeit(f2):7.1f} us")
