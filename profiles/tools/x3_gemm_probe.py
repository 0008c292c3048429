"""Per-shape timing of the split-fp16 token GEMM in its two input forms — fp32 activations split while staged (linear) vs
activations already split into fp16 planes (linear_planes) — next to the exact-fp32 MFMA kernel.  Warm, 30 launches each.
    python profiles/tools/x3_gemm_probe.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def planes(t):
    s = t * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()


print(f"{'shape':>26} | {'f32 us':>8} {'TF':>6} | {'x3 us':>8} {'TF':>6} | {'x3 planes-in us':>15} {'TF':>6} | {'planes in+out us':>16} | max |d| planes vs x3")
for m, k0, k1, n in ((32768, 256, 0, 768), (32768, 256, 256, 512), (32768, 512, 0, 256), (65536, 256, 0, 768), (65536, 256, 256, 512),
                     (65536, 512, 0, 256), (65536, 256, 0, 512), (32768, 256, 0, 1024), (32768, 1024, 0, 113), (2048, 256, 0, 768)):
    x = torch.randn(m, k0, device=dev)
    x2 = torch.randn(m, k1, device=dev) if k1 else None
    w = torch.randn(n, k0 + k1, device=dev) / (k0 + k1) ** 0.5
    b = torch.randn(n, device=dev) * 0.1
    fl = 2.0 * m * (k0 + k1) * n
    t32 = timeit(lambda: ops.linear(x, w, b, x2=x2, precision="f32"))
    t3 = timeit(lambda: ops.linear(x, w, b, x2=x2, precision="x3"))
    px, px2 = planes(x), (planes(x2) if k1 else None)
    tp = timeit(lambda: ops.linear_planes(px, w, b, x2=px2))
    tpp = timeit(lambda: ops.linear_planes(px, w, b, x2=px2, out="planes"))
    d = float((ops.linear_planes(px, w, b, x2=px2) - ops.linear(x, w, b, x2=x2, precision="x3")).abs().max())
    print(f"{m:>7}x{k0:>4}+{k1:<4}x{n:<5} | {t32:8.1f} {fl / t32 / 1e6:6.0f} | {t3:8.1f} {fl / t3 / 1e6:6.0f} | {tp:15.1f} {fl / tp / 1e6:6.0f} | {tpp:16.1f} | {d:.2e}")
