"""Feasibility probe: what LayerNorm scale/shift + exact erf GELU cost when they are applied to the A operand of the wide
split-fp16 GEMM while it is staged (PRAM_GEMM_ABLATE=64), next to the stand-alone LayerNorm + GELU pass they would replace.
    python profiles/tools/x3_gelu_staging_probe.py            (run once plain, once with PRAM_GEMM_ABLATE=64)"""
import os, sys
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
for m, k, n in ((65536, 512, 256), (32768, 512, 256), (32768, 1024, 113), (4096, 512, 256), (2048, 512, 256)):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev) * 0.1
    r = torch.randn(m, n, device=dev)
    g, bt = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    t = timeit(lambda: ops.linear(x, w, b, residual=r, precision="x3"))
    tl = timeit(lambda: ops.layernorm_gelu_(x, g, bt))
    print(f"ablate={os.environ.get('PRAM_GEMM_ABLATE','0'):>2} {m:>6} x {k:>4} -> {n:<4}: gemm {t:7.1f} us   stand-alone LayerNorm+GELU {tl:6.1f} us")
