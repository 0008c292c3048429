"""Where a workgroup of the wide split-fp16 GEMM spends its cycles: PRAM_GEMM_ABLATE=4 makes wave 0 of every workgroup add
shader-clock deltas per main-loop phase to device counters (gemm_core_x3w.h); this prints them per shape as a share of the
workgroup's lifetime.      PRAM_GEMM_ABLATE=4 python profiles/tools/x3_gemm_phases.py [w256|w128]"""
import ctypes
import os
import sys
from pathlib import Path

os.environ.setdefault("PRAM_GEMM_ABLATE", "4")
if len(sys.argv) > 1:
    os.environ["PRAM_X3_TILE"] = sys.argv[1]
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
buf = (ctypes.c_ulonglong * 72)()


def phases(reset=True):
    torch.cuda.synchronize()
    _lib.check(L.pram_debug_gemm_phases(buf, int(reset)), "pram_debug_gemm_phases")
    return [int(v) for v in buf]


print(f"{'shape':>24} | {'us':>7} | wg | cyc/wg loop | issue+mfma   wait   commit  barrier | epilogue | per chunk: mfma / wait / commit / barrier (cycles)")
for m, k0, k1, n in ((32768, 256, 0, 768), (32768, 256, 256, 512), (65536, 256, 0, 768), (65536, 256, 256, 512), (65536, 512, 0, 256),
                     (65536, 256, 0, 512), (65536, 1024, 0, 1024)):
    x = torch.randn(m, k0, device=dev)
    x2 = torch.randn(m, k1, device=dev) if k1 else None
    w = torch.randn(n, k0 + k1, device=dev) / (k0 + k1) ** 0.5
    b = torch.randn(n, device=dev) * 0.1
    for _ in range(5):
        ops.linear(x, w, b, x2=x2, precision="x3")
    phases()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        ops.linear(x, w, b, x2=x2, precision="x3")
    e1.record()
    p = phases()
    us = e0.elapsed_time(e1) / reps * 1e3
    wg = p[5]
    if wg == 0:
        print(f"{m:>7}x{k0:>4}+{k1:<4}x{n:<5} | {us:7.1f} | (narrow tile: not instrumented)")
        continue
    loop, ep = p[4] / wg, p[6] / wg
    nk = (k0 + k1) // 32
    sh = [p[i] / wg for i in range(4)]
    tot = loop + ep
    print(f"{m:>7}x{k0:>4}+{k1:<4}x{n:<5} | {us:7.1f} | {wg // reps:>4} | {loop:9.0f} | " + "  ".join(f"{100 * v / tot:5.1f}%" for v in sh) +
          f" | {100 * ep / tot:5.1f}% | " + " / ".join(f"{v / nk:.0f}" for v in sh))
    if (m, k0, k1, n) == (65536, 512, 0, 256):
        base = min(p[8 + 8 * w_] for w_ in range(8))
        print("      wave: loop top | loads issued | k-step 0 issued | k-step 1 issued | loads landed | commit done | past barrier   (cycles from the first wave's loop top, workgroup 0, chunk 3)")
        for w_ in range(8):
            print(f"      {w_}: " + " ".join(f"{p[8 + 8 * w_ + i] - base:6d}" for i in range(7)))
