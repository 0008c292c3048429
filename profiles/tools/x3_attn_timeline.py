"""Per-workgroup timeline of one attention_x3 launch (a -DPRAM_PROFILING build of attention_x3.hip: PRAM_HIP_LIB=.../libpram_hip_prof.so):
when every workgroup started and ended (100 MHz wall clock), on which XCD / CU, and how many shader clocks it took.
    python profiles/tools/x3_attn_timeline.py [B N]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from pram_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 2048)
zero = len(sys.argv) > 3 and sys.argv[3] == "zeros"


def planes(t):
    s = t * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()


q, k, v = (torch.zeros(B * N, 256, device=dev) if zero else torch.randn(B * N, 256, device=dev) for _ in range(3))
pq, pk = planes(q), planes(k)
vt = ops.value_planes_t(planes(v), B, 4, N)
f = lambda: ops.attention_x3(pq, pk, vt, B, 4, N, N, 0.125)
for _ in range(10):
    f()
torch.cuda.synchronize()
L = _lib.load()
ph = np.zeros((8, 4), dtype=np.uint64)
fp = L.pram_debug_attention_phases
fp.restype, fp.argtypes = C.c_int, [C.c_void_p, C.c_int]
assert fp(ph.ctypes.data, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} {'zeros' if zero else 'randn'}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
assert fp(ph.ctypes.data, 0) == 0
tiles = 10 * B * 4 * (N // 256) * (N // 64 - 1)      # tile passes of mid() per wave slot
print("shader clocks per tile and wave: scores + soft-max | P V | LDS writes (+ load wait) | barrier | sum")
for w in range(8):
    v = ph[w].astype(np.float64) / tiles
    print(f"  wave {w}: {v[0]:7.0f} | {v[1]:7.0f} | {v[2]:7.0f} | {v[3]:7.0f} | {v.sum():7.0f}")
fn = L.pram_debug_attention_timeline
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
buf = np.zeros((4096, 4), dtype=np.uint64)
assert fn(buf.ctypes.data) == 0
nwg = B * 4 * (N // 256)
t = buf[:nwg].astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # us
dur = en - st
xcc = (t[:, 2] >> 32) & 0xf
hw = t[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
clk = t[:, 3] / np.maximum(dur, 1e-9) / 1e3      # GHz
print(f"workgroups {nwg}: span {en.max():.1f} us; duration min / median / max {dur.min():.1f} / {np.median(dur):.1f} / {dur.max():.1f} us; shader clock {np.median(clk):.2f} GHz (min {clk.min():.2f}, max {clk.max():.2f})")
first = st < 5.0
print(f"first round: {first.sum()} workgroups start < 5 us, their durations {dur[first].min():.1f} / {np.median(dur[first]):.1f} / {dur[first].max():.1f}; later ones {dur[~first].min() if (~first).any() else 0:.1f} / {np.median(dur[~first]) if (~first).any() else 0:.1f} / {dur[~first].max() if (~first).any() else 0:.1f}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: {m.sum():4d} workgroups, duration median {np.median(dur[m]):6.1f} us (min {dur[m].min():6.1f}, max {dur[m].max():6.1f}), last end {en[m].max():6.1f} us, clock {np.median(clk[m]):.2f} GHz")
hist, edges = np.histogram(en, bins=12)
print("end-time histogram (us):", ", ".join(f"{edges[i]:.0f}-{edges[i+1]:.0f}: {hist[i]}" for i in range(len(hist))))
