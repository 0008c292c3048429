"""Per-shape timing of the three MFMA kernel families at the shapes of the hot path (and a long-K
control that separates steady-state main-loop efficiency from prologue/epilogue overhead).

    python profiles/kernel_sweep.py            # on a GPU box; prints one line per shape

Not part of the product or the tests: a measurement helper whose output is kept under profiles/.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pram_amd import ops


def timeit(fn, reps=20, warm=40, rounds=3):
    """min over `rounds` of the mean of `reps` launches, after `warm` untimed ones (the first tens of launches on
    fresh allocations run 10-15 % slower: clocks / TLB / infinity-cache warm-up)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    print("== linear (M, N, K)")
    for (m, n, k) in [(32768, 768, 256), (32768, 512, 512), (32768, 256, 512), (65536, 768, 256), (65536, 512, 256),
                      (65536, 512, 512), (65536, 256, 512), (65536, 512, 4096), (8192, 512, 4096), (4096, 256, 512)]:
        a, w, b = rnd(m, k), rnd(n, k), rnd(n)
        t = timeit(lambda: ops.linear(a, w, b))
        print(f"linear m={m} n={n} k={k}: {t*1e6:8.1f} us  {2.0*m*n*k/t/1e12:6.1f} TFLOP/s")
    print("== conv (B, H, W, Cin, Cout, ks, stride)")
    B = 16
    for (h, w_, ci, co, ks, st) in [(480, 640, 4, 64, 3, 1), (480, 640, 64, 64, 3, 2), (240, 320, 64, 128, 3, 1),
                                     (240, 320, 128, 128, 3, 2), (120, 160, 128, 256, 3, 1), (120, 160, 256, 256, 3, 1),
                                     (120, 160, 256, 256, 1, 1), (120, 160, 256, 256, 3, 2), (60, 80, 256, 256, 3, 1),
                                     (120, 160, 256, 128, 1, 1), (60, 80, 256, 65, 1, 1)]:
        x = rnd(B, h, w_, ci)
        wt = rnd(co, ks, ks, ci)
        bias = rnd(co)
        t = timeit(lambda: ops.conv2d_nhwc(x, wt, bias, None, None, ks=ks, stride=st, relu=True), reps=10)
        ho, wo = (h + 2 * (ks // 2) - ks) // st + 1, (w_ + 2 * (ks // 2) - ks) // st + 1
        fl = 2.0 * B * ho * wo * co * ks * ks * ci
        print(f"conv {h}x{w_} {ci}->{co} k{ks} s{st}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s")
    print("== attention (S sequences, T tokens)")
    for (S, T) in [(16, 2048), (32, 2048), (8, 4096), (1, 2048)]:
        q, k, v = rnd(S * T, 256), rnd(S * T, 256), rnd(S * T, 256)
        t = timeit(lambda: ops.attention(q, k, v, S, 4, T, T, 0.125, None, None))
        fl = 4.0 * S * 4 * T * T * 64
        print(f"attention S={S} T={T}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
