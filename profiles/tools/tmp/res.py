import sys, torch
sys.path.insert(0, "/root/repo")
from pram_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in (32768, 65536):
    x = torch.randn(m, 512, device=dev); w = torch.randn(256, 512, device=dev) / 22; b = torch.randn(256, device=dev); r = torch.randn(m, 256, device=dev)
    out = torch.empty(m, 256, device=dev)
    t0 = timeit(lambda: ops.linear(x, w, b, precision="x3", out=out))
    t1 = timeit(lambda: ops.linear(x, w, b, residual=r, precision="x3", out=out))
    t2 = timeit(lambda: ops.linear(x, w, b, precision="x3", out=out))
    t3 = timeit(lambda: ops.linear(x, w, b, residual=r, precision="x3", out=out))
    print(f"M={m} K=512 N=256: no residual {t0:.1f} / {t2:.1f} us, with residual {t1:.1f} / {t3:.1f} us")
