import sys, torch
sys.path.insert(0, "/root/repo")
from pram_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bs = torch.tensor(1.0, device=dev)
for B in (1, 2, 4, 8, 16):
    d = torch.randn(B, 2048, 2048, device=dev)
    t = timeit(lambda: ops.sinkhorn_match(d, bs, 20, 0.2))
    print(f"B={B}: {t:8.1f} us total, {t / B:7.1f} us per pair")
