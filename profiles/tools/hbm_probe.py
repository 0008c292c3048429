"""What the HBM of this box sustains for plain streaming (torch device-to-device copy, read-only reduction, fill): the practical
ceiling the HBM-bound kernels (Sinkhorn, LayerNorm+GELU, V^T transpose, conv1a's output) are compared with.
    python profiles/tools/hbm_probe.py"""
import torch

dev = torch.device("cuda:0")


def timeit(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    t_copy = timeit(lambda: y.copy_(x))
    t_read = timeit(lambda: x.sum())
    t_fill = timeit(lambda: y.fill_(1.0))
    gb = mb / 1024.0
    print(f"{mb:5d} MB: copy {2 * gb / t_copy / 1e3:5.2f} TB/s (read + write)   read-only sum {gb / t_read / 1e3:5.2f} TB/s   fill {gb / t_fill / 1e3:5.2f} TB/s")
