import sys, torch
sys.path.insert(0, "/root/repo")
from pram_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, N in ((32, 2048), (16, 4096), (16, 2048), (1, 2048)):
    q = torch.randn(B * N, 256, device=dev).half(); k = torch.randn(B * N, 256, device=dev).half(); v = torch.randn(B * N, 256, device=dev).half()
    vt = ops.value_t16(v, B, 4, N)
    fl = 4.0 * B * 4 * N * N * 64
    t_old = timeit(lambda: ops.attention_h16(q, k, v, B, 4, N, N, 0.125))
    t_new = timeit(lambda: ops.attention_h16t(q, k, vt, B, 4, N, N, 0.125))
    t_vt = timeit(lambda: ops.value_t16(v, B, 4, N))
    print(f"B={B:3d} N={N}: attention_h16 {t_old:8.1f} us {fl / t_old / 1e6:7.1f} TF | attention_h16t {t_new:8.1f} us {fl / t_new / 1e6:7.1f} TF (+ transpose {t_vt:.1f} us)")
