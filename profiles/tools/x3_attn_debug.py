import sys, torch
sys.path.insert(0, '.')
from pram_amd import ops
dev = torch.device("cuda:0")
def planes(t):
    s = t * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()
def run(q, k, v, B, N):
    vt = ops.value_planes_t(planes(v), B, 4, N)
    return ops.attention_x3(planes(q), planes(k), vt, B, 4, N, N, 0.125)
def ref(q, k, v, B, N):
    q, k, v = (t.double().view(B, N, 4, 64).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
    return (a @ v).transpose(1, 2).reshape(B * N, 256)
torch.manual_seed(0)
for B, N in ((1, 64), (1, 128), (1, 256), (2, 2048), (16, 2048)):
    for name, mk in (("q=0", lambda: (torch.zeros(B*N,256), torch.randn(B*N,256), torch.randn(B*N,256))),
                     ("v=1", lambda: (torch.randn(B*N,256), torch.randn(B*N,256), torch.ones(B*N,256))),
                     ("rand", lambda: (torch.randn(B*N,256), torch.randn(B*N,256), torch.randn(B*N,256)))):
        q, k, v = (t.to(dev) for t in mk())
        o = run(q, k, v, B, N)
        r = ref(q, k, v, B, N)
        e = (o.double() - r).abs()
        bad = (e > 1e-4)
        print(f"B={B} N={N} {name}: max err {e.max().item():.3e}; bad rows {int(bad.any(1).sum())}/{B*N}; bad cols {sorted(set((bad.any(0).nonzero().flatten() % 64).tolist()))[:20]}; first bad rows {bad.any(1).nonzero().flatten()[:12].tolist()}")
