"""GPU: seeded random shapes through the split-fp16 kernels (the default path) against fp64 — the tile choice (narrow / 128 x 256 /
256 x 256), ragged edges in every dimension, concatenated inputs, split-plane outputs, ragged sequences, cross-attention shifts.
The hand-picked shapes of test_gpu_x3.py pin the error bars; this file hunts for shape-dependent addressing bugs."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from pram_amd import weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def err(a, ref64):
    return float((a.double().cpu() - ref64).abs().max())


def _planes(t2d):
    s = t2d * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()


@pytest.mark.parametrize("seed", range(6))
def test_linear_x3_random_shapes(dev, seed):
    from pram_amd import ops
    rng = random.Random(1000 + seed)
    for case in range(6):
        m = rng.choice([1, 7, 64, 129, 300, 1000, 2049, 4096, 8200, 20000])
        k0 = 32 * rng.randint(1, 16)
        k1 = rng.choice([0, 0, 32 * rng.randint(1, 8)])
        n = rng.choice([8, 24, 64, 113, 128, 256, 264, 512, 768, 1000])
        x = W.normal(seed, f"fz/x{case}", (m, k0))
        x2 = W.normal(seed, f"fz/y{case}", (m, k1)) if k1 else None
        w = W.normal(seed, f"fz/w{case}", (n, k0 + k1), 1.0 / math.sqrt(k0 + k1))
        b = W.normal(seed, f"fz/b{case}", (n,), 0.1)
        res = W.normal(seed, f"fz/r{case}", (m, n)) if rng.random() < 0.5 else None
        xa = torch.cat([x, x2], -1) if k1 else x
        ref = F.linear(xa.double(), w.double(), b.double())
        if res is not None:
            ref = ref + res.double()
        kw = dict(x2=None if x2 is None else x2.to(dev), residual=None if res is None else res.to(dev), precision="x3")
        got = ops.linear(x.to(dev), w.to(dev), b.to(dev), **kw)
        e = err(got, ref)
        assert e < 6e-6, (m, k0, k1, n, e)
        if res is None and n % 8 == 0:
            # split-plane output (hi + lo = 16 x the fp32 result, to the rounding of the two fp16 planes)
            _, pl = ops.linear(x.to(dev), w.to(dev), b.to(dev), x2=kw["x2"], precision="x3", split_out="only")
            back = (pl[0].float() + pl[1].float()) / 16.0
            assert float((back - got).abs().max()) <= 2e-6 * max(1.0, float(got.abs().max())), (m, k0, k1, n)


@pytest.mark.parametrize("seed", range(4))
def test_ragged_linear_x3_random(dev, seed):
    from pram_amd import ops
    rng = random.Random(2000 + seed)
    S = rng.randint(1, 6)
    T = rng.choice([64, 200, 300, 512, 1030])
    n = rng.choice([64, 256, 512])
    x = W.normal(seed, "fzr/x", (S * T, 256)).to(dev)
    w = W.normal(seed, "fzr/w", (n, 256), 1.0 / 16).to(dev)
    lens_l = [rng.choice([0, 1, T // 3, T - 1, T]) for _ in range(S)]
    lens = torch.tensor(lens_l, dtype=torch.int32, device=dev)
    full = ops.linear(x, w, None, precision="x3")
    out = torch.full((S * T, n), 3.0, device=dev)
    ops.linear(x, w, None, precision="x3", out=out, lens=lens, t_pad=T)
    o3, f3 = out.view(S, T, n), full.view(S, T, n)
    for s_, ln in enumerate(lens_l):
        assert torch.equal(o3[s_, :ln], f3[s_, :ln]) and bool((o3[s_, ln:] == 3.0).all())


@pytest.mark.parametrize("seed", range(5))
def test_attention_x3_random_shapes(dev, seed):
    from pram_amd import ops
    rng = random.Random(3000 + seed)
    B = rng.randint(1, 4)
    M = rng.choice([1, 33, 64, 130, 500, 1024, 1500])
    N = rng.choice([1, 40, 64, 65, 300, 1024, 1100, 2048])
    Hh = 4
    q = W.normal(seed, "fza/q", (B, M, 256), 1.1)
    k = W.normal(seed, "fza/k", (B, N, 256), 1.1)
    v = W.normal(seed, "fza/v", (B, N, 256))
    qlens = [rng.randint(1, M) for _ in range(B)]
    klens = [rng.randint(1, N) for _ in range(B)]
    sp = lambda t, L: t.view(B, L, Hh, 64).permute(0, 2, 1, 3).double()
    s = torch.einsum("bhid,bhjd->bhij", sp(q, M), sp(k, N)) * 0.125
    mask = torch.arange(N)[None, :] >= torch.tensor(klens)[:, None]
    pr = torch.softmax(s.masked_fill(mask[:, None, None, :], float("-inf")), -1)
    ref = torch.einsum("bhij,bhjd->bhid", pr, sp(v, N)).permute(0, 2, 1, 3).reshape(B, M, 256)
    qd, kd, vd = q.view(B * M, -1).to(dev), k.view(B * N, -1).to(dev), v.view(B * N, -1).to(dev)
    ql = torch.tensor(qlens, dtype=torch.int32, device=dev)
    kl = torch.tensor(klens, dtype=torch.int32, device=dev)
    pq, pk = _planes(qd), _planes(kd)
    out, lse = ops.attention_x3(pq, pk, ops.value_planes_t(_planes(vd), B, Hh, N, kl), B, Hh, M, N, 0.125, ql, kl, want_lse=True)
    out = out.view(B, M, 256)
    col = ops.attention_colmean_x3(pq, pk, lse, B, Hh, M, N, 0.125, ql, kl)
    for b in range(B):
        e = err(out[b, :qlens[b]], ref[b, :qlens[b]])
        assert e < (1e-4 if N >= 1024 else 5e-6), (B, M, N, qlens, klens, e)
        cref = pr[b, :, :qlens[b], :klens[b]].mean(dim=(0, 1))
        assert err(col[b, :klens[b]], cref) < 2e-5 * float(cref.max()) + 1e-9, (B, M, N, qlens, klens)


@pytest.mark.parametrize("seed", range(4))
def test_conv_x3_random_shapes(dev, seed):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    rng = random.Random(4000 + seed)
    for case in range(3):
        cin = rng.choice([32, 64, 128, 256])
        cout = rng.choice([16, 64, 65, 128, 256])
        ks = rng.choice([1, 3])
        stride = rng.choice([1, 2])
        bsz, h, w = rng.randint(1, 3), rng.choice([9, 24, 40, 61]), rng.choice([8, 33, 64, 80])
        x = W.normal(seed, f"fzc/x{case}", (bsz, cin, h, w))
        wt = W.normal(seed, f"fzc/w{case}", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
        b = W.normal(seed, f"fzc/b{case}", (cout,), 0.1)
        ref = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=ks // 2)).permute(0, 2, 3, 1)
        got = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ResNet4x._ohwi(wt).to(dev), b.to(dev), None, None, ks=ks, stride=stride,
                              relu=True, precision="x3")
        assert tuple(got.shape) == tuple(ref.shape) and err(got, ref) < 1e-5, (cin, cout, ks, stride, bsz, h, w)
