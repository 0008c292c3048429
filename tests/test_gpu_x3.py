"""GPU (MI355X): the split-fp16 ("x3") MFMA path — three fp16 MFMAs per fp32-class product — kernel by kernel against
fp64 references, next to the exact-fp32 MFMA kernels on the same inputs.  The bar is "fp32 class": the x3 error must
stay within a small factor of the f32-MFMA kernel's own rounding error (and far below single-product fp16)."""
import math

import pytest
import torch
import torch.nn.functional as F

from pram_amd import weights as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rnd(seed, name, shape, std=1.0):
    return W.normal(seed, name, shape, std)


def err(a, ref64):
    return float((a.double().cpu() - ref64).abs().max())


@pytest.mark.parametrize("m,k0,k1,n", [(300, 256, 0, 768), (1000, 256, 256, 512), (77, 1024, 0, 113), (128, 128, 0, 256),
                                       (513, 512, 0, 256), (4096, 512, 0, 512), (65, 32, 0, 64)])
def test_linear_x3_is_fp32_class(dev, m, k0, k1, n):
    from pram_amd import ops
    x = rnd(1, "lin/x", (m, k0))
    x2 = rnd(1, "lin/x2", (m, k1)) if k1 else None
    w = rnd(1, "lin/w", (n, k0 + k1), 1.0 / math.sqrt(k0 + k1))
    b = rnd(1, "lin/b", (n,), 0.1)
    res = rnd(1, "lin/r", (m, n))
    xa = torch.cat([x, x2], -1) if k1 else x
    ref = F.linear(xa.double(), w.double(), b.double()) * 0.25 + res.double()
    args = (x.to(dev), w.to(dev), b.to(dev))
    kw = dict(x2=None if x2 is None else x2.to(dev), residual=res.to(dev), alpha=0.25)
    e32 = err(ops.linear(*args, precision="f32", **kw), ref)
    ex3 = err(ops.linear(*args, precision="x3", **kw), ref)
    e16 = err(ops.linear(*args, precision="f16", **kw), ref) if (k0 + k1) % 64 == 0 and k0 % 64 == 0 else float("nan")
    print(f"linear {m}x{k0}+{k1}x{n}: |err| f32 {e32:.2e}  x3 {ex3:.2e}  f16 {e16:.2e}")
    assert ex3 < 4e-6 and ex3 < 4 * e32 + 1e-6


def test_linear_x3_small_and_large_magnitudes(dev):
    """The scales keep hi / lo inside fp16's range: tiny activations (lo subnormal or flushed) and large ones (|x| in the
    thousands) keep fp32-class RELATIVE accuracy of the result."""
    from pram_amd import ops
    w = rnd(2, "mag/w", (256, 256), 1.0 / 16.0)
    for mag in (1e-4, 1e-2, 1.0, 50.0, 1000.0):
        x = rnd(2, "mag/x", (512, 256)) * mag
        ref = F.linear(x.double(), w.double())
        e = err(ops.linear(x.to(dev), w.to(dev), precision="x3"), ref)
        rel = e / float(ref.abs().max())
        rel32 = err(ops.linear(x.to(dev), w.to(dev), precision="f32"), ref) / float(ref.abs().max())
        print(f"|x| ~ {mag:g}: relative error x3 {rel:.2e} (f32 {rel32:.2e}), absolute {e:.2e}")
        # documented bound (gemm_core_x3.h): 22 bits of every activation above 2^-7; below that lo is a subnormal fp16 and
        # an activation loses at most 2^-29 ABSOLUTE (the MFMA keeps fp16 subnormals) — a tensor of only tiny values
        # therefore keeps ~1e-5 relative, not 1e-6
        assert rel < (2e-6 if mag >= 1e-2 else 5e-5) and (mag >= 1e-2 or e < 1e-8)
    # weights of very different magnitude: the per-tensor weight scale is a power of two chosen on the host
    for wmag in (1e-5, 1e-2, 30.0):
        ws = w * wmag * 16
        x = rnd(2, "mag/x", (512, 256))
        ref = F.linear(x.double(), ws.double())
        rel = err(ops.linear(x.to(dev), ws.to(dev), precision="x3"), ref) / float(ref.abs().max())
        assert rel < 2e-6, (wmag, rel)


def test_linear_x3_rotary_and_split_planes(dev):
    from pram_amd import ops
    m = 700
    x = rnd(3, "rot/x", (m, 256))
    w = rnd(3, "rot/w", (768, 256), 1.0 / 16.0)
    b = rnd(3, "rot/b", (768,), 0.1)
    ang = W.uniform(3, "rot/a", (m, 32), -3.0, 3.0)
    cos, sin = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
    o32 = ops.linear(x.to(dev), w.to(dev), b.to(dev), rotary=(cos, sin, 512), precision="f32")
    ox3, (hi, lo) = ops.linear(x.to(dev), w.to(dev), b.to(dev), rotary=(cos, sin, 512), precision="x3", split_out="also")
    only = ops.linear(x.to(dev), w.to(dev), b.to(dev), rotary=(cos, sin, 512), precision="x3", split_out="only")
    assert only[0] is None and torch.equal(only[1][0], hi) and torch.equal(only[1][1], lo)
    assert H.maxdiff(o32, ox3) < 4e-6
    # planes: value * 16 = hi + lo to 2^-21 relative (2^-22 truncation, round to nearest on both parts)
    rec = (hi.double() + lo.double()) / 16.0
    d = (rec - ox3.double()).abs()
    big = ox3.double().abs() >= 2.0 ** -7
    rel = (d[big] / ox3.double().abs()[big]).max().item()
    print(f"split planes: max relative reconstruction error {rel:.2e} (|v| >= 2^-7), max absolute below {d[~big].max().item():.2e}")
    assert rel < 2.0 ** -21 and d[~big].max().item() <= 2.0 ** -29


def _attn_ref(q, k, v, scale, qlens=None, klens=None):
    """fp64 reference: q [B,H,M,64], k / v [B,H,N,64]"""
    s = torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale
    if klens is not None:
        n = k.shape[2]
        mask = torch.arange(n)[None, :] >= torch.tensor(klens)[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return torch.einsum("bhij,bhjd->bhid", torch.softmax(s, -1), v.double())


def _planes(ops, t2d):
    """fp32 [rows, cols] -> (hi, lo) planes of t * 16 (what the projection epilogue writes)"""
    s = t2d * 16.0
    hi = s.half()
    return hi.contiguous(), (s - hi.float()).half().contiguous()


@pytest.mark.parametrize("B,M,N,ragged", [(2, 2048, 2048, False), (3, 300, 517, True), (1, 64, 64, False), (2, 1000, 130, True)])
def test_attention_x3_vs_fp64(dev, B, M, N, ragged):
    from pram_amd import ops
    Hh = 4
    q = rnd(5, "ax/q", (B, M, Hh * 64), 1.2)
    k = rnd(5, "ax/k", (B, N, Hh * 64), 1.2)
    v = rnd(5, "ax/v", (B, N, Hh * 64))
    qlens = [M - 37 * i for i in range(B)] if ragged else None
    klens = [N - 61 * i for i in range(B)] if ragged else None
    scale = 0.125
    sp = lambda t, L: t.view(B, L, Hh, 64).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(q, M), sp(k, N), sp(v, N), scale, qlens, klens).permute(0, 2, 1, 3).reshape(B, M, Hh * 64)
    qd, kd, vd = q.view(B * M, -1).to(dev), k.view(B * N, -1).to(dev), v.view(B * N, -1).to(dev)
    ql = None if qlens is None else torch.tensor(qlens, dtype=torch.int32, device=dev)
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device=dev)
    o32 = ops.attention(qd, kd, vd, B, Hh, M, N, scale, ql, kl, precision="f32").view(B, M, -1)
    ox3, lse = ops.attention_x3(_planes(ops, qd), _planes(ops, kd), ops.value_planes_t(_planes(ops, vd), B, Hh, N, kl), B, Hh, M, N, scale,
                                ql, kl, want_lse=True)
    ox3 = ox3.view(B, M, -1)
    _, lse32 = ops.attention(qd, kd, vd, B, Hh, M, N, scale, ql, kl, want_lse=True, precision="f32")
    e32 = ex3 = el = 0.0
    for b in range(B):
        mq = M if qlens is None else qlens[b]
        e32 = max(e32, err(o32[b, :mq], ref[b, :mq]))
        ex3 = max(ex3, err(ox3[b, :mq], ref[b, :mq]))
        el = max(el, float((lse[b, :, :mq] - lse32[b, :, :mq]).abs().max()))
    print(f"attention B{B} {M}x{N}: |err| f32 {e32:.2e}  x3 {ex3:.2e}; lse |x3 - f32| {el:.2e}")
    # S = K Q^T is a full three-term split product (the log-sum-exp agrees with the f32-MFMA kernel to 1e-5); the probabilities
    # enter P V as one fp16 each, normalised by the sum of the same rounded values: a 2^-12 relative perturbation of the softmax
    # weights, independent per key -> errors of a few 1e-5 on O(1) values (vs 1e-6 for the exact kernels, 1e-3 for the fp16 path)
    # below 1024 keys the kernel splits P as well (three MFMAs per product): fp32-class like the GEMMs
    assert (ex3 < 1e-4 and el < 1e-4) if N >= 1024 else (ex3 < 3e-6 and el < 1e-5)


def test_attention_x3_cross_equals_two_directions(dev):
    """kv_shift (both directions of the matcher's cross attention in one launch) == two separate launches, bit for bit."""
    from pram_amd import ops
    B, T = 2, 384
    qk = rnd(6, "cx/qk", (2 * B * T, 256), 1.1).to(dev)
    v = rnd(6, "cx/v", (2 * B * T, 256)).to(dev)
    lens = torch.tensor([384, 300, 200, 384], dtype=torch.int32, device=dev)
    pq, pv = _planes(ops, qk), _planes(ops, v)
    both = ops.attention_x3(pq, pq, ops.value_planes_t(pv, 2 * B, 4, T, lens), 2 * B, 4, T, T, 0.125, lens, lens, kv_shift=B).view(2 * B, T, 256)
    half = lambda pl, lo, hi: (pl[0][lo * T:hi * T], pl[1][lo * T:hi * T])
    vt1 = ops.value_planes_t(half(pv, B, 2 * B), B, 4, T, lens[B:].contiguous())
    vt0 = ops.value_planes_t(half(pv, 0, B), B, 4, T, lens[:B].contiguous())
    d01 = ops.attention_x3(half(pq, 0, B), half(pq, B, 2 * B), vt1, B, 4, T, T, 0.125, lens[:B].contiguous(), lens[B:].contiguous()).view(B, T, 256)
    d10 = ops.attention_x3(half(pq, B, 2 * B), half(pq, 0, B), vt0, B, 4, T, T, 0.125, lens[B:].contiguous(), lens[:B].contiguous()).view(B, T, 256)
    for b in range(B):
        assert torch.equal(both[b, :int(lens[b])], d01[b, :int(lens[b])])
        assert torch.equal(both[B + b, :int(lens[B + b])], d10[b, :int(lens[B + b])])


def test_attention_x3_spike_and_empty(dev):
    """One key dominating a row (online-softmax rescale path) and an empty key set (context defined as 0)."""
    from pram_amd import ops
    B, M, N = 1, 256, 1024
    q = rnd(8, "sp/q", (B * M, 256))
    k = rnd(8, "sp/k", (B * N, 256))
    v = rnd(8, "sp/v", (B * N, 256))
    k[700] = q[5] * 6.0          # a spike late in the key walk
    sp = lambda t, L: t.view(B, L, 4, 64).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(q, M), sp(k, N), sp(v, N), 0.125).permute(0, 2, 1, 3).reshape(M, 256)
    vt = ops.value_planes_t(_planes(ops, v.to(dev)), B, 4, N)
    out = ops.attention_x3(_planes(ops, q.to(dev)), _planes(ops, k.to(dev)), vt, B, 4, M, N, 0.125)
    assert err(out, ref) < 1e-4          # 1024 keys: probabilities as one fp16
    kl = torch.zeros(1, dtype=torch.int32, device=dev)
    out0 = ops.attention_x3(_planes(ops, q.to(dev)), _planes(ops, k.to(dev)), vt, B, 4, M, N, 0.125, None, kl)
    assert float(out0.abs().max()) == 0.0


@pytest.mark.parametrize("S,T,n,ragged", [(3, 320, 768, True), (2, 2048, 768, False), (4, 128, 512, True), (1, 64, 768, False)])
def test_qkv_projection_writes_the_value_planes_transposed(dev, S, T, n, ragged):
    """linear_qkv_planes == linear(split_out) + value_planes_t, bit for bit: the q / k columns as row-major planes, the value
    heads straight into the key-permuted V^T planes (zeros beyond each sequence's length), wide and narrow tiles, with rotary."""
    from pram_amd import ops
    heads, hid = 4, 256
    x = rnd(21, "qkv/x", (S * T, 256)).to(dev)
    w = rnd(21, "qkv/w", (n, 256), 1.0 / 16).to(dev)
    b = rnd(21, "qkv/b", (n,), 0.1).to(dev)
    ang = W.uniform(21, "qkv/a", (S * T, 32), -3.0, 3.0)
    cos, sin = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
    rot = (cos, sin, n - hid) if n == 768 else None
    lens = torch.tensor([T - 37 * i if i % 2 == 0 else max(T - 200, 0) for i in range(S)], dtype=torch.int32, device=dev) if ragged else None
    _, pl = ops.linear(x, w, b, rotary=rot, split_out="only", lens=lens, t_pad=T if ragged else 0, precision="x3")
    vt_ref = ops.value_planes_t((pl[0][:, n - hid:], pl[1][:, n - hid:]), S, heads, T, lens)
    pq, vt = ops.linear_qkv_planes(x, w, b, heads, T, rotary=rot, lens=lens)
    for s_ in range(S):
        ln = T if lens is None else int(lens[s_])
        rows = slice(s_ * T, s_ * T + ln)
        for pl_i in (0, 1):
            assert torch.equal(pq[pl_i][rows], pl[pl_i][rows, :n - hid])
            # positions of whole 64-token blocks that hold a valid token are defined (zeros past the length); later blocks are never read
            nb = (ln + 63) // 64 * 64
            assert torch.equal(vt[pl_i][s_, :, :, :nb], vt_ref[pl_i].view(S, heads, 64, -1)[s_, :, :, :nb])


@pytest.mark.parametrize("B,M,N,ragged", [(2, 640, 640, False), (3, 300, 517, True), (2, 2048, 2048, True)])
def test_attention_colmean_x3_vs_fp64(dev, B, M, N, ragged):
    """AdaGML's token scores on the split path: the mean over heads and query rows of the soft-max matrix, from the planes and
    the row log-sum-exps of the attention call — against fp64, next to the exact-fp32 kernel; ragged lengths, an empty query set."""
    from pram_amd import ops
    Hh = 4
    q = rnd(15, "cm/q", (B, M, Hh * 64), 1.2)
    k = rnd(15, "cm/k", (B, N, Hh * 64), 1.2)
    v = rnd(15, "cm/v", (B, N, Hh * 64))
    qlens = [M - 37 * i for i in range(B)] if ragged else [M] * B
    klens = [N - 61 * i for i in range(B)] if ragged else [N] * B
    if ragged:
        qlens[-1] = 0                    # no queries: the means are reported as 0
    scale = 0.125
    sp = lambda t, L: t.view(B, L, Hh, 64).permute(0, 2, 1, 3).double()
    s = torch.einsum("bhid,bhjd->bhij", sp(q, M), sp(k, N)) * scale
    ref = torch.zeros(B, N, dtype=torch.float64)
    for b in range(B):
        if qlens[b] > 0:
            pr = torch.softmax(s[b, :, :qlens[b], :klens[b]], -1)
            ref[b, :klens[b]] = pr.mean(dim=(0, 1))
    qd, kd, vd = q.view(B * M, -1).to(dev), k.view(B * N, -1).to(dev), v.view(B * N, -1).to(dev)
    ql = torch.tensor(qlens, dtype=torch.int32, device=dev)
    kl = torch.tensor(klens, dtype=torch.int32, device=dev)
    pq, pk = _planes(ops, qd), _planes(ops, kd)
    _, lse = ops.attention_x3(pq, pk, ops.value_planes_t(_planes(ops, vd), B, Hh, N, kl), B, Hh, M, N, scale, ql, kl, want_lse=True)
    cx3 = ops.attention_colmean_x3(pq, pk, lse, B, Hh, M, N, scale, ql, kl)
    _, lse32 = ops.attention(qd, kd, vd, B, Hh, M, N, scale, ql, kl, want_lse=True, precision="f32")
    c32 = ops.attention_colmean(qd, kd, lse32, B, Hh, M, N, scale, ql, kl)
    assert torch.isfinite(cx3).all()
    ex3, e32 = err(cx3, ref), err(c32, ref)
    rel = ex3 / float(ref.max())
    print(f"colmean B{B} {M}x{N}: |err| f32 {e32:.2e}  x3 {ex3:.2e} (largest mean {float(ref.max()):.2e})")
    assert rel < 2e-5 and ex3 < 4 * e32 + 1e-8
    for b in range(B):
        assert float(cx3[b, klens[b]:].abs().max() if klens[b] < N else 0.0) == 0.0


def test_attention_colmean_x3_cross_equals_two_directions(dev):
    from pram_amd import ops
    B, T = 2, 384
    qk = rnd(16, "cmx/qk", (2 * B * T, 256), 1.1).to(dev)
    v = rnd(16, "cmx/v", (2 * B * T, 256)).to(dev)
    lens = torch.tensor([384, 300, 200, 384], dtype=torch.int32, device=dev)
    pq, pv = _planes(ops, qk), _planes(ops, v)
    _, lse = ops.attention_x3(pq, pq, ops.value_planes_t(pv, 2 * B, 4, T, lens), 2 * B, 4, T, T, 0.125, lens, lens, want_lse=True, kv_shift=B)
    both = ops.attention_colmean_x3(pq, pq, lse, 2 * B, 4, T, T, 0.125, lens, lens, kv_shift=B)
    half = lambda pl, lo, hi: (pl[0][lo * T:hi * T], pl[1][lo * T:hi * T])
    l0, l1 = lens[:B].contiguous(), lens[B:].contiguous()
    # queries of set 0 over the keys of set 1 -> one mean per set-1 token (rows B..2B-1), and the other way round
    c1 = ops.attention_colmean_x3(half(pq, 0, B), half(pq, B, 2 * B), lse[:B].contiguous(), B, 4, T, T, 0.125, l0, l1)
    c0 = ops.attention_colmean_x3(half(pq, B, 2 * B), half(pq, 0, B), lse[B:].contiguous(), B, 4, T, T, 0.125, l1, l0)
    assert torch.equal(both[B:], c1) and torch.equal(both[:B], c0)


@pytest.mark.parametrize("B,M,N", [(3, 2048, 2048), (2, 300, 517), (1, 64, 64)])
def test_bgemm_planes_is_fp32_class(dev, B, M, N):
    """The matcher's score matrix mdesc0 . mdesc1^T with both operands as split planes (batched launch, wide and narrow tiles)."""
    from pram_amd import ops
    a = rnd(31, "bg/a", (B * M, 256), 0.5)
    b = rnd(31, "bg/b", (B * N, 256), 0.5)
    ref = torch.einsum("bmk,bnk->bmn", a.view(B, M, 256).double(), b.view(B, N, 256).double()) * 0.25
    ldc = (N + 3) // 4 * 4
    got = ops.bgemm_nt_planes(_planes(ops, a.to(dev)), _planes(ops, b.to(dev)), B, M, N, alpha=0.25, ldc=ldc)[:, :, :N]
    c32 = ops.bgemm_nt(a.view(B, M, 256).to(dev), b.view(B, N, 256).to(dev), alpha=0.25, ldc=ldc)[:, :, :N]
    ex3, e32 = err(got, ref), err(c32, ref)
    print(f"bgemm {B}x{M}x{N}: |err| f32 {e32:.2e}  x3 {ex3:.2e}")
    assert ex3 < 4e-6 and ex3 < 4 * e32 + 1e-6


@pytest.mark.parametrize("B,M,N,ragged", [(2, 2048, 2048, False), (3, 300, 517, True), (1, 64, 64, False)])
def test_attention_h16t_single_product_mode(dev, B, M, N, ragged):
    """The fp16 path's attention on the pipelined kernel (hi planes only, one MFMA per product): against fp64 fed the same
    fp16-rounded q / k / v the error is that of fp16 probabilities (2^-12 relative each), far inside the fp16 path's own bar;
    ragged lengths, cross-attention shift and lse like the split-path kernel."""
    from pram_amd import ops
    Hh = 4
    q = rnd(25, "h16/q", (B, M, Hh * 64), 1.2).half()
    k = rnd(25, "h16/k", (B, N, Hh * 64), 1.2).half()
    v = rnd(25, "h16/v", (B, N, Hh * 64)).half()
    qlens = [M - 37 * i for i in range(B)] if ragged else None
    klens = [N - 61 * i for i in range(B)] if ragged else None
    sp = lambda t, L: t.float().view(B, L, Hh, 64).permute(0, 2, 1, 3)
    ref = _attn_ref(sp(q, M), sp(k, N), sp(v, N), 0.125, qlens, klens).permute(0, 2, 1, 3).reshape(B, M, Hh * 64)
    qd, kd, vd = q.view(B * M, -1).to(dev), k.view(B * N, -1).to(dev), v.view(B * N, -1).to(dev)
    ql = None if qlens is None else torch.tensor(qlens, dtype=torch.int32, device=dev)
    kl = None if klens is None else torch.tensor(klens, dtype=torch.int32, device=dev)
    out, lse = ops.attention_h16t(qd, kd, ops.value_t16(vd, B, Hh, N, kl), B, Hh, M, N, 0.125, ql, kl, want_lse=True)
    old = ops.attention_h16(qd, kd, vd, B, Hh, M, N, 0.125, ql, kl).view(B, M, -1)
    out = out.view(B, M, -1)
    e_new = e_old = 0.0
    for b in range(B):
        mq = M if qlens is None else qlens[b]
        e_new = max(e_new, err(out[b, :mq], ref[b, :mq]))
        e_old = max(e_old, err(old[b, :mq], ref[b, :mq]))
    print(f"attention_h16t B{B} {M}x{N}: |err| vs fp64 (same fp16 operands) {e_new:.2e}; attention_h16 {e_old:.2e}")
    assert torch.isfinite(lse).all() and e_new < 5e-4


def test_conv_x3_is_fp32_class(dev):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    for cin, cout, ks, stride, h, w in ((64, 64, 3, 2, 40, 56), (128, 256, 3, 1, 30, 40), (256, 65, 1, 1, 15, 20), (256, 256, 3, 1, 24, 32),
                                        (64, 128, 3, 1, 33, 47)):
        x = rnd(7, "cv/x", (2, cin, h, w))
        wt = rnd(7, "cv/w", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
        b = rnd(7, "cv/b", (cout,), 0.1)
        sc = W.uniform(7, "cv/s", (cout,), 0.8, 1.2)
        sh = rnd(7, "cv/t", (cout,), 0.1)
        ref = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=ks // 2) * sc.double().view(1, -1, 1, 1)
                         + sh.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
        xa = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wa = ResNet4x._ohwi(wt).to(dev)
        kw = dict(ks=ks, stride=stride, relu=True)
        e32 = err(ops.conv2d_nhwc(xa, wa, b.to(dev), sc.to(dev), sh.to(dev), precision="f32", **kw), ref)
        ex3 = err(ops.conv2d_nhwc(xa, wa, b.to(dev), sc.to(dev), sh.to(dev), precision="x3", **kw), ref)
        print(f"conv {cin}->{cout} k{ks} s{stride}: |err| f32 {e32:.2e}  x3 {ex3:.2e}")
        assert ex3 < 1e-5 and ex3 < 4 * e32 + 1e-6      # fp32 accumulation over up to 2304 terms: the order of the sum moves the last bits


@pytest.mark.parametrize("prec", ["x3", "f32"])
def test_models_meet_the_fp32_bars_on_both_exact_paths(dev, prec):
    """End to end against the fp32 oracle on the split-fp16 path (the default) and on the exact-fp32 MFMA path: SegNetViT
    logits <= 1e-3 with identical arg-max, GML and AdaGML indices exact — the same bars for both (full-size variants live
    in test_gpu_models.py, which runs on the process default precision)."""
    from oracle import ref_cpu as R
    from pram_amd import ops
    from pram_amd.nets.adagml import AdaGML
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    n = 1024
    seg = load_segnet('segnetvit', 113, 256, 15, 1024)
    seg.load_state_dict(H.segnet_sd(113), strict=True)
    seg.to(dev).eval().set_precision(prec)
    desc, kp, _ = W.synthetic_tokens(4, n)
    ref = R.segnetvit_forward(H.segnet_sd(113), desc[None], kp[None], (1, 3, 480, 640))
    got = seg({'seg_descriptors': desc[None].to(dev), 'keypoints': kp[None].to(dev), 'image': torch.empty(1, 3, 480, 640)})['prediction']
    d = H.maxdiff(got, ref)
    print(f"SegNetViT {prec} N={n}: logits |d|max {d:.2e}")
    assert d < 1e-3 and torch.equal(got.argmax(-1).cpu(), ref.argmax(-1))
    for cls, sd, fn in ((GML, H.gml_sd(), R.gml_produce_matches), (AdaGML, H.adagml_sd(), R.adagml_produce_matches)):
        net = cls({})
        net.load_state_dict(sd, strict=True)
        net.to(dev).eval().set_precision(prec)
        data, _ = H.pair_data(3, 640, 768)
        ref = fn(sd, data)
        got = net.produce_matches({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()})
        ds = H.maxdiff(got['matching_scores0'], ref['matching_scores0'])
        print(f"{cls.__name__} {prec} 640x768: matches {(ref['matches0'] >= 0).sum().item()}, scores |d|max {ds:.2e}")
        assert torch.equal(got['matches0'].cpu(), ref['matches0']) and ds < 1e-3
