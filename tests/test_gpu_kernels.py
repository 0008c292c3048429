"""GPU (MI355X): each HIP kernel, called through the C ABI (pram_amd.ops -> libpram_hip.so), against the
CPU oracle on the same seeded inputs.  fp32 tolerances are written next to each check; index outputs
are compared bit-exactly."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from pram_amd import weights as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rnd(seed, name, shape, std=1.0):
    return W.normal(seed, name, shape, std)


# ------------------------------------------------------------------ linear / LN / fourier
@pytest.mark.parametrize("m,k0,k1,n", [(300, 256, 0, 768), (1000, 256, 256, 512), (77, 1024, 0, 113), (128, 128, 0, 256),
                                       (513, 512, 0, 256), (64, 4, 0, 256), (200, 256, 0, 4)])
def test_linear(dev, m, k0, k1, n):
    from pram_amd import ops
    x = rnd(1, "lin/x", (m, k0))
    x2 = rnd(1, "lin/x2", (m, k1)) if k1 else None
    w = rnd(1, "lin/w", (n, k0 + k1), 1.0 / math.sqrt(k0 + k1))
    b = rnd(1, "lin/b", (n,), 0.1)
    res = rnd(1, "lin/r", (m, n))
    ref = F.linear(torch.cat([x, x2], -1) if k1 else x, w, b) * 0.25 + res
    out = ops.linear(x.to(dev), w.to(dev), b.to(dev), x2=None if x2 is None else x2.to(dev), residual=res.to(dev), alpha=0.25)
    assert H.maxdiff(out, ref) < 2e-5          # fp32 accumulate, different summation order
    if not k1:
        assert H.maxdiff(ops.linear(x.to(dev), w.to(dev)), F.linear(x, w)) < 2e-5


@pytest.mark.parametrize("m,k0,k1,n", [(300, 256, 0, 768), (1000, 256, 256, 512), (77, 1024, 0, 113), (513, 512, 0, 256)])
def test_linear_f16_path(dev, m, k0, k1, n):
    """C5 fp16 MFMA GEMM: exact (to fp32 accumulation order) against a reference fed the SAME fp16-rounded operands,
    and within the documented 2e-2 of the fp32 result."""
    from pram_amd import ops
    x = rnd(1, "lin/x", (m, k0))
    x2 = rnd(1, "lin/x2", (m, k1)) if k1 else None
    w = rnd(1, "lin/w", (n, k0 + k1), 1.0 / math.sqrt(k0 + k1))
    b = rnd(1, "lin/b", (n,), 0.1)
    res = rnd(1, "lin/r", (m, n))
    xa = torch.cat([x, x2], -1) if k1 else x
    ref = F.linear(xa, w, b) * 0.25 + res
    ref16 = F.linear(xa.half().double(), w.half().double(), b.double()).float() * 0.25 + res
    old = ops.gemm_precision
    ops.gemm_precision = "f16"
    try:
        out = ops.linear(x.to(dev), w.to(dev), b.to(dev), x2=None if x2 is None else x2.to(dev), residual=res.to(dev), alpha=0.25)
    finally:
        ops.gemm_precision = old
    print(f"f16 linear {m}x{k0}+{k1}x{n}: vs fp16-operand fp64 {H.maxdiff(out, ref16):.2e}, vs fp32 {H.maxdiff(out, ref):.2e}")
    assert H.maxdiff(out, ref16) < 2e-5 and H.maxdiff(out, ref) < 2e-2


def test_conv_f16_path(dev):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    for cin, cout, ks, stride, h, w in ((64, 64, 3, 2, 40, 56), (128, 256, 3, 1, 30, 40), (256, 65, 1, 1, 15, 20), (256, 256, 3, 1, 24, 32)):
        x = rnd(7, "cv/x", (2, cin, h, w))
        wt = rnd(7, "cv/w", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
        b = rnd(7, "cv/b", (cout,), 0.1)
        ref16 = F.conv2d(x.half().double(), wt.half().double(), b.double(), stride=stride, padding=ks // 2).float()
        old = ops.gemm_precision
        ops.gemm_precision = "f16"
        try:
            out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ResNet4x._ohwi(wt).to(dev), b.to(dev), ks=ks, stride=stride)
        finally:
            ops.gemm_precision = old
        assert H.maxdiff(out.permute(0, 3, 1, 2), ref16) < 2e-5


def test_linear_mfma_layout_asymmetric(dev):
    """A = I-like check with asymmetric B: catches row/col swaps in the MFMA fragment maps."""
    from pram_amd import ops
    m = n = k = 128
    x = torch.eye(m, k)
    w = (torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 251) / 7.0
    out = ops.linear(x.to(dev), w.to(dev), precision="f32")
    assert torch.equal(out.cpu(), w.t().contiguous())          # exact-fp32 MFMA: 1.0 * w is w
    out3 = ops.linear(x.to(dev), w.to(dev), precision="x3")
    assert float(((out3.cpu() - w.t()).abs() / w.t().clamp_min(1.0)).max()) <= 2.0 ** -21      # split-fp16: w to 22 bits


def test_qkv_rotary_epilogue(dev):
    """qkv projection with the permuted weight + rotary epilogue == reference layout + apply_cached_rotary_emb."""
    from pram_amd import ops
    from pram_amd.nets import _blocks as blk
    B, N = 2, 200
    x = rnd(2, "rot/x", (B, N, 256))
    sd = {"p.qkv.weight": rnd(2, "rot/w", (768, 256), 1 / 16.0), "p.qkv.bias": rnd(2, "rot/b", (768,), 0.1)}
    sd.update({"p.proj.weight": torch.zeros(256, 256), "p.proj.bias": torch.zeros(256), "p.mlp.0.weight": torch.zeros(512, 512),
               "p.mlp.0.bias": torch.zeros(512), "p.mlp.1.weight": torch.ones(512), "p.mlp.1.bias": torch.zeros(512),
               "p.mlp.3.weight": torch.zeros(256, 512), "p.mlp.3.bias": torch.zeros(256)})
    kp = torch.stack([W.synthetic_tokens(0, N)[1], W.synthetic_tokens(1, N)[1]])
    Wr = rnd(2, "rot/wr", (32, 2))
    cos_r, sin_r = R.fourier_encoding(Wr, R.normalize_keypoints(kp, (1, 3, 480, 640)))
    cos, sin = ops.fourier_encoding(kp.to(dev), Wr.to(dev), 320.0, 240.0, 448.0)
    assert H.maxdiff(cos, cos_r) < 2e-6 and H.maxdiff(sin, sin_r) < 2e-6     # cosf/sinf vs torch.cos/sin
    qkv_ref = F.linear(x, sd["p.qkv.weight"], sd["p.qkv.bias"]).reshape(B, N, 4, 64, 3).permute(0, 2, 1, 3, 4)
    q_ref = R.rotary(qkv_ref[..., 0], cos_r, sin_r)
    k_ref = R.rotary(qkv_ref[..., 1], cos_r, sin_r)
    v_ref = qkv_ref[..., 2]
    pk = blk.pack_self_block(sd, "p", dev)
    out = ops.linear(x.to(dev).view(B * N, 256), pk["qkv_w"], pk["qkv_b"], rotary=(cos.view(-1, 32), sin.view(-1, 32), 512))
    out = out.cpu().view(B, N, 3, 4, 64)
    perm = torch.cat([torch.arange(0, 64, 2), torch.arange(1, 64, 2)])
    assert H.maxdiff(out[:, :, 0].permute(0, 2, 1, 3), q_ref[..., perm]) < 1e-5
    assert H.maxdiff(out[:, :, 1].permute(0, 2, 1, 3), k_ref[..., perm]) < 1e-5
    assert H.maxdiff(out[:, :, 2].permute(0, 2, 1, 3), v_ref) < 1e-5


@pytest.mark.parametrize("rows,cols", [(100, 512), (33, 1024), (257, 256)])
def test_layernorm_gelu(dev, rows, cols):
    from pram_amd import ops
    x = rnd(3, "ln/x", (rows, cols), 2.0)
    g, b = W.uniform(3, "ln/g", (cols,), 0.8, 1.2), W.uniform(3, "ln/b", (cols,), -0.1, 0.1)
    ref = F.gelu(F.layer_norm(x, (cols,), g, b, 1e-5))
    out = ops.layernorm_gelu_(x.to(dev).clone(), g.to(dev), b.to(dev))
    assert H.maxdiff(out, ref) < 5e-6


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, scale):
    a = torch.softmax(torch.einsum("bhid,bhjd->bhij", q.double(), k.double()) * scale, -1)
    return torch.einsum("bhij,bhjd->bhid", a, v.double()), a


@pytest.mark.parametrize("B,M,N", [(1, 128, 64), (2, 300, 300), (1, 70, 1000), (2, 513, 129), (1, 2048, 2048)])
def test_attention_vs_fp64(dev, B, M, N):
    """full-tensor fp64 reference; tolerance 2e-5 abs on O(1) outputs (fp32 MFMA, online softmax)."""
    from pram_amd import ops
    Hh = 4
    q, k, v = rnd(4, "at/q", (B, Hh, M, 64)), rnd(4, "at/k", (B, Hh, N, 64)), rnd(4, "at/v", (B, Hh, N, 64))
    ref, a = _attn_ref(q, k, v, 0.125)
    to2d = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, Hh * 64).contiguous().to(dev)
    out, lse = ops.attention(to2d(q, M), to2d(k, N), to2d(v, N), B, Hh, M, N, 0.125, want_lse=True)
    out = out.cpu().view(B, M, Hh, 64).permute(0, 2, 1, 3)
    assert H.maxdiff(out, ref) < 2e-5
    col = ops.attention_colmean(to2d(q, M), to2d(k, N), lse, B, Hh, M, N, 0.125)
    assert H.maxdiff(col, a.mean(1).mean(1)) < 1e-6


@pytest.mark.parametrize("B,M,N", [(1, 128, 64), (2, 300, 333), (1, 70, 1000), (1, 2048, 2048)])
def test_attention_f16_path(dev, B, M, N):
    """C5 'fp16 MFMA path': fp16 operands, fp32 accumulate.  Documented tolerance: 4e-3 abs on O(1) outputs vs fp64
    (operand rounding 2^-11), and much tighter vs an fp64 reference fed the SAME fp16-rounded q, k, v (layout check)."""
    from pram_amd import ops
    Hh = 4
    q, k, v = rnd(4, "at/q", (B, Hh, M, 64)), rnd(4, "at/k", (B, Hh, N, 64)), rnd(4, "at/v", (B, Hh, N, 64))
    ref, _ = _attn_ref(q, k, v, 0.125)
    ref16, _ = _attn_ref(q.half().float(), k.half().float(), v.half().float(), 0.125)
    to2d = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, Hh * 64).contiguous().to(dev)
    out, lse = ops.attention(to2d(q, M), to2d(k, N), to2d(v, N), B, Hh, M, N, 0.125, want_lse=True, precision="f16")
    out = out.cpu().view(B, M, Hh, 64).permute(0, 2, 1, 3)
    d, d16 = H.maxdiff(out, ref), H.maxdiff(out, ref16)
    print(f"f16 attention {B}x{M}x{N}: |o - fp64| {d:.2e}, |o - fp64(fp16 inputs)| {d16:.2e}")
    assert d < 4e-3 and d16 < 1.5e-3
    # ragged: padded element == unpadded run, bit for bit
    if B == 2:
        L = torch.tensor([M, M - 77], dtype=torch.int32, device=dev)
        Lk = torch.tensor([N, N - 100], dtype=torch.int32, device=dev)
        o2 = ops.attention(to2d(q, M), to2d(k, N), to2d(v, N), B, Hh, M, N, 0.125, L, Lk, precision="f16").view(B, M, 256)
        one = lambda t: t[1].permute(1, 0, 2).reshape(t.shape[2], Hh * 64).contiguous().to(dev)
        solo = ops.attention(one(q[:, :, :M - 77]), one(k[:, :, :N - 100]), one(v[:, :, :N - 100]),
                             1, Hh, M - 77, N - 100, 0.125, precision="f16")
        assert torch.equal(o2[1, :M - 77], solo)


def test_attention_spiked_key_forces_rescale(dev):
    """One key dominates one query late in the sequence: the running max jumps mid-stream (online-softmax rescale)."""
    from pram_amd import ops
    B, Hh, M, N = 1, 4, 64, 512
    q, k, v = rnd(5, "sp/q", (B, Hh, M, 64)), rnd(5, "sp/k", (B, Hh, N, 64)), rnd(5, "sp/v", (B, Hh, N, 64))
    k[0, :, 300] = q[0, :, 7] * 6.0
    k[0, :, 5] = q[0, :, 9] * 6.0
    ref, _ = _attn_ref(q, k, v, 0.125)
    to2d = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, Hh * 64).contiguous().to(dev)
    out = ops.attention(to2d(q, M), to2d(k, N), to2d(v, N), B, Hh, M, N, 0.125).cpu().view(B, M, Hh, 64).permute(0, 2, 1, 3)
    assert H.maxdiff(out, ref) < 2e-5


def test_attention_ragged_equals_unpadded(dev):
    """lens: a padded batch element must produce exactly its B = 1 result (no masks in the reference)."""
    from pram_amd import ops
    Hh, T = 4, 320
    lens = [320, 200, 77]
    q, k, v = rnd(6, "rg/q", (3, T, 256)), rnd(6, "rg/k", (3, T, 256)), rnd(6, "rg/v", (3, T, 256))
    L = torch.tensor(lens, dtype=torch.int32, device=dev)
    f2 = lambda t: t.reshape(3 * T, 256).to(dev)
    out = ops.attention(f2(q), f2(k), f2(v), 3, Hh, T, T, 0.125, L, L).view(3, T, 256)
    for b, n in enumerate(lens):
        solo = ops.attention(q[b, :n].contiguous().to(dev), k[b, :n].contiguous().to(dev), v[b, :n].contiguous().to(dev),
                             1, Hh, n, n, 0.125)
        assert torch.equal(out[b, :n], solo)


@pytest.mark.parametrize("S,T,lens", [(1, 2048, None), (2, 1500, [1500, 700]), (3, 1100, [1100, 513, 40]), (1, 600, [512])])
def test_attention_split_equals_fused(dev, S, T, lens):
    """Small launches run the key chunks as separate workgroups + a combine pass; the fold is the one the fused kernel
    applies, in the same order: outputs and log-sum-exp are identical bit for bit (so a batch element still equals
    its B = 1 run whichever mode either launch used)."""
    from pram_amd import ops
    Hh = 4
    q, k, v = rnd(8, "sf/q", (S * T, 256)).to(dev), rnd(8, "sf/k", (S * T, 256)).to(dev), rnd(8, "sf/v", (S * T, 256)).to(dev)
    L = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    assert ops._lib.load().pram_attention_workspace_bytes(S, Hh, T, T) > 0
    try:
        ops.attention_split = True
        o1, l1 = ops.attention(q, k, v, S, Hh, T, T, 0.125, L, L, want_lse=True)
        ops.attention_split = False
        o0, l0 = ops.attention(q, k, v, S, Hh, T, T, 0.125, L, L, want_lse=True)
    finally:
        ops.attention_split = True
    for b in range(S):
        n = lens[b] if lens else T
        assert torch.equal(o1.view(S, T, 256)[b, :n], o0.view(S, T, 256)[b, :n])
        assert torch.equal(l1[b, :, :n], l0[b, :, :n])
    # and against fp64
    b, n = 0, (lens[0] if lens else T)
    f = lambda t: t.view(S, T, 4, 64)[b, :n].permute(1, 0, 2).double().cpu()[None]
    ref, _ = _attn_ref(f(q), f(k), f(v), 0.125)
    got = o1.view(S, T, 4, 64)[b, :n].permute(1, 0, 2).cpu()[None]
    assert H.maxdiff(got, ref) < 2e-5


@pytest.mark.parametrize("prec", ["f32", "f16"])
def test_attention_cross_equals_two_calls(dev, prec):
    """the single-launch cross attention (both directions, ragged lens) == two separate launches, bit for bit,
    outputs, log-sum-exp and the AdaGML column means alike"""
    from pram_amd import ops
    Hh, T, B = 4, 300, 3
    qk, v = rnd(7, "cx/qk", (2 * B * T, 256)).to(dev), rnd(7, "cx/v", (2 * B * T, 256)).to(dev)
    lens = torch.tensor([300, 211, 64, 97, 300, 130], dtype=torch.int32, device=dev)
    half = B * T
    o, lse = ops.attention_cross(qk, v, B, Hh, T, 0.125, lens, want_lse=True, precision=prec)
    o0, l0 = ops.attention(qk[:half], qk[half:], v[half:], B, Hh, T, T, 0.125, lens[:B], lens[B:], want_lse=True, precision=prec)
    o1, l1 = ops.attention(qk[half:], qk[:half], v[:half], B, Hh, T, T, 0.125, lens[B:], lens[:B], want_lse=True, precision=prec)
    for s in range(2 * B):
        n = int(lens[s])
        want_o, want_l = (o0, l0) if s < B else (o1, l1)
        i = s % B
        assert torch.equal(o.view(2 * B, T, 256)[s, :n], want_o.view(B, T, 256)[i, :n])
        assert torch.equal(lse[s, :, :n], want_l[i, :, :n])
    if prec == "f32":
        col = ops.attention_cross_colmean(qk, lse, B, Hh, T, 0.125, lens)
        c1 = ops.attention_colmean(qk[:half], qk[half:], l0, B, Hh, T, T, 0.125, lens[:B], lens[B:])   # per set-1 token
        c0 = ops.attention_colmean(qk[half:], qk[:half], l1, B, Hh, T, T, 0.125, lens[B:], lens[:B])   # per set-0 token
        assert torch.equal(col[:B], c0) and torch.equal(col[B:], c1)
    # no lens
    o2 = ops.attention_cross(qk, v, B, Hh, T, 0.125, precision=prec)
    o20 = ops.attention(qk[:half], qk[half:], v[half:], B, Hh, T, T, 0.125, precision=prec)
    assert torch.equal(o2[:half], o20)


def test_fp16_operands_in_hbm_equal_the_staged_rounding(dev):
    """C5 path: the projection can hand q / k / v over as fp16 (pram_linear_f16_h16) and attention_h16 reads them
    directly.  Both steps only move WHERE the round-to-fp16 happens, so: the fp16 copy equals the fp32 output rounded,
    and attention_h16 equals attention(precision="f16") on the fp32 tensors — bit for bit, self and cross, ragged."""
    from pram_amd import ops
    Hh, T, S = 4, 333, 4
    x = rnd(9, "h16/x", (S * T, 256)).to(dev)
    w, b = rnd(9, "h16/w", (768, 256), 0.06).to(dev), rnd(9, "h16/b", (768,), 0.1).to(dev)
    old = ops.gemm_precision
    try:
        ops.gemm_precision = "f16"
        o32 = ops.linear(x, w, b)
        o32b, o16 = ops.linear(x, w, b, half_copy="also")
        _, o16only = ops.linear(x, w, b, half_copy="only")
    finally:
        ops.gemm_precision = old
    assert torch.equal(o32, o32b) and torch.equal(o16, o32.half()) and torch.equal(o16only, o16)
    lens = torch.tensor([333, 200, 64, 301], dtype=torch.int32, device=dev)
    q, k, v = o32[:, :256], o32[:, 256:512], o32[:, 512:]
    ref, lse = ops.attention(q, k, v, S, Hh, T, T, 0.125, lens, lens, want_lse=True, precision="f16")
    got, lse2 = ops.attention_h16(o16[:, :256], o16[:, 256:512], o16[:, 512:], S, Hh, T, T, 0.125, lens, lens, want_lse=True)
    for s_ in range(S):
        n = int(lens[s_])
        assert torch.equal(got.view(S, T, 256)[s_, :n], ref.view(S, T, 256)[s_, :n]) and torch.equal(lse2[s_, :, :n], lse[s_, :, :n])
    # cross pairing (kv_shift) against the fp32-input single-launch cross attention on the fp16 MFMA path
    refx = ops.attention_cross(q, v, S // 2, Hh, T, 0.125, lens, precision="f16")
    gotx = ops.attention_h16(o16[:, :256], o16[:, :256], o16[:, 512:], S, Hh, T, T, 0.125, lens, lens, kv_shift=S // 2)
    for s_ in range(S):
        n = int(lens[s_])
        assert torch.equal(gotx.view(S, T, 256)[s_, :n], refx.view(S, T, 256)[s_, :n])
    with pytest.raises(Exception):
        ops.linear(x, w, b, half_copy="only")          # needs the fp16 GEMM path


# ------------------------------------------------------------------ sinkhorn / matches
def _sink_input(tag, m, n):
    M = W.normal(11, f"sink/{tag}", (2, m, n), 2.0)
    for b in range(2):
        idx = torch.argsort(W.uniform(12 + b, f"sink/perm/{tag}", (m,)))[: min(m, n)]
        M[b, idx, torch.arange(min(m, n))] += 6.0
    return M


@pytest.mark.parametrize("tag", ["257x193", "100x300"])
def test_sinkhorn_golden(dev, golden, tag):
    """against the reference's sink_algorithm / dual_softmax / compute_matches outputs (golden)."""
    from pram_amd.nets import gml
    from pram_amd import ops
    g = golden(f"sinkhorn_{tag}")
    m, n = int(g["m"]), int(g["n"])
    M = _sink_input(tag, m, n).to(dev)
    bs = torch.tensor(1.0, device=dev)
    p = gml.sink_algorithm(M, bs, 20).cpu().numpy()
    # entries span 1e-11 .. 1e2 (dustbin corner): 1e-6 abs + 2e-6 relative
    err = np.abs(p - g["p"]) - 2e-6 * np.abs(g["p"])
    assert err.max() < 1e-6, err.max()
    d = gml.dual_softmax(M, bs).cpu().numpy()
    assert np.abs(d - g["dual"]).max() < 1e-5
    for thr, k0, k1 in ((0.0, "m0_p0", "m1_p0"), (0.2, "m0_p02", "m1_p02")):
        r = ops.sinkhorn_match(M.contiguous(), bs, 20, thr)
        assert np.array_equal(r["matches0"].cpu().numpy(), g[k0]) and np.array_equal(r["matches1"].cpu().numpy(), g[k1])
    r = ops.sinkhorn_match(M.contiguous(), bs, 20, 0.0)
    assert np.abs(r["matching_scores0"].cpu().numpy() - g["s0"]).max() < 1e-5
    assert np.abs(r["matching_scores1"].cpu().numpy() - g["s1"]).max() < 1e-5


def test_sinkhorn_2049(dev):
    """full-size 2048 x 2048 problem vs the oracle; indices exact, P within 1e-6 abs + 2e-6 rel."""
    from pram_amd import ops
    m = n = 2048
    M = W.normal(21, "sink/big", (1, m, n), 2.0)
    idx = torch.argsort(W.uniform(22, "sink/bigperm", (m,)))
    M[0, idx, torch.arange(n)] += 8.0
    p_ref = R.sink_algorithm(M, torch.tensor(1.0), 20)
    i0, i1, s0, s1 = R.compute_matches(p_ref, 0.2)
    r = ops.sinkhorn_match(M.to(dev).contiguous(), torch.tensor(1.0, device=dev), 20, 0.2, want_p=True)
    assert float(((r["p"].cpu() - p_ref).abs() - 2e-6 * p_ref.abs()).max()) < 1e-6
    assert torch.equal(r["matches0"].cpu(), i0) and torch.equal(r["matches1"].cpu(), i1)
    assert H.maxdiff(r["matching_scores0"], s0) < 1e-5
    assert (i0 >= 0).sum() > 500


def test_compute_matches_api(dev, golden):
    """GML.compute_matches on a materialised score matrix == the reference's outputs (golden)."""
    from pram_amd.nets.gml import GML
    g = golden("sinkhorn_257x193")
    net = GML({})
    i0, i1, s0, s1 = net.compute_matches(torch.from_numpy(g["p"]).to(dev), p=0.2)
    assert np.array_equal(i0.cpu().numpy(), g["m0_p02"]) and np.array_equal(i1.cpu().numpy(), g["m1_p02"])
    assert np.abs(s0.cpu().numpy() - g["s0"]).max() < 1e-7 and np.abs(s1.cpu().numpy() - g["s1"]).max() < 1e-7


def test_argmax_ties_lowest_index(dev):
    from pram_amd import ops
    M = torch.zeros(1, 6, 8)
    M[0, 2, 3] = M[0, 2, 5] = 4.0      # row tie
    M[0, 4, 1] = M[0, 5, 1] = 4.0      # column tie
    p_ref = R.sink_algorithm(M, torch.tensor(1.0), 20)
    i0, i1, _, _ = R.compute_matches(p_ref, 0.0)
    r = ops.sinkhorn_match(M.to(dev).contiguous(), torch.tensor(1.0, device=dev), 20, 0.0)
    assert torch.equal(r["matches0"].cpu(), i0) and torch.equal(r["matches1"].cpu(), i1)


# ------------------------------------------------------------------ SFD2 kernels
@pytest.mark.parametrize("cin,cout,ks,stride,h,w", [(64, 64, 3, 2, 40, 56), (128, 256, 3, 1, 30, 40), (256, 65, 1, 1, 15, 20),
                                                    (256, 256, 3, 2, 24, 32), (256, 128, 1, 1, 17, 23)])
def test_conv2d(dev, cin, cout, ks, stride, h, w):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    x = rnd(7, "cv/x", (2, cin, h, w))
    wt = rnd(7, "cv/w", (cout, cin, ks, ks), 1.0 / math.sqrt(cin * ks * ks))
    b = rnd(7, "cv/b", (cout,), 0.1)
    sc, sh = W.uniform(7, "cv/s", (cout,), 0.8, 1.2), W.uniform(7, "cv/t", (cout,), -0.1, 0.1)
    y = F.conv2d(x, wt, b, stride=stride, padding=ks // 2)
    res = rnd(7, "cv/r", tuple(y.shape))
    ref = F.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ResNet4x._ohwi(wt).to(dev), b.to(dev), sc.to(dev), sh.to(dev),
                          residual=res.permute(0, 2, 3, 1).contiguous().to(dev), ks=ks, stride=stride, relu=True)
    assert H.maxdiff(out.permute(0, 3, 1, 2), ref) < 2e-5


def test_conv1a_rgb(dev):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    x = rnd(8, "c1/x", (2, 3, 48, 64))
    wt = rnd(8, "c1/w", (64, 3, 3, 3), 0.2)
    b = rnd(8, "c1/b", (64,), 0.1)
    ref = F.conv2d(x, wt, b, padding=1)
    out = ops.conv2d_nhwc(ops.image_to_nhwc4(x.to(dev)), ResNet4x._ohwi(wt, 4).to(dev), b.to(dev), ks=3)
    assert H.maxdiff(out.permute(0, 3, 1, 2), ref) < 1e-5
    assert torch.equal(ops.nhwc_to_nchw(out).cpu(), out.permute(0, 3, 1, 2).contiguous().cpu())


def test_grouped_conv(dev):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    x = rnd(9, "gc/x", (2, 256, 18, 22))
    wt = rnd(9, "gc/w", (256, 8, 3, 3), 1.0 / math.sqrt(72))
    sc, sh = W.uniform(9, "gc/s", (256,), 0.8, 1.2), W.uniform(9, "gc/t", (256,), -0.1, 0.1)
    ref = F.relu(F.conv2d(x, wt, padding=1, groups=32) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    out = ops.conv3x3_grouped_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ResNet4x._ohwi(wt).to(dev), sc.to(dev), sh.to(dev), 32, True)
    assert H.maxdiff(out.permute(0, 3, 1, 2), ref) < 1e-5


def test_score_map(dev):
    from pram_amd import ops
    lg = rnd(10, "sm/l", (2, 65, 12, 16), 3.0)
    ref = R.score_map_from_logits(lg)
    out = ops.score_map(lg.permute(0, 2, 3, 1).contiguous().to(dev))
    assert H.maxdiff(out, ref) < 1e-6


@pytest.mark.parametrize("rad", [3, 4])
def test_nms_bit_exact(dev, golden, rad):
    from pram_amd import ops
    g = golden(f"nms_crafted_r{rad}")
    out = ops.simple_nms(torch.from_numpy(g["score"]).to(dev), rad)
    assert np.array_equal(out.cpu().numpy(), g["nms"])
    # a larger random map with plateaus / exact ties, against the oracle
    s = torch.floor(W.uniform(11, "nms/q", (2, 150, 210), 0.0, 40.0)) / 40.0
    assert torch.equal(ops.simple_nms(s.to(dev), rad).cpu(), R.simple_nms(s, rad))


def test_select_keypoints_bit_exact(dev, golden):
    """fed the reference's score map: keypoints (canonical order) and scores must be identical."""
    from pram_amd import ops
    for tag in ("small_k64", "small_fallback", "small_all"):
        g = golden(f"sfd2_{tag}")
        score = torch.from_numpy(g["score_map"])
        nms_ref = R.simple_nms(score, 4)
        nms = ops.simple_nms(score.to(dev), 4)
        assert torch.equal(nms.cpu(), nms_ref)
        k = int(g["max_keypoints"])
        kp, sc, cnt = ops.select_keypoints(nms, 0.005, int(g["min_keypoints"]), 4, k, fallback_ref=0)
        cnt = cnt.tolist()
        for b in range(2):
            assert cnt[b] == len(g[f"kp{b}"]), (tag, b, cnt[b], len(g[f"kp{b}"]))
            assert np.array_equal(kp[b, :cnt[b]].cpu().numpy().astype(np.int16), g[f"kp{b}"]), (tag, b)
            assert np.array_equal(sc[b, :cnt[b]].cpu().numpy(), g[f"sc{b}"])


def test_select_keypoints_ties_and_flat_map(dev):
    """many exact ties (quantised scores) and > k candidates: canonical (score desc, index asc) order == oracle."""
    from pram_amd import ops
    s = torch.floor(W.uniform(12, "sel/q", (2, 96, 128), 0.0, 16.0)) / 16.0 * 0.5 + 0.01
    for k in (50, 333, 4096):
        kps, scs = R.select_keypoints(s, 0.005, 10, 4, k)
        kp, sc, cnt = ops.select_keypoints(s.to(dev).contiguous(), 0.005, 10, 4, k, fallback_ref=0)
        cnt = cnt.tolist()
        for b in range(2):
            assert cnt[b] == len(kps[b])
            assert torch.equal(kp[b, :cnt[b]].cpu(), kps[b]) and torch.equal(sc[b, :cnt[b]].cpu(), scs[b])
    # empty image
    z = torch.zeros(1, 64, 64)
    kp, sc, cnt = ops.select_keypoints(z.to(dev), 0.005, 10, 4, 100, fallback_ref=0)
    assert cnt.tolist() == [0]


def test_sample_descriptors(dev):
    from pram_amd import ops
    for c, norm in ((128, True), (256, False)):
        fmap = rnd(13, "sa/f", (1, c, 30, 40))
        kp = torch.stack([torch.floor(W.uniform(13, "sa/x", (500,), 0.0, 160.0)), torch.floor(W.uniform(13, "sa/y", (500,), 0.0, 120.0))], -1)
        ref = R.sample_descriptors(kp, fmap, 4, norm)
        out = ops.sample_nhwc(fmap.permute(0, 2, 3, 1).contiguous().to(dev), kp[None].to(dev), None, 4, norm)
        assert H.maxdiff(out[0].t(), ref) < 2e-6
    sm = W.uniform(13, "sa/sm", (1, 120, 160), 0.0, 1.0)
    ref_sc = sm[0, kp[:, 1].long(), kp[:, 0].long()]
    assert torch.equal(ops.score_lookup(sm.to(dev), kp[None].to(dev), None)[0].cpu(), ref_sc)
