"""GPU (MI355X): round-4 additions — ragged fp16 GEMM, the range guard armed by launches (per-model precision inside an f32 process,
the split kernels the fp16 path uses on purpose), the record of a graph replay that fell back, zero-filled ragged heads."""
import pytest
import torch

from pram_amd import ops, weights as W
from pram_amd._lib import PramHipError
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_ragged_fp16_gemm_stores_valid_rows_only_and_stays_on_the_fp16_kernel(dev):
    """linear(precision='f16', lens=...) runs the fp16 GEMM itself (no detour through the split kernels: nothing to report to the
    range guard), equals the dense fp16 result on the valid rows bit for bit and leaves every other row of `out` untouched."""
    T, S, K, N = 192, 5, 256, 320
    x = W.normal(21, "r4/x", (S * T, K), 1.0).to(dev)
    x2 = W.normal(21, "r4/x2", (S * T, K), 1.0).to(dev)
    w = W.normal(21, "r4/w", (N, 2 * K), 0.05).to(dev)
    b = W.normal(21, "r4/b", (N,), 0.1).to(dev)
    res = W.normal(21, "r4/r", (S * T, N), 1.0).to(dev)
    lens = torch.tensor([192, 0, 77, 128, 1], dtype=torch.int32, device=dev)
    ops.x3_range_exceeded(dev)
    dense = ops.linear(x, w, b, x2=x2, residual=res, alpha=0.5, precision="f16")
    out = torch.full((S * T, N), -7.0, device=dev)
    got = ops.linear(x, w, b, x2=x2, residual=res, alpha=0.5, precision="f16", lens=lens, t_pad=T, out=out)
    assert got.data_ptr() == out.data_ptr() and not ops.x3_launched(dev)
    valid = (torch.arange(T, device=dev)[None] < lens[:, None]).reshape(-1)
    assert torch.equal(got[valid], dense[valid])
    assert bool((got[~valid] == -7.0).all())
    # narrow output (the 128-row tile) and a value past the split format's range: fine on this path (fp16 holds 65504)
    wn = W.normal(21, "r4/wn", (48, K), 0.05).to(dev)
    xb = x.clone()
    xb[3, 3] = 5000.0
    outn = torch.full((S * T, 48), 3.0, device=dev)
    gn = ops.linear(xb, wn, precision="f16", lens=lens, t_pad=T, out=outn)
    dn = ops.linear(xb, wn, precision="f16")
    assert torch.equal(gn[valid], dn[valid]) and bool((gn[~valid] == 3.0).all()) and bool(torch.isfinite(gn[valid]).all())
    assert not ops.x3_launched(dev)


def _hot_segnet(dev, gain, n_class=113):
    from pram_amd.nets.load_segnet import load_segnet
    sd = dict(H.segnet_sd())
    sd["input_proj.weight"] = sd["input_proj.weight"] * gain
    sd["input_proj.bias"] = sd["input_proj.bias"] * gain
    m = load_segnet('segnetvit', n_class, 256, 15, 1024)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval(), sd


def _tokens(dev, N, seed=31):
    desc = W.normal(seed, "r4/d", (1, N, 256), 0.05).to(dev)
    kp = torch.stack([torch.floor(W.uniform(seed, "r4/kx", (N,), 4.0, 636.0)), torch.floor(W.uniform(seed, "r4/ky", (N,), 4.0, 476.0))], -1)[None].to(dev)
    return {"seg_descriptors": desc, "keypoints": kp, "image": torch.empty(1, 3, 480, 640)}


def test_guard_sees_a_models_own_split_precision_inside_an_f32_process(dev):
    """Process default f32 (or f16), model.set_precision('x3'), residual stream beyond the split format's range: the outermost
    guarded call — the pipeline's, outside the model's precision scope — must still notice and fall back."""
    from pram_amd.nets.sfd2 import ResNet4x
    from pram_amd.pipeline import GraphedPipeline, QueryPipeline
    net, _ = _hot_segnet(dev, 6.0e3)
    data = _tokens(dev, 192)
    old = ops.gemm_precision, ops.attention_precision
    for base in ("f32", "f16"):
        ops.gemm_precision = ops.attention_precision = base
        try:
            ops.x3_range_exceeded(dev)
            want = net.set_precision("f32")(data)["prediction"]
            assert not ops.x3_launched(dev)                         # an exact-fp32 forward has nothing to report
            net.set_precision("x3").set_act_scale(16.0)
            with ops.guard_scope("raise"):
                with pytest.raises(PramHipError):
                    net(data)
            got = net(data)["prediction"]                           # default policy: lower the model's activation scale, re-run
            assert net.act_scale == 1.0 and bool(torch.isfinite(got).all())
            assert float((got - want).abs().max()) / float(want.abs().max()) < 1e-4
            net.set_act_scale(16.0)
            # ... and through the pipeline, whose guarded call is the outermost one and sits outside every model's scope
            sfd2 = ResNet4x()
            sfd2.load_state_dict(H.sfd2_sd(), strict=True)
            sfd2 = sfd2.to(dev).eval().set_precision("f32")
            pipe = QueryPipeline(sfd2, net, None, max_keypoints=128, min_keypoints=8)
            img = torch.stack([W.synthetic_image(1, 96, 128), W.synthetic_image(2, 96, 128)]).to(dev)
            with pytest.raises(PramHipError):
                pipe.run(img, None, stages="er", guard="raise")
            out = pipe.run(img, None, stages="er")                  # trips, lowers the scales of the models that ran, re-runs
            n = int(out["counts"][0])
            assert net.act_scale == 1.0 and bool(torch.isfinite(out["prediction"][0, :n]).all())
            net.set_act_scale(16.0)
            sfd2.set_act_scale(16.0)
            g = GraphedPipeline(pipe, img, None, stages="er", record=True)      # captured at the default scale: its replay trips
            assert g.uses_x3
            ops.x3_range_exceeded(dev)
            res = g.run(img)
            assert bool(torch.isfinite(res["prediction"][0, :n]).all()) and torch.equal(res["prediction"], out["prediction"])
            # the record a caller reads after run() is the fallback's, not the overflowed replay's
            assert g.record is not None and bool(torch.isfinite(g.record[0, :n]).all())
            assert torch.equal(g.record, QueryPipeline.pack_record(out))
        finally:
            net.set_precision(None)
            ops.gemm_precision, ops.attention_precision = old
    ops.x3_range_exceeded(dev)


def test_fp16_path_matcher_reports_its_split_kernels_to_the_guard(dev):
    """The fp16 path keeps GML's final descriptor projection and score matrix on the split kernels on purpose; their range is
    checked although the process precision says 'f16' (the guard used to look at the settings only)."""
    from pram_amd.nets.gml import GML
    sd = dict(H.gml_sd())
    net = GML({}).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    data = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in H.pair_data(0, 256, 256)[0].items()}
    ops.x3_range_exceeded(dev)
    with ops.guard_scope("deferred"):
        net.set_precision("f16").produce_matches(data)
    assert ops.x3_launched(dev)
    assert not ops.x3_range_exceeded(dev)
    ref = net.set_precision("f16").produce_matches(data)
    # an out-projection that leaves the split format's range (the planes carry out / d^(1/4) * 16)
    hot = dict(sd)
    last = net.n_layers - 1
    hot[f"out_proj.{last}.weight"] = sd[f"out_proj.{last}.weight"] * 4.0e5
    net2 = GML({}).eval()
    net2.load_state_dict(hot, strict=True)
    net2 = net2.to(dev).set_precision("f16")
    with ops.guard_scope("raise"):
        with pytest.raises(PramHipError):
            net2.produce_matches(data)
    out = net2.produce_matches(data)                               # fallback: exact fp32, finite
    assert bool(torch.isfinite(out["matching_scores0"]).all())
    assert ref["matches0"].shape == out["matches0"].shape
    ops.x3_range_exceeded(dev)


def test_ragged_sc_head_is_zero_beyond_a_frames_keypoints(dev):
    """with_sc: the 'sc' rows beyond lens read zero like 'prediction' (they used to be uninitialised memory)."""
    from pram_amd.nets.segnetvit import SegNetViT
    net = SegNetViT({"n_class": 113, "with_sc": True}).eval()
    net.load_state_dict(W.make_state_dict('segnetvit', net.state_dict(), seed=7), strict=True)
    net = net.to(dev)
    N = 128
    d = _tokens(dev, N)
    data = {"seg_descriptors": d["seg_descriptors"].repeat(2, 1, 1), "keypoints": d["keypoints"].repeat(2, 1, 1), "image": torch.empty(2, 3, 480, 640),
            "lens": torch.tensor([N, 40], dtype=torch.int32, device=dev)}
    for prec in ("x3", "f32", "f16"):
        torch.empty(1 << 22, device=dev).fill_(float("nan"))        # poison what the allocator hands out next
        out = net.set_precision(prec)(data)
        assert bool((out["sc"][1, 40:] == 0).all()) and bool((out["prediction"][1, 40:] == 0).all()), prec
        assert bool(torch.isfinite(out["sc"]).all()) and bool(torch.isfinite(out["prediction"]).all()), prec
        full = net({k: (v[:1] if torch.is_tensor(v) and v.shape[0] == 2 else v) for k, v in data.items() if k != "lens"})
        assert float((out["sc"][0] - full["sc"][0]).abs().max()) < (1e-5 if prec != "f16" else 1e-1)
    ops.x3_range_exceeded(dev)


def test_sinkhorn_on_non_finite_scores_never_faults(dev):
    """All-NaN (and partly NaN) score matrices: rows / columns without a comparable value are reported unmatched — the arg-max
    sentinel used to be used as an address in the mutual check (a GPU memory fault under the deferred guard policy)."""
    g = torch.Generator().manual_seed(5)
    for m, n in ((300, 257), (64, 2049)):
        d = torch.randn(2, m, (n + 3) // 4 * 4, generator=g).to(dev)      # rows padded to 16 bytes, n valid columns (as GML hands it over)
        d[0] = float("nan")
        d[1, 3] = float("nan")
        d[1, :, 5] = float("inf")
        bin_score = torch.ones(1, device=dev)
        for dual in (False, True):
            r = ops.sinkhorn_match(d.clone(), bin_score, 20, 0.2, dual_softmax=dual, n_valid=n)
            torch.cuda.synchronize()
            assert bool((r["matches0"][0] == -1).all()) and bool((r["matches1"][0] == -1).all())
            assert int(r["matches0"][1].max()) < n and int(r["matches1"][1].max()) < m


def test_sfd2_backbone_grouped_matrix_pipe_is_fp32_class(dev):
    """conv4's ResBlocks: 1x1 -> planes -> grouped 3x3 on the matrix pipe (the default) against the vector kernel's exact-fp32
    grouped 3x3 (nets/sfd2.py:107-124)."""
    from pram_amd.nets import sfd2 as S
    net = S.ResNet4x()
    net.load_state_dict(H.sfd2_sd(), strict=True)
    net = net.to(dev).eval()
    img = torch.stack([W.synthetic_image(1, 96, 128), W.synthetic_image(2, 96, 128)]).to(dev)
    saved_g = ops.GROUPED_X3
    try:
        ops.GROUPED_X3 = False
        b = net._backbone(img)[-1]
        ops.GROUPED_X3 = True
        c = net._backbone(img)[-1]
    finally:
        ops.GROUPED_X3 = saved_g
    assert float((c - b).abs().max()) <= 4e-6 * float(b.abs().max())


def test_select_keypoints_beyond_the_in_lds_sort(dev):
    """max_keypoints > 8192 (the reference's top_k_keypoints has no bound, nets/sfd2.py:38-50): the top-k is sorted in the workspace
    instead of LDS — same canonical order, bit for bit against the oracle; ties at the cut-off and the fewer-than-k case included."""
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(17)
    Hh, Ww = 200, 256
    nms = torch.rand(2, Hh, Ww, generator=g) * 0.9 + 0.05
    nms[0].view(-1)[::3] = 0.0                                   # ~34 k candidates in frame 0 ...
    nms[1].view(-1)[::2] = 0.0                                   # ... ~25 k in frame 1
    nms[1, 50:60, 50:150] = 0.4321                               # a plateau of equal scores across the cut-off
    for k in (10000, 16384, 20000, 40000):
        kp, sc, cnt = ops.select_keypoints(nms.to(dev), 0.005, 128, 4, k, fallback_ref=-1)
        okp, osc = R.select_keypoints(nms, 0.005, 128, 4, k, per_image_fallback=True)
        for b in range(2):
            n = int(cnt[b])
            assert n == len(osc[b]) == min(k, n), (k, b, n, len(osc[b]))
            assert torch.equal(sc[b, :n].cpu(), osc[b]) and torch.equal(kp[b, :n].cpu(), okp[b]), (k, b)


@pytest.mark.parametrize("shape", [(2, 96, 128), (1, 33, 47), (1, 16, 32), (1, 5, 7), (2, 480, 640)])
def test_fused_conv1(dev, shape):
    """pram_sfd2_conv1_x3_f32 = conv1a -> conv1b (bias, BN, ReLU each; stride 2 on the second): against an fp64 reference of both
    layers (fp32-class), and against the two-kernel path (whose conv1a is the exact-fp32 MFMA kernel: agreement to rounding, not
    bit for bit); even / odd frame sizes, frames smaller than a tile, the bench's 480 x 640; range guard on an out-of-range pixel."""
    B, Hh, Ww = shape
    F = torch.nn.functional
    wa = W.normal(5, "c1/wa", (64, 3, 3, 4), 0.25)
    wa[..., 3] = 0.0                                               # the NHWC4 image's fourth channel does not exist
    wb = W.normal(5, "c1/wb", (64, 3, 3, 64), 0.06)
    par = {k: (W.uniform(5, "c1/" + k, (64,), 0.6, 1.4) if k[0] == "s" else W.normal(5, "c1/" + k, (64,), 0.2)) for k in ("ba", "sa", "ta", "bb", "sb", "tb")}
    img = W.normal(6, f"c1/img{shape}", (B, 3, Hh, Ww), 1.0)
    x4 = ops.image_to_nhwc4(img.to(dev))
    d = {k: v.to(dev) for k, v in par.items()}
    ops.x3_range_exceeded(dev)
    got = ops.sfd2_conv1(x4, wa.to(dev), d["ba"], d["sa"], d["ta"], wb.to(dev), d["bb"], d["sb"], d["tb"])
    assert not ops.x3_range_exceeded(dev)
    y = ops.conv2d_nhwc(x4, wa.to(dev), d["ba"], d["sa"], d["ta"], ks=3, stride=1, relu=True)
    two = ops.conv2d_nhwc(y, wb.to(dev), d["bb"], d["sb"], d["tb"], ks=3, stride=2, relu=True, precision="x3")
    bn = lambda v, s, t: v * par[s].double().view(1, -1, 1, 1) + par[t].double().view(1, -1, 1, 1)
    r = torch.relu(bn(F.conv2d(img.double(), wa[..., :3].double().permute(0, 3, 1, 2), par["ba"].double(), padding=1), "sa", "ta"))
    r = torch.relu(bn(F.conv2d(r, wb.double().permute(0, 3, 1, 2), par["bb"].double(), stride=2, padding=1), "sb", "tb")).permute(0, 2, 3, 1)
    assert tuple(got.shape) == tuple(two.shape) == tuple(r.shape)
    scale = float(r.abs().max())
    e_f, e_2 = float((got.double().cpu() - r).abs().max()), float((two.double().cpu() - r).abs().max())
    assert e_f <= max(2.0 * e_2, 2e-6 * scale), (e_f, e_2, scale)
    assert float((got - two).abs().max()) <= 4e-6 * scale
    # the image as the caller holds it (NCHW, three planes): the same bits without the repack kernel
    direct = ops.sfd2_conv1(img.to(dev).contiguous(), wa.to(dev), d["ba"], d["sa"], d["ta"], wb.to(dev), d["bb"], d["sb"], d["tb"])
    assert torch.equal(direct, got)
    if shape == (2, 96, 128):
        bad = img.clone()
        bad[1, 2, 40, 50] = 5.0e3
        ops.sfd2_conv1(ops.image_to_nhwc4(bad.to(dev)), wa.to(dev), d["ba"], d["sa"], d["ta"], wb.to(dev), d["bb"], d["sb"], d["tb"])
        assert ops.x3_range_exceeded(dev)


def _planes(x):
    """the split operand of a tensor as the kernels compute it: hi = fp16(16 x), lo = fp16(16 x - hi)"""
    x16 = x.float() * 16.0
    hi = x16.half()
    return hi, (x16 - hi.float()).half()


@pytest.mark.parametrize("shape", [(2, 24, 32, 256), (1, 13, 21, 64), (1, 3, 5, 128), (3, 40, 48, 256), (16, 120, 160, 256)])
@pytest.mark.parametrize("bn_relu", [True, False])
def test_grouped_conv_on_the_matrix_pipe(dev, shape, bn_relu):
    """pram_conv3x3_grouped_planes_x3_f32 (block-diagonal split-fp16 product from fp16 planes, windows by LDS-DMA) against the
    exact-fp32 vector kernel and an fp64 reference of the grouped convolution: fp32-class; ragged tiles, frames smaller than a
    tile, more tiles than resident workgroups take (the persistent walk and its double buffer), the bench's 16 x 120 x 160 x 256."""
    B, Hh, Ww, C = shape
    G = C // 8
    w = W.normal(7, f"gx/w{C}", (C, 3, 3, 8), 0.12)
    x = W.normal(8, f"gx/x{shape}", (B, Hh, Ww, C), 1.0)
    sc = W.uniform(7, f"gx/s{C}", (C,), 0.6, 1.4).to(dev) if bn_relu else None
    sh = W.normal(7, f"gx/t{C}", (C,), 0.2).to(dev) if bn_relu else None
    xd, wd = x.to(dev), w.to(dev)
    hi, lo = _planes(xd)
    got = ops.conv3x3_grouped_planes(hi, lo, wd, sc, sh, G, bn_relu)
    valu = ops.conv3x3_grouped_nhwc(xd, wd, sc, sh, G, bn_relu)
    if B * Hh * Ww <= 8192:
        r = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1, groups=G)
        if bn_relu:
            r = torch.relu(r * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1))
        r = r.permute(0, 2, 3, 1)
        scale = float(r.abs().max())
        e_x, e_v = float((got.double().cpu() - r).abs().max()), float((valu.double().cpu() - r).abs().max())
        assert e_x <= max(2.0 * e_v, 2e-6 * scale), (e_x, e_v, scale)
    scale = float(valu.abs().max())
    assert float((got - valu).abs().max()) <= 4e-6 * scale
    again = ops.conv3x3_grouped_planes(hi, lo, wd, sc, sh, G, bn_relu)
    assert torch.equal(got, again)


@pytest.mark.parametrize("shape", [(2, 24, 32, 256, 256), (1, 7, 9, 64, 128), (1, 30, 40, 256, 64)])
def test_conv_writes_the_next_layers_split_operand(dev, shape):
    """pram_conv2d_nhwc_x3_planes = pram_conv2d_nhwc_x3_f32 with the split of the consumer's staging applied in the epilogue:
    the planes are bit for bit the split of the fp32 kernel's result; the range guard trips when 16 y leaves fp16."""
    B, Hh, Ww, Cin, Cout = shape
    w = W.normal(9, f"pl/w{shape}", (Cout, 1, 1, Cin), 0.08)
    x = W.normal(9, f"pl/x{shape}", (B, Hh, Ww, Cin), 1.0)
    sc, sh = W.uniform(9, f"pl/s{Cout}", (Cout,), 0.6, 1.4).to(dev), W.normal(9, f"pl/t{Cout}", (Cout,), 0.2).to(dev)
    xd, wd = x.to(dev), w.to(dev)
    ops.x3_range_exceeded(dev)
    hi, lo = ops.conv2d_nhwc_planes(xd, wd, None, sc, sh, ks=1, relu=True)
    assert not ops.x3_range_exceeded(dev)
    y = ops.conv2d_nhwc(xd, wd, None, sc, sh, ks=1, relu=True, precision="x3")
    eh, el = _planes(y)
    assert torch.equal(hi, eh) and torch.equal(lo, el)
    if Cout == 256:
        ops.conv2d_nhwc_planes(xd, wd, None, sc * 4000.0, sh, ks=1, relu=True)
        assert ops.x3_range_exceeded(dev)


@pytest.mark.parametrize("shape", [(2, 24, 32, 256, 128, 1), (1, 13, 21, 64, 96, 1), (3, 9, 7, 32, 64, 3), (16, 120, 160, 256, 128, 1)])
def test_conv_with_channel_normalisation_in_its_epilogue(dev, shape, monkeypatch):
    """pram_conv2d_nhwc_x3_l2norm_f32 = convolution -> F.normalize over the channels of each pixel in one kernel, against the two
    kernels (same values, another summation order: agreement to rounding) and an fp64 reference; ragged row tiles, narrow outputs,
    a 3 x 3 layer, the descriptor head's own shape; an all-zero pixel stays zero (the 1e-12 floor)."""
    B, Hh, Ww, cin, cout, ks = shape
    x = W.normal(13, f"ln/x{shape}", (B, Hh, Ww, cin), 1.0)
    x[0, 0, 0] = 0.0
    w = W.normal(13, f"ln/w{shape}", (cout, ks, ks, cin), (ks * ks * cin) ** -0.5)
    bias = W.normal(13, f"ln/b{cout}", (cout,), 0.1)
    xd, wd = x.to(dev), w.to(dev)
    bd = None if ks == 3 else bias.to(dev)      # (without a bias the zero pixel of the 3 x 3 case is not zero either: its neighbours)
    monkeypatch.setattr(ops, "FUSED_L2NORM", True)
    got = ops.conv2d_nhwc(xd, wd, bd, ks=ks, precision="x3", l2norm=True)
    monkeypatch.setattr(ops, "FUSED_L2NORM", False)
    two = ops.conv2d_nhwc(xd, wd, bd, ks=ks, precision="x3", l2norm=True)
    assert float((got - two).abs().max()) <= 4e-7
    if B * Hh * Ww <= 4096:
        r = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None if ks == 3 else bias.double(), padding=ks // 2)
        r = torch.nn.functional.normalize(r, dim=1).permute(0, 2, 3, 1)
        assert float((got.double().cpu() - r).abs().max()) <= 2e-6
    n = got.double().pow(2).sum(-1).sqrt()
    assert float((n - 1.0).abs().max()) <= 1e-6 or ks == 1      # unit vectors ...
    if ks == 1:
        wz = torch.zeros_like(wd)
        z = ops.conv2d_nhwc(xd, wz, None, ks=1, precision="x3", l2norm=True)
        assert float(z.abs().max()) == 0.0                        # ... and zero stays zero
