"""GPU (MI355X): round-3 additions — the range guard of the split-fp16 path, the key-split mode of the split-fp16 attention,
ragged GEMMs on the fp16 path (AdaGML), the result-record kernel, RCCL with one rank, the secondary matcher shape."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from pram_amd import ops, weights as W
from pram_amd._lib import PramHipError
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _to(data, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}


# ------------------------------------------------------------------------------------------------ range guard
def test_range_guard_flags_gemm_operands_and_plane_outputs(dev):
    """fp16(16 x) overflows for a finite |x| >= 4094.97: the staging of the split-fp16 GEMM and its plane-writing epilogue report
    it in the status word; in-range values, NaN inputs (which propagate as NaN like the reference's fp32 arithmetic) and the
    exact-fp32 kernels do not."""
    w = W.normal(3, "rg/w", (256, 256), 0.05).to(dev)
    x = W.normal(3, "rg/x", (512, 256), 1.0).to(dev)
    ops.x3_range_exceeded(dev)                                     # clear
    y = ops.linear(x, w, precision="x3")
    assert not ops.x3_range_exceeded(dev) and bool(torch.isfinite(y).all())
    xb = x.clone()
    xb[137, 19] = 4094.0                                           # 16 x = 65504: the largest fp16, still fine
    y = ops.linear(xb, w, precision="x3")
    assert not ops.x3_range_exceeded(dev) and bool(torch.isfinite(y).all())
    xb[137, 19] = 4095.0                                           # 16 x = 65520 rounds to +inf
    y = ops.linear(xb, w, precision="x3")
    assert ops.x3_range_exceeded(dev)                              # reports and resets
    assert not ops.x3_range_exceeded(dev)
    assert not bool(torch.isfinite(y[137]).all())                  # what the caller would otherwise have got: NaN in that row
    y32 = ops.linear(xb, w, precision="f32")
    assert bool(torch.isfinite(y32).all()) and not ops.x3_range_exceeded(dev)
    xn = x.clone()
    xn[5, 5] = float("nan")
    y = ops.linear(xn, w, precision="x3")
    assert not ops.x3_range_exceeded(dev) and bool(torch.isnan(y[5]).all()) and bool(torch.isfinite(y[6]).all())
    xi = x.clone()
    xi[9, 0] = float("inf")
    ops.linear(xi, w, precision="x3")
    assert ops.x3_range_exceeded(dev)
    # plane-writing epilogue: in-range operands, out-of-range RESULT (alpha pushes it past 4094.97)
    _, pl = ops.linear(x, w, alpha=3.0e4, split_out="only", precision="x3")
    assert ops.x3_range_exceeded(dev)
    _, pl = ops.linear(x, w, alpha=1.0, split_out="only", precision="x3")
    assert not ops.x3_range_exceeded(dev)
    # convolution staging
    img = W.normal(4, "rg/img", (1, 24, 32, 64), 1.0).to(dev)
    wc = W.normal(4, "rg/wc", (64, 3, 3, 64), 0.05).to(dev)
    ops.conv2d_nhwc(img, wc, precision="x3")
    assert not ops.x3_range_exceeded(dev)
    img[0, 3, 3, 3] = -5.0e3
    ops.conv2d_nhwc(img, wc, precision="x3")
    assert ops.x3_range_exceeded(dev)


def _hot_segnet(dev, gain):
    from pram_amd.nets.load_segnet import load_segnet
    sd = dict(H.segnet_sd())
    sd["input_proj.weight"] = sd["input_proj.weight"] * gain        # residual stream O(gain): no normalisation on it (segnetvit.py:105-106)
    sd["input_proj.bias"] = sd["input_proj.bias"] * gain
    m = load_segnet('segnetvit', 113, 256, 15, 1024)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval(), sd


def test_range_guard_lowers_the_activation_scale_on_a_hot_residual_stream(dev, monkeypatch):
    """A SegNetViT whose residual stream leaves the range of the default planes (|x| >= 4094.97 at scale 16): the default policy
    lowers the MODEL's activation scale (16 -> 1: |x| < 65520) and re-runs on the split kernels — no exact-fp32 fallback, the
    result in the split path's own accuracy class against the oracle — and the model keeps the scale: the next call does not
    trip at all.  "raise" raises; "deferred" hands the NaN-derived output back with the flag set — never a silently wrong result."""
    net, sd = _hot_segnet(dev, 6.0e3)
    N = 192
    desc = (W.normal(11, "rg/d", (1, N, 256), 0.05)).to(dev)
    kp = torch.stack([torch.floor(W.uniform(11, "rg/x", (N,), 4.0, 636.0)), torch.floor(W.uniform(11, "rg/y", (N,), 4.0, 476.0))], -1)[None].to(dev)
    data = {"seg_descriptors": desc, "keypoints": kp, "image": torch.empty(1, 3, 480, 640)}
    ops.x3_range_exceeded(dev)
    with ops.guard_scope("deferred"):
        raw = net.set_precision("x3")(data)["prediction"]
    assert ops.x3_range_exceeded(dev, reset=True), "fixture must leave the range (else the test proves nothing)"
    assert not bool(torch.isfinite(raw).all())
    with ops.guard_scope("raise"):
        with pytest.raises(PramHipError):
            net(data)
    assert net.act_scale == 16.0 and ops.current_act_scale() == 16.0
    ev0 = dict(ops.guard_events)
    with ops.guard_scope("fallback"):
        got = net(data)["prediction"]
    assert net.act_scale == 1.0, net.act_scale                                    # lowered, and kept
    assert ops.guard_events["rescaled"] == ev0["rescaled"] + 1 and ops.guard_events["f32_fallback"] == ev0["f32_fallback"]
    assert ops.current_act_scale() == 16.0                                        # the thread's own setting is untouched outside the model
    assert bool(torch.isfinite(got).all())
    o = R.segnetvit_forward(sd, desc.cpu(), kp.cpu(), (1, 3, 480, 640))
    peak = float(o.abs().max())
    rel = float((got.cpu() - o).abs().max()) / peak
    agree = float((got.cpu().argmax(-1) == o.argmax(-1)).float().mean())
    ref32 = net.set_precision("f32")(data)["prediction"]
    rel32 = float((ref32.cpu() - o).abs().max()) / peak
    print(f"hot SegNetViT (gain 6e3, |logit| max {peak:.3g}): split path at act_scale 1: rel {rel:.2e}, arg-max {agree:.4f}; exact-fp32 kernels rel {rel32:.2e}")
    # activations of ~10^4 behind logits of ~10: the planes' 22-bit significand shows as ~4e-5 of the logits' peak (4.6e-4 absolute,
    # inside the 1e-3 parity bar; the exact-fp32 kernels' 24 bits give 1e-6)
    assert rel < 1e-4 and rel * peak < 1e-3 and agree == 1.0, (rel, agree)
    ev1 = dict(ops.guard_events)
    with ops.guard_scope("fallback"):
        again = net.set_precision("x3")(data)["prediction"]
    assert torch.equal(again, got) and ops.guard_events == ev1                   # sticky: no trip, no re-run, same bits
    net.set_precision(None)
    # the same thing up front: calibrate on a sample before serving / capturing
    net_c, _ = _hot_segnet(dev, 6.0e3)
    assert net_c.set_precision("x3").calibrate_act_scale(lambda: net_c(data)) == 1.0 and net_c.act_scale == 1.0
    ev2 = dict(ops.guard_events)
    with ops.guard_scope("raise"):
        assert torch.equal(net_c(data)["prediction"], got)                       # in range now: nothing to raise about, the same bits
    assert ops.guard_events == ev2
    cool, _ = _hot_segnet(dev, 1.0)
    assert cool.calibrate_act_scale(lambda: cool(data)) == 16.0
    # a stream the allowed scales cannot carry ends on the exact-fp32 kernels, as before (here: scales limited to >= 1, |x| ~ 7e4, attention logits ~ 1e9)
    monkeypatch.setattr(ops, "ACT_SCALE_MIN", 1.0)
    net3, _ = _hot_segnet(dev, 6.0e4)
    with ops.guard_scope("fallback"):
        g3 = net3.set_precision("x3")(data)["prediction"]
    assert ops.guard_events["f32_fallback"] == ev1["f32_fallback"] + 1 and net3.act_scale == 1.0
    assert bool(torch.isfinite(g3).all()) and torch.equal(g3, net3.set_precision("f32")(data)["prediction"])
    monkeypatch.undo()
    # an in-range model is untouched by the guard (same bits as with the guard off)
    net2, _ = _hot_segnet(dev, 1.0)
    with ops.guard_scope("deferred"):
        a = net2(data)["prediction"]
    with ops.guard_scope("fallback"):
        b = net2(data)["prediction"]
    assert torch.equal(a, b) and not ops.x3_range_exceeded(dev) and net2.act_scale == 16.0


def test_activation_scale_is_exact_rescaling_of_the_planes(dev):
    """The activation scale is a power of two: in-range data gives the same fp32 GEMM bits under scale 16 and scale 4 (both parts
    stay normal fp16 numbers), planes written under a scale are consumed under it, and a scale that is not a power of two in
    [2^-12, 16] is refused."""
    w = W.normal(3, "as/w", (256, 256), 0.05).to(dev)
    x = W.normal(3, "as/x", (512, 256), 1.0).to(dev)
    y16 = ops.linear(x, w, precision="x3")
    with ops.act_scale_scope(4.0):
        assert ops.current_act_scale() == 4.0
        y4 = ops.linear(x, w, precision="x3")
        _, pl = ops.linear(x, w, split_out="only", precision="x3")
        z4 = ops.linear_planes(pl, w)
    assert ops.current_act_scale() == 16.0
    _, pl16 = ops.linear(x, w, split_out="only", precision="x3")
    z16 = ops.linear_planes(pl16, w)
    ref = (x.double() @ w.double().t()).float()
    assert float((y16 - ref).abs().max()) < 2e-5 and float((y4 - y16).abs().max()) < 2e-6
    ref2 = (ref.double() @ w.double().t()).float()
    assert float((z4 - ref2).abs().max()) < 2e-5 and float((z16 - ref2).abs().max()) < 2e-5
    assert torch.equal(pl[0].float() * 4.0, pl16[0].float()) or float((pl[0].float() * 4.0 - pl16[0].float()).abs().max()) <= float(pl16[0].float().abs().max()) * 2 ** -10
    big = x * 3.0e4                                                  # |x| up to ~1.3e5: beyond scale 16 and scale 1, inside 1/16
    ops.x3_range_exceeded(dev)
    ops.linear(big, w, precision="x3")
    assert ops.x3_range_exceeded(dev)
    with ops.act_scale_scope(1.0 / 16.0):
        yb = ops.linear(big, w, precision="x3")
    assert not ops.x3_range_exceeded(dev)
    refb = (big.double() @ w.double().t()).float()
    assert float((yb - refb).abs().max()) / float(refb.abs().max()) < 2e-6
    with pytest.raises(PramHipError):
        with ops.act_scale_scope(3.0):
            pass
    assert ops.current_act_scale() == 16.0


@pytest.mark.parametrize("gain,seed", [(30.0, 7), (300.0, 7), (1500.0, 21), (3000.0, 33)])
def test_default_path_keeps_its_accuracy_class_up_to_the_range_edge(dev, gain, seed):
    """The parity fixtures live at O(10) activations (seed-7 synthetic weights); a trained checkpoint need not.  SegNetViT with its
    input projection scaled so that the residual stream reaches 10^2 ... a few 10^3 — still inside the split format's |x| < 4094.97 —
    with other input seeds: no range flag, no fallback, and the logits stay in the fp32 class relative to their own magnitude (the split is
    a floating-point format: its error scales with the values)."""
    net, sd = _hot_segnet(dev, gain)
    N = 256
    desc = (W.normal(seed, "edge/d", (1, N, 256), 0.05)).to(dev)
    kp = torch.stack([torch.floor(W.uniform(seed, "edge/x", (N,), 4.0, 636.0)), torch.floor(W.uniform(seed, "edge/y", (N,), 4.0, 476.0))], -1)[None].to(dev)
    data = {"seg_descriptors": desc, "keypoints": kp, "image": torch.empty(1, 3, 480, 640)}
    ops.x3_range_exceeded(dev)
    with ops.guard_scope("deferred"):
        got = net.set_precision("x3")(data)["prediction"]
    hit = ops.x3_range_exceeded(dev, reset=True)
    net.set_precision(None)
    o = R.segnetvit_forward(sd, desc.cpu(), kp.cpu(), (1, 3, 480, 640))
    peak = float(o.abs().max())
    if hit:      # the fixture left the range after all (gain x seed): then it must be the guard's case, not a silent one
        assert not bool(torch.isfinite(got).all())
        pytest.skip(f"gain {gain}: activations beyond 4094.97 (guard case, covered above)")
    rel = float((got.cpu() - o).abs().max()) / peak
    agree = float((got.cpu().argmax(-1) == o.argmax(-1)).float().mean())
    print(f"gain {gain:g} seed {seed}: |logit| max {peak:.3g}, relative error {rel:.2e}, arg-max agreement {agree:.4f}")
    assert rel < 2e-5 and agree == 1.0, (rel, agree)


def test_nan_and_inf_inputs_behave_like_the_reference(dev):
    """NaN / Inf in the descriptors: the reference's fp32 arithmetic spreads non-finite values over the whole output (attention
    mixes every token into every other); so does this path, through whichever kernels the guard picks — never finite garbage."""
    net, sd = _hot_segnet(dev, 1.0)
    N = 128
    desc = W.normal(12, "nn/d", (1, N, 256), 0.05)
    kp = torch.stack([torch.floor(W.uniform(12, "nn/x", (N,), 4.0, 636.0)), torch.floor(W.uniform(12, "nn/y", (N,), 4.0, 476.0))], -1)[None]
    for bad in (float("nan"), float("inf")):
        d = desc.clone()
        d[0, 7, 3] = bad
        o = R.segnetvit_forward(sd, d, kp, (1, 3, 480, 640))
        got = net({"seg_descriptors": d.to(dev), "keypoints": kp.to(dev), "image": torch.empty(1, 3, 480, 640)})["prediction"].cpu()
        assert not bool(torch.isfinite(o).any()) and not bool(torch.isfinite(got).any()), bad
    ops.x3_range_exceeded(dev)


def test_pipeline_guard_and_graph_replay(dev):
    """QueryPipeline reads the guard once per run; a replayed hipGraph checks after the replay and re-runs eagerly under the
    pipeline's policy (lower activation scale first, exact fp32 last)."""
    from pram_amd.nets.gml import GML
    from pram_amd.nets.sfd2 import ResNet4x
    from pram_amd.pipeline import GraphedPipeline, QueryPipeline
    sfd2 = ResNet4x()
    sfd2.load_state_dict(H.sfd2_sd(), strict=True)
    seg, _ = _hot_segnet(dev, 6.0e3)
    pipe = QueryPipeline(sfd2.to(dev).eval(), seg, None, max_keypoints=128, min_keypoints=8)
    img = torch.stack([W.synthetic_image(1, 96, 128), W.synthetic_image(2, 96, 128)]).to(dev)
    ops.x3_range_exceeded(dev)
    with pytest.raises(PramHipError):
        pipe.run(img, None, stages="er", guard="raise")
    raw = pipe.run(img, None, stages="er", guard="deferred")
    n = int(raw["counts"][0])
    assert ops.x3_range_exceeded(dev) and not bool(torch.isfinite(raw["prediction"][0, :n]).all())
    # a graph captured while the models still carry the default scale: its replay trips, run() re-runs eagerly under the
    # pipeline's policy — which lowers the scale of the models that ran (the recogniser's 16 -> 1; the extractor's too: the
    # guard cannot tell which model tripped) — and returns that result
    g = GraphedPipeline(pipe, img, None, stages="er")
    assert seg.act_scale == 16.0
    ops.x3_range_exceeded(dev)
    got = g.run(img)
    assert seg.act_scale == 1.0 and not ops.x3_range_exceeded(dev)
    with ops.forced_precision("f32"):
        want = pipe.run(img, None, stages="er")["prediction"]
    peak = float(want[0, :n].abs().max())
    assert bool(torch.isfinite(got["prediction"][0, :n]).all())
    assert float((got["prediction"][0, :n] - want[0, :n]).abs().max()) / peak < 1e-4
    # the eager pipeline now runs in range on the split kernels: no trip, no re-run
    ev = dict(ops.guard_events)
    out = pipe.run(img, None, stages="er")
    assert ops.guard_events == ev and not ops.x3_range_exceeded(dev)
    assert float((out["prediction"][0, :n] - want[0, :n]).abs().max()) / peak < 1e-4
    # ... and a graph captured NOW carries the lowered scale: its replays stay on the split path
    g2 = GraphedPipeline(pipe, img, None, stages="er")
    got2 = g2.run(img)
    assert ops.guard_events == ev and torch.equal(got2["prediction"], out["prediction"])


# ------------------------------------------------------------------------------------------------ probabilities in P V
def test_default_path_logits_and_landmarks_over_several_token_sets(dev):
    """SegNetViT at 2048 tokens on four synthetic token sets (flat attention: the case in which rounding every probability to ONE
    fp16 shows): the default path keeps the logits within 1e-4 of the fp32 oracle and every landmark arg-max; the opt-in
    one-fp16 mode (pram_attention_x3_set_p_split(0)) is an order of magnitude further out and sits AT the 1e-3 bar (7e-4 ... 1.3e-3
    depending on where the running maximum happens to round each probability: it is not a parity configuration)."""
    from tests.test_gpu_configs import _segnet, _tokens
    L = ops._lib.load()
    assert L.pram_attention_x3_set_p_split(-1) == 1, "two-part probabilities are the default"
    net = _segnet(dev, 113)
    worst = {1: 0.0, 0: 0.0}
    for idx in (0, 1, 2, 3):
        desc, kp = _tokens(2048, idx=idx)
        ref = R.segnetvit_forward(H.segnet_sd(113), desc, kp, (1, 3, 480, 640))
        data = {"seg_descriptors": desc.to(dev), "keypoints": kp.to(dev), "image": torch.empty(1, 3, 480, 640)}
        out = net(data)["prediction"].cpu()
        d = float((out - ref).abs().max())
        worst[1] = max(worst[1], d)
        assert d < 1e-4 and bool((out.argmax(-1) == ref.argmax(-1)).all()), (idx, d)
        try:
            L.pram_attention_x3_set_p_split(0)
            out1 = net(data)["prediction"].cpu()
        finally:
            L.pram_attention_x3_set_p_split(1)
        worst[0] = max(worst[0], float((out1 - ref).abs().max()))
    print(f"SegNetViT N=2048 nc113, 4 token sets: |logit - oracle| two-part P {worst[1]:.2e}, one-fp16 P {worst[0]:.2e}")
    assert 5 * worst[1] < worst[0] < 2.5e-3


# ------------------------------------------------------------------------------------------------ key-split attention
@pytest.fixture
def chunk_keys():
    """Sets the process-wide key-chunk size of the split-fp16 attention for one test and restores the default afterwards."""
    L = ops._lib.load()
    default = L.pram_attention_x3_set_chunk_keys(0)

    def set_(keys):
        assert L.pram_attention_x3_set_chunk_keys(keys) == keys
    yield set_
    L.pram_attention_x3_set_chunk_keys(default)


@pytest.mark.parametrize("shape", [(1, 2048, 2048, 512), (2, 1500, 1100, 512), (1, 2048, 2048, 1024), (1, 1024, 4096, 2048), (3, 640, 1030, 512),
                                   (1, 2048, 4096, 1024)])
def test_attention_x3_split_equals_fused(dev, shape, chunk_keys):
    """Under-filled launches run groups of key chunks as a second grid dimension + a fold kernel; fused launches fold the same
    chunks in registers, in the same order: bit-identical outputs and log-sum-exps (so a padded batch element still equals its
    B = 1 run), at every chunk size."""
    S, M, N, ck = shape
    chunk_keys(ck)
    T = max(M, N)
    Tp = (T + 63) // 64 * 64
    x = W.normal(31, f"sp/x{S}{M}{N}", (S * Tp, 256), 1.0).to(dev)
    wq = W.normal(31, "sp/w", (768, 256), 0.06).to(dev)
    ql = torch.tensor([M - 17 * i for i in range(S)], dtype=torch.int32, device=dev)
    kl = torch.tensor([N - 29 * i for i in range(S)], dtype=torch.int32, device=dev)
    lens = torch.maximum(ql, kl)
    pl, vt = ops.linear_qkv_planes(x, wq, None, 4, Tp, lens=lens)
    q3, k3 = (pl[0][:, :256], pl[1][:, :256]), (pl[0][:, 256:512], pl[1][:, 256:512])
    L = ops._lib.load()
    assert L.pram_attention_x3_is_split(S, 4, Tp, Tp) > 1, "shape must qualify for the split mode"
    saved = ops.attention_split
    try:
        ops.attention_split = True
        o1, l1 = ops.attention_x3(q3, k3, vt, S, 4, Tp, Tp, 0.125, ql, kl, want_lse=True)          # split: chunks as a grid dimension
        L.pram_attention_x3_set_split_target(0)
        o2, l2 = ops.attention_x3(q3, k3, vt, S, 4, Tp, Tp, 0.125, ql, kl, want_lse=True)          # fused (a workspace is offered, not used)
        ops.attention_split = False
        o3, l3 = ops.attention_x3(q3, k3, vt, S, 4, Tp, Tp, 0.125, ql, kl, want_lse=True)          # fused, no workspace
    finally:
        ops.attention_split = saved
        L.pram_attention_x3_set_split_target(-1)
    for s in range(S):
        m = int(ql[s])
        for o, l in ((o2, l2), (o3, l3)):
            assert torch.equal(o1[s * Tp:s * Tp + m], o[s * Tp:s * Tp + m]), s
            assert torch.equal(l1[s, :, :m], l[s, :, :m]), s
    # values: against fp64 soft-max attention on the same (split) operands, V recovered from the row-major value planes of an
    # ordinary projection call
    _, plv = ops.linear(x, wq, None, split_out="only", lens=lens, t_pad=Tp)
    full = (plv[0].double() + plv[1].double()) / 16
    for s in range(S):
        m, n = int(ql[s]), int(kl[s])
        qq = full[s * Tp:s * Tp + m, :256].view(m, 4, 64).transpose(0, 1)
        kk = full[s * Tp:s * Tp + n, 256:512].view(n, 4, 64).transpose(0, 1)
        vv = full[s * Tp:s * Tp + n, 512:].view(n, 4, 64).transpose(0, 1)
        want = (torch.softmax(qq @ kk.transpose(1, 2) * 0.125, -1) @ vv).transpose(0, 1).reshape(m, 256)
        d = float((o1[s * Tp:s * Tp + m].double() - want).abs().max())
        assert d < 2e-4, (s, d)


@pytest.mark.parametrize("p_split", [1, 0])
def test_x3_attention_many_workgroups_every_mode(dev, chunk_keys, p_split):
    """640 workgroups (two resident per CU, 2560 waves) x four 512-key chunks, repeated: every sequence against fp64 soft-max
    attention, fused and split, with the probabilities as two fp16 parts and as one.  A fused variant that folded parked chunks
    after the last tile failed exactly this launch on ~1 wave in 2500 (one output register, 16 lanes) and nothing smaller."""
    S, T = 10, 2048
    chunk_keys(512)
    L = ops._lib.load()
    x = W.normal(31, "mw/x", (S * T, 256), 1.0).to(dev)
    wq = W.normal(31, "mw/w", (768, 256), 0.06).to(dev)
    pl, vt = ops.linear_qkv_planes(x, wq, None, 4, T)
    q3, k3 = (pl[0][:, :256], pl[1][:, :256]), (pl[0][:, 256:512], pl[1][:, 256:512])
    _, plv = ops.linear(x, wq, None, split_out="only")
    full = (plv[0].double() + plv[1].double()) / 16
    want = torch.empty(S * T, 256, dtype=torch.float64, device=dev)
    for s in range(S):
        qq, kk, vv = (full[s * T:(s + 1) * T, c * 256:(c + 1) * 256].view(T, 4, 64).transpose(0, 1) for c in range(3))
        want[s * T:(s + 1) * T] = (torch.softmax(qq @ kk.transpose(1, 2) * 0.125, -1) @ vv).transpose(0, 1).reshape(T, 256)
    prev = L.pram_attention_x3_set_p_split(-1)
    try:
        L.pram_attention_x3_set_p_split(p_split)
        outs = []
        for target in (0, 4096):          # 0: never split (fused); 4096: four key groups of one chunk each + the fold kernel
            L.pram_attention_x3_set_split_target(target)
            assert (L.pram_attention_x3_is_split(S, 4, T, T) > 1) == (target > 0)
            for rep in range(12):
                o = ops.attention_x3(q3, k3, vt, S, 4, T, T, 0.125)
                d = float((o.double() - want).abs().max())
                assert d < (2e-6 if p_split else 2e-4), (target, rep, d)
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
    finally:
        L.pram_attention_x3_set_split_target(-1)
        L.pram_attention_x3_set_p_split(prev)


def test_x3_attention_tile_loop_forms_agree_bit_for_bit(dev):
    """attention_x3_pipe_kernel has two forms of its tile loop (attention_x3.hip: vector / matrix PHASES across the two waves of a SIMD
    for launches that fill the chip, the INTERLEAVED form for grids of one wave per SIMD) with the same MFMA order per accumulator:
    16 sequences of 2048 keys in one launch (256-row workgroups, phases) give every sequence exactly the bits of its own launch
    (128-row workgroups, interleaved), ragged lengths included; and both sit on the fp64 soft-max."""
    S, T = 16, 2048
    x = W.normal(31, "forms/x", (S * T, 256), 1.0).to(dev)
    wq = W.normal(31, "forms/w", (768, 256), 0.06).to(dev)
    lens = torch.tensor([T - 37 * (i % 5) for i in range(S)], dtype=torch.int32, device=dev)
    pl, vt = ops.linear_qkv_planes(x, wq, None, 4, T, lens=lens)
    q3, k3 = (pl[0][:, :256], pl[1][:, :256]), (pl[0][:, 256:512], pl[1][:, 256:512])
    o_all = ops.attention_x3(q3, k3, vt, S, 4, T, T, 0.125, lens, lens)
    _, plv = ops.linear(x, wq, None, split_out="only", lens=lens, t_pad=T)
    full = (plv[0].double() + plv[1].double()) / 16
    for s in range(0, S, 3):
        n = int(lens[s])
        sl = slice(s * T, (s + 1) * T)
        o_one = ops.attention_x3((q3[0][sl], q3[1][sl]), (k3[0][sl], k3[1][sl]), tuple(v[s:s + 1].contiguous() for v in vt),
                                 1, 4, T, T, 0.125, lens[s:s + 1], lens[s:s + 1])
        assert torch.equal(o_all[s * T:s * T + n], o_one[:n]), s
        qq, kk, vv = (full[s * T:s * T + n, c * 256:(c + 1) * 256].view(n, 4, 64).transpose(0, 1) for c in range(3))
        want = (torch.softmax(qq @ kk.transpose(1, 2) * 0.125, -1) @ vv).transpose(0, 1).reshape(n, 256)
        assert float((o_all[s * T:s * T + n].double() - want).abs().max()) < 2e-6, s


@pytest.mark.parametrize("ck", [512, 4096])
def test_batched_matcher_equals_b1_with_split_attention(dev, chunk_keys, ck):
    """GML on 2048-keypoint sets: a batch of 5 pairs (fused attention launches) gives every pair exactly the bits of its own B = 1
    call (split attention launches with 512-key chunks; one chunk per sequence at the default 4096)."""
    from pram_amd.nets.gml import GML
    chunk_keys(ck)
    net = GML({})
    net.load_state_dict(H.gml_sd(), strict=True)
    net = net.to(dev).eval()
    ds = [H.pair_data(i, 2048, 2048, device=dev)[0] for i in (1, 2, 3, 4, 5)]
    cat = {k: torch.cat([d[k] for d in ds], 0) for k in ds[0] if torch.is_tensor(ds[0][k])}
    cat["image_shape0"] = cat["image_shape1"] = (1, 3, 640, 480)
    rb = net.produce_matches(cat, p=0.0)
    for i, d in enumerate(ds):
        r1 = net.produce_matches(d, p=0.0)
        assert torch.equal(rb["matches0"][i], r1["matches0"][0]) and torch.equal(rb["matching_scores0"][i], r1["matching_scores0"][0]), i


# ------------------------------------------------------------------------------------------------ ragged GEMM on the fp16 path
def test_adagml_fp16_path_commits_only_the_stopping_pairs(dev):
    """AdaGML under precision 'f16': pairs stop at different layers and their matching descriptors are committed by a RAGGED
    GEMM into a persistent buffer.  The fp16 GEMM has no ragged mode (it would overwrite the rows of pairs that stopped
    earlier with projections of stale tokens), so ragged calls run on a kernel that honours lens: the batched call equals the
    B = 1 calls pair by pair, and agrees with the default path on (almost) all matches."""
    from pram_amd.nets.adagml import AdaGML
    sd = dict(H.adagml_sd())
    for k in list(sd):
        if k.endswith("predict.3.bias"):
            sd[k] = sd[k] + 0.08                                   # pairs clear check_if_stop before the last layer
    net = AdaGML({})
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    ds = [H.pair_data(i, 448, 448, device=dev)[0] for i in (1, 2, 3, 4)]
    cat = {k: torch.cat([d[k] for d in ds], 0) for k in ds[0] if torch.is_tensor(ds[0][k])}
    cat["image_shape0"] = cat["image_shape1"] = (1, 3, 640, 480)
    px = {}
    rx = net.set_precision("x3").produce_matches(cat, p=0.0, probes=px)
    pb = {}
    rb = net.set_precision("f16").produce_matches(cat, p=0.0, probes=pb)
    stops = [int(s) for s in pb["stop_layer"]]
    print("adagml f16 stop layers", stops, "x3", [int(s) for s in px["stop_layer"]])
    assert len(set(stops)) > 1, "fixture should stop the pairs at different layers"
    for i, d in enumerate(ds):
        p1 = {}
        r1 = net.produce_matches(d, p=0.0, probes=p1)
        assert int(p1["stop_layer"][0]) == stops[i]
        assert torch.equal(rb["matches0"][i], r1["matches0"][0]), i
        assert H.maxdiff(rb["matching_scores0"][i], r1["matching_scores0"][0]) == 0.0
    same = float((rb["matches0"] == rx["matches0"]).float().mean())
    assert same > 0.9, same
    net.set_precision(None)


def test_ragged_linear_on_the_fp16_path_leaves_other_rows_untouched(dev):
    x = W.normal(41, "rl/x", (4 * 128, 256), 1.0).to(dev)
    w = W.normal(41, "rl/w", (256, 256), 0.06).to(dev)
    lens = torch.tensor([128, 0, 37, 0], dtype=torch.int32, device=dev)
    out = torch.full((4 * 128, 256), 7.0, device=dev)
    ops.linear(x, w, out=out, lens=lens, t_pad=128, precision="f16")
    assert float((out[128:256] - 7).abs().max()) == 0.0 and float((out[384:] - 7).abs().max()) == 0.0
    assert float((out[256 + 37:384] - 7).abs().max()) == 0.0
    # the fp16 GEMM itself since round 4 (it used to detour through the split kernels): the dense fp16 result, bit for bit
    dense = ops.linear(x, w, precision="f16")
    assert torch.equal(out[:128], dense[:128]) and torch.equal(out[256:256 + 37], dense[256:256 + 37])
    want = x.double() @ w.double().t()
    assert float((out[:128].double() - want[:128]).abs().max()) < 5e-3


# ------------------------------------------------------------------------------------------------ MLP tail as a GEMM pair
@pytest.mark.parametrize("shape", [(16384, 256, 256, 512, 256), (2048, 256, 256, 512, 256), (4096, 256, 0, 1024, 113), (20000, 256, 256, 512, 256),
                                   (300, 128, 0, 256, 96)])
def test_mlp_tail_pair_vs_fp64_and_vs_three_kernels(dev, shape):
    """Linear -> LayerNorm -> GELU -> Linear + residual as two kernels (centred first GEMM + row sums of squares; LayerNorm + GELU
    applied to the second GEMM's operand while staged, GELU through a fitted Gaussian tail) against an fp64 reference and against the three-kernel
    path (stand-alone LayerNorm + GELU with exact erff), wide and narrow tiles, ragged rows."""
    m, k0, k1, hid, n = shape
    x = W.normal(61, f"mt/x{m}", (m, k0), 1.0).to(dev)
    x2 = W.normal(61, f"mt/x2{m}", (m, k1), 1.0).to(dev) if k1 else None
    w0 = W.normal(61, "mt/w0", (hid, k0 + k1), (k0 + k1) ** -0.5).to(dev)
    b0 = (W.normal(61, "mt/b0", (hid,), 0.3) + 0.7).to(dev)              # a common offset: what the centring removes
    g = (1.0 + W.normal(61, "mt/g", (hid,), 0.2)).to(dev)
    bt = W.normal(61, "mt/bt", (hid,), 0.2).to(dev)
    w3 = W.normal(61, "mt/w3", (n, hid), hid ** -0.5).to(dev)
    b3 = W.normal(61, "mt/b3", (n,), 0.1).to(dev)
    res = W.normal(61, f"mt/r{m}", (m, n), 1.0).to(dev)
    w0c, b0c = [t.to(dev) for t in ops.center_linear(w0, b0)]
    got = ops.mlp_tail(x, w0c, b0c, g, bt, w3, b3, x2=x2, residual=res)
    h = ops.linear(x, w0, b0, x2=x2, precision="x3")
    ops.layernorm_gelu_(h, g, bt)
    three = ops.linear(h, w3, b3, residual=res, precision="x3")
    xa = torch.cat([x, x2], 1).double() if k1 else x.double()
    hh = torch.nn.functional.layer_norm(xa @ w0.double().t() + b0.double(), (hid,), g.double(), bt.double(), 1e-5)
    want = torch.nn.functional.gelu(hh) @ w3.double().t() + b3.double() + res.double()
    d_pair, d_three = float((got.double() - want).abs().max()), float((three.double() - want).abs().max())
    print(f"mlp tail {shape}: pair vs fp64 {d_pair:.2e}, three kernels vs fp64 {d_three:.2e}")
    assert d_pair < 2e-5 and d_pair < 3 * d_three + 2e-6
    # ragged: rows beyond their sequence's length are left untouched, the others are the same bits as the full call
    if m % 4 == 0:
        t_pad = m // 4
        lens = torch.tensor([t_pad, 0, t_pad - 37, 5], dtype=torch.int32, device=dev)
        out = torch.full((m, n), 7.0, device=dev)
        ops.mlp_tail(x, w0c, b0c, g, bt, w3, b3, x2=x2, residual=res, lens=lens, t_pad=t_pad, out=out)
        for s_, ln in enumerate(lens.tolist()):
            a = s_ * t_pad
            assert torch.equal(out[a:a + ln], got[a:a + ln]) and float((out[a + ln:a + t_pad] - 7).abs().max() if ln < t_pad else 0.0) == 0.0, s_


def test_erf_of_the_fused_gelu_against_erff(dev):
    """The fitted Gaussian tail inside the fused GELU (7.5e-8 |t|), pinned through the kernel: an identity first GEMM feeds a dense grid of hidden
    values through LayerNorm-free scaling (gamma = std, beta = mean of the grid undo the normalisation), the second GEMM is an
    identity too — the output is GELU(grid), compared with torch's erf-based GELU in fp64."""
    hid = 256
    grid = torch.linspace(-9.0, 9.0, 64 * hid, dtype=torch.float64).view(64, hid)
    mu, sd = grid.mean(1, keepdim=True), grid.std(1, unbiased=False, keepdim=True)
    # rows of the grid as inputs of an identity Linear: after centring the first GEMM emits grid - mean; the LayerNorm divides by
    # sqrt(var + eps); gamma / beta are per column, so each ROW is restored only up to its own (mu, sd): use one row at a time
    eye = torch.eye(hid, device=dev)
    zero = torch.zeros(hid, device=dev)
    worst = 0.0
    for r in range(0, 64, 7):
        row = grid[r:r + 1].float().to(dev).repeat(64, 1).contiguous()
        w0c, b0c = [t.to(dev) for t in ops.center_linear(eye, zero)]
        g = torch.full((hid,), float(torch.sqrt(sd[r] ** 2 + 1e-5)), device=dev)
        bt = torch.full((hid,), float(mu[r]), device=dev)
        got = ops.mlp_tail(row, w0c, b0c, g, bt, eye, None)[0].double().cpu()
        want = torch.nn.functional.gelu(grid[r])
        worst = max(worst, float((got - want).abs().max()))
    print(f"fused GELU (fitted tail) vs fp64 GELU on [-9, 9]: max |d| = {worst:.2e}")
    assert worst < 3e-6          # split-fp16 identity products + fp32 LayerNorm arithmetic around a 1.5e-7 erf


# ------------------------------------------------------------------------------------------------ fp16 path, fp16 intermediates
def test_fp16_path_with_fp16_intermediates_keeps_its_accuracy_class(dev):
    """PRAM_F16_ACT=1 (opt-in): q / k / v, the attention context and the MLP's hidden layer in fp16 in HBM — the values the
    consuming kernels would round to fp16 anyway (the hidden layer: before its LayerNorm instead of after its GELU).  Same
    accuracy class as the fp32-intermediate fp16 path: bars at 2 x the measured distances to the fp32 oracle."""
    from pram_amd.nets import _blocks as blk
    from tests.test_gpu_configs import _segnet, _tokens
    desc, kp = _tokens(2048, idx=2)
    ref = R.segnetvit_forward(H.segnet_sd(113), desc, kp, (1, 3, 480, 640))
    net = _segnet(dev, 113).set_precision("f16")
    data = {"seg_descriptors": desc.to(dev), "keypoints": kp.to(dev), "image": torch.empty(1, 3, 480, 640)}
    saved = blk.F16_ACT
    try:
        out = {}
        for act in (False, True):
            blk.F16_ACT = act
            out[act] = net(data)["prediction"].cpu()
    finally:
        blk.F16_ACT = saved
        net.set_precision(None)
    d_old, d_new = H.maxdiff(out[False], ref), H.maxdiff(out[True], ref)
    agree = float((out[True].argmax(-1) == ref.argmax(-1)).float().mean())
    print(f"fp16 path N=2048 nc113: |logit - oracle| fp32 intermediates {d_old:.3e}, fp16 intermediates {d_new:.3e}, argmax agreement {agree:.4f}")
    assert d_new < 8e-2 and d_new < 2.0 * d_old + 1e-2 and agree >= 0.99


# ------------------------------------------------------------------------------------------------ glue kernels
def test_pack_record_kernel(dev):
    B, k, km = 3, 40, 25
    kp = torch.floor(W.uniform(51, "pr/k", (B, k, 2), 0.0, 600.0)).to(dev)
    sc = W.uniform(51, "pr/s", (B, k), 0.0, 1.0).to(dev)
    lm = torch.randint(-1, 100, (B, k), dtype=torch.int32).to(dev)
    m0 = torch.randint(-1, 50, (B, km), dtype=torch.int64).to(dev)
    ms = W.uniform(51, "pr/m", (B, km), 0.0, 1.0).to(dev)
    rec = ops.pack_record(kp, sc, lm, m0, ms)
    want = torch.zeros(B, k, 6, device=dev)
    want[:, :, 0:2], want[:, :, 2], want[:, :, 3] = kp, sc, lm.float()
    want[:, :, 4] = -1
    want[:, :km, 4], want[:, :km, 5] = m0.float(), ms
    assert torch.equal(rec, want)
    rec = ops.pack_record(kp, sc)
    assert torch.equal(rec[:, :, :3], want[:, :, :3]) and float(rec[:, :, 3:].abs().max()) == 0.0
    f = ops._filled((5, 7), dev, torch.int32, -2)
    assert f.dtype == torch.int32 and bool((f == -2).all()) and float(ops._filled((3, 3), dev).abs().max()) == 0.0


def test_secondary_matcher_shape_512_by_1024(dev):
    """the 512 best keypoints of a query against 1024-keypoint reference sets (unequal, zero-padded token sets);
    GML indices against the oracle."""
    from pram_amd.nets.gml import GML
    net = GML({})
    net.load_state_dict(H.gml_sd(), strict=True)
    data, _ = H.pair_data(6, 512, 1024)
    ref = R.gml_produce_matches(H.gml_sd(), data, p=0.0)
    r = net.to(dev).eval().produce_matches(_to(data, dev), p=0.0)
    assert torch.equal(r["matches0"].cpu(), ref["matches0"]) and torch.equal(r["matches1"].cpu(), ref["matches1"])
    assert H.maxdiff(r["matching_scores0"], ref["matching_scores0"]) < 1e-3


# ------------------------------------------------------------------------------------------------ RCCL, one rank
def test_gather_records_through_rccl_with_one_rank(dev):
    """The one collective of the path pushed through RCCL itself (backend 'nccl' on ROCm) with world_size 1 — what a 1-GPU box
    can execute of the multi-GPU path: communicator set-up on the device, all_gather_into_tensor issued from the main stream
    after a lane stream produced the record (bench.py's stream pattern), even and padded (shard_sizes) forms."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from pram_amd import ops
from pram_amd.pipeline import gather_records
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
lane = torch.cuda.Stream(device=dev)
kp = torch.arange(2 * 64 * 2, device=dev, dtype=torch.float32).view(2, 64, 2)
with torch.cuda.stream(lane):
    rec = ops.pack_record(kp, kp[:, :, 0].contiguous())
main = torch.cuda.current_stream(dev)
main.wait_stream(lane)
rec.record_stream(main)
# world size 1 short-circuits in gather_records: drive the collective itself
full = torch.empty_like(rec)
dist.all_gather_into_tensor(full, rec.contiguous())
t = torch.tensor([3.0], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(full, rec) and float(t) == 3.0
assert torch.equal(gather_records(rec), rec) and torch.equal(gather_records(rec, [2]), rec)
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
''' % str(root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


# ------------------------------------------------------------------------------------------------ 3x3 conv with the window in LDS
@pytest.mark.parametrize("shape", [(4, 120, 160, 256, 256), (12, 60, 80, 128, 256), (12, 37, 45, 64, 512), (16, 120, 160, 32, 256),
                                   (2, 240, 320, 64, 128), (9, 37, 45, 96, 128)])      # the last two: the 128-channel form (conv2a)
def test_conv3x3_halo_equals_chunked(dev, shape, monkeypatch):
    """3x3 / stride-1 layers with >= 256 (or exactly 128) output channels keep the input window of a workgroup resident in LDS (staged and split once
    per 32-channel slab instead of once per tap): same K order, same products per accumulator -> the SAME bits as the per-tap kernel
    (PRAM_CONV_HALO=0), with bias, BatchNorm scale / shift, residual and ReLU, on whole and on ragged tiles; both against fp64."""
    B, Hh, Ww, cin, cout = shape
    x = W.normal(41, f"hc/x{shape}", (B, Hh, Ww, cin), 1.0).to(dev)
    w = W.normal(41, f"hc/w{shape}", (cout, 3, 3, cin), (9 * cin) ** -0.5).to(dev)
    bias = W.normal(41, "hc/b", (cout,), 0.1).to(dev)
    sc = (1.0 + W.normal(41, "hc/s", (cout,), 0.1)).to(dev)
    sh = W.normal(41, "hc/t", (cout,), 0.1).to(dev)
    res = W.normal(41, f"hc/r{shape}", (B, Hh, Ww, cout), 1.0).to(dev)
    for kw in (dict(relu=False), dict(scale=sc, shift=sh, residual=res, relu=True)):
        monkeypatch.setenv("PRAM_CONV_HALO", "1")
        a = ops.conv2d_nhwc(x, w, bias, kw.get("scale"), kw.get("shift"), residual=kw.get("residual"), ks=3, relu=kw["relu"], precision="x3")
        monkeypatch.setenv("PRAM_CONV_HALO", "0")
        b = ops.conv2d_nhwc(x, w, bias, kw.get("scale"), kw.get("shift"), residual=kw.get("residual"), ks=3, relu=kw["relu"], precision="x3")
        assert torch.equal(a, b)
        want = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
        if "scale" in kw:
            want = torch.relu(want * sc.double() + sh.double() + res.double())
        assert float((a.double() - want).abs().max()) < 2e-5
