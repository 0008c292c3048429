"""GPU (MI355X): corner cases raised in review — zero Sinkhorn iterations, empty sets inside ragged batches, a one-layer
AdaGML, the load_sfd2 entry point on a checkpoint file, the 'NNM' registry row, graph-private scratch memory."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from pram_amd import weights as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _to(data, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}


def test_sinkhorn_zero_iterations_is_the_row_softmax(dev):
    """sink_algorithm(M, bin, 0) = softmax_row of the augmented matrix (u = v = 1, nets/gml.py:27-46); matches come from it."""
    from pram_amd.nets.gml import GML, sink_algorithm
    M = W.normal(21, "s0/M", (2, 70, 93), 2.0)
    b = torch.tensor(0.7)
    for it in (0, 1, 3):
        want = R.sink_algorithm(M, b, it)
        got = sink_algorithm(M.to(dev), b.to(dev), it)
        assert float(((got.cpu() - want).abs() / want.abs().clamp_min(1.0)).max()) < 1e-6, it      # the corner cell holds m + n mass
    net = GML({"sinkhorn_iterations": 0})
    net.load_state_dict(H.gml_sd(), strict=True)
    data, _ = H.pair_data(1, 200, 230)
    ref = R.gml_produce_matches(H.gml_sd(), data, p=0.0, sinkhorn_iterations=0)
    r = net.to(dev).eval().produce_matches(_to(data, dev), p=0.0)
    assert torch.equal(r["matches0"].cpu(), ref["matches0"]) and H.maxdiff(r["matching_scores0"], ref["matching_scores0"]) < 1e-3
    assert bool(torch.isfinite(r["matching_scores0"]).all())


@pytest.mark.parametrize("prec", ["x3", "f32"])
def test_empty_sets_in_ragged_batches_stay_finite(dev, prec):
    """A pair whose query (or reference) set is empty sits next to normal pairs in one batch: the normal pairs keep their
    B = 1 results, the empty pair reports "no matches", and nothing on the way is NaN / uninitialised."""
    from pram_amd import ops
    from pram_amd.nets.adagml import AdaGML
    from pram_amd.nets.gml import GML
    # kernel level: attention with an empty key set writes zeros, column means of an empty query set are zeros
    q = W.normal(22, "e/q", (2 * 96, 256)).to(dev)
    kl = torch.tensor([96, 0], dtype=torch.int32, device=dev)
    ql = torch.tensor([96, 40], dtype=torch.int32, device=dev)
    out, lse = ops.attention(q, q, q, 2, 4, 96, 96, 0.125, ql, kl, want_lse=True, out=torch.full((192, 256), float("nan"), device=dev), precision="f32")
    assert float(out[96:136].abs().max()) == 0.0 and bool(torch.isfinite(out[:96]).all())
    col = ops.attention_colmean(q, q, torch.zeros(2, 4, 96, device=dev), 2, 4, 96, 96, 0.125,
                                torch.tensor([96, 0], dtype=torch.int32, device=dev), torch.tensor([96, 96], dtype=torch.int32, device=dev))
    assert bool(torch.isfinite(col).all()) and float(col[1].abs().max()) == 0.0
    for cls, sd in ((GML, H.gml_sd()), (AdaGML, H.adagml_sd())):
        net = cls({})
        net.load_state_dict(sd, strict=True)
        net = net.to(dev).eval().set_precision(prec)
        ds = [H.pair_data(i, 300, 300, device=dev)[0] for i in range(3)]
        cat = {k: torch.cat([d[k] for d in ds], 0) for k in ds[0] if torch.is_tensor(ds[0][k])}
        cat["image_shape0"] = cat["image_shape1"] = (1, 3, 640, 480)
        cat["lens0"] = torch.tensor([300, 0, 300], dtype=torch.int32, device=dev)
        cat["lens1"] = torch.tensor([300, 300, 0], dtype=torch.int32, device=dev)
        r = net.produce_matches(cat, p=0.0)
        solo = net.produce_matches(ds[0], p=0.0)
        assert torch.equal(r["matches0"][0], solo["matches0"][0]) and torch.equal(r["matching_scores0"][0], solo["matching_scores0"][0])
        assert bool((r["matches0"][1:] == -1).all()) and bool(torch.isfinite(r["matching_scores0"]).all())
        assert float(r["matching_scores0"][1:].abs().max()) == 0.0


def test_adagml_single_layer(dev):
    """n_layers = 1: the loop never reaches the pruning branch and the matcher must fall through to out_proj[0] on the
    un-pruned tokens, like the reference (nets/adagml.py:374)."""
    from pram_amd.nets.adagml import AdaGML
    net = AdaGML({"n_layers": 1})
    sd = W.make_state_dict("adagml", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    data, _ = H.pair_data(2, 320, 288)
    ref = R.adagml_produce_matches(sd, data, p=0.0, n_layers=1)
    r = net.to(dev).eval().produce_matches(_to(data, dev), p=0.0)
    assert int((ref["matches0"] >= 0).sum()) > 0
    assert torch.equal(r["matches0"].cpu(), ref["matches0"]) and H.maxdiff(r["matching_scores0"], ref["matching_scores0"]) < 1e-3


def test_load_sfd2_from_checkpoint_file(dev):
    """a7: load_sfd2(path) = ResNet4x(3, 128) + torch.load(path)['state_dict'] strict (nets/sfd2.py:592-596); a checkpoint
    with a missing / renamed key must fail the strict load."""
    from pram_amd.nets.sfd2 import load_sfd2
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "sfd2.pth")
        torch.save({"state_dict": H.sfd2_sd(), "epoch": 3}, p)
        net = load_sfd2(p).to(dev).eval()
        bad = {("x." + k if i == 0 else k): v for i, (k, v) in enumerate(H.sfd2_sd().items())}
        torch.save({"state_dict": bad}, os.path.join(td, "bad.pth"))
        with pytest.raises(RuntimeError):
            load_sfd2(os.path.join(td, "bad.pth"))
        torch.save({"model": H.sfd2_sd()}, os.path.join(td, "wrongkey.pth"))
        with pytest.raises(KeyError):
            load_sfd2(os.path.join(td, "wrongkey.pth"))
    img = W.synthetic_image(1, 96, 128)[None]
    r = net.extract_local_global({"image": img.to(dev)}, {"max_keypoints": 64, "min_keypoints": 8})
    o = R.sfd2_extract_local_global(H.sfd2_sd(), img, max_keypoints=64, min_keypoints=8)
    assert H.maxdiff(r["score_map"], o["score_map"]) < 1e-4 and len(r["keypoints"][0]) == len(o["keypoints"][0])


def test_nnm_conf_builds_the_nearest_neighbor_plugin(dev, golden):
    """confs['NNM'] (match_features_batch.py:52-60) -> matchers/nearest_neighbor.py; 'gm' / 'superglue' say why they are absent."""
    from pram_amd.localization import formats
    from pram_amd.localization.match_features_batch import build_matcher, confs, match_from_stores
    from pram_amd.localization.matchers.nearest_neighbor import NearestNeighbor
    assert confs["NNM"]["model"] == {"name": "nearest_neighbor", "do_mutual_check": True, "distance_threshold": None}
    model = build_matcher("NNM", device=dev)
    assert isinstance(model, NearestNeighbor)
    g = golden("nn_m300_n260")
    pair = W.synthetic_match_pair(7, 300, 260)
    d0, d1 = pair["descriptors0"].t()[None].contiguous(), pair["descriptors1"].t()[None].contiguous()
    r = model({"descriptors0": d0.to(dev), "descriptors1": d1.to(dev)})
    assert np.array_equal(r["matches0"].cpu().numpy().astype(np.int32), g["m0_mutual"])
    for name in ("gm", "superglue"):
        with pytest.raises(NotImplementedError):
            build_matcher(name)
    with pytest.raises(KeyError):
        build_matcher("nope")
    # the pair-matching driver accepts the descriptor-only plugin too
    sq, sr, sm = formats.DictStore(), formats.DictStore(), formats.DictStore()
    for store, name, dd, kk in ((sq, "q.jpg", pair["descriptors0"], pair["keypoints0"]), (sr, "r.jpg", pair["descriptors1"], pair["keypoints1"])):
        formats.write_features(store, name, {"descriptors": dd.t().numpy(), "keypoints": kk.numpy(), "scores": np.ones(len(kk), np.float32),
                                             "image_size": np.array([640, 480])})
    assert match_from_stores("NNM", [("q.jpg", "r.jpg")], sq, sr, sm, device=dev) == 1
    m, s = formats.read_matches(sm, "q.jpg", "r.jpg")
    assert m.dtype == np.int16 and np.array_equal(m.astype(np.int32), g["m0_mutual"][0])


def test_graphed_pipelines_own_their_scratch(dev):
    """Two captured pipelines replayed on different streams, with an eager run of a LARGER problem in between: every replay
    still equals the eager result (scratch buffers are private to a graph and never regrown under it)."""
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    from pram_amd.nets.sfd2 import ResNet4x
    from pram_amd.pipeline import GraphedPipeline, QueryPipeline
    sfd2, seg, gml = ResNet4x(), load_segnet('segnetvit', 113, 256, 15, 1024), GML({})
    for m, sd in ((sfd2, H.sfd2_sd()), (seg, H.segnet_sd(113)), (gml, H.gml_sd())):
        m.load_state_dict(sd, strict=True)
        m.to(dev).eval()
    pipe = QueryPipeline(sfd2, seg, gml, max_keypoints=192, min_keypoints=8)
    mk = lambda j, n=2, h=96, w=128: torch.stack([W.synthetic_image(20 * j + i, h, w) for i in range(n)]).to(dev)

    def refset(img, k=None):
        ex = sfd2.extract_batched(img, pipe.cfg)
        return {"descriptors": ex["descriptors"].flip(1).contiguous(), "keypoints": ex["keypoints"].flip(1).contiguous(),
                "scores": ex["scores"].flip(1).contiguous()}
    a, b = mk(0), mk(1)
    ra, rb = refset(a), refset(b)
    want_a = QueryPipeline.pack_record(pipe.run(a, ra)).clone()
    want_b = QueryPipeline.pack_record(pipe.run(b, rb)).clone()
    ga, gb = GraphedPipeline(pipe, a, ra), GraphedPipeline(pipe, b, rb)
    big = mk(3, n=6, h=160, w=224)
    pipe.run(big, refset(big))                       # a larger eager problem regrows the eager workspaces
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_a = QueryPipeline.pack_record(ga.run(a, ra))
        with torch.cuda.stream(s2):
            got_b = QueryPipeline.pack_record(gb.run(b, rb))
        torch.cuda.synchronize()
        assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
    with pytest.raises(ValueError):
        GraphedPipeline(pipe, a, None, stages="er").run(a, ra)


@pytest.mark.parametrize("prec", ["x3", "f32"])
def test_ragged_linear_touches_only_valid_rows(dev, prec):
    """lens / t_pad: valid rows equal the plain call bit for bit, rows beyond their sequence's length keep what `out` held
    (AdaGML commits out_proj of the pairs that stop at a layer into a persistent buffer), whole tiles of dead rows are skipped."""
    from pram_amd import ops
    T, S, K, N = 300, 5, 256, 256
    x = W.normal(41, "rag/x", (S * T, K)).to(dev)
    w = W.normal(41, "rag/w", (N, K), 1.0 / 16).to(dev)
    b = W.normal(41, "rag/b", (N,), 0.1).to(dev)
    lens = torch.tensor([300, 0, 17, 129, 0], dtype=torch.int32, device=dev)
    full = ops.linear(x, w, b, precision=prec)
    out = torch.full((S * T, N), -7.0, device=dev)
    ops.linear(x, w, b, precision=prec, out=out, lens=lens, t_pad=T)
    o3, f3 = out.view(S, T, N), full.view(S, T, N)
    for s_, n in enumerate(lens.tolist()):
        assert torch.equal(o3[s_, :n], f3[s_, :n])
        assert bool((o3[s_, n:] == -7.0).all())
    if prec == "x3":
        # the split-plane output obeys the same rule (AdaGML keeps its matching descriptors in persistent planes)
        _, ref_pl = ops.linear(x, w, b, precision="x3", split_out="only")
        keep = torch.full((2, S * T, N), 5.0, device=dev, dtype=torch.float16)
        ops.linear(x, w, b, precision="x3", split_out="only", out_planes=(keep[0], keep[1]), lens=lens, t_pad=T)
        for pl_i in (0, 1):
            k3, r3 = keep[pl_i].view(S, T, N), ref_pl[pl_i].view(S, T, N)
            for s_, n in enumerate(lens.tolist()):
                assert torch.equal(k3[s_, :n], r3[s_, :n]) and bool((k3[s_, n:] == 5.0).all())
    h = ops.linear(x, w, b, precision=prec).clone()
    g, bt = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    ref = ops.layernorm_gelu_(h.clone(), g, bt)
    got = ops.layernorm_gelu_(h.clone(), g, bt, lens=lens, t_pad=T).view(S, T, N)
    for s_, n in enumerate(lens.tolist()):
        assert torch.equal(got[s_, :n], ref.view(S, T, N)[s_, :n]) and torch.equal(got[s_, n:], h.view(S, T, N)[s_, n:])


def test_packed_weights_follow_in_place_parameter_edits(dev):
    """The packed / split device copies of the weights are keyed on the parameters' version counters: an in-place edit
    (optimizer step, p.mul_() under no_grad) must show up in the next forward; refresh_packed() covers writes through .data."""
    from pram_amd.nets.gml import GML
    net = GML({})
    sd = W.make_state_dict("gml", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    data, _ = H.pair_data(1, 320, 288)
    before = net.produce_matches(_to(data, dev))["matching_scores0"].clone()
    heads = [p for n, p in net.named_parameters() if n.startswith("out_proj.") and n.endswith(".weight")]
    assert heads
    with torch.no_grad():
        for p in heads:
            p.mul_(1.05)
    sd2 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ref = R.gml_produce_matches(sd2, data)
    r = net.produce_matches(_to(data, dev))
    assert H.maxdiff(r["matching_scores0"], before) > 1e-4
    assert torch.equal(r["matches0"].cpu(), ref["matches0"]) and H.maxdiff(r["matching_scores0"], ref["matching_scores0"]) < 1e-3
    for p in heads:
        p.data.div_(1.05)                    # invisible to the version counter
    net.refresh_packed()
    again = net.produce_matches(_to(data, dev))["matching_scores0"]
    assert H.maxdiff(again, before) < 1e-4


def test_profiling_entry_points_are_inert_without_their_switch(dev):
    """pram_debug_gemm_phases reads (and clears) the wide GEMM's phase counters; without PRAM_GEMM_ABLATE=4 in the environment the
    product kernels never touch them, so they stay zero across a GEMM."""
    import ctypes
    import os
    from pram_amd import _lib, ops
    assert os.environ.get("PRAM_GEMM_ABLATE") is None
    L = _lib.load()
    buf = (ctypes.c_ulonglong * 72)()
    _lib.check(L.pram_debug_gemm_phases(buf, 1), "pram_debug_gemm_phases")
    x = W.normal(51, "dbg/x", (4096, 256)).to(dev)
    w = W.normal(51, "dbg/w", (512, 256), 1.0 / 16).to(dev)
    ops.linear(x, w, None, precision="x3")
    torch.cuda.synchronize()
    _lib.check(L.pram_debug_gemm_phases(buf, 0), "pram_debug_gemm_phases")
    assert all(int(v) == 0 for v in buf)
