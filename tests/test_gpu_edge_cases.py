"""GPU: edge cases of the C ABI — empty and minimal inputs, ragged tails, maximum supported sizes, argument errors."""
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
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_empty_and_single_row_linear(dev):
    from pram_amd import ops
    w = W.normal(1, "e/w", (256, 256), 0.06).to(dev)
    assert tuple(ops.linear(torch.zeros(0, 256, device=dev), w).shape) == (0, 256)
    x = W.normal(1, "e/x", (1, 256))
    assert H.maxdiff(ops.linear(x.to(dev), w), F.linear(x, w.cpu())) < 1e-5


def test_attention_single_key_and_single_query(dev):
    from pram_amd import ops
    q, k, v = W.normal(2, "e/q", (1, 256)), W.normal(2, "e/k", (1, 256)), W.normal(2, "e/v", (1, 256))
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), 1, 4, 1, 1, 0.125)
    assert H.maxdiff(out, v) < 1e-6            # softmax over one key = 1
    # one query against 129 keys (tail tile of 1 key)
    k2, v2 = W.normal(3, "e/k2", (129, 256)), W.normal(3, "e/v2", (129, 256))
    out = ops.attention(q.to(dev), k2.to(dev), v2.to(dev), 1, 4, 1, 129, 0.125).cpu().view(4, 64)
    qh, kh, vh = q.view(4, 64), k2.view(129, 4, 64), v2.view(129, 4, 64)
    a = torch.softmax(torch.einsum("hd,nhd->hn", qh.double(), kh.double()) * 0.125, -1)
    assert H.maxdiff(out, torch.einsum("hn,nhd->hd", a, vh.double())) < 1e-5


def test_attention_zero_length_batch_element_is_skipped(dev):
    from pram_amd import ops
    x = W.normal(4, "e/z", (2 * 130, 256)).to(dev)
    lens = torch.tensor([130, 0], dtype=torch.int32, device=dev)
    out = torch.full((2 * 130, 256), 7.0, device=dev)
    ops.attention(x, x, x, 2, 4, 130, 130, 0.125, lens, lens, out=out)
    assert torch.isfinite(out).all() and bool((out[130:] == 7.0).all())      # untouched rows of the empty element


def test_sinkhorn_minimal_and_maximal(dev):
    from pram_amd import ops
    bs = torch.tensor(1.0, device=dev)
    M = torch.tensor([[[3.0]]])
    r = ops.sinkhorn_match(M.to(dev), bs, 20, 0.0, want_p=True)
    assert H.maxdiff(r["p"], R.sink_algorithm(M, torch.tensor(1.0), 20)) < 1e-6 and r["matches0"].tolist() == [[0]]
    # 4096 x 4096 (BASELINE C4/C5 size, the 17-float4-per-lane instantiation): indices vs the oracle
    m = n = 4096
    Mb = W.normal(23, "sink/max", (1, m, n), 2.0)
    idx = torch.argsort(W.uniform(24, "sink/maxperm", (m,)))
    Mb[0, idx, torch.arange(n)] += 9.0
    i0, i1, s0, _ = R.compute_matches(R.sink_algorithm(Mb, torch.tensor(1.0), 20), 0.2)
    rb = ops.sinkhorn_match(Mb.to(dev).contiguous(), bs, 20, 0.2)
    assert torch.equal(rb["matches0"].cpu(), i0) and torch.equal(rb["matches1"].cpu(), i1)
    assert H.maxdiff(rb["matching_scores0"], s0) < 1e-5 and (i0 >= 0).sum() > 1000
    with pytest.raises(Exception):
        ops.sinkhorn_match(torch.zeros(1, 8, 4400, device=dev), bs, 1, 0.2)       # more than 4351 columns


def test_sinkhorn_ragged_equals_unpadded(dev):
    from pram_amd import ops
    bs = torch.tensor(1.0, device=dev)
    M = W.normal(25, "sink/rag", (2, 300, 260), 2.0)
    ml = torch.tensor([300, 211], dtype=torch.int32, device=dev)
    nl = torch.tensor([260, 97], dtype=torch.int32, device=dev)
    r = ops.sinkhorn_match(M.to(dev).contiguous(), bs, 20, 0.0, m_lens=ml, n_lens=nl)
    solo = ops.sinkhorn_match(M[1:, :211, :97].contiguous().to(dev), bs, 20, 0.0)
    assert torch.equal(r["matches0"][1, :211], solo["matches0"][0]) and torch.equal(r["matches1"][1, :97], solo["matches1"][0])
    assert bool((r["matches0"][1, 211:] == -1).all()) and bool((r["matching_scores0"][1, 211:] == 0).all())
    assert torch.allclose(r["matching_scores0"][1, :211], solo["matching_scores0"][0], rtol=1e-5, atol=1e-7)


def test_select_keypoints_max_k_and_border_only(dev):
    from pram_amd import ops
    s = torch.floor(W.uniform(26, "sel/max", (1, 200, 240), 0.0, 64.0)) / 64.0 * 0.9 + 0.01      # every pixel a candidate
    kps, scs = R.select_keypoints(s, 0.005, 10, 4, 8192)
    kp, sc, cnt = ops.select_keypoints(s.to(dev).contiguous(), 0.005, 10, 4, 8192, fallback_ref=0)
    assert cnt.tolist() == [8192] and torch.equal(kp[0].cpu(), kps[0]) and torch.equal(sc[0].cpu(), scs[0])
    # one more than the in-LDS sort holds: sorted in the workspace since round 4 (the reference has no bound), same canonical order
    kps1, scs1 = R.select_keypoints(s, 0.005, 10, 4, 8193)
    kp1, sc1, cnt1 = ops.select_keypoints(s.to(dev).contiguous(), 0.005, 10, 4, 8193, fallback_ref=0)
    assert cnt1.tolist() == [8193] and torch.equal(kp1[0].cpu(), kps1[0]) and torch.equal(sc1[0].cpu(), scs1[0])
    with pytest.raises(Exception):
        ops.select_keypoints(s.to(dev).contiguous(), 0.005, 10, 4, 0)
    # all candidates inside the removed border -> zero keypoints
    z = torch.zeros(1, 64, 64)
    z[0, :4] = 0.5
    z[0, :, -4:] = 0.5
    assert ops.select_keypoints(z.to(dev), 0.005, 0, 4, 100, fallback_ref=0)[2].tolist() == [0]


def test_keep_all_keypoints(dev):
    """max_keypoints < 0 keeps every candidate in nonzero() order (nets/sfd2.py:324 skips top_k_keypoints): selection
    bit-exact against the oracle on the same NMS map (more than the 8192 an in-LDS top-k could hold), and through the
    model API."""
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    s = torch.floor(W.uniform(28, "sel/all", (2, 120, 160), 0.0, 64.0)) / 64.0 * 0.9 + 0.01
    s[1, :, 80:] = 0.0
    kps, scs = R.select_keypoints(s, 0.005, 10, 4, -1, per_image_fallback=True)
    assert len(kps[0]) == 112 * 152 > 8192
    kp, sc, cnt = ops.select_keypoints(s.to(dev).contiguous(), 0.005, 10, 4, 120 * 160, fallback_ref=-1)
    assert cnt.tolist() == [len(k) for k in kps]
    for b in range(2):
        assert torch.equal(kp[b, :cnt[b]].cpu(), kps[b]) and torch.equal(sc[b, :cnt[b]].cpu(), scs[b])
    net = ResNet4x()
    net.load_state_dict(H.sfd2_sd(), strict=True)
    net = net.to(dev).eval()
    img = W.synthetic_image(6, 96, 128)[None]
    o = R.sfd2_extract_local_global(H.sfd2_sd(), img, max_keypoints=-1, min_keypoints=8)
    r = net.extract_local_global({"image": img.to(dev)}, {"max_keypoints": -1, "min_keypoints": 8})
    n = len(o["keypoints"][0])
    assert abs(len(r["keypoints"][0]) - n) <= max(2, n // 50)
    same = (r["keypoints"][0].cpu()[:, None, :] == o["keypoints"][0][None]).all(-1).any(1).float().mean().item()
    assert same > 0.95 and tuple(r["descriptors"][0].shape) == (128, len(r["keypoints"][0]))


def test_nms_radius_zero_and_non_multiple_tile(dev):
    from pram_amd import ops
    s = W.uniform(27, "nms/odd", (2, 45, 77), 0.0, 1.0)
    for r in (0, 1, 2):
        assert torch.equal(ops.simple_nms(s.to(dev), r).cpu(), R.simple_nms(s, r))
    with pytest.raises(Exception):
        ops.simple_nms(s.to(dev), 5)


def test_conv_non_multiple_sizes_and_errors(dev):
    from pram_amd import ops
    from pram_amd.nets.sfd2 import ResNet4x
    x = W.normal(28, "cv/odd", (1, 64, 37, 53))
    wt = W.normal(28, "cv/oddw", (96, 64, 3, 3), 0.04)
    for stride in (1, 2):
        ref = F.conv2d(x, wt, stride=stride, padding=1)
        out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), ResNet4x._ohwi(wt).to(dev), ks=3, stride=stride)
        assert tuple(out.shape) == (1, ref.shape[2], ref.shape[3], 96) and H.maxdiff(out.permute(0, 3, 1, 2), ref) < 2e-5
    with pytest.raises(Exception):      # cin must be 4 or a multiple of 32
        ops.conv2d_nhwc(torch.zeros(1, 8, 8, 24, device=dev), torch.zeros(32, 3, 3, 24, device=dev), ks=3)


def test_models_on_tiny_inputs(dev):
    """N = 1 token / M = 1 vs N = 3 matcher: shapes and values against the oracle."""
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    seg = load_segnet('segnetvit', 113, 256, 15, 1024)
    seg.load_state_dict(H.segnet_sd(113), strict=True)
    seg = seg.to(dev).eval()
    desc, kp, _ = W.synthetic_tokens(0, 1)
    out = seg({"seg_descriptors": desc[None].to(dev), "keypoints": kp[None].to(dev), "image": torch.empty(1, 3, 480, 640)})["prediction"]
    assert H.maxdiff(out, R.segnetvit_forward(H.segnet_sd(113), desc[None], kp[None], (1, 3, 480, 640))) < 1e-3
    g = GML({})
    g.load_state_dict(H.gml_sd(), strict=True)
    g = g.to(dev).eval()
    data, _ = H.pair_data(0, 1, 3)
    r = g.produce_matches({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}, p=0.0)
    o = R.gml_produce_matches(H.gml_sd(), data, p=0.0)
    assert torch.equal(r["matches0"].cpu(), o["matches0"]) and torch.equal(r["matches1"].cpu(), o["matches1"])


def test_frame_staging_equals_the_reference_preparation(dev):
    """pram_stage_frames_u8 = `torch.from_numpy(img / 255).permute(2, 0, 1).cuda().float()` + tvf.Normalize(mean, std)
    (localization/loc_by_rec_online.py:98-106, nets/sfd2.py:14-17), bit for bit: every byte value in every channel, a 480 x 640
    batch, and a frame whose pixel count is only a multiple of 4."""
    from pram_amd import ops
    from pram_amd.nets.sfd2 import RGB_mean, RGB_std
    lut = ops.frame_lut(RGB_mean, RGB_std, dev)
    g = torch.Generator().manual_seed(5)
    for B, Hh, Ww in ((2, 480, 640), (1, 6, 10), (3, 16, 16)):
        img = torch.randint(0, 256, (B, Hh, Ww, 3), generator=g, dtype=torch.uint8)
        if (B, Hh, Ww) == (3, 16, 16):
            img.view(-1)[:768] = torch.arange(256, dtype=torch.uint8).repeat(3)      # every byte value, in every channel phase
        want = []
        for b in range(B):
            x = torch.from_numpy(img[b].numpy() / 255).permute(2, 0, 1).float()      # numpy float64 division, then .float()
            m = torch.tensor(RGB_mean).view(3, 1, 1)
            s = torch.tensor(RGB_std).view(3, 1, 1)
            want.append(x.sub_(m).div_(s))                                          # tvf.Normalize
        got = ops.stage_frames(img.to(dev), lut)
        assert torch.equal(got.cpu(), torch.stack(want)), (B, Hh, Ww)
