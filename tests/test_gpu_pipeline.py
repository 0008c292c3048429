"""GPU: the batched QueryPipeline (extract -> sample -> recognise -> match, ragged counts on the device) against
the oracle run per query, and the record packing that the multi-GPU gather ships."""
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


def _models(dev):
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    from pram_amd.nets.sfd2 import ResNet4x
    sfd2, seg, gml = ResNet4x(), load_segnet('segnetvit', 113, 256, 15, 1024), GML({})
    for m, sd in ((sfd2, H.sfd2_sd()), (seg, H.segnet_sd(113)), (gml, H.gml_sd())):
        m.load_state_dict(sd, strict=True)
        m.to(dev).eval()
    return sfd2, seg, gml


def test_pipeline_matches_per_query_oracle(dev):
    from pram_amd.pipeline import QueryPipeline
    sfd2, seg, gml = _models(dev)
    k = 160
    # frame 1 has fewer candidates than k after the border cut (ragged batch), frame 0 more
    img = torch.stack([W.synthetic_image(1, 96, 128), W.synthetic_image(4, 96, 128) * 0.3])
    pipe = QueryPipeline(sfd2, seg, gml, max_keypoints=k, min_keypoints=8)
    # reference sets: noisy permuted copies of each query's own descriptors (built from a first extraction)
    ex = sfd2.extract_batched(img.to(dev), pipe.cfg)
    counts = ex["counts"].tolist()
    assert counts[0] == k and 0 < counts[1] <= k
    ref = {"descriptors": ex["descriptors"].flip(1).contiguous(), "keypoints": ex["keypoints"].flip(1).contiguous(),
           "scores": ex["scores"].flip(1).contiguous()}
    out = pipe.run(img.to(dev), ref, stages="erm")
    assert out["counts"].tolist() == counts
    rec = QueryPipeline.pack_record(out)
    assert tuple(rec.shape) == (2, k, 6)
    for b in range(2):
        n = counts[b]
        o = R.sfd2_extract_local_global(H.sfd2_sd(), img[b:b + 1], max_keypoints=k, min_keypoints=8, per_image_fallback=True)
        assert n == len(o["keypoints"][0])
        same = (out["keypoints"][b, :n].cpu() == o["keypoints"][0]).all(1).float().mean().item()
        assert same > 0.95, same
        # stage-isolated: feed the GPU's own keypoints / descriptors to the oracle's recogniser and matcher
        kp = out["keypoints"][b:b + 1, :n].cpu()
        _, segd = R.sfd2_sample(o["score_map"], o["mid_features"], kp[0], norm_desc=False)
        logits = R.segnetvit_forward(H.segnet_sd(113), segd.t()[None], kp, (1, 3, 96, 128))
        assert H.maxdiff(out["prediction"][b, :n], logits[0]) < 1e-3
        post = R.add_segmentations(logits[0], 0.95)
        ids = (logits[0].argmax(-1) - 1)
        assert torch.equal(out["landmark"][b, :n].cpu().long(), ids)
        assert torch.equal(rec[b, :n, 3].cpu().long(), ids)
        # matcher: query set = this frame's n keypoints, reference set = the flipped padded set (all k rows)
        data = {"descriptors0": out["descriptors"][b:b + 1, :n].cpu(), "keypoints0": kp, "scores0": out["scores"][b:b + 1, :n].cpu(),
                "descriptors1": ref["descriptors"][b:b + 1].cpu(), "keypoints1": ref["keypoints"][b:b + 1].cpu(),
                "scores1": ref["scores"][b:b + 1].cpu(), "image_shape0": (1, 3, 128, 96), "image_shape1": (1, 3, 128, 96)}
        m = R.gml_produce_matches(H.gml_sd(), data, p=0.2)
        assert torch.equal(out["matches0"][b, :n].cpu(), m["matches0"][0])
        assert H.maxdiff(out["matching_scores0"][b, :n], m["matching_scores0"][0]) < 1e-3
    assert bool((rec[1, counts[1]:, 4] == -1).all())        # padded rows of the short query carry "no match"


def test_batches_in_flight_on_separate_streams_equal_serial(dev):
    """bench.py --inflight: several batches run concurrently on their own HIP streams (workspaces are per stream);
    every batch must produce exactly what it produces alone."""
    from pram_amd.pipeline import QueryPipeline
    sfd2, seg, gml = _models(dev)
    k = 256
    pipe = QueryPipeline(sfd2, seg, gml, max_keypoints=k, min_keypoints=8)
    batches = [torch.stack([W.synthetic_image(10 * j + i, 128, 160) for i in range(3)]).to(dev) for j in range(4)]
    refs = []
    for img in batches:
        ex = sfd2.extract_batched(img, pipe.cfg)
        refs.append({"descriptors": ex["descriptors"].flip(1).contiguous(), "keypoints": ex["keypoints"].flip(1).contiguous(),
                     "scores": ex["scores"].flip(1).contiguous(), "lens": ex["counts"].clone()})
    serial = [QueryPipeline.pack_record(pipe.run(img, ref)).clone() for img, ref in zip(batches, refs)]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for rep in range(3):                      # a few rounds: a race shows up as a rare mismatch
        got = []
        for j, (img, ref) in enumerate(zip(batches, refs)):
            with torch.cuda.stream(lanes[j % len(lanes)]):
                got.append(QueryPipeline.pack_record(pipe.run(img, ref)))
        torch.cuda.synchronize()
        for a, b in zip(got, serial):
            assert torch.equal(a, b)


@pytest.mark.parametrize("split", [True, False])
def test_graphed_pipeline_equals_eager(dev, split):
    """the whole step captured (no host synchronisation inside) and replayed on new inputs == eager run: as four graphs on two
    streams (extract | match || recognise | record: what a forking step gets by default) and as one hipGraph"""
    from pram_amd.pipeline import GraphedPipeline, QueryPipeline
    sfd2, seg, gml = _models(dev)
    pipe = QueryPipeline(sfd2, seg, gml, max_keypoints=192, min_keypoints=8)
    mk = lambda j: torch.stack([W.synthetic_image(20 * j + i, 96, 128) for i in range(2)]).to(dev)
    imgs = [mk(0), mk(1), mk(2)]
    ex = sfd2.extract_batched(imgs[0], pipe.cfg)
    ref = {"descriptors": ex["descriptors"].flip(1).contiguous(), "keypoints": ex["keypoints"].flip(1).contiguous(),
           "scores": ex["scores"].flip(1).contiguous()}
    g = GraphedPipeline(pipe, imgs[0], ref, split=split, record=True)
    assert g.split == split
    for img in imgs[1:] + imgs[:1]:
        want = QueryPipeline.pack_record(pipe.run(img, ref)).clone()
        got = QueryPipeline.pack_record(g.run(img, ref))
        assert torch.equal(got, want)
        assert torch.equal(g.record, want)          # the captured record
    with pytest.raises(ValueError):
        g.run(imgs[0][:1], ref)
