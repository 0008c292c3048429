"""Edge formats and the batch pair-matching driver (a17)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from pram_amd.localization import formats as F


def test_pair_names_and_retrieval(tmp_path):
    assert F.names_to_pair("seq1/frame-0.png", "db/a/b.jpg") == "seq1-frame-0.png/db-a-b.jpg" == R.names_to_pair("seq1/frame-0.png", "db/a/b.jpg")
    assert F.names_to_pair_old("a/b", "c") == "a-b_c"
    p = tmp_path / "pairs.txt"
    p.write_text("q1 r1\nq1 r2\nq2 r1\n")
    assert F.parse_retrieval(p) == {"q1": ["r1", "r2"], "q2": ["r1"]}


def test_find_unique_new_pairs():
    pairs = [("a", "b"), ("b", "a"), ("a", "c"), ("a", "b"), ("c", "d")]
    assert F.find_unique_new_pairs(pairs) == [("a", "b"), ("a", "c"), ("c", "d")]
    st = F.DictStore()
    F.write_matches(st, F.names_to_pair("c", "a"), {"matches0": np.zeros(3, np.int16)})      # reverse order, new naming
    st.create_group(F.names_to_pair_old("c", "d"))                                             # old naming
    assert F.find_unique_new_pairs(pairs, st) == [("a", "b")]


def test_match_encoding_matches_reference_casts():
    m = torch.tensor([[5, -1, 32767, 40000, 0]], dtype=torch.int64)          # 40000 wraps in int16, as .short() does
    s = torch.tensor([[0.2, 0.0, 0.33333334, 0.99951172, 1e-8]], dtype=torch.float32)
    enc = F.encode_matches(m[0], s[0])
    want = R.writer_encode({"matches0": m, "matching_scores0": s})
    assert enc["matches0"].dtype == np.int16 and enc["matching_scores0"].dtype == np.float16
    assert np.array_equal(enc["matches0"], want["matches0"]) and np.array_equal(enc["matches0"], np.array([5, -1, 32767, -25536, 0], np.int16))
    assert np.array_equal(enc["matching_scores0"].view(np.uint16), want["matching_scores0"].view(np.uint16))
    st = F.DictStore()
    F.write_matches(st, "q/r", enc)
    F.write_matches(st, "q/r", enc)            # an existing group is replaced, not an error
    m2, s2 = F.read_matches(st, "q", "r")
    assert np.array_equal(m2, enc["matches0"]) and np.array_equal(s2, enc["matching_scores0"])
    assert "matching_scores0" not in F.encode_matches(m[0])


def test_feature_encoding_and_pair_item_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    n = 37
    pred = {"keypoints": rng.integers(4, 600, (n, 2)).astype(np.float64), "scores": rng.random(n),
            "descriptors": rng.standard_normal((n, 128))}
    image_hw, orig_wh = (480, 640), np.array([1296, 968])
    enc = F.encode_features(pred, image_hw, orig_wh)
    want = R.feature_encode(pred, (1, 3) + image_hw, orig_wh)
    assert enc["descriptors"].shape == (128, n)
    for k in want:
        assert np.array_equal(enc[k], want[k]), k
    # the affine map by hand: x' = (x + .5) * float32(1296 / 640) - .5
    sx = np.float32(1296 / 640)
    assert np.array_equal(enc["keypoints"][:, 0], (pred["keypoints"][:, 0] + .5) * sx - .5)
    st = F.DictStore()
    F.write_features(st, "seq/a.png", enc)
    F.write_features(st, "seq/b.png", enc)
    st.save(tmp_path / "feats.npz")
    st2 = F.DictStore.load(tmp_path / "feats.npz")
    item = F.read_feature_pair(st2, "seq/a.png", st2, "seq/b.png")
    want_item = R.pair_item(dict(st["seq/a.png"].items()), dict(st["seq/b.png"].items()))
    assert set(item) == set(want_item)
    for k in item:
        if k.startswith("image") and not k.startswith("image_size"):
            assert tuple(item[k].shape) == tuple(want_item[k].shape) == (1, 968, 1296)
        else:
            assert torch.equal(item[k], want_item[k]), k
    assert item["descriptors0"].shape == (n, 128) and item["descriptors0"].dtype == torch.float32


def test_open_store_without_h5py_says_so():
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is installed here")
    except ImportError:
        with pytest.raises(ImportError, match="DictStore"):
            F.open_store("/tmp/x.h5")


@pytest.mark.gpu
def test_match_pairs_batched_equals_reference_loop(tmp_path):
    """match_pairs (padded batches, device-side int16 / fp16 casts) == the reference's loop: one pair at a time through
    the plugin, writer_fn's casts (match_features_batch.py:214-228,119-129)."""
    from pram_amd import weights as Wt
    from pram_amd.nets.gml import GML
    dev = torch.device("cuda:0")
    net = GML({}).eval()
    net.load_state_dict(Wt.make_state_dict("gml", net.state_dict(), seed=7), strict=True)
    net = net.to(dev)
    feats = F.DictStore()
    names, sizes = [f"seq/f{i}.png" for i in range(5)], [300, 211, 256, 97, 180]
    for i, (nm, n) in enumerate(zip(names, sizes)):
        sp = Wt.synthetic_match_pair(40 + i, n, n)
        pred = {"keypoints": sp["keypoints0"].double().numpy(), "scores": sp["scores0"].double().numpy(),
                "descriptors": sp["descriptors0"].double().numpy()}
        F.write_features(feats, nm, F.encode_features(pred, (480, 640), np.array([640, 480])))
    pairs = [(names[0], names[1]), (names[2], names[3]), (names[4], names[0]), (names[1], names[2])]
    out = F.DictStore()
    assert F.match_pairs(net, pairs, feats, feats, out, batch_size=3, device=dev) == 4
    for a, b in pairs:
        item = F.read_feature_pair(feats, a, feats, b)
        data = {k: (v[None].to(dev) if not k.startswith("image") else v[None]) for k, v in item.items()}
        pred = net.produce_matches(data)
        want = R.writer_encode({k: v.cpu() for k, v in pred.items()})
        m, s = F.read_matches(out, a, b)
        assert np.array_equal(m, want["matches0"]) and np.array_equal(s.view(np.uint16), want["matching_scores0"].view(np.uint16))
    # the plugin route (match_from_paths: dynamic_load -> strict checkpoint load -> loop), skipping pairs already stored
    from pram_amd.localization.match_features_batch import match_from_stores
    ck = tmp_path / "gml.pth"
    torch.save({"model": {k: v.cpu() for k, v in net.state_dict().items()}}, ck)
    out2 = F.DictStore()
    F.write_matches(out2, F.names_to_pair(*pairs[0][::-1]), {"matches0": np.zeros(3, np.int16)})   # reverse order present -> skipped
    assert match_from_stores("gml", pairs + [pairs[1]], feats, feats, out2, weight_path=str(ck), device=dev, batch_size=2) == 3
    for a, b in pairs[1:]:
        m, s = F.read_matches(out, a, b)
        m2, s2 = F.read_matches(out2, a, b)
        assert np.array_equal(m, m2) and np.array_equal(s.view(np.uint16), s2.view(np.uint16))
    assert F.names_to_pair(*pairs[0]) not in out2


def test_formats_reproduce_the_reference_files(golden):
    """tests/golden/formats_pinned.npz holds what the reference's OWN code wrote / read (extract_features.main's export loop,
    FeaturePairsDataset.__getitem__, writer_fn) when oracle/gen_golden.py::gen_formats ran it against an in-memory h5py
    stand-in; the product functions must reproduce those arrays bit for bit, dtypes included."""
    import numpy as np
    import torch
    from pram_amd import weights as W
    from pram_amd.localization import formats as F2
    g = golden("formats_pinned")
    store = F2.DictStore()
    for i, (name, n, hw, orig) in enumerate((("db/3.jpg", 57, (480, 640), (1280, 960)), ("q/1.jpg", 40, (240, 320), (320, 240)))):
        pred = {"keypoints": np.floor(W.uniform(40 + i, "ff/k", (n, 2), 4.0, 200.0).numpy()).astype(np.float64),
                "scores": W.uniform(40 + i, "ff/s", (n,), 0.0, 1.0).numpy().astype(np.float64),
                "descriptors": W.normal(40 + i, "ff/d", (n, 128), 1.0).numpy().astype(np.float64)}
        enc = F2.encode_features(pred, hw, np.array(orig))
        for k in ("descriptors", "keypoints", "scores", "image_size"):
            want = g[f"feat_{i}_{k}"]
            assert np.asarray(enc[k]).dtype == want.dtype and np.array_equal(enc[k], want), (name, k)
        F2.write_features(store, name, enc)
    item = F2.read_feature_pair(store, "q/1.jpg", store, "db/3.jpg")
    for k, v in item.items():
        if k in ("image0", "image1"):
            assert tuple(v.shape) == tuple(g[f"item_{k}_shape"])
        else:
            assert v.dtype == torch.float32 and np.array_equal(v.numpy(), g[f"item_{k}"]), k
    enc = F2.encode_matches(torch.from_numpy(g["write_m0"])[0], torch.from_numpy(g["write_s0"])[0])
    assert enc["matches0"].dtype == np.int16 and np.array_equal(enc["matches0"], g["write_matches0"])
    assert enc["matching_scores0"].dtype == np.float16 and np.array_equal(enc["matching_scores0"], g["write_matching_scores0"])
