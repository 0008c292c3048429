"""ORACLE TOOLING — runs ONLY in the build container (needs /root/reference; never on the GPU box).

Imports the reference PRAM code (read-only, with the two harness-side shims of SURVEY.md H5/H6),
loads the build's deterministic weights into the reference modules with ``strict=True`` (which
proves state-dict schema compatibility), runs the reference on CPU fp32, checks the CPU
restatement ``oracle/ref_cpu.py`` against it, and writes small golden fixtures to
``tests/golden/*.npz``.  Fixtures hold data only: seeds/sizes, outputs (sub-sampled where large)
and checksums; inputs and weights are regenerated from ``pram_amd.weights``.

    python oracle/gen_golden.py            # regenerate everything
"""
from __future__ import annotations

import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))

from oracle import ref_cpu as R  # noqa: E402
from pram_amd import weights as W  # noqa: E402


def import_reference():
    """Shim H6: stub torchvision.transforms.{Compose,Normalize}; shim H5: CPU sink_algorithm for adagml."""
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")

        class Normalize:
            def __init__(self, mean, std):
                self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

            def __call__(self, x):
                return (x - self.mean) / self.std

        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, x):
                for t in self.ts:
                    x = t(x)
                return x

        tvt.Normalize, tvt.Compose = Normalize, Compose
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    sys.path.insert(0, str(REF))
    import nets.sfd2 as ref_sfd2
    import nets.segnetvit as ref_segvit
    import nets.load_segnet as ref_load
    import nets.gml as ref_gml
    import nets.adagml as ref_adagml
    import nets.utils as ref_utils
    ref_adagml.sink_algorithm = ref_gml.sink_algorithm
    return dict(sfd2=ref_sfd2, segvit=ref_segvit, load=ref_load, gml=ref_gml, adagml=ref_adagml, utils=ref_utils)


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def sub(t: torch.Tensor, n: int = 4096):
    """Deterministic sub-sample (every k-th flat element) + checksum of the full tensor."""
    f = t.reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([f.sum().item(), f.abs().sum().item()])


def save(name, **arrs):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote tests/golden/{name}.npz ({(OUT / f'{name}.npz').stat().st_size / 1024:.1f} KiB)")


def gen_segnetvit(ref):
    print("SegNetViT")
    # c513 = BASELINE C5's class count (configs/config_train_aachen_sfd2.yaml); N kept small: the fixture pins the head width
    for tag, (B, N, C) in {"b2_n512_c113": (2, 512, 113), "b1_n300_c161": (1, 300, 161), "b1_n640_c513": (1, 640, 513)}.items():
        model = ref["load"].load_segnet("segnetvit", C, 256, 15, 1024).eval()
        sd = W.make_state_dict("segnetvit", model.state_dict(), seed=7)
        model.load_state_dict(sd, strict=True)
        toks = [W.synthetic_tokens(i, N) for i in range(B)]
        desc = torch.stack([t[0] for t in toks])
        kp = torch.stack([t[1] for t in toks])
        img = torch.empty(B, 3, 480, 640)
        with torch.no_grad():
            out_ref = model({"seg_descriptors": desc, "keypoints": kp, "image": img})["prediction"]
            probes = {}
            out_or = R.segnetvit_forward(sd, desc, kp, img.shape, probes=probes)
        d = maxdiff(out_ref, out_or)
        print(f"  {tag}: |ref - oracle|max = {d:.3e}  logits range [{out_ref.min():.3f},{out_ref.max():.3f}]"
              f"  argmax classes used {out_ref.argmax(-1).unique().numel()}")
        assert d < 2e-5, d
        s, cs = sub(out_ref, 8192)
        l0, _ = sub(probes["layer0"], 2048)
        save(f"segnetvit_{tag}", B=B, N=N, C=C, seed=7, logits_sub=s, logits_checksum=cs,
             argmax=out_ref.argmax(-1).numpy().astype(np.int16), layer0_sub=l0,
             logits_rows=out_ref[:, :8].numpy())


    # with_sc head (no shipped config enables it): 3 layers, one small frame
    seg = ref["segvit"].SegNetViT({"n_class": 21, "n_layers": 3, "with_sc": True, "hidden_dim": 256, "output_dim": 1024}).eval()
    sd = W.make_state_dict("segnetvit", seg.state_dict(), seed=7)
    seg.load_state_dict(sd, strict=True)
    d0, k0 = W.synthetic_tokens(9, 200)[:2]
    img = torch.empty(1, 3, 480, 640)
    with torch.no_grad():
        r = seg({"seg_descriptors": d0[None], "keypoints": k0[None], "image": img})
        probes = {}
        o = R.segnetvit_forward(sd, d0[None], k0[None], img.shape, n_layers=3, probes=probes)
    assert maxdiff(r["prediction"], o) < 2e-5 and maxdiff(r["sc"], probes["sc"]) < 2e-5
    print(f"  with_sc: |ref - oracle|max prediction {maxdiff(r['prediction'], o):.2e}, sc {maxdiff(r['sc'], probes['sc']):.2e}")
    save("segnetvit_with_sc", sc=r["sc"].numpy(), prediction_rows=r["prediction"][:, :8].numpy(), n_class=21, n_layers=3, N=200)


def gen_sinkhorn(ref):
    print("Sinkhorn / dual-softmax / compute_matches")
    g = ref["gml"]
    for tag, (m, n) in {"257x193": (257, 193), "100x300": (100, 300)}.items():
        M = W.normal(11, f"sink/{tag}", (2, m, n), 2.0)
        # plant a permutation signal so argmax / mutual checks are non-trivial
        for b in range(2):
            idx = torch.argsort(W.uniform(12 + b, f"sink/perm/{tag}", (m,)))[: min(m, n)]
            M[b, idx, torch.arange(min(m, n))] += 6.0
        bin_score = torch.tensor(1.0)
        with torch.no_grad():
            p_ref = g.sink_algorithm(M, bin_score, 20)
            p_or = R.sink_algorithm(M, bin_score, 20)
            ds_ref = g.dual_softmax(M, bin_score)
            ds_or = R.dual_softmax(M, bin_score)
            net = g.GML({})
            res = {}
            for p in (0.0, 0.2):
                i0, i1, s0, s1 = net.compute_matches(p_ref, p)
                o0, o1, t0, t1 = R.compute_matches(p_or, p)
                assert torch.equal(i0, o0) and torch.equal(i1, o1), "compute_matches indices differ"
                assert maxdiff(s0, t0) < 1e-6 and maxdiff(s1, t1) < 1e-6
                res[p] = (i0, i1, s0, s1)
        print(f"  {tag}: sinkhorn diff {maxdiff(p_ref, p_or):.2e}  dual diff {maxdiff(ds_ref, ds_or):.2e} "
              f" matches(p=0.2) {(res[0.2][0] >= 0).sum().item()}")
        assert maxdiff(p_ref, p_or) < 1e-6 and maxdiff(ds_ref, ds_or) < 1e-6
        save(f"sinkhorn_{tag}", m=m, n=n, p=p_ref.numpy(), dual=ds_ref.numpy(),
             m0_p0=res[0.0][0].numpy().astype(np.int32), m1_p0=res[0.0][1].numpy().astype(np.int32),
             m0_p02=res[0.2][0].numpy().astype(np.int32), m1_p02=res[0.2][1].numpy().astype(np.int32),
             s0=res[0.0][2].numpy(), s1=res[0.0][3].numpy())


def _pair_data(idx, m, n, shape_key="image_shape"):
    d = W.synthetic_match_pair(idx, m, n)
    data = {k + "": v[None] for k, v in d.items() if k != "gt"}
    if shape_key == "image_shape":
        data["image_shape0"] = (1, 3, 640, 480)   # (W,H)-swapped tuple exactly as singlemap3d.py:147 passes it
        data["image_shape1"] = (1, 3, 640, 480)
    else:
        data["image0"] = torch.empty(1, 1, 480, 640)
        data["image1"] = torch.empty(1, 1, 480, 640)
    return data, d["gt"]


def gen_gml(ref):
    print("GML")
    g = ref["gml"]
    net = g.GML({}).eval()
    sd = W.make_state_dict("gml", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    # m512_n1024 = SURVEY 8(d)'s secondary shape (a query's voted keypoints against a compressed reference frame)
    for tag, (m, n, key) in {"m384_n512": (384, 512, "image_shape"), "m256_n256_img": (256, 256, "image"),
                             "m512_n1024": (512, 1024, "image_shape")}.items():
        data, gt = _pair_data(0, m, n, key)
        with torch.no_grad():
            r_def = net(data)
            r_p0 = net.produce_matches(data, p=0.0)
            probes = {}
            o_def = R.gml_produce_matches(sd, data, p=0.2, probes=probes)
            o_p0 = R.gml_produce_matches(sd, data, p=0.0)
        for a, b in ((r_def, o_def), (r_p0, o_p0)):
            for k in ("matches0", "matches1"):
                assert torch.equal(a[k], b[k]), f"{tag} {k} differs from oracle"
            for k in ("matching_scores0", "matching_scores1"):
                assert maxdiff(a[k], b[k]) < 1e-5, (tag, k, maxdiff(a[k], b[k]))
        nm = (r_def["matches0"] >= 0).sum().item()
        nm0 = (r_p0["matches0"] >= 0).sum().item()
        correct = ((r_p0["matches0"][0] == gt) & (gt >= 0)).sum().item()
        print(f"  {tag}: matches p=0.2: {nm}, p=0: {nm0} (gt-consistent {correct}); max score {r_p0['matching_scores0'].max():.3f}")
        ds, dcs = sub(probes["dist"], 4096)
        save(f"gml_{tag}", m=m, n=n, key=key, seed=7,
             m0_def=r_def["matches0"].numpy().astype(np.int32), m1_def=r_def["matches1"].numpy().astype(np.int32),
             m0_p0=r_p0["matches0"].numpy().astype(np.int32), m1_p0=r_p0["matches1"].numpy().astype(np.int32),
             s0=r_p0["matching_scores0"].numpy(), s1=r_p0["matching_scores1"].numpy(),
             dist_sub=ds, dist_checksum=dcs, desc0_sub=sub(probes["desc0"], 2048)[0])
    # wrapper path: dynamic_load -> strict state-dict load -> forward
    import localization.matchers as ref_matchers
    from localization.base_model import dynamic_load
    with tempfile.TemporaryDirectory() as td:
        wp = os.path.join(td, "gml.pth")
        torch.save({"model": sd}, wp)
        Model = dynamic_load(ref_matchers, "gml")
        wrapped = Model({"name": "gml", "weight_path": wp, "sinkhorn_iterations": 20}).eval()
        data, _ = _pair_data(0, 384, 512)
        with torch.no_grad():
            rw = wrapped(data)
        g0 = np.load(OUT / "gml_m384_n512.npz")
        assert np.array_equal(rw["matches0"].numpy().astype(np.int32), g0["m0_def"])
        print("  wrapper path (dynamic_load + strict load) reproduces the direct call")


def gen_adagml(ref):
    print("AdaGML")
    a = ref["adagml"]
    net = a.AdaGML({}).eval()
    sd = W.make_state_dict("adagml", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    for tag, (m, n, pidx) in {"m640_n768": (640, 768, 1), "m300_n280": (300, 280, 2), "m512_n1024": (512, 1024, 3)}.items():
        data, gt = _pair_data(pidx, m, n)
        with torch.no_grad():
            r = net.produce_matches(data, p=0.0)
            probes = {}
            o = R.adagml_produce_matches(sd, data, p=0.0, probes=probes)
        assert torch.equal(r["matches0"], o["matches0"]), "adagml matches0 differs from oracle"
        assert maxdiff(r["matching_scores0"], o["matching_scores0"]) < 1e-5
        print(f"  {tag}: stop layer {probes['stop_layer']}, sizes {probes['sizes']}, "
              f"matches {(r['matches0'] >= 0).sum().item()}")
        save(f"adagml_{tag}", m=m, n=n, seed=7, pair_index=pidx, m0_p0=r["matches0"].numpy().astype(np.int32),
             s0=r["matching_scores0"].numpy(), stop_layer=probes["stop_layer"],
             ind0=probes["ind0"].numpy().astype(np.int32), ind1=probes["ind1"].numpy().astype(np.int32),
             sizes=np.array(probes["sizes"], dtype=np.int32), conf0_l0=probes["conf0_0"].numpy(),
             conf0_l1=probes["conf0_1"].numpy())


def gen_adagml_run(ref):
    """AdaGML.run (mode=1): normalised keypoints packed in x, returns matched original ids."""
    print("AdaGML.run")
    a = ref["adagml"]
    net = a.AdaGML({}).eval()
    sd = W.make_state_dict("adagml", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    pair = W.synthetic_match_pair(6, 400, 360)
    nk0 = R.normalize_keypoints(pair["keypoints0"], (1, 3, 640, 480))
    nk1 = R.normalize_keypoints(pair["keypoints1"], (1, 3, 640, 480))
    data = {"desc1": pair["descriptors0"][None], "desc2": pair["descriptors1"][None],
            "x1": torch.cat([nk0, pair["scores0"][:, None]], -1)[None], "x2": torch.cat([nk1, pair["scores1"][:, None]], -1)[None]}
    with torch.no_grad():
        r = net(data, mode=1)
        o = R.adagml_run(sd, data)
    assert torch.equal(r["index0"], o["index0"]) and torch.equal(r["index1"], o["index1"])
    # tiny-set guard: biases that prune almost everything
    sd2 = dict(sd)
    for k in list(sd2):
        if k.endswith("predict.3.bias"):
            sd2[k] = torch.tensor([-20.0])
    net.load_state_dict(sd2, strict=True)
    with torch.no_grad():
        r2 = net(data, mode=1)
        o2 = R.adagml_run(sd2, data)
    assert torch.equal(r2["index0"], o2["index0"]) and r2["index0"].tolist() == [0]
    print(f"  run: {len(r['index0'])} matches; tiny-set guard returns zeros")
    save("adagml_run_m400_n360", index0=r["index0"].numpy().astype(np.int32), index1=r["index1"].numpy().astype(np.int32))


def gen_normalize(ref):
    print("normalize_keypoints (W,H)-swap quirk")
    k = torch.tensor([[[0.0, 0.0], [320.0, 240.0], [639.0, 479.0]]])
    a = ref["utils"].normalize_keypoints(k, (1, 3, 480, 640))
    b = ref["utils"].normalize_keypoints(k, (1, 3, 640, 480))
    assert maxdiff(a, R.normalize_keypoints(k, (1, 3, 480, 640))) < 1e-7
    assert maxdiff(b, R.normalize_keypoints(k, (1, 3, 640, 480))) < 1e-7
    save("normalize_keypoints", kpts=k.numpy(), hw_480_640=a.numpy(), hw_640_480=b.numpy())


def gen_sfd2(ref):
    print("SFD2")
    s = ref["sfd2"]
    net = s.ResNet4x(3, 128).eval()
    sd = W.make_state_dict("sfd2", net.state_dict(), seed=7)
    net.load_state_dict(sd, strict=True)
    # (1) full frame 480x640, k = 2048
    img = W.synthetic_image(0)[None]
    cfg = {"min_keypoints": 128, "max_keypoints": 2048}
    with torch.no_grad():
        r = net.extract_local_global({"image": img.clone()}, cfg)
        o = R.sfd2_extract_local_global(sd, img, max_keypoints=2048)
    for k in ("score_map", "desc_map", "mid_features"):
        print(f"  {k}: |ref-oracle| = {maxdiff(r[k], o[k]):.2e}  range [{r[k].min():.4f}, {r[k].max():.4f}]")
        assert maxdiff(r[k], o[k]) < 1e-5
    # canonicalise the reference's topk order (ties unspecified): sort by (score desc, flat idx asc)
    kp_r, sc_r = r["keypoints"][0], r["scores"][0]
    flat = (kp_r[:, 1] * 640 + kp_r[:, 0]).long()
    order = np.lexsort((flat.numpy(), -sc_r.numpy().astype(np.float64)))
    kp_rc, sc_rc = kp_r[order], sc_r[order]
    n_ties = int((sc_rc[1:] == sc_rc[:-1]).sum())
    n_cand = int((R.simple_nms(r["score_map"], 4) >= 0.005).sum())
    print(f"  keypoints {tuple(kp_r.shape)}, candidates {n_cand}, exact score ties among selected {n_ties}")
    assert torch.equal(kp_rc, o["keypoints"][0]) and torch.equal(sc_rc, o["scores"][0]), "keypoint set differs"
    d_r = r["descriptors"][0][:, order]
    assert maxdiff(d_r, o["descriptors"][0]) < 1e-5
    with torch.no_grad():
        sc2, seg = net.sample(r["score_map"], r["mid_features"], kp_rc, norm_desc=False)
        sc2o, sego = R.sfd2_sample(o["score_map"], o["mid_features"], o["keypoints"][0], norm_desc=False)
    assert maxdiff(seg, sego) < 1e-5 and torch.equal(sc2, sc2o)
    nms_ref = s.simple_nms(r["score_map"], 4)
    assert torch.equal(nms_ref, o["nms"])
    save("sfd2_frame0", seed=7, H=480, W=640, k=2048,
         score_sub=sub(r["score_map"], 8192)[0], score_checksum=sub(r["score_map"])[1],
         out4_sub=sub(r["mid_features"], 8192)[0], out4_checksum=sub(r["mid_features"])[1],
         desc_map_sub=sub(r["desc_map"], 8192)[0],
         keypoints=kp_rc.numpy().astype(np.int16), scores=sc_rc.numpy(),
         descriptors_sub=d_r[:, ::16].numpy(), seg_desc_sub=seg[:, ::16].numpy(),
         nms_nonzero=torch.nonzero(nms_ref[0]).numpy().astype(np.int16), n_candidates=n_cand)
    # (2) small frames: batch of 2 (fallback looks at element 0 only), min_keypoints fallback, k > candidates
    img2 = torch.stack([W.synthetic_image(1, 96, 128), W.synthetic_image(2, 96, 128)])
    for tag, cfgx in {"small_k64": dict(max_keypoints=64, min_keypoints=8),
                      "small_fallback": dict(max_keypoints=4096, min_keypoints=100000),
                      "small_all": dict(max_keypoints=4096, min_keypoints=0)}.items():
        with torch.no_grad():
            r2 = net.extract_local_global({"image": img2.clone()}, cfgx)
            o2 = R.sfd2_extract_local_global(sd, img2, **cfgx)
        arrs = {}
        for b in range(2):
            kp, sc = r2["keypoints"][b], r2["scores"][b]
            if len(kp) == cfgx["max_keypoints"]:
                fl = (kp[:, 1] * 128 + kp[:, 0]).long()
                od = np.lexsort((fl.numpy(), -sc.numpy().astype(np.float64)))
                kp, sc = kp[od], sc[od]
            assert torch.equal(kp, o2["keypoints"][b]) and torch.equal(sc, o2["scores"][b]), (tag, b)
            arrs[f"kp{b}"] = kp.numpy().astype(np.int16)
            arrs[f"sc{b}"] = sc.numpy()
        print(f"  {tag}: counts {[len(k) for k in r2['keypoints']]}")
        save(f"sfd2_{tag}", **arrs, score_map=r2["score_map"].numpy(), **{k: v for k, v in cfgx.items()})
    # (3) crafted NMS tie / plateau case
    sm = torch.zeros(1, 40, 48)
    sm[0, 10, 10] = 0.5
    sm[0, 10, 13] = 0.5     # tie inside one window: both are maxima of their windows
    sm[0, 20:23, 20:23] = 0.3  # plateau
    sm[0, 30, 5] = 0.2
    sm[0, 33, 8] = 0.25
    sm[0, 0, 47] = 0.9
    sm += W.uniform(5, "nms/noise", (1, 40, 48), 0.0, 0.01)
    for rad in (3, 4):
        a = s.simple_nms(sm, rad)
        assert torch.equal(a, R.simple_nms(sm, rad))
        save(f"nms_crafted_r{rad}", score=sm.numpy(), nms=a.numpy(), radius=rad)


def _stub_missing_modules():
    """MagicMock stand-ins (harness only) for modules the reference imports at module scope but that the
    pinned functions never touch: pycolmap, cv2, h5py, ...."""
    import importlib
    from unittest import mock
    for name in ['pycolmap', 'cv2', 'h5py', 'open3d', 'pypangolin', 'tensorboardX', 'matplotlib', 'matplotlib.pyplot',
                 'sklearn', 'sklearn.cluster', 'skimage', 'skimage.io', 'tqdm', 'progressbar']:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = mock.MagicMock()


def gen_edges(ref):
    """recogniser epilogue, NearestNeighbor matcher, projection matching, offline extraction."""
    print("edges of the path")
    _stub_missing_modules()
    import localization.frame as ref_frame
    import localization.multimap3d as ref_mm
    import localization.matchers.nearest_neighbor as ref_nn
    # ---- NearestNeighbor (mutual / ratio / distance)
    pair = W.synthetic_match_pair(7, 300, 260)
    d0, d1 = pair["descriptors0"].t()[None].contiguous(), pair["descriptors1"].t()[None].contiguous()
    arrs = {}
    for tag, conf in {"mutual": {}, "ratio": {"ratio_threshold": 0.9, "do_mutual_check": False},
                      "dist": {"distance_threshold": 0.7}}.items():
        model = ref_nn.NearestNeighbor(conf).eval()
        with torch.no_grad():
            r = model({"descriptors0": d0, "descriptors1": d1})
        o = R.nn_forward(d0, d1, **{k: v for k, v in {**ref_nn.NearestNeighbor.default_conf, **conf}.items()})
        assert torch.equal(r["matches0"], o["matches0"]) and maxdiff(r["matching_scores0"], o["matching_scores0"]) < 1e-6, tag
        arrs[f"m0_{tag}"] = r["matches0"].numpy().astype(np.int32)
        arrs[f"s0_{tag}"] = r["matching_scores0"].numpy()
        print(f"  NN {tag}: {(r['matches0'] >= 0).sum().item()} matches")
    save("nn_m300_n260", **arrs)
    # ---- recogniser epilogue on oracle logits
    sd = W.make_state_dict("segnetvit", ref["load"].load_segnet("segnetvit", 113, 256, 15, 1024).state_dict(), seed=7)
    desc, kp, _ = W.synthetic_tokens(3, 400)
    with torch.no_grad():
        logits = R.segnetvit_forward(sd, desc[None], kp[None], (1, 3, 480, 640))[0]
    logits[:, 0] += 6.0   # make background competitive so the 0.95 filter has something to do
    fr = ref_frame.Frame.__new__(ref_frame.Frame)
    fr.keypoints = np.zeros((400, 3))
    fr.descriptors = np.zeros((400, 128))
    fr.initialize_localization_variables = lambda: None
    import io, contextlib
    for thr, tag in ((0.95, "thr095"), (0.2, "thr02"), (0.0, "thr0")):
        fr.keypoints = np.arange(400 * 3, dtype=float).reshape(400, 3)
        fr.descriptors = np.zeros((400, 128))
        with contextlib.redirect_stdout(io.StringIO()):
            fr.add_segmentations(logits.clone(), thr)
        o = R.add_segmentations(logits.clone(), thr)
        kept_ref = (fr.keypoints[:, 0] / 3).astype(int)
        kept_or = np.arange(400) if o["keep"] is None else np.nonzero(o["keep"].numpy())[0]
        assert np.array_equal(kept_ref, kept_or), tag
        assert np.array_equal(fr.seg_ids, o["seg_ids"].numpy()) and np.abs(fr.seg_scores - o["seg_scores"].numpy()).max() < 1e-7
        print(f"  add_segmentations thr={thr}: kept {len(kept_ref)}/400, distinct ids {len(np.unique(fr.seg_ids))}")
        arrs = dict(kept=kept_ref.astype(np.int32), seg_ids=fr.seg_ids.astype(np.int32), seg_scores_sub=fr.seg_scores[:, ::8])
        if tag == "thr095":
            with contextlib.redirect_stdout(io.StringIO()):
                ps_ref = ref_mm.MultiMap3D.process_segmentations(None, torch.from_numpy(fr.seg_scores), topk=20)
            ps_or = R.process_segmentations(torch.from_numpy(fr.seg_scores), topk=20)
            assert len(ps_ref) == len(ps_or)
            for a, b in zip(ps_ref, ps_or):
                assert a[0] == b[0] and np.array_equal(a[1], b[1]) and abs(a[2] - b[2]) < 1e-7
            arrs["ps_sids"] = np.array([a[0] for a in ps_ref], dtype=np.int32)
            arrs["ps_counts"] = np.array([len(a[1]) for a in ps_ref], dtype=np.int32)
            arrs["ps_scores"] = np.array([a[2] for a in ps_ref], dtype=np.float64)
            print(f"  process_segmentations: top sids {arrs['ps_sids'][:6].tolist()} counts {arrs['ps_counts'][:6].tolist()}")
        save(f"segpost_{tag}", **arrs)
    np.savez_compressed(OUT / "segpost_logits.npz", logits=logits.numpy())
    # ---- projection matching (restated from singlemap3d.py:416-433; pinned against the inline reference expressions)
    pair = W.synthetic_match_pair(9, 500, 700)
    q_k, q_d, r_d = pair["keypoints0"], pair["descriptors0"], pair["descriptors1"]
    uv = (pair["keypoints1"] + W.normal(9, "proj/jitter", (700, 2), 3.0)).t().contiguous()
    thr = 8.0
    pe = torch.sqrt(torch.sum((q_k[..., None] - uv[:2][None]) ** 2, dim=1))
    dd = torch.sqrt(2 - 2 * q_d @ r_d.t() + 1e-6)
    dd[pe >= 2 * thr] = dd[pe >= 2 * thr] + 100
    dists, ids = torch.topk(dd, k=2, largest=False, dim=1)
    rmask = ((dists[:, 0] / dists[:, 1]) <= 0.995) * (dists[:, 0] < 100)
    om, oi, od = R.match_by_projection(q_k, q_d, uv, r_d, thr)
    assert torch.equal(rmask, om) and torch.equal(ids[:, 0][rmask], oi[rmask]) and maxdiff(dists, od) < 1e-6
    print(f"  match_by_projection: {int(rmask.sum())}/500 pass the ratio test")
    save("projmatch_m500_n700", ratio_mask=rmask.numpy(), ids=ids[:, 0].numpy().astype(np.int32), dists=dists.numpy(),
         uv=uv.numpy(), threshold=thr)
    # ---- offline extraction (extract_sfd2_return); harness shim: .cuda() is the identity on this CPU-only box
    s = ref["sfd2"]
    net = s.ResNet4x(3, 128).eval()
    sd2 = W.make_state_dict("sfd2", net.state_dict(), seed=7)
    net.load_state_dict(sd2, strict=True)
    raw = W.uniform(77, "offline/img", (1, 3, 96, 128), 0.0, 1.0) * 0.5 + 0.5 * torch.nn.functional.interpolate(
        W.uniform(77, "offline/coarse", (1, 3, 7, 9), 0.0, 1.0), size=(96, 128), mode="bilinear", align_corners=True)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, kw in {"s1": dict(conf_th=0.001, topK=100), "ms": dict(conf_th=0.001, topK=150, scales=[1.0, 0.75])}.items():
            r = s.extract_sfd2_return(net, raw.clone(), **kw)
            o = R.extract_sfd2_return(sd2, raw.clone(), **kw)
            # canonicalise the reference's unstable argsort among equal scores
            def canon(d):
                key = np.lexsort((d["keypoints"][:, 0], d["keypoints"][:, 1], -d["scores"]))
                return d["keypoints"][key], d["scores"][key], d["descriptors"][key]
            rk, rs, rd = canon(r)
            ok_, os_, od_ = canon(o)
            assert np.array_equal(rk, ok_) and np.abs(rs - os_).max() < 1e-7 and np.abs(rd - od_).max() < 1e-5, tag
            print(f"  extract_sfd2_return {tag}: {len(rk)} keypoints")
            save(f"sfd2_offline_{tag}", keypoints=rk, scores=rs, descriptors_sub=rd[:, ::8], **{k: np.array(v) for k, v in kw.items()})
        # mask-labelled variant (sfd2.py:508-571).  Harness shim: the reference spells float64 as np.float, which numpy >= 1.24
        # no longer has.  Segmentation ids: vertical bands, band 0 unlabelled; the three topK regimes + topK = -1.
        if not hasattr(np, "float"):
            np.float = float
        mask = np.zeros((96, 128, 3), np.uint8)
        for band, (b_, g_, r_) in enumerate([(0, 0, 0), (7, 0, 0), (0, 3, 0), (5, 1, 2)]):
            mask[:, band * 32:(band + 1) * 32] = (b_, g_, r_)
        full = R.extract_sfd2_return(sd2, raw.clone(), conf_th=0.001, topK=-1)
        n_all = len(full["scores"])
        n_lab = int((full["keypoints"][:, 0] >= 32).sum())
        for tag, topk in {"few": max(1, n_lab // 2), "all": n_all + 10, "mix": n_lab + (n_all - n_lab) // 2, "none": -1}.items():
            r = s.extract_sfd2_return(net, raw.clone(), conf_th=0.001, mask=mask, topK=topk)
            o = R.extract_sfd2_return(sd2, raw.clone(), conf_th=0.001, mask=mask, topK=topk)
            assert r["labels"].dtype == np.int32 and len(r["labels"]) == len(o["labels"]), tag
            # same multiset of (keypoint, label) records; order equal up to the reference's unstable ties
            def canon(d):
                n = len(d["labels"]) if tag != "none" else None
                lab = d["labels"] if tag != "none" else np.zeros(len(d["scores"]), np.int32)
                key = np.lexsort((d["keypoints"][:, 0], d["keypoints"][:, 1], -d["scores"], -(lab != 0).astype(int)))
                return d["keypoints"][key], d["scores"][key], lab[key], d["descriptors"][key]
            for a, b in zip(canon(r), canon(o)):
                assert np.abs(np.asarray(a, float) - np.asarray(b, float)).max() < 1e-5, tag
            if tag == "none":
                assert np.array_equal(np.sort(r["labels"]), np.sort(o["labels"]))
            print(f"  extract_sfd2_return mask/{tag}: topK {topk} -> {len(o['scores'])} keypoints, {len(o['labels'])} labels")
            save(f"sfd2_offline_mask_{tag}", keypoints=o["keypoints"], scores=o["scores"], labels=o["labels"],
                 descriptors_sub=o["descriptors"][:, ::8], topK=np.array(topk), mask_bands=np.array([[0, 0, 0], [7, 0, 0], [0, 3, 0], [5, 1, 2]]))
    finally:
        torch.Tensor.cuda = orig_cuda


class _MemH5:
    """Harness-only stand-in for the h5py module: ``File(path, mode)`` -> an in-memory group store per path (context manager
    included).  It gives the reference's own I/O code something to run against; no reference code is replaced."""

    def __init__(self):
        from pram_amd.localization.formats import DictStore

        class Store(DictStore):
            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def close(self):
                pass

            def create_group(self, key):
                node, leaf = self._walk(key, True)
                g = node._d[leaf] = Store()
                return g
        self.Store = Store
        self.files = {}

    def File(self, path, mode="r", **kw):
        return self.files.setdefault(str(path), self.Store())


def gen_projection(ref):
    """SingleMap3D.refine_pose_by_projection EXECUTED from the imported reference on a synthetic map
    (object built with __new__, pycolmap's pose solver replaced by a recorder that keeps what it is handed)."""
    print("projection refinement (singlemap3d.py:367-452)")
    _stub_missing_modules()
    import localization.singlemap3d as ref_sm
    from types import SimpleNamespace
    seed, N, M, D = 31, 1500, 600, 128
    imw, imh, thr = 640, 480, 12
    # map points in front of / around the camera; some behind, some far, some outside the image
    xyz = np.stack([W.uniform(seed, "pj/x", (N,), -6.0, 6.0).numpy().astype(np.float64),
                    W.uniform(seed, "pj/y", (N,), -4.5, 4.5).numpy().astype(np.float64),
                    W.uniform(seed, "pj/z", (N,), -2.0, 14.0).numpy().astype(np.float64)], 1)
    xyz[:7, 2] += 150.0                                    # beyond the 100 m depth cut
    descs = torch.nn.functional.normalize(W.normal(seed, "pj/d", (N, D), 1.0), dim=-1).numpy()
    qvec = np.array([0.9987503, 0.02, -0.04, 0.02]); qvec /= np.linalg.norm(qvec)
    tvec = np.array([0.15, -0.1, 0.3])
    fx, fy, cx, cy = 525.0, 530.0, 320.0, 240.0
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Rm = ref_sm.qvec2rotmat(qvec)
    cam_pts = xyz @ Rm.T + tvec
    uv = (cam_pts @ Kmat.T)
    uv = uv[:, :2] / uv[:, 2:3]
    vis = np.nonzero((cam_pts[:, 2] > 0.5) & (uv[:, 0] > 5) & (uv[:, 0] < imw - 5) & (uv[:, 1] > 5) & (uv[:, 1] < imh - 5))[0]
    # query keypoints: noisy projections of visible points (integer pixels like SFD2's), descriptors = noisy copies; plus clutter
    pick = vis[np.argsort(W.uniform(seed, "pj/pick", (len(vis),), 0.0, 1.0).numpy())[:M - 150]]
    kp = np.floor(uv[pick] + W.normal(seed, "pj/kn", (len(pick), 2), 2.0).numpy())
    qd = descs[pick] + W.normal(seed, "pj/dn", (len(pick), D), 0.35 / D ** 0.5).numpy()
    kp = np.concatenate([kp, np.stack([np.floor(W.uniform(seed, "pj/cx", (150,), 4.0, imw - 4.0).numpy()),
                                       np.floor(W.uniform(seed, "pj/cy", (150,), 4.0, imh - 4.0).numpy())], 1)])
    qd = np.concatenate([qd, W.normal(seed, "pj/cd", (150, D), 1.0).numpy()])
    qd = qd / np.linalg.norm(qd, axis=1, keepdims=True)
    keypoints = np.concatenate([kp, W.uniform(seed, "pj/sc", (len(kp), 1), 0.0, 1.0).numpy()], 1).astype(np.float32)
    # the reference object: two covisible reference frames sharing the points
    sm = ref_sm.SingleMap3D.__new__(ref_sm.SingleMap3D)
    ids = np.arange(N) * 3 + 11                            # non-contiguous point ids
    sm.point3Ds = {int(i): SimpleNamespace(xyz=xyz[j], descriptor=descs[j], seg_id=int(j % 17)) for j, i in enumerate(ids)}
    sm.reference_frames = {7: SimpleNamespace(point3D_ids=ids[:900]), 9: SimpleNamespace(point3D_ids=ids[600:])}
    sm.covisible_graph = {7: [9], 9: [7]}
    sm.config = {"localization": {"threshold": thr, "covisibility_frame": 2}}
    sm.find_reference_frames = lambda matched_point3D_ids, candidate_frame_ids=None: [7, 9]
    handed = {}

    def solver(pts2d, pts3d, camera, estimation_options=None, refinement_options=None):
        handed.update(pts2d=np.array(pts2d), pts3d=np.array(pts3d))
        from unittest import mock
        return {"num_inliers": 0, "inliers": np.zeros(len(pts2d), bool), "cam_from_world": mock.MagicMock()}
    ref_sm.pycolmap.absolute_pose_estimation = solver
    q_frame = SimpleNamespace(qvec=qvec, tvec=tvec, camera=SimpleNamespace(width=imw, height=imh), get_intrinsics=lambda: Kmat,
                              reference_frame_id=7, keypoints=keypoints, descriptors=qd.astype(np.float32))
    import io, contextlib
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self        # harness shim: this box has no GPU
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ret = sm.refine_pose_by_projection(q_frame)
    finally:
        torch.Tensor.cuda = orig_cuda
    all_ids = np.unique(ids)                               # the method sorts the point ids
    order = np.searchsorted(ids, all_ids)
    Tcw = np.eye(4)
    Tcw[:3, :3], Tcw[:3, 3] = Rm, tvec
    oo = R.refine_matches_by_projection(keypoints, qd.astype(np.float32), xyz[order], descs[order], Kmat, Tcw, imw, imh, thr)
    mk_ids = oo["matched_keypoint_ids"].numpy()
    mp_ids = oo["matched_point_ids"].numpy()
    assert np.array_equal(ret["matched_keypoint_ids"], mk_ids), "matched keypoint ids differ"
    assert np.array_equal(ret["matched_point3D_ids"], all_ids[mp_ids]), "matched map points differ"
    assert np.array_equal(ret["matched_xyzs"], xyz[order][mp_ids]) and np.array_equal(handed["pts3d"], xyz[order][mp_ids])
    assert np.array_equal(handed["pts2d"], keypoints[mk_ids][:, :2] + 0.5)
    assert np.array_equal(ret["matched_sids"], np.array([sm.point3Ds[int(i)].seg_id for i in all_ids])[mp_ids])
    n_keep = int(oo["point_mask"].sum())
    print(f"  {N} map points, {n_keep} in the frustum, {len(mk_ids)}/{len(keypoints)} keypoints pass the ratio test "
          f"({int((np.isin(all_ids[mp_ids], ids[pick])).sum())} of them planted)")
    save("projrefine_n1500_m600", point_mask=oo["point_mask"].numpy(), matched_keypoint_ids=mk_ids.astype(np.int32),
         matched_point_ids=mp_ids.astype(np.int32), dists=oo["dists"].numpy(), seed=seed, threshold=thr, imw=imw, imh=imh,
         qvec=qvec, tvec=tvec, K=Kmat, xyz=xyz[order], keypoints=keypoints, pick_ids=np.searchsorted(all_ids, ids[pick]).astype(np.int32),
         descs_key=np.array([seed, N, M, D]))


def gen_formats(ref):
    """the reference's own file-format code (FeaturePairsDataset.__getitem__, writer_fn,
    find_unique_new_pairs, the export loop of extract_features.main, parsers) executed against an in-memory h5py stand-in."""
    print("edge formats (match_features_batch.py:89-129,165-186; extract_features.py:200-238; parsers.py:57-80)")
    _stub_missing_modules()
    from pathlib import Path
    from types import SimpleNamespace
    import localization.match_features_batch as ref_mfb
    import localization.extract_features as ref_ef
    import colmap_utils.parsers as ref_par
    from pram_amd.localization import formats as F2
    mem = _MemH5()
    ref_mfb.h5py = mem
    ref_ef.h5py = mem
    # ---- names / pair lists
    for a, b in (("db/1.jpg", "query/night/2.jpg"), ("a", "b"), ("x/y/z.png", "x/y/z.png")):
        assert ref_par.names_to_pair(a, b) == R.names_to_pair(a, b) == F2.names_to_pair(a, b)
        assert ref_par.names_to_pair_old(a, b) == R.names_to_pair(a, b, separator="_") == F2.names_to_pair_old(a, b)
    with tempfile.TemporaryDirectory() as td:
        pf = Path(td) / "pairs.txt"
        lines = ["q/1.jpg db/3.jpg", "q/1.jpg db/4.jpg", "q/2.jpg db/3.jpg", "db/3.jpg q/1.jpg", "q/1.jpg db/3.jpg"]
        pf.write_text("\n".join(lines) + "\n")
        want = ref_par.parse_retrieval(pf)
        assert want == R.parse_retrieval(pf) == F2.parse_retrieval(pf)
        pairs_all = [(q, r) for q, rs in want.items() for r in rs]
        # ---- feature export: the loop of extract_features.main on two fake frames
        feats = {}
        for i, (name, n, hw, orig) in enumerate((("db/3.jpg", 57, (480, 640), (1280, 960)), ("q/1.jpg", 40, (240, 320), (320, 240)))):
            feats[name] = dict(image=np.zeros((3,) + hw, np.float32), original_size=np.array(orig),
                               pred={"keypoints": np.floor(W.uniform(40 + i, "ff/k", (n, 2), 4.0, 200.0).numpy()).astype(np.float64),
                                     "scores": W.uniform(40 + i, "ff/s", (n,), 0.0, 1.0).numpy().astype(np.float64),
                                     "descriptors": W.normal(40 + i, "ff/d", (n, 128), 1.0).numpy().astype(np.float64)})
        names = list(feats)
        ref_ef.args = SimpleNamespace(image_list=None)
        ref_ef.get_model = lambda **kw: (SimpleNamespace(cuda=lambda: None), None)
        state = {"i": 0}

        def fake_extractor(model, img, **kw):
            d = feats[names[state["i"]]]["pred"]
            state["i"] += 1
            return {k: v.copy() for k, v in d.items()}
        ref_ef.get_model = lambda **kw: (SimpleNamespace(cuda=lambda: SimpleNamespace()), fake_extractor)
        ref_ef.ImageDataset = lambda *a, **k: [{"name": nm, "image": feats[nm]["image"], "original_size": feats[nm]["original_size"]} for nm in names]
        real_loader = torch.utils.data.DataLoader
        ref_ef.torch.utils.data.DataLoader = lambda ds, num_workers=0: real_loader(ds, num_workers=0)
        try:
            conf = {"output": "feats-x", "preprocessing": {}, "model": {"name": "resnet4x", "model_fn": "-", "use_stability": False, "outdim": 128,
                                                                         "max_keypoints": 100, "conf_th": 0.001, "scales": [1.0]}}
            fpath = ref_ef.main(conf, Path(td) / "images", Path(td) / "out")
        finally:
            ref_ef.torch.utils.data.DataLoader = real_loader
        fstore = mem.files[str(fpath)]
        arrs = {}
        for nm in names:
            grp = fstore[nm]
            enc_o = R.feature_encode({k: v.copy() for k, v in feats[nm]["pred"].items()}, (1, 3) + feats[nm]["image"].shape[1:], feats[nm]["original_size"])
            enc_p = F2.encode_features({k: v.copy() for k, v in feats[nm]["pred"].items()}, feats[nm]["image"].shape[1:], feats[nm]["original_size"])
            for k in ("descriptors", "keypoints", "scores", "image_size"):
                a = np.asarray(grp[k])
                assert a.dtype == np.asarray(enc_o[k]).dtype == np.asarray(enc_p[k]).dtype and np.array_equal(a, enc_o[k]) and np.array_equal(a, enc_p[k]), (nm, k)
                arrs[f"feat_{names.index(nm)}_{k}"] = a
        print(f"  feature export: {len(names)} groups, keypoint dtype {arrs['feat_0_keypoints'].dtype}, descriptors {arrs['feat_0_descriptors'].shape}")
        # ---- reading a pair back: FeaturePairsDataset.__getitem__
        ds = ref_mfb.FeaturePairsDataset([("q/1.jpg", "db/3.jpg")], fpath, fpath)
        item = ds[0]
        it_o = R.pair_item({k: np.asarray(v) for k, v in fstore["q/1.jpg"].items()}, {k: np.asarray(v) for k, v in fstore["db/3.jpg"].items()})
        it_p = F2.read_feature_pair(fstore, "q/1.jpg", fstore, "db/3.jpg")
        assert set(item) == set(it_o) == set(it_p)
        for k in item:
            assert item[k].dtype == it_o[k].dtype == it_p[k].dtype and tuple(item[k].shape) == tuple(it_o[k].shape) == tuple(it_p[k].shape), k
            if k not in ("image0", "image1"):
                assert torch.equal(item[k], it_o[k]) and torch.equal(item[k], it_p[k]), k
                arrs[f"item_{k}"] = item[k].numpy()
        arrs["item_image0_shape"], arrs["item_image1_shape"] = np.array(item["image0"].shape), np.array(item["image1"].shape)
        # ---- writer_fn: int16 / fp16 records, replacing an existing group
        mpath = Path(td) / "matches.h5"
        m0 = torch.tensor([[5, -1, 40000, 7, -1, 123]])                 # 40000 wraps in int16 exactly like .short()
        s0 = torch.tensor([[0.123456, 0.0, 0.99951171875, 1e-5, 0.5, 0.333333]])
        pname = ref_par.names_to_pair("q/1.jpg", "db/3.jpg")
        ref_mfb.writer_fn((pname, {"matches0": torch.zeros(1, 3, dtype=torch.long)}), mpath)      # a stale group ...
        ref_mfb.writer_fn((pname, {"matches0": m0, "matching_scores0": s0}), mpath)                # ... is replaced
        mstore = mem.files[str(mpath)]
        enc_o = R.writer_encode({"matches0": m0, "matching_scores0": s0})
        enc_p = F2.encode_matches(m0[0], s0[0])
        for k in ("matches0", "matching_scores0"):
            a = np.asarray(mstore[pname][k])
            assert a.dtype == enc_o[k].dtype == enc_p[k].dtype and np.array_equal(a, enc_o[k]) and np.array_equal(a, enc_p[k]), k
            arrs[f"write_{k}"] = a
        mstore2 = F2.DictStore()
        F2.write_matches(mstore2, pname, {"matches0": np.zeros(3, np.int16)})
        F2.write_matches(mstore2, pname, enc_p)
        assert np.array_equal(np.asarray(mstore2[pname]["matches0"]), arrs["write_matches0"])
        # ---- find_unique_new_pairs: without a file, and against the stored group (both orders / both namings)
        mpath.write_text("")                                            # match_path.exists() is what the reference tests
        assert set(ref_mfb.find_unique_new_pairs(pairs_all)) == R.find_unique_new_pairs(pairs_all) == set(F2.find_unique_new_pairs(pairs_all))
        got = set(ref_mfb.find_unique_new_pairs(pairs_all, mpath))
        assert got == R.find_unique_new_pairs(pairs_all, mstore) == set(F2.find_unique_new_pairs(pairs_all, mstore)), got
        assert ("q/1.jpg", "db/3.jpg") not in got and ("db/3.jpg", "q/1.jpg") not in got and len(got) == 2
        print(f"  pairs: {len(pairs_all)} listed -> {len(R.find_unique_new_pairs(pairs_all))} unique -> {len(got)} new")
    save("formats_pinned", **arrs, write_m0=m0.numpy(), write_s0=s0.numpy())


def gen_schema(ref):
    """State-dict key/shape schema of the reference modules (what load_state_dict(strict=True) needs)."""
    print("state-dict schema")
    mods = {
        "sfd2": ref["sfd2"].ResNet4x(3, 128),
        "segnetvit_c113": ref["load"].load_segnet("segnetvit", 113, 256, 15, 1024),
        "gml": ref["gml"].GML({}),
        "adagml": ref["adagml"].AdaGML({}),
    }
    arrs = {}
    for name, m in mods.items():
        sd = m.state_dict()
        arrs[name + "_keys"] = np.array(list(sd.keys()))
        arrs[name + "_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        print(f"  {name}: {len(sd)} tensors, {sum(v.numel() for v in sd.values())} elements")
    save("schema", **arrs)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    only = sys.argv[1:] or ["schema", "normalize", "sinkhorn", "segnetvit", "gml", "adagml", "adagml_run", "sfd2", "edges", "projection", "formats"]
    for name in only:
        globals()[f"gen_{name}"](ref)
    print("all reference-vs-oracle checks passed; fixtures written to", OUT)


if __name__ == "__main__":
    main()
