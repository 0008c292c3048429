"""CPU: host logic, state-dict schema, C-ABI library loads and exports every declared symbol,
and the product path refuses to run without the GPU (no CPU fallback)."""
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from pram_amd import weights as W
from tests import helpers as H

ROOT = Path(__file__).resolve().parents[1]


def test_header_symbols_exported(hip_lib):
    hdr = (ROOT / "include" / "pram_hip.h").read_text()
    names = sorted(set(re.findall(r"\b(pram_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 24
    from pram_amd import _lib
    assert sorted(_lib.exported_symbols()) == names, set(names) ^ set(_lib.exported_symbols())
    for n in names:
        assert hasattr(hip_lib, n), n
    assert hip_lib.pram_hip_version() >= 100
    assert hip_lib.pram_sinkhorn_workspace_bytes(1, 2048, 2048) > 2049 * 2049 * 4


def test_state_dict_schema_matches_reference(golden):
    """keys and shapes recorded from the reference modules == the host modules' (strict load works)."""
    from pram_amd.nets.adagml import AdaGML
    from pram_amd.nets.gml import GML
    from pram_amd.nets.load_segnet import load_segnet
    from pram_amd.nets.sfd2 import ResNet4x
    g = golden("schema")
    mods = {"sfd2": ResNet4x(), "segnetvit_c113": load_segnet("segnetvit", 113, 256, 15, 1024), "gml": GML({}), "adagml": AdaGML({})}
    for name, m in mods.items():
        sd = m.state_dict()
        assert list(sd.keys()) == list(g[name + "_keys"]), name
        assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g[name + "_shapes"]), name


def test_weights_deterministic():
    a = W.uniform(7, "x", (1000,))
    b = W.uniform(7, "x", (1000,))
    assert torch.equal(a, b) and not torch.equal(a, W.uniform(8, "x", (1000,)))
    assert abs(float(a.mean())) < 0.1 and float(a.min()) >= -1 and float(a.max()) < 1
    sd = H.gml_sd()
    assert float(sd["bin_score"]) == 1.0
    # fixed fingerprint: any change to the generator invalidates every golden fixture
    assert abs(float(sd["input_proj.weight"].double().abs().sum()) - 2512.004) < 0.01, float(sd["input_proj.weight"].double().abs().sum())


def test_qkv_packing_is_a_permutation():
    from pram_amd.nets import _blocks as blk
    sd = {"p.qkv.weight": torch.arange(768 * 256, dtype=torch.float32).view(768, 256), "p.qkv.bias": torch.arange(768, dtype=torch.float32)}
    sd.update({"p.proj.weight": torch.zeros(256, 256), "p.proj.bias": torch.zeros(256), "p.mlp.0.weight": torch.zeros(512, 512),
               "p.mlp.0.bias": torch.zeros(512), "p.mlp.1.weight": torch.ones(512), "p.mlp.1.bias": torch.zeros(512),
               "p.mlp.3.weight": torch.zeros(256, 512), "p.mlp.3.bias": torch.zeros(256)})
    pk = blk.pack_self_block(sd, "p", "cpu")
    assert sorted(pk["qkv_b"].tolist()) == list(range(768))
    # head 1, q, packed column 3 = original even dim 6 -> row 1*192 + 6*3 + 0
    assert pk["qkv_b"][1 * 64 + 3].item() == 1 * 192 + 6 * 3 + 0
    # head 2, k, packed column 32+5 = original odd dim 11 -> row 2*192 + 11*3 + 1
    assert pk["qkv_b"][256 + 2 * 64 + 37].item() == 2 * 192 + 11 * 3 + 1
    # head 3, v, natural dim 9
    assert pk["qkv_b"][512 + 3 * 64 + 9].item() == 3 * 192 + 9 * 3 + 2


def test_keypoint_norm_constants_quirk():
    from pram_amd.nets.utils import keypoint_norm_constants, normalize_keypoints
    assert keypoint_norm_constants((1, 3, 480, 640)) == (320.0, 240.0, 448.0)
    assert keypoint_norm_constants((1, 3, 640, 480)) == (240.0, 320.0, 448.0)      # (W,H)-swapped tuple: centre (H/2, W/2)
    k = torch.tensor([[[320.0, 240.0]]])
    assert torch.allclose(normalize_keypoints(k, (1, 3, 640, 480)), torch.tensor([[[80 / 448.0, -80 / 448.0]]]))


def test_no_cpu_fallback():
    from pram_amd._lib import PramHipError
    from pram_amd.nets.load_segnet import load_segnet
    m = load_segnet("segnetvit", 113, 256, 15, 1024).eval()
    desc, kp, _ = W.synthetic_tokens(0, 32)
    with pytest.raises(PramHipError):
        m({"seg_descriptors": desc[None], "keypoints": kp[None], "image": torch.empty(1, 3, 480, 640)})
    from pram_amd.nets.gml import GML
    d, _ = H.pair_data(0, 32, 32)
    with pytest.raises(PramHipError):
        GML({}).eval()(d)
    from pram_amd.nets.sfd2 import ResNet4x
    with pytest.raises(PramHipError):
        ResNet4x().eval().extract_local_global({"image": torch.zeros(1, 3, 64, 64)})


def test_load_segnet_and_plugin_registry(tmp_path):
    from pram_amd.nets.load_segnet import load_segnet
    with pytest.raises(NotImplementedError):
        load_segnet("segnet", 113, 256, 15, 1024)
    import pram_amd.localization.matchers as matchers
    from pram_amd.localization.base_model import BaseModel, dynamic_load
    from pram_amd.localization.match_features_batch import confs
    for name in ("gml", "adagml"):
        cls = dynamic_load(matchers, confs[name]["model"]["name"])
        assert issubclass(cls, BaseModel)
    wp = tmp_path / "gml.pth"
    torch.save({"model": H.gml_sd()}, wp)
    m = dynamic_load(matchers, "gml")({"name": "gml", "weight_path": str(wp), "sinkhorn_iterations": 20})
    assert m.net.sinkhorn_iterations == 20 and m.net.match_threshold == 0.2
    bad = dict(H.gml_sd())
    bad.pop("bin_score")
    torch.save({"model": bad}, wp)
    with pytest.raises(RuntimeError):
        dynamic_load(matchers, "gml")({"name": "gml", "weight_path": str(wp)})


def test_config_values_of_7scenes(golden):
    """C1 plumbing: the values the recogniser factory needs from configs/config_train_7scenes_sfd2.yaml."""
    import yaml
    cfg = yaml.safe_load((ROOT / "tests" / "golden" / "config_7scenes_values.yaml").read_text())
    from pram_amd.nets.load_segnet import load_segnet
    m = load_segnet(cfg["network"], cfg["n_class"], 256 if cfg["use_mid_feature"] else 128, cfg["layers"], cfg["output_dim"])
    assert m.config["n_class"] == 113 and m.n_layers == 15 and m.config["output_dim"] == 1024
    assert cfg["localization"]["matching_method"] == "gml"


def test_gm_module_mirrors_the_reference_state():
    """nets/gm.py: the class is unconstructible in the reference and says so here; its free functions are GML's."""
    import pytest
    from pram_amd.nets import gm, gml
    assert gm.sink_algorithm is gml.sink_algorithm and gm.dual_softmax is gml.dual_softmax
    with pytest.raises(TypeError, match="cannot be constructed"):
        gm.GM({})
