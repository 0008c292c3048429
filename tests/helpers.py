"""Shared test helpers: deterministic weights / inputs and oracle state dicts."""
from __future__ import annotations

import functools

import torch

from pram_amd import weights as W


@functools.lru_cache(maxsize=None)
def segnet_sd(n_class: int = 113):
    from pram_amd.nets.load_segnet import load_segnet
    m = load_segnet('segnetvit', n_class, 256, 15, 1024)
    return W.make_state_dict('segnetvit', m.state_dict(), seed=7)


@functools.lru_cache(maxsize=None)
def gml_sd():
    from pram_amd.nets.gml import GML
    return W.make_state_dict('gml', GML({}).state_dict(), seed=7)


@functools.lru_cache(maxsize=None)
def adagml_sd():
    from pram_amd.nets.adagml import AdaGML
    return W.make_state_dict('adagml', AdaGML({}).state_dict(), seed=7)


@functools.lru_cache(maxsize=None)
def sfd2_sd():
    from pram_amd.nets.sfd2 import ResNet4x
    return W.make_state_dict('sfd2', ResNet4x().state_dict(), seed=7)


def pair_data(idx, m, n, key="image_shape", device="cpu"):
    d = W.synthetic_match_pair(idx, m, n)
    data = {k: v[None].to(device) for k, v in d.items() if k != "gt"}
    if key == "image_shape":
        data["image_shape0"] = (1, 3, 640, 480)
        data["image_shape1"] = (1, 3, 640, 480)
    else:
        data["image0"] = torch.empty(1, 1, 480, 640)
        data["image1"] = torch.empty(1, 1, 480, 640)
    return data, d["gt"]


def subsample(t: torch.Tensor, n: int = 4096):
    f = t.reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float()


def maxdiff(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max()) if a.numel() else 0.0
