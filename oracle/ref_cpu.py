"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement (plain torch ops, functional style over a ``state_dict``) of the PRAM
per-query hot path.  It is the checker for the HIP kernels: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED.  Every function below is checked against the imported reference
(``/root/reference``, torch CPU) by ``oracle/gen_golden.py`` (run in the build container; the
reference cannot travel to the GPU box) and against the committed fixtures in ``tests/golden/``
by ``tests/test_oracle_golden.py``.  ``nets/gm.py``'s ``GM`` class is unconstructible in the
reference — its shared free functions are pinned through ``nets/gml.py``.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
EPS_SINK = 1e-8  # nets/gml.py:17


# ------------------------------------------------------------------------------------------
# shared helpers
# ------------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln_gelu(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # nn.LayerNorm(eps=1e-5, affine) + exact (erf) GELU — nets/segnetvit.py:92-93
    c = x.shape[-1]
    return F.gelu(F.layer_norm(x, (c,), sd[p + ".weight"], sd[p + ".bias"], 1e-5))


def normalize_keypoints(kpts: torch.Tensor, image_shape) -> torch.Tensor:
    """nets/utils.py:17-24.  ``_,_,height,width = image_shape``; centre (W/2,H/2), scale 0.7*max.
    The matcher call sites pass (1,3,width,height) (localization/singlemap3d.py:147) — the
    formula is applied to whatever tuple is given, the quirk is not corrected."""
    _, _, height, width = image_shape
    size = torch.tensor([float(width), float(height)], dtype=kpts.dtype)
    center = size / 2
    scaling = size.max() * 0.7
    return (kpts - center) / scaling


def fourier_encoding(Wr: torch.Tensor, norm_kpts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """nets/segnetvit.py:35-40 / nets/gml.py:68-73: cos/sin of Wr·xy.  Returned un-interleaved
    ([B,N,32] each); ``repeat_interleave(2)`` is applied implicitly by :func:`rotary`."""
    proj = norm_kpts @ Wr.t()
    return torch.cos(proj), torch.sin(proj)


def rotary(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """nets/segnetvit.py:15-23: t*cos + rotate_half(t)*sin, pairs (2i,2i+1) -> (-t[2i+1], t[2i]).
    t: [B,H,N,64]; cos/sin: [B,N,32]."""
    b, h, n, d = t.shape
    tp = t.reshape(b, h, n, d // 2, 2)
    c = cos[:, None]
    s = sin[:, None]
    e = tp[..., 0] * c + (-tp[..., 1]) * s
    o = tp[..., 1] * c + tp[..., 0] * s
    return torch.stack([e, o], -1).reshape(b, h, n, d)


def attention(q, k, v, scale: float, want_colmean: bool = False):
    """nets/segnetvit.py:73-76 (and nets/adagml.py:145-148 for the column mean)."""
    attn = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * scale, -1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    if want_colmean:
        return out, attn.mean(1).mean(1)
    return out


def self_block(sd: SD, p: str, x: torch.Tensor, cos, sin, heads: int = 4, want_colmean: bool = False):
    """SelfMultiHeadAttention.forward — nets/segnetvit.py:97-106, nets/gml.py:128-137,
    nets/adagml.py:170-179.  qkv channel c = h*192 + d*3 + {q,k,v}."""
    b, n, c = x.shape
    dh = c // heads
    qkv = _lin(sd, p + ".qkv", x).reshape(b, n, heads, dh, 3).permute(0, 2, 1, 3, 4)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = rotary(q, cos, sin)
    k = rotary(k, cos, sin)
    r = attention(q, k, v, dh ** -0.5, want_colmean)
    ctx = r[0] if want_colmean else r
    msg = _lin(sd, p + ".proj", ctx.permute(0, 2, 1, 3).reshape(b, n, c))
    h1 = _ln_gelu(sd, p + ".mlp.1", _lin(sd, p + ".mlp.0", torch.cat([x, msg], -1)))
    y = x + _lin(sd, p + ".mlp.3", h1)
    return (y, r[1]) if want_colmean else y


def cross_block(sd: SD, p: str, x0: torch.Tensor, x1: torch.Tensor, heads: int = 4, want_colmean: bool = False):
    """CrossMultiHeadAttention.forward — nets/gml.py:164-186 / nets/adagml.py:207-229.
    Shared to_qk, both sides scaled by dh^-1/4, one ``sim``; row softmax and column softmax."""
    b, m, c = x0.shape
    n = x1.shape[1]
    dh = c // heads
    s4 = (dh ** -0.5) ** 0.5

    def split(t, l):
        return t.reshape(b, l, heads, dh).permute(0, 2, 1, 3)

    qk0 = split(_lin(sd, p + ".to_qk", x0), m) * s4
    qk1 = split(_lin(sd, p + ".to_qk", x1), n) * s4
    v0 = split(_lin(sd, p + ".to_v", x0), m)
    v1 = split(_lin(sd, p + ".to_v", x1), n)
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    a01 = torch.softmax(sim, -1)
    a10 = torch.softmax(sim.transpose(-2, -1), -1)
    m0 = torch.einsum("bhij,bhjd->bhid", a01, v1).permute(0, 2, 1, 3).reshape(b, m, c)
    m1 = torch.einsum("bhji,bhid->bhjd", a10, v0).permute(0, 2, 1, 3).reshape(b, n, c)
    m0 = _lin(sd, p + ".proj", m0)
    m1 = _lin(sd, p + ".proj", m1)

    def mlp(x, msg):
        return x + _lin(sd, p + ".mlp.3", _ln_gelu(sd, p + ".mlp.1", _lin(sd, p + ".mlp.0", torch.cat([x, msg], -1))))

    y0, y1 = mlp(x0, m0), mlp(x1, m1)
    if want_colmean:
        # nets/adagml.py:229: (mean attn10 over heads, rows) -> per set-0 token; (attn01) -> per set-1 token
        return y0, y1, a10.mean(1).mean(1), a01.mean(1).mean(1)
    return y0, y1


# ------------------------------------------------------------------------------------------
# SegNetViT — nets/segnetvit.py:124-203
# ------------------------------------------------------------------------------------------
def segnetvit_forward(sd: SD, seg_descriptors: torch.Tensor, keypoints: Optional[torch.Tensor] = None,
                      image_shape=None, norm_keypoints: Optional[torch.Tensor] = None,
                      n_layers: int = 15, heads: int = 4, probes: Optional[dict] = None) -> torch.Tensor:
    if norm_keypoints is None:
        if image_shape is None:
            raise ValueError("Require image shape for keypoint coordinate normalization")
        norm_keypoints = normalize_keypoints(keypoints, image_shape)
    cos, sin = fourier_encoding(sd["kenc.Wr.weight"], norm_keypoints)
    x = _lin(sd, "input_proj", seg_descriptors)
    for i in range(n_layers):
        x = self_block(sd, f"gnn.layers.{i}", x, cos, sin, heads)
        if probes is not None:
            probes[f"layer{i}"] = x
    if probes is not None and "sc.0.weight" in sd:      # with_sc head on the same tokens — nets/segnetvit.py:166-172,199-201
        probes["sc"] = _lin(sd, "sc.3", _ln_gelu(sd, "sc.1", _lin(sd, "sc.0", x)))
    h = _ln_gelu(sd, "seg.1", _lin(sd, "seg.0", x))
    return _lin(sd, "seg.3", h)


# ------------------------------------------------------------------------------------------
# Sinkhorn / matches — nets/gml.py:20-46, 304-319 (== nets/gm.py:17-43,249-264, nets/adagml.py:24-50,491-506)
# ------------------------------------------------------------------------------------------
def _augment(M: torch.Tensor, dustbin: torch.Tensor) -> torch.Tensor:
    b, m, n = M.shape
    bins = dustbin.reshape(1, 1, 1).to(M)
    M = torch.cat([M, bins.expand(b, m, 1)], -1)
    return torch.cat([M, bins.expand(b, 1, n + 1)], -2)


def sink_algorithm(M: torch.Tensor, dustbin: torch.Tensor, iteration: int) -> torch.Tensor:
    """Plain-domain Sinkhorn: p = softmax_row(aug); u,v <- 1; it x {u = r/(p v + eps); v = c/(p^T u + eps)};
    p*u*v with r = (1,..,1,m+1), c = (1,..,1,n+1).  nets/gml.py:27-46."""
    A = _augment(M, dustbin)
    b, m1, n1 = A.shape
    r = torch.ones(b, m1, dtype=M.dtype)
    r[:, -1] = m1
    c = torch.ones(b, n1, dtype=M.dtype)
    c[:, -1] = n1
    p = torch.softmax(A, -1)
    u = torch.ones_like(r)
    v = torch.ones_like(c)
    for _ in range(iteration):
        u = r / ((p * v[:, None, :]).sum(-1) + EPS_SINK)
        v = c / ((p * u[:, :, None]).sum(-2) + EPS_SINK)
    return p * u[:, :, None] * v[:, None, :]


def dual_softmax(M: torch.Tensor, dustbin: torch.Tensor) -> torch.Tensor:
    """nets/gml.py:20-24."""
    A = _augment(M, dustbin)
    return torch.exp(torch.log_softmax(A, -1) + torch.log_softmax(A, 1))


def compute_matches(scores: torch.Tensor, p: float = 0.2):
    """nets/gml.py:304-319.  argmax = first occurrence (torch CPU ``max(dim)`` semantics)."""
    inner = scores[:, :-1, :-1]
    max0, max1 = inner.max(2), inner.max(1)
    i0, i1 = max0.indices, max1.indices
    ar0 = torch.arange(i0.shape[1])[None]
    ar1 = torch.arange(i1.shape[1])[None]
    mutual0 = ar0 == i1.gather(1, i0)
    mutual1 = ar1 == i0.gather(1, i1)
    zero = scores.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values, zero)
    ms1 = torch.where(mutual1, ms0.gather(1, i1), zero)
    valid0 = mutual0 & (ms0 > p)
    valid1 = mutual1 & valid0.gather(1, i1)
    i0 = torch.where(valid0, i0, i0.new_tensor(-1))
    i1 = torch.where(valid1, i1, i1.new_tensor(-1))
    return i0, i1, ms0, ms1


# ------------------------------------------------------------------------------------------
# GML — nets/gml.py:189-319
# ------------------------------------------------------------------------------------------
def _matcher_norm_kpts(data: dict):
    if "norm_keypoints0" in data and "norm_keypoints1" in data:
        return data["norm_keypoints0"], data["norm_keypoints1"]
    if "image0" in data and "image1" in data:
        return (normalize_keypoints(data["keypoints0"], data["image0"].shape).float(),
                normalize_keypoints(data["keypoints1"], data["image1"].shape).float())
    if "image_shape0" in data and "image_shape1" in data:
        return (normalize_keypoints(data["keypoints0"], data["image_shape0"]).float(),
                normalize_keypoints(data["keypoints1"], data["image_shape1"]).float())
    raise ValueError("Require image shape for keypoint coordinate normalization")


def gml_produce_matches(sd: SD, data: dict, p: float = 0.2, n_layers: int = 9, sinkhorn_iterations: int = 20,
                        with_sinkhorn: bool = True, probes: Optional[dict] = None) -> dict:
    nk0, nk1 = _matcher_norm_kpts(data)
    d0 = _lin(sd, "input_proj", data["descriptors0"])
    d1 = _lin(sd, "input_proj", data["descriptors1"])
    c0, s0 = fourier_encoding(sd["poseenc.Wr.weight"], nk0)
    c1, s1 = fourier_encoding(sd["poseenc.Wr.weight"], nk1)
    for i in range(n_layers):
        d0 = self_block(sd, f"self_attn.{i}", d0, c0, s0)
        d1 = self_block(sd, f"self_attn.{i}", d1, c1, s1)
        d0, d1 = cross_block(sd, f"cross_attn.{i}", d0, d1)
    dim = d0.shape[-1]
    md0 = _lin(sd, f"out_proj.{n_layers - 1}", d0) / dim ** 0.25
    md1 = _lin(sd, f"out_proj.{n_layers - 1}", d1) / dim ** 0.25
    dist = torch.einsum("bmd,bnd->bmn", md0, md1)
    score = sink_algorithm(dist, sd["bin_score"], sinkhorn_iterations) if with_sinkhorn \
        else dual_softmax(dist, sd["bin_score"])
    if probes is not None:
        probes.update(desc0=d0, desc1=d1, dist=dist, score=score)
    i0, i1, ms0, ms1 = compute_matches(score, p)
    return {"matches0": i0, "matches1": i1, "matching_scores0": ms0, "matching_scores1": ms1}


# ------------------------------------------------------------------------------------------
# AdaGML — nets/adagml.py:232-536
# ------------------------------------------------------------------------------------------
def pooling_layer(sd: SD, p: str, x: torch.Tensor, score: torch.Tensor) -> torch.Tensor:
    """PoolingLayer.forward — nets/adagml.py:132-138."""
    s = _lin(sd, p + ".score_enc.3", _ln_gelu(sd, p + ".score_enc.1", _lin(sd, p + ".score_enc.0", score)))
    xx = _lin(sd, p + ".proj", x)
    h = _ln_gelu(sd, p + ".predict.1", _lin(sd, p + ".predict.0", torch.cat([xx, s], -1)))
    return torch.sigmoid(_lin(sd, p + ".predict.3", h))


def adagml_confidence_threshold(layer_index: int, n_layers: int = 9) -> float:
    """nets/adagml.py:516-520."""
    return float(np.clip(0.5 + 0.1 * np.exp(-4.0 * layer_index / n_layers), 0, 1))


def adagml_produce_matches(sd: SD, data: dict, p: float = 0.2, n_layers: int = 9, n_min_tokens: int = 256,
                           sinkhorn_iterations: int = 20, probes: Optional[dict] = None) -> dict:
    """AdaGML.produce_matches — nets/adagml.py:307-404 (B = 1 semantics: mask indexing drops the batch)."""
    if "norm_keypoints0" in data and "norm_keypoints1" in data:
        nk0, nk1 = data["norm_keypoints0"], data["norm_keypoints1"]
    elif "image0" in data and "image1" in data:
        nk0 = normalize_keypoints(data["keypoints0"], data["image0"].shape)
        nk1 = normalize_keypoints(data["keypoints1"], data["image1"].shape)
    elif "image_shape0" in data and "image_shape1" in data:
        nk0 = normalize_keypoints(data["keypoints0"], data["image_shape0"])
        nk1 = normalize_keypoints(data["keypoints1"], data["image_shape1"])
    else:
        raise ValueError("Require image shape for keypoint coordinate normalization")
    d0 = _lin(sd, "input_proj", data["descriptors0"])
    d1 = _lin(sd, "input_proj", data["descriptors1"])
    c0, s0 = fourier_encoding(sd["poseenc.Wr.weight"], nk0)
    c1, s1 = fourier_encoding(sd["poseenc.Wr.weight"], nk1)
    nb, m, _ = d0.shape
    n = d1.shape[1]
    ind0 = torch.arange(m)
    ind1 = torch.arange(n)
    ni = 0
    sizes = []
    for ni in range(n_layers):
        d0, a00 = self_block(sd, f"self_attn.{ni}", d0, c0, s0, want_colmean=True)
        d1, a11 = self_block(sd, f"self_attn.{ni}", d1, c1, s1, want_colmean=True)
        d0, d1, a01, a10 = cross_block(sd, f"cross_attn.{ni}", d0, d1, want_colmean=True)
        conf0 = pooling_layer(sd, f"pooling.{ni}", d0, torch.stack([a00, a01], -1)).squeeze(-1)
        conf1 = pooling_layer(sd, f"pooling.{ni}", d1, torch.stack([a11, a10], -1)).squeeze(-1)
        if probes is not None:
            probes[f"conf0_{ni}"] = conf0
            probes[f"conf1_{ni}"] = conf1
        if ni >= 1:
            thr = adagml_confidence_threshold(ni, n_layers)
            if d0.shape[1] >= n_min_tokens:
                k0 = conf0[0] > thr
                ind0, d0, c0, s0 = ind0[k0], d0[:, k0], c0[:, k0], s0[:, k0]
            if d1.shape[1] >= n_min_tokens:
                k1 = conf1[0] > thr
                ind1, d1, c1, s1 = ind1[k1], d1[:, k1], c1[:, k1], s1[:, k1]
            sizes.append((ni, d0.shape[1], d1.shape[1]))
            conf = torch.cat([conf0, conf1], -1)
            pos = 1.0 - (conf < thr).float().sum() / (m + n)
            if pos > 0.95:
                break
    dim = d0.shape[-1]
    md0 = _lin(sd, f"out_proj.{ni}", d0) / dim ** 0.25
    md1 = _lin(sd, f"out_proj.{ni}", d1) / dim ** 0.25
    dist = torch.einsum("bmd,bnd->bmn", md0, md1)
    score = sink_algorithm(dist, sd["bin_score"], sinkhorn_iterations)
    i0, _, ms0, _ = compute_matches(score, p)
    valid = i0[0] > -1
    full_i = torch.full((nb, m), -1, dtype=i0.dtype)
    full_i[:, ind0[valid]] = ind1[i0[0][valid]]
    full_s = torch.zeros(nb, m)
    full_s[:, ind0] = ms0
    if probes is not None:
        probes.update(stop_layer=ni, ind0=ind0, ind1=ind1, sizes=sizes)
    return {"matches0": full_i, "matching_scores0": full_s}


def adagml_run(sd: SD, data: dict, p: float = 0.2, n_layers: int = 9, n_min_tokens: int = 256, sinkhorn_iterations: int = 20) -> dict:
    """AdaGML.run — nets/adagml.py:406-489: pre-normalised keypoints in x[:, :, :2], B = 1, returns matched original ids."""
    d0 = _lin(sd, "input_proj", data["desc1"])
    d1 = _lin(sd, "input_proj", data["desc2"])
    c0, s0 = fourier_encoding(sd["poseenc.Wr.weight"], data["x1"][:, :, :2])
    c1, s1 = fourier_encoding(sd["poseenc.Wr.weight"], data["x2"][:, :, :2])
    m, n = d0.shape[1], d1.shape[1]
    ind0, ind1 = torch.arange(m), torch.arange(n)
    ni = 0
    for ni in range(n_layers):
        d0, a00 = self_block(sd, f"self_attn.{ni}", d0, c0, s0, want_colmean=True)
        d1, a11 = self_block(sd, f"self_attn.{ni}", d1, c1, s1, want_colmean=True)
        d0, d1, a01, a10 = cross_block(sd, f"cross_attn.{ni}", d0, d1, want_colmean=True)
        conf0 = pooling_layer(sd, f"pooling.{ni}", d0, torch.stack([a00, a01], -1)).squeeze(-1)
        conf1 = pooling_layer(sd, f"pooling.{ni}", d1, torch.stack([a11, a10], -1)).squeeze(-1)
        if ni >= 1:
            thr = adagml_confidence_threshold(ni, n_layers)
            if d0.shape[1] >= n_min_tokens:
                k0 = conf0[0] > thr
                ind0, d0, c0, s0 = ind0[k0], d0[:, k0], c0[:, k0], s0[:, k0]
            if d1.shape[1] >= n_min_tokens:
                k1 = conf1[0] > thr
                ind1, d1, c1, s1 = ind1[k1], d1[:, k1], c1[:, k1], s1[:, k1]
            if d0.shape[1] <= 5 or d1.shape[1] <= 5:
                return {"index0": torch.zeros(1).long(), "index1": torch.zeros(1).long()}
            conf = torch.cat([conf0, conf1], -1)
            if 1.0 - (conf < thr).float().sum() / (m + n) > 0.95:
                break
    dim = d0.shape[-1]
    md0 = _lin(sd, f"out_proj.{ni}", d0) / dim ** 0.25
    md1 = _lin(sd, f"out_proj.{ni}", d1) / dim ** 0.25
    score = sink_algorithm(torch.einsum("bmd,bnd->bmn", md0, md1), sd["bin_score"], sinkhorn_iterations)
    i0, _, _, _ = compute_matches(score, p)
    valid = i0[0] > -1
    return {"index0": ind0[valid], "index1": ind1[i0[0][valid]]}


# ------------------------------------------------------------------------------------------
# SFD2 — nets/sfd2.py
# ------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _conv_bn_relu(sd: SD, p: str, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    """``conv()`` helper with use_bn=True — nets/sfd2.py:78-91 (conv has bias AND BN)."""
    y = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=stride, padding=1)
    return F.relu(_bn(sd, p + ".1", y))


def _resblock(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResBlock.forward — nets/sfd2.py:107-124 (1x1, 3x3 groups=32, 1x1, all bias-free + BN)."""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    y = F.relu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], padding=1, groups=32)))
    y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
    return F.relu(y + x)


def sfd2_backbone(sd: SD, image: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Conv stack of ResNet4x.extract_local_global — nets/sfd2.py:281-293,331-333."""
    o1a = _conv_bn_relu(sd, "conv1a", image)
    o1b = _conv_bn_relu(sd, "conv1b", o1a, 2)
    o2a = _conv_bn_relu(sd, "conv2a", o1b)
    o2b = _conv_bn_relu(sd, "conv2b", o2a, 2)
    o3a = _conv_bn_relu(sd, "conv3a", o2b)
    o3b = _conv_bn_relu(sd, "conv3b", o3a)
    o4 = o3b
    for i in range(3):
        o4 = _resblock(sd, f"conv4.{i}", o4)
    pa = F.conv2d(o4, sd["convPa.0.weight"], sd["convPa.0.bias"], stride=2, padding=1)
    pa = F.relu(_bn(sd, "convPa.1", pa))
    pa = F.conv2d(pa, sd["convPa.3.weight"], sd["convPa.3.bias"], padding=1)
    logits = F.conv2d(pa, sd["convPb.weight"], sd["convPb.bias"])
    da = F.conv2d(o4, sd["convDa.0.weight"], sd["convDa.0.bias"], padding=1)
    da = F.relu(_bn(sd, "convDa.1", da))
    da = F.conv2d(da, sd["convDa.3.weight"], sd["convDa.3.bias"], padding=1)
    desc = F.conv2d(da, sd["convDb.weight"], sd["convDb.bias"])
    return dict(out1b=o1b, out2b=o2b, out3b=o3b, out4=o4, logits=logits, desc_raw=desc)


def score_map_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """softmax over 65 ch, drop dustbin, 8x8 depth-to-space — nets/sfd2.py:294-300.
    score[b, 8*hc+i, 8*wc+j] = semi[b, 8*i+j, hc, wc]."""
    semi = torch.softmax(logits, 1)[:, :-1]
    b, _, hc, wc = semi.shape
    s = semi.permute(0, 2, 3, 1).reshape(b, hc, wc, 8, 8).permute(0, 1, 3, 2, 4)
    return s.reshape(b, hc * 8, wc * 8)


def simple_nms(scores: torch.Tensor, radius: int) -> torch.Tensor:
    """nets/sfd2.py:20-35: 1 + 2 suppression rounds, window 2r+1, exact fp32 equality."""
    def mp(x):
        return F.max_pool2d(x, kernel_size=2 * radius + 1, stride=1, padding=radius)

    zeros = torch.zeros_like(scores)
    max_mask = scores == mp(scores)
    for _ in range(2):
        supp = mp(max_mask.float()) > 0
        ss = torch.where(supp, zeros, scores)
        new_max = ss == mp(ss)
        max_mask = max_mask | (new_max & (~supp))
    return torch.where(max_mask, scores, zeros)


def select_keypoints(nms_scores: torch.Tensor, conf_th: float, min_keypoints: int, border: int,
                     max_keypoints: int, per_image_fallback: bool = False):
    """Threshold / min-keypoint fallback / border / top-k — nets/sfd2.py:306-329.

    Canonical tie rule: torch.topk's order among equal scores is unspecified,
    so the selected set is ordered by (score desc, flat index y*W+x asc).  When fewer than
    ``max_keypoints`` candidates survive, the row-major ``nonzero`` order is kept (sfd2.py:46-48).
    The fallback test looks at batch element 0 only (sfd2.py:311) unless ``per_image_fallback``.
    Returns lists of keypoints [(N_i,2) float (x,y)] and scores [(N_i,)]."""
    b, h, w = nms_scores.shape
    kps, scs = [], []
    n0 = int((nms_scores[0] >= conf_th).sum())
    for i in range(b):
        ni = int((nms_scores[i] >= conf_th).sum()) if per_image_fallback else n0
        th = conf_th * 0.5 if ni <= min_keypoints else conf_th
        yx = torch.nonzero(nms_scores[i] >= th)
        sc = nms_scores[i][yx[:, 0], yx[:, 1]]
        keep = (yx[:, 0] >= border) & (yx[:, 0] < h - border) & (yx[:, 1] >= border) & (yx[:, 1] < w - border)
        yx, sc = yx[keep], sc[keep]
        if 0 <= max_keypoints < len(yx):
            order = torch.sort(sc, descending=True, stable=True).indices[:max_keypoints]
            yx, sc = yx[order], sc[order]
        kps.append(torch.flip(yx, [1]).float())
        scs.append(sc)
    return kps, scs


def sample_descriptors(keypoints: torch.Tensor, fmap: torch.Tensor, s: int = 4, norm: bool = True) -> torch.Tensor:
    """nets/sfd2.py:53-64 / ResNet4x.sample :348-363.  keypoints [N,2] (x,y); fmap [1,C,h,w].
    u = ((k - s/2 + .5) / (w*s - s/2 - .5, h*s - s/2 - .5))*2 - 1; bilinear, align_corners, zero pad."""
    _, c, h, w = fmap.shape
    k = keypoints - s / 2 + 0.5
    k = k / torch.tensor([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5], dtype=k.dtype)
    k = k * 2 - 1
    d = F.grid_sample(fmap, k.view(1, 1, -1, 2), mode="bilinear", align_corners=True).reshape(1, c, -1)
    if norm:
        d = F.normalize(d, p=2, dim=1)
    return d[0]


def sfd2_extract_local_global(sd: SD, image: torch.Tensor, conf_th: float = 0.005, remove_borders: int = 4,
                              min_keypoints: int = 128, max_keypoints: int = 4096,
                              per_image_fallback: bool = False) -> dict:
    """ResNet4x.extract_local_global — nets/sfd2.py:269-346."""
    bb = sfd2_backbone(sd, image)
    score = score_map_from_logits(bb["logits"])
    ih, iw = image.shape[2:]
    if score.shape[1] != ih or score.shape[2] != iw:
        score = F.interpolate(score[:, None], size=[ih, iw], align_corners=True, mode="bilinear")[:, 0]
    nms = simple_nms(score, 4)
    kps, scs = select_keypoints(nms, conf_th, min_keypoints, remove_borders, max_keypoints, per_image_fallback)
    desc_map = F.normalize(bb["desc_raw"], dim=1)
    descs = [sample_descriptors(k, d[None], 4) for k, d in zip(kps, desc_map)]
    return dict(score_map=score, desc_map=desc_map, mid_features=bb["out4"],
                global_descriptors=[bb["out1b"], bb["out2b"], bb["out3b"], bb["out4"]],
                keypoints=kps, scores=scs, descriptors=descs, nms=nms, logits=bb["logits"])


def sfd2_sample(score_map: torch.Tensor, semi_descs: torch.Tensor, kpts: torch.Tensor, s: int = 4,
                norm_desc: bool = True):
    """ResNet4x.sample — nets/sfd2.py:348-369; scores looked up at integer pixel of batch elem 0."""
    d = sample_descriptors(kpts, semi_descs, s, norm_desc)
    sc = score_map[0, kpts[:, 1].long(), kpts[:, 0].long()]
    return sc, d


# ------------------------------------------------------------------------------------------
# Edges of the path — recogniser epilogue, NN matcher, projection matching,
# offline extraction.  Same status: test infrastructure, pinned by oracle/gen_golden.py.
# ------------------------------------------------------------------------------------------
def _top2(x: torch.Tensor, largest: bool):
    """top-2 per row with the canonical tie rule (lowest index first); torch.topk's order among ties is unspecified."""
    order = torch.sort(x, dim=-1, descending=largest, stable=True)
    return order.values[..., :2], order.indices[..., :2]


def nn_find_nn(sim: torch.Tensor, ratio_thresh, distance_thresh):
    """localization/matchers/nearest_neighbor.py:5-17"""
    sim_nn, ind_nn = _top2(sim, True)
    if not ratio_thresh:
        sim_nn, ind_nn = sim_nn[..., :1], ind_nn[..., :1]
    dist_nn = 2 * (1 - sim_nn)
    mask = torch.ones(ind_nn.shape[:-1], dtype=torch.bool)
    if ratio_thresh:
        mask = mask & (dist_nn[..., 0] <= (ratio_thresh ** 2) * dist_nn[..., 1])
    if distance_thresh:
        mask = mask & (dist_nn[..., 0] <= distance_thresh ** 2)
    matches = torch.where(mask, ind_nn[..., 0], ind_nn.new_tensor(-1))
    scores = torch.where(mask, (sim_nn[..., 0] + 1) / 2, sim_nn.new_tensor(0))
    return matches, scores


def nn_forward(desc0: torch.Tensor, desc1: torch.Tensor, ratio_threshold=None, distance_threshold=None,
               do_mutual_check=True) -> dict:
    """NearestNeighbor._forward — nearest_neighbor.py:39-56; descriptors [B, D, N]."""
    sim = torch.einsum('bdn,bdm->bnm', desc0, desc1)
    m0, s0 = nn_find_nn(sim, ratio_threshold, distance_threshold)
    if do_mutual_check:
        m1, _ = nn_find_nn(sim.transpose(1, 2), ratio_threshold, distance_threshold)
        inds0 = torch.arange(m0.shape[-1])
        loop = torch.gather(m1, -1, torch.where(m0 > -1, m0, m0.new_tensor(0)))
        m0 = torch.where((m0 > -1) & (inds0 == loop), m0, m0.new_tensor(-1))
    return {'matches0': m0, 'matching_scores0': s0}


def add_segmentations(segmentations: torch.Tensor, filtering_threshold: float) -> dict:
    """Frame.add_segmentations — localization/frame.py:96-121 (the keypoint/descriptor filtering is returned as 'keep')."""
    seg_scores = torch.softmax(segmentations, dim=-1)
    keep = None
    if filtering_threshold > 0:
        non_bg = seg_scores[:, 0] < filtering_threshold
        if torch.sum(non_bg) >= 0.4 * seg_scores.shape[0]:
            keep = non_bg
            segmentations, seg_scores = segmentations[non_bg], seg_scores[non_bg]
    return {'keep': keep, 'segmentations': segmentations, 'seg_scores': seg_scores,
            'seg_ids': segmentations.max(dim=-1)[1] - 1}


def process_segmentations(segs: torch.Tensor, topk: int = 10):
    """MultiMap3D.process_segmentations — localization/multimap3d.py:348-379; full sort in canonical tie order."""
    order = torch.sort(segs, dim=-1, descending=True, stable=True)
    pred_values, pred_ids = order.values.numpy(), order.indices.numpy()
    out, used = [], []
    for k in range(segs.shape[-1]):
        values_k, ids_k = pred_values[:, k], pred_ids[:, k]
        out_k = []
        for sid in np.unique(ids_k):
            if sid == 0 or sid in used:
                continue
            used.append(sid)
            ids = np.where(ids_k == sid)[0]
            out_k.append((ids.shape[0], sid, ids, np.mean(values_k[ids])))
        for v in sorted(out_k, key=lambda item: item[0], reverse=True):
            out.append((v[1], v[2], v[3]))
            if len(out) >= topk:
                return out
    return out


def match_by_projection(q_kpts, q_descs, proj_uvs, ref_descs, threshold: float):
    """Descriptor matching inside SingleMap3D.refine_pose_by_projection — localization/singlemap3d.py:416-433."""
    proj_error = q_kpts[:, :2][..., None] - proj_uvs[:2][None]
    proj_error = torch.sqrt(torch.sum(proj_error ** 2, dim=1))
    out_of_range = proj_error >= 2 * threshold
    desc_dist = torch.sqrt(2 - 2 * q_descs @ ref_descs.t() + 1e-6)
    desc_dist[out_of_range] = desc_dist[out_of_range] + 100
    dists, ids = _top2(desc_dist, False)
    ratios = dists[:, 0] / dists[:, 1]
    return (ratios <= 0.995) & (dists[:, 0] < 100), ids[:, 0], dists


def project_map_points(xyzs, K, Tcw, imw: int, imh: int):
    """SingleMap3D.refine_pose_by_projection, projection + frustum test — localization/singlemap3d.py:405-415, float64 like
    the reference.  -> (proj_uvs [3, n_keep] of the survivors, mask [N])."""
    X = torch.as_tensor(np.asarray(xyzs), dtype=torch.float64)
    homo = torch.cat([X, torch.ones(X.shape[0], 1, dtype=X.dtype)], dim=1)
    proj = torch.as_tensor(np.asarray(K), dtype=torch.float64) @ (torch.as_tensor(np.asarray(Tcw), dtype=torch.float64) @ homo.t())[:3, :]
    proj[0] /= proj[2]
    proj[1] /= proj[2]
    mask = (proj[2] > 0) * (proj[2] < 100) * (proj[0] >= 0) * (proj[0] < imw) * (proj[1] >= 0) * (proj[1] < imh)
    return proj[:, mask], mask


def refine_matches_by_projection(q_kpts, q_descs, xyzs, descs, K, Tcw, imw: int, imh: int, threshold: float) -> dict:
    """The device part of SingleMap3D.refine_pose_by_projection — localization/singlemap3d.py:405-444: what it hands to the pose
    solver (matched keypoint ids, the map points they matched)."""
    proj_uvs, mask = project_map_points(xyzs, K, Tcw, imw, imh)
    kp = torch.as_tensor(np.asarray(q_kpts))[:, :2]
    proj_error = torch.sqrt(torch.sum((kp[..., None] - proj_uvs[:2][None]) ** 2, dim=1))
    out_of_range = proj_error >= 2 * threshold
    qd = torch.as_tensor(np.asarray(q_descs)).float()
    rd = torch.as_tensor(np.asarray(descs)).float()[mask]
    desc_dist = torch.sqrt(2 - 2 * qd @ rd.t() + 1e-6)
    desc_dist[out_of_range] = desc_dist[out_of_range] + 100
    dists, ids = _top2(desc_dist, False)
    ratio_mask = ((dists[:, 0] / dists[:, 1]) <= 0.995) & (dists[:, 0] < 100)
    keep = torch.nonzero(mask).flatten()
    return {"point_mask": mask, "ratio_mask": ratio_mask, "matched_keypoint_ids": torch.nonzero(ratio_mask).flatten(),
            "matched_point_ids": keep[ids[:, 0][ratio_mask]], "dists": dists}


def mask_labelling(keypoints, scores, descriptors, mask, topK=-1) -> dict:
    """extract_sfd2_return's mask branch — nets/sfd2.py:508-571, kept as the reference's per-keypoint loop.  Ties in the
    two score sorts are broken by (score desc, index asc) instead of numpy's unspecified quicksort order."""
    labels, others = [], []
    kp_l, sc_l, de_l, kp_u, sc_u, de_u = [], [], [], [], [], []
    id_img = np.int32(mask[:, :, 2]) * 256 * 256 + np.int32(mask[:, :, 1]) * 256 + np.int32(mask[:, :, 0])
    for i in range(keypoints.shape[0]):
        x, y = keypoints[i, 0], keypoints[i, 1]
        gid = id_img[int(y), int(x)]
        if gid == 0:
            kp_u.append(keypoints[i]); sc_u.append(scores[i]); de_u.append(descriptors[i]); others.append(0)
        else:
            kp_l.append(keypoints[i]); sc_l.append(scores[i]); de_l.append(descriptors[i]); labels.append(gid)
    if topK > 0:
        if topK <= len(kp_l):
            idxes = np.argsort(-np.array(sc_l, float), kind="stable")[:topK]
            keypoints = np.array(kp_l, float)[idxes]
            scores = np.array(sc_l, float)[idxes]
            labels = np.array(labels, np.int32)[idxes]
            descriptors = np.array(de_l, float)[idxes]
        elif topK >= len(kp_l) + len(kp_u):
            keypoints, scores, descriptors = kp_l, sc_l, de_l
            for i in range(len(others)):
                keypoints.append(kp_u[i]); scores.append(sc_u[i]); descriptors.append(de_u[i]); labels.append(others[i])
        else:
            n = topK - len(kp_l)
            idxes = np.argsort(-np.array(sc_u, float), kind="stable")[:n]
            keypoints, scores, descriptors = kp_l, sc_l, de_l
            for i in idxes:
                keypoints.append(kp_u[i]); scores.append(sc_u[i]); descriptors.append(de_u[i]); labels.append(others[i])
    return {"keypoints": np.array(keypoints, float), "descriptors": np.array(descriptors, float),
            "scores": np.array(scores, float), "labels": np.array(labels, np.int32)}


def extract_sfd2_return(sd: SD, img: torch.Tensor, conf_th=0.001, topK=-1, scales=(1.0,), mask=None):
    """extract_sfd2_return — nets/sfd2.py:386-589.  img [1,3,H,W] in [0,1], un-normalised."""
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    img = ((img.squeeze() - mean) / std)[None]
    _, _, H, W = img.shape
    all_pts, all_descs = [], []
    for s in scales:
        new_img = img if s == 1.0 else F.interpolate(img, size=(int(H * s), int(W * s)), mode='bilinear', align_corners=True)
        nh, nw = new_img.shape[2:]
        bb = sfd2_backbone(sd, new_img)
        heat = score_map_from_logits(bb["logits"])
        desc_map = F.normalize(bb["desc_raw"], dim=1)
        if heat.shape[1] != nh or heat.shape[2] != nw:
            heat = F.interpolate(heat[:, None], size=[nh, nw], mode='bilinear', align_corners=True)[:, 0]
        sc_map = simple_nms(heat, 3)[0]
        yx = torch.nonzero(sc_map > conf_th)
        sc = sc_map[yx[:, 0], yx[:, 1]]
        order = torch.sort(sc, descending=True, stable=True).indices
        yx, sc = yx[order], sc[order]
        x, y = yx[:, 1], yx[:, 0]
        keep = ~((x < 4) | (x >= W - 4) | (y < 4) | (y >= H - 4))
        x, y, sc = x[keep], y[keep], sc[keep]
        if x.numel() == 0:
            continue
        grid = torch.stack([x.float() / (float(nw) / 2.) - 1., y.float() / (float(nh) / 2.) - 1.], -1).view(1, 1, -1, 2)
        d = F.grid_sample(desc_map, grid, mode='bilinear', align_corners=True).reshape(desc_map.shape[1], -1).numpy()
        d = d / np.linalg.norm(d, axis=0)[np.newaxis, :]
        all_pts.append(torch.stack([x.float() * W / nw, y.float() * H / nh, sc], 1).numpy())
        all_descs.append(d.T)
    if not all_pts:
        return None, None, None
    pts, descs = np.vstack(all_pts), np.vstack(all_descs)
    kp, scs = pts[:, :2], pts[:, 2]
    if mask is not None:
        return mask_labelling(kp, scs, descs, mask, topK)
    if topK > 0:
        idx = np.argsort(-np.array(scs, dtype=float), kind="stable")[:topK]
        kp, scs, descs = kp[idx], scs[idx], descs[idx]
    return {"keypoints": np.array(kp, dtype=float), "descriptors": np.array(descs, dtype=float), "scores": np.array(scs, dtype=float)}


# ------------------------------------------------------------------ edge formats
# Parity status of this block: PINNED.  oracle/gen_golden.py::gen_formats executes the reference's own
# FeaturePairsDataset.__getitem__, writer_fn, find_unique_new_pairs (localization/match_features_batch.py), the export loop of
# localization/extract_features.py::main and colmap_utils/parsers.py with h5py replaced by an in-memory group store (h5py is not
# installed; the container itself is the one thing that cannot run here) and asserts equality with the functions below.
def names_to_pair(name0: str, name1: str, separator: str = "/") -> str:
    """colmap_utils/parsers.py:79-80"""
    return separator.join((name0.replace("/", "-"), name1.replace("/", "-")))


def writer_encode(pred: dict) -> dict:
    """localization/match_features_batch.py:119-129 (writer_fn): batch element 0, indices -> int16, scores -> fp16"""
    out = {"matches0": pred["matches0"][0].cpu().short().numpy()}
    if "matching_scores0" in pred:
        out["matching_scores0"] = pred["matching_scores0"][0].cpu().half().numpy()
    return out


def feature_encode(pred: dict, image_shape, original_size) -> dict:
    """localization/extract_features.py:215-232: descriptors transposed to [D, N]; keypoints mapped to the original
    image; image_size = original (w, h).  image_shape = data['image'].shape (.., h, w)."""
    import numpy as np
    out = dict(pred)
    out["descriptors"] = pred["descriptors"].transpose()
    out["image_size"] = original_size = np.asarray(original_size)
    if "keypoints" in pred:
        size = np.array(tuple(image_shape)[-2:][::-1])
        scales = (original_size / size).astype(np.float32)
        out["keypoints"] = (pred["keypoints"] + .5) * scales[None] - .5
    return out


def pair_item(grp0: dict, grp1: dict) -> dict:
    """localization/match_features_batch.py:95-113 (FeaturePairsDataset.__getitem__) on two feature groups given as
    {name: ndarray}"""
    data = {}
    for grp, sfx in ((grp0, "0"), (grp1, "1")):
        for k, v in grp.items():
            data[k + sfx] = torch.from_numpy(v.__array__()).float()
            if k == "descriptors":
                data[k + sfx] = data[k + sfx].t()
        data["image" + sfx] = torch.empty((1,) + tuple(grp["image_size"])[::-1])
    return data


def parse_retrieval(path) -> dict:
    """colmap_utils/parsers.py:57-63"""
    from collections import defaultdict
    out = defaultdict(list)
    with open(path, "r") as f:
        for line in f.read().rstrip("\n").split("\n"):
            q, r = line.split(" ")
            out[q].append(r)
    return dict(out)


def find_unique_new_pairs(pairs_all, existing=None):
    """localization/match_features_batch.py:165-186: (j, i) dropped when (i, j) came first; pairs already stored under either
    order / either naming dropped.  ``existing`` = container of stored group names (anything supporting ``in``).  The reference
    returns list(set(...)) — arbitrary order — so this returns a set."""
    pairs = set()
    for i, j in pairs_all:
        if (j, i) not in pairs:
            pairs.add((i, j))
    if existing is None:
        return pairs
    old_name = lambda a, b: names_to_pair(a, b, separator="_")
    return {(i, j) for i, j in pairs
            if not (names_to_pair(i, j) in existing or names_to_pair(j, i) in existing or old_name(i, j) in existing or old_name(j, i) in existing)}
