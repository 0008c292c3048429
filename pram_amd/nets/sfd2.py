"""SFD2 keypoint / descriptor extractor ("ResNet4x") on the HIP kernels.

Same state-dict schema, methods and returned dicts as the reference (nets/sfd2.py:127-369,592-596).
Feature maps live in NHWC on the device (what the implicit-GEMM convolutions and the bilinear
gather want); the NCHW-shaped tensors the reference returns (`desc_map`, `mid_features`,
`global_descriptors`) are zero-copy permuted views of that storage (torch channels_last).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from . import _blocks as blk

import os as _os
# conv1a -> conv1b as one kernel on the split-fp16 path (pram_sfd2_conv1_x3_f32); 0: two kernels (conv1a on the exact-fp32 MFMA kernel)
FUSED_CONV1 = _os.environ.get("PRAM_FUSED_CONV1", "1") != "0"

RGB_mean = [0.485, 0.456, 0.406]
RGB_std = [0.229, 0.224, 0.225]


def norm_RGB(img: torch.Tensor) -> torch.Tensor:
    """tvf.Normalize(mean, std) (nets/sfd2.py:14-17) without torchvision."""
    mean = img.new_tensor(RGB_mean).view(-1, 1, 1)
    std = img.new_tensor(RGB_std).view(-1, 1, 1)
    return (img - mean) / std


def _conv_bn(cin, cout, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class _Res(nn.Module):
    """Parameter holder for ResBlock (nets/sfd2.py:94-105): 1x1 / grouped 3x3 / 1x1, bias-free, BN each."""

    def __init__(self, planes=256, groups=32):
        super().__init__()
        self.conv1 = nn.Conv2d(planes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes)


def _head(stride):
    return nn.Sequential(nn.Conv2d(256, 256, 3, stride, 1), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                         nn.Conv2d(256, 256, 3, 1, 1))


class ResNet4x(blk.PackedCache, nn.Module):
    default_config = {'conf_th': 0.005, 'remove_borders': 4, 'min_keypoints': 128, 'max_keypoints': 4096}

    def __init__(self, inputdim=3, outdim=128, desc_compressor=None):
        super().__init__()
        assert inputdim == 3 and outdim == 128 and desc_compressor is None
        self.outdim = outdim
        self.conv1a, self.conv1b = _conv_bn(3, 64), _conv_bn(64, 64, 2)
        self.conv2a, self.conv2b = _conv_bn(64, 128), _conv_bn(128, 128, 2)
        self.conv3a, self.conv3b = _conv_bn(128, 256), _conv_bn(256, 256)
        self.conv4 = nn.Sequential(_Res(), _Res(), _Res())
        self.convPa, self.convDa = _head(2), _head(1)
        self.convPb = nn.Conv2d(256, 65, 1)
        self.convDb = nn.Conv2d(256, outdim, 1)

    # ------------------------------------------------------------------ packed weights
    @staticmethod
    def _ohwi(w: torch.Tensor, pad_to: Optional[int] = None) -> torch.Tensor:
        w = w.detach().float().permute(0, 2, 3, 1)
        if pad_to is not None and w.shape[-1] < pad_to:
            w = torch.cat([w, w.new_zeros(*w.shape[:-1], pad_to - w.shape[-1])], -1)
        return w.contiguous()

    @staticmethod
    def _bn(sd, p):
        scale = sd[p + ".weight"].double() / torch.sqrt(sd[p + ".running_var"].double() + 1e-5)
        shift = sd[p + ".bias"].double() - sd[p + ".running_mean"].double() * scale
        return scale.float(), shift.float()

    def _build_packed(self, dev):
        sd = self.state_dict()
        P: Dict[str, torch.Tensor] = {}
        for name in ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b"):
            P[name + ".w"] = self._ohwi(sd[name + ".0.weight"], 4 if name == "conv1a" else None)
            P[name + ".b"] = sd[name + ".0.bias"].float()
            P[name + ".s"], P[name + ".t"] = self._bn(sd, name + ".1")
        for i in range(3):
            p = f"conv4.{i}"
            for j in (1, 2, 3):
                P[f"{p}.w{j}"] = self._ohwi(sd[f"{p}.conv{j}.weight"])
                P[f"{p}.s{j}"], P[f"{p}.t{j}"] = self._bn(sd, f"{p}.bn{j}")
        for head in ("convPa", "convDa"):
            P[head + ".w0"] = self._ohwi(sd[head + ".0.weight"])
            P[head + ".b0"] = sd[head + ".0.bias"].float()
            P[head + ".s0"], P[head + ".t0"] = self._bn(sd, head + ".1")
            P[head + ".w3"] = self._ohwi(sd[head + ".3.weight"])
            P[head + ".b3"] = sd[head + ".3.bias"].float()
        for head in ("convPb", "convDb"):
            P[head + ".w"] = self._ohwi(sd[head + ".weight"])
            P[head + ".b"] = sd[head + ".bias"].float()
        return {k: v.contiguous().to(dev) for k, v in P.items()}

    # ------------------------------------------------------------------ conv stack (NHWC in/out)
    def _backbone(self, image: torch.Tensor):
        blk.require_cuda(image, "ResNet4x")
        P = self._packed_get(self._build_packed)
        fused1 = FUSED_CONV1 and ops.gemm_prec() in ("x3", "f16")
        x = image.float().contiguous() if fused1 and image.shape[-1] != 4 else ops.image_to_nhwc4(image.float())

        def cbr(x, n, stride=1):
            return ops.conv2d_nhwc(x, P[n + ".w"], P[n + ".b"], P[n + ".s"], P[n + ".t"], ks=3, stride=stride, relu=True)

        if fused1:
            # conv1a -> conv1b in one launch: the 480 x 640 x 64 map between them never goes to HBM (split-fp16 arithmetic; the fp16
            # path takes it too: faster than its own two kernels and more accurate)
            o1b = ops.sfd2_conv1(x, P["conv1a.w"], P["conv1a.b"], P["conv1a.s"], P["conv1a.t"],
                                 P["conv1b.w"], P["conv1b.b"], P["conv1b.s"], P["conv1b.t"])
        else:
            o1b = cbr(cbr(x, "conv1a"), "conv1b", 2)
        o2b = cbr(cbr(o1b, "conv2a"), "conv2b", 2)
        o3b = cbr(cbr(o2b, "conv3a"), "conv3b")
        o4 = o3b
        for i in range(3):
            p = f"conv4.{i}"
            if ops.GROUPED_X3 and ops.gemm_prec() == "x3" and o4.shape[-1] % 64 == 0 and o4.numel() < (1 << 29):      # (32-bit buffer offsets)
                # the 1x1 hands its result over as split fp16 planes; the grouped 3x3 runs on the matrix pipe from them
                yh, yl = ops.conv2d_nhwc_planes(o4, P[p + ".w1"], None, P[p + ".s1"], P[p + ".t1"], ks=1, relu=True)
                y = ops.conv3x3_grouped_planes(yh, yl, P[p + ".w2"], P[p + ".s2"], P[p + ".t2"], groups=o4.shape[-1] // 8, relu=True)
            else:
                y = ops.conv2d_nhwc(o4, P[p + ".w1"], None, P[p + ".s1"], P[p + ".t1"], ks=1, relu=True)
                y = ops.conv3x3_grouped_nhwc(y, P[p + ".w2"], P[p + ".s2"], P[p + ".t2"], groups=32, relu=True)
            o4 = ops.conv2d_nhwc(y, P[p + ".w3"], None, P[p + ".s3"], P[p + ".t3"], residual=o4, ks=1, relu=True)
        return P, o1b, o2b, o3b, o4

    def _score_head(self, P, o4):
        pa = ops.conv2d_nhwc(o4, P["convPa.w0"], P["convPa.b0"], P["convPa.s0"], P["convPa.t0"], ks=3, stride=2, relu=True)
        pa = ops.conv2d_nhwc(pa, P["convPa.w3"], P["convPa.b3"], ks=3)
        logits = ops.conv2d_nhwc(pa, P["convPb.w"], P["convPb.b"], ks=1)
        return logits, ops.score_map(logits)

    def _desc_head(self, P, o4):
        da = ops.conv2d_nhwc(o4, P["convDa.w0"], P["convDa.b0"], P["convDa.s0"], P["convDa.t0"], ks=3, relu=True)
        da = ops.conv2d_nhwc(da, P["convDa.w3"], P["convDa.b3"], ks=3)
        return ops.conv2d_nhwc(da, P["convDb.w"], P["convDb.b"], ks=1, l2norm=True)           # convDb, then F.normalize(desc, dim=1)

    @staticmethod
    def _nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
        return x_nhwc.permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    @blk.with_model_precision
    def det(self, x):
        """-> (score [B,H,W], desc [B,128,H/4,W/4]) — nets/sfd2.py:172-199"""
        P, _, _, _, o4 = self._backbone(x)
        _, score = self._score_head(P, o4)
        return score, self._nchw_view(self._desc_head(P, o4))

    @torch.no_grad()
    @blk.with_model_precision
    def forward(self, batch):
        """nets/sfd2.py:201-233"""
        P, _, _, _, o4 = self._backbone(batch['image'])
        logits, score = self._score_head(P, o4)
        desc = self._nchw_view(self._desc_head(P, o4))
        lg = self._nchw_view(logits)
        return {'dense_features': desc, 'scores': score, 'logits': lg, 'semi_map': torch.softmax(lg, 1)[:, :-1]}

    def extract_patches(self, batch):
        """nets/sfd2.py:235-267 — the same dense outputs as forward()"""
        return self.forward(batch)

    @torch.no_grad()
    @blk.with_model_precision
    def extract_batched(self, image: torch.Tensor, config: dict, per_image_fallback: bool = True):
        """Device-resident, sync-free form of extract_local_global for batches of independent queries:
        padded keypoints [B,k,2], scores [B,k], descriptors [B,k,128], counts int32 [B] (device)."""
        cfg = {**self.default_config, **config}
        b, _, ih, iw = image.shape
        P, o1b, o2b, o3b, o4 = self._backbone(image)
        _, score = self._score_head(P, o4)
        if score.shape[1] != ih or score.shape[2] != iw:      # nets/sfd2.py:301-303
            score = ops.resize_bilinear(score, ih, iw)
        nms = ops.simple_nms(score, 4)
        if cfg['max_keypoints'] < 0:
            # keep every candidate in nonzero() (row-major) order (nets/sfd2.py:324 skips top_k_keypoints): select with a
            # bound that cannot be exceeded, then trim the padded buffers to the longest set — one host read of the
            # counts, which the reference pays in nonzero() as well
            kpts, scores, counts = ops.select_keypoints(nms, cfg['conf_th'], cfg['min_keypoints'], cfg['remove_borders'],
                                                        ih * iw, -1 if per_image_fallback else 0)
            k_eff = max(int(counts.max().item()), 1)
            kpts, scores = kpts[:, :k_eff].contiguous(), scores[:, :k_eff].contiguous()
        else:
            kpts, scores, counts = ops.select_keypoints(nms, cfg['conf_th'], cfg['min_keypoints'], cfg['remove_borders'],
                                                        cfg['max_keypoints'], -1 if per_image_fallback else 0)
        desc_map = self._desc_head(P, o4)
        descs = ops.sample_nhwc(desc_map, kpts, counts, 4, True)
        return dict(score_map=score, desc_map=desc_map, mid_features=o4, global_nhwc=[o1b, o2b, o3b, o4],
                    keypoints=kpts, scores=scores, descriptors=descs, counts=counts)

    @torch.no_grad()
    def extract_local_global(self, data, config={'conf_th': 0.005, 'remove_borders': 4, 'min_keypoints': 128,
                                                 'max_keypoints': 4096}):
        """nets/sfd2.py:269-346.  The min-keypoints fallback looks at batch element 0 only, like the
        reference; selected keypoints follow the canonical (score desc, flat index asc) order."""
        r = self.extract_batched(data['image'], config, per_image_fallback=False)
        counts: List[int] = r['counts'].tolist()      # the one host sync (the reference syncs in nonzero())
        return {
            'score_map': r['score_map'],
            'desc_map': self._nchw_view(r['desc_map']),
            'mid_features': self._nchw_view(r['mid_features']),
            'global_descriptors': [self._nchw_view(t) for t in r['global_nhwc']],
            'keypoints': [r['keypoints'][i, :c] for i, c in enumerate(counts)],
            'scores': [r['scores'][i, :c] for i, c in enumerate(counts)],
            'descriptors': [r['descriptors'][i, :c].t() for i, c in enumerate(counts)],
        }

    @torch.no_grad()
    def sample(self, score_map, semi_descs, kpts, s=4, norm_desc=True):
        """nets/sfd2.py:348-369: (scores [N], descriptors [C,N]) at kpts [N,2] of a [1,C,h,w] map."""
        blk.require_cuda(kpts, "ResNet4x.sample")
        b, c, h, w = semi_descs.shape
        assert b == 1, "ResNet4x.sample is a B = 1 API in the reference (score_map[0, ...])"
        nhwc = semi_descs.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        k = kpts.float().reshape(1, -1, 2)
        d = ops.sample_nhwc(nhwc.float(), k, None, s, bool(norm_desc))
        sc = ops.score_lookup(score_map[:1].float(), k, None)
        return sc[0], d[0].t()

    @torch.no_grad()
    def sample_batched(self, score_map, mid_nhwc, kpts, counts, s=4, norm_desc=True):
        """Batched sample(): [B,k,C] token-major descriptors + [B,k] scores, no host sync."""
        return ops.score_lookup(score_map, kpts, counts), ops.sample_nhwc(mid_nhwc, kpts, counts, s, bool(norm_desc))


class DescriptorCompressor(blk.PackedCache, nn.Module):
    """nets/sfd2.py:372-383: Conv1d(inputdim -> outdim, kernel 1) over descriptors [B, C, N], then L2 normalisation
    along C — a per-keypoint linear map (used by the training CLI's optional descriptor compression, main.py:51-63)."""

    def __init__(self, inputdim: int, outdim: int):
        super().__init__()
        self.inputdim, self.outdim = inputdim, outdim
        self.conv = nn.Conv1d(in_channels=inputdim, out_channels=outdim, kernel_size=1, padding=0, bias=True)

    @torch.no_grad()
    @blk.with_model_precision
    def forward(self, x):
        blk.require_cuda(x, "DescriptorCompressor.forward")
        P = self._packed_get(lambda dev: {"w": self.conv.weight.detach().float().reshape(self.outdim, self.inputdim).contiguous().to(dev),
                                          "b": self.conv.bias.detach().float().contiguous().to(dev)})
        b, c, n = x.shape
        rows = x.float().transpose(1, 2).reshape(b * n, c)
        y = ops.linear(rows, P["w"], P["b"])
        ops.l2norm_rows_(y)
        return y.view(b, n, self.outdim).transpose(1, 2)


def load_sfd2(weight_path):
    """nets/sfd2.py:592-596"""
    net = ResNet4x(inputdim=3, outdim=128)
    net.load_state_dict(torch.load(weight_path, map_location='cpu')['state_dict'], strict=True)
    return net


@torch.no_grad()
def label_keypoints_by_mask(keypoints, scores, descriptors, mask, topK=-1):
    """The mask branch of extract_sfd2_return (nets/sfd2.py:508-571): a keypoint's label is the 24-bit id the BGR
    segmentation image holds at (int(y), int(x)); labelled keypoints come first, the budget ``topK`` is filled with
    the best labelled ones, then with the best unlabelled ones (label 0).  Host-side numpy, like the reference.
    Reproduced as they are: with topK <= 0 all keypoints come back in their original order while ``labels`` lists
    only the labelled ones; score ties follow (score desc, original index asc) (numpy's default argsort leaves them
    unspecified); the reference's ``np.float`` (gone from numpy >= 1.24) is float64."""
    import numpy as np
    mask = np.asarray(mask)
    id_img = np.int32(mask[:, :, 2]) * 256 * 256 + np.int32(mask[:, :, 1]) * 256 + np.int32(mask[:, :, 0])
    keypoints, scores, descriptors = np.asarray(keypoints), np.asarray(scores), np.asarray(descriptors)
    gid = id_img[keypoints[:, 1].astype(np.int64), keypoints[:, 0].astype(np.int64)]      # int(): truncation, coordinates >= 0
    lab, unl = np.flatnonzero(gid != 0), np.flatnonzero(gid == 0)
    labels = gid[lab].astype(np.int32)
    best = lambda idx, n: idx[np.argsort(-scores[idx].astype(float), kind="stable")[:n]]
    if topK > 0:
        if topK <= lab.size:
            sel = best(lab, topK)
            labels = gid[sel].astype(np.int32)
        elif topK >= lab.size + unl.size:
            sel = np.concatenate([lab, unl])
            labels = np.concatenate([labels, np.zeros(unl.size, np.int32)])
        else:
            extra = best(unl, topK - lab.size)
            sel = np.concatenate([lab, extra])
            labels = np.concatenate([labels, np.zeros(extra.size, np.int32)])
        keypoints, scores, descriptors = keypoints[sel], scores[sel], descriptors[sel]
    return {"keypoints": np.array(keypoints, float), "descriptors": np.array(descriptors, float),
            "scores": np.array(scores, float), "labels": np.array(labels, np.int32)}


def extract_sfd2_return(model, img, conf_th=0.001, mask=None, topK=-1, min_keypoints=0, **kwargs):
    """Offline extraction variant (nets/sfd2.py:386-589): det() per scale, NMS radius 3, ``>`` threshold,
    sort by score, border 4, descriptors at the keypoints, float64 numpy outputs.  The dense work (convs,
    NMS, resize, descriptor sampling) runs in HIP; the per-keypoint bookkeeping is the reference's numpy.
    Ties in the score sort follow (score desc, row-major index asc) — numpy's argsort order among equal
    scores is unspecified in the reference."""
    import numpy as np
    dev = next(model.parameters()).device
    img = norm_RGB(img.squeeze())[None].to(dev).float()
    B, one, H, W = img.shape
    all_pts, all_descs = [], []
    for s in kwargs.get('scales', [1.0]):
        if s == 1.0:
            new_img = img
        else:
            nh, nw = int(H * s), int(W * s)
            new_img = ops.resize_bilinear(img, nh, nw)
        nh, nw = new_img.shape[2:]
        heatmap, coarse_desc = model.det(new_img)
        if heatmap.size(1) != nh or heatmap.size(2) != nw:
            heatmap = ops.resize_bilinear(heatmap, nh, nw)
        scores = ops.simple_nms(heatmap, 3)[0]
        yx = torch.nonzero(scores > conf_th)
        sc = scores[yx[:, 0], yx[:, 1]]
        order = torch.sort(sc, descending=True, stable=True).indices
        yx, sc = yx[order], sc[order]
        x, y = yx[:, 1], yx[:, 0]
        keep = ~((x < 4) | (x >= W - 4) | (y < 4) | (y >= H - 4))     # NB: reference tests against the ORIGINAL W, H
        x, y, sc = x[keep], y[keep], sc[keep]
        if x.numel() == 0:
            continue
        D = coarse_desc.size(1)
        if coarse_desc.size(2) == nh and coarse_desc.size(3) == nw:
            desc = coarse_desc[0, :, y, x]
        else:
            grid = torch.stack([x.float() / (float(nw) / 2.) - 1., y.float() / (float(nh) / 2.) - 1.], -1)[None]
            nhwc = coarse_desc.permute(0, 2, 3, 1)
            nhwc = nhwc if nhwc.is_contiguous() else nhwc.contiguous()
            desc = ops.sample_nhwc(nhwc, grid, None, 0, False)[0].t()
            desc = desc / torch.linalg.norm(desc, dim=0, keepdim=True)
        # rescale on the host in numpy float32 like the reference (sfd2.py:483-484): torch's GPU division by a
        # scalar multiplies by the reciprocal and lands 1 ulp away from the correctly rounded quotient
        pts = torch.stack([x.float(), y.float(), sc], 1).cpu().numpy()
        pts[:, 0] = pts[:, 0] * W / nw
        pts[:, 1] = pts[:, 1] * H / nh
        all_pts.append(pts)
        all_descs.append(desc.t().cpu().numpy())
    if not all_pts:
        return None, None, None
    all_pts, all_descs = np.vstack(all_pts), np.vstack(all_descs)
    keypoints, scores, descriptors = all_pts[:, 0:2], all_pts[:, 2], all_descs
    if mask is not None:
        return label_keypoints_by_mask(keypoints, scores, descriptors, mask, topK)
    if topK > 0:
        idxes = np.argsort(-np.array(scores, dtype=float), kind="stable")[:topK]
        keypoints, scores, descriptors = keypoints[idxes], scores[idxes], descriptors[idxes]
    return {"keypoints": np.array(keypoints, dtype=float), "descriptors": np.array(descriptors, dtype=float),
            "scores": np.array(scores, dtype=float)}
