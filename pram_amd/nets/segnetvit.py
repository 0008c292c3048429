"""SegNetViT — sparse-token landmark recogniser on the HIP kernels.

Same constructor config, state-dict schema and ``forward(data) -> {'prediction': [B,N,C]}`` as the
reference (nets/segnetvit.py:124-203), so ``load_state_dict(torch.load(p)['model'], strict=True)``
(inference.py:33-39) works unchanged.  The nn.Linear / nn.LayerNorm members are parameter
containers only: forward() never calls them, it runs pram_amd/csrc kernels.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from .. import ops
from . import _blocks as blk
from .utils import keypoint_norm_constants


class _FourierEnc(nn.Module):
    """Parameter holder for LearnableFourierPositionalEncoding (segnetvit.py:26-40): Wr 2 -> F/2."""

    def __init__(self, M: int, dim: int, gamma: float = 1.0):
        super().__init__()
        self.Wr = nn.Linear(M, dim // 2, bias=False)
        nn.init.normal_(self.Wr.weight.data, mean=0, std=gamma ** -2)


class _SelfLayer(nn.Module):
    """Parameter holder with the reference's SelfMultiHeadAttention key names (segnetvit.py:79-95)."""

    def __init__(self, feat_dim: int, hidden_dim: int):
        super().__init__()
        self.qkv = nn.Linear(feat_dim, hidden_dim * 3)
        self.proj = nn.Linear(hidden_dim, hidden_dim)
        self.mlp = nn.Sequential(nn.Linear(feat_dim + hidden_dim, feat_dim * 2),
                                 nn.LayerNorm(feat_dim * 2, elementwise_affine=True), nn.GELU(),
                                 nn.Linear(feat_dim * 2, feat_dim))


class _Stack(nn.Module):
    def __init__(self, n_layers: int, dim: int):
        super().__init__()
        self.layers = nn.ModuleList([_SelfLayer(dim, dim) for _ in range(n_layers)])


class SegNetViT(blk.PackedCache, nn.Module):
    default_config = {
        'descriptor_dim': 256, 'output_dim': 1024, 'n_class': 512, 'keypoint_encoder': [32, 64, 128, 256],
        'n_layers': 15, 'num_heads': 4, 'hidden_dim': 256, 'with_score': False, 'with_global': False,
        'with_cls': False, 'with_sc': False,
    }

    def __init__(self, config={}):
        super().__init__()
        self.config = {**self.default_config, **config}
        c = self.config
        if c['hidden_dim'] != 256 or c['num_heads'] != 4:
            raise NotImplementedError("HIP kernels are specialised for hidden_dim 256 / 4 heads of 64")
        self.with_cls, self.with_sc, self.with_score = c['with_cls'], c['with_sc'], c['with_score']
        self.n_layers = c['n_layers']
        self.gnn = _Stack(c['n_layers'], c['hidden_dim'])
        self.kenc = _FourierEnc(2, c['hidden_dim'] // c['num_heads'])
        self.input_proj = nn.Linear(c['descriptor_dim'], c['hidden_dim'])
        self.seg = nn.Sequential(nn.Linear(c['hidden_dim'], c['output_dim']),
                                 nn.LayerNorm(c['output_dim'], elementwise_affine=True), nn.GELU(),
                                 nn.Linear(c['output_dim'], c['n_class']))
        if self.with_sc:       # second head on the same tokens (segnetvit.py:166-172,199-201); no shipped config enables it
            self.sc = nn.Sequential(nn.Linear(c['hidden_dim'], c['output_dim']),
                                    nn.LayerNorm(c['output_dim'], elementwise_affine=True), nn.GELU(),
                                    nn.Linear(c['output_dim'], 3))

    # ---- packed device weights
    def _build_packed(self, dev):
        sd = self.state_dict()
        f = lambda k: sd[k].detach().float().contiguous().to(dev)
        heads = ["seg"] + (["sc"] if self.with_sc else [])
        extra = {f"{h}{i}_{n[0]}": f(f"{h}.{i}.{n}") for h in heads[1:] for i in (0, 1, 3) for n in ("weight", "bias")}
        # the first Linear of a head feeds a LayerNorm: centred over its outputs at pack time (ops.center_linear), so that the
        # LayerNorm + GELU can ride inside the second GEMM on the split-fp16 path (ops.mlp_tail)
        cen = {}
        for h in heads:
            w, b = ops.center_linear(sd[f"{h}.0.weight"], sd[f"{h}.0.bias"])
            cen[f"{h}0_w"], cen[f"{h}0_b"] = w.to(dev), b.to(dev)
        return {
            **extra,
            **cen,
            "Wr": f("kenc.Wr.weight"),
            "in_w": f("input_proj.weight"), "in_b": f("input_proj.bias"),
            "layers": [blk.pack_self_block(sd, f"gnn.layers.{i}", dev) for i in range(self.n_layers)],
            "seg1_w": f("seg.1.weight"), "seg1_b": f("seg.1.bias"), "seg3_w": f("seg.3.weight"), "seg3_b": f("seg.3.bias"),
        }

    def preprocess(self, data):
        """-> (descriptors [B,N,D], (cos, sin) [B,N,32])  — segnetvit.py:174-186"""
        desc0 = data['seg_descriptors']
        blk.require_cuda(desc0, "SegNetViT.forward")
        P = self._packed_get(self._build_packed)
        if 'norm_keypoints' in data.keys():
            enc = ops.fourier_encoding(data['norm_keypoints'], P["Wr"], 0.0, 0.0, 1.0)
        elif 'image' in data.keys():
            cx, cy, sc = keypoint_norm_constants(data['image'].shape)
            enc = ops.fourier_encoding(data['keypoints'], P["Wr"], cx, cy, sc)
        else:
            raise ValueError('Require image shape for keypoint coordinate normalization')
        return desc0, enc

    @torch.no_grad()
    @blk.with_model_precision
    def forward(self, data: Dict[str, torch.Tensor]):
        desc, (cos, sin) = self.preprocess(data)
        P = self._packed_get(self._build_packed)
        B, N, D = desc.shape
        lens: Optional[torch.Tensor] = data.get('lens')      # extension: ragged batches (int32 [B], device)
        x = ops.linear(desc.float().reshape(B * N, D), P["in_w"], P["in_b"])
        cos, sin = cos.reshape(B * N, 32), sin.reshape(B * N, 32)
        for lp in P["layers"]:
            x = blk.self_block(x, lp, cos, sin, B, N, lens)
        kw = dict(lens=lens, t_pad=N)      # ragged like the layers: rows beyond a frame's keypoint count are never read or written
        # the logits of rows beyond a frame's keypoint count read zero (they are the tensor the caller sees), not leftovers
        o0 = None if lens is None else ops._filled((B * N, P["seg3_w"].shape[0]), x.device)
        if ops.gemm_prec() == "x3" and blk.FUSED_MLP and P["seg3_w"].shape[0] > 64:
            out = ops.mlp_tail(x, P["seg0_w"], P["seg0_b"], P["seg1_w"], P["seg1_b"], P["seg3_w"], P["seg3_b"], out=o0, **kw)
        else:
            h = ops.linear(x, P["seg0_w"], P["seg0_b"], **kw)
            ops.layernorm_gelu_(h, P["seg1_w"], P["seg1_b"], **kw)
            out = ops.linear(h, P["seg3_w"], P["seg3_b"], out=o0, **kw)
        output = {'prediction': out.view(B, N, -1)}
        if self.with_sc:      # three outputs: narrower than the fused tail takes
            h = ops.linear(x, P["sc0_w"], P["sc0_b"], **kw)
            ops.layernorm_gelu_(h, P["sc1_w"], P["sc1_b"], **kw)
            s0 = None if lens is None else ops._filled((B * N, 3), x.device)      # zeros beyond a frame's keypoint count, like 'prediction'
            output['sc'] = ops.linear(h, P["sc3_w"], P["sc3_b"], out=s0, **kw).view(B, N, 3)
        return output
