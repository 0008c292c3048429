"""GML matcher (LightGlue-style: 9 x [self, cross] attention + Sinkhorn) on the HIP kernels.

Same config keys, state-dict schema and outputs as the reference (nets/gml.py:189-319).  The free
functions ``sinkhorn`` / ``sink_algorithm`` / ``dual_softmax`` / ``compute_matches`` that
nets/gm.py duplicates (gm.py:17-43,249-264) are exposed here over the same kernels.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from .. import ops
from . import _blocks as blk
from .segnetvit import _FourierEnc, _SelfLayer
from .utils import keypoint_norm_constants

eps = 1e-8


class _CrossLayer(nn.Module):
    """Parameter holder with CrossMultiHeadAttention's key names (nets/gml.py:143-162)."""

    def __init__(self, feat_dim: int, hidden_dim: int):
        super().__init__()
        self.to_qk = nn.Linear(feat_dim, hidden_dim)
        self.to_v = nn.Linear(feat_dim, hidden_dim)
        self.proj = nn.Linear(hidden_dim, hidden_dim)
        self.mlp = nn.Sequential(nn.Linear(feat_dim + hidden_dim, feat_dim * 2),
                                 nn.LayerNorm(feat_dim * 2, elementwise_affine=True), nn.GELU(),
                                 nn.Linear(feat_dim * 2, feat_dim))


# ---- free functions (nets/gml.py:20-46,304-319) over the HIP kernels -------------------------------
def sink_algorithm(M: torch.Tensor, dustbin: torch.Tensor, iteration: int) -> torch.Tensor:
    """[B,m,n] scores -> [B,m+1,n+1] transport plan (plain-domain Sinkhorn with dustbins)."""
    blk.require_cuda(M, "sink_algorithm")
    Mp, n = _pad_cols(M)
    return ops.sinkhorn_match(Mp, dustbin, iteration, 0.0, want_p=True, n_valid=n)["p"]


def dual_softmax(M: torch.Tensor, dustbin: torch.Tensor) -> torch.Tensor:
    blk.require_cuda(M, "dual_softmax")
    Mp, n = _pad_cols(M)
    return ops.sinkhorn_match(Mp, dustbin, 0, 0.0, want_p=True, dual_softmax=True, n_valid=n)["p"]


def _pad_cols(M: torch.Tensor):
    M = M.float().contiguous()
    return M, M.shape[-1]


def normalize_inputs(data: dict):
    """-> per-set (kpts, cx, cy, scale) following nets/gml.py:254-264's key precedence."""
    k0, k1 = data['keypoints0'], data['keypoints1']
    if 'norm_keypoints0' in data.keys() and 'norm_keypoints1' in data.keys():
        return (data['norm_keypoints0'], 0.0, 0.0, 1.0), (data['norm_keypoints1'], 0.0, 0.0, 1.0)
    if 'image0' in data.keys() and 'image1' in data.keys():
        return (k0, *keypoint_norm_constants(data['image0'].shape)), (k1, *keypoint_norm_constants(data['image1'].shape))
    if 'image_shape0' in data.keys() and 'image_shape1' in data.keys():
        return (k0, *keypoint_norm_constants(data['image_shape0'])), (k1, *keypoint_norm_constants(data['image_shape1']))
    raise ValueError('Require image shape for keypoint coordinate normalization')


_const_cache = {}


def _const_i32(device, n: int, value: int) -> torch.Tensor:
    """A cached [n] int32 device tensor filled with ``value`` (the "every set is full" lengths): no kernel in steady state."""
    key = (str(device), n, value)
    t = _const_cache.get(key)
    if t is None:
        t = _const_cache[key] = torch.full((n,), value, device=device, dtype=torch.int32)
    return t


def stack_pair(desc0: torch.Tensor, desc1: torch.Tensor, lens0, lens1):
    """[B,M,D],[B,N,D] -> X [2B, T, D] (T = max(M,N), zero padded) and lens int32 [2B] or None.  Contiguous fp32 inputs of equal
    size are placed with two device-to-device copies (no framework kernel on the path)."""
    B, M, D = desc0.shape
    N = desc1.shape[1]
    T = max(M, N)
    dev = desc0.device
    if M == N:
        X = torch.empty(2 * B, T, D, device=dev, dtype=torch.float32)
        X[:B].copy_(desc0)
        X[B:].copy_(desc1)
    else:
        X = ops._filled((2 * B, T, D), dev)
        X[:B, :M] = desc0
        X[B:, :N] = desc1
    lens = None
    if lens0 is not None or lens1 is not None or M != N:
        lens = torch.empty(2 * B, device=dev, dtype=torch.int32)
        lens[:B].copy_(lens0 if lens0 is not None else _const_i32(dev, B, M))
        lens[B:].copy_(lens1 if lens1 is not None else _const_i32(dev, B, N))
    return X, T, lens


def stack_encodings(k0, k1, enc0, enc1, Wr, T: int):
    """Fourier encodings of both keypoint sets as one [2B * T, 32] (cos, sin) pair, rows beyond a set's size zero: the encoder
    writes straight into its half when the sets are equally large (no copies)."""
    B, M = k0.shape[0], k0.shape[1]
    N = k1.shape[1]
    dev = k0.device
    if M == N == T:
        cos = torch.empty(2 * B, T, 32, device=dev, dtype=torch.float32)
        sin = torch.empty(2 * B, T, 32, device=dev, dtype=torch.float32)
        ops.fourier_encoding(k0.float(), Wr, *enc0, out=(cos[:B], sin[:B]))
        ops.fourier_encoding(k1.float(), Wr, *enc1, out=(cos[B:], sin[B:]))
    else:
        cos, sin = ops._filled((2 * B, T, 32), dev), ops._filled((2 * B, T, 32), dev)
        c0, s0 = ops.fourier_encoding(k0.float(), Wr, *enc0)
        c1, s1 = ops.fourier_encoding(k1.float(), Wr, *enc1)
        cos[:B, :M], sin[:B, :M], cos[B:, :N], sin[B:, :N] = c0, s0, c1, s1
    return cos, sin


class GML(blk.PackedCache, nn.Module):
    default_config = {
        'descriptor_dim': 128, 'hidden_dim': 256, 'weights': 'indoor', 'keypoint_encoder': [32, 64, 128, 256],
        'GNN_layers': ['self', 'cross'] * 9, 'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'with_pose': False,
        'n_layers': 9, 'n_min_tokens': 256, 'with_sinkhorn': True, 'ac_fn': 'relu', 'norm_fn': 'bn',
    }

    def __init__(self, config):
        super().__init__()
        self.config = {**self.default_config, **config}
        c = self.config
        if c['hidden_dim'] != 256:
            raise NotImplementedError("HIP kernels are specialised for hidden_dim 256 / 4 heads of 64")
        self.n_layers = c['n_layers']
        self.with_sinkhorn = c['with_sinkhorn']
        self.match_threshold = c['match_threshold']
        self.sinkhorn_iterations = c['sinkhorn_iterations']
        self.input_proj = nn.Linear(c['descriptor_dim'], c['hidden_dim'])
        self.self_attn = nn.ModuleList([_SelfLayer(c['hidden_dim'], c['hidden_dim']) for _ in range(self.n_layers)])
        self.cross_attn = nn.ModuleList([_CrossLayer(c['hidden_dim'], c['hidden_dim']) for _ in range(self.n_layers)])
        self.poseenc = _FourierEnc(2, c['hidden_dim'] // 4)
        self.out_proj = nn.ModuleList([nn.Linear(c['hidden_dim'], c['hidden_dim']) for _ in range(self.n_layers)])
        self.register_parameter('bin_score', torch.nn.Parameter(torch.tensor(1.)))

    def _build_packed(self, dev):
        sd = self.state_dict()
        f = lambda k: sd[k].detach().float().contiguous().to(dev)
        return {
            "Wr": f("poseenc.Wr.weight"), "in_w": f("input_proj.weight"), "in_b": f("input_proj.bias"),
            "self": [blk.pack_self_block(sd, f"self_attn.{i}", dev) for i in range(self.n_layers)],
            "cross": [blk.pack_cross_block(sd, f"cross_attn.{i}", dev) for i in range(self.n_layers)],
            "out_w": [f(f"out_proj.{i}.weight") for i in range(self.n_layers)],
            "out_b": [f(f"out_proj.{i}.bias") for i in range(self.n_layers)],
            "bin": f("bin_score").reshape(1),
        }

    def forward(self, data, mode=0):
        if not self.training:
            return self.produce_matches(data=data)
        raise NotImplementedError("training is outside the hot path (nets/gml.py:247 is `pass` in the reference too)")

    @torch.no_grad()
    @blk.with_model_precision
    def produce_matches(self, data: dict, p=0.2, **kwargs):
        desc0, desc1 = data['descriptors0'], data['descriptors1']
        blk.require_cuda(desc0, "GML.produce_matches")
        (k0, cx0, cy0, sc0), (k1, cx1, cy1, sc1) = normalize_inputs(data)
        P = self._packed_get(self._build_packed)
        B, M, _ = desc0.shape
        N = desc1.shape[1]
        X, T, lens = stack_pair(desc0, desc1, data.get('lens0'), data.get('lens1'))
        cos, sin = stack_encodings(k0, k1, (cx0, cy0, sc0), (cx1, cy1, sc1), P["Wr"], T)
        cos, sin = cos.view(-1, 32), sin.view(-1, 32)
        x = ops.linear(X.view(2 * B * T, -1), P["in_w"], P["in_b"])
        nI = self.n_layers
        for i in range(nI):
            x = blk.self_block(x, P["self"][i], cos, sin, 2 * B, T, lens)
            x = blk.cross_block(x, P["cross"][i], B, T, lens)
        d = x.shape[-1]
        ldc = (T + 3) // 4 * 4
        if (blk._split_path() or blk._half_path()) and d % 32 == 0:
            # the matching descriptors leave the projection as split planes and meet on the fp16 matrix pipe as well
            # ragged like everything upstream: rows beyond a set's size hold whatever the (ragged) producers left there
            # (fp16 path too: the score matrix feeds a Sinkhorn whose plan is fp32 — the split kernels are both faster than the
            # exact-fp32 ones and more accurate than a single fp16 product)
            _, pl = ops.linear(x, P["out_w"][nI - 1], P["out_b"][nI - 1], alpha=1.0 / d ** .25, split_out="only", lens=lens, t_pad=T,
                               precision="x3")
            dist = ops.bgemm_nt_planes((pl[0][:B * T], pl[1][:B * T]), (pl[0][B * T:], pl[1][B * T:]), B, T, T, ldc=ldc)
        else:
            md = ops.linear(x, P["out_w"][nI - 1], P["out_b"][nI - 1], alpha=1.0 / d ** .25, lens=lens, t_pad=T).view(2 * B, T, d)
            dist = ops.bgemm_nt(md[:B], md[B:], ldc=ldc)
        r = ops.sinkhorn_match(dist, P["bin"], self.sinkhorn_iterations, p,
                               m_lens=None if lens is None else lens[:B], n_lens=None if lens is None else lens[B:],
                               dual_softmax=not self.with_sinkhorn, n_valid=T)
        return {
            'matches0': r['matches0'][:, :M], 'matches1': r['matches1'][:, :N],
            'matching_scores0': r['matching_scores0'][:, :M], 'matching_scores1': r['matching_scores1'][:, :N],
        }

    # API parity helpers (nets/gml.py:296-319)
    def compute_score(self, dist, dustbin, iteration):
        return sink_algorithm(dist, dustbin, iteration) if self.with_sinkhorn else dual_softmax(dist, dustbin)

    def compute_matches(self, scores, p=0.2):
        """nets/gml.py:304-319 on a materialised [B, M+1, N+1] score matrix (API parity; produce_matches itself
        uses the match extraction fused into the last Sinkhorn pass).  Row / column arg-max by the wave-reduction
        top-2 kernel (first occurrence on ties, like torch.max)."""
        blk.require_cuda(scores, "GML.compute_matches")
        inner = scores[:, :-1, :-1].float().contiguous()
        max0, _, indices0 = ops.row_top2(inner, largest=True)
        max1, _, indices1 = ops.row_top2(inner.transpose(1, 2).contiguous(), largest=True)
        ar0 = torch.arange(indices0.shape[1], device=scores.device)[None]
        ar1 = torch.arange(indices1.shape[1], device=scores.device)[None]
        mutual0 = ar0 == indices1.gather(1, indices0)
        mutual1 = ar1 == indices0.gather(1, indices1)
        zero = scores.new_tensor(0)
        mscores0 = torch.where(mutual0, max0, zero)
        mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
        valid0 = mutual0 & (mscores0 > p)
        valid1 = mutual1 & valid0.gather(1, indices1)
        return (torch.where(valid0, indices0, indices0.new_tensor(-1)), torch.where(valid1, indices1, indices1.new_tensor(-1)),
                mscores0, mscores1)
