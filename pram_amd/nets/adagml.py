"""AdaGML — adaptive GML (per-layer confidence pooling, token pruning, early exit) on the HIP kernels.

Same config, state-dict schema and ``produce_matches`` output as the reference
(nets/adagml.py:232-404): ``matches0`` / ``matching_scores0`` scattered back to the full query set.
The reference is B = 1 only (its mask indexing drops the batch dimension) and its Python ``if``
synchronises with the device every layer; here a batch of pairs runs in one device-resident pass with
per-pair token counts / stop flags kept in device tensors (no host read), each pair computing exactly
its B = 1 result.  An emptied token set yields "no matches" instead of the reference's IndexError.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import ops
from . import _blocks as blk
from .gml import GML, normalize_inputs, sink_algorithm, dual_softmax, stack_encodings  # noqa: F401  (API parity)


class _Pooling(nn.Module):
    """Parameter holder for PoolingLayer (nets/adagml.py:114-130)."""

    def __init__(self, hidden_dim: int, score_dim: int = 2):
        super().__init__()
        self.score_enc = nn.Sequential(nn.Linear(score_dim, hidden_dim), nn.LayerNorm(hidden_dim, elementwise_affine=True),
                                       nn.GELU(), nn.Linear(hidden_dim, hidden_dim))
        self.proj = nn.Linear(hidden_dim, hidden_dim)
        self.predict = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.LayerNorm(hidden_dim, elementwise_affine=True),
                                     nn.GELU(), nn.Linear(hidden_dim, 1))


class AdaGML(GML):
    default_config = {
        'descriptor_dim': 128, 'hidden_dim': 256, 'weights': 'indoor', 'keypoint_encoder': [32, 64, 128, 256],
        'GNN_layers': ['self', 'cross'] * 9, 'sinkhorn_iterations': 20, 'match_threshold': 0.2, 'with_pose': True,
        'n_layers': 9, 'n_min_tokens': 256, 'with_sinkhorn': True, 'min_confidence': 0.9,
        'classification_background_weight': 0.05, 'pretrained': True,
    }

    def __init__(self, config):
        super().__init__(config)
        self.n_min_tokens = self.config['n_min_tokens']
        self.min_confidence = self.config['min_confidence']
        self.pooling = nn.ModuleList([_Pooling(self.config['hidden_dim'], 2) for _ in range(self.n_layers)])

    def _build_packed(self, dev):
        P = super()._build_packed(dev)
        sd = self.state_dict()
        f = lambda k: sd[k].detach().float().contiguous().to(dev)
        pools = []
        for i in range(self.n_layers):
            p = f"pooling.{i}"
            w0 = sd[p + ".score_enc.0.weight"].detach().float()
            pools.append({
                "se0_w": torch.cat([w0, w0.new_zeros(w0.shape[0], 2)], 1).contiguous().to(dev),   # K 2 -> 4 (zero pad)
                "se0_b": f(p + ".score_enc.0.bias"), "se1_w": f(p + ".score_enc.1.weight"), "se1_b": f(p + ".score_enc.1.bias"),
                "se3_w": f(p + ".score_enc.3.weight"), "se3_b": f(p + ".score_enc.3.bias"),
                "proj_w": f(p + ".proj.weight"), "proj_b": f(p + ".proj.bias"),
                "pr0_w": f(p + ".predict.0.weight"), "pr0_b": f(p + ".predict.0.bias"),
                "pr1_w": f(p + ".predict.1.weight"), "pr1_b": f(p + ".predict.1.bias"),
                # N 1 -> 4 rows (zero pad) keeps the logit column 16-B aligned; column 0 is the logit
                "pr3_w": torch.cat([sd[p + ".predict.3.weight"].detach().float().cpu(),
                                    torch.zeros(3, sd[p + ".predict.3.weight"].shape[1])], 0).contiguous().to(dev),
                "pr3_b": torch.cat([sd[p + ".predict.3.bias"].detach().float().cpu(), torch.zeros(3)]).contiguous().to(dev),
            })
        P["pool"] = pools
        return P

    def forward(self, data, mode=0):
        if not self.training:
            if mode == 0:
                return self.produce_matches(data=data)
            return self.run(data=data)
        raise NotImplementedError("training is outside the hot path")

    @torch.no_grad()
    def run(self, data, p=0.2):
        """AdaGML.run (nets/adagml.py:406-489): the evaluation-time entry with pre-normalised keypoints packed as
        x = [kx, ky, ..., score]; returns the matched ORIGINAL indices {'index0', 'index1'} of one pair (B = 1), or
        two zero tensors when pruning leaves <= 5 tokens in a set (adagml.py:458-462)."""
        d = {'descriptors0': data['desc1'], 'descriptors1': data['desc2'],
             'keypoints0': data['x1'][:, :, :2], 'keypoints1': data['x2'][:, :, :2],
             'norm_keypoints0': data['x1'][:, :, :2].contiguous(), 'norm_keypoints1': data['x2'][:, :, :2].contiguous(),
             'scores0': data['x1'][:, :, -1], 'scores1': data['x2'][:, :, -1]}
        assert d['descriptors0'].shape[0] == 1, "AdaGML.run is a B = 1 API in the reference"
        internal = {}
        self.produce_matches(d, p=p, _internal=internal)
        dev = d['descriptors0'].device
        if bool(internal['tiny'][0].item()):
            z = torch.zeros(size=(1,), device=dev).long()
            return {'index0': z, 'index1': z.clone()}
        n0 = int(internal['lens'][0].item())
        m0 = internal['matches0'][0, :n0]
        valid = m0 > -1
        ind0, ind1 = internal['ind'][0].long(), internal['ind'][1].long()
        return {'index0': ind0[:n0][valid], 'index1': ind1[m0[valid]]}

    def confidence_threshold(self, layer_index: int):
        """nets/adagml.py:516-520"""
        return float(np.clip(0.5 + 0.1 * np.exp(-4.0 * layer_index / self.n_layers), 0, 1))

    # API parity helpers (nets/adagml.py:491-514,522-538); produce_matches applies the same rules inside its kernels
    def compute_score(self, dist, dustbin, iteration):
        return GML.compute_score(self, dist, dustbin, iteration)

    def compute_matches(self, scores, p=0.2):
        return GML.compute_matches(self, scores, p)

    def check_if_stop(self, confidences0: torch.Tensor, confidences1: torch.Tensor, layer_index: int, num_points: int) -> torch.Tensor:
        confidences = torch.cat([confidences0, confidences1], -1)
        pos = 1.0 - (confidences < self.confidence_threshold(layer_index)).float().sum() / num_points
        return pos > 0.95

    def stop_iteration(self, m_last, n_last, m_current, n_current, confidence=0.975):
        return (m_current + n_current) / (m_last + n_last) > confidence

    def _pool_logit(self, pp, x, score4, lens=None, T: int = 0):
        """PoolingLayer.forward up to the pre-sigmoid logit (nets/adagml.py:132-136); rows beyond lens are skipped.  -> [rows, 4]:
        the last Linear is padded to four outputs (16-byte rows), the logit is column 0."""
        kw = dict(lens=lens, t_pad=T)
        s = ops.linear(score4, pp["se0_w"], pp["se0_b"], **kw)
        ops.layernorm_gelu_(s, pp["se1_w"], pp["se1_b"], **kw)
        s = ops.linear(s, pp["se3_w"], pp["se3_b"], **kw)
        xx = ops.linear(x, pp["proj_w"], pp["proj_b"], **kw)
        h = ops.linear(xx, pp["pr0_w"], pp["pr0_b"], x2=s, **kw)
        ops.layernorm_gelu_(h, pp["pr1_w"], pp["pr1_b"], **kw)
        return ops.linear(h, pp["pr3_w"], pp["pr3_b"], **kw)

    @torch.no_grad()
    @blk.with_model_precision
    def produce_matches(self, data: dict, p: float = 0.2, **kwargs):
        """Batched, device-resident AdaGML (nets/adagml.py:307-404 per pair).  All nI layers are enqueued;
        per-pair state (token counts, survivor ids, stop flag, the matching descriptors of the layer the
        pair stopped at) lives in device tensors and is committed with masked selects, so the early exit of
        the reference's Python loop costs no device->host read.  A pair that has stopped gets token count 0
        for the remaining layers (its attention workgroups exit immediately)."""
        desc0, desc1 = data['descriptors0'], data['descriptors1']
        blk.require_cuda(desc0, "AdaGML.produce_matches")
        _ = data['scores0'], data['scores1']     # read like the reference (KeyError if absent), unused in compute
        probes = kwargs.get('probes')
        internal = kwargs.get('_internal')
        (k0, cx0, cy0, sc0), (k1, cx1, cy1, sc1) = normalize_inputs(data)
        P = self._packed_get(self._build_packed)
        B, m, _ = desc0.shape
        n = desc1.shape[1]
        T = max(m, n)
        dev = desc0.device
        X = torch.zeros(2 * B, T, desc0.shape[2], device=dev, dtype=torch.float32)
        X[:B, :m], X[B:, :n] = desc0, desc1
        cos, sin = stack_encodings(k0, k1, (cx0, cy0, sc0), (cx1, cy1, sc1), P["Wr"], T)
        ind = torch.zeros(2 * B, T, device=dev, dtype=torch.int32)
        ind[:B, :m] = torch.arange(m, device=dev, dtype=torch.int32)
        ind[B:, :n] = torch.arange(n, device=dev, dtype=torch.int32)
        ind = ind.contiguous()
        l0 = data.get('lens0')
        l1 = data.get('lens1')
        lens = torch.cat([l0.int() if l0 is not None else torch.full((B,), m, device=dev, dtype=torch.int32),
                          l1.int() if l1 is not None else torch.full((B,), n, device=dev, dtype=torch.int32)]).contiguous()
        num_points = (lens[:B] + lens[B:]).float().contiguous()            # m + n of the ORIGINAL sets (adagml.py:370)
        # per-pair state on the device (ops.adagml_layer_state, one kernel per layer): which pairs still run, their token counts,
        # whether a still-running pair was pruned to <= 5 tokens (run()), the layer a pair stopped at and what it held then
        active = ops._filled((B,), dev, torch.int32, 1)
        tiny = ops._filled((B,), dev, torch.int32)
        stop_layer = ops._filled((B,), dev, torch.int32, -1)
        d = self.config['hidden_dim']
        planes_md = blk._split_path() and d % 32 == 0       # matching descriptors as split planes: the score matrix runs on the fp16 pipe too
        md_final = None if planes_md else ops._filled((2 * B, T, d), dev)
        md_planes = ops._filled((2, 2 * B * T, d // 2), dev).view(torch.float16) if planes_md else None
        lens_final = lens.clone()
        ind_final = ind.clone()
        lens_eff = lens
        x = ops.linear(X.view(2 * B * T, -1), P["in_w"], P["in_b"])
        nI = self.n_layers
        for ni in range(nI):
            last = ni == nI - 1
            x, col_self = blk.self_block(x, P["self"][ni], cos.view(-1, 32), sin.view(-1, 32), 2 * B, T, lens_eff, want_colmean=True)
            x, col_cross = blk.cross_block(x, P["cross"][ni], B, T, lens_eff, want_colmean=True)
            score4 = ops.adagml_scores4(col_self, col_cross)
            logit4 = self._pool_logit(P["pool"][ni], x, score4, lens_eff, T)
            lens_new = n_below = None
            if ni >= 1:
                thr = self.confidence_threshold(ni)
                x3, cos, sin, ind, lens_new, n_below, conf = ops.adagml_prune(
                    logit4, thr, self.n_min_tokens, lens_eff, x.view(2 * B, T, -1), cos, sin, ind, want_conf=probes is not None, ld_logit=4)
                x = x3.view(2 * B * T, -1)
                if probes is not None:
                    probes[f"conf_{ni}"] = conf
            elif probes is not None:
                probes[f"conf_{ni}"] = torch.sigmoid(logit4[:, 0].contiguous().view(2 * B, T))
            # commit the pruned counts, check_if_stop (adagml.py:522-531) and the per-pair commits of the pairs stopping here
            active, lens, lens_stop, lens_eff = ops.adagml_layer_state(active, lens, lens_new, n_below, num_points, tiny, stop_layer, lens_final,
                                                                       ind, ind_final, B, T, ni, last)
            if ni >= 1 or last:
                # out_proj only for the pairs that stop at this layer, written straight into their rows of md_final: the ragged
                # GEMM skips every tile of the other pairs (lens 0) and leaves their rows as they are
                if planes_md:
                    ops.linear(x, P["out_w"][ni], P["out_b"][ni], alpha=1.0 / d ** .25, split_out="only", out_planes=(md_planes[0], md_planes[1]),
                               lens=lens_stop, t_pad=T)
                else:
                    ops.linear(x, P["out_w"][ni], P["out_b"][ni], alpha=1.0 / d ** .25, out=md_final.view(2 * B * T, d), lens=lens_stop, t_pad=T)
        ldc = (T + 3) // 4 * 4
        if planes_md:
            dist = ops.bgemm_nt_planes((md_planes[0][:B * T], md_planes[1][:B * T]), (md_planes[0][B * T:], md_planes[1][B * T:]), B, T, T, ldc=ldc)
        else:
            dist = ops.bgemm_nt(md_final[:B].contiguous(), md_final[B:].contiguous(), ldc=ldc)
        lf0, lf1 = lens_final[:B].contiguous(), lens_final[B:].contiguous()
        r = ops.sinkhorn_match(dist, P["bin"], self.sinkhorn_iterations, p, m_lens=lf0, n_lens=lf1,
                               dual_softmax=not self.with_sinkhorn, n_valid=T)
        out_m, out_s = ops.adagml_scatter(r['matches0'], r['matching_scores0'], ind_final[:B].contiguous(),
                                          ind_final[B:].contiguous(), lf0, m)
        if probes is not None:
            probes.update(stop_layer=stop_layer, ind=ind_final, lens=lens_final)
        if internal is not None:
            internal.update(matches0=r['matches0'], ind=ind_final, lens=lens_final, tiny=tiny, stop_layer=stop_layer)
        return {'matches0': out_m, 'matching_scores0': out_s}
