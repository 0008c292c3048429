"""Matcher plugin 'nearest_neighbor' (reference: localization/matchers/nearest_neighbor.py:5-56):
mutual nearest neighbours on descriptor similarity, optional ratio / distance tests.
sim = A·Bᵀ on the f32 MFMA GEMM, top-2 per row by wave reduction (ties -> lowest index)."""
import torch

from pram_amd import ops
from pram_amd.localization.base_model import BaseModel
from pram_amd.nets._blocks import require_cuda


def find_nn(sim, ratio_thresh, distance_thresh, n_valid=None):
    """sim [B,M,ld] -> matches int64 [B,M] (-1 none), scores [B,M]  (nearest_neighbor.py:5-17)"""
    s0, s1, i0 = ops.row_top2(sim, largest=True, n_valid=n_valid)
    d0, d1 = 2 * (1 - s0), 2 * (1 - s1)
    mask = torch.ones_like(i0, dtype=torch.bool)
    if ratio_thresh:
        mask = mask & (d0 <= (ratio_thresh ** 2) * d1)
    if distance_thresh:
        mask = mask & (d0 <= distance_thresh ** 2)
    matches = torch.where(mask, i0, i0.new_tensor(-1))
    scores = torch.where(mask, (s0 + 1) / 2, s0.new_tensor(0))
    return matches, scores


def mutual_check(m0, m1):
    """nearest_neighbor.py:20-25"""
    inds0 = torch.arange(m0.shape[-1], device=m0.device)
    loop = torch.gather(m1, -1, torch.where(m0 > -1, m0, m0.new_tensor(0)))
    ok = (m0 > -1) & (inds0 == loop)
    return torch.where(ok, m0, m0.new_tensor(-1))


class NearestNeighbor(BaseModel):
    default_conf = {'ratio_threshold': None, 'distance_threshold': None, 'do_mutual_check': True}
    required_inputs = ['descriptors0', 'descriptors1']

    def _init(self, conf):
        pass

    @torch.no_grad()
    def _forward(self, data):
        d0, d1 = data['descriptors0'], data['descriptors1']      # [B, D, N], [B, D, M]  ('bdn,bdm->bnm')
        require_cuda(d0, "NearestNeighbor")
        a = d0.transpose(1, 2).contiguous().float()
        b = d1.transpose(1, 2).contiguous().float()
        n, m = a.shape[1], b.shape[1]
        sim = ops.bgemm_nt(a, b, ldc=(m + 3) // 4 * 4)
        matches0, scores0 = find_nn(sim, self.conf['ratio_threshold'], self.conf['distance_threshold'], n_valid=m)
        if self.conf['do_mutual_check']:
            sim_t = ops.bgemm_nt(b, a, ldc=(n + 3) // 4 * 4)
            matches1, _ = find_nn(sim_t, self.conf['ratio_threshold'], self.conf['distance_threshold'], n_valid=n)
            matches0 = mutual_check(matches0, matches1)
        return {'matches0': matches0, 'matching_scores0': scores0}
