"""Matcher plugin 'nearest_neighbor': mutual nearest neighbours on descriptor similarity with optional ratio /
distance tests (behaviour of the reference's localization/matchers/nearest_neighbor.py:5-56).

sim = A·Bᵀ runs on the f32 MFMA GEMM; the per-row best / second-best (ties -> lowest index) come from one
wave-reduction kernel instead of torch.topk."""
import torch

from pram_amd import ops
from pram_amd.localization.base_model import BaseModel
from pram_amd.nets._blocks import require_cuda


def _nearest(a: torch.Tensor, b: torch.Tensor, ratio, max_dist):
    """a [B,M,D], b [B,N,D] unit descriptors -> (match index int64 [B,M] or -1, score [B,M]).
    Squared L2 distance of unit vectors is 2(1 - sim); accept when d0 <= ratio^2 * d1 and d0 <= max_dist^2."""
    n = b.shape[1]
    sim = ops.bgemm_nt(a, b, ldc=(n + 3) // 4 * 4)
    best, second, idx = ops.row_top2(sim, largest=True, n_valid=n)
    d_best, d_second = 2 * (1 - best), 2 * (1 - second)
    accept = torch.ones_like(idx, dtype=torch.bool)
    if ratio:
        accept &= d_best <= (ratio ** 2) * d_second
    if max_dist:
        accept &= d_best <= max_dist ** 2
    no_match = torch.full_like(idx, -1)
    return torch.where(accept, idx, no_match), torch.where(accept, (best + 1) / 2, torch.zeros_like(best))


def _keep_mutual(m01: torch.Tensor, m10: torch.Tensor) -> torch.Tensor:
    """Drop i -> j unless j -> i."""
    rows = torch.arange(m01.shape[-1], device=m01.device)
    back = torch.gather(m10, -1, m01.clamp(min=0))
    return torch.where((m01 >= 0) & (back == rows), m01, torch.full_like(m01, -1))


class NearestNeighbor(BaseModel):
    default_conf = {'ratio_threshold': None, 'distance_threshold': None, 'do_mutual_check': True}

    def _init(self, conf):
        pass

    @torch.no_grad()
    def _forward(self, data):
        d0, d1 = data['descriptors0'], data['descriptors1']        # [B, D, N] and [B, D, M], channel-major like the reference
        require_cuda(d0, "NearestNeighbor")
        a = d0.transpose(1, 2).contiguous().float()
        b = d1.transpose(1, 2).contiguous().float()
        ratio, max_dist = self.conf['ratio_threshold'], self.conf['distance_threshold']
        matches0, scores0 = _nearest(a, b, ratio, max_dist)
        if self.conf['do_mutual_check']:
            matches1, _ = _nearest(b, a, ratio, max_dist)
            matches0 = _keep_mutual(matches0, matches1)
        return {'matches0': matches0, 'matching_scores0': scores0}
