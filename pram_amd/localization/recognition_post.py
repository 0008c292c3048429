"""Recogniser epilogue — SURVEY.md §8(f) row 1.

Reference: Frame.add_segmentations (localization/frame.py:96-121) and
MultiMap3D.process_segmentations (localization/multimap3d.py:348-379).  The per-token work (softmax,
background test, argmax, the full descending sort over classes) runs in HIP; the landmark voting
loop over sorted columns is the reference's host logic, kept on the host."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from pram_amd import ops
from pram_amd.nets._blocks import require_cuda


@torch.no_grad()
def add_segmentations(segmentations: torch.Tensor, filtering_threshold: float) -> dict:
    """segmentations [N, C] logits (one frame).  Returns what Frame.add_segmentations stores:
    'keep' (bool [N] or None when no filtering happened), 'segmentations', 'seg_scores', 'seg_ids'."""
    require_cuda(segmentations, "add_segmentations")
    seg = segmentations.float().contiguous()[None]
    ids, mask, cnt, scores = ops.seg_epilogue(seg, None, filtering_threshold if filtering_threshold > 0 else 2.0, want_scores=True)
    ids, mask, scores, seg = ids[0], mask[0].bool(), scores[0], seg[0]
    keep = None
    if filtering_threshold > 0 and int(cnt.item()) >= 0.4 * seg.shape[0]:      # frame.py:106
        keep = mask
        seg, scores, ids = seg[keep], scores[keep], ids[keep]
    return {'keep': keep, 'segmentations': seg, 'seg_scores': scores, 'seg_ids': ids.long()}


@torch.no_grad()
def process_segmentations(segs: torch.Tensor, topk: int = 10) -> List[Tuple[int, np.ndarray, float]]:
    """multimap3d.py:348-379: rank landmarks by how many tokens put them at sorted position k.
    segs [N, C] seg scores.  -> list of (sid, token ids, mean score), at most ``topk`` entries."""
    require_cuda(segs, "process_segmentations")
    vals, idx = ops.row_sort_desc(segs.float().contiguous())      # == torch.topk(segs, k=C), canonical tie order
    pred_values, pred_ids = vals.cpu().numpy(), idx.cpu().numpy()
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


@torch.no_grad()
def match_by_projection(q_kpts: torch.Tensor, q_descs: torch.Tensor, proj_uvs: torch.Tensor, ref_descs: torch.Tensor,
                        threshold: float):
    """Descriptor matching of SingleMap3D.refine_pose_by_projection (singlemap3d.py:416-433):
    q_kpts [M,2], q_descs [M,D], proj_uvs [2,N] projected map points, ref_descs [N,D].
    Returns (ratio_mask bool [M], ids int64 [M] index of the nearest reference, dists [M,2])."""
    require_cuda(q_descs, "match_by_projection")
    m, n = q_descs.shape[0], ref_descs.shape[0]
    sim = ops.bgemm_nt(q_descs.float().contiguous()[None], ref_descs.float().contiguous()[None], ldc=(n + 3) // 4 * 4)[0]
    d0, d1, i0 = ops.proj_dist_top2(sim, q_kpts[:, :2], proj_uvs[:2], 2 * threshold, n_valid=n)
    ratios = d0 / d1
    ratio_mask = (ratios <= 0.995) & (d0 < 100)
    return ratio_mask, i0, torch.stack([d0, d1], 1)
