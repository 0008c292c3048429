"""Recogniser epilogue and projection refinement — SURVEY.md §8(f) rows 1 and 2.

Reference: Frame.add_segmentations (localization/frame.py:96-121), MultiMap3D.process_segmentations
(localization/multimap3d.py:348-379) and the device part of SingleMap3D.refine_pose_by_projection
(localization/singlemap3d.py:405-444).  Everything per token / per map point runs in HIP — softmax, background test, argmax,
the full descending sort over classes, the landmark vote, the float64 projection with its frustum test and ordered
compaction, the masked descriptor distances with their top-2 — and only the handful of winners / survivors crosses to the host."""
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
    """Landmark ranking of multimap3d.py:348-379 for one frame.  segs [N, C] seg scores on the GPU.
    -> at most ``topk`` (landmark id, token ids, mean score) entries, best first.

    The class lists are sorted on the device (row_sort_desc: the reference's torch.topk(k=C), canonical tie order) and the
    vote runs there too (ops.seg_vote: per sorted position a class histogram over the tokens, landmarks taken by token count,
    each landmark once): the [N, C] score matrix never leaves the GPU, only the winners' ids, token lists and means do."""
    require_cuda(segs, "process_segmentations")
    vals, idx = ops.row_sort_desc(segs.float().contiguous())
    sid, _, cnt, nwin, tokens, mean = ops.seg_vote(vals, idx, topk)
    n = int(nwin.item())
    if n == 0:
        return []
    sid, cnt, mean = sid[:n].cpu().numpy(), cnt[:n].cpu().numpy(), mean[:n].cpu().numpy()
    toks = tokens[:n, :int(cnt.max())].cpu().numpy()
    return [(int(sid[w]), toks[w, :cnt[w]].astype(np.int64), mean[w]) for w in range(n)]


@torch.no_grad()
def project_map_points(xyzs, K, Tcw, im_w: int, im_h: int):
    """singlemap3d.py:405-415 in float64 like the reference: proj = K (Tcw [X 1])[:3], u = p0 / p2, v = p1 / p2, kept when
    0 < depth < 100 and (u, v) inside the image.  xyzs [N,3], K [3,3], Tcw [4,4] (numpy or tensors).
    -> (proj_uvs [3, n_keep] float64 = u, v, depth of the survivors in their original order, mask bool [N])."""
    dev = torch.device("cuda", torch.cuda.current_device())
    f = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, dtype=torch.float64).to(dev)
    uvd, mask, keep, uvk, count = ops.project_points(f(xyzs).reshape(-1, 3), f(K), f(Tcw), float(im_w), float(im_h))
    n = int(count.item())                                   # the reference synchronises here too (boolean indexing)
    keep = keep[:n].long()
    return torch.cat([uvk[:, :n], uvd[2:3, keep]], 0), mask.bool()


@torch.no_grad()
def refine_matches_by_projection(q_kpts, q_descs, xyzs, descs, K, Tcw, im_w: int, im_h: int, threshold: float):
    """The device part of SingleMap3D.refine_pose_by_projection (singlemap3d.py:405-444): project the map points, keep those in
    the frustum, match query descriptors against them with the projection-radius penalty and the 0.995 ratio test.
    q_kpts [M, >=2] float32, q_descs [M, D], xyzs [N, 3] float64, descs [N, D].
    -> dict(point_mask bool [N], ratio_mask bool [M], matched_keypoint_ids int64 [m], matched_point_ids int64 [m] (indices into
    the N map points), dists float32 [M, 2]) — what the reference hands to the pose solver (mkpts = q_kpts[ids], mxyzs =
    xyzs[matched_point_ids])."""
    dev = torch.device("cuda", torch.cuda.current_device())
    f64 = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a, dtype=torch.float64).to(dev)
    f32 = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dev).float()
    uvd, mask, keep, uvk, count = ops.project_points(f64(xyzs).reshape(-1, 3), f64(K), f64(Tcw), float(im_w), float(im_h))
    n = int(count.item())
    keep = keep[:n].long()
    qk, qd = f32(q_kpts)[:, :2].contiguous(), f32(q_descs).contiguous()
    m = qd.shape[0]
    out = {"point_mask": mask.bool()}
    if n < 2 or m == 0:        # the reference's topk(k = 2) needs two candidates
        out.update(ratio_mask=torch.zeros(m, dtype=torch.bool, device=dev), matched_keypoint_ids=torch.zeros(0, dtype=torch.long, device=dev),
                   matched_point_ids=torch.zeros(0, dtype=torch.long, device=dev), dists=torch.zeros(m, 2, device=dev))
        return out
    rd = f32(descs)[keep].contiguous()
    sim = ops.bgemm_nt(qd[None], rd[None], ldc=(n + 3) // 4 * 4)[0]
    d0, d1, i0 = ops.proj_dist_top2_f64uv(sim, qk, uvk, 2.0 * float(threshold), n)
    ratio_mask = ((d0 / d1) <= 0.995) & (d0 < 100)
    ids = i0[ratio_mask]
    out.update(ratio_mask=ratio_mask, matched_keypoint_ids=torch.nonzero(ratio_mask).flatten(), matched_point_ids=keep[ids],
               dists=torch.stack([d0, d1], 1))
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
