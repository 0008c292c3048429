"""Batched per-query hot path: SFD2 extract -> sample -> SegNetViT recognise -> GML/AdaGML match.

Reproduces the model-call sequence of the reference's per-query loop
(localization/loc_by_rec_online.py:109-133 followed by the matcher call of
localization/singlemap3d.py:143-154) for a batch of independent query frames, entirely on the
device: ragged keypoint counts travel as an int32 ``counts`` array, nothing synchronises with the
host until the caller reads the result record.  Queries shard across GPUs with no data-path
collective; ``gather_records`` is the single RCCL all-gather of the fixed-size result records
(SURVEY.md §8(e)).

Recognition and matching both depend only on the extraction, so with few queries in flight (the
reference's online loop is one frame at a time: 64 attention workgroups for 256 CUs) the matcher
runs on a second HIP stream next to the recogniser and the two half-empty launch sequences fill
the chip together; results are identical (same kernels, same launch configurations).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


import threading as _threading

# (device index, main stream handle) -> a stream that was MEASURED to run beside that main stream.  Process-wide and shared on
# purpose: every pipeline / captured graph that forks beside the same main stream uses the SAME side stream (their branches are
# ordered on it — graphs replayed on one main stream run one after the other anyway).  Guarded by _BRANCH_LOCK; entries live for
# the process (a handle the runtime re-uses for a new stream keeps its measured partner: placement is for speed only).
_BRANCH = {}
_BRANCH_LOCK = _threading.Lock()


def _branch_stream(device, main: Optional[torch.cuda.Stream] = None) -> torch.cuda.Stream:
    """A stream for the forked branch (recognise || match) that really runs beside ``main`` (default: the current stream).

    ROCm multiplexes a process's HIP streams onto a few hardware queues by creation order, so a stream created after a handful of
    others can share its queue with the stream it is meant to overlap: the two branches of the step then run one after the other
    and nothing says so (measured: one-query latency 3.9 -> 5.1 ms = the serial sum, in a process that had created ~25 streams
    before; a high-priority stream made it 10.4; profiles/r05_stream_queue_collision.txt).  So the choice is measured, once per
    (device, main stream): two ~0.15 ms spin kernels (torch.cuda._sleep), one per stream; a candidate whose pair takes about the
    time of one is kept.  Placement is for speed only; any stream gives the same results.

    The measurement synchronises the device (~30 times) and must not run inside a stream capture: QueryPipeline.warm_up() /
    GraphedPipeline's constructor do it ahead of time; a first forked run() that finds itself inside a capture raises instead of
    invalidating it."""
    import time
    dev = torch.device(device)
    main = main or torch.cuda.current_stream(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), main.cuda_stream)
    with _BRANCH_LOCK:
        if key in _BRANCH:
            return _BRANCH[key]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("pram_amd.pipeline: the side stream of a forked step has to be measured before a stream capture — call "
                           "QueryPipeline.warm_up(device) (or run one step) outside the capture first")

    def spin(cands, cycles=300_000):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for st in cands:
            with torch.cuda.stream(st):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    spin([main])
    one = min(spin([main]) for _ in range(3))
    best, best_t = None, float("inf")
    for _ in range(8):
        c = torch.cuda.Stream(device=dev)
        spin([main, c])
        t = min(spin([main, c]) for _ in range(2))
        if t < best_t:
            best, best_t = c, t
        if t < 1.5 * one:
            break
    with _BRANCH_LOCK:
        return _BRANCH.setdefault(key, best)      # (another thread may have measured the same pair meanwhile: one winner)


class QueryPipeline:
    """guard: range guard of the split-fp16 path for one run() (ops.guarded_call): "fallback" (default) reads the device status
    word once after the last launch — the one host synchronisation of a run — and re-runs the batch on the exact-fp32 kernels
    when an activation left the fp16 parts' range; "raise" raises instead; "deferred" never synchronises (batches in flight,
    hipGraph capture): the caller asks ``ops.x3_range_exceeded()`` when it reads the results.
    match_keypoints: only the first (= best-scoring) M keypoints of every query enter the matcher (SURVEY.md §8(d)'s secondary
    shape: the keypoints voted to one landmark, localization/multimap3d.py:131-139); 0 = all."""

    def __init__(self, sfd2, segnet, matcher, max_keypoints: int = 2048, min_keypoints: int = 128, bg_threshold: float = 0.95,
                 overlap_below: int = 8, guard: str = "fallback", match_keypoints: int = 0):
        self.sfd2, self.segnet, self.matcher = sfd2, segnet, matcher
        self.bg_threshold = bg_threshold
        self.cfg = {'min_keypoints': min_keypoints, 'max_keypoints': max_keypoints}
        self.overlap_below = overlap_below      # batches smaller than this run recognise || match on two streams
        self.guard = guard
        self.match_keypoints = int(match_keypoints)
        self._side, self._side_main = None, None

    def warm_up(self, device, main: Optional[torch.cuda.Stream] = None) -> None:
        """Measure the side stream of forked steps (batches below overlap_below) beside ``main`` (default: the current stream) NOW:
        ~30 device synchronisations that would otherwise happen inside the first forked run() — and must not happen inside a
        stream capture or beside another thread's timing."""
        dev = torch.device(device)
        main = main or torch.cuda.current_stream(dev)
        self._side, self._side_main = _branch_stream(dev, main), main.cuda_stream

    def _match(self, ex, ref, W, H):
        kpts, scores, counts, desc = ex['keypoints'], ex['scores'], ex['counts'], ex['descriptors']
        km = self.match_keypoints
        if 0 < km < kpts.shape[1]:
            kpts, scores, desc = kpts[:, :km].contiguous(), scores[:, :km].contiguous(), desc[:, :km].contiguous()
            counts = torch.clamp(counts, max=km)
        data = {
            'descriptors0': desc, 'keypoints0': kpts, 'scores0': scores, 'lens0': counts,
            # the reference hands the matcher (1, 3, width, height) (singlemap3d.py:147,152)
            'image_shape0': (1, 3, W, H),
            'descriptors1': ref['descriptors'], 'keypoints1': ref['keypoints'], 'scores1': ref['scores'],
            'image_shape1': (1, 3, W, H),
        }
        if 'lens' in ref:
            data['lens1'] = ref['lens']
        return self.matcher.produce_matches(data) if hasattr(self.matcher, 'produce_matches') else self.matcher(data)

    @torch.no_grad()
    def run(self, images: torch.Tensor, ref: Optional[Dict[str, torch.Tensor]] = None, stages: str = "erm",
            guard: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """images [B,3,H,W] normalised fp32 on the GPU.  ref: reference-frame sets to match against
        (descriptors [B,Nr,128], keypoints [B,Nr,2], scores [B,Nr], optional lens int32 [B]).
        stages: any of 'e' (extract), 'r' (recognise), 'm' (match).  guard: see the class docstring (None = self.guard)."""
        with ops.guard_scope(guard or self.guard):
            return ops.guarded_call(lambda: self._run(images, ref, stages), images.device)

    def _extract(self, images):
        return self.sfd2.extract_batched(images, self.cfg, per_image_fallback=True)

    def _recognise(self, ex, images, out):
        kpts, counts = ex['keypoints'], ex['counts']
        _, seg = self.sfd2.sample_batched(ex['score_map'], ex['mid_features'], kpts, counts, norm_desc=False)
        pred = self.segnet({'seg_descriptors': seg, 'keypoints': kpts, 'image': images, 'lens': counts})['prediction']
        out['prediction'] = pred
        # recogniser epilogue (Frame.add_segmentations, frame.py:96-121): landmark id = argmax - 1,
        # background mask at the reference's pre_filtering_th (configs/config_train_7scenes_sfd2.yaml:98)
        ids, non_bg, n_non_bg, _ = ops.seg_epilogue(pred, counts, self.bg_threshold)
        out['landmark'], out['non_bg'], out['n_non_bg'] = ids, non_bg, n_non_bg

    def _forks(self, batch: int, ref, stages: str) -> bool:
        return 'm' in stages and ref is not None and 'r' in stages and batch < self.overlap_below

    def _run(self, images, ref, stages):
        B, _, H, W = images.shape
        ex = self._extract(images)
        out = {'keypoints': ex['keypoints'], 'scores': ex['scores'], 'counts': ex['counts'], 'descriptors': ex['descriptors']}
        do_match = 'm' in stages and ref is not None
        forked = self._forks(B, ref, stages)
        if forked:
            main = torch.cuda.current_stream(images.device)
            if self._side is None or self._side.device != images.device or self._side_main != main.cuda_stream:
                self._side, self._side_main = _branch_stream(images.device, main), main.cuda_stream
            self._side.wait_stream(main)                 # extraction results are ready for the side stream
            with torch.cuda.stream(self._side):
                m = self._match(ex, ref, W, H)
        if 'r' in stages:
            self._recognise(ex, images, out)
        if do_match:
            if forked:
                main.wait_stream(self._side)
                for t in (m['matches0'], m['matching_scores0']):
                    t.record_stream(main)               # allocated on the side stream, consumed on the caller's
            else:
                m = self._match(ex, ref, W, H)
            out['matches0'] = m['matches0']
            out['matching_scores0'] = m['matching_scores0']
        return out

    @staticmethod
    def pack_record(out: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Fixed-size per-query record [B, k, 6] fp32: x, y, score, landmark id, match index, match score (one kernel,
        pram_pack_record_f32; keypoints beyond the matcher's share carry match index -1)."""
        return ops.pack_record(out['keypoints'], out['scores'], out.get('landmark'), out.get('matches0'), out.get('matching_scores0'))


def gather_records(rec: torch.Tensor, shard_sizes=None) -> torch.Tensor:
    """One all-gather of the result records over RCCL/xGMI (backend 'nccl' on ROCm; gloo on CPU tests).

    ``shard_sizes`` = queries held by every rank, in rank order (what ``shard_range`` gives each rank — known on every
    rank without communication).  Uneven shards (65 queries over 8 GPUs) are padded to the largest shard for the
    collective and trimmed afterwards, so the result is the [sum(shard_sizes), ...] record array in query order; with
    ``shard_sizes=None`` all ranks must hold the same number of queries."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    ws = dist.get_world_size()
    tail = tuple(rec.shape[1:])
    if shard_sizes is None:
        full = torch.empty((ws * rec.shape[0],) + tail, device=rec.device, dtype=rec.dtype)
        dist.all_gather_into_tensor(full, rec.contiguous())
        return full
    sizes = [int(x) for x in shard_sizes]
    if len(sizes) != ws or sizes[dist.get_rank()] != rec.shape[0]:
        raise ValueError(f"shard_sizes {sizes} do not describe this job (world {ws}, this rank holds {rec.shape[0]})")
    big = max(sizes)
    if big == 0:
        return rec
    mine = rec.contiguous()
    if rec.shape[0] < big:
        mine = torch.zeros((big,) + tail, device=rec.device, dtype=rec.dtype)
        mine[:rec.shape[0]] = rec
    full = torch.empty((ws * big,) + tail, device=rec.device, dtype=rec.dtype)
    dist.all_gather_into_tensor(full, mine)
    if min(sizes) == big:
        return full
    full = full.view((ws, big) + tail)
    return torch.cat([full[r, :sizes[r]] for r in range(ws)], 0)


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous chunk [lo, hi) of ``n_items`` queries for ``rank`` (balanced; first ranks get the remainder)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class GraphedPipeline:
    """QueryPipeline.run captured once as a hipGraph and replayed.

    The step has no host synchronisation and fixed launch geometry for a given (batch, frame size, reference-set
    size), so the whole sequence — ~330 kernels on up to two streams — can be recorded and re-issued with one call.
    GPU time does not change (the step is GPU-bound); what changes is the host: ~3 ms of Python / ctypes launches per
    step become one graph launch, which is what matters when the host thread is also solving poses or feeding
    several GPUs.  Inputs are copied into the captured buffers; outputs are the captured tensors (valid until the next
    replay — clone what must outlive it).  ``record=True`` also captures the result record (``self.record``).
    Range guard: a capture cannot synchronise, so the captured run is "deferred"; ``run()`` reads the status word after the replay
    (``guard`` = the pipeline's policy) and re-runs eagerly on the exact-fp32 kernels when it is set; ``replay()`` is the bare
    re-issue for callers that keep several steps in flight and check ``ops.x3_range_exceeded()`` themselves."""

    def __init__(self, pipe: QueryPipeline, images: torch.Tensor, ref: Optional[Dict[str, torch.Tensor]] = None, stages: str = "erm",
                 warmup: int = 2, record: bool = False, split: Optional[bool] = None, stream: Optional[torch.cuda.Stream] = None):
        self.pipe, self.stages = pipe, stages
        self.images = images.clone()
        self.ref = None if ref is None else {k: v.clone() for k, v in ref.items()}
        self.with_record = record
        self.record = None
        self._ws = {}                 # this graph's private scratch (ops.workspace_scope): never shared, never regrown by others
        # stream: the stream the graph will be REPLAYED on (default: the current one) — the forked branch gets a stream measured to
        # run beside it (_branch_stream)
        self._side = _branch_stream(images.device, stream) if pipe.overlap_below > images.shape[0] else None
        # A step that forks (recognise || match, small batches) is captured as FOUR graphs — extract | match | recognise | record —
        # replayed on two streams: the runtime issues the branches of ONE hipGraph one after the other (measured: a two-branch graph
        # of 2 x 150 kernels replays in 838 us, the same 300 in one chain in 499 us; in the traced step the recogniser's first kernel
        # starts when the matcher is almost through), and it is the overlap of the two that a one-query step lives on.
        forks = pipe._forks(images.shape[0], ref, stages)
        self.split = forks if split is None else (bool(split) and forks)
        side = torch.cuda.Stream(device=images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with ops.workspace_scope(self._ws):
            with torch.cuda.stream(side):                # warm up off the default stream: packs weights, sizes workspaces
                for _ in range(warmup):
                    self._run_eager()
            torch.cuda.current_stream(images.device).wait_stream(side)
            torch.cuda.synchronize(images.device)
            n0 = ops.x3_launch_count(images.device)
            if not self.split:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = self._run_eager()
            else:
                self._capture_split()
            # whether a replay can set the range guard's status word is a property of what was CAPTURED (per-model precisions
            # included), not of the process-wide precision settings
            self.uses_x3 = ops.x3_launch_count(images.device) > n0

    def _capture_split(self):
        pipe, images = self.pipe, self.images
        _, _, H, W = images.shape
        self.g_e, self.g_m, self.g_r, self.g_f = (torch.cuda.CUDAGraph() for _ in range(4))
        # the matcher is captured on the stream it will be replayed on: scratch memory is keyed by stream (ops._workspace), and
        # the two branches must not share any; each graph allocates from its own pool (a pool shared by graphs that run at the
        # same time would hand the temporaries of one to the other)
        cap = torch.cuda.Stream(device=images.device)
        with ops.guard_scope("deferred"):
            with torch.cuda.graph(self.g_e, stream=cap):
                ex = pipe._extract(images)
            out = {'keypoints': ex['keypoints'], 'scores': ex['scores'], 'counts': ex['counts'], 'descriptors': ex['descriptors']}
            with torch.cuda.graph(self.g_m, stream=self._side):
                m = pipe._match(ex, self.ref, W, H)
            with torch.cuda.graph(self.g_r, stream=cap):
                pipe._recognise(ex, images, out)
            out['matches0'], out['matching_scores0'] = m['matches0'], m['matching_scores0']
            if self.with_record:
                with torch.cuda.graph(self.g_f, stream=cap):
                    self.record = QueryPipeline.pack_record(out)
        self._ex = ex                 # keeps the extractor's tensors (graph-pool memory the other graphs read) alive
        self.out = out

    def _replay(self):
        if self.uses_x3:
            ops.mark_x3(self.images.device)      # replays launch nothing through the bindings: arm the range guard's read here
        if not self.split:
            self.graph.replay()
            return
        main = torch.cuda.current_stream(self.images.device)
        self.g_e.replay()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self.g_m.replay()
        self.g_r.replay()
        main.wait_stream(self._side)
        if self.with_record:
            self.g_f.replay()

    def _run_eager(self):
        # the forked branch runs on the side stream measured for the stream this graph replays on (_branch_stream: shared with
        # whatever else forks beside that stream — their branches are ordered on it); the pipeline's own choice is put back after
        dev = self.images.device
        saved = (self.pipe._side, self.pipe._side_main)
        self.pipe._side, self.pipe._side_main = self._side, torch.cuda.current_stream(dev).cuda_stream
        try:
            out = self.pipe.run(self.images, self.ref, self.stages, guard="deferred")
            if self.with_record:
                self.record = QueryPipeline.pack_record(out)
            return out
        finally:
            self.pipe._side, self.pipe._side_main = saved

    def replay(self) -> Dict[str, torch.Tensor]:
        """Re-issue the captured step on the current stream with the inputs already in the captured buffers; no synchronisation,
        no range-guard check."""
        self._replay()
        return self.out

    @torch.no_grad()
    def run(self, images: torch.Tensor, ref: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        if tuple(images.shape) != tuple(self.images.shape):
            raise ValueError(f"graph was captured for images {tuple(self.images.shape)}, got {tuple(images.shape)}")
        if ref is not None:
            if self.ref is None:
                raise ValueError("this graph was captured without reference sets (ref=None): it has no matcher stage to feed")
            for k, v in ref.items():
                if k not in self.ref or tuple(self.ref[k].shape) != tuple(v.shape):
                    raise ValueError(f"ref[{k!r}]: captured {tuple(self.ref[k].shape) if k in self.ref else None}, got {tuple(v.shape)}")
        self.images.copy_(images)
        if ref is not None:
            for k, v in ref.items():
                self.ref[k].copy_(v)
        self._replay()
        guard = self.pipe.guard
        if guard != "deferred" and self.uses_x3 and ops.x3_range_exceeded(images.device):
            if guard == "raise":
                from ._lib import PramHipError
                raise PramHipError("split-fp16 path: an activation beyond |x| < 4094.97 in the replayed step")
            # The captured buffers (self.out, self.record) hold the overflowed replay: the result of THIS call is the eager re-run
            # returned below, under the pipeline's own policy — "fallback" lowers the activation scale of the models that ran
            # (they keep it: a graph captured afterwards replays in range; THIS graph keeps tripping and re-running eagerly until
            # it is rebuilt) and only goes to the exact-fp32 kernels when no scale carries the values.  The re-run gets the CAPTURED inputs (self.images / self.ref were just refreshed from the
            # arguments: `ref=None` means "the captured reference sets", not "no matcher"), and its record is copied INTO the
            # captured record tensor — self.record stays the buffer every later replay writes, so a caller that keeps reading
            # g.record after replay() sees that replay's record, not this call's.  One status word serves the whole device: a caller that
            # keeps OTHER replays in flight on other streams must use replay() and check ops.x3_range_exceeded() itself once they
            # are done — the reset here is not ordered against them.
            out = self.pipe.run(self.images, self.ref, self.stages, guard=guard)
            if self.with_record:
                self.record.copy_(QueryPipeline.pack_record(out))
            return out
        return self.out
