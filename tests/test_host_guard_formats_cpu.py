"""CPU: host logic added in round 3 — the synthetic reference sets of bench.py at unequal sizes, the centred first Linear of the
MLP pair, the fitted GELU tail, the rank pinning helper, and that the header, the ctypes table and the library agree on the new
entry points."""
import os
import re
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_reference_sets_of_the_secondary_shape_are_consistent():
    """--match-kpts 512 --ref-kpts 1024: twins are drawn among the first 512 query keypoints only; gt points at the twin's row."""
    import bench
    g = torch.Generator().manual_seed(3)
    B, k, D, M, N = 2, 96, 16, 40, 64
    desc = torch.nn.functional.normalize(torch.randn(B, k, D, generator=g), dim=-1)
    kp = torch.floor(torch.rand(B, k, 2, generator=g) * 400) + 4
    sc = torch.rand(B, k, generator=g)
    counts = [k, 30]                                    # query 1 has only 30 real keypoints: padded rows have no twin
    ref, gt = bench.make_reference_sets(desc, kp, sc, counts, 5000, n_ref=N, m_match=M)
    assert tuple(ref["descriptors"].shape) == (B, N, D) and tuple(ref["keypoints"].shape) == (B, N, 2) and tuple(gt.shape) == (B, M)
    n_in = min(N - N // 4, M)
    for b in range(B):
        tw = gt[b][gt[b] >= 0]
        assert tw.numel() == tw.unique().numel() and int(tw.max()) < n_in      # one-to-one, inlier rows only
        for i in torch.nonzero(gt[b] >= 0).flatten().tolist():
            assert i < counts[b]
            assert torch.equal(ref["keypoints"][b, gt[b, i]], kp[b, i])         # the twin sits at the query keypoint's position
            assert float((ref["descriptors"][b, gt[b, i]] * desc[b, i]).sum()) > 0.5
    assert int((gt[1] >= 0).sum()) <= 30
    # default shape: unchanged semantics (k inliers - k/4 outliers, twins for every real keypoint among the inliers)
    ref2, gt2 = bench.make_reference_sets(desc, kp, sc, [k, k], 5000)
    assert tuple(gt2.shape) == (B, k) and int((gt2[0] >= 0).sum()) == k - k // 4


def test_centred_first_linear_leaves_the_layernorm_unchanged():
    """LayerNorm is invariant to a common shift of its inputs: Linear with weights centred over the outputs gives h - mean(h),
    the same LayerNorm output, and mean(out^2) is the variance (what the GEMM pair of ops.mlp_tail relies on)."""
    from pram_amd.ops import center_linear
    g = torch.Generator().manual_seed(5)
    w, b = torch.randn(48, 32, generator=g, dtype=torch.float64), torch.randn(48, generator=g, dtype=torch.float64) + 3.0
    x = torch.randn(7, 32, generator=g, dtype=torch.float64)
    wc, bc = center_linear(w, b)
    h = x @ w.t() + b
    hc = x @ wc.double().t() + bc.double()
    assert float((hc - (h - h.mean(1, keepdim=True))).abs().max()) < 1e-5
    assert float((hc.pow(2).mean(1) - h.var(1, unbiased=False)).abs().max()) < 1e-4
    ln = torch.nn.functional.layer_norm
    assert float((ln(hc, (48,)) - ln(h, (48,))).abs().max()) < 1e-5


def test_fitted_gaussian_tail_of_the_fused_gelu():
    """The coefficients compiled into linear.hip::gelu_erf reproduce GELU to 7.5e-8 |t| (fp32 Horner, like the kernel)."""
    src = (ROOT / "pram_amd" / "csrc" / "linear.hip").read_text()
    body = src[src.index("__device__ __forceinline__ float gelu_erf(float t)"):]
    body = body[:body.index("\n}\n")]
    c = [np.float32(v) for v in re.findall(r"(-?\d\.\d+(?:e-?\d+)?)f", body) if v not in ("6.0", "0.0", "1.0")]
    assert len(c) == 8, c                                # degree 7, highest power first in the Horner chain
    from scipy.special import erfc
    t = np.linspace(-9.0, 9.0, 180001)
    a = np.minimum(np.abs(t), 6.0).astype(np.float32)
    q = np.full(a.shape, c[0], np.float32)
    for k in c[1:]:
        q = q * a + k
    s = np.exp2(-q.astype(np.float64))
    gelu = t * np.where(t >= 0, 1.0 - s, s)
    want = t * (1.0 - 0.5 * erfc(t / np.sqrt(2)))
    err = np.abs(gelu - want)
    assert float((err / np.maximum(np.abs(t), 1.0)).max()) < 1.5e-7, float(err.max())


def test_rank_pinning_spreads_over_the_allowed_cores():
    import bench
    saved = os.sched_getaffinity(0)
    try:
        allowed = sorted(saved)
        if len(allowed) >= 2:
            cores = []
            for r in range(2):
                os.sched_setaffinity(0, saved)
                cores.append(bench.pin_rank_thread(r, 2))
            assert cores[0] != cores[1] and all(c in allowed for c in cores)
        os.sched_setaffinity(0, saved)
        assert bench.pin_rank_thread(0, len(allowed) + 1) is None        # fewer cores than ranks: left alone
    finally:
        os.sched_setaffinity(0, saved)


def test_new_entry_points_are_declared_bound_and_exported():
    from pram_amd import _lib
    header = (ROOT / "include" / "pram_hip.h").read_text()
    new = ["pram_set_status_word", "pram_read_status_word", "pram_attention_x3_workspace_bytes", "pram_attention_x3_is_split",
           "pram_attention_x3_set_split_target", "pram_attention_x3_set_chunk_keys", "pram_attention_x3_mfma_per_tile",
           "pram_linear_x3_ssq_parts", "pram_linear_x3_ssq_f32", "pram_linear_x3_lngelu_f32", "pram_adagml_layer_state",
           "pram_adagml_scores4_f32", "pram_adagml_prune_ld_f32", "pram_pack_record_f32", "pram_fill_u32"]
    L = _lib.load()
    for name in new:
        assert re.search(r"\b" + name + r"\s*\(", header), name
        assert name in _lib.exported_symbols() and hasattr(L, name), name
    assert L.pram_hip_version() >= 110
    # probabilities as two fp16 parts by default (three MFMAs per P V product, 48 per tile); as one fp16: 40 from 1024 keys on
    prev = L.pram_attention_x3_set_p_split(-1)
    try:
        assert L.pram_attention_x3_set_p_split(1) == 1
        assert L.pram_attention_x3_mfma_per_tile(2048) == 48 and L.pram_attention_x3_mfma_per_tile(512) == 48
        assert L.pram_attention_x3_set_p_split(0) == 0
        assert L.pram_attention_x3_mfma_per_tile(2048) == 40 and L.pram_attention_x3_mfma_per_tile(512) == 48
    finally:
        L.pram_attention_x3_set_p_split(prev)
    assert L.pram_linear_x3_ssq_parts(1000, 512, 512) == 8 and L.pram_linear_x3_ssq_parts(1, 1024, 256) == 16
    # the key-chunk geometry (no GPU needed): one chunk per 4096 keys by default -> no workspace, no split at 2048 or 4096 keys
    assert L.pram_attention_x3_set_chunk_keys(0) == 4096
    assert L.pram_attention_x3_workspace_bytes(8, 4, 4096, 4096) == 0 and L.pram_attention_x3_is_split(1, 4, 4096, 4096) == 1
    assert L.pram_attention_x3_workspace_bytes(1, 4, 2048, 2048) == 0 and L.pram_attention_x3_is_split(1, 4, 2048, 2048) == 1
    assert L.pram_attention_x3_workspace_bytes(1, 4, 2048, 8192) > 0 and L.pram_attention_x3_is_split(1, 4, 2048, 8192) == 2
    assert L.pram_attention_x3_set_chunk_keys(512) == 512
    try:
        assert L.pram_attention_x3_is_split(1, 4, 2048, 2048) == 4 and L.pram_attention_x3_is_split(2, 4, 2048, 2048) == 2
        assert L.pram_attention_x3_is_split(16, 4, 2048, 2048) == 1 and L.pram_attention_x3_is_split(1, 4, 512, 512) == 1
    finally:
        L.pram_attention_x3_set_chunk_keys(4096)


def test_guard_lowers_scales_before_it_leaves_the_split_path(monkeypatch):
    """ops.guarded_call's policy on a host without a GPU (the device reads are stubbed): a tripped guard divides the activation scale of
    every model that ran inside the call by 16 and re-runs; the models keep it; only when no scale is left (ACT_SCALE_MIN) does the
    call run under forced_precision('f32'); 'raise' raises; 'deferred' and an un-armed device return at once; nested guarded calls
    leave the decision to the outermost one."""
    from pram_amd import ops
    from pram_amd._lib import PramHipError

    class Model:
        act_scale = 16.0

    trips = {"left": 0}
    runs = []
    monkeypatch.setattr(ops, "x3_launched", lambda device=None: True)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)

    def exceeded(device=None, reset=True):
        if trips["left"] > 0:
            trips["left"] -= 1
            return True
        return False
    monkeypatch.setattr(ops, "x3_range_exceeded", exceeded)
    a, b = Model(), Model()

    def call():
        for m in (a, b):
            ops.guarded_call(lambda m=m: ops.note_model_ran(m), "dev")      # nested model calls: registered, never deciding
        runs.append((a.act_scale, b.act_scale, ops.gemm_prec()))
        return len(runs)

    ev = dict(ops.guard_events)
    with ops.guard_scope("fallback"):
        trips["left"] = 2                                                   # trips at 16 and at 1, fits at 1/16
        assert ops.guarded_call(call, "dev") == 3
    assert [r[:2] for r in runs] == [(16.0, 16.0), (1.0, 1.0), (1.0 / 16, 1.0 / 16)] and a.act_scale == 1.0 / 16
    assert ops.guard_events["rescaled"] == ev["rescaled"] + 2 and ops.guard_events["f32_fallback"] == ev["f32_fallback"]
    runs.clear()
    with ops.guard_scope("fallback"):
        trips["left"] = 5                                                   # nothing fits: 1/16 -> 1/256 -> the exact-fp32 kernels
        ops.guarded_call(call, "dev")
    assert [r[0] for r in runs] == [1.0 / 16, 1.0 / 256, 1.0 / 256] and runs[-1][2] == "f32" and runs[0][2] != "f32"
    assert ops.guard_events["f32_fallback"] == ev["f32_fallback"] + 1 and a.act_scale == 1.0 / 256
    trips["left"] = 1
    with ops.guard_scope("raise"):
        try:
            ops.guarded_call(call, "dev")
            raise AssertionError("the raise policy must raise")
        except PramHipError:
            pass
    trips["left"] = 1
    n = len(runs)
    with ops.guard_scope("deferred"):
        ops.guarded_call(call, "dev")
    assert len(runs) == n + 1 and trips["left"] == 1                        # never asked
