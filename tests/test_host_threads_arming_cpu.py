"""CPU: host logic of round 4 — precision / guard scopes are per host thread, the guard is armed by what was launched (not by the
precision settings), and the header, the ctypes table and the library agree on the new entry points."""
import re
import sys
import threading
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_precision_and_guard_scopes_are_thread_local():
    """One host thread per GPU in one process: a model-level precision override, a forced fallback or a guard policy entered by
    one thread must not leak into another; the module attributes stay the process-wide defaults."""
    from pram_amd import ops
    assert ops.gemm_prec() == ops.gemm_precision and ops.attn_prec() == ops.attention_precision
    seen, go, done = {}, threading.Event(), threading.Event()

    def other():
        go.wait(10)
        seen["gemm"], seen["attn"], seen["guard"] = ops.gemm_prec(), ops.attn_prec(), ops.guard_policy()
        with ops.precision_scope("f16"):
            seen["own"] = ops.gemm_prec()
        done.set()

    t = threading.Thread(target=other)
    t.start()
    base = ops.gemm_precision
    other_p = "f32" if base != "f32" else "x3"
    with ops.precision_scope(other_p), ops.guard_scope("raise"):
        assert ops.gemm_prec() == other_p and ops.attn_prec() == other_p and ops.guard_policy() == "raise"
        with ops.forced_precision("f32"):
            with ops.precision_scope("x3"):              # a model's own setting inside the guard's re-run: the forced path wins
                assert ops.gemm_prec() == "f32" and ops.attn_prec() == "f32"
            go.set()
            assert done.wait(10)
        assert ops.gemm_prec() == other_p
    t.join()
    assert seen == {"gemm": base, "attn": ops.attention_precision, "guard": ops.x3_guard, "own": "f16"}
    assert ops.gemm_prec() == base and ops.guard_policy() == ops.x3_guard
    # the process-wide default still is what an assignment to the module attribute sets (tests and bench.py do that)
    old = ops.gemm_precision
    try:
        ops.gemm_precision = "f16"
        assert ops.gemm_prec() == "f16"
        with ops.precision_scope(None):                   # a model without its own precision: no override
            assert ops.gemm_prec() == "f16"
    finally:
        ops.gemm_precision = old


def test_guard_is_armed_by_launches_not_by_settings():
    """guarded_call consults the status word only when a split-fp16 kernel was launched on the device since the word was last
    read — whatever the precision settings say (a model with .precision = 'x3' inside an f32 process went unchecked)."""
    from pram_amd import ops
    dev = torch.device("cuda", 0)
    calls = []
    saved = ops.x3_range_exceeded
    ops.x3_range_exceeded = lambda device=None, reset=True: (calls.append(device), ops._x3_pending.discard(0), False)[-1]
    saved_capt = torch.cuda.is_current_stream_capturing
    torch.cuda.is_current_stream_capturing = lambda: False
    try:
        ops._x3_pending.discard(0)
        old = ops.gemm_precision, ops.attention_precision
        ops.gemm_precision = ops.attention_precision = "f32"
        try:
            assert ops.guarded_call(lambda: 1, dev) == 1 and calls == []              # nothing launched: no synchronisation
            n0 = ops.x3_launch_count(dev)
            assert ops.guarded_call(lambda: (ops._mark_x3(dev), 2)[-1], dev) == 2     # an x3 launch under f32 settings: checked
            assert len(calls) == 1 and ops.x3_launch_count(dev) == n0 + 1
            assert ops.guarded_call(lambda: ops.guarded_call(lambda: (ops._mark_x3(dev), 3)[-1], dev), dev) == 3
            assert len(calls) == 2                                                    # only the outermost call checks
            with ops.guard_scope("deferred"):
                assert ops.guarded_call(lambda: (ops._mark_x3(dev), 4)[-1], dev) == 4
            assert len(calls) == 2 and ops.x3_launched(dev)                           # left pending for whoever reads the word
        finally:
            ops.gemm_precision, ops.attention_precision = old
    finally:
        ops.x3_range_exceeded = saved
        torch.cuda.is_current_stream_capturing = saved_capt
        ops._x3_pending.discard(0)


def test_round4_entry_points_declared_bound_and_exported():
    from pram_amd import _lib
    header = (ROOT / "include" / "pram_hip.h").read_text()
    names = ["pram_linear_f16_ragged_f32"]
    lib = _lib.load()
    for n in names:
        assert re.search(r"\b" + n + r"\s*\(", header), n
        assert n in _lib._SIGS, n
        assert getattr(lib, n) is not None


def test_every_entry_point_survives_null_arguments():
    """The C ABI validates before it touches anything: every exported entry called with NULL pointers and zero sizes returns an
    error code (or, for the size / query helpers, a harmless value) — never a crash — and the launching ones leave a message in
    pram_last_error.  Run in a child process so that a regression is a failed assertion, not a dead test session."""
    import subprocess
    code = r"""
import sys
sys.path.insert(0, %r)
from pram_amd import _lib
L = _lib.load()
helpers = {"pram_hip_version", "pram_last_error", "pram_fill_u32", "pram_linear_x3_ssq_parts", "pram_attention_x3_is_split",
           "pram_attention_x3_mfma_per_tile", "pram_attention_x3_set_chunk_keys", "pram_attention_x3_set_p_split",
           "pram_attention_x3_set_split_target", "pram_x3_set_act_scale"}
for name, (res, args) in sorted(_lib._SIGS.items()):
    vals = [None if a is _lib.P else (0.0 if a is _lib.F else 0) for a in args]
    r = getattr(L, name)(*vals)
    if name in helpers or name.endswith("_workspace_bytes"):
        continue
    assert isinstance(r, int) and r < 0, (name, r)
    assert L.pram_last_error(), name
print("ok", len(_lib._SIGS))
""" % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_activation_scale_setter_is_validated_and_thread_local():
    """pram_x3_set_act_scale: powers of two in [2^-12, 16] only, query with 0, per host thread (no GPU needed: nothing launches)."""
    import threading
    from pram_amd import _lib, ops
    L = _lib.load()
    assert L.pram_x3_set_act_scale(0.0) == 16.0
    assert L.pram_x3_set_act_scale(3.0) == -1.0 and L.pram_x3_set_act_scale(32.0) == -1.0 and L.pram_x3_set_act_scale(2.0 ** -13) == -1.0
    assert L.pram_x3_set_act_scale(0.0) == 16.0                       # refused values change nothing
    seen = {}

    def other():
        seen["before"] = L.pram_x3_set_act_scale(0.0)
        L.pram_x3_set_act_scale(0.25)
        seen["after"] = L.pram_x3_set_act_scale(0.0)

    with ops.act_scale_scope(1.0):
        assert ops.current_act_scale() == 1.0
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert ops.current_act_scale() == 1.0                         # the other thread's setting is its own
    assert seen == {"before": 16.0, "after": 0.25} and ops.current_act_scale() == 16.0
    with pytest.raises(_lib.PramHipError):
        with ops.act_scale_scope(5.0):
            pass
