"""Shared attention-block runners for SegNetViT / GML / AdaGML on the HIP kernels.

Tokens travel as one contiguous fp32 matrix [S*T, 256] (S sequences padded to T rows) plus an
optional int32 device array ``lens`` [S]; nothing here synchronises with the host.

Weight packing (done once per device, cached by the owning module):
  * ``qkv``: the reference views Linear(256->768) output as (head 4, dim 64, {q,k,v})
    (nets/segnetvit.py:98-100), i.e. row c = h*192 + d*3 + s.  Rows are re-ordered to
    [q | k | v] x [head] x [dim], and inside every q/k head the 32 even rotary dims come first,
    then the 32 odd ones.  q·k is invariant under a common permutation of the head dim, and the
    rotary pair (2i, 2i+1) lands in columns (i, i+32) of the same 64-wide tile, which the GEMM
    epilogue rotates lane-locally (PRAM_LIN_ROTARY).
  * cross attention: ``to_qk`` and ``to_v`` are stacked into one [512,256] projection; the
    reference's two dh^-1/4 factors (nets/gml.py:174) are applied as one dh^-1/2 score scale.
  * ``proj`` is folded into ``mlp.0`` (inference-time weight folding, like the BatchNorm fold):
    mlp.0(cat[x, proj(ctx)]) = W0a x + (W0b Wp) ctx + (b0 + W0b bp), products formed once in fp64.
    One 256x256 GEMM and one [tokens,256] round trip per block disappear; results move by ~1e-7
    relative (covered by the same parity tests).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops

HEADS = 4
DH = 64


def _rot_perm() -> torch.Tensor:
    d = torch.arange(DH)
    return torch.cat([d[0::2], d[1::2]])


def _fold_proj_into_mlp0(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    """-> {'mlp0_w' [2h, 2h], 'mlp0_b' [2h]} acting on cat[x, ctx] (ctx = attention output, before proj)."""
    w0 = sd[prefix + ".mlp.0.weight"].detach().double().cpu()
    b0 = sd[prefix + ".mlp.0.bias"].detach().double().cpu()
    wp = sd[prefix + ".proj.weight"].detach().double().cpu()
    bp = sd[prefix + ".proj.bias"].detach().double().cpu()
    hid = wp.shape[0]
    w0a, w0b = w0[:, :w0.shape[1] - hid], w0[:, w0.shape[1] - hid:]
    w, b = torch.cat([w0a, w0b @ wp], 1), b0 + w0b @ bp
    # mlp.0 feeds a LayerNorm, which is invariant to a common shift of its inputs: the weights are centred over the outputs
    # (fp64), so mlp.0's output is h - mean(h) on every path — and on the split-fp16 path the LayerNorm + GELU ride inside the
    # second GEMM's staging (ops.mlp_tail) instead of being a pass of their own over the hidden layer
    return {"mlp0_w": (w - w.mean(0, keepdim=True)).float(), "mlp0_b": (b - b.mean()).float()}


def pack_self_block(sd: Dict[str, torch.Tensor], prefix: str, device) -> Dict[str, torch.Tensor]:
    w = sd[prefix + ".qkv.weight"].detach().float()
    b = sd[prefix + ".qkv.bias"].detach().float()
    hid = w.shape[0] // 3
    w = w.view(HEADS, DH, 3, -1)
    b = b.view(HEADS, DH, 3)
    perm = _rot_perm()
    wq, wk, wv = w[:, perm, 0], w[:, perm, 1], w[:, :, 2]
    bq, bk, bv = b[:, perm, 0], b[:, perm, 1], b[:, :, 2]
    wp = torch.cat([wq.reshape(hid, -1), wk.reshape(hid, -1), wv.reshape(hid, -1)], 0)
    bp = torch.cat([bq.reshape(hid), bk.reshape(hid), bv.reshape(hid)], 0)
    out = {"qkv_w": wp, "qkv_b": bp, **_fold_proj_into_mlp0(sd, prefix)}
    for k in ("mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias"):
        out[k] = sd[f"{prefix}.{k}"].detach().float()
    return {k: v.contiguous().to(device) for k, v in out.items()}


def pack_cross_block(sd: Dict[str, torch.Tensor], prefix: str, device) -> Dict[str, torch.Tensor]:
    out = {
        "qkv_w": torch.cat([sd[prefix + ".to_qk.weight"], sd[prefix + ".to_v.weight"]], 0).detach().float(),
        "qkv_b": torch.cat([sd[prefix + ".to_qk.bias"], sd[prefix + ".to_v.bias"]], 0).detach().float(),
    }
    out.update(_fold_proj_into_mlp0(sd, prefix))
    for k in ("mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias"):
        out[k] = sd[f"{prefix}.{k}"].detach().float()
    return {k: v.contiguous().to(device) for k, v in out.items()}


def _half_path() -> bool:
    """both the token GEMMs and the attention run on the fp16 MFMA path (BASELINE C5): q / k / v can travel as fp16"""
    return ops.gemm_prec() == "f16" and ops.attn_prec() == "f16"


import os as _os
FUSED_MLP = _os.environ.get("PRAM_FUSED_MLP", "1") != "0"    # LayerNorm + GELU inside the second GEMM of the MLP tail (0: three kernels)
# fp16 path: q / k / v, the attention context and the MLP's hidden layer travel as fp16 in HBM (pram_linear_f16_qkv_h16,
# pram_attention_h16t_h16, pram_linear_f16_ssq_h16, pram_linear_f16_lngelu_f32).  OFF by default: same accuracy class, 45 % fewer
# bytes per block — and 3 % SLOWER at the C5 shape (347.6 vs 359.1 q/s): the 128 x 128 fp16 GEMM is bound by the latency of its
# register staging, not by bytes, and the LayerNorm + GELU transform lands on its critical path.
F16_ACT = _os.environ.get("PRAM_F16_ACT", "0") == "1"
FUSED_VT = _os.environ.get("PRAM_FUSED_VT", "1") != "0"      # the projection epilogue writes the V^T planes (0: separate transpose kernel)


def _split_path() -> bool:
    """both families on the split-fp16 path: the projection hands q / k / v over as (hi, lo) fp16 planes"""
    return ops.gemm_prec() == "x3" and ops.attn_prec() == "x3"


def _cols(planes, lo: int, hi: int):
    return planes[0][:, lo:hi], planes[1][:, lo:hi]


def _mlp_tail(x: torch.Tensor, ctx: torch.Tensor, p: Dict[str, torch.Tensor], lens=None, T: int = 0) -> torch.Tensor:
    """x + mlp(cat[x, proj(ctx)])  (nets/segnetvit.py:104-106), proj folded into mlp.0.  lens / T: ragged sequences — tiles of
    rows beyond every sequence's length are skipped (their rows are never read downstream)."""
    if ops.gemm_prec() == "x3" and FUSED_MLP:
        return ops.mlp_tail(x, p["mlp0_w"], p["mlp0_b"], p["mlp.1.weight"], p["mlp.1.bias"], p["mlp.3.weight"], p["mlp.3.bias"],
                            x2=ctx, residual=x, lens=lens, t_pad=T)
    h = ops.linear(x, p["mlp0_w"], p["mlp0_b"], x2=ctx, lens=lens, t_pad=T)
    ops.layernorm_gelu_(h, p["mlp.1.weight"], p["mlp.1.bias"], lens=lens, t_pad=T)
    return ops.linear(h, p["mlp.3.weight"], p["mlp.3.bias"], residual=x, lens=lens, t_pad=T)


def self_block(x: torch.Tensor, p: Dict[str, torch.Tensor], cos: torch.Tensor, sin: torch.Tensor, S: int, T: int,
               lens: Optional[torch.Tensor], want_colmean: bool = False):
    """x [S*T, 256] -> same.  SelfMultiHeadAttention.forward (nets/segnetvit.py:97-106)."""
    hid = HEADS * DH
    scale = DH ** -0.5
    if _split_path():
        # split-fp16 path: the projection writes q | k | v as (hi, lo) planes (plus fp32 when the column means need q / k)
        if T % 64 == 0 and FUSED_VT:       # the projection writes the values transposed itself
            pl, v3 = ops.linear_qkv_planes(x, p["qkv_w"], p["qkv_b"], HEADS, T, rotary=(cos, sin, 2 * HEADS * DH), lens=lens)
        else:
            _, pl = ops.linear(x, p["qkv_w"], p["qkv_b"], rotary=(cos, sin, 2 * HEADS * DH), split_out="only", lens=lens, t_pad=T)
            v3 = ops.value_planes_t(_cols(pl, 2 * hid, 3 * hid), S, HEADS, T, lens)
        q3, k3 = _cols(pl, 0, hid), _cols(pl, hid, 2 * hid)
        if want_colmean:
            ctx, lse = ops.attention_x3(q3, k3, v3, S, HEADS, T, T, scale, lens, lens, want_lse=True)
            col = ops.attention_colmean_x3(q3, k3, lse, S, HEADS, T, T, scale, lens, lens)
            return _mlp_tail(x, ctx, p, lens, T), col
        return _mlp_tail(x, ops.attention_x3(q3, k3, v3, S, HEADS, T, T, scale, lens, lens), p, lens, T)
    if _half_path() and not want_colmean:
        # fp16 path: the projection writes q | k | v as fp16 only (what the fp16 attention would round them to anyway)
        if F16_ACT and T % 64 == 0:
            # fp16 intermediates in HBM: the projection writes q | k as fp16 rows and v transposed, the attention writes an fp16
            # context, the MLP's hidden layer travels as fp16 and is normalised inside the second GEMM (ops.mlp_tail_f16)
            h16, vt = ops.linear_qkv_h16(x, p["qkv_w"], p["qkv_b"], HEADS, T, rotary=(cos, sin, 2 * HEADS * DH), lens=lens)
            ctx16 = ops.attention_h16t(h16[:, :hid], h16[:, hid:2 * hid], vt, S, HEADS, T, T, scale, lens, lens, out16=True)
            return ops.mlp_tail_f16(x, ctx16, p["mlp0_w"], p["mlp0_b"], p["mlp.1.weight"], p["mlp.1.bias"], p["mlp.3.weight"], p["mlp.3.bias"])
        _, h16 = ops.linear(x, p["qkv_w"], p["qkv_b"], rotary=(cos, sin, 2 * HEADS * DH), half_copy="only")
        vt = ops.value_t16(h16[:, 2 * hid:], S, HEADS, T, lens)
        ctx = ops.attention_h16t(h16[:, :hid], h16[:, hid:2 * hid], vt, S, HEADS, T, T, scale, lens, lens)
        return _mlp_tail(x, ctx, p, lens, T)
    qkv = ops.linear(x, p["qkv_w"], p["qkv_b"], rotary=(cos, sin, 2 * HEADS * DH), lens=lens, t_pad=T)
    q, k, v = qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:]
    if want_colmean:
        ctx, lse = ops.attention(q, k, v, S, HEADS, T, T, scale, lens, lens, want_lse=True)
        col = ops.attention_colmean(q, k, lse, S, HEADS, T, T, scale, lens, lens)
        return _mlp_tail(x, ctx, p, lens, T), col
    ctx = ops.attention(q, k, v, S, HEADS, T, T, scale, lens, lens)
    return _mlp_tail(x, ctx, p, lens, T)


def cross_block(x: torch.Tensor, p: Dict[str, torch.Tensor], B: int, T: int, lens: Optional[torch.Tensor],
                want_colmean: bool = False):
    """x [2B*T, 256]: sequences 0..B-1 are set 0, B..2B-1 set 1.  CrossMultiHeadAttention.forward
    (nets/gml.py:164-186): m0 = softmax_row(sim) v1, m1 = softmax_row(sim^T) v0."""
    hid = HEADS * DH
    scale = DH ** -0.5     # (dh^-1/4)^2
    if _split_path():
        if T % 64 == 0 and FUSED_VT:
            pl, v3 = ops.linear_qkv_planes(x, p["qkv_w"], p["qkv_b"], HEADS, T, lens=lens)
        else:
            _, pl = ops.linear(x, p["qkv_w"], p["qkv_b"], split_out="only", lens=lens, t_pad=T)
            v3 = ops.value_planes_t(_cols(pl, hid, 2 * hid), 2 * B, HEADS, T, lens)
        qk3 = _cols(pl, 0, hid)
        if want_colmean:
            ctx, lse = ops.attention_x3(qk3, qk3, v3, 2 * B, HEADS, T, T, scale, lens, lens, want_lse=True, kv_shift=B)
            col = ops.attention_colmean_x3(qk3, qk3, lse, 2 * B, HEADS, T, T, scale, lens, lens, kv_shift=B)
            return _mlp_tail(x, ctx, p, lens, T), col
        return _mlp_tail(x, ops.attention_x3(qk3, qk3, v3, 2 * B, HEADS, T, T, scale, lens, lens, kv_shift=B), p, lens, T)
    if _half_path() and not want_colmean:
        if F16_ACT and T % 64 == 0:
            qk16, vt = ops.linear_qkv_h16(x, p["qkv_w"], p["qkv_b"], HEADS, T, lens=lens)
            ctx16 = ops.attention_h16t(qk16, qk16, vt, 2 * B, HEADS, T, T, scale, lens, lens, kv_shift=B, out16=True)
            return ops.mlp_tail_f16(x, ctx16, p["mlp0_w"], p["mlp0_b"], p["mlp.1.weight"], p["mlp.1.bias"], p["mlp.3.weight"], p["mlp.3.bias"])
        _, h16 = ops.linear(x, p["qkv_w"], p["qkv_b"], half_copy="only")
        qk16, v16 = h16[:, :hid], h16[:, hid:]
        vt = ops.value_t16(v16, 2 * B, HEADS, T, lens)
        ctx = ops.attention_h16t(qk16, qk16, vt, 2 * B, HEADS, T, T, scale, lens, lens, kv_shift=B)
        return _mlp_tail(x, ctx, p, lens, T)
    qkv = ops.linear(x, p["qkv_w"], p["qkv_b"], lens=lens, t_pad=T)          # [2B*T, 512] = [qk | v]
    qk, v = qkv[:, :hid], qkv[:, hid:]
    # one launch for both directions: sequence s attends to sequence (s + B) mod 2B
    if want_colmean:
        ctx, lse = ops.attention_cross(qk, v, B, HEADS, T, scale, lens, want_lse=True)
        col = ops.attention_cross_colmean(qk, lse, B, HEADS, T, scale, lens)
        # rows 0..B-1: attn10 column means -> per set-0 token ; rows B..2B-1: attn01 column means -> per set-1 token
        return _mlp_tail(x, ctx, p, lens, T), col
    ctx = ops.attention_cross(qk, v, B, HEADS, T, scale, lens)
    return _mlp_tail(x, ctx, p, lens, T)


def with_model_precision(fn):
    """Run a model method under the model's own ``precision`` (None = the process default, ops.*_precision)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with ops.precision_scope(getattr(self, "precision", None)):
            # range guard of the split-fp16 path (ops.guarded_call): a value the planes cannot carry first lowers this model's
            # activation scale (self.act_scale, kept) and re-runs; beyond every scale the call re-runs on the exact-fp32 kernels
            # (or raises) instead of coming back as NaN-derived indices
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                return fn(self, *a, **k)      # raises the "no CPU path" error itself

            def call():
                ops.note_model_ran(self)
                with ops.act_scale_scope(getattr(self, "act_scale", None)):      # read at every (re-)run: the guard may have lowered it
                    return fn(self, *a, **k)
            return ops.guarded_call(call, dev)
    return wrapper


class PackedCache:
    """Mixin: device-side packed weights, rebuilt after load_state_dict / .to() / .cuda(); per-model MFMA path."""
    precision: Optional[str] = None      # "f32" | "x3" | "f16"; None follows ops.gemm_prec() / ops.attn_prec()
    # split-fp16 path: the activation planes of this model carry value * act_scale (ops: "activation scale").  16 covers
    # |x| < 4094.97; the range guard divides it by 16 when an activation does not fit and the model keeps the result — or set it
    # up front with set_act_scale() / calibrate_act_scale() for a checkpoint whose range is known.
    act_scale: float = 16.0

    def set_act_scale(self, s: float):
        s = float(s)
        import math
        if not (s > 0 and math.log2(s) == int(math.log2(s)) and 2.0 ** -12 <= s <= 16.0):
            raise ValueError(f"act_scale {s!r}: expected a power of two in [2^-12, 16]")
        self.act_scale = s
        return self

    def calibrate_act_scale(self, run):
        """Settle the activation scale on representative inputs BEFORE serving (or capturing a hipGraph): `run` is a callable that
        pushes a calibration sample through this model's guarded entry point (``lambda: model(data)``, ``lambda:
        matcher.produce_matches(pair)``).  Starts from the default 16, lets the range guard lower it as far as the sample needs
        (policy "fallback") and returns the scale the model keeps.  A sample inside the default range costs one forward."""
        self.act_scale = ops.ACT_SCALE_DEFAULT
        with ops.guard_scope("fallback"):
            run()
        return self.act_scale

    def set_precision(self, p: Optional[str]):
        self.precision = None if p is None else ops._check_precision(p)
        return self

    def _packed_get(self, builder):
        dev = next(self.parameters()).device
        # the version counters catch in-place edits (optimizer steps, p.mul_() under no_grad); writes through `.data`
        # bypass them — call refresh_packed() after those
        ver = 0
        for t in self.parameters():
            ver += t._version
        for t in self.buffers():
            ver += t._version
        key = (str(dev), ver)
        cache = self.__dict__.setdefault("_packed_store", {})
        if key not in cache:
            cache.clear()
            with torch.no_grad():
                cache[key] = builder(dev)
        return cache[key]

    def refresh_packed(self):
        """Drop the packed / split device copies of the weights; the next forward rebuilds them from the parameters."""
        self._packed_invalidate()
        return self

    def _packed_invalidate(self):
        self.__dict__.setdefault("_packed_store", {}).clear()      # derived forms (fp16 / split planes) die with the packed tensors

    def _apply(self, fn, *a, **k):  # .to/.cuda/.float
        r = super()._apply(fn, *a, **k)
        self._packed_invalidate()
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed_invalidate()
        return r


def require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        from .._lib import PramHipError
        raise PramHipError(f"{what}: tensors must live on the GPU — pram_amd has no CPU path "
                           "(the CPU restatement under oracle/ is test infrastructure only)")
