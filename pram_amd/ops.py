"""Torch-facing wrappers over the C ABI (include/pram_hip.h).  PyTorch here is plumbing only:
device memory (torch.empty), the current HIP stream, and nothing else — every op below is a
hand-written gfx950 kernel in pram_amd/csrc.  All tensors must be fp32 CUDA tensors."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_INT64 = torch.int64


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.PramHipError(f"{name}: expected a CUDA tensor (pram_amd has no CPU path)")
    if t.dtype != dtype:
        raise _lib.PramHipError(f"{name}: expected {dtype}, got {t.dtype}")


def _rows2d(t: torch.Tensor, name: str) -> Tuple[int, int]:
    """(rows, ld) of a tensor viewed as a row-major matrix over its last dim."""
    _chk(t, name)
    if t.stride(-1) != 1:
        raise _lib.PramHipError(f"{name}: last dim must be contiguous")
    return t.numel() // t.shape[-1], t.shape[-1]


def _filled(shape, device, dtype=torch.float32, word: int = 0) -> torch.Tensor:
    """torch.zeros / torch.full for 32-bit dtypes without a framework kernel: torch.empty + hipMemsetD32Async on the current stream
    (``word`` = the 32-bit pattern; -2 as int32 is 0xFFFFFFFE)."""
    t = torch.empty(shape, device=device, dtype=dtype)
    assert t.element_size() == 4
    if t.numel():
        _lib.check(_lib.load().pram_fill_u32(t.data_ptr(), word & 0xFFFFFFFF, t.numel(), _st()), "pram_fill_u32")
    return t


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, x2: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, alpha: float = 1.0, out: Optional[torch.Tensor] = None,
           rotary: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, half_copy: str = "no", split_out: str = "no",
           precision: Optional[str] = None, lens: Optional[torch.Tensor] = None, t_pad: int = 0, out_planes=None):
    """out = alpha * ([x | x2] @ w.T + bias) + residual.  x [..., k0] (contiguous rows), w [n, k0+k1].
    precision: None = ops.gemm_precision; "f32" exact-fp32 MFMA, "x3" split-fp16 (three fp16 MFMAs per product, fp32-class
    accuracy), "f16" single fp16 product (BASELINE C5).  Shapes a fast path cannot take (K not a multiple of 32 / 64)
    run on the exact-fp32 kernel.
    half_copy (f16 path only): "also" -> returns (out fp32, out fp16), "only" -> returns (None, out fp16): the fp16 operand
    of attention_h16.
    split_out (x3 path only): "also" -> (out fp32, (hi, lo)), "only" -> (None, (hi, lo)): the result * 16 as two fp16 planes,
    the operand format of attention_x3.
    lens / t_pad (every path; with half_copy the fp16 copy is dense): ragged token matrix — rows are sequences of t_pad rows with lens[s] valid ones; output tiles
    without a valid row are skipped and left untouched (and only valid rows are stored).
    out_planes (with split_out): a (hi, lo) pair of existing fp16 buffers to write into instead of fresh ones."""
    L = _lib.load()
    x = x.contiguous()
    m, k0 = _rows2d(x, "x")
    if lens is not None:
        assert t_pad > 0 and m % t_pad == 0 and lens.dtype == torch.int32 and lens.numel() == m // t_pad
    k1 = 0
    if x2 is not None:
        x2 = x2.contiguous()
        m2, k1 = _rows2d(x2, "x2")
        assert m2 == m
    _chk(w, "w")
    w = w.contiguous()
    n = w.shape[0]
    assert w.shape[1] == k0 + k1, (w.shape, k0, k1)
    prec = _check_precision(_tl("forced") or precision or gemm_prec())
    want32 = not (half_copy == "only" or split_out == "only")
    if out is None and want32:
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = residual.contiguous()
    flags, rc, rs, rcols = 0, None, None, 0
    if rotary is not None:
        rc, rs, rcols = rotary
        flags = 1
    K = k0 + k1
    use16 = prec == "f16" and K % 64 == 0 and (k1 == 0 or k0 % 64 == 0)
    x3_shape = K % 32 == 0 and (k1 == 0 or k0 % 32 == 0)
    usex3 = prec == "x3" and x3_shape
    if half_copy != "no":
        if not use16:
            raise _lib.PramHipError("linear(half_copy=...) needs the fp16 GEMM path (precision 'f16', K % 64 == 0)")
        out16 = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float16)
        o32 = out if half_copy == "also" else None
        if m:
            _lib.check(L.pram_linear_f16_h16(_p(x), k0, k0, _p(x2), k1, k1, _p(_w16(w)), _p(bias), _p(residual), n, _p(o32), n,
                                             _p(out16), n, m, n, float(alpha), flags, _p(rc), _p(rs), int(rcols), _st()),
                       "pram_linear_f16_h16")
        return o32, out16
    if split_out != "no":
        if not usex3:
            raise _lib.PramHipError("linear(split_out=...) needs the split-fp16 GEMM path (precision 'x3', K % 32 == 0)")
        if out_planes is not None:      # persistent plane buffers (AdaGML commits the rows of the pairs stopping at a layer)
            planes = out_planes
            assert planes[0].dtype == torch.float16 and planes[0].is_contiguous() and planes[1].is_contiguous() and planes[0].shape[-1] == n
        else:
            planes = torch.empty(2, *x.shape[:-1], n, device=x.device, dtype=torch.float16)
        o32 = out if split_out == "also" else None
        if m:
            wh, wl, ws = split_weight(w)
            _lib.check(L.pram_linear_x3_ragged_f32(_p(x), k0, k0, _p(x2), k1, k1, _p(wh), _p(wl), ws, _p(bias), _p(residual), n,
                                                   _p(o32), n, _p(planes[0]), _p(planes[1]), n, m, n, float(alpha), flags, _p(rc), _p(rs),
                                                   int(rcols), _p(lens), int(t_pad), _st()), "pram_linear_x3_f32")
        return o32, (planes[0], planes[1])
    if m == 0:          # empty token set: nothing to launch (an empty tensor has a null data pointer)
        return out
    if usex3:
        wh, wl, ws = split_weight(w)
        _lib.check(L.pram_linear_x3_ragged_f32(_p(x), k0, k0, _p(x2), k1, k1, _p(wh), _p(wl), ws, _p(bias), _p(residual), n, _p(out), n,
                                               None, None, 0, m, n, float(alpha), flags, _p(rc), _p(rs), int(rcols), _p(lens), int(t_pad),
                                               _st()), "pram_linear_x3_f32")
        return out
    if use16:      # ragged calls included: tiles without a valid row are skipped, only valid rows are stored (as on the other paths)
        _lib.check(L.pram_linear_f16_ragged_f32(_p(x), k0, k0, _p(x2), k1, k1, _p(_w16(w)), _p(bias), _p(residual),
                                                n, _p(out), n, m, n, float(alpha), flags, _p(rc), _p(rs), int(rcols), _p(lens), int(t_pad), _st()),
                   "pram_linear_f16_f32")
        return out
    _lib.check(L.pram_linear_ragged_f32(_p(x), k0, k0, _p(x2), k1, k1, _p(w), _p(bias), _p(residual),
                                        n, _p(out), n, m, n, float(alpha), flags, _p(rc), _p(rs), int(rcols), _p(lens), int(t_pad), _st()),
               "pram_linear_f32")
    return out


def mlp_tail(x: torch.Tensor, w0c: torch.Tensor, b0c: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w3: torch.Tensor,
             b3: Optional[torch.Tensor], *, x2: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             eps: float = 1e-5, lens: Optional[torch.Tensor] = None, t_pad: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """residual + Linear(w3, b3)(GELU(LayerNorm(Linear(w0, b0)([x | x2])))) on the split-fp16 path as TWO kernels: the first GEMM
    (w0c / b0c = the weights centred over their outputs, center_linear()) writes h - mean(h) and the rows' sums of squares, the
    second normalises and GELUs its operand while it stages it (pram_linear_x3_ssq_f32 / pram_linear_x3_lngelu_f32).  Shapes the
    fast path cannot take (K % 32 != 0, hidden > 1024) raise."""
    L = _lib.load()
    x = x.contiguous()
    m, k0 = _rows2d(x, "x")
    k1 = 0
    if x2 is not None:
        x2 = x2.contiguous()
        m2, k1 = _rows2d(x2, "x2")
        assert m2 == m
    hid = w0c.shape[0]
    n = w3.shape[0]
    if (k0 + k1) % 32 or (k1 and k0 % 32) or hid % 32 or hid > 1024 or w0c.shape[1] != k0 + k1 or w3.shape[1] != hid:
        raise _lib.PramHipError(f"mlp_tail: unsupported shapes K={k0}+{k1}, hidden={hid}")
    if lens is not None:
        assert t_pad > 0 and m % t_pad == 0 and lens.dtype == torch.int32 and lens.numel() == m // t_pad
    if out is None:
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    if m == 0:
        return out
    parts = int(L.pram_linear_x3_ssq_parts(m, hid, k0 + k1))
    h = torch.empty(m, hid, device=x.device, dtype=torch.float32)
    ssq = torch.empty(parts, m, device=x.device, dtype=torch.float32)
    wh, wl, ws = split_weight(w0c.contiguous())
    _lib.check(L.pram_linear_x3_ssq_f32(_p(x), k0, k0, _p(x2), k1, k1, _p(wh), _p(wl), ws, _p(b0c), _p(h), hid, _p(ssq), m, hid,
                                        _p(lens), int(t_pad), _st()), "pram_linear_x3_ssq_f32")
    if residual is not None:
        residual = residual.contiguous()
    wh, wl, ws = split_weight(w3.contiguous())
    _lib.check(L.pram_linear_x3_lngelu_f32(_p(h), hid, hid, _p(wh), _p(wl), ws, _p(b3), _p(residual), n, _p(out), n, m, n, _p(ssq), parts,
                                           _p(gamma), _p(beta), float(eps), _p(lens), int(t_pad), _st()), "pram_linear_x3_lngelu_f32")
    return out


def linear_qkv_h16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], heads: int, t_seq: int,
                   rotary: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, lens: Optional[torch.Tensor] = None):
    """The q | k | v (or qk | v) projection on the fp16 path with fp16 outputs: -> (fp16 [m, n - heads * 64] q / k columns,
    vt16 = the transposed key-permuted fp16 values [m / t_seq, heads, 64, t_seq] of attention_h16t).  t_seq % 64 == 0."""
    L = _lib.load()
    x = x.contiguous()
    m, k0 = _rows2d(x, "x")
    n = w.shape[0]
    col0 = n - heads * 64
    assert t_seq % 64 == 0 and m % t_seq == 0 and k0 % 64 == 0 and col0 > 0 and col0 % 64 == 0
    out16 = torch.empty(m, col0, device=x.device, dtype=torch.float16)
    vt = torch.empty(m // t_seq, heads, 64, t_seq, device=x.device, dtype=torch.float16)
    flags, rc, rs, rcols = 0, None, None, 0
    if rotary is not None:
        rc, rs, rcols = rotary
        flags = 1
    if m:
        _lib.check(L.pram_linear_f16_qkv_h16(_p(x), k0, k0, _p(_w16(w.contiguous())), _p(bias), _p(out16), col0, _p(vt), col0, heads, t_seq, m, n,
                                             flags, _p(rc), _p(rs), int(rcols), _p(lens), _st()), "pram_linear_f16_qkv_h16")
    return out16, vt


def mlp_tail_f16(x: torch.Tensor, ctx16: torch.Tensor, w0c: torch.Tensor, b0c: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                 w3: torch.Tensor, b3: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """x + Linear(w3)(GELU(LayerNorm(Linear(w0c)([x | ctx])))) on the fp16 MFMA path with fp16 intermediates: x fp32 [m, k0] (the
    residual stream), ctx16 fp16 [m, k1] (attention_h16t(out16=True)); the hidden layer travels as fp16."""
    L = _lib.load()
    x = x.contiguous()
    m, k0 = _rows2d(x, "x")
    assert ctx16.dtype == torch.float16 and ctx16.is_contiguous() and ctx16.shape[0] == m
    k1 = ctx16.shape[1]
    hid, n = w0c.shape[0], w3.shape[0]
    assert k0 % 64 == 0 and k1 % 64 == 0 and hid % 64 == 0 and hid <= 1024 and n > 64 and w0c.shape[1] == k0 + k1 and w3.shape[1] == hid and n == k0
    out = torch.empty(m, n, device=x.device, dtype=torch.float32)
    if m == 0:
        return out
    parts = (hid + 63) // 64
    h16 = torch.empty(m, hid, device=x.device, dtype=torch.float16)
    ssq = torch.empty(parts, m, device=x.device, dtype=torch.float32)
    _lib.check(L.pram_linear_f16_ssq_h16(_p(x), k0, k0, _p(ctx16), k1, k1, _p(_w16(w0c.contiguous())), _p(b0c), _p(h16), hid, _p(ssq), m, hid, _st()),
               "pram_linear_f16_ssq_h16")
    _lib.check(L.pram_linear_f16_lngelu_f32(_p(h16), hid, hid, _p(_w16(w3.contiguous())), _p(b3), _p(x), n, _p(out), n, m, n, _p(ssq), parts,
                                            _p(gamma), _p(beta), float(eps), _st()), "pram_linear_f16_lngelu_f32")
    return out


def center_linear(w: torch.Tensor, b: torch.Tensor):
    """(w - mean over the outputs, b - mean) in fp64 -> fp32: a Linear whose output is h - mean(h).  LayerNorm(h) only needs the
    centred values (it is invariant to the shift), so a Linear that feeds a LayerNorm can be centred once at pack time."""
    wd, bd = w.detach().double().cpu(), b.detach().double().cpu()
    return (wd - wd.mean(0, keepdim=True)).float().contiguous(), (bd - bd.mean()).float().contiguous()


def linear_qkv_planes(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], heads: int, t_seq: int,
                      rotary: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, lens: Optional[torch.Tensor] = None):
    """The q | k | v (or qk | v) projection of an attention block on the split-fp16 path, values written transposed:
    -> ((hi, lo) row-major planes [m, n - heads * 64] of the q / k columns, (vt_hi, vt_lo) = value_planes_t of the value columns).
    Rows are sequences of t_seq tokens (t_seq % 64 == 0); lens as in linear()."""
    L = _lib.load()
    x = x.contiguous()
    m, k0 = _rows2d(x, "x")
    n = w.shape[0]
    col0 = n - heads * 64
    assert t_seq % 64 == 0 and m % t_seq == 0 and k0 % 32 == 0 and col0 > 0 and col0 % 64 == 0
    planes = torch.empty(2, m, col0, device=x.device, dtype=torch.float16)
    vt = torch.empty(2, m // t_seq, heads, 64, t_seq, device=x.device, dtype=torch.float16)
    flags, rc, rs, rcols = 0, None, None, 0
    if rotary is not None:
        rc, rs, rcols = rotary
        flags = 1
    if m:
        wh, wl, ws = split_weight(w.contiguous())
        _lib.check(L.pram_linear_x3_qkv_f32(_p(x), k0, k0, _p(wh), _p(wl), ws, _p(bias), _p(planes[0]), _p(planes[1]), col0, _p(vt[0]), _p(vt[1]),
                                            col0, heads, t_seq, m, n, flags, _p(rc), _p(rs), int(rcols), _p(lens), _st()), "pram_linear_x3_qkv_f32")
    return (planes[0], planes[1]), (vt[0], vt[1])


def linear_planes(x, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, x2=None, residual: Optional[torch.Tensor] = None,
                  alpha: float = 1.0, rotary: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None, out: str = "f32"):
    """linear() on the split-fp16 path with the activations ALREADY split: x (and x2) are (hi, lo) pairs of fp16 2-D views
    (value * s = hi + lo, s = the activation scale in force: 16 unless a model / act_scale_scope lowered it) written by a producing kernel's epilogue.  out: "f32" -> fp32 tensor; "planes" -> (hi, lo) planes of
    the result * s; "both" -> (fp32, (hi, lo))."""
    L = _lib.load()
    xh, xl = x
    assert xh.dtype == torch.float16 and xh.dim() == 2 and xh.stride(1) == 1 and xh.stride(0) == xl.stride(0)
    m, k0 = xh.shape
    k1 = 0
    if x2 is not None:
        assert x2[0].shape[0] == m and x2[0].stride(1) == 1 and x2[0].stride(0) == x2[1].stride(0)
        k1 = x2[0].shape[1]
    w = w.contiguous()
    n = w.shape[0]
    assert w.shape[1] == k0 + k1 and k0 % 32 == 0 and k1 % 32 == 0
    o32 = torch.empty(m, n, device=xh.device, dtype=torch.float32) if out in ("f32", "both") else None
    planes = torch.empty(2, m, n, device=xh.device, dtype=torch.float16) if out in ("planes", "both") else None
    flags, rc, rs, rcols = 0, None, None, 0
    if rotary is not None:
        rc, rs, rcols = rotary
        flags = 1
    if residual is not None:
        residual = residual.contiguous()
    if m:
        wh, wl, ws = split_weight(w)
        _lib.check(L.pram_linear_x3p_f32(_p(xh), _p(xl), xh.stride(0), k0, _p(x2[0]) if k1 else None, _p(x2[1]) if k1 else None,
                                         x2[0].stride(0) if k1 else 0, k1, _p(wh), _p(wl), ws, _p(bias), _p(residual), n, _p(o32), n,
                                         _p(planes[0]) if planes is not None else None, _p(planes[1]) if planes is not None else None,
                                         n, m, n, float(alpha), flags, _p(rc), _p(rs), int(rcols), _st()), "pram_linear_x3p_f32")
    pl = None if planes is None else (planes[0], planes[1])
    return o32 if out == "f32" else (pl if out == "planes" else (o32, pl))


def bgemm_nt(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, ldc: Optional[int] = None) -> torch.Tensor:
    """c[z] = alpha * a[z] @ b[z].T ; a [B,M,K], b [B,N,K] -> c [B,M,ldc] (view [:, :, :N] is the result)."""
    L = _lib.load()
    a, b = a.contiguous(), b.contiguous()
    _chk(a, "a"), _chk(b, "b")
    B, M, K = a.shape
    N = b.shape[1]
    ldc = ldc or N
    c = torch.empty(B, M, ldc, device=a.device, dtype=torch.float32)
    _lib.check(L.pram_bgemm_nt_f32(_p(a), K, M * K, _p(b), K, N * K, _p(c), ldc, M * ldc, B, M, N, K, float(alpha), _st()),
               "pram_bgemm_nt_f32")
    return c


def bgemm_nt_planes(a, b, batch: int, m: int, n: int, alpha: float = 1.0, ldc: Optional[int] = None) -> torch.Tensor:
    """bgemm_nt on the split-fp16 path: a / b are (hi, lo) pairs of fp16 2-D views [batch * m, K] / [batch * n, K] (value * 16 =
    hi + lo, e.g. row ranges of linear(split_out=...)'s planes) -> c [batch, m, ldc] fp32."""
    L = _lib.load()
    for pair in (a, b):
        for t in pair:
            assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
        assert pair[0].stride(0) == pair[1].stride(0)
    K = a[0].shape[1]
    assert b[0].shape[1] == K and b[0].stride(0) == K and a[0].shape[0] == batch * m and b[0].shape[0] == batch * n
    ldc = ldc or n
    c = torch.empty(batch, m, ldc, device=a[0].device, dtype=torch.float32)
    lda = a[0].stride(0)
    _lib.check(L.pram_bgemm_nt_x3p_f32(_p(a[0]), _p(a[1]), lda, m * lda, _p(b[0]), _p(b[1]), K, n * K, _p(c), ldc, m * ldc, batch, m, n, K,
                                       float(alpha), _st()), "pram_bgemm_nt_x3p_f32")
    return c


def layernorm_gelu_(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                    lens: Optional[torch.Tensor] = None, t_pad: int = 0) -> torch.Tensor:
    L = _lib.load()
    rows, cols = _rows2d(x, "x")
    assert x.is_contiguous()
    _lib.check(L.pram_layernorm_gelu_ragged_f32(_p(x), cols, _p(x), cols, _p(gamma), _p(beta), rows, cols, float(eps), _p(lens), int(t_pad),
                                                _st()), "pram_layernorm_gelu_f32")
    return x


def fourier_encoding(kpts: torch.Tensor, wr: torch.Tensor, cx: float, cy: float, scale: float, out=None):
    """kpts [..., 2] -> (cos, sin) [..., 32]; out: a (cos, sin) pair of contiguous [rows, 32] views to write into."""
    L = _lib.load()
    kpts = kpts.contiguous().float()
    _chk(kpts, "kpts")
    rows = kpts.numel() // 2
    if out is not None:
        cos, sin = out
        assert cos.is_contiguous() and sin.is_contiguous() and cos.numel() == rows * 32 and sin.numel() == rows * 32
    else:
        cos = torch.empty(*kpts.shape[:-1], 32, device=kpts.device, dtype=torch.float32)
        sin = torch.empty_like(cos)
    _lib.check(L.pram_fourier_encoding_f32(_p(kpts), _p(wr), float(cx), float(cy), float(scale), _p(cos), _p(sin), rows, _st()),
               "pram_fourier_encoding_f32")
    return cos, sin


# bench.py's roofline probe: when set to a list, every attention launch is bracketed by HIP events on the
# launch stream and (algorithmic flops, start, stop) is appended.  None (default) = no events.
attention_probe = None

# MFMA path of the three matrix families (token GEMMs + convolutions / attention):
#   "f32" exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: the f32 vector rate);
#   "x3"  split-fp16: every fp32 operand = hi + lo in fp16, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate —
#         fp32-class accuracy (passes the same 1e-3 / indices-exact gates) at 5.3x less matrix time;
#   "f16" single fp16 product (BASELINE config C5's "fp16 MFMA path", own documented tolerance).
# Process defaults come from PRAM_PRECISION (both) or PRAM_GEMM_PRECISION / PRAM_ATTENTION_PRECISION; every op also takes
# an explicit ``precision=`` and every model a ``.precision`` attribute that overrides them.
import os as _os
import weakref as _weakref
PRECISIONS = ("f32", "x3", "f16")
default_precision = _os.environ.get("PRAM_PRECISION", "x3")
attention_precision = _os.environ.get("PRAM_ATTENTION_PRECISION", default_precision)
gemm_precision = _os.environ.get("PRAM_GEMM_PRECISION", default_precision)


def _check_precision(p: str) -> str:
    if p not in PRECISIONS:
        raise _lib.PramHipError(f"unknown precision {p!r} (expected one of {PRECISIONS})")
    return p


def set_precision(p: str) -> None:
    """Process-wide default for both the GEMM / convolution family and the attention family (thread-local scopes override it)."""
    global attention_precision, gemm_precision
    attention_precision = gemm_precision = _check_precision(p)


def current_precision() -> str:
    g, a = gemm_prec(), attn_prec()
    return a if a == g else f"gemm {g} / attention {a}"


import threading as _threading

# Scopes (precision_scope / forced_precision / guard_scope) and the guard's nesting depth are PER HOST THREAD: a process that drives
# several GPUs from one thread each must not see another thread's model-level override or nesting depth.  The module attributes
# gemm_precision / attention_precision / x3_guard stay the process-wide DEFAULTS (what set_precision() and the environment set);
# gemm_prec() / attn_prec() / guard_policy() return what is in force for the calling thread.
_tls = _threading.local()


def _tl(name, default=None):
    return getattr(_tls, name, default)


def gemm_prec() -> str:
    """MFMA path of the token GEMMs / convolutions in force for the calling thread (scope override, else the process default)."""
    return _tl("forced") or _tl("gemm") or gemm_precision


def attn_prec() -> str:
    """MFMA path of the attention family in force for the calling thread."""
    return _tl("forced") or _tl("attn") or attention_precision


def guard_policy() -> str:
    return _tl("guard") or x3_guard


class forced_precision:
    """``with ops.forced_precision("f32"): ...`` — every op and every model this thread runs inside the block uses that MFMA
    path, whatever the models' own ``.precision`` says (the range guard's fallback run)."""

    def __init__(self, p: str):
        self.p = _check_precision(p)

    def __enter__(self):
        self.saved = _tl("forced")
        _tls.forced = self.p

    def __exit__(self, *exc):
        _tls.forced = self.saved
        return False


class precision_scope:
    """``with ops.precision_scope("x3"): ...`` — what a model with a ``.precision`` attribute wraps its forward in (None: no
    override).  Thread-local; an enclosing forced_precision wins."""

    def __init__(self, p: Optional[str]):
        self.p = None if p is None else _check_precision(p)

    def __enter__(self):
        self.saved = (_tl("gemm"), _tl("attn"))
        if self.p is not None:
            _tls.gemm = _tls.attn = self.p

    def __exit__(self, *exc):
        _tls.gemm, _tls.attn = self.saved
        return False


# ---- range guard of the split-fp16 path (include/pram_hip.h, "range guard").  Every x3 kernel that splits fp32 values reports
# a finite |value| >= 4094.97 (its hi part would be +-inf in fp16) in a device status word owned here, one per device.  Launches
# are asynchronous, so the word is read where the host synchronises anyway:
#   * the model entry points (forward / produce_matches / extract_*: nets/_blocks.with_model_precision) read it after their last
#     launch and, when it is set, re-run the call on the exact-fp32 kernels (x3_guard = "fallback", the default), raise
#     (x3_guard = "raise"), or leave it to the caller (x3_guard = "deferred": no synchronisation; ask x3_range_exceeded());
#   * pipeline.QueryPipeline does the same once per run; bench.py defers it to the end of the timed region and reports it.
# WHETHER there is anything to read is decided by what was launched, not by the precision settings: every binding that launches a
# splitting kernel marks its device (_mark_x3: all of them fetch their weight planes through split_weight), so a model that runs
# x3 under its own .precision inside an f32 process, or the x3 kernels the fp16 path uses on purpose (GML's matching descriptors),
# arm the guard all the same — and a pure f32 / f16 run never synchronises for it.
x3_guard = _os.environ.get("PRAM_X3_GUARD", "fallback")
X3_GUARDS = ("fallback", "raise", "deferred")
_status_words = {}
_x3_pending = set()       # device indices with split-fp16 launches since their status word was last read
_x3_counts = {}           # device index -> split-fp16 launches so far (x3_launch_count: what a hipGraph capture compares)


def _dev_index(device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def _mark_x3(device) -> None:
    i = _dev_index(device)
    _x3_pending.add(i)
    _x3_counts[i] = _x3_counts.get(i, 0) + 1


def mark_x3(device) -> None:
    """For callers that re-issue captured work (hipGraph replays launch nothing through these bindings): arm the range guard's
    read for ``device`` without counting a launch — x3_launched() is True again until the status word is next read."""
    _x3_pending.add(_dev_index(device))


def x3_launch_count(device=None) -> int:
    """Split-fp16 launches issued (or captured) on ``device`` so far: a caller that replays captured work compares the count
    around the capture to know whether its replays can set the status word."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return _x3_counts.get(_dev_index(device), 0)


def x3_launched(device=None) -> bool:
    """True if a split-fp16 kernel was launched on ``device`` since its status word was last read (no synchronisation)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return _dev_index(device) in _x3_pending


def _x3_status(device) -> torch.Tensor:
    """The status word of ``device`` (allocated and registered with the library on first use)."""
    key = _dev_index(device)
    w = _status_words.get(key)
    if w is None:
        with torch.cuda.device(key):
            w = torch.zeros(4, device=torch.device("cuda", key), dtype=torch.int32)
            _lib.check(_lib.load().pram_set_status_word(w.data_ptr()), "pram_set_status_word")
        _status_words[key] = w
    return w


def x3_range_exceeded(device=None, reset: bool = True) -> bool:
    """True if a split-fp16 kernel met a value beyond the format's range since the last reset (synchronises with the device:
    the kernels that could set the word have to be done)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _x3_status(device)
    torch.cuda.synchronize(device)
    _x3_pending.discard(_dev_index(device))      # everything launched so far has reported
    hit = bool(int(w[0].item()) & 1)
    if hit and reset:
        w.zero_()
    return hit


class guard_scope:
    """``with ops.guard_scope("deferred"): ...`` — the range-guard policy of the model calls this thread makes inside."""

    def __init__(self, mode: str):
        if mode not in X3_GUARDS:
            raise _lib.PramHipError(f"unknown x3 guard {mode!r} (expected one of {X3_GUARDS})")
        self.mode = mode

    def __enter__(self):
        self.saved = _tl("guard")
        _tls.guard = self.mode

    def __exit__(self, *exc):
        _tls.guard = self.saved
        return False


# ---- activation scale of the split-fp16 path (include/pram_hip.h, "activation scale").  The planes carry value * s, s = 16 by
# default (range |x| < 4094.97).  A model whose activations are larger — trained checkpoints — does not have to leave the split
# path: it lowers s (a power of two; s = 1 carries |x| < 65520, s = 1/16 a million) and keeps the same 22-bit significand.  The
# scale is a property of the MODEL (PackedCache.act_scale, applied around its forward by nets/_blocks.with_model_precision) and
# is found by the range guard itself: a tripped guard first lowers the scale of the models that ran (sticky: the cliff is met
# once) and re-runs on the split kernels; only a value no scale can carry goes to the exact-fp32 kernels.
ACT_SCALE_DEFAULT = 16.0
ACT_SCALE_MIN = 2.0 ** -8
guard_events = {"rescaled": 0, "f32_fallback": 0}      # what the range guard did so far in this process (tests / bench read it)


def current_act_scale() -> float:
    return float(_lib.load().pram_x3_set_act_scale(0.0))


class act_scale_scope:
    """``with ops.act_scale_scope(s): ...`` — the split-fp16 launches of this thread inside the block carry their activation planes
    as value * s (None: no change).  Producers and consumers of planes must run under the same scale."""

    def __init__(self, s: Optional[float]):
        self.s = None if s is None else float(s)

    def __enter__(self):
        self.saved = None
        if self.s is not None:
            self.saved = float(_lib.load().pram_x3_set_act_scale(self.s))
            if self.saved < 0:
                raise _lib.PramHipError(f"activation scale {self.s!r}: expected a power of two in [2^-12, 16]")

    def __exit__(self, *exc):
        if self.saved is not None:
            _lib.load().pram_x3_set_act_scale(self.saved)
        return False


def _ran_models():
    """models (PackedCache) whose guarded forward ran inside the current outermost guarded call of this thread"""
    return _tl("ran", None)


def note_model_ran(model) -> None:
    ran = _tl("ran", None)
    if ran is not None and all(m is not model for m in ran):
        ran.append(model)


def guarded_call(fn, device):
    """Run ``fn()`` (a model entry point) under the range guard: only the OUTERMOST guarded call of a thread checks the status
    word, after its last launch, and only if a split-fp16 kernel was launched on the device since the word was last read; nothing
    is checked while a stream is capturing (a hipGraph cannot synchronise: GraphedPipeline checks after the replay).
    A tripped guard ("fallback", the default) first divides the activation scale of every model that ran inside the call by 16
    (down to ACT_SCALE_MIN; the models keep it) and re-runs on the split kernels; when no scale is left to give, or nothing that
    ran has one, the call re-runs on the exact-fp32 kernels."""
    depth = _tl("depth", 0)
    if depth == 0:
        _tls.ran = []
    _tls.depth = depth + 1
    try:
        out = fn()
    finally:
        _tls.depth = depth
    policy = guard_policy()
    if depth != 0:
        return out
    ran, _tls.ran = _tl("ran", None) or [], None
    if policy == "deferred" or not x3_launched(device):
        return out
    if policy not in X3_GUARDS:
        raise _lib.PramHipError(f"unknown x3 guard {policy!r} (expected one of {X3_GUARDS})")
    if torch.cuda.is_current_stream_capturing() or not x3_range_exceeded(device):
        return out
    if policy == "raise":
        raise _lib.PramHipError("split-fp16 path: an activation beyond the range of its planes (|x| >= 65520 / act_scale; "
                                "4094.97 at the default scale) — lower the models' act_scale, or re-run with precision 'f32' "
                                "(PRAM_X3_GUARD=fallback does both by itself)")
    while True:
        scalable = [m for m in ran if getattr(m, "act_scale", None) and m.act_scale / 16.0 >= ACT_SCALE_MIN]
        if not scalable:
            break
        for m in scalable:
            m.act_scale = m.act_scale / 16.0
        guard_events["rescaled"] += 1
        _tls.depth, _tls.ran = depth + 1, None
        try:
            out = fn()
        finally:
            _tls.depth = depth
        if not x3_range_exceeded(device):
            return out
    guard_events["f32_fallback"] += 1
    _tls.depth = depth + 1
    try:
        with forced_precision("f32"):
            return fn()
    finally:
        _tls.depth = depth


# Derived forms of a (static) weight tensor — its fp16 copy, its split planes — live exactly as long as the tensor object
# they were derived from and are rebuilt when it is modified in place (tensor._version), never keyed by a device address
# that the allocator may hand to another tensor.
_derived_cache = {}


def _derived(w: torch.Tensor, kind: str, fn):
    ent = _derived_cache.get(id(w))
    if ent is None or ent[0]() is not w or ent[1] != w._version:
        key = id(w)
        ent = (_weakref.ref(w, lambda _r, k=key: _derived_cache.pop(k, None)), w._version, {})
        _derived_cache[key] = ent
    v = ent[2].get(kind)
    if v is None:
        with torch.no_grad():
            v = ent[2][kind] = fn(w)
    return v


def _w16(w: torch.Tensor) -> torch.Tensor:
    """fp16 copy of a (static) weight tensor."""
    return _derived(w, "h16", lambda t: t.half().contiguous())


def split_weight_raw(t: torch.Tensor):
    """(hi, lo, scale) of a tensor, uncached: see split_weight."""
    import math
    amax = float(t.abs().max())
    scale = 2.0 ** math.floor(math.log2(16384.0 / amax)) if amax > 0 and math.isfinite(amax) else 1.0
    ts = t.float() * scale
    hi = ts.half()
    lo = (ts - hi.float()).half()
    return hi.contiguous(), lo.contiguous(), float(scale)


def split_weight(w: torch.Tensor):
    """-> (hi, lo, scale): w * scale = hi + lo as two fp16 tensors of w's shape; scale = the power of two that puts
    max|w| into [2^13, 2^14) (exact to apply and to undo), so hi + lo carries 22 bits of every weight above 2^-17 max|w|."""
    make = split_weight_raw
    _x3_status(w.device)      # every split-fp16 GEMM / convolution comes through here: the range guard's status word is registered,
    _mark_x3(w.device)        # and the device is marked as having something to report (guarded_call / x3_launched)
    return _derived(w, "x3", make)


attention_split = True      # False: never hand the kernel a split workspace (tests compare both modes bit for bit)


def _attention_ws(L, batch, heads, m_max, n_max, device):
    """workspace for the split (small-launch) mode, per stream: two streams may run attention concurrently"""
    nb = int(L.pram_attention_workspace_bytes(batch, heads, m_max, n_max)) if attention_split else 0
    if nb == 0:
        return None, 0
    return _workspace(nb, device, "attention"), nb


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, heads: int, m_max: int, n_max: int,
              scale: float, q_lens: Optional[torch.Tensor] = None, k_lens: Optional[torch.Tensor] = None,
              want_lse: bool = False, out: Optional[torch.Tensor] = None, precision: Optional[str] = None):
    """q/k/v: 2-D row-major views (possibly column slices of a wider buffer): q [batch*m_max, >=heads*64]."""
    L = _lib.load()
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, nm)
        assert t.dim() == 2 and t.stride(1) == 1
    if out is None:
        out = torch.empty(batch * m_max, heads * 64, device=q.device, dtype=torch.float32)
    lse = torch.empty(batch, heads, m_max, device=q.device, dtype=torch.float32) if want_lse else None
    probe = attention_probe
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    prec = _check_precision(_tl("forced") or precision or attn_prec())
    if prec == "x3":
        prec = "f32"      # fp32 operands in HBM: the split path starts at the projection (attention_x3 takes its planes)
    if prec == "f32":
        ws, nb = _attention_ws(L, batch, heads, m_max, n_max, q.device)
        rc = L.pram_attention_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                  _p(lse), _p(q_lens), _p(k_lens), batch, heads, m_max, n_max, float(scale), _p(ws), nb, _st())
    else:
        rc = L.pram_attention_f16_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                      _p(lse), _p(q_lens), _p(k_lens), batch, heads, m_max, n_max, float(scale), _st())
    _lib.check(rc, "pram_attention_" + prec)
    if probe is not None:
        e1.record()
        # algorithmic FLOPs of QK^T + PV: 4 * m_b * n_b * 64 per (batch element, head), from the ACTUAL ragged
        # lengths (device tensors, summed by the caller after synchronising)
        probe.append((q_lens, k_lens, m_max, n_max, heads, batch, e0, e1, "self", 16))
    return (out, lse) if want_lse else out


def attention_colmean(q: torch.Tensor, k: torch.Tensor, lse2: torch.Tensor, batch: int, heads: int, m_max: int,
                      n_max: int, scale: float, q_lens=None, k_lens=None) -> torch.Tensor:
    L = _lib.load()
    out = _filled((batch, n_max), q.device)
    _lib.check(L.pram_attention_colmean_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(lse2), _p(out), _p(q_lens),
                                            _p(k_lens), batch, heads, m_max, n_max, float(scale), _st()),
               "pram_attention_colmean_f32")
    return out


def attention_h16(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, heads: int, m_max: int, n_max: int, scale: float,
                  q_lens: Optional[torch.Tensor] = None, k_lens: Optional[torch.Tensor] = None, want_lse: bool = False,
                  out: Optional[torch.Tensor] = None, kv_shift: int = 0):
    """attention(precision="f16") on fp16 q / k / v (2-D views, possibly column slices of one fp16 projection output):
    same bits out, half the operand traffic.  kv_shift: see attention_cross."""
    L = _lib.load()
    for t in (q, k, v):
        assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
    if out is None:
        out = torch.empty(batch * m_max, heads * 64, device=q.device, dtype=torch.float32)
    lse = torch.empty(batch, heads, m_max, device=q.device, dtype=torch.float32) if want_lse else None
    probe = attention_probe
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(L.pram_attention_h16_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                        _p(lse), _p(q_lens), _p(k_lens), batch, heads, m_max, n_max, float(scale), int(kv_shift),
                                        _st()), "pram_attention_h16_f32")
    if probe is not None:
        e1.record()
        kl = k_lens if (k_lens is None or not kv_shift) else torch.roll(k_lens, -kv_shift)
        probe.append((q_lens, kl, m_max, n_max, heads, batch, e0, e1, "cross" if kv_shift else "self", 16))
    return (out, lse) if want_lse else out


def value_t16(v16: torch.Tensor, seqs: int, heads: int, t_max: int, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp16 row-major values [seqs * t_max, >= heads * 64] -> the transposed, key-permuted fp16 values [seqs, heads, 64, tv] of
    attention_h16t (the single-plane form of value_planes_t)."""
    L = _lib.load()
    assert v16.is_cuda and v16.dtype == torch.float16 and v16.dim() == 2 and v16.stride(1) == 1
    tv = (t_max + 63) // 64 * 64
    out = torch.empty(seqs, heads, 64, tv, device=v16.device, dtype=torch.float16)
    _lib.check(L.pram_attention_x3_vt(_p(v16), None, v16.stride(0), _p(out), None, _p(lens), seqs, heads, t_max, _st()), "pram_attention_x3_vt")
    return out


def attention_h16t(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, batch: int, heads: int, m_max: int, n_max: int, scale: float,
                   q_lens: Optional[torch.Tensor] = None, k_lens: Optional[torch.Tensor] = None, want_lse: bool = False,
                   out: Optional[torch.Tensor] = None, kv_shift: int = 0, out16: bool = False):
    """The fp16 path's attention on the software-pipelined kernel: fp16 q / k (2-D views), vt = value_t16(...) of the key side.
    One fp16 MFMA per product, probabilities rounded to fp16 (tolerance of the fp16 path)."""
    L = _lib.load()
    for t in (q, k):
        assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
    assert vt.is_contiguous() and vt.dtype == torch.float16
    if out is None:
        out = torch.empty(batch * m_max, heads * 64, device=q.device, dtype=torch.float16 if out16 else torch.float32)
    lse = torch.empty(batch, heads, m_max, device=q.device, dtype=torch.float32) if want_lse else None
    probe = attention_probe
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if out16:      # the context as fp16: the operand format of the fp16 path's next GEMM
        assert out.dtype == torch.float16
        _lib.check(L.pram_attention_h16t_h16(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), _p(out), out.stride(0), _p(lse), _p(q_lens),
                                             _p(k_lens), batch, heads, m_max, n_max, float(scale), int(kv_shift), _st()), "pram_attention_h16t_h16")
    else:
        _lib.check(L.pram_attention_h16t_f32(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), _p(out), out.stride(0), _p(lse), _p(q_lens),
                                             _p(k_lens), batch, heads, m_max, n_max, float(scale), int(kv_shift), _st()), "pram_attention_h16t_f32")
    if probe is not None:
        e1.record()
        kl = k_lens if (k_lens is None or not kv_shift) else torch.roll(k_lens, -kv_shift)
        probe.append((q_lens, kl, m_max, n_max, heads, batch, e0, e1, "cross" if kv_shift else "self", 16))
    return (out, lse) if want_lse else out


def value_planes_t(v, seqs: int, heads: int, t_max: int, lens: Optional[torch.Tensor] = None):
    """(hi, lo) row-major value planes [seqs * t_max, >= heads * 64] (column slices of linear(split_out=...)'s planes) ->
    the transposed, key-permuted planes [seqs, heads, 64, tv] attention_x3 stages with 16-byte copies; zeros beyond lens."""
    L = _lib.load()
    tv = (t_max + 63) // 64 * 64
    out = torch.empty(2, seqs, heads, 64, tv, device=v[0].device, dtype=torch.float16)
    assert v[0].stride(0) == v[1].stride(0) and v[0].stride(1) == 1 and v[0].dtype == torch.float16
    _lib.check(L.pram_attention_x3_vt(_p(v[0]), _p(v[1]), v[0].stride(0), _p(out[0]), _p(out[1]), _p(lens), seqs, heads, t_max, _st()),
               "pram_attention_x3_vt")
    return out[0], out[1]


def attention_x3(q, k, vt, batch: int, heads: int, m_max: int, n_max: int, scale: float,
                 q_lens: Optional[torch.Tensor] = None, k_lens: Optional[torch.Tensor] = None, want_lse: bool = False,
                 out: Optional[torch.Tensor] = None, kv_shift: int = 0):
    """Split-fp16 flash attention.  q / k: (hi, lo) pairs of fp16 2-D views (column slices of the planes written by
    linear(split_out=...)); vt: the (hi, lo) planes of value_planes_t for the key side.  fp32-class results, three fp16
    MFMAs per product.  kv_shift: see attention_cross."""
    L = _lib.load()
    for pair in (q, k):
        for t in pair:
            assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
        assert pair[0].stride(0) == pair[1].stride(0)
    tv = (n_max + 63) // 64 * 64
    assert vt[0].is_contiguous() and vt[1].is_contiguous() and vt[0].dtype == torch.float16 and vt[0].numel() == batch * heads * 64 * tv
    if out is None:
        out = torch.empty(batch * m_max, heads * 64, device=q[0].device, dtype=torch.float32)
    lse = torch.empty(batch, heads, m_max, device=q[0].device, dtype=torch.float32) if want_lse else None
    probe = attention_probe
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # from 1024 keys on the keys are reduced in chunks; under-filled launches (one or two query frames) run the chunks as a grid
    # dimension through a workspace: same bits, shorter serial walk.  attention_split = False (tests): no workspace -> always fused.
    nb = int(L.pram_attention_x3_workspace_bytes(batch, heads, m_max, n_max)) if attention_split else 0
    ws = _workspace(nb, q[0].device, "attention_x3") if nb else None
    _x3_status(q[0].device)
    _lib.check(L.pram_attention_x3_f32(_p(q[0]), _p(q[1]), q[0].stride(0), _p(k[0]), _p(k[1]), k[0].stride(0), _p(vt[0]), _p(vt[1]),
                                       _p(out), out.stride(0), _p(lse), _p(q_lens), _p(k_lens), batch, heads,
                                       m_max, n_max, float(scale), int(kv_shift), _p(ws), nb, _st()), "pram_attention_x3_f32")
    if probe is not None:
        e1.record()
        kl = k_lens if (k_lens is None or not kv_shift) else torch.roll(k_lens, -kv_shift)
        probe.append((q_lens, kl, m_max, n_max, heads, batch, e0, e1, "cross" if kv_shift else "self",
                      int(L.pram_attention_x3_mfma_per_tile(n_max))))
    return (out, lse) if want_lse else out


def attention_colmean_x3(q, k, lse2: torch.Tensor, batch: int, heads: int, m_max: int, n_max: int, scale: float,
                         q_lens=None, k_lens=None, kv_shift: int = 0) -> torch.Tensor:
    """Column means of the soft-max matrix attention_x3 just produced (its q / k planes, its lse) -> [batch, n_max] fp32, row
    (b + kv_shift) % batch holding the means over sequence b's queries; keys beyond their length read 0."""
    L = _lib.load()
    for pair in (q, k):
        for t in pair:
            assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
        assert pair[0].stride(0) == pair[1].stride(0)
    out = _filled((batch, n_max), lse2.device)
    _lib.check(L.pram_attention_x3_colmean_f32(_p(q[0]), _p(q[1]), q[0].stride(0), _p(k[0]), _p(k[1]), k[0].stride(0), _p(lse2), _p(out),
                                               _p(q_lens), _p(k_lens), batch, heads, m_max, n_max, float(scale), int(kv_shift), _st()),
               "pram_attention_x3_colmean_f32")
    return out


def attention_cross(qk: torch.Tensor, v: torch.Tensor, pairs: int, heads: int, t_max: int, scale: float,
                    lens: Optional[torch.Tensor] = None, want_lse: bool = False, out: Optional[torch.Tensor] = None,
                    precision: Optional[str] = None):
    """Both directions of the matcher's cross attention in one launch.  qk / v: [2*pairs*t_max, >= heads*64] row-major
    views; sequences 0..pairs-1 are set 0 and attend to set 1 (pairs..2*pairs-1) and vice versa.  lens int32 [2*pairs].
    Equal, bit for bit, to attention(qk0, qk1, v1) and attention(qk1, qk0, v0) written to the two halves of out."""
    L = _lib.load()
    for t, nm in ((qk, "qk"), (v, "v")):
        _chk(t, nm)
        assert t.dim() == 2 and t.stride(1) == 1
    S = 2 * pairs
    if out is None:
        out = torch.empty(S * t_max, heads * 64, device=qk.device, dtype=torch.float32)
    lse = torch.empty(S, heads, t_max, device=qk.device, dtype=torch.float32) if want_lse else None
    probe = attention_probe
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    prec = _check_precision(_tl("forced") or precision or attn_prec())
    if prec == "x3":
        prec = "f32"
    if prec == "f32":
        ws, nb = _attention_ws(L, S, heads, t_max, t_max, qk.device)
        rc = L.pram_attention_cross_f32(_p(qk), qk.stride(0), _p(v), v.stride(0), _p(out), out.stride(0), _p(lse), _p(lens),
                                        pairs, heads, t_max, float(scale), _p(ws), nb, _st())
    else:
        rc = L.pram_attention_cross_f16_f32(_p(qk), qk.stride(0), _p(v), v.stride(0), _p(out), out.stride(0), _p(lse), _p(lens),
                                            pairs, heads, t_max, float(scale), _st())
    _lib.check(rc, "pram_attention_cross_" + prec)
    if probe is not None:
        e1.record()
        probe.append((lens, None if lens is None else torch.roll(lens, -pairs), t_max, t_max, heads, S, e0, e1, "cross", 16))
    return (out, lse) if want_lse else out


def attention_cross_colmean(qk: torch.Tensor, lse2: torch.Tensor, pairs: int, heads: int, t_max: int, scale: float,
                            lens=None) -> torch.Tensor:
    """[2*pairs, t_max]: row kb = per token of sequence kb, mean attention received from the other set's queries."""
    L = _lib.load()
    out = _filled((2 * pairs, t_max), qk.device)
    _lib.check(L.pram_attention_cross_colmean_f32(_p(qk), qk.stride(0), _p(lse2), _p(out), _p(lens), pairs, heads, t_max,
                                                  float(scale), _st()), "pram_attention_cross_colmean_f32")
    return out


_ws_cache = {}
_ws_scopes = []          # innermost active workspace_scope store
_ws_captured = []        # scratch allocated while a stream was capturing outside any scope: owned by the process


class workspace_scope:
    """Route every scratch allocation of the enclosed launches into ``store`` (a dict owned by the caller) instead of
    the process-wide cache.  ``GraphedPipeline`` warms up and captures inside its own scope, so the pointers baked into
    its hipGraph belong to that graph alone: a later eager run that needs a larger workspace, or a second graph
    captured on torch's shared capture stream, can neither free nor share them.  A buffer that has to grow inside a
    scope is retired, not freed (an earlier capture may still point at it)."""

    def __init__(self, store: dict):
        self.store = store

    def __enter__(self):
        _ws_scopes.append(self.store)
        return self.store

    def __exit__(self, *exc):
        _ws_scopes.pop()
        return False


def _workspace(nbytes: int, device, tag: str = "") -> torch.Tensor:
    """Scratch memory owned by (device, purpose, stream): launches on different streams never share a workspace, so
    several batches can be in flight at once.  See workspace_scope for hipGraph capture."""
    key = (str(device), tag, _st())
    scoped = bool(_ws_scopes)
    store = _ws_scopes[-1] if scoped else _ws_cache
    ws = store.get(key)
    if ws is not None and ws.numel() >= nbytes:
        return ws
    new = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
    if scoped:
        if ws is not None:
            store.setdefault("_retired", []).append(ws)
        store[key] = new
    elif torch.cuda.is_current_stream_capturing():
        # captured outside a scope: never enters (or evicts from) the eager cache, and is never freed
        _ws_captured.append(new)
    else:
        store[key] = new
    return new


# plans of one Sinkhorn call: at most this many bytes per group of pairs (0 = one call for the whole batch)
sinkhorn_group_bytes = int(_os.environ.get("PRAM_SINKHORN_GROUP_MB", "135")) << 20      # 0: one group (see sinkhorn_match)


def sinkhorn_match(dist: torch.Tensor, bin_score: torch.Tensor, iters: int, threshold: float,
                   m_lens=None, n_lens=None, want_p: bool = False, dual_softmax: bool = False, n_valid: Optional[int] = None):
    """dist [B, M, ldd] (first n_valid (default ldd) columns valid).  Returns dict with matches0/1 (int64),
    matching_scores0/1 and optionally the full assignment matrix 'p' [B, M+1, N+1]."""
    L = _lib.load()
    _chk(dist, "dist")
    assert dist.is_contiguous() and dist.dim() == 3
    B, M, ldd = dist.shape
    N = n_valid or ldd
    m0 = torch.empty(B, M, device=dist.device, dtype=_INT64)
    m1 = torch.empty(B, N, device=dist.device, dtype=_INT64)
    s0 = torch.empty(B, M, device=dist.device, dtype=torch.float32)
    s1 = torch.empty(B, N, device=dist.device, dtype=torch.float32)
    p = torch.zeros(B, M + 1, N + 1, device=dist.device, dtype=torch.float32) if want_p else None
    bs = bin_score.reshape(1).float()
    # The 20 iterations stream the whole [m + 1, n + 1] plan of every pair: a group of pairs whose plans fit the 256 MB infinity
    # cache together iterates out of it (35 us per pair and call instead of 67 from HBM, profiles/r02_sinkhorn_batch.txt), so a
    # large batch runs group after group — same kernels, every pair its own rows: results do not depend on the grouping.
    # Default (round 5, profiles/r05_sinkhorn_group_ab.txt): groups of <= 135 MB (8 pairs of 2049 x 2049: +0.7 % queries/s in three
    # alternations on one box), but only when a group still holds at least four pairs — at 4097 x 4097 (67 MB per pair) two-pair
    # groups double the call's 42 dependent launches for nothing (-0.8 %).
    per_pair = (M + 1) * ((N + 4) // 4 * 4) * 4
    group = B
    if sinkhorn_group_bytes > 0 and sinkhorn_group_bytes // max(per_pair, 1) >= 4:
        group = max(1, min(B, sinkhorn_group_bytes // max(per_pair, 1)))
    for b0 in range(0, B, group):
        b1 = min(B, b0 + group)
        nb = b1 - b0
        ws = _workspace(L.pram_sinkhorn_workspace_bytes(nb, M, N), dist.device, "sinkhorn")
        sl = lambda t: None if t is None else _p(t[b0:b1])
        if dual_softmax:
            rc = L.pram_dual_softmax_match_f32(_p(dist[b0:b1]), ldd, sl(m_lens), sl(n_lens), _p(bs), float(threshold), sl(p), N + 1,
                                               _p(m0[b0:b1]), _p(m1[b0:b1]), _p(s0[b0:b1]), _p(s1[b0:b1]), nb, M, N, _p(ws), _st())
        else:
            rc = L.pram_sinkhorn_match_f32(_p(dist[b0:b1]), ldd, sl(m_lens), sl(n_lens), _p(bs), int(iters), float(threshold), sl(p),
                                           N + 1, _p(m0[b0:b1]), _p(m1[b0:b1]), _p(s0[b0:b1]), _p(s1[b0:b1]), nb, M, N, _p(ws), _st())
        _lib.check(rc, "pram_sinkhorn_match_f32")
    out = {"matches0": m0, "matches1": m1, "matching_scores0": s0, "matching_scores1": s1}
    if want_p:
        out["p"] = p
    return out


def adagml_prune(logit, thr, n_min_tokens, lens_in, x, cos, sin, ind, want_conf=False, ld_logit: int = 1):
    """-> (x_out, cos_out, sin_out, ind_out, lens_out int32 [S], n_below int32 [S], conf or None).  ld_logit > 1: ``logit`` is the
    [S * T, ld_logit] output of the pooling head's padded last Linear and the logits are its column 0, read in place."""
    L = _lib.load()
    if ld_logit > 1:
        S, T = x.shape[0], x.shape[1]
        assert logit.is_contiguous() and logit.numel() == S * T * ld_logit
    else:
        S, T = logit.shape
    ldx = x.shape[-1]
    # rows at and beyond the new length of a set are never written — and never read: every consumer is ragged (lens)
    x_o, cos_o, sin_o, ind_o = torch.empty_like(x), torch.empty_like(cos), torch.empty_like(sin), _filled(tuple(ind.shape), ind.device, torch.int32)
    lens_o = torch.empty(S, device=x.device, dtype=torch.int32)
    n_below = torch.empty(S, device=x.device, dtype=torch.int32)
    conf = _filled((S, T), x.device) if want_conf else None
    row_map = torch.empty(S, T, device=x.device, dtype=torch.int32)
    _lib.check(L.pram_adagml_prune_ld_f32(_p(logit), int(ld_logit), float(thr), int(n_min_tokens), _p(lens_in), _p(x), _p(cos), _p(sin), _p(ind),
                                          _p(x_o), _p(cos_o), _p(sin_o), _p(ind_o), _p(lens_o), _p(n_below), _p(conf), _p(row_map),
                                          S, T, ldx, _st()),
               "pram_adagml_prune_f32")
    return x_o, cos_o, sin_o, ind_o, lens_o, n_below, conf


def adagml_scores4(col_self: torch.Tensor, col_cross: torch.Tensor) -> torch.Tensor:
    """[S, T] self / cross attention scores per token -> [S * T, 4] rows (self, cross, 0, 0): the pooling head's input."""
    L = _lib.load()
    assert col_self.is_contiguous() and col_cross.is_contiguous() and col_self.shape == col_cross.shape
    out = torch.empty(col_self.numel(), 4, device=col_self.device, dtype=torch.float32)
    _lib.check(L.pram_adagml_scores4_f32(_p(col_self), _p(col_cross), _p(out), col_self.numel(), _st()), "pram_adagml_scores4_f32")
    return out


def adagml_layer_state(active, lens, lens_new, n_below, num_points, tiny, stop_layer, lens_final, ind, ind_final, pairs: int, t_max: int,
                       layer: int, last: bool):
    """One layer of the batched AdaGML bookkeeping (pram_adagml_layer_state) -> (active', lens', lens_stop, lens_eff); tiny,
    stop_layer, lens_final and ind_final are updated in place."""
    L = _lib.load()
    dev = lens.device
    active_o = torch.empty_like(active)
    lens_o = torch.empty_like(lens)
    lens_stop = torch.empty_like(lens)
    lens_eff = torch.empty_like(lens)
    _lib.check(L.pram_adagml_layer_state(_p(active), _p(active_o), _p(lens), _p(lens_o), _p(lens_new), _p(n_below), _p(num_points), _p(tiny),
                                         _p(stop_layer), _p(lens_final), _p(lens_stop), _p(lens_eff), _p(ind), _p(ind_final), int(pairs),
                                         int(t_max), int(layer), int(bool(last)), _st()), "pram_adagml_layer_state")
    return active_o, lens_o, lens_stop, lens_eff


def adagml_scatter(matches0, mscores0, ind0, ind1, lens0, m_full):
    L = _lib.load()
    B, T = matches0.shape
    out_m = _filled((B, m_full, 2), matches0.device, torch.int32, -1).view(_INT64).view(B, m_full)      # int64 -1 = two 0xFFFFFFFF words
    out_s = _filled((B, m_full), matches0.device)
    _lib.check(L.pram_adagml_scatter_f32(_p(matches0), _p(mscores0), _p(ind0), _p(ind1), _p(lens0), B, T, m_full, _p(out_m),
                                         _p(out_s), _st()), "pram_adagml_scatter_f32")
    return out_m, out_s


# ------------------------------------------------------------------------------------- SFD2
def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias=None, scale=None, shift=None, residual=None, ks: int = 3,
                stride: int = 1, relu: bool = False, precision: Optional[str] = None, l2norm: bool = False) -> torch.Tensor:
    """x [B,H,W,Cin] contiguous NHWC; w [Cout,ks,ks,Cin].  precision: see linear() (conv1a's 4-channel input always runs
    on the exact-fp32 kernel).  l2norm: F.normalize over the channels of every output pixel behind the layer — inside the
    convolution's epilogue on the split-fp16 path when Cout <= 128, as a second kernel (l2norm_rows_) otherwise."""
    L = _lib.load()
    _chk(x, "x")
    assert x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    out = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    prec = _check_precision(_tl("forced") or precision or gemm_prec())
    if prec == "x3" and Cin % 32 == 0:
        wh, wl, ws = split_weight(w)
        if l2norm and Cout <= 128 and FUSED_L2NORM:
            _lib.check(L.pram_conv2d_nhwc_x3_l2norm_f32(_p(x), B, H, W, Cin, _p(wh), _p(wl), ws, _p(bias), _p(scale), _p(shift), _p(residual),
                                                        _p(out), Cout, ks, stride, int(relu), _st()), "pram_conv2d_nhwc_x3_l2norm_f32")
            return out
        _lib.check(L.pram_conv2d_nhwc_x3_f32(_p(x), B, H, W, Cin, _p(wh), _p(wl), ws, _p(bias), _p(scale), _p(shift), _p(residual),
                                             _p(out), Cout, ks, stride, int(relu), _st()), "pram_conv2d_nhwc_x3_f32")
    elif prec == "f16" and Cin % 64 == 0:
        _lib.check(L.pram_conv2d_nhwc_f16_f32(_p(x), B, H, W, Cin, _p(_w16(w)), _p(bias), _p(scale), _p(shift), _p(residual),
                                              _p(out), Cout, ks, stride, int(relu), _st()), "pram_conv2d_nhwc_f16_f32")
    else:
        _lib.check(L.pram_conv2d_nhwc_f32(_p(x), B, H, W, Cin, _p(w), _p(bias), _p(scale), _p(shift), _p(residual), _p(out),
                                          Cout, ks, stride, int(relu), _st()), "pram_conv2d_nhwc_f32")
    return l2norm_rows_(out) if l2norm else out


FUSED_L2NORM = _os.environ.get("PRAM_FUSED_L2NORM", "1") != "0"      # split-fp16 path: F.normalize inside the producing convolution
GROUPED_X3 = _os.environ.get("PRAM_GROUPED_X3", "1") != "0"      # split-fp16 path: the grouped 3x3 on the matrix pipe (0: vector ALU)


def conv3x3_grouped_nhwc(x: torch.Tensor, w: torch.Tensor, scale, shift, groups: int, relu: bool) -> torch.Tensor:
    """Grouped 3x3 (8 channels per group) of the ResBlock, exact fp32 on the vector ALU."""
    L = _lib.load()
    assert x.is_contiguous() and w.is_contiguous()
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    _lib.check(L.pram_conv3x3_grouped_nhwc_f32(_p(x), B, H, W, Cc, _p(w), _p(scale), _p(shift), _p(out), groups, int(relu), _st()),
               "pram_conv3x3_grouped_nhwc_f32")
    return out


def conv2d_nhwc_planes(x: torch.Tensor, w: torch.Tensor, bias=None, scale=None, shift=None, residual=None, ks: int = 1,
                       stride: int = 1, relu: bool = False):
    """conv2d_nhwc on the split-fp16 path with the result as fp16 planes (hi, lo), hi + lo = 16 y: the split operand of the next
    split-fp16 layer (conv3x3_grouped_planes), written once by the producer instead of being recomputed by the consumer."""
    L = _lib.load()
    _chk(x, "x")
    assert x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    hi = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float16)
    lo = torch.empty_like(hi)
    wh, wl, ws = split_weight(w)
    _lib.check(L.pram_conv2d_nhwc_x3_planes(_p(x), B, H, W, Cin, _p(wh), _p(wl), ws, _p(bias), _p(scale), _p(shift), _p(residual),
                                            _p(hi), _p(lo), Cout, ks, stride, int(relu), _st()), "pram_conv2d_nhwc_x3_planes")
    return hi, lo


def conv3x3_grouped_planes(hi: torch.Tensor, lo: torch.Tensor, w: torch.Tensor, scale, shift, groups: int, relu: bool) -> torch.Tensor:
    """Grouped 3x3 (8 channels per group) on the matrix pipe (pram_conv3x3_grouped_planes_x3_f32): (hi, lo) from
    conv2d_nhwc_planes, w [C, 3, 3, 8] fp32; fp32 NHWC result."""
    L = _lib.load()
    assert hi.is_contiguous() and lo.is_contiguous() and w.is_contiguous() and hi.dtype == lo.dtype == torch.float16
    B, H, W, Cc = hi.shape
    out = torch.empty(B, H, W, Cc, device=hi.device, dtype=torch.float32)
    wh, wl, ws = split_weight(w)
    _lib.check(L.pram_conv3x3_grouped_planes_x3_f32(_p(hi), _p(lo), B, H, W, Cc, _p(wh), _p(wl), ws, _p(scale), _p(shift), _p(out),
                                                    groups, int(relu), _st()), "pram_conv3x3_grouped_planes_x3_f32")
    return out


def sfd2_conv1(x4: torch.Tensor, wa: torch.Tensor, ba, sa, ta, wb: torch.Tensor, bb, sb, tb) -> torch.Tensor:
    """SFD2's conv1a -> conv1b (3x3 / stride 1 / 3 -> 64, then 3x3 / stride 2 / 64 -> 64, each bias -> BN -> ReLU; nets/sfd2.py:135-139,
    281-282) in ONE launch on the split-fp16 path (pram_sfd2_conv1_x3_f32): the 1.26 GB conv1a map never exists.  x4: the NHWC4 image
    (image_to_nhwc4) or the fp32 NCHW image [B, 3, H, W] itself; wa [64, 3, 3, 4], wb [64, 3, 3, 64] as conv2d_nhwc takes them.
    -> [B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64]."""
    L = _lib.load()
    assert x4.is_contiguous() and x4.dtype == torch.float32 and tuple(wa.shape) == (64, 3, 3, 4) and tuple(wb.shape) == (64, 3, 3, 64)
    nchw3 = x4.shape[-1] != 4
    if nchw3:
        assert x4.dim() == 4 and x4.shape[1] == 3
        B, _, H, W = x4.shape
    else:
        B, H, W, _ = x4.shape

    def pad48(t):      # [64][36] -> [64][48]: K padded to three 16-deep steps, then the usual split
        flat = t.reshape(64, 36).float()
        return split_weight_raw(torch.cat([flat, flat.new_zeros(64, 12)], 1).contiguous())
    _x3_status(x4.device)
    _mark_x3(x4.device)
    wah, wal, wsa = _derived(wa, "x3pad48", pad48)
    wbh, wbl, wsb = split_weight(wb)
    out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64, device=x4.device, dtype=torch.float32)
    _lib.check(L.pram_sfd2_conv1_x3_f32(_p(x4), B, H, W, _p(wah), _p(wal), wsa, _p(ba), _p(sa), _p(ta), _p(wbh), _p(wbl), wsb,
                                        _p(bb), _p(sb), _p(tb), _p(out), int(nchw3), _st()), "pram_sfd2_conv1_x3_f32")
    return out


def image_to_nhwc4(img: torch.Tensor) -> torch.Tensor:
    L = _lib.load()
    img = img.contiguous()
    _chk(img, "image")
    B, Cc, H, W = img.shape
    assert Cc == 3
    out = torch.empty(B, H, W, 4, device=img.device, dtype=torch.float32)
    _lib.check(L.pram_image_to_nhwc4_f32(_p(img), _p(out), B, H, W, _st()), "pram_image_to_nhwc4_f32")
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    L = _lib.load()
    assert x.is_contiguous()
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, H, W, device=x.device, dtype=torch.float32)
    _lib.check(L.pram_nhwc_to_nchw_f32(_p(x), _p(out), B, H, W, Cc, _st()), "pram_nhwc_to_nchw_f32")
    return out


def score_map(logits_nhwc: torch.Tensor) -> torch.Tensor:
    L = _lib.load()
    assert logits_nhwc.is_contiguous() and logits_nhwc.shape[-1] == 65
    B, Hc, Wc, _ = logits_nhwc.shape
    out = torch.empty(B, Hc * 8, Wc * 8, device=logits_nhwc.device, dtype=torch.float32)
    _lib.check(L.pram_score_map_f32(_p(logits_nhwc), _p(out), B, Hc, Wc, _st()), "pram_score_map_f32")
    return out


def simple_nms(score: torch.Tensor, radius: int) -> torch.Tensor:
    L = _lib.load()
    score = score.contiguous()
    _chk(score, "score")
    B, H, W = score.shape
    out = torch.empty_like(score)
    ws = _workspace(L.pram_simple_nms_workspace_bytes(B, H, W), score.device, "nms")
    _lib.check(L.pram_simple_nms_f32(_p(score), _p(out), B, H, W, int(radius), _p(ws), _st()), "pram_simple_nms_f32")
    return out


def select_keypoints(nms: torch.Tensor, conf_th: float, min_keypoints: int, border: int, max_keypoints: int,
                     fallback_ref: int = 0):
    """-> kpts [B,k,2] (x,y), scores [B,k], counts [B] int32 (device)."""
    L = _lib.load()
    assert nms.is_contiguous()
    B, H, W = nms.shape
    k = int(max_keypoints)
    ws = _workspace(L.pram_select_keypoints_workspace_bytes(B, H, W, k), nms.device, "select")
    kpts = _filled((B, k, 2), nms.device)
    scores = _filled((B, k), nms.device)
    counts = _filled((B,), nms.device, torch.int32)
    _lib.check(L.pram_select_keypoints_f32(_p(nms), B, H, W, float(conf_th), int(min_keypoints), int(border), k,
                                           int(fallback_ref), _p(kpts), _p(scores), _p(counts), _p(ws), _st()),
               "pram_select_keypoints_f32")
    return kpts, scores, counts


def sample_nhwc(fmap: torch.Tensor, kpts: torch.Tensor, lens: Optional[torch.Tensor], s: int, l2norm: bool) -> torch.Tensor:
    """fmap [B,fh,fw,C] NHWC, kpts [B,N,2] -> [B,N,C]"""
    L = _lib.load()
    assert fmap.is_contiguous()
    kpts = kpts.contiguous()
    B, fh, fw, Cc = fmap.shape
    N = kpts.shape[1]
    out = _filled((B, N, Cc), fmap.device)
    _lib.check(L.pram_sample_nhwc_f32(_p(fmap), B, fh, fw, Cc, _p(kpts), _p(lens), N, int(s), int(l2norm), _p(out), _st()),
               "pram_sample_nhwc_f32")
    return out


def l2norm_rows_(x: torch.Tensor) -> torch.Tensor:
    L = _lib.load()
    assert x.is_contiguous()
    cols = x.shape[-1]
    _lib.check(L.pram_l2norm_rows_f32(_p(x), x.numel() // cols, cols, _st()), "pram_l2norm_rows_f32")
    return x


def score_lookup(score_map_: torch.Tensor, kpts: torch.Tensor, lens: Optional[torch.Tensor]) -> torch.Tensor:
    L = _lib.load()
    score_map_ = score_map_.contiguous()
    kpts = kpts.contiguous()
    B, H, W = score_map_.shape
    Bk, N = kpts.shape[0], kpts.shape[1]
    out = _filled((Bk, N), kpts.device)
    stride = 0 if B == 1 else H * W
    _lib.check(L.pram_score_lookup_f32(_p(score_map_), stride, H, W, _p(kpts), _p(lens), Bk, N, _p(out), _st()),
               "pram_score_lookup_f32")
    return out


# ------------------------------------------------------------------------------------- edges of the path
def resize_bilinear(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """F.interpolate(x, size=(oh, ow), mode='bilinear', align_corners=True) for [..., h, w] planar maps."""
    L = _lib.load()
    x = x.contiguous().float()
    _chk(x, "x")
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    out = torch.empty(*x.shape[:-2], oh, ow, device=x.device, dtype=torch.float32)
    _lib.check(L.pram_resize_bilinear_f32(_p(x), _p(out), planes, h, w, int(oh), int(ow), _st()), "pram_resize_bilinear_f32")
    return out


def seg_epilogue(logits: torch.Tensor, lens: Optional[torch.Tensor], bg_threshold: float, want_scores: bool = False):
    """logits [B,N,C] -> seg_ids int32 [B,N], non_bg_mask int32 [B,N], n_non_bg int32 [B], seg_scores or None."""
    L = _lib.load()
    logits = logits.contiguous()
    _chk(logits, "logits")
    B, N, Cc = logits.shape
    ids = _filled((B, N), logits.device, torch.int32, -2)
    mask = _filled((B, N), logits.device, torch.int32)
    cnt = torch.empty(B, device=logits.device, dtype=torch.int32)      # cleared by the entry itself
    sc = _filled(tuple(logits.shape), logits.device) if want_scores else None
    _lib.check(L.pram_seg_epilogue_f32(_p(logits), _p(lens), B, N, Cc, float(bg_threshold), _p(sc), _p(ids), _p(mask), _p(cnt), _st()),
               "pram_seg_epilogue_f32")
    return ids, mask, cnt, sc


def row_sort_desc(x: torch.Tensor):
    L = _lib.load()
    x = x.contiguous()
    rows, cols = _rows2d(x, "x")
    vals = torch.empty_like(x)
    idx = torch.empty(x.shape, device=x.device, dtype=_INT64)
    _lib.check(L.pram_row_sort_desc_f32(_p(x), cols, rows, cols, _p(vals), _p(idx), _st()), "pram_row_sort_desc_f32")
    return vals, idx


def row_top2(x: torch.Tensor, largest: bool, n_valid: Optional[int] = None, row_lens=None, col_lens=None):
    """x [B,M,ld] -> (best [B,M], second [B,M], index of best int64 [B,M])"""
    L = _lib.load()
    assert x.is_contiguous() and x.dim() == 3
    B, M, ld = x.shape
    N = n_valid or ld
    v0 = torch.zeros(B, M, device=x.device, dtype=torch.float32)
    v1 = torch.zeros(B, M, device=x.device, dtype=torch.float32)
    i0 = torch.full((B, M), -1, device=x.device, dtype=_INT64)
    _lib.check(L.pram_row_top2_f32(_p(x), ld, M * ld, _p(row_lens), _p(col_lens), B, M, N, int(largest), _p(v0), _p(v1), _p(i0), _st()),
               "pram_row_top2_f32")
    return v0, v1, i0


def proj_dist_top2(sim: torch.Tensor, kpts: torch.Tensor, proj_uv: torch.Tensor, rng: float, n_valid: Optional[int] = None):
    """sim [M, ld] fp32, kpts [M,2], proj_uv [2,N] -> (d0 [M], d1 [M], i0 int64 [M])"""
    L = _lib.load()
    assert sim.is_contiguous() and sim.dim() == 2
    M, ld = sim.shape
    N = n_valid or ld
    kpts, proj_uv = kpts.contiguous().float(), proj_uv.contiguous().float()
    d0 = torch.zeros(M, device=sim.device, dtype=torch.float32)
    d1 = torch.zeros(M, device=sim.device, dtype=torch.float32)
    i0 = torch.full((M,), -1, device=sim.device, dtype=_INT64)
    _lib.check(L.pram_proj_dist_top2_f32(_p(sim), ld, _p(kpts), _p(proj_uv), M, N, float(rng), _p(d0), _p(d1), _p(i0), _st()),
               "pram_proj_dist_top2_f32")
    return d0, d1, i0


def proj_dist_top2_f64uv(sim: torch.Tensor, kpts: torch.Tensor, proj_uv: torch.Tensor, rng: float, n_valid: int):
    """proj_dist_top2 with float64 projections [2, ldu] (first n_valid columns valid): the pixel error and its `>= rng` test
    are float64, as in the reference (float32 keypoints - float64 projections)."""
    L = _lib.load()
    assert sim.is_contiguous() and sim.dim() == 2 and proj_uv.dtype == torch.float64 and proj_uv.is_contiguous() and proj_uv.dim() == 2
    M, ld = sim.shape
    kpts = kpts.contiguous().float()
    d0 = torch.zeros(M, device=sim.device, dtype=torch.float32)
    d1 = torch.zeros(M, device=sim.device, dtype=torch.float32)
    i0 = torch.full((M,), -1, device=sim.device, dtype=_INT64)
    _lib.check(L.pram_proj_dist_top2_f64uv(_p(sim), ld, _p(kpts), _p(proj_uv), proj_uv.shape[1], M, int(n_valid), float(rng), _p(d0), _p(d1),
                                           _p(i0), _st()), "pram_proj_dist_top2_f64uv")
    return d0, d1, i0


def project_points(xyz: torch.Tensor, K: torch.Tensor, Tcw: torch.Tensor, im_w: float, im_h: float):
    """xyz [N,3], K [3,3], Tcw [4,4] float64 on the device -> (uvd [3,N] f64, mask int32 [N], keep_idx int32 [N], uv_keep f64 [2,N],
    count int32 [1]) — projection, frustum test and ordered compaction of singlemap3d.py:405-415; nothing synchronises."""
    L = _lib.load()
    for t, nm in ((xyz, "xyz"), (K, "K"), (Tcw, "Tcw")):
        _chk(t, nm, torch.float64)
    xyz, K, Tcw = xyz.contiguous(), K.contiguous(), Tcw.contiguous()
    n = xyz.shape[0]
    dev = xyz.device
    uvd = torch.empty(3, n, device=dev, dtype=torch.float64)
    mask = torch.empty(n, device=dev, dtype=torch.int32)
    keep = torch.empty(n, device=dev, dtype=torch.int32)
    uvk = torch.empty(2, max(n, 1), device=dev, dtype=torch.float64)
    count = torch.zeros(1, device=dev, dtype=torch.int32)
    _lib.check(L.pram_project_points_f64(_p(xyz), _p(K), _p(Tcw), n, float(im_w), float(im_h), _p(uvd), _p(mask), _p(keep), _p(uvk),
                                         _p(count), _st()), "pram_project_points_f64")
    return uvd, mask, keep, uvk, count


def seg_vote(sorted_vals: torch.Tensor, sorted_ids: torch.Tensor, topk: int):
    """Landmark vote over the sorted class lists of row_sort_desc ([N, C] each) -> device tensors
    (win_sid [topk], win_rank [topk], win_count [topk], n_win [1], tokens [topk, N], mean_score [topk])."""
    L = _lib.load()
    assert sorted_vals.is_contiguous() and sorted_ids.is_contiguous() and sorted_ids.dtype == _INT64
    n, c = sorted_vals.shape
    dev = sorted_vals.device
    sid = torch.zeros(topk, device=dev, dtype=torch.int32)
    rank = torch.zeros(topk, device=dev, dtype=torch.int32)
    cnt = torch.zeros(topk, device=dev, dtype=torch.int32)
    nwin = torch.zeros(1, device=dev, dtype=torch.int32)
    tokens = torch.zeros(topk, max(n, 1), device=dev, dtype=torch.int32)
    mean = torch.zeros(topk, device=dev, dtype=torch.float32)
    _lib.check(L.pram_seg_vote(_p(sorted_ids), _p(sorted_vals), n, c, int(topk), _p(sid), _p(rank), _p(cnt), _p(nwin), _p(tokens), _p(mean),
                               _st()), "pram_seg_vote")
    return sid, rank, cnt, nwin, tokens, mean


def frame_lut(mean, std, device) -> torch.Tensor:
    """[3, 256] fp32: what the reference's frame preparation makes of byte v in channel c — `img / 255` in float64 (numpy),
    `.float()`, then tvf.Normalize's `sub_(mean).div_(std)` in fp32 (localization/loc_by_rec_online.py:98-106) — computed with exactly
    those operations, on the host."""
    v = (torch.arange(256, dtype=torch.float64) / 255).float()[None].repeat(3, 1)
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1)
    return v.sub_(m).div_(s).contiguous().to(device)


def stage_frames(frames_u8: torch.Tensor, lut: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 [B, H, W, 3] (cv2 layout, on the device) -> normalised fp32 [B, 3, H, W] (pram_stage_frames_u8): the device half of the
    reference's per-frame preparation; `lut` from frame_lut()."""
    L = _lib.load()
    _chk(frames_u8, "frames", torch.uint8)
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.dim() == 4 and frames_u8.shape[-1] == 3
    B, H, W, _ = frames_u8.shape
    if not (lut.is_cuda and lut.device == frames_u8.device and lut.dtype == torch.float32 and lut.is_contiguous() and tuple(lut.shape) == (3, 256)):
        raise _lib.PramHipError("stage_frames: lut must be the contiguous fp32 [3, 256] table of frame_lut() on the frames' device")
    if out is None:
        out = torch.empty(B, 3, H, W, device=frames_u8.device, dtype=torch.float32)
    if not (out.is_cuda and out.device == frames_u8.device and out.is_contiguous() and tuple(out.shape) == (B, 3, H, W) and out.dtype == torch.float32):
        raise _lib.PramHipError("stage_frames: out must be a contiguous fp32 [B, 3, H, W] tensor on the frames' device")
    _lib.check(L.pram_stage_frames_u8(_p(frames_u8), _p(lut), _p(out), B, H, W, _st()), "pram_stage_frames_u8")
    return out


def pack_record(kpts: torch.Tensor, scores: torch.Tensor, landmark: Optional[torch.Tensor] = None,
                matches0: Optional[torch.Tensor] = None, mscores0: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> [B, k, 6] fp32: x, y, score, landmark id, match index, match score (pram_pack_record_f32); matches0 / mscores0 may cover
    only the first km <= k keypoints of every query."""
    L = _lib.load()
    kpts, scores = kpts.contiguous(), scores.contiguous()
    _chk(kpts, "kpts"), _chk(scores, "scores")
    B, k = scores.shape
    km = 0
    if matches0 is not None:
        matches0, mscores0 = matches0.contiguous(), mscores0.contiguous()
        assert matches0.dtype == _INT64 and mscores0.dtype == torch.float32 and matches0.shape == mscores0.shape
        km = matches0.shape[1]
    if landmark is not None:
        landmark = landmark.contiguous()
        assert landmark.dtype == torch.int32
    rec = torch.empty(B, k, 6, device=scores.device, dtype=torch.float32)
    _lib.check(L.pram_pack_record_f32(_p(kpts), _p(scores), _p(landmark), _p(matches0), _p(mscores0), B, k, km, _p(rec), _st()),
               "pram_pack_record_f32")
    return rec
