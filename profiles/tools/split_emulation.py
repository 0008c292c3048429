"""Numerical feasibility probe for the split-fp16 ("f16x3") MFMA path — CPU only, test infrastructure.

Every matrix product of the oracle (F.linear, F.conv2d with groups == 1, the attention / distance einsums) is
replaced by an emulation of what the split kernels compute:

    x = hi + lo,  hi = fp16(x * s),  lo = fp16(x * s - hi)        (s: power-of-two scale, exact)
    a . b  ~=  (a_hi . b_hi + a_hi . b_lo + a_lo . b_hi) / (s_a s_b)        (fp32 accumulate; lo.lo dropped)

and the end-to-end outputs are compared with the plain-fp32 oracle on the same inputs.  Products of two fp16
values are exact in fp32, so torch's CPU fp32 matmul on the fp16-valued operands differs from the MFMA only in
summation order.  Usage:  python profiles/tools/split_emulation.py [segnet|gml|adagml|sfd2 ...] [--n 2048]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ref_cpu as R  # noqa: E402
from pram_amd import weights as W  # noqa: E402

MODE = {"on": False, "terms": 3, "act_scale": 16.0}
_real_linear, _real_conv2d, _real_einsum = F.linear, F.conv2d, torch.einsum


def _pow2_scale_for(w: torch.Tensor, target: float = 16384.0) -> float:
    m = float(w.abs().max())
    if m == 0.0:
        return 1.0
    import math
    return 2.0 ** math.floor(math.log2(target / m))


def split(x: torch.Tensor, s: float):
    xs = x * s
    hi = xs.half().float()
    lo = (xs - hi).half().float()
    assert torch.isfinite(hi).all(), "fp16 overflow in the hi part"
    return hi, lo


def _combine(f, a, b, sa, sb):
    ah, al = split(a, sa)
    bh, bl = split(b, sb)
    out = f(ah, bh)
    if MODE["terms"] >= 3:
        out = out + (f(ah, bl) + f(al, bh))
    return out / (sa * sb)


def linear(x, w, b=None):
    if not MODE["on"]:
        return _real_linear(x, w, b)
    y = _combine(lambda p, q: _real_linear(p, q), x, w, MODE["act_scale"], _pow2_scale_for(w))
    return y if b is None else y + b


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if not MODE["on"] or groups != 1:
        return _real_conv2d(x, w, b, stride, padding, dilation, groups)
    y = _combine(lambda p, q: _real_conv2d(p, q, None, stride, padding, dilation, 1), x, w, MODE["act_scale"], _pow2_scale_for(w))
    return y if b is None else y + b.view(1, -1, 1, 1)


def einsum(eq, a, b):
    if not MODE["on"]:
        return _real_einsum(eq, a, b)
    # softmax probabilities (in [0, 1]) get the large scale, everything else the activation scale
    pa = bool(a.min() >= 0 and a.max() <= 1.0)
    pb = bool(b.min() >= 0 and b.max() <= 1.0)
    sa = 16384.0 if pa else MODE["act_scale"]
    sb = 16384.0 if pb else MODE["act_scale"]
    if MODE.get("p_hi_only") and (pa != pb):
        # P V with the probabilities rounded to ONE fp16 (two MFMAs: P_hi V_hi + P_hi V_lo); the softmax normalisation uses the
        # same rounded probabilities, as the kernel would
        P, V, sp, sv = (a, b, sa, sb) if pa else (b, a, sb, sa)
        ph = (P * sp).half().float()
        vh, vl = split(V, sv)
        f = (lambda x, y: _real_einsum(eq, x, y)) if pa else (lambda x, y: _real_einsum(eq, y, x))
        out = (f(ph, vh) + f(ph, vl)) / (sp * sv)
        # renormalise: sum of the rounded row / sum of the exact row (rows of P sum to 1)
        red = P.dim() - 1 if pa else None
        ratio = (ph / sp).sum(-1, keepdim=True) if pa else None
        if pa and eq == "bhij,bhjd->bhid":
            return out / ratio
        if (not pa) and eq == "bhji,bhid->bhjd":      # a10 = softmax(sim^T): P = b? (cross block writes einsum(a10, v0))
            return out
        return out
    return _combine(lambda p, q: _real_einsum(eq, p, q), a, b, sa, sb)


F.linear, F.conv2d, torch.einsum = linear, conv2d, einsum


def both(fn):
    MODE["on"] = False
    ref = fn()
    MODE["on"] = True
    got = fn()
    MODE["on"] = False
    return ref, got


def sd_for(name, module):
    return W.make_state_dict(name, module.state_dict(), seed=7)


def run_segnet(n, nc):
    from pram_amd.nets.load_segnet import load_segnet
    sd = sd_for("segnetvit", load_segnet('segnetvit', nc, 256, 15, 1024))
    desc = W.normal(11, "emu/segdesc", (1, n, 256), 0.05)
    kp = torch.stack([torch.floor(W.uniform(11, "emu/kx", (n,), 4.0, 636.0)), torch.floor(W.uniform(11, "emu/ky", (n,), 4.0, 476.0))], -1)[None]
    ref, got = both(lambda: R.segnetvit_forward(sd, desc, kp, (1, 3, 480, 640)))
    d = (ref - got).abs().max().item()
    am = (ref.argmax(-1) == got.argmax(-1)).float().mean().item()
    print(f"segnetvit N={n} nc{nc}: logits |d|max {d:.3e} (|ref|max {ref.abs().max():.2f}), argmax agreement {am:.6f}")


def run_gml(n, ada=False):
    from pram_amd.nets.adagml import AdaGML
    from pram_amd.nets.gml import GML
    name = "adagml" if ada else "gml"
    sd = sd_for(name, AdaGML({}) if ada else GML({}))
    pair = W.synthetic_match_pair(5, n, n)
    data = {k: v[None] for k, v in pair.items() if k != "gt"}
    data["image_shape0"] = data["image_shape1"] = (1, 3, 640, 480)
    f = R.adagml_produce_matches if ada else R.gml_produce_matches
    ref, got = both(lambda: f(sd, data))
    same = (ref["matches0"] == got["matches0"]).float().mean().item()
    d = (ref["matching_scores0"] - got["matching_scores0"]).abs().max().item()
    print(f"{name} {n}x{n}: matches0 identical {same:.6f} ({int((ref['matches0'] >= 0).sum())} matches), scores |d|max {d:.3e}")


def run_sfd2(h, w, k):
    from pram_amd.nets.sfd2 import ResNet4x
    sd = sd_for("sfd2", ResNet4x())
    img = W.synthetic_image(3, h, w)[None]
    ref, got = both(lambda: R.sfd2_extract_local_global(sd, img, max_keypoints=k, min_keypoints=128, per_image_fallback=True))
    ds = (ref["score_map"] - got["score_map"]).abs().max().item()
    dd = (ref["desc_map"] - got["desc_map"]).abs().max().item()
    a, b = ref["keypoints"][0], got["keypoints"][0]
    sa = {(int(x), int(y)) for x, y in a.tolist()}
    sb = {(int(x), int(y)) for x, y in b.tolist()}
    order = a.shape == b.shape and bool((a == b).all())
    print(f"sfd2 {w}x{h} k={k}: score_map |d|max {ds:.3e}, desc_map |d|max {dd:.3e}, keypoint sets equal {sa == sb} "
          f"({len(sa & sb)}/{len(sa)}), same order {order}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["segnet", "gml", "sfd2"])
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--nc", type=int, default=113)
    ap.add_argument("--terms", type=int, default=3, help="3 = hi.hi + hi.lo + lo.hi ; 1 = plain fp16 (for contrast)")
    ap.add_argument("--act-scale", type=float, default=16.0)
    ap.add_argument("--p-hi-only", action="store_true", help="P V from fp16(P) alone: two MFMAs per product instead of three")
    args = ap.parse_args()
    MODE["terms"], MODE["act_scale"], MODE["p_hi_only"] = args.terms, args.act_scale, args.p_hi_only
    torch.set_num_threads(8)
    with torch.no_grad():
        for wh in args.what:
            if wh == "segnet":
                run_segnet(args.n, args.nc)
            elif wh == "gml":
                run_gml(args.n)
            elif wh == "adagml":
                run_gml(args.n, ada=True)
            elif wh == "sfd2":
                run_sfd2(240, 320, 1024) if args.n < 2048 else run_sfd2(480, 640, 2048)
