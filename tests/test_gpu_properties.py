"""GPU: size-independent properties at BASELINE.json's full sizes (640x480 frames, 2048 keypoints) — things that must hold
whatever the weights are, checked where an element-by-element CPU oracle would take minutes."""
import pytest
import torch

from pram_amd import weights as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(hip_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_nms_idempotent_and_sparse_at_full_frame(dev):
    from pram_amd import ops
    s = torch.stack([W.uniform(40 + i, "prop/score", (480, 640), 0.0, 1.0) for i in range(4)]).to(dev)
    once = ops.simple_nms(s, 4)
    assert torch.equal(ops.simple_nms(once, 4), once)                       # survivors are still maxima of what is left
    kept = once > 0
    assert torch.equal(once[kept], s[kept]) and 0.005 < kept.float().mean().item() < 0.05
    # no two survivors within the suppression radius of each other (Chebyshev distance <= 4)
    dil = torch.nn.functional.max_pool2d(kept.float()[:, None], 9, 1, 4)[:, 0]
    cnt = torch.nn.functional.avg_pool2d(kept.float()[:, None], 9, 1, 4, divisor_override=1)[:, 0]
    assert bool(((cnt <= 1) | ~kept).all()) and bool((dil >= kept.float()).all())


def test_selection_sorted_bounded_and_inside_border_at_full_frame(dev):
    from pram_amd import ops
    s = torch.stack([W.uniform(50 + i, "prop/sel", (480, 640), 0.0, 1.0) ** 4 for i in range(4)]).to(dev)
    nms = ops.simple_nms(s, 4)
    kp, sc, cnt = ops.select_keypoints(nms, 0.005, 128, 4, 2048, fallback_ref=-1)
    for b in range(4):
        n = int(cnt[b])
        assert n == 2048
        k, v = kp[b, :n], sc[b, :n]
        assert bool((v[:-1] >= v[1:]).all()) and bool((v >= 0.005).all())
        flat = k[:, 1] * 640 + k[:, 0]
        tie = v[:-1] == v[1:]
        assert bool((flat[:-1][tie] < flat[1:][tie]).all())                   # canonical tie order
        assert bool(((k[:, 0] >= 4) & (k[:, 0] < 636) & (k[:, 1] >= 4) & (k[:, 1] < 476)).all())
        assert torch.equal(v, nms[b, k[:, 1].long(), k[:, 0].long()]) and flat.unique().numel() == n
        # nothing better was left behind
        rest = nms[b].clone()
        rest[k[:, 1].long(), k[:, 0].long()] = 0
        rest[:4] = 0; rest[-4:] = 0; rest[:, :4] = 0; rest[:, -4:] = 0
        assert rest.max() <= v[-1]


def test_attention_linear_in_v_and_blind_to_key_order_at_2048(dev):
    from pram_amd import ops
    S, T, Hh = 2, 2048, 4
    g = lambda tag: W.normal(61, tag, (S * T, 256), 1.0).to(dev)
    q, k, v1, v2 = g("prop/q"), g("prop/k"), g("prop/v1"), g("prop/v2")
    a1 = ops.attention(q, k, v1, S, Hh, T, T, 0.125)
    a2 = ops.attention(q, k, v2, S, Hh, T, T, 0.125)
    mix = ops.attention(q, k, 0.75 * v1 - 2.0 * v2, S, Hh, T, T, 0.125)
    assert H.maxdiff(mix, 0.75 * a1 - 2.0 * a2) < 2e-5                          # softmax weights do not depend on V
    # every output is a convex combination of the values of its own sequence and head
    vmin = v1.view(S, T, Hh, 64).amin(1, keepdim=True)
    vmax = v1.view(S, T, Hh, 64).amax(1, keepdim=True)
    o = a1.view(S, T, Hh, 64)
    assert bool(((o >= vmin - 1e-5) & (o <= vmax + 1e-5)).all())
    # permuting the keys (with their values) inside each sequence changes the summation order only
    perm = torch.argsort(W.uniform(62, "prop/perm", (T,), 0.0, 1.0)).to(dev)
    idx = (torch.arange(S, device=dev)[:, None] * T + perm[None]).reshape(-1)
    assert H.maxdiff(ops.attention(q, k[idx], v1[idx], S, Hh, T, T, 0.125), a1) < 2e-5


def test_matches_are_mutual_and_thresholded_at_2048(dev):
    from pram_amd import ops
    B, N = 2, 2048
    M = W.normal(63, "prop/dist", (B, N, N), 1.0)
    for b in range(B):                                    # plant a noisy permutation so that a good share matches
        p = torch.argsort(W.uniform(64 + b, "prop/p", (N,), 0.0, 1.0))
        M[b, torch.arange(N), p] += 12.0
    r = ops.sinkhorn_match(M.to(dev).contiguous(), torch.tensor(1.0, device=dev), 20, 0.2)
    m0, m1, s0, s1 = r["matches0"], r["matches1"], r["matching_scores0"], r["matching_scores1"]
    assert int((m0 >= 0).sum()) > 0.8 * B * N
    for b in range(B):
        i = torch.nonzero(m0[b] >= 0)[:, 0]
        assert torch.equal(m1[b, m0[b, i]], i)                                  # mutual
        j = torch.nonzero(m1[b] >= 0)[:, 0]
        assert torch.equal(m0[b, m1[b, j]], j)
        assert bool((s0[b, i] > 0.2).all()) and torch.equal(s1[b, m0[b, i]], s0[b, i])
        assert bool((s0[b] >= 0).all())


def test_conv_is_linear_in_its_input_at_full_frame(dev):
    from pram_amd import ops
    x1 = W.normal(65, "prop/x1", (2, 240, 320, 64), 1.0).to(dev)
    x2 = W.normal(65, "prop/x2", (2, 240, 320, 64), 1.0).to(dev)
    w = W.normal(65, "prop/w", (128, 3, 3, 64), 0.05).to(dev)
    f = lambda x: ops.conv2d_nhwc(x, w, None, None, None, ks=3, stride=1, relu=False)
    assert H.maxdiff(f(x1 + 0.5 * x2), f(x1) + 0.5 * f(x2)) < 2e-5
    # zero padding: shifting the frame by one pixel shifts the interior of the result
    y = f(x1)
    x1s = torch.zeros_like(x1)
    x1s[:, 1:] = x1[:, :-1]
    assert H.maxdiff(f(x1s)[:, 3:-2], y[:, 2:-3]) < 1e-6
