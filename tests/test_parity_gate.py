"""The matcher part of bench.py's parity gate (bench.compare_matches) on constructed cases: the two recognised discrete decisions of
the reference — acceptance-threshold ties and sub-threshold mutual flips on a PROVEN near-tie (nets/gml.py:304-319) — are excused,
and nothing wider is.  The near-tie is built on purpose: a Sinkhorn assignment whose row 5 has two best columns that agree to a few
ulp, one of them mutual, the other not.  CPU part: the oracle alone; GPU part: the HIP Sinkhorn on the same matrix."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

THR = 0.2
ROW, C_MUT, C_OTHER, ROW_OTHER = 5, 3, 9, 7


def near_tie_scores(m=32, n=40, seed=3):
    """A score matrix whose assignment row ROW has its two best entries in columns C_MUT and C_OTHER, tuned (bisection on the
    fp32 oracle) until they agree to the last bits; column C_MUT's best row is ROW (mutual), column C_OTHER's best row is
    ROW_OTHER (not mutual).  Everything is far below the acceptance threshold (flat assignment, ~1 / n per entry)."""
    g = torch.Generator().manual_seed(seed)
    M = torch.randn(1, m, n, generator=g) * 0.3
    M[0, ROW, C_MUT] = 2.0
    M[0, ROW_OTHER, C_OTHER] = 3.0
    bin_score = torch.tensor(1.0)

    def gap(x):
        M[0, ROW, C_OTHER] = x
        P = R.sink_algorithm(M, bin_score, 20)[0, :-1, :-1]
        return float(P[ROW, C_OTHER] - P[ROW, C_MUT])

    lo, hi = torch.tensor(1.0), torch.tensor(4.0)
    assert gap(lo) < 0 < gap(hi)
    for _ in range(60):
        mid = (lo + hi) / 2
        if mid == lo or mid == hi:
            break
        if gap(mid) < 0:
            lo = mid
        else:
            hi = mid
    g_lo, g_hi = gap(lo), gap(hi)
    x = lo if abs(g_lo) <= abs(g_hi) else hi
    assert abs(gap(x)) <= 1e-6, (g_lo, g_hi)
    M[0, ROW, C_OTHER] = x
    return M, bin_score


def oracle_result(M, bin_score):
    P = R.sink_algorithm(M, bin_score, 20)
    i0, i1, s0, s1 = R.compute_matches(P, THR)
    return P[0, :-1, :-1], i0[0], s0[0]


def test_constructed_near_tie_is_what_it_claims():
    M, bs = near_tie_scores()
    P, m_ref, s_ref = oracle_result(M, bs)
    top = P[ROW].topk(2)
    assert set(top.indices.tolist()) == {C_MUT, C_OTHER} and float(top.values[0] - top.values[1]) <= 1e-6
    assert int(P[:, C_MUT].argmax()) == ROW and int(P[:, C_OTHER].argmax()) == ROW_OTHER
    assert float(P.max()) < THR and bool((m_ref == -1).all())          # all sub-threshold: indices are -1 whoever wins


def test_gate_excuses_exactly_the_proven_mutual_flip():
    M, bs = near_tie_scores()
    P, m_ref, s_ref = oracle_result(M, bs)
    # the other implementation names the other partner of the tie: row ROW's mutual flag flips, its score goes p <-> 0
    s_got = s_ref.clone()
    s_got[ROW] = 0.0 if float(s_ref[ROW]) != 0.0 else float(P[ROW, C_MUT])
    ok, rep = bench.compare_matches(m_ref.clone(), s_got, m_ref, s_ref, THR, P)
    assert ok and rep["mutual_flips_below_threshold"] == 1 and rep["mutual_flips"][0]["i"] == ROW
    assert rep["mutual_flips"][0]["row_top2_gap"] <= bench.MUTUAL_FLIP_GAP and rep["scores_maxdiff"] < 1e-6
    # ... without the oracle's assignment matrix there is no proof: not excused (AdaGML's restatement hands none out)
    ok, rep = bench.compare_matches(m_ref.clone(), s_got, m_ref, s_ref, THR, None)
    assert not ok and rep["mutual_flips_below_threshold"] == 0 and rep["scores_maxdiff"] >= 1e-3
    # ... the same p <-> 0 difference on a row with a clear winner is a bug, not a tie
    clear = ROW_OTHER
    assert float(P[clear].topk(2).values[0] - P[clear].topk(2).values[1]) > 1e-3 and float(s_ref[clear]) > 0.0
    s_bad = s_ref.clone()
    s_bad[clear] = 0.0
    ok, rep = bench.compare_matches(m_ref.clone(), s_bad, m_ref, s_ref, THR, P)
    assert not ok and rep["mutual_flips_below_threshold"] == 0
    # ... and so is a flip ABOVE the threshold (it would change a reported match), or more flips than MAX_MUTUAL_FLIPS
    s_hi, s_hi_ref = s_ref.clone(), s_ref.clone()
    s_hi_ref[ROW], s_hi[ROW] = 0.5, 0.0
    ok, _ = bench.compare_matches(m_ref.clone(), s_hi, m_ref, s_hi_ref, THR, P)
    assert not ok


def test_gate_threshold_ties_and_plain_mismatches():
    m_ref = torch.tensor([4, -1, 2, 7])
    s_ref = torch.tensor([0.9, 0.0, 0.2000004, 0.5])
    # the same candidate straddles `score > 0.2`: excused, counted
    m_got, s_got = torch.tensor([4, -1, -1, 7]), torch.tensor([0.9, 0.0, 0.1999996, 0.5])
    ok, rep = bench.compare_matches(m_got, s_got, m_ref, s_ref, THR, None)
    assert ok and rep["threshold_ties"] == 1 and rep["indices_identical"]
    # a different partner, or a dropped match away from the threshold, is a mismatch
    ok, rep = bench.compare_matches(torch.tensor([4, -1, 3, 7]), s_ref, m_ref, s_ref, THR, None)
    assert not ok and not rep["indices_identical"]
    ok, rep = bench.compare_matches(torch.tensor([4, -1, 2, -1]), s_ref, m_ref, s_ref, THR, None)
    assert not ok
    # scores beyond the bar
    ok, rep = bench.compare_matches(m_ref, s_ref + torch.tensor([2e-3, 0, 0, 0]), m_ref, s_ref, THR, None)
    assert not ok and rep["scores_maxdiff"] >= 1e-3


@pytest.mark.gpu
def test_hip_sinkhorn_on_the_constructed_near_tie(hip_lib):
    """The HIP Sinkhorn + compute_matches on the near-tie: whichever partner it names, the gate passes — by identity or by ONE
    proven flip on row ROW — and every other row agrees to 1e-6."""
    from pram_amd import ops
    dev = torch.device("cuda:0")
    M, bs = near_tie_scores()
    P, m_ref, s_ref = oracle_result(M, bs)
    r = ops.sinkhorn_match(M.to(dev).contiguous(), bs.to(dev), 20, THR, want_p=True)
    torch.cuda.synchronize()
    m_got, s_got = r["matches0"][0].cpu(), r["matching_scores0"][0].cpu()
    assert float((r["p"][0, :-1, :-1].cpu() - P).abs().max()) < 1e-6
    ok, rep = bench.compare_matches(m_got, s_got, m_ref, s_ref, THR, P)
    assert ok, rep
    assert rep["mutual_flips_below_threshold"] <= 1 and rep["indices_identical"]
    if rep["mutual_flips_below_threshold"]:
        assert rep["mutual_flips"][0]["i"] == ROW and rep["mutual_flips"][0]["row_top2_gap"] <= 1e-6
    others = torch.arange(len(s_ref)) != ROW
    assert float((s_got - s_ref)[others].abs().max()) < 1e-6
