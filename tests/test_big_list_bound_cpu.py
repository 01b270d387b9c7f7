"""CPU restatement of the bound behind the coarse filter's big lists (csrc/flat_collect.hip "lists beyond 128 entries", round 6): P
disjoint row ranges, 16 row classes each (class = row & 15), per range the r-th best of its 16 class maxima of the COARSE scores s
(|s - exact| <= E), T = the worst of the ranges' values.  Claims checked on random data with adversarial noise:
  1. at least P * r distinct rows have s >= T (so the exact k-th best value, k <= P * r, is >= T - E);
  2. every row of the exact top-k has s >= T - 2E -- the frozen scan's pass condition -- whatever part of the rows the ranges cover;
  3. the same with rows excluded (IDSelector: they are no evidence and no candidates) and with ties."""
import numpy as np
import pytest


def _bound(s, ranges, r, allowed):
    T = np.inf
    for lo, hi in ranges:
        cm = np.full(16, -np.inf, dtype=np.float32)
        for c in range(16):
            rows = np.arange(lo + ((c - lo) % 16), hi, 16)
            rows = rows[allowed[rows]]
            if rows.size:
                cm[c] = s[rows].max()
        srt = np.sort(cm)[::-1]
        T = min(T, srt[r - 1])  # the r-th best class maximum (-inf: fewer than r classes with a row)
    return T


@pytest.mark.parametrize("n,k,frac,ties,sel", [(40_000, 200, 1.0, False, False), (40_000, 1000, 0.25, False, False), (60_000, 2048, 0.25, True, False),
                                                  (30_000, 129, 0.5, False, True), (50_000, 700, 1.0, True, True)])
def test_range_bound_keeps_every_row_of_the_result(n, k, frac, ties, sel):
    rs = np.random.RandomState(n + k)
    exact = rs.randn(n).astype(np.float32)
    if ties:
        exact = np.round(exact * 8) / 8  # many exact ties, also at the k-th value
    E = np.float32(0.05)
    # the coarse score: anywhere within E of the exact value, pushed against the claim (good rows low, bad rows high)
    order = np.argsort(-exact, kind="stable")
    noise = rs.uniform(-1, 1, n).astype(np.float32) * E
    noise[order[:k]] = -E * rs.uniform(0.5, 1.0, k).astype(np.float32)
    noise[order[k : 4 * k]] = E * rs.uniform(0.5, 1.0, 3 * k).astype(np.float32)
    s = exact + noise
    allowed = np.ones(n, dtype=bool)
    if sel:
        allowed = rs.rand(n) < 0.6
    per = 8
    P = -(-k // per)
    r = -(-k // P)
    stride = (n // P) // 64 * 64
    length = max(64, int(stride * frac) // 64 * 64)
    ranges = [(p * stride, p * stride + length) for p in range(P)]
    T = _bound(s, ranges, r, allowed)
    if not np.isfinite(T):
        pytest.skip("a range with fewer than r admissible classes: no bound (the kernel passes everything then)")
    # 1. enough distinct rows at least that good
    assert int(np.count_nonzero(allowed & (s >= T))) >= P * r >= k
    # 2. / 3. the exact top-k among the admissible rows all pass T - 2E
    adm = np.flatnonzero(allowed)
    topk = adm[np.argsort(-exact[adm], kind="stable")[:k]]
    kth = exact[topk[-1]]
    assert kth >= T - E - 1e-6
    tied_or_better = adm[exact[adm] >= kth]  # every row tied with the k-th value must be a candidate too (FAISS's tie rules pick among them)
    assert np.all(s[tied_or_better] >= T - 2 * E - 1e-6)
