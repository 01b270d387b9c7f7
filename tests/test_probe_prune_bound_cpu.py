"""The probe-pruning rule of the IVF coarse-filter path (csrc/ivf_collect.hip ivf_probe_prune_kernel), restated in numpy and confronted
with what it claims: a probed list is pruned only if EVERY row of it is strictly farther (in IVFFlatScanner's COMPUTED f32 distance)
than k rows of the lists in front of the witness rank -- so leaving it out cannot change a label or a distance of IndexIVF::search
(faiss/IndexIVF.cpp search_preassigned, reached from src/faiss_extension.cpp:631).  The arithmetic is emulated the way FAISS computes it:
coarse distances by the BLAS-branch formula (xn + cn) - 2 ip in float32 with a k-ordered fma chain for ip, scanner distances as the
k-ordered float32 chain of (x_k - y_k)^2.  Data with offsets, large norms, tiny lists and giant radii.  CPU only; the device kernel is
compared bit for bit against the unpruned search in tests/test_ivf_probe_prune_gpu.py."""
import numpy as np
import pytest

U = 2.0**-24


def chain_ip(x, c):
    """fvec_inner_product as a k-ordered float32 fma chain (each product exact, one rounding per step)"""
    acc = np.float32(0.0)
    for k in range(len(x)):
        acc = np.float32(np.float64(x[k]) * np.float64(c[k]) + np.float64(acc))
    return acc


def chain_l2(x, y):
    """IVFFlatScanner's fvec_L2sqr: t = x_k - y_k rounded, acc = fma(t, t, acc)"""
    acc = np.float32(0.0)
    for k in range(len(x)):
        t = np.float32(x[k] - y[k])
        acc = np.float32(np.float64(t) * np.float64(t) + np.float64(acc))
    return acc


def norm2(x):
    return chain_ip(x, x)


def coarse_blas(x, cents):
    """exhaustive_L2sqr_blas: dis = (xn + cn) - 2 ip, clamped at 0, all float32"""
    xn = norm2(x)
    out = np.empty(len(cents), dtype=np.float32)
    for j, c in enumerate(cents):
        d = np.float32(np.float32(xn + norm2(c)) - np.float32(2.0) * chain_ip(x, c))
        out[j] = d if d > 0 else np.float32(0.0)
    return out


def prune_rule(x, cD, cI, k, cn, list_max, sizes, d):
    """the kernel's decision for one query: True = probe p is pruned (all in double, as on the device)"""
    eps = 2.0 * (d + 2.0) * U
    nx = np.sqrt(float((x.astype(np.float64) ** 2).sum())) * (1.0 + 1e-9)
    np_ = len(cI)
    up = np.full(np_, np.inf)
    lo2 = np.full(np_, -1.0)
    for p in range(np_):
        j = cI[p]
        if j < 0:
            continue
        cd = float(cD[p])
        nc = np.sqrt(max(float(cn[j]), 0.0)) * (1.0 + 1e-7)
        ec = 2.0 * (d + 2.0) * U * (nx + nc) ** 2
        R = np.sqrt(float(list_max[j])) * 1.0001
        up[p] = (np.sqrt(cd + ec) + R) ** 2 * (1.0 + eps)
        gap = np.sqrt(max(cd - ec, 0.0)) - R
        if gap > 0:
            lo2[p] = gap * gap * (1.0 - eps)
    W, m, cum = 0.0, 0, 0
    while m < np_ and cum < k:
        cum += int(sizes[cI[m]]) if cI[m] >= 0 else 0
        W = max(W, up[m])
        m += 1
    can = cum >= k and np.isfinite(W)
    return np.array([can and p >= m and lo2[p] > W for p in range(np_)]), m, W


def make_case(kind, rs, d, nlist, n):
    if kind == "separated":
        cent = rs.randn(nlist // 3, d) * 1.0
        xb = cent[rs.randint(0, len(cent), n)] + 0.1 * rs.randn(n, d)
    elif kind == "offset":  # large norms against small spreads: the formula's absolute error matters
        cent = 50.0 + rs.randn(nlist // 3, d) * 2.0
        xb = cent[rs.randint(0, len(cent), n)] + 0.3 * rs.randn(n, d)
    elif kind == "giant":  # a few cells swallow many clusters (large radii), others are tiny
        cent = rs.randn(nlist * 2, d) * 1.5
        xb = cent[rs.randint(0, len(cent), n)] + 0.05 * rs.randn(n, d)
    else:  # uniform: hardly anything can be pruned -- the rule must not prune wrongly either
        xb = rs.rand(n, d)
    return xb.astype(np.float32)


@pytest.mark.parametrize("kind", ["separated", "offset", "giant", "uniform"])
@pytest.mark.parametrize("k", [1, 10, 40])
def test_pruned_lists_hold_no_row_of_the_result(kind, k):
    rs = np.random.RandomState(sum(map(ord, kind)) * 101 + k)
    d, nlist, n, nq, nprobe = 24, 24, 900, 12, 12
    xb = make_case(kind, rs, d, nlist, n)
    # centroids: a few Lloyd steps from random rows (any centroids do: the rule only uses the assignment it is given)
    cents = xb[rs.choice(n, nlist, replace=False)].copy()
    for _ in range(4):
        a = ((xb[:, None, :].astype(np.float64) - cents[None].astype(np.float64)) ** 2).sum(-1).argmin(1)
        for j in range(nlist):
            if (a == j).any():
                cents[j] = xb[a == j].mean(0).astype(np.float32)
    assign = np.array([int(coarse_blas(y, cents).argmin()) for y in xb])
    sizes = np.bincount(assign, minlength=nlist)
    # list_max: the largest ||y - c||^2 of every list, float32 chain on the float32 residual (csrc/ivf_collect.hip ivf_rows_to_bf16_kernel)
    list_max = np.zeros(nlist, dtype=np.float32)
    for i, y in enumerate(xb):
        r = (y - cents[assign[i]]).astype(np.float32)
        list_max[assign[i]] = max(list_max[assign[i]], norm2(r))
    cn = np.array([norm2(c) for c in cents], dtype=np.float32)
    xq = (xb[rs.choice(n, nq)] + (0.05 * rs.randn(nq, d)).astype(np.float32)).astype(np.float32)
    pruned_total = 0
    for x in xq:
        cd_all = coarse_blas(x, cents)
        order = np.lexsort((np.arange(nlist), cd_all))[:nprobe]  # (dis, id) ascending, as the quantiser returns
        cD, cI = cd_all[order], order
        pr, m, W = prune_rule(x, cD, cI, k, cn, list_max, sizes, d)
        pruned_total += int(pr.sum())
        # the result IndexIVF::search would compute over ALL probed lists, in the scanner's arithmetic
        rows = [i for i in range(n) if assign[i] in set(cI.tolist())]
        dist = {i: chain_l2(x, xb[i]) for i in rows}
        kth = sorted(dist.values())[min(k, len(rows)) - 1] if rows else None
        for p in np.nonzero(pr)[0]:
            assert p >= m >= 1
            in_list = [i for i in rows if assign[i] == cI[p]]
            # every row of a pruned list is STRICTLY worse than the k-th computed result (ties cannot involve it either)
            assert all(dist[i] > kth for i in in_list), (kind, k, p, cI[p])
            # ... because the lists in front of the witness rank hold >= k rows and W bounds every computed distance of theirs from above
            front = [i for i in rows if assign[i] in set(cI[:m].tolist())]
            assert len(front) >= k and all(float(dist[i]) <= W for i in front), (kind, k, m)
    if kind in ("separated", "offset"):
        assert pruned_total > 0, "the rule never fired on data it is made for"
