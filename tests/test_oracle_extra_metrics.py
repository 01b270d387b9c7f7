"""Oracle check (CPU): the extra-metric restatement (oracle/orc_core.c extra_distance, after
faiss/utils/extra_distances-inl.h VectorDistance<mt>) against float64 numpy formulas of the published definitions.
The reference repository holds no golden vectors for these metrics (its tests use L2 / inner product only), so this is
the pin the oracle has for them: definitions, ordering (Jaccard keeps the largest) and id translation."""
import numpy as np
import pytest

from oracle import oracle as orc

FORMULAS = {
    2: lambda a, b: np.abs(a - b).sum(-1),
    3: lambda a, b: np.abs(a - b).max(-1),
    20: lambda a, b: (np.abs(a - b) / (np.abs(a) + np.abs(b))).sum(-1),
    21: lambda a, b: np.abs(a - b).sum(-1) / np.abs(a + b).sum(-1),
    22: lambda a, b: 0.5 * (-(a * np.log((a + b) / 2 / a)) - (b * np.log((a + b) / 2 / b))).sum(-1),
    23: lambda a, b: np.minimum(a, b).sum(-1) / np.maximum(a, b).sum(-1),
}


@pytest.mark.parametrize("metric", sorted(FORMULAS))
@pytest.mark.parametrize("desc", ["Flat", "IDMap,Flat"])
def test_oracle_extra_metric_matches_definition(metric, desc):
    rs = np.random.RandomState(metric)
    d, n, nq, k = 13, 700, 6, 8
    xb = (rs.rand(n, d) * 0.95 + 0.05).astype(np.float32)
    xq = (rs.rand(nq, d) * 0.95 + 0.05).astype(np.float32)
    o = orc.Index(d, desc, metric)
    ids = np.arange(n, dtype=np.int64) * 3 + 5
    if desc == "Flat":
        o.add(xb)
        ids = np.arange(n, dtype=np.int64)
    else:
        o.add_with_ids(xb, ids)
    D, I = o.search(xq, k)
    ref = FORMULAS[metric](xq[:, None, :].astype(np.float64), xb[None].astype(np.float64))
    order = np.argsort(-ref if metric == 23 else ref, axis=1, kind="stable")[:, :k]
    assert np.array_equal(I, ids[order])
    assert np.allclose(D, np.take_along_axis(ref, order, 1), rtol=2e-5, atol=1e-6)


def test_oracle_lp_exponent():
    rs = np.random.RandomState(9)
    xb = rs.rand(300, 10).astype(np.float32)
    xq = rs.rand(3, 10).astype(np.float32)
    o = orc.Index(10, "Flat", 4)
    o.add(xb)
    D0, I0 = o.search(xq, 4)  # metric_arg 0 (what the glue leaves): |x-y|^0 = 1 per dimension
    assert np.all(D0 == 10.0) and np.array_equal(I0, np.tile(np.arange(4), (3, 1)))
    orc.set_metric_arg(3.0)
    try:
        D, I = o.search(xq, 4)
    finally:
        orc.set_metric_arg(0.0)
    ref = (np.abs(xq[:, None, :].astype(np.float64) - xb[None]) ** 3).sum(-1)
    order = np.argsort(ref, axis=1, kind="stable")[:, :4]
    assert np.array_equal(I, order) and np.allclose(D, np.take_along_axis(ref, order, 1), rtol=2e-5)
