"""Pins the CPU oracle against every golden vector the reference holds for the hot path
(SURVEY.md 8c): test/sql/faiss.test, faiss2.test, faiss3.test, faiss4.test, faiss7.test,
'faiss_add_ids_with_train copy.test'.  All are d=8, N=1000, nq=10, k=2, inner product."""
import numpy as np
import pytest

from helpers import bitmap_from_ids, goldens, load_csv_fixtures
from oracle import oracle as orc

# sqllogictest compares floats approximately; FAISS's own SIMD summation order is
# build-dependent, so the pin is labels exact + distances to 1e-6 relative.
GOLD_RTOL = 1e-6


def test_flat_ip_matches_faiss_test():
    """faiss.test:7-38 -- faiss_create('flat8', 8, 'Flat'); add 1000; search k=2 -> 20 distances"""
    ids, xb, _, xq = load_csv_fixtures()
    ix = orc.Index(8, "Flat")  # default metric = INNER_PRODUCT
    for i0 in range(0, 1000, 256):  # arrives in DataChunks
        ix.add(xb[i0 : i0 + 256])
    assert ix.ntotal == 1000
    D, I = ix.search(xq, 2)
    g = np.array(goldens()["flat_ip_k2_distances"], dtype=np.float64)
    np.testing.assert_allclose(D.reshape(-1), g, rtol=GOLD_RTOL)


def test_idmap_flat_labels_match_faiss2_test():
    """faiss2.test:7-42 -- IDMap,Flat with explicit labels; label multiset of the join"""
    ids, xb, _, xq = load_csv_fixtures()
    ix = orc.Index(8, "IDMap,Flat")
    ix.add_with_ids(xb, ids)
    D, I = ix.search(xq, 2)
    assert sorted(I.reshape(-1).tolist()) == goldens()["idmap_flat_ip_k2_labels_multiset"]


def test_idmap_flat_rank_label_distance_match_faiss3_test():
    """faiss3.test:22-44"""
    ids, xb, _, xq = load_csv_fixtures()
    ix = orc.Index(8, "IDMap,Flat")
    ix.add_with_ids(xb, ids)
    D, I = ix.search(xq, 2)
    g = goldens()["idmap_flat_ip_k2"]
    got = [(j, int(I[q, j]), float(D[q, j])) for q in range(10) for j in range(2)]
    assert [(r, l) for r, l, _ in got] == [(r, l) for r, l, _ in g]
    np.testing.assert_allclose([d for _, _, d in got], [d for _, _, d in g], rtol=GOLD_RTOL)


def test_bitmap_filter_matches_faiss3_test():
    """faiss3.test:46-68 -- faiss_search_filter(..., 'column0>100', 'column0', 'training'):
    IDSelectorBitmap over external ids (src/faiss_extension.cpp:959)."""
    ids, xb, _, xq = load_csv_fixtures()
    ix = orc.Index(8, "IDMap,Flat")
    ix.add_with_ids(xb, ids)
    bm = bitmap_from_ids(ids, ids > 100)
    D, I = ix.search(xq, 2, sel=("bitmap", bm))
    g = goldens()["idmap_flat_ip_k2_filter_id_gt_100"]
    got_labels = [int(I[q, j]) for q in range(10) for j in range(2)]
    assert got_labels == [l for _, l, _ in g]
    # golden distances are round(distance, 5)
    np.testing.assert_allclose([float(D[q, j]) for q in range(10) for j in range(2)], [d for _, _, d in g], atol=1e-5)
    # same result through IDSelectorBatch (faiss_search_filter_set, :1008)
    D2, I2 = ix.search(xq, 2, sel=("batch", ids[ids > 100]))
    assert np.array_equal(I, I2) and np.array_equal(D, D2)


def test_add_with_ids_on_plain_flat_error_text():
    """faiss4.test:19-22 / faiss6.test:27-30: the glue greps this substring (:523)"""
    ids, xb, _, _ = load_csv_fixtures()
    ix = orc.Index(8, "Flat")
    with pytest.raises(orc.OracleError, match="add_with_ids not implemented for this type of index"):
        ix.add_with_ids(xb, ids)
    ix.add(xb)  # the follow-up plain add succeeds (faiss4.test:24-25)
    assert ix.ntotal == 1000


def test_small_index_pads_with_minus_one():
    """faiss7.test:15-25 -- N=1 < k=2 with a filter that excludes the only row"""
    ix = orc.Index(2, "IDMap,Flat")
    ix.add_with_ids(np.array([[0.0040321066, 0.023423655]], np.float32), np.array([231]))
    q = np.array([[-0.04529257, 0.024853613]], np.float32)
    D, I = ix.search(q, 2)
    assert I.tolist() == [[231, -1]]
    assert D[0, 1] == -np.finfo(np.float32).max  # CMin neutral
    bm = bitmap_from_ids(np.array([231]), np.array([231 % 2 == 0]))
    D, I = ix.search(q, 2, sel=("bitmap", bm))
    assert I.tolist() == [[-1, -1]]


def test_idmap_ivf1_train_then_add_single_vector():
    """'faiss_add_ids_with_train copy.test':7-11 -- IDMap,IVF1,Flat, one vector: train(1)+add_with_ids"""
    ix = orc.Index(2, "IDMap,IVF1,Flat")
    assert not ix.is_trained
    x = np.array([[0.0040321066, 0.023423655]], np.float32)
    ix.train(x)
    assert ix.is_trained
    ix.add_with_ids(x, np.array([231]))
    D, I = ix.search(x, 1)
    assert I.tolist() == [[231]]


def test_train_too_few_points_error_text():
    """substring matched at src/faiss_extension.cpp:400,592"""
    ix = orc.Index(4, "IVF8,Flat")
    with pytest.raises(orc.OracleError, match="should be at least as large as number of clusters"):
        ix.train(np.zeros((3, 4), np.float32))


def test_numpy_cross_check_of_goldens():
    """Independent float32 numpy restatement (Q @ X.T, stable descending argsort) agrees."""
    ids, xb, _, xq = load_csv_fixtures()
    S = xq @ xb.T
    order = np.argsort(-S, axis=1, kind="stable")[:, :2]
    D, I = orc.Index(8, "Flat").search(xq, 2) if False else (None, None)
    ix = orc.Index(8, "Flat")
    ix.add(xb)
    D, I = ix.search(xq, 2)
    assert np.array_equal(I, order)
    np.testing.assert_allclose(D, np.take_along_axis(S, order, 1), rtol=2e-6)
