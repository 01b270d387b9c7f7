"""Shared test helpers: golden fixtures and comparison utilities."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_csv_fixtures():
    """training.csv / queries.csv of the reference (test/sql/): col0 = id, col1..8 = vector."""
    tr = np.loadtxt(os.path.join(GOLDEN, "training.csv"), delimiter=",", dtype=np.float64)
    qu = np.loadtxt(os.path.join(GOLDEN, "queries.csv"), delimiter=",", dtype=np.float64)
    # the reference casts LIST<DOUBLE> -> FLOAT (src/faiss_extension.cpp:292-293)
    return (
        tr[:, 0].astype(np.int64),
        tr[:, 1:].astype(np.float32),
        qu[:, 0].astype(np.int64),
        qu[:, 1:].astype(np.float32),
    )


def goldens():
    return json.load(open(os.path.join(GOLDEN, "reference_goldens.json")))


def bitmap_from_ids(ids, keep_mask):
    """The reference's mask: bit `id` set iff the filter is true for the row whose selector
    column equals id; sized max_id/8+1 bytes (src/faiss_extension.cpp:765-778)."""
    ids = np.asarray(ids, dtype=np.int64)
    nbytes = int(ids.max()) // 8 + 1
    bm = np.zeros(nbytes, dtype=np.uint8)
    for i in ids[np.asarray(keep_mask, dtype=bool)]:
        bm[i >> 3] |= np.uint8(1 << (i & 7))
    return bm


def assert_same_results(D, I, D_ref, I_ref, metric_is_l2, rtol=0.0, what=""):
    """Labels bit-exact; distances bit-exact when rtol == 0 else within rtol (relative)."""
    I = np.asarray(I)
    I_ref = np.asarray(I_ref)
    if not np.array_equal(I, I_ref):
        bad = np.argwhere(I != I_ref)
        q = bad[0][0]
        raise AssertionError(
            f"{what}: labels differ in {len(bad)} slots; first query {q}:\n got {I[q]}\n ref {I_ref[q]}\n"
            f" gotD {np.asarray(D)[q]}\n refD {np.asarray(D_ref)[q]}"
        )
    if rtol == 0.0:
        a = np.asarray(D, dtype=np.float32).view(np.uint32)
        b = np.asarray(D_ref, dtype=np.float32).view(np.uint32)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            q = bad[0][0]
            raise AssertionError(f"{what}: distances not bit-exact in {len(bad)} slots; q={q}: {D[q]} vs {D_ref[q]}")
    else:
        np.testing.assert_allclose(D, D_ref, rtol=rtol, atol=0, err_msg=what)
