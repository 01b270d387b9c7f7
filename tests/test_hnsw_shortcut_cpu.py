"""CPU restatement of the HNSW build's full-list short cut (csrc/hnsw.hip add_link, round 6) against the pairwise pass it replaces
(faiss/impl/HNSW.cpp shrink_neighbor_list, as oracle/orc_hnsw.c restates it): a list that is FULL and is the kept sequence of its
last shrink is shrunk again with one more candidate -- by the full heuristic and by the short cut -- over many insertions in a row;
the two must produce the same list every time, on data with exact distance ties as well."""
import numpy as np
import pytest


def _dist(a, b):
    d = a - b
    return np.float32(np.dot(d, d))


def shrink_full(cands, vecs, src, L):
    """candidates: ids; sorted by (distance to src, id); FAISS's loop: keep c unless a kept s is closer to c than src is"""
    order = sorted(cands, key=lambda c: (_dist(vecs[src], vecs[c]), c))
    kept = []
    for c in order:
        d1 = _dist(vecs[src], vecs[c])
        if all(not (_dist(vecs[c], vecs[s]) < d1) for s in kept):
            kept.append(c)
            if len(kept) >= L:
                break
    return kept


def shrink_shortcut(kept_prev, dest, vecs, src, L):
    """kept_prev: the (full, clean) list = kept sequence of the last shrink; dest joins"""
    order = sorted(kept_prev + [dest], key=lambda c: (_dist(vecs[src], vecs[c]), c))
    r = order.index(dest)
    d_dest = _dist(vecs[src], vecs[dest])
    if any(_dist(vecs[dest], vecs[s]) < d_dest for s in order[:r]):
        return list(kept_prev), True  # dest is pruned: the list stays what it is
    out = order[: r + 1]
    for c in order[r + 1 :]:
        if not (_dist(vecs[c], vecs[dest]) < _dist(vecs[src], vecs[c])):
            out.append(c)
    return out[:L], len(out[:L]) == L


@pytest.mark.parametrize("d,L,integer", [(48, 8, False), (96, 16, False), (24, 8, True), (64, 32, False), (40, 16, True)])
def test_short_cut_equals_the_pairwise_pass(d, L, integer):
    rs = np.random.RandomState(d * 100 + L)
    n = 600
    # a HUB: src sits at the centre of the cloud, so a candidate is rarely closer to a kept row than to src and the list stays full --
    # the situation of the build's hot vertices on high-dimensional rows; a third of the rows are near-copies (or, integer data, exact
    # copies) of earlier ones, which do get pruned -- or prune
    vecs = (rs.randint(-2, 3, size=(n, d)) if integer else rs.randn(n, d)).astype(np.float32)
    for i in range(2 * L, n):
        if rs.rand() < 0.33:
            j = rs.randint(1, i)
            vecs[i] = vecs[j] if integer else vecs[j] + 0.01 * rs.randn(d).astype(np.float32)
    vecs[0] = 0
    src = 0
    # the list becomes full by plain appends (no heuristic), then its first shrink makes it "clean"
    lst = list(range(1, L + 1))
    clean = False
    took = 0
    for dest in range(L + 1, n):
        full = shrink_full(lst + [dest], vecs, src, L)
        if clean and len(lst) == L:
            got, still = shrink_shortcut(sorted(lst, key=lambda c: (_dist(vecs[src], vecs[c]), c)), dest, vecs, src, L)
            assert sorted(got) == sorted(full), (dest, got, full)
            took += 1
        if len(lst) < L:  # room: FAISS appends without looking
            lst = lst + [dest]
            clean = False
        else:
            lst = full
            clean = len(full) == L
    assert took > 20  # the short cut's case does occur
