"""Tuning knobs are PER INDEX (round 5).  DuckDB searches different indexes from different worker threads at the
same time (SURVEY 8b "Threading"; the reference takes only a per-index lock, src/faiss_extension.cpp:629).  Rounds 1-4 kept ~35
knobs as process-wide ints behind a per-index set_option: one index's A/B switch changed the path of every other index and raced
with its launches.  Here: index A pins a split count and searches in a loop on one thread while index B, on another thread,
toggles the same knob (and others) between searches -- A's launches must keep A's value on every search, both must keep returning
the oracle's bits."""
import threading

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2 = orc.METRIC_L2


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def test_two_indexes_two_threads_one_toggles_options(mf):
    d, n, nq, k = 64, 40_000, 64, 10
    xa, xb = orc.synth_uniform(n, d, 101), orc.synth_uniform(n, d, 202)
    qa, qb = orc.synth_uniform(nq, d, 303), orc.synth_uniform(nq, d, 404)
    A, B = mf.index_factory(d, "Flat", L2), mf.index_factory(d, "Flat", L2)
    A.add(xa)
    B.add(xb)
    for ix in (A, B):
        ix.set_option("prefilter", 0)  # the exact f32 kernel: its split count is visible in last_kernel_info()
    A.set_option("mfma_nsplit", 6)
    refA = orc.flat_search(L2, xa, qa, k)
    refB = orc.flat_search(L2, xb, qb, k)
    errors = []
    stop = threading.Event()

    def run_a():
        try:
            for _ in range(150):
                D, I = A.search(qa, k)
                ki = A.last_kernel_info()
                assert ki["nsplit"] == 6, "index A ran with another index's split count: %r" % (ki,)
                assert np.array_equal(I, refA[1]) and np.array_equal(D.view(np.uint32), refA[0].view(np.uint32))
        except Exception as e:  # noqa: BLE001
            errors.append(("A", repr(e)))
        finally:
            stop.set()

    def run_b():
        try:
            i = 0
            while not stop.is_set() or i < 20:
                ns = (2, 11, 0, 3)[i % 4]
                B.set_option("mfma_nsplit", ns)
                B.set_option("mfma_global_lists", i % 2)
                B.set_option("cl_bound_mode", i % 2)
                D, I = B.search(qb, k)
                if ns:
                    assert B.last_kernel_info()["nsplit"] == ns
                assert np.array_equal(I, refB[1]) and np.array_equal(D.view(np.uint32), refB[0].view(np.uint32))
                i += 1
                if i > 2000:
                    break
        except Exception as e:  # noqa: BLE001
            errors.append(("B", repr(e)))

    ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
    ta.start(), tb.start()
    ta.join(), tb.join()
    assert not errors, errors


def test_an_ivf_index_forwards_tuning_to_itself_and_its_quantizer(mf):
    """IVF options that are tuning knobs reach both the IVF index's own launches and its coarse quantizer's (csrc/ivf.hip set_option):
    the coarse quantiser on the vector ALU, the scan's items dealt round-robin -- same bits as the defaults."""
    d, nlist, n = 64, 64, 60_000
    xb = orc.synth_clustered(n, d, 5, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(300, d, 6, n_centers=nlist, sigma=0.2)
    g, h = mf.index_factory(d, f"IVF{nlist},Flat", L2), mf.index_factory(d, f"IVF{nlist},Flat", L2)
    g.train(xb)
    g.add(xb)
    h.ivf_set_centroids(g.ivf_centroids())
    h.add(xb)
    h.set_option("ivf_coarse_mfma", 0)
    h.set_option("ivf_cl_xcd", 0)
    D0, I0 = g.search(xq, 10, nprobe=8)
    D1, I1 = h.search(xq, 10, nprobe=8)
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))
