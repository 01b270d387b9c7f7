"""REAL multi-device runs (VERDICT r4 #5a).  Every other "multi-GPU" GPU test in this tree spreads VIRTUAL shards over device 0
because the pool's boxes have one GPU; these run only where `mvs_device_count() >= 2` and are skipped elsewhere:

  * the in-library ShardedIndex (csrc/sharded.hip; reference hook src/gpu/gpu.cpp:34-63, `faiss_to_gpu(name, -1)`) on DISTINCT
    devices with both exchanges -- peer copies into device 0 (shard_exchange 0) and one ncclAllGather (shard_exchange 1) --
    against the unsharded index and the oracle, bit for bit;
  * `python bench.py --gpus 2` with no launcher around it: two ranks over RCCL on two devices, the merged result checked against
    the oracle inside the run, and the line's own census of what the collective library saw (config.collective).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _need(mf, n):
    have = mf.device_count()
    if have < n:
        pytest.skip("needs %d visible GPUs, this box has %d" % (n, have))
    return have


def _same(a, b, what):
    assert np.array_equal(a[1], b[1]), what + ": labels"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), what + ": distances"


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("exchange", [0, 1])
def test_sharded_index_on_distinct_devices(mf, metric, exchange):
    have = _need(mf, 2)
    G = min(have, 8)
    d, nb, k = 64, 60_000, 10
    rs = np.random.RandomState(100 + 10 * metric + exchange)
    xb = rs.rand(nb, d).astype(np.float32)
    xb[rs.randint(0, nb, 4000)] = xb[rs.randint(0, nb, 4000)]  # duplicate rows: exact ties across shards
    xq = np.concatenate([rs.rand(150, d).astype(np.float32), xb[rs.randint(0, nb, 50)]])
    one, sh, o = mf.index_factory(d, "Flat", metric), mf.index_factory(d, "Flat", metric), orc.Index(d, "Flat", metric)
    sh.shard_to_gpus(list(range(G)))
    assert sh.shard_info()["devices"] == list(range(G))
    sh.set_option("shard_exchange", exchange)
    for a in (one, sh, o):
        for i0 in range(0, nb, 2048):
            a.add(xb[i0 : i0 + 2048])
    for q in (xq, xq[:7]):
        ref = one.search(q, k)
        _same(sh.search(q, k), ref, "sharded on %d devices (exchange %d) vs unsharded, nq=%d" % (G, exchange, len(q)))
        _same(ref, o.search(q, k), "unsharded vs oracle")


def test_sharded_ivf_on_distinct_devices(mf):
    have = _need(mf, 2)
    G = min(have, 8)
    d, nb, nlist, k = 64, 80_000, 64, 10
    xb = orc.synth_clustered(nb, d, 31, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(300, d, 32, n_centers=nlist, sigma=0.2)
    one, sh = mf.index_factory(d, f"IVF{nlist},Flat", L2), mf.index_factory(d, f"IVF{nlist},Flat", L2)
    one.train(xb)
    one.add(xb)
    sh.shard_to_gpus(list(range(G)))
    sh.ivf_set_centroids(one.ivf_centroids())
    sh.add(xb)
    _same(sh.search(xq, k, nprobe=8), one.search(xq, k, nprobe=8), "row-sharded IVF on %d devices vs one device" % G)


def test_bench_self_launch_two_ranks_over_rccl(mf):
    _need(mf, 2)
    env = dict(os.environ)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVS_DEVICE", "MVS_BENCH_SHARED_GPU", "MVS_BENCH_BACKEND"):
        env.pop(v, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "2000000"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["config"]["row_shards"] == 2
    col = j["config"]["collective"]
    assert col["backend"] == "nccl" and col["world_size"] == 2 and col["allreduce_of_ones"] == 2 and col["distinct_devices"] == 2, col
    assert j["merged_labels_bit_exact_vs_oracle"] is True and j["merged_distances_bit_exact_vs_oracle"] is True


@pytest.mark.parametrize("exchange", [0, 1])
def test_c4_shape_flat_ip_768_on_distinct_devices(mf, exchange):
    """BASELINE.json configs[3] in small: IndexFlatIP d=768 k=10 row-sharded over every visible device (round 6) -- each
    shard's search runs flat_bf16_big_kernel (the wide stores' coarse filter), the records travel by peer copy / one ncclAllGather, the
    merge replays FAISS's inner-product tie rule across shards.  Normalised rows with duplicates: exact ties at the k-th score."""
    have = _need(mf, 2)
    G = min(have, 8)
    d, nb, nq, k = 768, 80_000 * G, 256, 10
    rs = np.random.RandomState(768 + exchange)
    xb = rs.randn(nb, d).astype(np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    xb[rs.randint(0, nb, 2000)] = xb[rs.randint(0, nb, 2000)]
    xq = np.concatenate([rs.randn(nq - 40, d).astype(np.float32), xb[rs.randint(0, nb, 40)]])
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    one, sh = mf.index_factory(d, "Flat", IP), mf.index_factory(d, "Flat", IP)
    sh.shard_to_gpus(list(range(G)))
    sh.set_option("shard_exchange", exchange)
    for a in (one, sh):
        a.set_option("prefilter", 2)
        for i0 in range(0, nb, 1 << 16):
            a.add(xb[i0 : i0 + (1 << 16)])
    ref = one.search(xq, k)
    assert one.last_kernel_info()["name"] == "flat_bf16_big_kernel"
    _same(sh.search(xq, k), ref, "C4 shape on %d devices (exchange %d) vs one device" % (G, exchange))
    assert sh.last_kernel_info()["name"] == "flat_bf16_big_kernel"
    Do, Io = orc.flat_search(IP, xb, xq[:32], k, force_path=orc.PATH_BLAS)
    assert np.array_equal(ref[1][:32], Io) and np.array_equal(ref[0][:32].view(np.uint32), Do.view(np.uint32))
