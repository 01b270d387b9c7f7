"""The N>1 path on CPU: two processes (gloo, world_size 2) each hold a row shard, exchange their (distance,label)
blocks with all_gather and rank 0 merges them; the merged result must equal the unsharded search.  Shard searches
are produced by the oracle here (no GPU in this container); the code under test is the exchange + host merge that
bench.py --gpus N and a multi-GPU host run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, metric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sharded import ShardExchange, shard_bounds

    orc.set_num_threads(2)
    n, d, nq, k = 20000, 32, 64, 10
    xb = orc.synth_uniform(n, d, 1234)
    xb[::97] = xb[5]  # duplicates -> ties across shards
    xq = orc.synth_uniform(nq, d, 4321)
    if metric == orc.METRIC_INNER_PRODUCT:  # small integers: nearly every query has rows tied at its k-th score
        xb = np.floor(xb * 5 - 2).astype(np.float32)
        xq = np.floor(xq * 5 - 2).astype(np.float32)
    r0, r1 = shard_bounds(n, rank, world)
    D, I = orc.flat_search(metric, xb[r0:r1], xq, k, force_path=orc.PATH_BLAS)
    I = np.where(I >= 0, I + r0, -1)  # global labels (mvs_index_set_label_offset on the device path)
    if metric == orc.METRIC_L2:
        xch = ShardExchange(nq, k, "cpu")
        Dm, Im = xch.merge(metric, torch.from_numpy(D), torch.from_numpy(I))
    else:
        # inner product: pure-order k+1 lists per shard + the tie protocol.  The device calls are stood in for by numpy:
        # pure order = (score desc, row asc); tie candidates = the k smallest rows of the shard with score >= T.
        sc = xq @ xb[r0:r1].T  # integer-valued data below: exact in f32, so ties are real ties
        order = np.lexsort((np.broadcast_to(np.arange(r1 - r0), sc.shape), -sc), axis=1)[:, : k + 1]
        Dp = np.take_along_axis(sc, order, axis=1).astype(np.float32)
        Ip = (order + r0).astype(np.int64)

        def tie_candidates(xf, T):
            s2 = xf.numpy() @ xb[r0:r1].T
            out = np.full((len(T), k), -1, dtype=np.int64)
            for f in range(len(T)):
                rows = np.nonzero(s2[f] >= T[f].item())[0][:k] + r0
                out[f, : len(rows)] = rows
            return torch.from_numpy(out)

        xch = ShardExchange(nq, k, "cpu", ip_ties=True)
        Dm, Im = xch.merge_ip_exact(torch.from_numpy(Dp), torch.from_numpy(Ip), torch.from_numpy(xq), tie_candidates)
    if rank == 0:
        Dr, Ir = orc.flat_search(metric, xb, xq, k, force_path=orc.PATH_BLAS)
        D11, _ = orc.flat_search(metric, xb, xq, k + 1, force_path=orc.PATH_BLAS)
        ret["same_D"] = bool(np.array_equal(Dm, Dr))
        ret["same_I"] = bool(np.array_equal(Im, Ir))  # every query, boundary ties included
        ret["n_ok"] = int((D11[:, k - 1] == D11[:, k]).sum()) if metric == orc.METRIC_INNER_PRODUCT else nq
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [1, 0])
def test_two_rank_exchange_and_merge_equals_unsharded(metric):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + metric
    mp.spawn(_worker, args=(2, port, metric, ret), nprocs=2, join=True)
    assert ret["same_D"] and ret["same_I"] and ret["n_ok"] > 16, dict(ret)


def _ivf_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sharded import ShardExchange, replicate_ivf_centroids, shard_bounds

    orc.set_num_threads(2)
    n, d, nq, k, nprobe = 12000, 16, 48, 10, 5
    xb = orc.synth_uniform(n, d, 1234)
    xq = orc.synth_uniform(nq, d, 4321)
    r0, r1 = shard_bounds(n, rank, world)
    # the oracle index stands in for the per-GPU shard index (same method names as mi355_faiss.Index)
    ix = orc.Index(d, "IDMap,IVF32,Flat", orc.METRIC_L2)
    replicate_ivf_centroids(ix, xb if rank == 0 else None, src=0)
    assert ix.is_trained
    ix.add_with_ids(xb[r0:r1], np.arange(r0, r1, dtype=np.int64))
    D, I = ix.search(xq, k, nprobe=nprobe)
    xch = ShardExchange(nq, k, "cpu")
    Dm, Im = xch.merge(orc.METRIC_L2, torch.from_numpy(D), torch.from_numpy(I))
    if rank == 0:
        one = orc.Index(d, "IDMap,IVF32,Flat", orc.METRIC_L2)
        one.train(xb)
        one.add_with_ids(xb, np.arange(n, dtype=np.int64))
        Dr, Ir = one.search(xq, k, nprobe=nprobe)
        ret["same_D"] = bool(np.array_equal(Dm, Dr))
        ret["same_I"] = bool(np.array_equal(Im, Ir))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ivf_replicated_centroids_equals_unsharded():
    """centroids replicated (rank 0 trains, broadcast), every inverted list row-sharded -> the merged
    result is the single-index result, bit for bit"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31700 + (os.getpid() % 2000)
    mp.spawn(_ivf_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["same_D"] and ret["same_I"]


def _qgroup_worker(rank, world, port, qgroups, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sharded import ShardExchange

    orc.set_num_threads(1)
    n, d, nq, k = 12000, 24, 45, 10  # (45 queries: the last query group is shorter than the first)
    xb = orc.synth_uniform(n, d, 1234)
    xb[::89] = xb[7]  # duplicates -> ties across shards
    xq = orc.synth_uniform(nq, d, 4321)
    xch = ShardExchange(nq, k, "cpu", qgroups=qgroups)
    r0, r1 = xch.row_bounds(n)
    qa, qb = xch.query_range()
    D, I = orc.flat_search(orc.METRIC_L2, xb[r0:r1], xq[qa:qb], k, force_path=orc.PATH_BLAS)
    I = np.where(I >= 0, I + r0, -1)
    Dm, Im = xch.merge(orc.METRIC_L2, torch.from_numpy(D), torch.from_numpy(I))
    if rank == 0:
        Dr, Ir = orc.flat_search(orc.METRIC_L2, xb, xq, k, force_path=orc.PATH_BLAS)
        ret["shape"] = tuple(Dm.shape)
        ret["same_D"] = bool(np.array_equal(Dm, Dr))
        ret["same_I"] = bool(np.array_equal(Im, Ir))
        ret["layout"] = (xch.G, xch.R)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,qgroups", [(4, 2), (2, 2)])
def test_query_groups_times_row_shards_equal_unsharded(world, qgroups):
    """bench.py --query-groups: G groups of R = world / G row shards, group g answers the g-th slice of the queries; one
    all-gather, one k-way merge per group -> the unsharded result (4 ranks = 2 x 2; 2 ranks = 2 x 1: query slices only)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33100 + (os.getpid() % 2000) + 7 * world
    mp.spawn(_qgroup_worker, args=(world, port, qgroups, ret), nprocs=world, join=True)
    assert ret["shape"] == (45, 10) and ret["layout"] == (qgroups, world // qgroups), dict(ret)
    assert ret["same_D"] and ret["same_I"], dict(ret)
