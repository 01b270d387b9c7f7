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


def _ivf_tie_worker(rank, world, port, metric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sharded import ShardExchange, shard_bounds

    orc.set_num_threads(2)
    is_l2 = metric == orc.METRIC_L2
    n, d, nq, k, nlist, nprobe = 9000, 16, 60, 10, 16, 4
    rs = np.random.RandomState(5)
    xb = rs.randint(-2, 3, size=(n, d)).astype(np.float32)  # integer coordinates: every distance is exact in f32, ties are real ties
    dup = rs.randint(0, n, n * 3 // 10)
    xb[dup] = xb[rs.randint(0, n, len(dup))]
    xq = np.concatenate([rs.randint(-2, 3, size=(nq - 20, d)).astype(np.float32), xb[rs.randint(0, n, 20)]])
    one = orc.Index(d, f"IVF{nlist},Flat", metric)
    one.train(xb)
    one.add(xb)
    # the shard of this rank, stood in for by numpy (no GPU here): list l of the shard = the rows of the unsharded list l that
    # fall into [r0, r1), in insertion order; probe order = the coarse quantiser's (an exact Flat search over the centroids)
    r0, r1 = shard_bounds(n, rank, world)
    lists = []
    for l in range(nlist):
        ids, codes = one.ivf_list(l)
        m = (ids >= r0) & (ids < r1)
        lists.append((ids[m], codes[m]))
    _, probes = orc.flat_search(metric, one.ivf_centroids(), xq, nprobe)

    def value(x, rows):
        return ((x[None, :] - rows) ** 2).sum(axis=1).astype(np.float32) if is_l2 else (rows @ x).astype(np.float32)

    def arrival(q):
        """(value, id, probe rank) of every row of this shard the query visits, in arrival order"""
        v, i, p = [], [], []
        for pr, l in enumerate(probes[q]):
            if l < 0:
                continue
            ids, codes = lists[int(l)]
            v.append(value(xq[q], codes)), i.append(ids), p.append(np.full(len(ids), pr, dtype=np.int64))
        return np.concatenate(v), np.concatenate(i), np.concatenate(p)

    kk = k + 1
    D = np.full((nq, kk), np.finfo(np.float32).max if is_l2 else -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((nq, kk), -1, dtype=np.int64)
    for q in range(nq):
        v, i, _ = arrival(q)
        o = np.lexsort((i, v if is_l2 else -v))[:kk]  # the shard's k + 1 best in the pure order
        D[q, : len(o)], I[q, : len(o)] = v[o], i[o]

    def tie_emit(fq, T):
        nf = len(fq)
        ev = np.zeros((nf, k), dtype=np.float32)
        ei = np.full((nf, k), -1, dtype=np.int64)
        ep = np.full((nf, k), -1, dtype=np.int32)
        for f in range(nf):
            v, i, p = arrival(int(fq[f]))
            keep = np.nonzero(v <= T[f].item() if is_l2 else v >= T[f].item())[0][:k]
            ev[f, : len(keep)], ei[f, : len(keep)], ep[f, : len(keep)] = v[keep], i[keep], p[keep]
        return torch.from_numpy(ev), torch.from_numpy(ei), torch.from_numpy(ep)

    xch = ShardExchange(nq, k, "cpu", ip_ties=True)  # (k + 1 entries per shard)
    Dm, Im = xch.merge_ivf_exact(metric, torch.from_numpy(D), torch.from_numpy(I), tie_emit)
    if rank == 0:
        Dr, Ir = one.search(xq, k, nprobe=nprobe)
        D11, _ = one.search(xq, k + 1, nprobe=nprobe)
        ret["same_D"] = bool(np.array_equal(Dm, Dr))
        ret["same_I"] = bool(np.array_equal(Im, Ir))
        ret["tied_queries"] = int((D11[:, k - 1] == D11[:, k]).sum())
        ret["bad"] = [int(q) for q in np.nonzero((Im != Ir).any(axis=1))[0][:5]]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [1, 0])
@pytest.mark.parametrize("world", [2, 3])
def test_ivf_row_shards_exact_ties_across_processes(metric, world):
    """VERDICT r4 #5b: the one-process-per-GPU host merged IVF shards in the pure order; FAISS's scanner heap keeps the rows tied at
    the k-th value by ARRIVAL order (probe rank, then list position).  ShardExchange.merge_ivf_exact runs the cross-process
    protocol (include/mi355_faiss.h); integer coordinates with 30 % duplicated rows tie most queries at rank k -- every query must
    equal the oracle's unsharded IVF search."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 34900 + (os.getpid() % 2000) + 11 * world + metric
    mp.spawn(_ivf_tie_worker, args=(world, port, metric, ret), nprocs=world, join=True)
    assert ret["tied_queries"] > 10, dict(ret)
    assert ret["same_D"] and ret["same_I"], dict(ret)


def _c4_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sharded import ShardExchange, shard_bounds

    orc.set_num_threads(2)
    IP = orc.METRIC_INNER_PRODUCT
    n, d, nq, k = 8000, 768, 48, 10
    rs = np.random.RandomState(4)
    xb = rs.randn(n, d).astype(np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    xb[rs.randint(0, n, 600)] = xb[rs.randint(0, n, 600)]  # duplicated rows land in different shards: exact ties at the k-th score
    xq = np.concatenate([rs.randn(nq - 16, d).astype(np.float32), xb[rs.randint(0, n, 16)]])
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    r0, r1 = shard_bounds(n, rank, world)
    # the shard's pure-order k + 1 list in the ORACLE's arithmetic (k-ordered chains: what the device's exact re-scoring produces)
    Dp, Ip = orc.flat_search(IP, xb[r0:r1], xq, min(k + 1, r1 - r0), force_path=orc.PATH_BLAS)
    kk = Dp.shape[1]
    order = np.lexsort((Ip, -Dp), axis=1)  # (score desc, row asc) -- the kernels' pure order
    Dp, Ip = np.take_along_axis(Dp, order, axis=1), np.take_along_axis(Ip, order, axis=1) + r0
    assert kk == k + 1

    def tie_candidates(xf, T):
        out = np.full((len(T), k), -1, dtype=np.int64)
        Df, If = orc.flat_search(IP, xb[r0:r1], xf.numpy(), min(256, r1 - r0), force_path=orc.PATH_BLAS)
        for f in range(len(T)):
            rows = np.sort(If[f][Df[f] >= T[f].item()])[:k] + r0
            out[f, : len(rows)] = rows
        return torch.from_numpy(out)

    xch = ShardExchange(nq, k, "cpu", ip_ties=True)
    Dm, Im = xch.merge_ip_exact(torch.from_numpy(Dp.copy()), torch.from_numpy(Ip.copy()), torch.from_numpy(xq), tie_candidates)
    if rank == 0:
        Dr, Ir = orc.flat_search(IP, xb, xq, k, force_path=orc.PATH_BLAS)
        D11, _ = orc.flat_search(IP, xb, xq, k + 1, force_path=orc.PATH_BLAS)
        ret["same_D"] = bool(np.array_equal(Dm.view(np.uint32), Dr.view(np.uint32)))
        ret["same_I"] = bool(np.array_equal(Im, Ir))
        ret["tied"] = int((D11[:, k - 1] == D11[:, k]).sum())
    dist.barrier()
    dist.destroy_process_group()


def test_c4_shape_flat_ip_768_four_ranks():
    """BASELINE.json configs[3] in small (IndexFlatIP d=768 k=10, row shards + all-gather merge): four gloo ranks, the
    shards' lists in the oracle's arithmetic, boundary ties between shards resolved by the second exchange -- equals the unsharded search."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 36100 + (os.getpid() % 2000)
    mp.spawn(_c4_worker, args=(4, port, ret), nprocs=4, join=True)
    assert ret["same_D"] and ret["same_I"], dict(ret)
