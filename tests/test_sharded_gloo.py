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
    r0, r1 = shard_bounds(n, rank, world)
    D, I = orc.flat_search(metric, xb[r0:r1], xq, k, force_path=orc.PATH_BLAS)
    I = np.where(I >= 0, I + r0, -1)  # global labels (mvs_index_set_label_offset on the device path)
    xch = ShardExchange(nq, k, "cpu")
    Dm, Im = xch.merge(metric, torch.from_numpy(D), torch.from_numpy(I))
    if rank == 0:
        Dr, Ir = orc.flat_search(metric, xb, xq, k, force_path=orc.PATH_BLAS)
        D11, _ = orc.flat_search(metric, xb, xq, k + 1, force_path=orc.PATH_BLAS)
        ok = D11[:, k - 1] != D11[:, k] if metric == orc.METRIC_INNER_PRODUCT else np.ones(nq, bool)
        ret["same_D"] = bool(np.array_equal(Dm, Dr))
        ret["same_I"] = bool(np.array_equal(Im[ok], Ir[ok]))
        ret["n_ok"] = int(ok.sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", [1, 0])
def test_two_rank_exchange_and_merge_equals_unsharded(metric):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + metric
    mp.spawn(_worker, args=(2, port, metric, ret), nprocs=2, join=True)
    assert ret["same_D"] and ret["same_I"] and ret["n_ok"] > 32
