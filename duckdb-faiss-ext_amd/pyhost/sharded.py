"""Row-sharded search across the GPUs of one node: one process per GPU, every rank searches its own
row shard with GLOBAL labels, then ONE exchange step -- an all-gather of the per-shard (distance, label) blocks over
RCCL/xGMI (backend "nccl" on ROCm; "gloo" in CPU tests) -- followed by the host k-way merge with the FAISS ordering
rule (csrc/merge_host.hip via mvs_merge_shards).  The payload is nq*k*12 bytes per rank: latency-bound.

The reference has no multi-GPU path at all (src/gpu/gpu.cpp:48 takes a single device); FAISS's own
index_cpu_to_gpu_multiple/IndexShards merges on host threads with the same heap rule.
"""
import numpy as np
import torch
import torch.distributed as dist

import mi355_faiss as mf


def shard_bounds(n, rank, world):
    """rows [r0, r1) of rank `rank`"""
    return n * rank // world, n * (rank + 1) // world


def replicate_ivf_centroids(ix, x_train=None, src=0, device=None, group=None):
    """IVF over row shards: ONE set of centroids on every GPU, every inverted list row-sharded, so that
    the union of the ranks' list scans is exactly the single-index scan and the merged top-k equals the unsharded result.
    Rank `src` trains on `x_train` (the reference trains once, src/faiss_extension.cpp:583), the nlist x d centroids
    (2 MB at IVF4096, d=128) are broadcast, and the other ranks install them instead of training."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == src:
        ix.train(x_train)
    if world == 1:
        return
    c = torch.empty((ix.nlist, ix.d), dtype=torch.float32, device=device if device is not None else "cpu")
    if rank == src:
        c.copy_(torch.from_numpy(ix.ivf_centroids()))
    dist.broadcast(c, src=src, group=group)
    if rank != src:
        ix.ivf_set_centroids(c.cpu().numpy())


class ShardExchange:
    """Pre-allocated buffers for the exchange step of one (nq, k) search shape."""

    def __init__(self, nq, k, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nq, self.k = nq, k
        if self.world > 1:
            self.gD = torch.empty((self.world, nq, k), dtype=torch.float32, device=device)
            self.gI = torch.empty((self.world, nq, k), dtype=torch.int64, device=device)
            pin = torch.device(device).type == "cuda"
            self.hD = torch.empty((self.world, nq, k), dtype=torch.float32, pin_memory=pin)
            self.hI = torch.empty((self.world, nq, k), dtype=torch.int64, pin_memory=pin)

    def all_gather(self, D, I):
        """D, I: this rank's [nq, k] results with global labels.  Returns ([world,nq,k], [world,nq,k]) on device."""
        if self.world == 1:
            return D.unsqueeze(0), I.unsqueeze(0)
        # concatenation along dim 0: [world*nq, k] is the same memory as [world, nq, k]
        dist.all_gather_into_tensor(self.gD.view(self.world * self.nq, self.k), D.contiguous(), group=self.group)
        dist.all_gather_into_tensor(self.gI.view(self.world * self.nq, self.k), I.contiguous(), group=self.group)
        return self.gD, self.gI

    def gather_async(self, D, I, merge_rank=0):
        """first half of merge(): all-gather + (on merge_rank) the device-to-pinned-host copy, all enqueued on the
        current stream; nothing here waits for the GPU, so the caller can go on enqueuing the next batch"""
        gD, gI = self.all_gather(D, I)
        self._pending = None
        if self.rank != merge_rank:
            return
        if self.world == 1:
            self._pending = (D, I, None)
            return
        self.hD.copy_(gD, non_blocking=True)
        self.hI.copy_(gI, non_blocking=True)
        ev = None
        if gD.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(gD.device))
        self._pending = (None, None, ev)

    def merge_host(self, metric):
        """second half: wait for the copy enqueued by gather_async, then the host k-way merge; (None, None) on the
        ranks that do not merge"""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is None:
            return None, None
        D, I, ev = pend
        if self.world == 1:
            return D.cpu().numpy(), I.cpu().numpy()
        if ev is not None:
            ev.synchronize()
        return mf.merge_shards(metric, self.hD.numpy(), self.hI.numpy())

    def merge(self, metric, D, I, merge_rank=0):
        """exchange + host merge; returns (D, I) numpy [nq, k] on merge_rank, (None, None) elsewhere"""
        self.gather_async(D, I, merge_rank)
        return self.merge_host(metric)
