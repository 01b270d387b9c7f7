"""Row-sharded search across the GPUs of one node, ONE PROCESS PER GPU (bench.py --gpus N under torchrun):
every rank searches its own row shard with GLOBAL labels, then ONE exchange step -- a single all-gather of packed
16-byte {value, label} records over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in CPU tests) -- followed by the host
k-way merge with the FAISS ordering rule on rank 0 (csrc/merge_host.hip).  The payload is nq*k*16 bytes per rank
(1.6 MB at the headline): latency-bound, ring bandwidth irrelevant.

Inner product: a shard hands over its k+1 best in the PURE order; if some query's k-th and (k+1)-th merged scores are
bit-equal, a second, tiny exchange collects per rank the k smallest global rows tied-or-better and rank 0 applies FAISS's
CMin-heap outcome (include/mi355_faiss.h, "inner-product boundary ties across processes").  Random float data never
takes that path; duplicate rows do.

The same partitioning exists INSIDE the library for a single process that owns several GPUs (csrc/sharded.hip,
faiss_to_gpu(name, -1)); the reference itself has no multi-GPU path (src/gpu/gpu.cpp:48 takes one device).
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
    Rank `src` trains on `x_train` -- ALL rows, as the reference does (src/faiss_extension.cpp:583), so the centroids are
    those of the 1-GPU run -- the nlist x d centroids (2 MB at IVF4096, d=128) are broadcast, and the other ranks
    install them instead of training."""
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


def pack_records(D, I):
    """[n, k] f32 + [n, k] i64 -> [n, k, 2] i64 records {value bits, label}: one collective instead of two"""
    rec = torch.empty(D.shape + (2,), dtype=torch.int64, device=D.device)
    rec[..., 0] = D.contiguous().view(torch.int32).to(torch.int64)
    rec[..., 1] = I
    return rec


def unpack_records(rec):
    """numpy [..., 2] i64 -> (f32 [...], i64 [...])"""
    D = rec[..., 0].astype(np.int32).view(np.float32)
    return np.ascontiguousarray(D), np.ascontiguousarray(rec[..., 1])


class ShardExchange:
    """Pre-allocated buffers for the exchange step of one (nq, kk) search shape; kk = k, or k + 1 for inner product
    with exact boundary ties (`ip_ties=True`: shards must then search with k + 1 and option ip_exact_ties = 0)."""

    def __init__(self, nq, k, device, group=None, ip_ties=False, metric=None):
        # metric given + GPU tensors: the merge runs ON THE DEVICE right behind the all-gather (mvs_merge_records_device) and
        # only the merged [nq, k] block crosses PCIe; otherwise the gathered blocks go to pinned host memory and
        # mvs_merge_shards merges them on the CPU (gloo tests, or when the caller does not say which order applies)
        self.metric = metric
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nq, self.k = nq, k
        self.kk = k + 1 if ip_ties else k
        self.ip_ties = ip_ties
        self.device = device
        if self.world > 1:
            self.grec = torch.empty((self.world, nq, self.kk, 2), dtype=torch.int64, device=device)
            pin = torch.device(device).type == "cuda"
            self.hrec = torch.empty((self.world, nq, self.kk, 2), dtype=torch.int64, pin_memory=pin)

    def gather_async(self, D, I, merge_rank=0):
        """first half: ONE all-gather of the packed records + (on merge_rank) the device-to-pinned-host copy, all enqueued
        on the current stream; nothing here waits for the GPU, so the caller can go on enqueuing the next batch"""
        self._pending = None
        if self.world == 1:
            self._pending = (D, I, None)
            return
        rec = pack_records(D, I)
        dist.all_gather_into_tensor(self.grec.view(self.world * self.nq, self.kk, 2), rec, group=self.group)
        if self.rank != merge_rank:
            return
        if self.grec.is_cuda and self.metric is not None and not self.ip_ties:
            Dm, Im = mf.merge_records_torch(self.metric, self.grec, self.k)
            if not hasattr(self, "hD"):
                self.hD = torch.empty((self.nq, self.k), dtype=torch.float32, pin_memory=True)
                self.hI = torch.empty((self.nq, self.k), dtype=torch.int64, pin_memory=True)
            self.hD.copy_(Dm, non_blocking=True)
            self.hI.copy_(Im, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.grec.device))
            self._pending = ("device", None, ev)
            return
        self.hrec.copy_(self.grec, non_blocking=True)
        ev = None
        if self.grec.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.grec.device))
        self._pending = (None, None, ev)

    def merge_host(self, metric):
        """second half (no ties): wait for the copy enqueued by gather_async, then the host k-way merge; (None, None) on
        the ranks that do not merge"""
        pend, self._pending = getattr(self, "_pending", None), None
        if pend is None:
            return None, None
        D, I, ev = pend
        if self.world == 1:
            return D.cpu().numpy(), I.cpu().numpy()
        if ev is not None:
            ev.synchronize()
        if isinstance(D, str):  # merged on the device by gather_async
            return self.hD.numpy().copy(), self.hI.numpy().copy()
        hD, hI = unpack_records(self.hrec.numpy())
        return mf.merge_shards(metric, hD, hI)

    def merge(self, metric, D, I, merge_rank=0):
        """exchange + host merge; returns (D, I) numpy [nq, k] on merge_rank, (None, None) elsewhere"""
        self.gather_async(D, I, merge_rank)
        return self.merge_host(metric)

    # ---- inner product with exact boundary ties: collective, every rank calls it with the same arguments --------
    def merge_ip_exact(self, D, I, xq, tie_candidates, merge_rank=0):
        """D, I: this rank's [nq, k+1] PURE-order results (global labels); xq: the replicated query batch (torch, on the
        exchange device); tie_candidates(xf, T) -> [nf, k] i64 global rows of THIS rank with score >= T (torch).
        Returns (D, I) numpy [nq, k] on merge_rank, (None, None) elsewhere."""
        assert self.ip_ties and D.shape[1] == self.kk
        k, kk = self.k, self.kk
        if self.world == 1:
            rawD, rawI = mf.merge_shards_raw(mf.METRIC_INNER_PRODUCT, D.cpu().numpy()[None], I.cpu().numpy()[None])
        else:
            self.gather_async(D, I, merge_rank)
            rawD = rawI = None
            if self.rank == merge_rank:
                if self.grec.is_cuda:  # merged top-(k+1) in the pure order, on the device
                    rD, rI = mf.merge_records_torch(mf.METRIC_INNER_PRODUCT, self.grec, kk, raw=True)
                    rawD, rawI = rD.cpu().numpy(), rI.cpu().numpy()
                else:
                    _, _, ev = self._pending
                    if ev is not None:
                        ev.synchronize()
                    hD, hI = unpack_records(self.hrec.numpy())
                    rawD, rawI = mf.merge_shards_raw(mf.METRIC_INNER_PRODUCT, hD, hI)
            self._pending = None
        # which queries have a bit-equal k-th and (k+1)-th score?  (decided on merge_rank, announced to everyone)
        flagged = np.zeros(0, dtype=np.int64)
        if rawD is not None:
            flagged = np.nonzero((rawI[:, k] >= 0) & (rawD[:, k] == rawD[:, k - 1]))[0].astype(np.int64)
        nf_t = torch.tensor([flagged.size], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(nf_t, src=merge_rank, group=self.group)
        nf = int(nf_t.item())
        first = np.zeros((0, k), dtype=np.int64)
        if nf > 0:
            fq = torch.empty(nf, dtype=torch.int64, device=self.device)
            T = torch.empty(nf, dtype=torch.float32, device=self.device)
            if rawD is not None:
                fq.copy_(torch.from_numpy(flagged))
                T.copy_(torch.from_numpy(np.ascontiguousarray(rawD[flagged, k - 1])))
            if self.world > 1:
                dist.broadcast(fq, src=merge_rank, group=self.group)
                dist.broadcast(T, src=merge_rank, group=self.group)
            rows = tie_candidates(xq[fq].contiguous(), T)  # [nf, k] global rows of this rank, ascending, -1 padded
            if self.world > 1:
                allr = torch.empty((self.world, nf, k), dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(allr.view(self.world * nf, k), rows.contiguous(), group=self.group)
                allr = allr.cpu().numpy()
            else:
                allr = rows.cpu().numpy()[None]
            if rawD is not None:
                cat = np.transpose(allr, (1, 0, 2)).reshape(nf, -1)
                cat = np.where(cat < 0, np.iinfo(np.int64).max, cat)
                cat.sort(axis=1)
                first = np.where(cat[:, :k] == np.iinfo(np.int64).max, -1, cat[:, :k])
        if rawD is None:
            return None, None
        return mf.finish_ip_ties(k, rawD, rawI, flagged, first)
