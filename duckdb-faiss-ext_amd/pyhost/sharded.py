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

    def __init__(self, nq, k, device, group=None, ip_ties=False, metric=None, qgroups=1):
        # metric given + GPU tensors: the merge runs ON THE DEVICE right behind the all-gather (mvs_merge_records_device) and
        # only the merged [nq, k] block crosses PCIe; otherwise the gathered blocks go to pinned host memory and
        # mvs_merge_shards merges them on the CPU (gloo tests, or when the caller does not say which order applies)
        # qgroups = G > 1: the ranks form G groups of R = world / G; group g holds the WHOLE database as R row shards and
        # answers the g-th slice of the queries (rank = g * R + row shard).  A shard's step cost has a part per (query, row)
        # pair and a part per query; 2 x 4 instead of 1 x 8 halves the second (bench.py --query-groups).
        self.metric = metric
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.G = max(1, int(qgroups))
        if self.world % self.G != 0 or (ip_ties and self.G > 1):
            raise ValueError("qgroups must divide the world size (and the inner-product tie protocol needs qgroups = 1)")
        self.R = self.world // self.G
        self.qg, self.rs = self.rank // self.R, self.rank % self.R
        self.nq, self.k = nq, k
        self.nq_g = (nq + self.G - 1) // self.G  # queries per group (the last group may hold fewer)
        self.kk = k + 1 if ip_ties else k
        self.ip_ties = ip_ties
        self.device = device
        if self.world > 1:
            self.grec = torch.empty((self.world, self.nq_g, self.kk, 2), dtype=torch.int64, device=device)
            pin = torch.device(device).type == "cuda"
            self.hrec = torch.empty((self.world, self.nq_g, self.kk, 2), dtype=torch.int64, pin_memory=pin)

    def query_range(self):
        """queries [qa, qb) this rank answers (all of them unless qgroups > 1)"""
        qa = min(self.nq, self.qg * self.nq_g)
        return qa, min(self.nq, qa + self.nq_g)

    def row_bounds(self, n):
        """rows [r0, r1) of this rank's shard"""
        return shard_bounds(n, self.rs, self.R)

    def gather_async(self, D, I, merge_rank=0):
        """first half: ONE all-gather of the packed records + (on merge_rank) the device-to-pinned-host copy, all enqueued
        on the current stream; nothing here waits for the GPU, so the caller can go on enqueuing the next batch.
        D, I: this rank's results for ITS queries (query_range(); all nq of them when qgroups = 1)"""
        self._pending = None
        if self.world == 1:
            self._pending = (D, I, None)
            return
        rec = pack_records(D, I)
        if rec.shape[0] < self.nq_g:  # the last query group: pad (the rows past nq are dropped after the merge)
            pad = torch.zeros((self.nq_g - rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
            pad[..., 1] = -1
            rec = torch.cat([rec, pad], dim=0)
        with mf.trace_range("mvs:exchange (all_gather of 16-byte records)"):  # (roctx: shows up in rocprofv3 --marker-trace)
            dist.all_gather_into_tensor(self.grec.view(self.world * self.nq_g, self.kk, 2), rec, group=self.group)
        if self.rank != merge_rank:
            return
        if self.grec.is_cuda and self.metric is not None and not self.ip_ties:
            if not hasattr(self, "hD"):
                self.hD = torch.empty((self.G * self.nq_g, self.k), dtype=torch.float32, pin_memory=True)
                self.hI = torch.empty((self.G * self.nq_g, self.k), dtype=torch.int64, pin_memory=True)
            for g in range(self.G):  # one k-way merge per query group over its R row shards
                Dm, Im = mf.merge_records_torch(self.metric, self.grec[g * self.R : (g + 1) * self.R], self.k)
                self.hD[g * self.nq_g : (g + 1) * self.nq_g].copy_(Dm, non_blocking=True)
                self.hI[g * self.nq_g : (g + 1) * self.nq_g].copy_(Im, non_blocking=True)
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
        with mf.trace_range("mvs:merge (wait for the records + k-way merge)"):
            return self._merge_host_body(metric, D, I, ev)

    def _merge_host_body(self, metric, D, I, ev):
        if ev is not None:
            ev.synchronize()
        if isinstance(D, str):  # merged on the device by gather_async
            return self.hD.numpy()[: self.nq].copy(), self.hI.numpy()[: self.nq].copy()
        hD, hI = unpack_records(self.hrec.numpy())
        if self.G == 1:
            return mf.merge_shards(metric, hD, hI)
        parts = [mf.merge_shards(metric, hD[g * self.R : (g + 1) * self.R], hI[g * self.R : (g + 1) * self.R]) for g in range(self.G)]
        return (np.concatenate([p[0] for p in parts])[: self.nq], np.concatenate([p[1] for p in parts])[: self.nq])

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

    # ---- IVF with exact distance ties (round 5): collective, every rank calls it with the same arguments ------------------------
    def merge_ivf_exact(self, metric, D, I, tie_emit, merge_rank=0, ids_ascending=None):
        """Row-sharded IVF, the heap's outcome under exact ties (include/mi355_faiss.h "IVF exact distance ties across PROCESSES";
        csrc/sharded.hip resolve_ties_ivf is the in-library twin).  D, I: this rank's [nq, k+1] PURE-order lists (the shard searched
        with k + 1 and option ivf_exact_ties = 0; labels = global rows); tie_emit(flagged i64 tensor, T f32 tensor) -> (v [nf, k] f32,
        id [nf, k] i64, rank [nf, k] i32) torch: this rank's first k rows not worse than T in arrival order, -1 padded
        (Index.ivf_tie_emit_torch).  Returns (D, I) numpy [nq, k] on merge_rank, (None, None) elsewhere."""
        assert self.ip_ties and D.shape[1] == self.kk  # (ip_ties = "k + 1 entries per shard": the same buffers serve both protocols)
        # (A_k below is ordered by (probe rank, stored id) -- FAISS's arrival order inside a list only while ids grew with
        # insertion order; pass Index.get_stat("ivf_ids_ascending") of the shard and the merge refuses what it cannot reproduce)
        if ids_ascending is not None and not ids_ascending:
            raise ValueError("merge_ivf_exact: the shard's ids do not grow with insertion order (add_with_ids with arbitrary ids): "
                             "the heap's tie outcome cannot be rebuilt from (probe rank, id); search with ivf_exact_ties = 0 and merge_shards instead")
        k, kk = self.k, self.kk
        is_l2 = metric == mf.METRIC_L2
        if self.world == 1:
            rawD, rawI = mf.merge_shards_raw(metric, D.cpu().numpy()[None], I.cpu().numpy()[None])
        else:
            self.gather_async(D, I, merge_rank)
            rawD = rawI = None
            if self.rank == merge_rank:
                if self.grec.is_cuda:
                    rD, rI = mf.merge_records_torch(metric, self.grec, kk, raw=True)
                    rawD, rawI = rD.cpu().numpy(), rI.cpu().numpy()
                else:
                    _, _, ev = self._pending
                    if ev is not None:
                        ev.synchronize()
                    hD, hI = unpack_records(self.hrec.numpy())
                    rawD, rawI = mf.merge_shards_raw(metric, hD, hI)
            self._pending = None
        flagged = np.zeros(0, dtype=np.int64)
        if rawD is not None:
            flagged = np.nonzero((rawI[:, k] >= 0) & (rawD[:, k] == rawD[:, k - 1]))[0].astype(np.int64)
        nf_t = torch.tensor([flagged.size], dtype=torch.int64, device=self.device)
        if self.world > 1:
            dist.broadcast(nf_t, src=merge_rank, group=self.group)
        nf = int(nf_t.item())
        em = None
        if nf > 0:
            fq = torch.empty(nf, dtype=torch.int64, device=self.device)
            T = torch.empty(nf, dtype=torch.float32, device=self.device)
            if rawD is not None:
                fq.copy_(torch.from_numpy(flagged))
                T.copy_(torch.from_numpy(np.ascontiguousarray(rawD[flagged, k - 1])))
            if self.world > 1:
                dist.broadcast(fq, src=merge_rank, group=self.group)
                dist.broadcast(T, src=merge_rank, group=self.group)
            v, ids, rk = tie_emit(fq, T)
            # one record per entry: {value bits, id, probe rank} as three int64 -> ONE all-gather
            rec = torch.stack([v.contiguous().view(torch.int32).to(torch.int64), ids.to(torch.int64), rk.to(torch.int64)], dim=-1).contiguous()
            if self.world > 1:
                allr = torch.empty((self.world, nf, k, 3), dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(allr.view(self.world * nf, k, 3), rec, group=self.group)
                em = allr.cpu().numpy()
            else:
                em = rec.cpu().numpy()[None]
        if rawD is None:
            return None, None
        # FAISS's print order of the pure lists: equal values by stored id -- ascending for L2, descending for inner product
        outD = np.ascontiguousarray(rawD[:, :k]).copy()
        outI = np.ascontiguousarray(rawI[:, :k]).copy()
        neutral = np.float32(np.finfo(np.float32).max if is_l2 else -np.finfo(np.float32).max)
        outD[outI < 0] = neutral
        if not is_l2:
            for q in range(outD.shape[0]):
                a = 0
                while a < k:
                    b = a + 1
                    while b < k and outI[q, b] >= 0 and outI[q, a] >= 0 and outD[q, b] == outD[q, a]:
                        b += 1
                    if b - a > 1:
                        outI[q, a:b] = outI[q, a:b][::-1]
                    a = b
        for f in range(nf):
            q = int(flagged[f])
            Tq = rawD[q, k - 1]
            e = em[:, f].reshape(-1, 3)
            e = e[e[:, 1] >= 0]
            ev = e[:, 0].astype(np.int32).view(np.float32)
            order = np.lexsort((e[:, 1], e[:, 2]))[:k]  # A_k: the first k of the union by (probe rank, id)
            av, aid = ev[order], e[order, 1]
            better = (rawD[q, :k] < Tq) if is_l2 else (rawD[q, :k] > Tq)
            better &= rawI[q, :k] >= 0
            res = list(zip(rawD[q, :k][better].tolist(), rawI[q, :k][better].tolist()))
            nbetter = len(res)
            in_ak = int(np.count_nonzero((av < Tq) if is_l2 else (av > Tq)))
            tied = np.sort(aid[av == Tq])
            gev = nbetter - in_ak  # rows better than T that arrived after A_k was complete: each evicted a tied row
            nt = len(tied)
            if is_l2:  # the gev LARGEST ids were evicted; ascending id
                keep = tied[: max(nt - gev, 0)].tolist()
            else:  # the gev SMALLEST ids were evicted; printed in descending id
                keep = tied[gev:][::-1].tolist()
                a = 0
                while a < nbetter:  # (the better part came out of the pure order: equal scores print in descending id)
                    b = a + 1
                    while b < nbetter and res[b][0] == res[a][0]:
                        b += 1
                    res[a:b] = res[a:b][::-1]
                    a = b
            res += [(float(Tq), int(t)) for t in keep]
            for j in range(k):
                if j < len(res):
                    outD[q, j], outI[q, j] = np.float32(res[j][0]), res[j][1]
                else:
                    outD[q, j], outI[q, j] = neutral, -1
        return outD, outI
