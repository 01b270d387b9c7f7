#!/usr/bin/env python3
"""What the in-library sharding (csrc/sharded.hip) costs on ONE device: the headline index as G virtual shards on device 0
next to (a) the unsharded index and (b) G x the step of ONE index holding N/G rows -- what the shards' own kernels cost when
they have to share the device.  overhead = T(G virtual shards) - G x T(N/G index): the fan-out's fixed cost per batch (query
distribution, record packing, exchange into the first device, device merge, finish, result copy).  Not a scaling number --
there is one GPU -- but on G real devices the G shard steps run side by side and only this overhead is added to one of them.
Both entries of the C ABI are timed: host pointers (mvs_index_search) and device pointers (mvs_index_search_device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, torch
import mi355_faiss as mf

n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, 10_000, 10
metric = mf.METRIC_L2 if os.environ.get("METRIC", "L2") == "L2" else mf.METRIC_INNER_PRODUCT


def build(rows):
    ix = mf.index_factory(d, "Flat", metric)
    for s0 in range(0, rows, 1 << 20):
        ix.add_torch(mf.synth_uniform_torch(min(1 << 20, rows - s0), d, 1234, row0=s0)); torch.cuda.synchronize()
    return ix


xq_t = mf.synth_uniform_torch(nq, d, 4321)
xq = xq_t.cpu().numpy()


def time_host(ix, reps=5):
    ix.search(xq, k)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); D, I = ix.search(xq, k); best = min(best, time.perf_counter() - t0)
    return best * 1e3, D, I


def time_dev(ix, reps=5):
    D = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
    ix.search_torch(xq_t, k, D=D, I=I); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); ix.search_torch(xq_t, k, D=D, I=I); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, D.cpu().numpy(), I.cpu().numpy()


ix = build(n)
th, D1, I1 = time_host(ix)
td, D1d, I1d = time_dev(ix)
assert np.array_equal(I1, I1d)
print(f"N={n} {os.environ.get('METRIC', 'L2')}: unsharded  host-pointer {th:.2f} ms  device-pointer {td:.2f} ms per 10k batch ({nq / td * 1e3:.0f} QPS)")
for G in (2, 4, 8):
    sh = ix.clone_to_gpu(0)
    sh.shard_to_gpus([0] * G)
    tgh, Dg, Ig = time_host(sh)
    tgd, Dgd, Igd = time_dev(sh)
    same = np.array_equal(I1, Ig) and np.array_equal(D1.view(np.uint32), Dg.view(np.uint32)) and np.array_equal(I1, Igd) and np.array_equal(D1.view(np.uint32), Dgd.view(np.uint32))
    del sh
    part = build(n // G)
    tp, _, _ = time_dev(part)
    del part
    print(f"{G} virtual shards on device 0: host-pointer {tgh:.2f} ms, device-pointer {tgd:.2f} ms; one N/{G} index {tp:.2f} ms -> "
          f"{G} x {tp:.2f} = {G * tp:.2f} ms; fan-out overhead {tgd - G * tp:+.2f} ms (device-pointer), {tgh - G * tp:+.2f} ms (host-pointer); "
          f"results identical to the unsharded index: {same}")
    print(f"   projected {G}-GPU step = N/{G} step + overhead = {tp + max(tgd - G * tp, 0):.2f} ms -> {td / (tp + max(tgd - G * tp, 0)):.2f}x of the 1-GPU step ({td:.2f} ms)")
