#!/usr/bin/env python3
"""Lists of 33 .. 128 entries on the wide stores (round 6; VERDICT r5 missing #3): ms per batch of the coarse filter (auto) against
the exact f32 kernel (option prefilter = 0) at the same k, FlatL2 / FlatIP, clustered L2-normalised rows (C4's data).
env: D (768), N (2 000 000), NQ (2048), KS ("10 32 33 64 100"), METRICS ("IP L2")."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import torch
import mi355_faiss as mf

d, n, nq = int(os.environ.get("D", 768)), int(os.environ.get("N", 2_000_000)), int(os.environ.get("NQ", 2048))
ks = [int(v) for v in os.environ.get("KS", "10 32 33 64 100").split()]
dev = "cuda:0"


def rows(m, seed, row0):
    x = mf.synth_clustered_torch(m, d, seed, row0=row0, n_centers=1024, sigma=1.0)
    return x / x.norm(dim=1, keepdim=True)


print(f"# wide-store list lengths: d={d} N={n} nq={nq}, clustered sigma 1 L2-normalised rows")
print(f"{'metric':<6} {'k':>4} {'filter ms':>10} {'kernel':<24} {'cand/query':>10} {'exact ms':>9} {'kernel':<20} {'x':>6}  bit-exact")
for mname in os.environ.get("METRICS", "IP L2").split():
    metric = mf.METRIC_L2 if mname == "L2" else mf.METRIC_INNER_PRODUCT
    ix = mf.index_factory(d, "Flat", metric)
    for s0 in range(0, n, 1 << 18):
        ix.add_torch(rows(min(1 << 18, n - s0), 1234, s0))
    xq = rows(nq, 4321, 0).contiguous()
    for k in ks:
        D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        res = []
        for mode in (-1, 0):
            ix.set_option("prefilter", mode)
            for _ in range(2):
                ix.search_torch(xq, k, D=D, I=I); torch.cuda.synchronize()
            c0 = ix.collect_stats()
            reps = 3 if mode < 0 else 1
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_torch(xq, k, D=D, I=I)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            c1 = ix.collect_stats()
            cand = (c1["candidates"] - c0["candidates"]) / max(1, c1["queries"] - c0["queries"])
            res.append((ms, ix.last_kernel_info()["name"], cand, D.clone(), I.clone()))
        same = bool(torch.equal(res[0][3].view(torch.int32), res[1][3].view(torch.int32)) and torch.equal(res[0][4], res[1][4]))
        print(f"{mname:<6} {k:>4} {res[0][0]:>10.2f} {res[0][1]:<24} {res[0][2]:>10.1f} {res[1][0]:>9.2f} {res[1][1]:<20} {res[1][0] / res[0][0]:>6.1f}  {same}", flush=True)
    ix.set_option("prefilter", -1)
    del ix
