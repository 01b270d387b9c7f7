"""HNSW search-kernel variant timing (G rows in flight, waves/CU cap, visited-set placement) in ONE process on one index.
    GAP_N=1000000 GAP_VARIANTS="((16,0,1),(2,0,0))" python tools/hnsw_variants.py
"""
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "duckdb-faiss-ext_amd", "pyhost"))
import torch, numpy as np, mi355_faiss as mf
d=768; n=int(os.environ.get("GAP_N","1000000"))
ix = mf.index_factory(d, "HNSW32", mf.METRIC_L2)
for s0 in range(0, n, 65536):
    xb = mf.synth_clustered_torch(min(65536, n-s0), d, 1234, row0=s0, sigma=1.0); xb /= xb.norm(dim=1, keepdim=True)
    ix.add_torch(xb)
xq = mf.synth_clustered_torch(10000, d, 4321, sigma=1.0); xq /= xq.norm(dim=1, keepdim=True)
torch.cuda.synchronize()
ef = 128
VARIANTS = eval(os.environ.get("GAP_VARIANTS", "((4,0,1),(8,0,1),(8,0,0),(4,0,0))"))
for g, w, h in VARIANTS:
    ix.set_option("hnsw_search_g", g); ix.set_option("hnsw_search_waves", w); ix.set_option("hnsw_visited_lds", h)
    print("== G=%d waves/CU cap=%d lds-visited=%d" % (g, w, h))
    D, I = ix.search_torch(xq, 10, efSearch=ef); torch.cuda.synchronize()
    if "Iref" not in globals():
        Dref, Iref = D.clone(), I.clone()
    print("   same results as first variant:", bool((I == Iref).all()) and bool((D == Dref).all()))
    for timing in (True,):
        ix.set_kernel_timing(timing)
        prev = 0.0
        for rep in range(2):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            ix.search_torch(xq, 10, D=D, I=I, efSearch=ef)
            t1 = time.perf_counter(); e1.record()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            nl, ms = ix.kernel_time_stats() if timing else (0, 0.0)
            print("ef=%d timing=%s rep %d: call returned after %.3f ms, synced after %.3f ms, torch-event %.3f ms, lib kernel event %.3f ms" % (ef, timing, rep, (t1-t0)*1e3, (t2-t0)*1e3, e0.elapsed_time(e1), ms - prev), flush=True)
            prev = ms
            ki = ix.last_kernel_info()
            print("      -> dist evals/query %.0f, %.0f GB/s algorithmic, grid %d" % (ki["bytes"] / (4*d+4) / 10000, ki["bytes"] / ((t2-t0)) / 1e9, ki["grid"]))
        ix.set_kernel_timing(False)
