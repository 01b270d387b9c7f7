"""HNSW32 build on UNIFORM rows (the ingest line's hardest case, 10 k rows/s in round 5): one build of `rows` x `d` for rocprofv3
(kernel trace / SQ counters) -- what binds the build kernel?  usage: python tools/hnsw_build_probe.py [rows=100000] [d=768] [efc=40]"""
import sys
import time

import torch

import mi355_faiss as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
efc = int(sys.argv[3]) if len(sys.argv) > 3 else 40
x = mf.synth_uniform_torch(n, d, 99, row0=0)
ix = mf.index_factory(d, "HNSW32", mf.METRIC_L2)
ix.set_ef_construction(efc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i0 in range(0, n, 2048):  # the glue's DataChunks
    ix.add_torch(x[i0 : i0 + 2048])
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"HNSW32 efConstruction {efc}: {n} uniform rows x {d} dims in {t:.2f} s = {n / t:.0f} rows/s; graph {ix.hnsw_graph_info() if hasattr(ix, 'hnsw_graph_info') else ''}")
