import sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/duckdb-faiss-ext_amd/pyhost")
import torch, mi355_faiss as mf
ix = mf.index_factory(128, "Flat", 1)
n = 10_000_000
for s0 in range(0, n, 1 << 20):
    ix.add_torch(mf.synth_uniform_torch(min(1 << 20, n - s0), 128, 1234, row0=s0)); torch.cuda.synchronize()
xq = mf.synth_uniform_torch(10000, 128, 4321)
L = mf.lib(); out = (C.c_ulonglong * 4)()
for seed in (0, 1):
  ix.set_option("pf_classes32", seed)
  ix.search_torch(xq, 10); torch.cuda.synchronize()
  L.mvs_debug_counters(out, 1)
  ix.search_torch(xq, 10); torch.cuda.synchronize()
  L.mvs_debug_counters(out, 1)
  print("seed", seed, end=" ")
  print("cycles_in_rare", out[3], "per event", out[3] / max(out[0], 1), "events", out[0], "candidates", out[1], "total_wave_cycles", out[2], "rare share", out[3] / max(out[2], 1), "wave-tiles", 312500 * 40 * 4)
