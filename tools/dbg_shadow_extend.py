"""debug: Flat shadow extended by add() -- compare every step with the exact f32 kernel"""
import torch

import mi355_faiss as mf

d, n0, step, nq, k = 128, 1_048_576, 65_536, 1024, 10
dev = torch.device("cuda", 0)
xb = mf.synth_clustered_torch(n0 + 3 * step, d, 1234, row0=0, n_centers=1024, sigma=0.1, device=dev)
xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=0.1, device=dev)
ix = mf.index_factory(d, "Flat", mf.METRIC_L2)
ex = mf.index_factory(d, "Flat", mf.METRIC_L2)
ex.set_option("prefilter", 0)
ex.set_option("flat_shadow", 0)
ix.set_option("flat_shadow", 1)
ix.add_torch(xb[:n0])
ex.add_torch(xb[:n0])
for i in range(4):
    D, I = ix.search_torch(xq, k)
    De, Ie = ex.search_torch(xq, k)
    torch.cuda.synchronize()
    nl = int((I != Ie).sum())
    nd = int((D.view(torch.int32) != De.view(torch.int32)).sum())
    bad = (I != Ie).any(dim=1).nonzero().flatten()[:5].tolist()
    print(i, ix.last_kernel_info()["name"], ix.shadow_stats(), "label diffs", nl, "dist diffs", nd, "queries", bad, flush=True)
    for q in bad[:2]:
        print("  got", I[q].tolist(), D[q].tolist())
        print("  ref", Ie[q].tolist(), De[q].tolist())
    if i < 3:
        ix.add_torch(xb[n0 + i * step : n0 + (i + 1) * step])
        ex.add_torch(xb[n0 + i * step : n0 + (i + 1) * step])
