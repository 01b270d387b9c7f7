"""Cross-shard merge ON THE DEVICE (mvs_merge_records_device, csrc/util_kernels.hip merge_records_kernel) must equal the host
merges it stands in for (mvs_merge_shards / mvs_merge_shards_raw, csrc/merge_host.hip): same labels, same distances, same
FAISS print order, with exact ties across shards, empty slots and k > candidates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("G,nq,kk", [(2, 100, 10), (8, 3000, 11), (5, 64, 1), (3, 50, 40), (64, 20, 16)])
def test_device_merge_equals_host_merge(mf, metric, G, nq, kk):
    import torch

    import sharded

    rs = np.random.RandomState(G * 1000 + kk)
    D = rs.randint(0, 30, size=(G, nq, kk)).astype(np.float32) / 4  # few distinct values: ties within and across shards
    I = rs.permutation(G * nq * kk * 2)[: G * nq * kk].reshape(G, nq, kk).astype(np.int64)
    I[rs.rand(G, nq, kk) < 0.1] = -1  # empty slots
    I[:, 0] = -1  # a query with no candidate at all
    rec = sharded.pack_records(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda())
    for kout in {kk, max(1, kk - 1)}:
        Dd, Id = mf.merge_records_torch(metric, rec, kout)
        Dh, Ih = mf.merge_shards(metric, D, I)
        assert np.array_equal(Id.cpu().numpy(), Ih[:, :kout]) or kout < kk
        if kout == kk:
            assert np.array_equal(Dd.cpu().numpy()[Ih >= 0], Dh[Ih >= 0])
        Dr, Ir = mf.merge_records_torch(metric, rec, kout, raw=True)
        Dhr, Ihr = mf.merge_shards_raw(metric, D, I)
        assert np.array_equal(Ir.cpu().numpy(), Ihr[:, :kout])
        ok = Ihr[:, :kout] >= 0
        assert np.array_equal(Dr.cpu().numpy()[ok], Dhr[:, :kout][ok])
