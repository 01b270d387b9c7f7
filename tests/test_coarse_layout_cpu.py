"""The LDS image of the IVF coarse quantiser's distance kernel (csrc/coarse_select.hip coarse_dist_mfma2_kernel, round 5), restated in
numpy: what every staging thread writes where, what every lane of the MFMA loop reads back, and that the two agree -- lane (ln, h) of
wave w gets, for MFMA step (block g, k-step j) of query block t, exactly x[32 t + ln][8 g + 2 j + h] and c[32 w + ln][8 g + 2 j + h],
i.e. every accumulator element is ONE chain over the dimensions in ascending order (IndexFlat's fvec_inner_product order, which
IndexIVF::search's quantizer->search computes: faiss/IndexIVF.cpp, reached from src/faiss_extension.cpp:631).  Also the bank rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS table) for the two instructions the layout was chosen for: ds_read_b128 conflict-free
in its four 16-lane groups, ds_write_b128 conflict-free in its eight 8-lane groups.  CPU only; the kernel itself is compared bit for
bit against the oracle in tests/test_coarse_matrix_gpu.py."""
import numpy as np
import pytest

K, P = 32, 36  # C2_K dims per slab, C2_P floats per LDS row


def interleave_rows(c):
    """csrc/common.h FlatGeom: a Flat index stores every four dims as [k0,k2,k1,k3] in rows with bit 4 clear, [k1,k3,k0,k2] with it set"""
    out = np.empty_like(c)
    for r in range(c.shape[0]):
        q = c[r].reshape(-1, 4)
        o = q[:, [1, 3, 0, 2]] if (r >> 4) & 1 else q[:, [0, 2, 1, 3]]
        out[r] = o.reshape(-1)
    return out


def stage(xt, ct_stored, c0, interleaved):
    """the 256 threads of a workgroup stage one slab: xt [128][32] query rows, ct_stored [128][32] centroid rows AS STORED.
    Returns the LDS image [256 * P] and the float offsets of every thread's four 16-byte writes (for the bank rule)."""
    lds = np.full(256 * P, np.nan, dtype=np.float32)
    writes = []  # (tid, float offset)
    for tid in range(256):
        r4, c8 = tid >> 2, tid & 3
        for i in range(2):
            base = (r4 + 64 * i) * P + 8 * c8
            a, b = xt[r4 + 64 * i, 8 * c8 : 8 * c8 + 4], xt[r4 + 64 * i, 8 * c8 + 4 : 8 * c8 + 8]
            lds[base : base + 4] = [a[0], a[2], b[0], b[2]]
            lds[base + 4 : base + 8] = [a[1], a[3], b[1], b[3]]
            u, v = ct_stored[r4 + 64 * i, 8 * c8 : 8 * c8 + 4], ct_stored[r4 + 64 * i, 8 * c8 + 4 : 8 * c8 + 8]
            ev, od = [u[0], u[2], v[0], v[2]], [u[1], u[3], v[1], v[3]]
            if interleaved:
                flip = ((c0 + r4 + 64 * i) >> 4) & 1
                lo, hi = [u[0], u[1], v[0], v[1]], [u[2], u[3], v[2], v[3]]
                ev, od = (hi, lo) if flip else (lo, hi)
            cb = (128 + r4 + 64 * i) * P + 8 * c8
            lds[cb : cb + 4] = ev
            lds[cb + 4 : cb + 8] = od
            writes += [(tid, base), (tid, base + 4), (tid, cb), (tid, cb + 4)]
    return lds, writes


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("c0", [0, 128, 16 * 7 * 8])
def test_every_lane_reads_its_dimensions_in_ascending_order(interleaved, c0):
    rs = np.random.RandomState(3 + c0)
    xt = rs.randn(128, K).astype(np.float32)
    ct = rs.randn(128, K).astype(np.float32)
    stored = ct.copy()
    if interleaved:  # the rows of the tile are rows c0 .. c0 + 127 of the store: the flip follows the GLOBAL row number
        big = np.zeros((c0 + 128, K), dtype=np.float32)
        big[c0:] = ct
        stored = interleave_rows(big)[c0:]
    lds, _ = stage(xt, stored, c0, interleaved)
    for wave in range(4):
        for lane in range(64):
            h, ln = lane >> 5, lane & 31
            xs = ln * P + 4 * h
            ys = xs + (128 + 32 * wave) * P
            dims = []
            for g in range(K // 8):  # block of 8 dims: one ds_read_b128 per operand
                bq = lds[ys + 8 * g : ys + 8 * g + 4]
                for j in range(4):  # k-step j of the block: v_mfma_f32_32x32x2_f32 takes dim 2 s + h from the lanes with l >> 5 == h
                    dim = 8 * g + 2 * j + h
                    dims.append(dim)
                    assert bq[j] == ct[32 * wave + ln, dim]
                    for t in range(4):
                        aq = lds[xs + 32 * t * P + 8 * g : xs + 32 * t * P + 8 * g + 4]
                        assert aq[j] == xt[32 * t + ln, dim]
            assert dims == sorted(dims) and dims == list(range(h, K, 2))  # with the partner half: 0, 1, 2, ... in MFMA order


def test_the_reads_and_writes_are_free_of_bank_conflicts():
    rs = np.random.RandomState(0)
    _, writes = stage(rs.randn(128, K).astype(np.float32), rs.randn(128, K).astype(np.float32), 0, False)
    # ds_read_b128: four groups of 16 lanes, one LDS cycle each when the 16 lanes hit 16 different 16-byte slots of the 256-byte bank row
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    for g in range(K // 8):
        for t in range(5):  # four query blocks + the centroid block (its base differs by a multiple of 32 rows: the same slots)
            for grp in groups:
                slots = {(((l & 31) * P + 4 * (l >> 5) + 32 * (t % 4) * P + 8 * g) // 4) % 16 for l in grp}
                assert len(slots) == 16, (g, t, grp)
    # ds_write_b128: eight groups of 8 contiguous lanes, bank = (byte address / 4) mod 32 per dword
    by_thread = {}
    for tid, off in writes:
        by_thread.setdefault(tid, []).append(off)
    for wave in range(4):
        for k in range(4):  # the k-th write instruction of the staging (query evens, query odds, centroid evens, centroid odds), first i only
            for g8 in range(8):
                banks = []
                for lane in range(8 * g8, 8 * g8 + 8):
                    off = by_thread[64 * wave + lane][k]
                    banks += [(off + e) % 32 for e in range(4)]
                assert len(set(banks)) == 32, (wave, k, g8)


def test_tiles_slabs_and_workgroups_cover_the_matrix_once():
    """the persistent launch: workgroup b takes tiles b, b + G, b + 2 G ...; tile -> (query block, centroid block); slabs of 32 dims"""
    for nq, nlist, d, G in [(10000, 4096, 128, 512), (129, 260, 36, 512), (77, 1024, 16, 512), (640, 1500, 96, 7)]:
        ntx = (nlist + 127) // 128
        ntiles = ntx * ((nq + 127) // 128)
        S = (d + K - 1) // K
        seen = np.zeros((ntiles, S), dtype=np.int32)
        for b in range(min(G, ntiles)):
            total = ((ntiles - 1 - b) // min(G, ntiles) + 1) * S
            tile, sl = b, 0
            for _ in range(total):
                seen[tile, sl] += 1
                sl += 1
                if sl == S:
                    sl, tile = 0, tile + min(G, ntiles)
        assert (seen == 1).all()
        cover = np.zeros((nq, nlist), dtype=np.int32)
        for tile in range(ntiles):
            q0, c0 = (tile // ntx) * 128, (tile % ntx) * 128
            cover[q0 : q0 + 128, c0 : c0 + 128] += 1
        assert (cover == 1).all()
