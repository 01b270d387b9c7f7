// csrc/flat_mfma.hip -- K2/K3: brute-force L2 / inner-product search as an f32 MFMA contraction with the
// top-k select fused into the accumulator epilogue (the nq x N distance matrix never reaches HBM).
//
// Replaces, for nq >= 20 and no selector, what the reference reaches through
//   entry.index->search(...)            /root/reference/src/faiss_extension.cpp:631
//     -> faiss::IndexFlat::search -> knn_L2sqr / knn_inner_product -> exhaustive_*_blas   [UPSTREAM FAISS]
// i.e. ip = sgemm(Y, X); dis = |x|^2 + |y|^2 - 2 ip (clamped at 0); per-query k-best heap.
//
// Arithmetic contract (bit-exact with oracle/orc_core.c search_blas):
//   v_mfma_f32_32x32x2_f32 is a k-ordered f32 fma chain (one rounding per product), so the accumulator
//   equals fmaf(x[d-1],y[d-1], ... fmaf(x[0],y[0],0)); the epilogue computes (xn + yn) - 2*ip with the
//   same two roundings as the oracle.  Zero padding of d up to dp adds fma(0,0,acc) = acc.
//
// Mapping (CDNA4, wave64):
//   A operand = 32 database rows (M), B operand = 32 queries (N): D[i][j] lands with the QUERY on the
//   lane (col = lane&31) and 16 database rows in the lane's 16 accumulator registers
//   (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  So a lane tests its 16 distances against its own
//   query's current k-th best with no cross-lane traffic; insertions (rare: ~k ln(N/k) per query) go to a
//   per-query list in LDS owned by that wave.
//   Workgroup = 4 waves x 32 queries; the database tile is staged once in LDS (padded rows => conflict
//   free ds_read_b32 in natural k order) and shared by the 4 waves.
//   d <= 128: the wave's 32 queries stay resident in registers as B fragments for the whole kernel.
//   d  > 128: 256-row tiles (8 accumulator tiles per wave), k streamed in units of 32.
#include "flat_fused.h"

#include <algorithm>
#include <cstring>

#include <cstdlib>

namespace mvs {

// =====================================================================================================
// v2 resident kernel (d <= 128): database tile staged with LDS-DMA (global_load_lds, no staging VGPRs, no
// ds_write pass) and A fragments software-pipelined through a two-group register ring so that an MFMA never
// waits on the ds_read issued just before it.
// =====================================================================================================
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(1))) const float glb_f32;

// Measured dead end (round 2): issuing the A-fragment reads by hand (inline ds_read_b64 + s_waitcnt lgkmcnt(0)) removes the
// eight s_waitcnt vmcnt(0) per tile that hipcc puts in front of LDS reads while an LDS-DMA is in flight, but loses the
// ds_read2st64_b64 pairing: 212.9 vs 207.9 ms at the headline, 262 vs 260 ms at d = 768 (same box).  The other workgroup
// of the CU covers those waits here (a tile is 8192 MFMA cycles); in the 5x shorter tiles of flat_bf16.hip it does not.
// ABL (ablation builds for profiling only; results are WRONG when != 0): bit0 = skip the epilogue,
// bit1 = stage only the first tile, bit2 = reuse the first A-fragment group for every MFMA (no ds_reads)
// STREAM = false (d <= 128): NT = 2 (64-row tiles), one unit per tile, the wave's query fragments stay in registers.
// STREAM = true  (d  > 128): NT = 4 (128-row tiles), k streamed in units of KC = 64; the accumulators persist over the
//   units of a tile and the B fragments of the NEXT unit are refilled group by group behind the MFMAs that just
//   consumed the current ones (one 64-byte-per-lane register set, no double buffer).
// GL: the per-query k-lists live directly in the partial-result buffers in global memory (L2-resident; touched only by
// the rare insertion path) instead of LDS, so that k > 12 does not push the workgroup over half a CU's LDS.
template <int KSTEPS, bool IS_L2, int ABL = 0, int NT = 2, bool STREAM = false, bool SEL = false, bool ITEMS = false,
          bool GL = false, bool TIE = false>
__global__ __launch_bounds__(256, 2) void flat_mfma_resident_kernel(const MfmaArgs a) {
	constexpr int KC = 2 * KSTEPS, BN = 32 * NT;
	// LDS image of a tile: [64 rows][C 16-byte chunks], UNPADDED so that one LDS-DMA dwordx4 instruction (1 KiB per
	// wave) lands whole; bank conflicts are removed by an XOR swizzle of the chunk position, applied on the SOURCE
	// address of the DMA and again on the ds_read_b128 address (cdna_hip_programming.md rule 21):
	//   chunk cg of row r lives at position cg ^ f(r),  f(r) = (r / R) & (min(C,16)-1),  R = max(1, 16/C)
	// => the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row.
	constexpr int C = KC / 4;                    // chunks per row
	constexpr int R = C >= 16 ? 1 : 16 / C;      // rows per 256-byte bank row
	constexpr int FM = (C >= 16 ? 16 : C) - 1;   // swizzle mask
	constexpr int NDMA = (BN * KC * 4) / 1024;   // LDS-DMA instructions per tile (whole workgroup)
	constexpr int CG = C >= 2 ? 2 : 1;           // chunks per A-fragment group (= 4 k-steps)
	constexpr int NG = C / CG;
	static_assert(KSTEPS % 4 == 0 && (BN * KC * 4) % 1024 == 0, "");

	extern __shared__ __attribute__((aligned(16))) float smem[];
	float *tbuf = smem;               // [2][BN][KC]  (swizzled, see above)
	float *nbuf = smem + 2 * BN * KC; // [2][BN]
	float *ld = nbuf + 2 * BN;        // [128][k]  (absent with GL)
	int *li = (int *)(ld + (GL ? 0 : QBLOCK * a.k));
	float *lthr = (float *)(li + (GL ? 0 : QBLOCK * a.k));
	int *lthrid = (int *)(lthr + QBLOCK);
	int *lpos = lthrid + QBLOCK;

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // provably wave-uniform: stays in SGPRs
	const int h = lane >> 5, c = lane & 31;
	const int k = a.k;
	int split = 0, qb = 0, q;
	bool qvalid;
	long long r_begin, r_end;
	const int ql = wave * WAVE_Q + c;
	int qblk32;
	if (ITEMS) {
		if (a.nitems_dev && (int)blockIdx.x >= *a.nitems_dev)
			return; // whole workgroup, before any barrier
		const int4 it = a.items[blockIdx.x];
		r_begin = it.x;
		r_end = it.y;
		qvalid = ql < it.w;
		q = qvalid ? a.qidx[it.z + ql] : 0;
		qblk32 = blockIdx.x * 4 + wave; // the item's queries were packed for it (pack_item_queries_kernel)
	} else {
		if (a.xcd_map) {
			const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
			split = (idx / a.nqb) * 8 + xcd;
			qb = idx % a.nqb;
		} else {
			split = blockIdx.x / a.nqb;
			qb = blockIdx.x % a.nqb;
		}
		q = qb * QBLOCK + ql;
		qvalid = q < a.nq;
		qblk32 = qb * 4 + wave;
		r_begin = (long long)split * a.split_rows;
		r_end = r_begin + a.split_rows;
	}
	const int nwin = a.slot_stride >> 4; // 16-slot windows of the shared threshold slots
	if (r_end > a.n)
		r_end = a.n;
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin + BN - 1) / BN) : 0;
	const int nch = STREAM ? a.nch : 1;
	const int nunits = ntiles * nch;

	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	float thr = qvalid ? neutral : (IS_L2 ? -INFINITY : INFINITY);
	const size_t ob = ITEMS ? ((size_t)blockIdx.x * QBLOCK + ql) * k : ((size_t)split * a.nq + (qvalid ? q : 0)) * k;
	float *ldq = GL ? a.pd + ob : ld + ql * k;
	int *liq = GL ? (int *)(a.pi + ob) : li + ql * k;
	if (h == 0) {
		if (!GL || qvalid)
			for (int j = 0; j < k; ++j) {
				ldq[j] = neutral;
				liq[j] = -1;
			}
		lthr[ql] = thr;
		lthrid[ql] = -1;
		lpos[ql] = 0;
	}
	const float xnq = (IS_L2 && qvalid) ? ((ITEMS && a.item_qn) ? a.item_qn[(size_t)blockIdx.x * QBLOCK + ql] : a.qn[q]) : 0.f;

	float qf[KSTEPS];
	const float4 *qsrc = (const float4 *)a.qf + (size_t)qblk32 * nch * (KSTEPS / 4) * 64 + lane;
#pragma unroll
	for (int s4 = 0; s4 < KSTEPS / 4; ++s4) {
		float4 v = qsrc[s4 * 64];
		qf[4 * s4 + 0] = v.x;
		qf[4 * s4 + 1] = v.y;
		qf[4 * s4 + 2] = v.z;
		qf[4 * s4 + 3] = v.w;
	}

	// LDS-DMA staging: instruction `inst` of a tile fills LDS bytes [1024 inst, 1024 inst + 1024); lane l owns
	// chunk slot 64 inst + l.  Wave w issues instructions w, w+4, ... -- ONE per A-fragment group of the MFMA loop,
	// so that the ~100-cycle issue cost of an LDS-DMA instruction hides under the MFMA issued just before it.
	constexpr int DMA_PER_WAVE = (NDMA + 3) / 4;
	static_assert(DMA_PER_WAVE <= NG, "one LDS-DMA instruction per MFMA group");
	auto dma_issue = [&](int u, int i) {
		const int inst = i * 4 + wave;
		if (NDMA % 4 == 0 || inst < NDMA) {
			const int tile = STREAM ? u / nch : u, ch = STREAM ? u - tile * nch : 0;
			const long long row0 = r_begin + (long long)tile * BN;
			// per-lane part recomputed at every issue from an opaque copy of the lane id: hipcc would otherwise hoist
			// the DMA_PER_WAVE loop-invariant offsets out of the tile loop and spill them (and reload them with a
			// vmcnt(0) in the MFMA loop)
			int lane_o = lane;
			MVS_OPAQUE_VGPR(lane_o);
			const int L = inst * 64 + lane_o; // chunk slot in the LDS image
			int r = L / C;
			const int p = L % C;
			const int cg = p ^ ((r / R) & FM);
			const long long lim = a.n - 1 - row0; // tail tile: clamp (rows >= nvalid are masked in the epilogue)
			r = r < lim ? r : (int)lim;
			const char *base = (const char *)(a.yb + (size_t)row0 * a.dp + ch * KC); // wave-uniform
			const unsigned boff = (unsigned)(r * a.dp + cg * 4) * 4u;
			__builtin_amdgcn_global_load_lds((glb_f32 *)(base + boff), (lds_f32 *)smem + (u & 1) * BN * KC + inst * 256,
			                                 16, 0, 0);
		}
	};
	auto dma_norms = [&](int tile) {
		if (IS_L2 && !TIE && wave < BN / 64) {
			long long gr = r_begin + (long long)tile * BN + wave * 64 + lane;
			if (gr >= a.n)
				gr = a.n - 1;
			__builtin_amdgcn_global_load_lds((glb_f32 *)(a.yn + gr),
			                                 (lds_f32 *)smem + 2 * BN * KC + (tile & 1) * BN + wave * 64, 4, 0, 0);
		}
	};

	f32x16 acc[NT];
	SlotBound sbound;
	SlotRegs sr;
	if (nunits > 0) {
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i)
			dma_issue(0, i);
		dma_norms(0);
	}
	__syncthreads();

	// ITEMS: a wave whose 32 query slots are all empty (item with <= 96 queries) only helps with the staging and keeps
	// in step at the barriers; its SIMD is left to the other workgroup of the CU
	bool wave_idle = false;
	if (ITEMS)
		wave_idle = wave * WAVE_Q >= a.items[blockIdx.x].w;
	for (int u = 0; u < nunits; ++u) {
		const int tile = STREAM ? u / nch : u, ch = STREAM ? u - tile * nch : 0;
		const bool stage_next = u + 1 < nunits && !(ABL & 2);
		const int window = tile % nwin;
		if (ITEMS && wave_idle) {
			if (stage_next) {
#pragma unroll
				for (int g = 0; g < DMA_PER_WAVE; ++g)
					dma_issue(u + 1, g);
			}
			if (ch == 0 && tile + 1 < ntiles)
				dma_norms(tile + 1);
			__syncthreads();
			continue;
		}
		if (ch == 0) {
#pragma unroll
			for (int t = 0; t < NT; ++t)
#pragma unroll
				for (int r = 0; r < 16; ++r)
					acc[t][r] = 0.f;
		}
		// A fragments: the database rows are PAIR-INTERLEAVED in HBM (FlatGeom::pair_interleaved), so the two k-steps
		// a lane half needs from chunk cg -- k = 4cg + h and k = 4cg + 2 + h -- are one aligned 8-byte word at byte
		// 8 * (h ^ bit4(row)) of the chunk: ds_read_b64, no selects, 32 distinct bank pairs per lane group.
		const float *Abase = tbuf + ((ABL & 2) ? 0 : (u & 1)) * BN * KC;
		const int fsw = (c / R) & FM;
		static_assert((32 / R) % (FM + 1) == 0, "row t*32 + c must have the same swizzle as row c");
		const int hoff = 2 * (h ^ ((c >> 4) & 1));
		float2 af[2][NT][CG];
#pragma unroll
		for (int t = 0; t < NT; ++t)
#pragma unroll
			for (int j = 0; j < CG; ++j)
				af[0][t][j] = *(const float2 *)(Abase + (t * 32 + c) * KC + ((j ^ fsw) * 4) + hoff);
		// B fragments of the next unit (STREAM): same query block, next k range
		const int un = u + 1 < nunits ? u + 1 : u;
		const float4 *qnext = qsrc + (size_t)((STREAM ? un % nch : 0) * (KSTEPS / 4)) * 64;
#pragma unroll
		for (int g = 0; g < NG; ++g) {
			// Order pinned with sched_barrier: [first k-step of group g] [ds_reads of group g+1, one LDS-DMA piece of
			// the next unit] [rest of group g].  hipcc otherwise sinks each ds_read to just before its MFMA (every
			// MFMA then waits out the LDS latency); issuing the next group's reads one k-step INTO the group puts
			// >= 6 MFMAs (384 cycles) between those reads and the wait at the head of the next group.
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int t = 0; t < NT; ++t)
				acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][t][0].x, qf[g * 2 * CG], acc[t], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
			if (g + 1 < NG) {
				// the swizzled chunk offsets are recomputed per group from an opaque copy of the swizzle (2 VALU per
				// read pair): hipcc would otherwise hoist all C of them out of the tile loop and spill
				int fo = fsw;
				MVS_OPAQUE_VGPR(fo);
#pragma unroll
				for (int t = 0; t < NT; ++t)
#pragma unroll
					for (int j = 0; j < CG; ++j)
						af[(g + 1) & 1][t][j] =
						    (ABL & 4) ? af[g & 1][t][j]
						              : *(const float2 *)(Abase + (t * 32 + c) * KC + ((((g + 1) * CG + j) ^ fo) * 4) + hoff);
			}
			if (g < DMA_PER_WAVE && stage_next)
				dma_issue(u + 1, g);
			if (g == 0 && ch == 0) {
				if (tile + 1 < ntiles && !(ABL & 2))
					dma_norms(tile + 1);
				// shared threshold slots of this lane's query: issued now, reduced in the epilogue
				slots_prefetch(sr, a.gslot + (size_t)(qvalid ? q : 0) * a.slot_stride, window, h);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int t = 0; t < NT; ++t)
				acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][t][0].y, qf[g * 2 * CG + 1], acc[t], 0, 0, 0);
			if (CG == 2) {
#pragma unroll
				for (int t = 0; t < NT; ++t)
					acc[t] =
					    __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][t][CG - 1].x, qf[g * 2 * CG + 2], acc[t], 0, 0, 0);
#pragma unroll
				for (int t = 0; t < NT; ++t)
					acc[t] =
					    __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][t][CG - 1].y, qf[g * 2 * CG + 3], acc[t], 0, 0, 0);
			}
			if (STREAM) {
				// rolling refill: the MFMAs above were the last readers of this group's B registers in this unit
				static_assert(!STREAM || CG == 2, "one float4 of query fragments per group");
				const float4 v = qnext[g * 64];
				qf[4 * g + 0] = v.x;
				qf[4 * g + 1] = v.y;
				qf[4 * g + 2] = v.z;
				qf[4 * g + 3] = v.w;
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		if (ch == nch - 1) {
			const long long row0 = r_begin + (long long)tile * BN;
			const int nvalid = (int)((r_end - row0) < BN ? (r_end - row0) : BN);
			if (ABL & 1) {
#pragma unroll
				for (int t = 0; t < NT; ++t)
					MVS_KEEP_VGPR(acc[t]); // keep the MFMA chain alive
			} else {
				const unsigned gkey = slots_update(sbound, slots_reduce(sr), window, nwin);
				unsigned long long rowmask[(NT + 1) / 2];
				if (SEL) { // lane l tests rows l, l + 64, ... of the tile once; every lane then reads the ballots
#pragma unroll
					for (int m = 0; m < (NT + 1) / 2; ++m) {
						const long long row = row0 + m * 64 + lane;
						bool ok = m * 64 + lane < nvalid;
						if (ok) {
							const long long lab = a.rowids ? a.rowids[row] : row;
							ok = mfma_sel_member(a.sel, a.idmap ? a.idmap[lab] : lab);
						}
						rowmask[m] = __builtin_amdgcn_ballot_w64(ok);
					}
				}
				tile_epilogue<NT, IS_L2, (ABL & 8) != 0, SEL, TIE, false, GL ? 2 : 1>(acc, nbuf + (tile & 1) * BN, row0, nvalid, xnq, thr, qvalid, gkey,
				                                              a.gslot + (size_t)(qvalid ? q : 0) * a.slot_stride, ldq, liq, k,
				                                              lthr + ql, lthrid + ql, lpos + ql, h, rowmask);
			}
		}
		__syncthreads(); // also drains this unit's LDS-DMA (vmcnt(0)) before the next unit reads it
	}

	if (!GL && h == 0 && qvalid) {
		float *od = a.pd + ob;
		int32_t *oi = a.pi + ob;
		for (int j = 0; j < k; ++j) {
			od[j] = ldq[j];
			oi[j] = liq[j];
		}
	}
}

// -------------------------------------------------------------------------------------------------

FlatGeom flat_geom_for(int d) {
	FlatGeom g;
	g.d = d;
	if (d <= 128) {
		static const int kcs[] = {8, 16, 32, 64, 128};
		g.kc = 128;
		for (int v : kcs)
			if (v >= d) {
				g.kc = v;
				break;
			}
		g.dp = g.kc;
		g.nch = 1;
		g.ntile = 2;
		g.pair_interleaved = true;
	} else {
		g.kc = 64;
		g.dp = (d + 63) / 64 * 64;
		g.nch = g.dp / 64;
		g.ntile = 4;
		g.pair_interleaved = true;
	}
	return g;
}

size_t qfrag_floats(const FlatGeom &g, int64_t nq) {
	const int64_t nblk32 = (nq + QBLOCK - 1) / QBLOCK * 4;
	return (size_t)nblk32 * 32 * g.dp;
}

static size_t mfma_lds_bytes(const FlatGeom &g, int64_t k, bool global_lists = false) {
	const size_t bn = g.bn();
	const size_t lda = (size_t)g.kc; // unpadded, swizzled rows
	return (2 * bn * lda + 2 * bn) * sizeof(float) + (global_lists ? 0 : (size_t)QBLOCK * k * 8) + QBLOCK * 12;
}

// largest k with the k-lists in LDS (selector and IVF item instances) / at all (lists in global memory beyond 12)
int64_t flat_mfma_max_k_lds(const FlatGeom &g) {
	const size_t fixed = mfma_lds_bytes(g, 0);
	return (int64_t)((160 * 1024 - fixed) / (QBLOCK * 8));
}
int64_t flat_mfma_max_k(const FlatGeom &) {
	return 256;
}

                             // with LDS lists, measured k = 16...80); 0 / -1: only when the LDS cannot hold them (k > 88)

FlatSearchPlan plan_flat_mfma(const FlatGeom &g, int64_t nq, int64_t n, int64_t k) {
	FlatSearchPlan p;
	p.nqb = (int)((nq + QBLOCK - 1) / QBLOCK);
	const int bn = g.bn();
	const int64_t ntiles = (n + bn - 1) / bn;
	// The grid is nqb x nsplit workgroups over 512 resident slots (2 per CU).  Thresholds are shared across
	// workgroups, so extra splits cost little; what matters is that the LAST round of workgroups is nearly full:
	// pick the split count (a multiple of 8: one XCD per split residue, see xcd_map) whose grid wastes the least.
	const int64_t slots = 2 * 256;
	const int64_t min_tiles = 16; // amortise the per-workgroup prologue
	int64_t max_split = ntiles / min_tiles;
	// few query blocks (small batches routed here for inner product): allow enough splits to fill the 512 slots
	int64_t split_cap = std::max<int64_t>(k <= 16 ? 256 : 128, std::min<int64_t>(512, slots / p.nqb));
	// K4 (merge_partials_kernel) holds nsplit*k candidates of one query in LDS
	split_cap = std::max<int64_t>(8, std::min<int64_t>(split_cap, (int64_t)(150 * 1024 / 8) / std::max<int64_t>(k, 1) - 1));
	if (max_split > split_cap)
		max_split = split_cap;
	int64_t nsplit = 1;
	p.xcd_map = false;
	if (tune().mfma_nsplit > 0) {
		nsplit = tune().mfma_nsplit;
	} else if (max_split >= 8) {
		double best_eff = -1;
		for (int64_t s = 8; s <= max_split; s += 8) {
			const int64_t w = s * p.nqb;
			if (w < slots && s + 8 <= max_split)
				continue; // fill the chip first
			const int64_t rounds = (w + slots - 1) / slots;
			double eff = (double)w / (double)(rounds * slots);
			// mild preference for >= 3 rounds (tail of the last round averages out)
			if (rounds < 3)
				eff -= 0.03 * (3 - rounds);
			// ... and, at equal efficiency, for MORE splits while k <= 16 and a split keeps >= 128 tiles (more rounds
			// average the tail better: +1.9 % at the headline, 32 -> 128 splits), for FEWER otherwise (every
			// (query, split) pair pays its own cold-start insertions: k = 100 is 50 % slower at 128 splits than at 32)
			eff += (k <= 16 && s * 128 <= ntiles ? 1e-4 : -1e-4) * s;
			if (eff > best_eff) {
				best_eff = eff;
				nsplit = s;
			}
		}
	} else if (max_split >= 1) {
		nsplit = max_split;
	}
	if (nsplit >= 8 && nsplit % 8 == 0)
		p.xcd_map = true;
	int64_t tiles_per_split = ntiles > 0 ? (ntiles + nsplit - 1) / nsplit : 1;
	p.split_rows = tiles_per_split * bn;
	p.nsplit = (int)nsplit;
	p.grid = p.nqb * p.nsplit;
	// two workgroups per CU need <= ~80 KB each: with the 64 KB of tile buffers that is k <= 12 for LDS-resident lists
	p.global_lists = tune().mfma_global_lists < 0 ? k > flat_mfma_max_k_lds(g) : (tune().mfma_global_lists > 0 && k > 12);
	if (k > flat_mfma_max_k_lds(g))
		p.global_lists = true;
	p.lds_bytes = mfma_lds_bytes(g, k, p.global_lists);
	return p;
}

// slots [0,k) of every query = key of the neutral value (nothing found yet); padding slots = 0
__global__ void init_gslot_kernel(unsigned *g, long long total, int stride, int k, int is_l2) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total)
		g[i] = (int)(i % stride) < k ? (is_l2 ? f2key(FLT_MAX) : ~f2key(-FLT_MAX)) : 0u;
}
int flat_mfma_slot_stride(int64_t k) {
	return (int)((k + 15) / 16 * 16);
}


template <int KSTEPS>
static void launch_resident_v2(int metric, const MfmaArgs &a, const FlatSearchPlan &p, hipStream_t st) {
#ifdef MVS_PROFILING // wrong-result ablation instances exist only in the profiling library (make profiling -> libmi355faiss_prof.so)
	if (KSTEPS == 64 && metric == METRIC_L2 && tune().mfma_variant >= 100) { // profiling ablations
		const int abl = tune().mfma_variant - 100;
#define MVS_ABL(N)                                                                                                     \
	if (abl == N) {                                                                                                    \
		auto kern = flat_mfma_resident_kernel<64, true, N>;                                                            \
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes)); \
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);                                          \
	}
		MVS_ABL(1) MVS_ABL(2) MVS_ABL(3) MVS_ABL(8) MVS_ABL(10)
#undef MVS_ABL
		MVS_HIP(hipGetLastError());
		return;
	}
#endif
	if (p.global_lists && a.sel.kind == MVS_SEL_NONE) {
		if (metric == METRIC_L2) {
			auto kern = flat_mfma_resident_kernel<KSTEPS, true, 0, 2, false, false, false, true>;
			ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
			hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
		} else {
			auto kern = flat_mfma_resident_kernel<KSTEPS, false, 0, 2, false, false, false, true>;
			ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
			hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
		}
		MVS_HIP(hipGetLastError());
		return;
	}
	if (metric == METRIC_L2) {
		auto kern = flat_mfma_resident_kernel<KSTEPS, true>;
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
		if (getenv("MVS_DEBUG")) {
			int nb = 0;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, 256, p.lds_bytes);
			fprintf(stderr, "[mvs] flat_mfma_resident_kernel<%d,L2> grid=%d lds=%zu nsplit=%d split_rows=%lld blocks/CU=%d\n",
			        KSTEPS, p.grid, p.lds_bytes, p.nsplit, (long long)p.split_rows, nb);
		}
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
	} else if (a.sel.kind != MVS_SEL_NONE) {
		auto kern = flat_mfma_resident_kernel<KSTEPS, false, 0, 2, false, true>;
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
	} else {
		auto kern = flat_mfma_resident_kernel<KSTEPS, false>;
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
	}
	MVS_HIP(hipGetLastError());
}

template <int KSTEPS, int NT, bool RESIDENT>
static void launch_inst(int metric, const MfmaArgs &a, const FlatSearchPlan &p, hipStream_t st) {
	if constexpr (RESIDENT) {
		launch_resident_v2<KSTEPS>(metric, a, p, st);
	} else {
		if (p.global_lists && a.sel.kind == MVS_SEL_NONE) {
			if (metric == METRIC_L2) {
				auto kern = flat_mfma_resident_kernel<KSTEPS, true, 0, NT, true, false, false, true>;
				ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
				hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
			} else {
				auto kern = flat_mfma_resident_kernel<KSTEPS, false, 0, NT, true, false, false, true>;
				ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
				hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
			}
			MVS_HIP(hipGetLastError());
			return;
		}
		if (metric == METRIC_L2) {
			auto kern = flat_mfma_resident_kernel<KSTEPS, true, 0, NT, true>;
			ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
			hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
		} else if (a.sel.kind != MVS_SEL_NONE) {
			auto kern = flat_mfma_resident_kernel<KSTEPS, false, 0, NT, true, true>;
			ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
			hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
		} else {
			auto kern = flat_mfma_resident_kernel<KSTEPS, false, 0, NT, true>;
			ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
			hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
		}
		MVS_HIP(hipGetLastError());
	}
}

void launch_flat_mfma(const FlatGeom &g, const FlatSearchPlan &p_in, int metric, const float *d_qf, const float *d_qnorm,
                      int64_t nq, FlatDB db, int64_t k, float *d_pd, int32_t *d_pi, unsigned *d_gthr, hipStream_t st,
                      const SelectorDev *sel, const int64_t *d_idmap) {
	if (nq <= 0)
		return;
	const int stride = flat_mfma_slot_stride(k);
	const long long gtotal = (long long)nq * stride;
	if (gtotal > 0)
		hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gthr, gtotal,
		                   stride, (int)k, metric == METRIC_L2 ? 1 : 0);
	FlatSearchPlan p = p_in;
	MfmaArgs a;
	memset(&a, 0, sizeof a);
	if (sel && sel->kind != MVS_SEL_NONE) {
		if (k > flat_mfma_max_k_lds(g))
			throw_faiss("mvs::launch_flat_mfma", __FILE__, "k = %lld too large for the selector instances", (long long)k);
		p.global_lists = false; // the SEL instances keep their k-lists in LDS
		p.lds_bytes = mfma_lds_bytes(g, k, false);
		if (metric != METRIC_IP)
			throw_faiss("mvs::launch_flat_mfma", __FILE__, "the fused kernel takes a selector for inner product only");
		a.sel = *sel;
		a.idmap = (const long long *)d_idmap;
	}
	a.gslot = d_gthr;
	a.slot_stride = stride;
	a.qf = d_qf;
	a.qn = d_qnorm;
	a.yb = db.vecs;
	a.yn = db.norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = db.n;
	a.split_rows = p.split_rows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.nqb = p.nqb;
	a.nsplit = p.nsplit;
	a.dp = g.dp;
	a.nch = g.nch;
	a.xcd_map = p.xcd_map ? 1 : 0;
	auto launch_one = [&](const MfmaArgs &aa, const FlatSearchPlan &pp) {
		if (g.nch == 1) {
			switch (g.kc) {
			case 8:
				launch_inst<4, 2, true>(metric, aa, pp, st);
				break;
			case 16:
				launch_inst<8, 2, true>(metric, aa, pp, st);
				break;
			case 32:
				launch_inst<16, 2, true>(metric, aa, pp, st);
				break;
			case 64:
				launch_inst<32, 2, true>(metric, aa, pp, st);
				break;
			default:
				launch_inst<64, 2, true>(metric, aa, pp, st);
				break;
			}
		} else {
			launch_inst<32, 4, false>(metric, aa, pp, st); // d > 128: 128-row tiles, k streamed in units of 64
		}
	};
	// Optional threshold warm-up (option mfma_warm = divisor): a pre-pass of the same kernel over the first
	// n/divisor rows leaves the shared class slots holding valid bounds before all workgroups start cold.
	if (tune().mfma_warm > 1 && db.n / tune().mfma_warm >= 4096) {
		const int64_t n_pre = db.n / tune().mfma_warm;
		FlatSearchPlan pp = plan_flat_mfma(g, nq, n_pre, k);
		if (pp.nsplit <= p.nsplit) {
			MfmaArgs ap = a;
			ap.n = n_pre;
			ap.split_rows = pp.split_rows;
			ap.nqb = pp.nqb;
			ap.nsplit = pp.nsplit;
			ap.xcd_map = pp.xcd_map ? 1 : 0;
			launch_one(ap, pp);
		}
	}
	launch_one(a, p);
}

// ---- tie pass (inner product): the k smallest row ids with score >= T_q, per query -------------------------------
// Same contraction as the search itself (so the scores are bit-identical to the ones the boundary value T came from),
// TIE epilogue, lists in global memory.  d_T[q] travels in the query-norm slot.  Partial lists as in launch_flat_mfma:
// pd holds 0 / FLT_MAX, pi the row ids.
template <int KSTEPS, int NT, bool STREAM>
static void launch_tie_inst(const MfmaArgs &a, const FlatSearchPlan &p, hipStream_t st) {
	auto kern = flat_mfma_resident_kernel<KSTEPS, true, 0, NT, STREAM, true, false, true, true>;
	ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));
	hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);
	MVS_HIP(hipGetLastError());
}
void launch_flat_mfma_tie(const FlatGeom &g, const FlatSearchPlan &p_in, const float *d_qf, const float *d_T, int64_t nq,
                          FlatDB db, int64_t k, float *d_pd, int32_t *d_pi, unsigned *d_gthr, hipStream_t st,
                          const SelectorDev *sel, const int64_t *d_idmap) {
	if (nq <= 0)
		return;
	const int stride = flat_mfma_slot_stride(k);
	const long long gtotal = (long long)nq * stride;
	hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gthr, gtotal, stride,
	                   (int)k, 1);
	FlatSearchPlan p = p_in;
	p.global_lists = true;
	p.lds_bytes = mfma_lds_bytes(g, k, true);
	MfmaArgs a;
	memset(&a, 0, sizeof a);
	if (sel)
		a.sel = *sel;
	a.idmap = (const long long *)d_idmap;
	a.gslot = d_gthr;
	a.slot_stride = stride;
	a.qf = d_qf;
	a.qn = d_T;
	a.yb = db.vecs;
	a.yn = db.norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = db.n;
	a.split_rows = p.split_rows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.nqb = p.nqb;
	a.nsplit = p.nsplit;
	a.dp = g.dp;
	a.nch = g.nch;
	a.xcd_map = p.xcd_map ? 1 : 0;
	if (g.nch == 1) {
		switch (g.kc) {
		case 8:
			launch_tie_inst<4, 2, false>(a, p, st);
			break;
		case 16:
			launch_tie_inst<8, 2, false>(a, p, st);
			break;
		case 32:
			launch_tie_inst<16, 2, false>(a, p, st);
			break;
		case 64:
			launch_tie_inst<32, 2, false>(a, p, st);
			break;
		default:
			launch_tie_inst<64, 2, false>(a, p, st);
			break;
		}
	} else {
		launch_tie_inst<32, 4, true>(a, p, st);
	}
}

// ---- IVF list scan as a segmented variant of the fused kernel ---------------------------------------------------
// Work item = (row segment of one inverted list, <= 128 of the queries probing it); rows in the Flat storage format
// (pair-interleaved, lists padded to 64 rows), the items' queries packed into B-fragment order per item.  Distances
// follow the Flat BLAS-branch arithmetic (inner product: the exact k-ordered chain; L2: ||x||^2 + ||y||^2 - 2<x,y>).
bool flat_mfma_items_supported(const FlatGeom &g, int64_t k) {
	return (g.nch > 1 || g.kc >= 64) && k <= flat_mfma_max_k_lds(g);
}
size_t flat_mfma_item_query_floats(const FlatGeom &g, int max_items) {
	return (size_t)max_items * QBLOCK * g.dp;
}
int flat_mfma_item_slots() {
	return QBLOCK;
}
template <int KSTEPS, int NT, bool STREAM>
static void launch_items_inst(int metric, bool has_sel, const MfmaArgs &a, int grid, size_t lds, hipStream_t st) {
#define MVS_ITEMS(L2, SEL)                                                                                             \
	{                                                                                                                  \
		auto kern = flat_mfma_resident_kernel<KSTEPS, L2, 0, NT, STREAM, SEL, true>;                                   \
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));        \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                                   \
	}
	if (metric == METRIC_L2) {
		if (has_sel)
			MVS_ITEMS(true, true)
		else
			MVS_ITEMS(true, false)
	} else {
		if (has_sel)
			MVS_ITEMS(false, true)
		else
			MVS_ITEMS(false, false)
	}
#undef MVS_ITEMS
	MVS_HIP(hipGetLastError());
}
void launch_flat_mfma_items(const FlatGeom &g, int metric, const float *d_qf, const float *d_qnorm, int64_t nq,
                            const float *d_rows, const float *d_norms, int64_t nrows, int64_t k, const void *d_items,
                            const int *d_nitems, int max_items, const int *d_qidx, const int64_t *d_rowids,
                            const SelectorDev *sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gthr,
                            hipStream_t st, const float *d_item_qn) {
	if (max_items <= 0)
		return;
	const int stride = flat_mfma_slot_stride(k);
	const long long gtotal = (long long)nq * stride;
	if (gtotal > 0)
		hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gthr, gtotal,
		                   stride, (int)k, metric == METRIC_L2 ? 1 : 0);
	MfmaArgs a;
	memset(&a, 0, sizeof a);
	const bool has_sel = sel && sel->kind != MVS_SEL_NONE;
	if (has_sel)
		a.sel = *sel;
	a.idmap = (const long long *)d_idmap;
	a.gslot = d_gthr;
	a.slot_stride = stride;
	a.qf = d_qf;
	a.qn = d_qnorm;
	a.yb = d_rows;
	a.yn = d_norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = nrows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.dp = g.dp;
	a.nch = g.nch;
	a.items = (const int4 *)d_items;
	a.nitems_dev = d_nitems;
	a.qidx = d_qidx;
	a.rowids = (const long long *)d_rowids;
	a.item_qn = d_item_qn;
	const size_t lds = mfma_lds_bytes(g, k);
	if (g.nch > 1)
		launch_items_inst<32, 4, true>(metric, has_sel, a, max_items, lds, st);
	else if (g.kc == 64)
		launch_items_inst<32, 2, false>(metric, has_sel, a, max_items, lds, st);
	else
		launch_items_inst<64, 2, false>(metric, has_sel, a, max_items, lds, st);
}

} // namespace mvs
