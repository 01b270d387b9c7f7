// csrc/flat_fused.h -- device-side pieces shared by the fused distance + top-k kernels (flat_mfma.hip: exact f32
// contraction; flat_bf16.hip: bf16x3 prefilter): kernel arguments, order-preserving keys, the per-query k-lists (heaps
// on (value, id)), cross-workgroup threshold slots and the accumulator epilogue.  See DESIGN.md 3.1-3.3.
#pragma once
#include "common.h"

#include "../../include/mi355_faiss.h"

namespace mvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Register-opacity / keep-alive helpers.  The "v" (VGPR) constraint only exists on the device pass; on the HOST pass
// of a kernel TEMPLATE an invalid constraint on a dependent type silently invalidates the instantiation and hipcc
// (ROCm 7.2) then emits no host stub for the kernel (undefined symbol at load time, no diagnostic).
#if defined(__HIP_DEVICE_COMPILE__)
#define MVS_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#define MVS_KEEP_VGPR(x) asm volatile("" ::"v"(x))
#else
#define MVS_OPAQUE_VGPR(x) ((void)(x))
#define MVS_KEEP_VGPR(x) ((void)(x))
#endif

struct MfmaArgs {
	const float *qf; // query fragments, layout [qblk32][ch][s4][lane][4]
	const float *qn; // query norms
	const float *yb; // database [n][dp]
	const float *yn; // database norms
	float *pd;       // partial distances [nsplit][nq][k]
	int32_t *pi;     // partial row ids
	unsigned *gslot; // [nq][slot_stride] shared threshold slots (keys); see "threshold sharing" below
	int slot_stride; // slots per query = k rounded up to a multiple of 16
	int nclass;      // row classes (row id mod nclass) the slots stand for; 0 = k
	int sched;       // prefilter kernel: 0 = none, 1 = s_setprio by wave slot parity, 2 = half-tile start stagger by slot parity, 3 = both
	long long n;
	long long split_rows;
	int nq, k, nqb, nsplit, dp, nch, xcd_map;
	// SEL instances only (inner product + IDSelector): FAISS's per-pair fvec_inner_product IS the k-ordered chain the
	// MFMA computes, so filtered IP search stays on the fused kernel and masks the rejected rows in the epilogue
	SelectorDev sel;
	const long long *idmap;
	// ITEMS instances (IVF list scan as a segmented variant of this kernel): one workgroup per work item =
	// (row segment of one inverted list, <= 128 of the queries that probe it)
	const int4 *items;       // {row_begin (multiple of 64), row_end, qoff, nq_item}
	const float *item_qn;    // ITEMS, L2 on residual rows: squared norm of (query - list centroid) per item slot [item][128]
	const int *nitems_dev;   // device-side item count; the grid is an upper bound
	const int *qidx;         // query number of slot qoff + s
	const long long *rowids; // stored id of every row (selector); row position is what the partial lists carry
};

// order-preserving float <-> uint key (atomicMin/Max on floats of either sign)
__device__ __forceinline__ unsigned f2key(float f) {
	const unsigned b = __float_as_uint(f);
	return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
	return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
// smallest float strictly greater than g >= 0 / largest float strictly smaller than g
__device__ __forceinline__ float next_up_nonneg(float g) {
	return __uint_as_float(__float_as_uint(g) + 1u);
}
__device__ __forceinline__ float next_down(float g) {
	if (g == 0.f)
		return __uint_as_float(0x80000001u);
	const unsigned b = __float_as_uint(g);
	return __uint_as_float(g > 0.f ? b - 1u : b + 1u);
}

__device__ __forceinline__ bool mfma_sel_member(const SelectorDev &s, long long id) {
	if (s.kind == MVS_SEL_BITMAP) {
		const unsigned long long u = (unsigned long long)id;
		if ((u >> 3) >= (unsigned long long)s.nbytes)
			return false;
		return (s.bitmap[u >> 3] >> (u & 7)) & 1;
	}
	if (s.kind == MVS_SEL_BATCH) {
		long long lo = 0, hi = s.nids;
		while (lo < hi) {
			const long long mid = (lo + hi) >> 1;
			if (s.sorted_ids[mid] < id)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo < s.nids && s.sorted_ids[lo] == id;
	}
	return true;
}

// membership order (DESIGN.md "ties"): L2 keeps the k smallest (dist, id); IP keeps the k largest score,
// equal scores prefer the smaller id.
template <bool IS_L2>
__device__ __forceinline__ bool cand_better(float v, int id, float tv, int tid) {
	if (IS_L2)
		return v < tv || (v == tv && id < tid);
	return v > tv || (v == tv && id < tid);
}

// replace the current worst entry of one query's list and find the new worst (rare path)
// (always inlined: a real call would force every value that lives across it -- query fragments, accumulators --
// into the callee-saved half of the VGPR file, i.e. spills)
template <bool IS_L2>
__device__ __forceinline__ bool entry_worse(float x, int xi, float y, int yi) {
	return IS_L2 ? (x > y || (x == y && xi > yi)) : (x < y || (x == y && xi > yi));
}
struct Thr {
	float v;
	int id;
	int pos;
};

// GLOBAL-memory k-lists (LSPACE != 1: k > 12 in the f32 kernel, e.g. the IVF coarse search with k = nprobe) stay heaps.
// The k-list of a query is a binary heap on (value, id) with the WORST entry at the root (slot 0 = the threshold):
// an insertion replaces the root and sifts down, <= log2(k) levels of two child reads, instead of rescanning all k
// slots for the new worst (k = 32: ~4x fewer dependent LDS / L2 round trips per insertion).  The all-neutral initial
// list is a valid heap; the merge kernels sort the partial lists, so the slot order never reaches the caller.
template <bool IS_L2>
__device__ __forceinline__ Thr list_insert(float *ld, int *li, int k, int /*pos*/, float v, int id) {
	int i = 0;
	for (;;) {
		const int l = 2 * i + 1, r = l + 1;
		if (l >= k)
			break;
		float cv = ld[l];
		int cid = li[l], c = l;
		if (r < k) {
			const float rv = ld[r];
			const int rid = li[r];
			if (entry_worse<IS_L2>(rv, rid, cv, cid)) {
				cv = rv;
				cid = rid;
				c = r;
			}
		}
		if (!entry_worse<IS_L2>(cv, cid, v, id))
			break; // the new entry is at least as bad as both children: it stays here
		ld[i] = cv;
		li[i] = cid;
		i = c;
	}
	ld[i] = v;
	li[i] = id;
	Thr t;
	t.v = ld[0];
	t.id = li[0];
	t.pos = 0;
	return t;
}


// "smaller is better" order-preserving key of a distance (L2) / score (IP)
template <bool IS_L2>
__device__ __forceinline__ unsigned bkey(float v) {
	return IS_L2 ? f2key(v) : ~f2key(v);
}
template <bool IS_L2>
__device__ __forceinline__ float bkey2f(unsigned k) {
	return key2f(IS_L2 ? k : ~k);
}

// Cross-workgroup threshold sharing.  Rows are split into k classes (row id mod k); slot j of a query holds the
// best value (as a "smaller is better" key) any workgroup has found among rows of class j.  If every slot is set
// there are k DISTINCT rows with value <= max_j slot_j, so the final k-th best value is <= that maximum: rows
// beyond it can never be in the result (ties are kept).  Updates are ONE fire-and-forget atomicMin -- no round trip
// on the inserting wave; readers fetch the slots of their query at the start of a tile (two lanes share a query and
// each loads half of a 16-slot WINDOW as 8-byte words) and reduce them in the epilogue.  For k > 16 the window
// rotates: the maximum over all ceil(k/16) windows, read over that many consecutive tiles, becomes the bound for the
// following tiles (slots only ever decrease, so a maximum assembled from older reads is still conservative).
// Relaxed agent-scope atomics only: a stale (looser) bound is always safe.  Slots [k, stride) are padding and hold
// key 0 (never the maximum).
constexpr int SLOT_WORDS = 4; // 8-byte words per lane: a lane pair covers one 16-slot window per tile

struct SlotRegs {
	unsigned long long w[SLOT_WORDS];
};
__device__ __forceinline__ void slots_prefetch(SlotRegs &sr, const unsigned *gslot_q, int window, int h) {
	const unsigned long long *src = (const unsigned long long *)(gslot_q + window * 16) + h * SLOT_WORDS;
#pragma unroll
	for (int j = 0; j < SLOT_WORDS; ++j)
		sr.w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// rotating-window state: `run` accumulates the current sweep, `bound` is the last completed sweep's maximum
struct SlotBound {
	unsigned run = 0u, bound = 0xFFFFFFFFu;
};
__device__ __forceinline__ unsigned slots_update(SlotBound &sb, unsigned part, int window, int nwin) {
	sb.run = sb.run > part ? sb.run : part;
	if (window == nwin - 1) { // wave-uniform
		sb.bound = sb.run;
		sb.run = 0u;
	}
	return sb.bound;
}
// max over this lane's words, then with the partner lane (l ^ 32) that holds the other half of the query's slots
__device__ __forceinline__ unsigned slots_reduce(const SlotRegs &sr) {
	unsigned m = 0u;
#pragma unroll
	for (int j = 0; j < SLOT_WORDS; ++j) {
		const unsigned lo = (unsigned)sr.w[j], hi = (unsigned)(sr.w[j] >> 32);
		m = m > lo ? m : lo;
		m = m > hi ? m : hi;
	}
	const unsigned o = (unsigned)__shfl_xor((int)m, 32);
	return m > o ? m : o;
}

// ---- fused epilogue of one row tile: distances + threshold test (lane-local per query) + rare insertion ----
// acc[t][r] holds ip(query = lane&31, row = t*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)).
// gkey = this query's shared bound (from the slot words the caller prefetched at the START of the tile, so that the
// L2/MALL round trip hides under the tile's MFMAs); all-ones until the first sweep over the slots is complete.
// TIE (instances of the tie pass, IS_L2 = true): the accumulator holds an inner-product score and xnq the query's
// boundary score T; a row counts as "distance 0" when score >= T and as +inf otherwise, so the smaller-is-better
// (value, id) lists end up holding the k SMALLEST ROW IDS with score >= T (see FlatIndex::search_flat, "ties").
#ifdef MVS_COUNT_EVENTS
static __device__ unsigned long long g_dbg_counters[4]; // rare-path events, candidates, insertions (diagnostic builds)
#endif
template <int LSPACE>
struct ListPtr { // generic (flat) pointers
	typedef float *F;
	typedef int *I;
	static __device__ __forceinline__ F f(int lo, int hi) {
		return (F)(uintptr_t)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
	}
	static __device__ __forceinline__ I i(int lo, int hi) {
		return (I)(uintptr_t)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
	}
};
template <>
struct ListPtr<1> { // LDS: the low word of a generic LDS pointer is the LDS offset
	typedef __attribute__((address_space(3))) float *F;
	typedef __attribute__((address_space(3))) int *I;
#if defined(__HIP_DEVICE_COMPILE__)
	static __device__ __forceinline__ F f(int lo, int) {
		return (F)(unsigned)lo;
	}
	static __device__ __forceinline__ I i(int lo, int) {
		return (I)(unsigned)lo;
	}
#else // host pass: pointers are 64 bits wide there; never executed
	static __device__ __forceinline__ F f(int, int) {
		return nullptr;
	}
	static __device__ __forceinline__ I i(int, int) {
		return nullptr;
	}
#endif
};
template <>
struct ListPtr<2> { // global memory
	typedef __attribute__((address_space(1))) float *F;
	typedef __attribute__((address_space(1))) int *I;
	static __device__ __forceinline__ F f(int lo, int hi) {
		return (F)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
	}
	static __device__ __forceinline__ I i(int lo, int hi) {
		return (I)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
	}
};

// The rare path of tile_epilogue as a function of its own.  OUTLINE instances (the bf16x3 kernel, 256 VGPRs in its hot
// loop) call it for real: the values that live across the call sit in callee-saved registers and the callee's own needs
// cost scratch traffic only when it runs, instead of SGPR / VGPR spills inside the tile loop.
template <int NT, bool IS_L2, int LSPACE>
__device__ __forceinline__ float rare_insert_body(const f32x16 (&acc)[NT], long long row0, int nvalid, float thr, bool qvalid,
                                                 float gval, bool any, unsigned *gslot_q, float *ldq, int *liq, int k, int h,
                                                 int nclass) {
		// ---- rare path: exact (value, id) insertion, WAVE-COOPERATIVE -------------------------------------------
		// Every lane marks its passing rows in a bit mask (against the shared bound at entry, T0; ties kept).  Then, round by
		// round, every lane pulls its next marked value out of the accumulators (a select tree: no dynamic register
		// indexing, no scratch) and the wave serialises over the lanes that have one: the 64 lanes insert it TOGETHER into
		// that lane's query list, which is kept SORTED (best first): lane i reads entry i, a ballot of "entry not worse than
		// the candidate" gives the insert position, the entries behind it move up by one, the last one falls out.  One LDS
		// (or L2, for global lists) round trip per candidate instead of a dependent chain of sift steps by a single lane, and
		// the two lanes that share a query (l, l + 32) need no turn taking.  The retained SET is what it always was -- the k
		// best under the (value, id) order -- and the merge kernels sort the partial lists anyway.
#ifdef MVS_COUNT_EVENTS
		const unsigned long long t_dbg0 = __builtin_amdgcn_s_memtime();
		if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)
			atomicAdd(&g_dbg_counters[0], 1ull);
#endif
		const float T0 = gval;
		unsigned long long m = 0ull;
		if (any) {
#pragma unroll
			for (int t = 0; t < NT; ++t)
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					float v = acc[t][r];
					if (IS_L2)
						v = v < 0.f ? 0.f : v; // FAISS: if (dis < 0) dis = 0
					const int rl = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
					if (rl < nvalid && (IS_L2 ? v <= T0 : v >= T0))
						m |= 1ull << (t * 16 + r);
				}
		}
		const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		const unsigned long long ldq_u = (unsigned long long)(uintptr_t)ldq, liq_u = (unsigned long long)(uintptr_t)liq,
		                         gsl_u = (unsigned long long)(uintptr_t)gslot_q;
		for (;;) {
			const bool has = m != 0ull;
			unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
			if (bal == 0ull)
				break;
			const int j = has ? __builtin_ctzll(m) : 0;
			m &= m - 1ull; // (0 stays 0)
			// value j of this lane: static select tree over the NT * 16 accumulator registers
			float sel[NT * 16];
#pragma unroll
			for (int i = 0; i < NT * 16; ++i)
				sel[i] = acc[i >> 4][i & 15];
#pragma unroll
			for (int w = NT * 16, bit = 0; w > 1; w >>= 1, ++bit)
#pragma unroll
				for (int i = 0; i < w / 2; ++i)
					sel[i] = ((j >> bit) & 1) ? sel[2 * i + 1] : sel[2 * i];
			float v = sel[0];
			if (IS_L2)
				v = v < 0.f ? 0.f : v;
			const int id = (int)(row0 + (j >> 4) * 32 + (j & 3) + 8 * ((j >> 2) & 3) + 4 * h);
			if (__popcll(bal) >= 6) {
				// MANY lanes hold a candidate (cold start of a workgroup, small databases, large k): serialising them through
				// the cooperative insert would cost one LDS round trip per candidate.  Every lane inserts into its own
				// query's list instead -- an insertion sort from the back, lanes in parallel; the two lanes that share a
				// query (l, l + 32) take turns.
				for (int hh = 0; hh < 2; ++hh) {
					const bool mine = has && h == hh;
					if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
						if (mine) {
							typedef typename ListPtr<LSPACE>::F FP;
							typedef typename ListPtr<LSPACE>::I IP_;
							FP ld_ = ListPtr<LSPACE>::f((int)ldq_u, (int)(ldq_u >> 32));
							IP_ li_ = ListPtr<LSPACE>::i((int)liq_u, (int)(liq_u >> 32));
							const unsigned ad0 = (unsigned)ldq_u, ai0 = (unsigned)liq_u; // LDS byte offsets (LSPACE == 1)
							auto rd = [&](int i, float &ev, int &ei) {
								if (LSPACE == 1) // by hand: see the cooperative insert below
									asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
									             : "=&v"(ev), "=&v"(ei)
									             : "v"(ad0 + 4u * (unsigned)i), "v"(ai0 + 4u * (unsigned)i)
									             : "memory");
								else {
									ev = ld_[i];
									ei = li_[i];
								}
							};
							auto wr = [&](int i, float ev, int ei) {
								if (LSPACE == 1)
									asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3" ::"v"(ad0 + 4u * (unsigned)i),
									             "v"(ai0 + 4u * (unsigned)i), "v"(ev), "v"(ei)
									             : "memory");
								else {
									ld_[i] = ev;
									li_[i] = ei;
								}
							};
							int i = k - 1;
							float ev;
							int ei;
							rd(i, ev, ei);
							if (entry_worse<IS_L2>(ev, ei, v, id)) {
								while (i > 0) {
									rd(i - 1, ev, ei);
									if (!entry_worse<IS_L2>(ev, ei, v, id))
										break;
									wr(i, ev, ei);
									--i;
								}
								wr(i, v, id);
								typedef __attribute__((address_space(1))) unsigned *GU;
								__hip_atomic_fetch_min((GU)gsl_u + (unsigned)id % (unsigned)nclass, bkey<IS_L2>(v), __ATOMIC_RELAXED,
								                       __HIP_MEMORY_SCOPE_AGENT);
							}
						}
						if (LSPACE == 1)
							asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
						else
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
						__builtin_amdgcn_wave_barrier();
					}
				}
				bal = 0ull;
			}
			while (bal != 0ull) {
				const int L = __builtin_ctzll(bal); // wave-uniform
				bal &= bal - 1ull;
				const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), L));
				const int cid = __builtin_amdgcn_readlane(id, L);
				typedef typename ListPtr<LSPACE>::F FP;
				typedef typename ListPtr<LSPACE>::I IP_;
				FP ldL = ListPtr<LSPACE>::f(__builtin_amdgcn_readlane((int)ldq_u, L), __builtin_amdgcn_readlane((int)(ldq_u >> 32), L));
				IP_ liL = ListPtr<LSPACE>::i(__builtin_amdgcn_readlane((int)liq_u, L), __builtin_amdgcn_readlane((int)(liq_u >> 32), L));
#ifdef MVS_COUNT_EVENTS
				if (lane == 0)
					atomicAdd(&g_dbg_counters[1], 1ull);
#endif
				bool inserted = false;
				if (LSPACE == 1 && k <= 64) {
					// LDS lists, by hand: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it compiles while an
					// LDS-DMA is in flight (and a wavefront fence waits for the fire-and-forget atomic below), i.e. every
					// candidate would drain the NEXT tile's staging and an L2 round trip.  Only lgkmcnt matters here.
					const unsigned ad = (unsigned)__builtin_amdgcn_readlane((int)ldq_u, L) + 4u * (unsigned)lane;
					const unsigned ai = (unsigned)__builtin_amdgcn_readlane((int)liq_u, L) + 4u * (unsigned)lane;
					float ev = 0.f;
					int ei = 0;
					if (lane < k)
						asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
						             : "=&v"(ev), "=&v"(ei)
						             : "v"(ad), "v"(ai)
						             : "memory");
					const bool notworse = lane < k && !entry_worse<IS_L2>(ev, ei, cv, cid);
					const int p = __popcll(__builtin_amdgcn_ballot_w64(notworse));
					if (p < k) {
						if (lane >= p && lane < k - 1)
							asm volatile("ds_write_b32 %0, %2 offset:4\n\tds_write_b32 %1, %3 offset:4" ::"v"(ad), "v"(ai), "v"(ev), "v"(ei)
							             : "memory");
						if (lane == p)
							asm volatile("ds_write_b32 %0, %2\n\tds_write_b32 %1, %3" ::"v"(ad), "v"(ai), "v"(cv), "v"(cid) : "memory");
						inserted = true;
					}
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				} else if (k <= 64) {
					float ev = 0.f;
					int ei = 0;
					if (lane < k) {
						ev = ldL[lane];
						ei = liL[lane];
					}
					const bool notworse = lane < k && !entry_worse<IS_L2>(ev, ei, cv, cid);
					const int p = __popcll(__builtin_amdgcn_ballot_w64(notworse)); // entries that stay in front
					if (p < k) {
						if (lane >= p && lane < k - 1) {
							ldL[lane + 1] = ev;
							liL[lane + 1] = ei;
						}
						if (lane == 0) {
							ldL[p] = cv;
							liL[p] = cid;
						}
						inserted = true;
					}
				} else {
					int p = 0;
					for (int b = 0; b < k; b += 64) {
						const int i = b + lane;
						bool notworse = false;
						if (i < k)
							notworse = !entry_worse<IS_L2>(ldL[i], liL[i], cv, cid);
						p += __popcll(__builtin_amdgcn_ballot_w64(notworse));
					}
					if (p < k) {
						for (int b = ((k - 1) >> 6) << 6; b >= 0; b -= 64) { // highest block first: its target slots were read
							const int i = b + lane;
							const bool mv = i >= p && i < k - 1;
							float ev = 0.f;
							int ei = 0;
							if (mv) {
								ev = ldL[i];
								ei = liL[i];
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
							if (mv) {
								ldL[i + 1] = ev;
								liL[i + 1] = ei;
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
						}
						if (lane == 0) {
							ldL[p] = cv;
							liL[p] = cid;
						}
						inserted = true;
					}
				}
				if (inserted && lane == 0) { // publish: best value of this row's class (fire and forget)
					// a GLOBAL-address-space atomic (vmcnt only).  Through a generic pointer it is a FLAT atomic, which also
					// counts on lgkmcnt: the next LDS wait of this wave would then sit out the atomic's whole L2 round trip,
					// once per inserted candidate.
					typedef __attribute__((address_space(1))) unsigned *GU;
					GU gsL = (GU)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(gsl_u >> 32), L) << 32) |
					              (unsigned)__builtin_amdgcn_readlane((int)gsl_u, L));
					__hip_atomic_fetch_min(gsL + (unsigned)cid % (unsigned)nclass, bkey<IS_L2>(cv), __ATOMIC_RELAXED,
					                       __HIP_MEMORY_SCOPE_AGENT);
				}
				if (!(LSPACE == 1 && k <= 64))
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
#ifdef MVS_COUNT_EVENTS
		if (lane == 0)
			atomicAdd(&g_dbg_counters[3], __builtin_amdgcn_s_memtime() - t_dbg0);
#endif
		// the list is sorted: its last entry is the query's current k-th best
		if (LSPACE == 1) {
			const unsigned at = (unsigned)ldq_u + 4u * (unsigned)(k - 1);
			float nt = thr;
			asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(nt) : "v"(at) : "memory");
			if (qvalid)
				thr = nt;
		} else if (qvalid) {
			thr = ldq[k - 1];
		}
	return thr;
}
template <bool IS_L2, int LSPACE>
__device__ __attribute__((noinline)) float rare_insert_outlined(f32x16 acc0, long long row0, int nvalid, float thr, bool qvalid,
                                                                float gval, bool any, unsigned *gslot_q, float *ldq, int *liq,
                                                                int k, int h, int nclass) {
	f32x16 acc[1] = {acc0};
	return rare_insert_body<1, IS_L2, LSPACE>(acc, row0, nvalid, thr, qvalid, gval, any, gslot_q, ldq, liq, k, h, nclass);
}

// YPRE: the row norms were already fetched into registers by the caller (ypre[t * 4 + g] = norms of rows
// t*32 + 8g + 4h .. +3); nb is then unused.
// LSPACE: where the k-lists live -- 1 = LDS, 2 = global memory (0 = unknown: flat accesses).  The cooperative insert
// rebuilds another lane's list pointer from readlane'd words, which hides the address space from the compiler.
template <int NT, bool IS_L2, bool SKIP_SLOW = false, bool SEL = false, bool TIE = false, bool YPRE = false, int LSPACE = 0,
          bool OUTLINE = false>
__device__ __forceinline__ void tile_epilogue(f32x16 (&acc)[NT], const float *nb, long long row0, int nvalid, float xnq,
                                              float &thr, bool qvalid, unsigned gkey, unsigned *gslot_q,
                                              float *ldq, int *liq, int k, float *lthr_q, int *lthrid_q, int *lpos_q,
                                              int h, const unsigned long long *rowmask = nullptr,
                                              const float4 *ypre = nullptr, int nclass_in = 0) {
	const int nclass = nclass_in > 0 ? nclass_in : k;
	float gval = IS_L2 ? FLT_MAX : -FLT_MAX;
	float teff = thr;
	if (qvalid) {
		const unsigned neutral_key = bkey<IS_L2>(IS_L2 ? FLT_MAX : -FLT_MAX);
		gval = bkey2f<IS_L2>(gkey < neutral_key ? gkey : neutral_key);
		// ties with the shared threshold are kept (<= / >=): one strict compare against its float neighbour
		teff = IS_L2 ? fminf(thr, next_up_nonneg(gval)) : fmaxf(thr, next_down(gval));
	}
	// best value of each group of 4 rows (kept for the rare path), then of all NT*16 rows.  NaN never wins.
	float gm[NT][4];
	float best = IS_L2 ? INFINITY : -INFINITY;
#pragma unroll
	for (int t = 0; t < NT; ++t) {
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
			if (IS_L2 && !TIE)
				y4 = YPRE ? ypre[t * 4 + g] : *(const float4 *)(nb + t * 32 + 8 * g + 4 * h);
			const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
			float v[4];
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				v[e] = acc[t][4 * g + e];
				if (IS_L2) {
					if (TIE)
						v[e] = v[e] >= xnq ? 0.f : INFINITY; // NaN scores never count (FAISS: strict compares)
					else
						v[e] = fmaf(-2.0f, v[e], xnq + yv[e]); // (xn + yn) - 2 ip, two roundings as the oracle
					acc[t][4 * g + e] = v[e];
				}
				if (SEL) { // rows the IDSelector rejects can never be a result
					const int rl = t * 32 + e + 8 * g + 4 * h;
					if (!((rowmask[rl >> 6] >> (rl & 63)) & 1ull)) {
						v[e] = IS_L2 ? INFINITY : -INFINITY;
						acc[t][4 * g + e] = v[e];
					}
				}
			}
			if (IS_L2)
				gm[t][g] = __builtin_fminf(__builtin_fminf(v[0], v[1]), __builtin_fminf(v[2], v[3]));
			else
				gm[t][g] = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), __builtin_fmaxf(v[2], v[3]));
		}
		if (IS_L2)
			best = __builtin_fminf(
			    best, __builtin_fminf(__builtin_fminf(gm[t][0], gm[t][1]), __builtin_fminf(gm[t][2], gm[t][3])));
		else
			best = __builtin_fmaxf(
			    best, __builtin_fmaxf(__builtin_fmaxf(gm[t][0], gm[t][1]), __builtin_fmaxf(gm[t][2], gm[t][3])));
	}
	const bool any = IS_L2 ? best < teff : best > teff;
	if (SKIP_SLOW) {
		MVS_KEEP_VGPR(any);
		return;
	}
	if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
		if constexpr (LSPACE == 1) {
			if constexpr (OUTLINE && NT == 1)
				thr = rare_insert_outlined<IS_L2, LSPACE>(acc[0], row0, nvalid, thr, qvalid, gval, any, gslot_q, ldq, liq, k, h, nclass);
			else
				thr = rare_insert_body<NT, IS_L2, LSPACE>(acc, row0, nvalid, thr, qvalid, gval, any, gslot_q, ldq, liq, k, h, nclass);
		} else {
			// Lists in GLOBAL memory (or of unknown placement): every list access is an L2 round trip, so an insertion must
			// be O(log k) accesses and the lanes must work in parallel -- binary heaps, one lane per query, the two lanes
			// that share a query taking turns.  (LDS lists, above, are sorted arrays filled cooperatively.)
			// ---- rare path: exact (value, id) insertion into this wave's per-query lists ----------
			// The lane's passing rows (against the thresholds at entry, T0) are numbered in scan order; each pass of the
			// loop below compacts the next two of them into registers (all indices static: no scratch, no calls inside
			// the unrolled scan) and inserts them.  Usually one lane has one candidate and one pass suffices.
			const float T0 = gval;
			for (int hh = 0; hh < 2; ++hh) { // lanes l and l+32 share a query: take turns
				const bool mine = any && h == hh;
				if (__builtin_amdgcn_ballot_w64(mine) == 0ull)
					continue;
				if (mine) {
					Thr cur;
					cur.v = *lthr_q;
					cur.id = *lthrid_q;
					cur.pos = *lpos_q;
					float T = T0;
					int done = 0, npass;
					do {
						float cv0 = 0.f, cv1 = 0.f;
						int cid0 = 0, cid1 = 0;
						npass = 0;
	#pragma unroll
						for (int t = 0; t < NT; ++t) {
	#pragma unroll
							for (int g = 0; g < 4; ++g) {
								if (IS_L2 ? gm[t][g] <= T0 : gm[t][g] >= T0) {
	#pragma unroll
									for (int e = 0; e < 4; ++e) {
										float v = acc[t][4 * g + e];
										if (IS_L2)
											v = v < 0.f ? 0.f : v; // FAISS: if (dis < 0) dis = 0
										const int rl = t * 32 + e + 8 * g + 4 * h;
										if (rl < nvalid && (IS_L2 ? v <= T0 : v >= T0)) {
											if (npass == done) {
												cv0 = v;
												cid0 = (int)(row0 + rl);
											} else if (npass == done + 1) {
												cv1 = v;
												cid1 = (int)(row0 + rl);
											}
											++npass;
										}
									}
								}
							}
						}
	#pragma unroll 1
						for (int i = 0; i < 2; ++i) { // a real loop: ONE inlined copy of the insertion code
							if (done + i < npass) {
								const float v = i ? cv1 : cv0;
								const int id = i ? cid1 : cid0;
								if ((IS_L2 ? v <= T : v >= T) && cand_better<IS_L2>(v, id, cur.v, cur.id)) {
									cur = list_insert<IS_L2>(ldq, liq, k, cur.pos, v, id);
									// publish: best value of this row's class (fire and forget)
									__hip_atomic_fetch_min((__attribute__((address_space(1))) unsigned *)(uintptr_t)gslot_q + (unsigned)id % (unsigned)nclass,
									                       bkey<IS_L2>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
								}
							}
						}
						done += 2;
					} while (done < npass);
					*lthr_q = cur.v;
					*lthrid_q = cur.id;
					*lpos_q = cur.pos;
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			thr = *lthr_q;
		}
	}
}


} // namespace mvs
