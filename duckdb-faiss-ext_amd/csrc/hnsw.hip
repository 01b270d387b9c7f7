// csrc/hnsw.hip -- faiss::IndexHNSWFlat on device ("HNSW<M>", "HNSW<M>,Flat").
//
// Replaces what the reference reaches through index_factory (:154), hnsw.efConstruction (:136-139), Index::add
// (:510/:512 -- <= 2048 rows per call, so the graph grows incrementally) and Index::search with
// SearchParametersHNSW{efSearch, sel} (:631, :691-702) of /root/reference/src/faiss_extension.cpp.  FAISS behaviour
// restated [UPSTREAM: faiss/impl/HNSW.cpp, faiss/IndexHNSW.cpp; line-by-line restatement in oracle/orc_hnsw.c]:
//   add    : levels from RandomGenerator(12345) with assign_probas (mult = 1/ln M), FAISS's flat offsets/neighbors
//            layout (2M slots at level 0, M above, -1 = empty); points inserted bucket by bucket from the highest
//            level down, each bucket shuffled with RandomGenerator(789); per point: greedy descent, then per level
//            search_neighbors_to_add (efConstruction) -> shrink_neighbor_list -> links both ways (add_link).
//   search : greedy descent on levels >= 1, level-0 best-first with the bounded candidate set (ef = max(efSearch,k)),
//            stop when efSearch stored distances are below the popped one; selector filters RESULTS only; inner
//            product is walked as the negated value and restored on output.
//
// MI355X design: ONE 64-lane wavefront owns one query (search) / one inserted point (build).
//   * a distance = the whole wave reads one 4*dp-byte row as coalesced float4 (dwordx4) loads, G rows in flight,
//     4 fma chains per lane, then a fixed DPP reduction tree -- the canonical HNSW arithmetic of oracle/orc_hnsw.c,
//     so distances, graphs and results are bit-identical to the oracle;
//   * the bounded sets (MinimaxHeap / result heap / construction result set) are SORTED 64-bit key arrays
//     ((ordered distance bits << 32) | id) in LDS: insert = ballot-popcount for the position + one lane-parallel
//     shift; pop-min, count_below and "current worst" are O(1) reads.  Heap shape is not part of FAISS's observable
//     behaviour, the (distance, id) order is;
//   * visited table = one byte per vertex per wave in HBM with a rolling stamp (VisitedTable);
//   * build concurrency mirrors FAISS's OpenMP loop: many waves insert points of one level bucket at the same time
//     under per-vertex spin locks (at most one lock held, exactly like add_with_locks); neighbour lists are read and
//     written with agent-scope relaxed atomics so the eight XCD L2s cannot serve stale lists.  One wave
//     (option hnsw_build_waves = 1) is the deterministic order and reproduces the oracle's graph bit for bit.
// The graph walk is HBM-latency/bandwidth work: ~ n_visited * (4d + 4) bytes per query; no MFMA.
#include "index.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

namespace mvs {

namespace {

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------ device helpers

__device__ __forceinline__ unsigned h_f2key(float f) {
	const unsigned b = __float_as_uint(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float h_key2f(unsigned k) {
	const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
	return __uint_as_float(b);
}
// low word: ((id + 1) << 1) | flag ; 0 = tombstone (MinimaxHeap's id = -1 keeps its distance)
__device__ __forceinline__ u64 mk_key(float dis, int id, unsigned flag = 0) {
	return ((u64)h_f2key(dis) << 32) | (u64)((((unsigned)(id + 1)) << 1) | flag);
}
__device__ __forceinline__ float key_dis(u64 k) {
	return h_key2f((unsigned)(k >> 32));
}
__device__ __forceinline__ int key_id(u64 k) {
	return (int)(((unsigned)k) >> 1) - 1;
}
__device__ __forceinline__ int rfl(int v) {
	return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ float rflf(float v) {
	return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ u64 rfl64(u64 v) {
	const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
	const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
	return ((u64)hi << 32) | lo;
}
// LDS traffic of one wave is in order in hardware; this only stops the compiler from reordering it
__device__ __forceinline__ void wave_fence() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// adjacent-pair binary tree over the 64 lanes (oracle/orc_hnsw.c dc_q); every lane ends with the total
__device__ __forceinline__ float wave_sum(float v) {
	v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
	v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
	v += dpp_mov<0x141>(v); // row_half_mirror: quad 0 <-> quad 1
	v += dpp_mov<0x140>(v); // row_mirror: lanes 0-7 <-> 8-15
	// xor 16 / xor 32 on the vector ALU (gfx950's v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips
	// through the LDS: with both operands = v the swap returns {v with its odd rows replaced by the even ones, v with its even rows
	// replaced by the odd ones}; their sum is v[lane] + v[lane ^ 16] in every lane (same two addends as before: same bits)
	{
		const unsigned u = __float_as_uint(v);
		const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
		v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
	}
	{
		const unsigned u = __float_as_uint(v);
		const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
		v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
	}
	return v;
}

template <int NI>
struct QV {
	float4 v[NI];
};

template <int NI>
__device__ __forceinline__ void load_row(QV<NI> &q, const float *row, int dp4, int lane) {
#pragma unroll
	for (int i = 0; i < NI; i++) {
		const int idx = lane + 64 * i;
		q.v[i] = idx < dp4 ? reinterpret_cast<const float4 *>(row)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
	}
}

template <int NI, bool IS_L2>
__device__ __forceinline__ float lane_partial(const QV<NI> &q, const QV<NI> &y) {
	float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
	for (int i = 0; i < NI; i++) {
		if (IS_L2) {
			const float t0 = q.v[i].x - y.v[i].x, t1 = q.v[i].y - y.v[i].y;
			const float t2 = q.v[i].z - y.v[i].z, t3 = q.v[i].w - y.v[i].w;
			a0 = fmaf(t0, t0, a0);
			a1 = fmaf(t1, t1, a1);
			a2 = fmaf(t2, t2, a2);
			a3 = fmaf(t3, t3, a3);
		} else {
			a0 = fmaf(q.v[i].x, y.v[i].x, a0);
			a1 = fmaf(q.v[i].y, y.v[i].y, a1);
			a2 = fmaf(q.v[i].z, y.v[i].z, a2);
			a3 = fmaf(q.v[i].w, y.v[i].w, a3);
		}
	}
	return (a0 + a1) + (a2 + a3);
}

template <int NI, bool IS_L2>
__device__ __forceinline__ float wave_dist1(const QV<NI> &q, const float *vecs, int dp4, int id, int lane) {
	QV<NI> y;
	load_row(y, vecs + (size_t)id * dp4 * 4, dp4, lane);
	const float t = wave_sum(lane_partial<NI, IS_L2>(q, y));
	return rflf(IS_L2 ? t : -t);
}

// distances from q to the rows nid[l] of the lanes l in `mask`; lane l receives its own distance.
// The G row loads of a group are issued back to back with NO control flow between them (a short last group re-reads
// its first row instead of branching): with "if (slot used) load" the compiler fuses load and reduction per slot and
// the rows arrive one HBM latency after the other.
template <int NI, bool IS_L2, int G>
__device__ __forceinline__ float eval_lanes(const QV<NI> &q, const float *vecs, int dp4, int nid, u64 mask, int lane) {
	float mydd = 0.f;
	while (mask) {
		int ls[G], ids[G];
		ls[0] = (int)__builtin_ctzll(mask);
		mask &= mask - 1;
#pragma unroll
		for (int g = 1; g < G; g++) {
			const bool more = mask != 0;
			const int l = more ? (int)__builtin_ctzll(mask) : ls[0];
			ls[g] = more ? l : -1;
			mask = more ? (mask & (mask - 1)) : mask;
			ids[g] = __builtin_amdgcn_readlane(nid, l);
		}
		ids[0] = __builtin_amdgcn_readlane(nid, ls[0]);
		QV<NI> y[G];
#pragma unroll
		for (int g = 0; g < G; g++)
			load_row(y[g], vecs + (size_t)ids[g] * dp4 * 4, dp4, lane);
#pragma unroll
		for (int g = 0; g < G; g++) {
			const float t = wave_sum(lane_partial<NI, IS_L2>(q, y[g]));
			if (lane == ls[g])
				mydd = IS_L2 ? t : -t;
		}
	}
	return mydd;
}

// ---- bf16 FIRST LOOK (search only; option hnsw_bf16) ---------------------------------------------------------------------------
// A second copy of the rows in bf16 (half the bytes).  Most neighbours of a hop are evaluated only to be thrown away: once the
// candidate set is full a neighbour matters only if its distance is below the worst candidate (or below the worst result, under
// a selector).  The walk therefore first computes every fresh neighbour's distance against the bf16 row and fetches the f32 row
// only when the EXACT distance could still be below that threshold thr:
//   L2   a = sum (q_i - y~_i)^2, y~ = bf16(y):  |a - D| <= 2^-7 ||y|| sqrt(D) + 2^-16 ||y||^2  (bf16 keeps 8 significant bits:
//        |y~_i - y_i| <= 2^-8 |y_i|; Cauchy-Schwarz), so D < thr implies a < thr + c1 sqrt(thr) + c2: a neighbour with
//        a >= thr' + c1 sqrt(thr') + c2, thr' = thr (1 + 1e-5), c1 = 2^-7 Ymax, c2 = 2^-16 Ymax^2 (Ymax = the largest row norm of
//        the index, margins for the f32 arithmetic of both sums included) is skipped -- it could not have changed any set;
//   IP   a = -<q, y~>:  |a - D| <= 2^-8 ||q|| ||y||: skipped when a >= thr + |thr| 1e-5 + (2^-8 + 2^-13) ||q|| Ymax.
// Every neighbour that passes gets the usual exact evaluation, so the walk's decisions -- and its results, bit for bit -- are those
// of the f32-only walk and of the oracle; "distance evaluations" keeps counting every fresh neighbour (what FAISS evaluates).
template <int NI>
struct QVH {
	uint2 v[NI]; // 4 bf16 per lane and step
};
template <int NI>
__device__ __forceinline__ void load_row_bf(QVH<NI> &y, const unsigned short *row, int dp4, int lane) {
#pragma unroll
	for (int i = 0; i < NI; i++) {
		const int idx = lane + 64 * i;
		y.v[i] = idx < dp4 ? reinterpret_cast<const uint2 *>(row)[idx] : make_uint2(0u, 0u);
	}
}
template <int NI, bool IS_L2>
__device__ __forceinline__ float lane_partial_bf(const QV<NI> &q, const QVH<NI> &y) {
	float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
	for (int i = 0; i < NI; i++) {
		const float y0 = __uint_as_float(y.v[i].x << 16), y1 = __uint_as_float(y.v[i].x & 0xffff0000u);
		const float y2 = __uint_as_float(y.v[i].y << 16), y3 = __uint_as_float(y.v[i].y & 0xffff0000u);
		if (IS_L2) {
			const float t0 = q.v[i].x - y0, t1 = q.v[i].y - y1, t2 = q.v[i].z - y2, t3 = q.v[i].w - y3;
			a0 = fmaf(t0, t0, a0);
			a1 = fmaf(t1, t1, a1);
			a2 = fmaf(t2, t2, a2);
			a3 = fmaf(t3, t3, a3);
		} else {
			a0 = fmaf(q.v[i].x, y0, a0);
			a1 = fmaf(q.v[i].y, y1, a1);
			a2 = fmaf(q.v[i].z, y2, a2);
			a3 = fmaf(q.v[i].w, y3, a3);
		}
	}
	return (a0 + a1) + (a2 + a3);
}
// approximate distances from q to the bf16 rows nid[l] of the lanes l in `mask` (structure of eval_lanes)
template <int NI, bool IS_L2, int G>
__device__ __forceinline__ float eval_lanes_bf(const QV<NI> &q, const unsigned short *vbf, int dp4, int nid, u64 mask, int lane) {
	float mydd = 0.f;
	while (mask) {
		int ls[G], ids[G];
		ls[0] = (int)__builtin_ctzll(mask);
		mask &= mask - 1;
#pragma unroll
		for (int g = 1; g < G; g++) {
			const bool more = mask != 0;
			const int l = more ? (int)__builtin_ctzll(mask) : ls[0];
			ls[g] = more ? l : -1;
			mask = more ? (mask & (mask - 1)) : mask;
			ids[g] = __builtin_amdgcn_readlane(nid, l);
		}
		ids[0] = __builtin_amdgcn_readlane(nid, ls[0]);
		QVH<NI> y[G];
#pragma unroll
		for (int g = 0; g < G; g++)
			load_row_bf(y[g], vbf + (size_t)ids[g] * dp4 * 4, dp4, lane);
#pragma unroll
		for (int g = 0; g < G; g++) {
			const float t = wave_sum(lane_partial_bf<NI, IS_L2>(q, y[g]));
			if (lane == ls[g])
				mydd = IS_L2 ? t : -t;
		}
	}
	return mydd;
}
// rows [r0, r1) of the f32 store -> bf16 copy; the largest squared row norm (float bits, atomicMax)
__global__ void hnsw_rows_to_bf16_kernel(const float *__restrict__ vecs, long long r0, long long r1, int dp,
                                         unsigned short *__restrict__ out, unsigned *__restrict__ max_bits) {
	const long long r = r0 + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (r >= r1)
		return;
	float n2 = 0.f;
	for (int c = lane; c < dp; c += 64) {
		const float v = vecs[(size_t)r * dp + c];
		const __bf16 h = (__bf16)v;
		out[(size_t)r * dp + c] = *reinterpret_cast<const unsigned short *>(&h);
		n2 = fmaf(v, v, n2);
	}
	for (int o = 32; o >= 1; o >>= 1)
		n2 += __shfl_xor(n2, o);
	if (lane == 0 && __float_as_uint(n2) > __hip_atomic_load(max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) // (agent scope: a plain load stays as first cached)
		atomicMax(max_bits, __float_as_uint(n2)); // (>= 0: the bit pattern orders like the value; NaN sorts above everything)
}

// keys[0..n) ascending; inserts nk, keeps at most cap entries (the largest falls off).  Returns the new count.
__device__ __forceinline__ int sorted_insert(u64 *keys, int n, int cap, u64 nk, int lane) {
	if (cap <= 256) {
		// every lane reads its (<= 4) elements once, the position is the popcount of "smaller than nk", and the
		// elements that are not smaller move one slot to the right -- all reads before all writes, one fence
		u64 v[4];
		int pos = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			v[b] = ~0ull;
			if (b * 64 < n) { // wave-uniform
				const int i = b * 64 + lane;
				if (i < n)
					v[b] = keys[i];
				pos += (int)__popcll(__builtin_amdgcn_ballot_w64(i < n && v[b] < nk));
			}
		}
		if (pos >= cap)
			return n;
		wave_fence();
#pragma unroll
		for (int b = 0; b < 4; ++b)
			if (b * 64 < n) {
				const int i = b * 64 + lane;
				if (i < n && !(v[b] < nk) && i + 1 < cap)
					keys[i + 1] = v[b];
			}
		if (lane == 0)
			keys[pos] = nk;
		wave_fence();
		return n < cap ? n + 1 : cap;
	}
	int pos = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		const bool lt = i < n && keys[i] < nk;
		pos += (int)__popcll(__builtin_amdgcn_ballot_w64(lt));
	}
	if (pos >= cap)
		return n;
	const int n2 = n < cap ? n + 1 : cap;
	for (int base = (n2 - 1) & ~63; base + 63 > pos; base -= 64) { // highest block first
		const int i = base + lane;
		const bool mv = i > pos && i < n2;
		u64 t = 0;
		if (mv)
			t = keys[i - 1];
		wave_fence();
		if (mv)
			keys[i] = t;
		wave_fence();
	}
	if (lane == 0)
		keys[pos] = nk;
	wave_fence();
	return n2;
}

struct GraphDev {
	const float *vecs; // [n][4*dp4]
	int dp4;
	const long long *offsets; // [n+1]
	int32_t *neighbors;
	int M;
};
__device__ __forceinline__ int nb_at(const GraphDev &g, int level) {
	return level == 0 ? 2 * g.M : g.M;
}
__device__ __forceinline__ int cum_at(const GraphDev &g, int level) {
	return level == 0 ? 0 : (level + 1) * g.M;
}
template <bool ATOMIC>
__device__ __forceinline__ int ld_nb(const int32_t *p) {
	if (ATOMIC)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}
__device__ __forceinline__ void st_nb(int32_t *p, int v) {
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lanes before the first empty slot (FAISS: "if (v < 0) break")
__device__ __forceinline__ u64 valid_prefix(u64 vmask) {
	const u64 inv = ~vmask;
	return inv ? (vmask & ((1ull << __builtin_ctzll(inv)) - 1ull)) : vmask;
}

// HNSW.cpp greedy_update_nearest
template <int NI, bool IS_L2, int G, bool ATOMIC>
__device__ __forceinline__ void greedy_update_nearest(const GraphDev &g, const QV<NI> &q, int level, int &nearest,
                                                      float &d_nearest, int lane, unsigned &ndis) {
	const int L = nb_at(g, level);
	for (;;) {
		const int prev = nearest;
		const long long base = g.offsets[nearest] + cum_at(g, level);
		for (int c0 = 0; c0 < L; c0 += 64) {
			const int j = c0 + lane;
			const int nid = j < L ? ld_nb<ATOMIC>(g.neighbors + base + j) : -1;
			const u64 vmask = __builtin_amdgcn_ballot_w64(nid >= 0);
			u64 m = valid_prefix(vmask);
			ndis += (unsigned)__popcll(m);
			const float mydd = eval_lanes<NI, IS_L2, G>(q, g.vecs, g.dp4, nid, m, lane);
			while (m) {
				const int l = (int)__builtin_ctzll(m);
				m &= m - 1;
				const float dd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydd), l));
				if (dd < d_nearest) {
					d_nearest = dd;
					nearest = __builtin_amdgcn_readlane(nid, l);
				}
			}
			if (~vmask)
				break;
		}
		if (nearest == prev)
			return;
	}
}

__device__ __forceinline__ bool sel_member_dev(const SelectorDev &s, long long id) {
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

__device__ __forceinline__ void clear_table(uint8_t *vis, long long nbytes16, int lane) {
	uint4 z = make_uint4(0, 0, 0, 0);
	for (long long i = lane; i < nbytes16; i += 64)
		reinterpret_cast<uint4 *>(vis)[i] = z;
}

// Visited set of one query.  Random single-byte probes of a per-wave table in HBM cap the whole walk at the DRAM
// small-access rate (tools/micro/gather_bw.hip: random 512-B reads reach 2 TB/s, 3 KB rows 6.6), so the set lives in
// LDS: exact open-addressing hash of vertex ids (0 = empty slot, key = id + 1, linear probing, ds_cmpst).  When it
// fills up (large ef) the wave migrates the ids into its HBM byte table and continues there -- still exact.
struct VisitedSet {
	unsigned *tab; // LDS, hsize entries
	unsigned hmask;
	int hshift;
	uint8_t *vis; // HBM fallback: one byte per vertex, rolling stamp
	uint8_t stamp;
	bool spilled;
	int count, limit;
};
__device__ __forceinline__ void visited_reset(VisitedSet &v, int lane) {
	const uint4 z = make_uint4(0, 0, 0, 0);
	for (unsigned i = lane; i < (v.hmask + 1) / 4; i += 64)
		reinterpret_cast<uint4 *>(v.tab)[i] = z;
	wave_fence();
	v.spilled = false;
	v.count = 0;
}
// 0 = already present, 1 = inserted now, 2 = probe chain too long (table crowded)
__device__ __forceinline__ int visited_probe(const VisitedSet &v, int id) {
	const unsigned key = (unsigned)id + 1u;
	unsigned h = (key * 2654435761u) >> v.hshift;
	for (int p = 0; p < 32; p++) {
		const unsigned old = atomicCAS(&v.tab[h], 0u, key);
		if (old == 0u)
			return 1;
		if (old == key)
			return 0;
		h = (h + 1) & v.hmask;
	}
	return 2;
}
__device__ __forceinline__ void visited_spill(VisitedSet &v, int lane) {
	wave_fence();
	for (unsigned i = lane; i <= v.hmask; i += 64) {
		const unsigned key = v.tab[i];
		if (key)
			v.vis[key - 1u] = v.stamp;
	}
	__builtin_amdgcn_s_waitcnt(0);
	wave_fence();
	v.spilled = true;
}
// marks the vertices nid of the lanes in `valid`; returns "was not visited before"
__device__ __forceinline__ bool visited_test_and_set(VisitedSet &v, int nid, bool valid, int lane) {
	bool fresh = false;
	if (!v.spilled) {
		int r = 0;
		if (valid)
			r = visited_probe(v, nid);
		fresh = r == 1;
		v.count += (int)__popcll(__builtin_amdgcn_ballot_w64(fresh));
		const u64 over = __builtin_amdgcn_ballot_w64(r == 2);
		if (over || v.count > v.limit) {
			visited_spill(v, lane);
			if (r == 2) {
				fresh = v.vis[nid] != v.stamp;
				if (fresh)
					v.vis[nid] = v.stamp;
			}
		}
		return fresh;
	}
	if (valid) {
		fresh = v.vis[nid] != v.stamp;
		if (fresh)
			v.vis[nid] = v.stamp;
	}
	return fresh;
}

// ---- sorted lists in REGISTERS (search, ef <= 128 and k <= 64) ---------------------------------------------------------------------
// The candidate and result lists of a walk are sorted arrays of (distance, id) keys.  In LDS every insertion is two dependent round
// trips (read all, fence, write the shifted tail, fence) and every hop reads the arrays again for pop_min / count_below; the walk's
// chain of dependent LDS latencies is what bounds it (replacing the two ds_bpermute of the wave reduction alone took C5 from 14.3 to
// 12.6 ms).  Entry i of a list lives in lane i & 63 of register v[i >> 6]: position = popcount of a ballot, the shift by one lane is
// one DPP move (wave_shr:1) per dword, a uniform element is two v_readlane -- no LDS, no fence.
__device__ __forceinline__ unsigned wave_shr1(unsigned v, unsigned lane0) { // lane l <- lane l - 1; lane 0 <- lane0 (uniform)
	return (unsigned)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 lane_get64(u64 v, int l) { // l uniform
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
	return ((u64)hi << 32) | lo;
}
template <int NB>
__device__ __forceinline__ u64 rl_at(const u64 (&v)[NB], int i) { // i uniform, 0 <= i < 64 NB
	u64 r = lane_get64(v[0], i & 63);
#pragma unroll
	for (int b = 1; b < NB; ++b) {
		const u64 t = lane_get64(v[b], i & 63);
		r = (i >> 6) == b ? t : r;
	}
	return r;
}
// sorted_insert on a register list: n entries, capacity cap <= 64 NB; returns the new n
template <int NB>
__device__ __forceinline__ int rl_insert(u64 (&v)[NB], int n, int cap, u64 nk, int lane) {
	int pos = 0;
#pragma unroll
	for (int b = 0; b < NB; ++b) {
		const int i = b * 64 + lane;
		pos += (int)__popcll(__builtin_amdgcn_ballot_w64(i < n && v[b] < nk));
	}
	if (pos >= cap)
		return n;
#pragma unroll
	for (int b = NB - 1; b >= 0; --b) { // highest block first: block b's lane 0 takes block b - 1's lane 63 as it was
		const int i = b * 64 + lane;
		const u64 carry = b > 0 ? lane_get64(v[b - 1], 63) : 0ull;
		const unsigned lo = wave_shr1((unsigned)v[b], (unsigned)carry);
		const unsigned hi = wave_shr1((unsigned)(v[b] >> 32), (unsigned)(carry >> 32));
		const u64 prev = ((u64)hi << 32) | lo;
		v[b] = i > pos ? prev : (i == pos ? nk : v[b]);
	}
	return n < cap ? n + 1 : cap;
}

// ------------------------------------------------------------------------------------------------ search kernel

// -DMVS_HNSW_PROFILE: per-phase shader-clock totals in stats[2..8] (tools/hbench.py prints them); off in the product
#ifdef MVS_HNSW_PROFILE
#define PROF_NOW() __builtin_readcyclecounter()
#define PROF_ADD(acc, a, b) acc += (b) - (a)
#else
#define PROF_NOW() 0ull
#define PROF_ADD(acc, a, b) ((void)(a), (void)(b))
#endif

struct SearchArgs {
	GraphDev g;
	int entry_point, max_level;
	const float *xq; // [nq][4*dp4]
	long long nq;
	int k, ef, efSearch;
	int hsize; // visited hash slots in LDS (power of two)
	const int32_t *nb0; // dense copy of the level-0 lists, [n][2M]: no offsets[] round trip in front of every hop
	SelectorDev sel;
	const long long *idmap;
	long long label_offset;
	uint8_t *visited; // [grid][vstride]
	long long vstride;
	unsigned *vstamp; // [grid] rolling stamp, persists across searches
	int *counter;     // dynamic query queue (walk lengths vary a lot between queries)
	float *D;
	long long *I;
	unsigned long long *stats; // [0] distance evaluations, [1] expanded vertices, [9] f32 rows fetched (bf16 first look)
	const unsigned short *vbf; // bf16 copy of the rows (BF instances)
	const unsigned *ymax_bits; // largest squared row norm (float bits)
};

// RL: the candidate and result lists live in registers instead of LDS -- 1: ef <= 128, k <= 64; 2: ef <= 256, k <= 256;
// round 5 (the harness asks for up to ~2 000 rows per query, go/main_test.go:26-32): 3: ef, k <= 512 -- HNSW128 d = 1536 N = 1 M, 43
// queries, k = 500: 3.0 ms per batch against 5.2 on LDS lists (one query: 1.39 vs 1.75).  A level of 32 blocks (ef, k <= 2 048) was
// measured SLOWER than the LDS lists (k = 1000: 9.7 vs 7.1 ms, one query 4.1 vs 1.9: every list operation walks all 32 blocks) and
// is not built: profiles/r5_harness_shapes.txt
template <int NI, bool IS_L2, int G, bool BF = false, int RL = 0>
__global__ __launch_bounds__(64) void hnsw_search_kernel(const SearchArgs a) {
	constexpr int NBC = RL == 3 ? 8 : (RL == 2 ? 4 : 2), NBR = RL == 3 ? 8 : (RL == 2 ? 4 : 1); // 64-entry blocks of the two lists
	extern __shared__ u64 smem[];
	u64 *ckeys = smem + a.hsize / 2; // MinimaxHeap candidates(ef); the visited hash sits in front (16-byte aligned)
	u64 *rkeys = ckeys + a.ef;       // result heap (k)
	u64 creg[NBC], rreg[NBR]; // (RL)
#pragma unroll
	for (int b = 0; b < NBC; ++b)
		creg[b] = ~0ull;
#pragma unroll
	for (int b = 0; b < NBR; ++b)
		rreg[b] = ~0ull;
	const int lane = threadIdx.x;
	const GraphDev &g = a.g;
	uint8_t *vis = a.visited + (size_t)blockIdx.x * a.vstride;
	unsigned stamp = a.vstamp[blockIdx.x], ndis = 0, nexp = 0;
	VisitedSet vs;
	vs.tab = reinterpret_cast<unsigned *>(smem);
	vs.hmask = a.hsize ? (unsigned)a.hsize - 1u : 0u;
	vs.hshift = a.hsize ? 32 - (31 - __builtin_clz((unsigned)a.hsize)) : 0;
	vs.vis = vis;
	vs.limit = a.hsize - a.hsize / 4 - 64; // spill before the table is 3/4 full
	unsigned long long p_desc = 0, p_pop = 0, p_nbr = 0, p_vis = 0, p_eval = 0, p_ins = 0, p_total = 0;
	(void)p_desc, (void)p_pop, (void)p_nbr, (void)p_vis, (void)p_eval, (void)p_ins, (void)p_total;
	const int ef = a.ef, k = a.k;
	unsigned nf32 = 0;
	float bf_c1 = 0.f, bf_c2 = 0.f, ymax = 0.f;
	if (BF) {
		const float y2 = __uint_as_float(*a.ymax_bits) * 1.0001f;
		ymax = sqrtf(y2) * 1.0001f;
		bf_c1 = 0.0078125f * ymax * 1.001f;
		bf_c2 = 1.52587890625e-05f * y2 * 1.01f;
	}
	for (;;) {
		int qn = 0;
		if (lane == 0)
			qn = atomicAdd(a.counter, 1);
		const long long qi = rfl(qn);
		if (qi >= a.nq)
			break;
		if (++stamp == 256) {
			clear_table(vis, a.vstride / 16, lane);
			stamp = 1;
		}
		vs.stamp = (uint8_t)stamp;
		if (a.hsize)
			visited_reset(vs, lane);
		else
			vs.spilled = true; // option hnsw_visited_lds = 0: HBM byte table only
		const unsigned long long tq0 = PROF_NOW();
		QV<NI> q;
		load_row(q, a.xq + (size_t)qi * g.dp4 * 4, g.dp4, lane);
		float bf_eip = 0.f; // inner product: (2^-8 + 2^-13) ||q|| Ymax
		if (BF && !IS_L2)
			bf_eip = (0.00390625f + 0.0001220703125f) * sqrtf(rflf(wave_sum(lane_partial<NI, false>(q, q))) * 1.0001f) * ymax;
		int nearest = a.entry_point;
		float d_nearest = wave_dist1<NI, IS_L2>(q, g.vecs, g.dp4, nearest, lane);
		ndis++;
		for (int level = a.max_level; level >= 1; level--)
			greedy_update_nearest<NI, IS_L2, G, false>(g, q, level, nearest, d_nearest, lane, ndis);
		// ---- search_from_candidates, level 0
		int nc = 0, nr = 0, nvalid = 0;
		// (the two lists: LDS arrays, or registers when RL)
		auto c_insert = [&](u64 key) { nc = RL ? rl_insert<NBC>(creg, nc, ef, key, lane) : sorted_insert(ckeys, nc, ef, key, lane); };
		auto r_insert = [&](u64 key) { nr = RL ? rl_insert<NBR>(rreg, nr, k, key, lane) : sorted_insert(rkeys, nr, k, key, lane); };
		auto c_at = [&](int i) -> u64 { return RL ? rl_at<NBC>(creg, i) : rfl64(ckeys[i]); };
		auto r_at = [&](int i) -> u64 { return RL ? rl_at<NBR>(rreg, i) : rfl64(rkeys[i]); };
		c_insert(mk_key(d_nearest, nearest));
		nvalid = 1;
		float rthr = FLT_MAX; // heap threshold: the k-heap starts full of (FLT_MAX, -1)
		{
			const bool pass = a.sel.kind == MVS_SEL_NONE || sel_member_dev(a.sel, a.idmap ? a.idmap[nearest] : nearest);
			if (pass && d_nearest < rthr) {
				r_insert(mk_key(d_nearest, nearest));
				rthr = nr < k ? FLT_MAX : key_dis(r_at(k - 1));
			}
		}
		(void)visited_test_and_set(vs, nearest, lane == 0, lane);
		PROF_ADD(p_desc, tq0, PROF_NOW());
		while (nvalid > 0) {
			const unsigned long long t0 = PROF_NOW();
			// pop_min: first live entry of the sorted array
			int pos = -1;
			if (RL) {
#pragma unroll
				for (int b = 0; b < NBC; ++b) {
					const int i = b * 64 + lane;
					const u64 m = __builtin_amdgcn_ballot_w64(i < nc && (unsigned)creg[b] != 0u);
					if (m && pos < 0)
						pos = b * 64 + (int)__builtin_ctzll(m);
				}
			} else {
				for (int base = 0; base < nc && pos < 0; base += 64) {
					const int i = base + lane;
					const bool live = i < nc && (unsigned)ckeys[i] != 0u;
					const u64 m = __builtin_amdgcn_ballot_w64(live);
					if (m)
						pos = base + (int)__builtin_ctzll(m);
				}
			}
			const u64 ck = c_at(pos);
			const int v0 = key_id(ck);
			if (RL) {
#pragma unroll
				for (int b = 0; b < NBC; ++b)
					if (b * 64 + lane == pos)
						creg[b] &= 0xffffffff00000000ull; // tombstone keeps its distance
			} else {
				if (lane == 0)
					ckeys[pos] = ck & 0xffffffff00000000ull; // tombstone keeps its distance
				wave_fence();
			}
			nvalid--;
			// count_below(d0) >= efSearch -> stop (check_relative_distance)
			int nbelow = 0;
			if (RL) {
#pragma unroll
				for (int b = 0; b < NBC; ++b) {
					const int i = b * 64 + lane;
					nbelow += (int)__popcll(__builtin_amdgcn_ballot_w64(i < nc && (unsigned)(creg[b] >> 32) < (unsigned)(ck >> 32)));
				}
			} else {
				for (int base = 0; base < nc; base += 64) {
					const int i = base + lane;
					const bool lt = i < nc && (unsigned)(ckeys[i] >> 32) < (unsigned)(ck >> 32);
					nbelow += (int)__popcll(__builtin_amdgcn_ballot_w64(lt));
				}
			}
			if (nbelow >= a.efSearch)
				break;
			nexp++;
			const unsigned long long t1 = PROF_NOW();
			PROF_ADD(p_pop, t0, t1);
			const int L = 2 * g.M;
			const int32_t *list0 = a.nb0 + (size_t)v0 * L;
			for (int c0 = 0; c0 < L; c0 += 64) {
				const unsigned long long t2 = PROF_NOW();
				const int j = c0 + lane;
				const int nid = j < L ? list0[j] : -1;
				const u64 vmask = __builtin_amdgcn_ballot_w64(nid >= 0);
				const u64 pm = valid_prefix(vmask);
				const bool valid = (pm >> lane) & 1ull;
				const unsigned long long t3 = PROF_NOW();
				PROF_ADD(p_nbr, t2, t3);
				const bool fresh = visited_test_and_set(vs, nid, valid, lane);
				const u64 fmask = __builtin_amdgcn_ballot_w64(fresh);
				const unsigned long long t4 = PROF_NOW();
				PROF_ADD(p_vis, t3, t4);
				ndis += (unsigned)__popcll(fmask);
				const float cmax = nc < ef ? FLT_MAX : key_dis(c_at(ef - 1));
				u64 need = fmask;
				if (BF && fmask != 0ull) {
					// a neighbour matters only below the worst candidate or (selectors) the worst result: nothing to skip until both
					// sets are full
					const float thr = fmaxf(cmax, rthr);
					if (thr < FLT_MAX) {
						// (2 G bf16 rows in flight -- same registers as G f32 rows -- measured slower on one box, two builds: 16.3 / 17.0
						// vs 14.2 / 14.4 ms at C5; the loads of G rows already cover the hop's latency, more only delay its first use)
						const float ap = eval_lanes_bf<NI, IS_L2, G>(q, a.vbf, g.dp4, nid, fmask, lane);
						float lim;
						if (IS_L2) {
							const float t = thr * 1.00001f;
							lim = (t + bf_c1 * sqrtf(t) + bf_c2) * 1.00001f;
						} else {
							lim = thr + fabsf(thr) * 1e-5f + bf_eip;
						}
						need = fmask & ~__builtin_amdgcn_ballot_w64(fresh && ap >= lim); // (NaN: not skipped)
					}
					nf32 += (unsigned)__popcll(need);
				}
				const float exd = eval_lanes<NI, IS_L2, G>(q, g.vecs, g.dp4, nid, need, lane);
				const float mydd = (!BF || ((need >> lane) & 1ull)) ? exd : FLT_MAX; // skipped: beyond every threshold
				const unsigned long long t5 = PROF_NOW();
				PROF_ADD(p_eval, t4, t5);
				bool pass = fresh;
				if (fresh && a.sel.kind != MVS_SEL_NONE)
					pass = sel_member_dev(a.sel, a.idmap ? a.idmap[nid] : nid);
				// thresholds only ever tighten, so lanes rejected now stay rejected during the sequential pass
				const bool maybe = fresh && (nc < ef || mydd < cmax || (pass && mydd < rthr));
				u64 mm = __builtin_amdgcn_ballot_w64(maybe);
				const u64 passmask = __builtin_amdgcn_ballot_w64(pass);
				while (mm) {
					const int l = (int)__builtin_ctzll(mm);
					mm &= mm - 1;
					const float dd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydd), l));
					const int id = __builtin_amdgcn_readlane(nid, l);
					const u64 key = mk_key(dd, id);
					if (((passmask >> l) & 1ull) && dd < rthr) { // res.add_result
						r_insert(key);
						rthr = nr < k ? FLT_MAX : key_dis(r_at(k - 1));
					}
					// candidates.push(v1, d)
					if (nc == ef) {
						const u64 last = c_at(ef - 1);
						if (dd >= key_dis(last))
							continue;
						if ((unsigned)last != 0u)
							nvalid--;
					}
					c_insert(key);
					nvalid++;
				}
				PROF_ADD(p_ins, t5, PROF_NOW());
				if (~vmask)
					break;
			}
		}
		PROF_ADD(p_total, tq0, PROF_NOW());
		// ---- heap_reorder + (IP) sign restore + label translation
		for (int j = lane; j < k; j += 64) {
			float dv = IS_L2 ? FLT_MAX : -FLT_MAX;
			long long lab = -1;
			if (j < nr) {
				u64 rk;
				if (RL) { // entry j lives in lane j & 63 of block j >> 6
					rk = rreg[0];
#pragma unroll
					for (int b = 1; b < NBR; ++b)
						rk = (j >> 6) == b ? rreg[b] : rk;
				} else {
					rk = rkeys[j];
				}
				const float dd = key_dis(rk);
				dv = IS_L2 ? dd : -dd;
				const int id = key_id(rk);
				lab = a.idmap ? a.idmap[id] : (long long)id + a.label_offset;
			}
			a.D[qi * k + j] = dv;
			a.I[qi * k + j] = lab;
		}
		wave_fence();
	}
	if (lane == 0) {
		a.vstamp[blockIdx.x] = stamp;
		if (a.stats) {
			atomicAdd(&a.stats[0], (unsigned long long)ndis);
			atomicAdd(&a.stats[1], (unsigned long long)nexp);
			if (BF)
				atomicAdd(&a.stats[9], (unsigned long long)nf32);
#ifdef MVS_HNSW_PROFILE
			atomicAdd(&a.stats[2], p_desc);
			atomicAdd(&a.stats[3], p_pop);
			atomicAdd(&a.stats[4], p_nbr);
			atomicAdd(&a.stats[5], p_vis);
			atomicAdd(&a.stats[6], p_eval);
			atomicAdd(&a.stats[7], p_ins);
			atomicAdd(&a.stats[8], p_total);
#endif
		}
	}
}

// nb0[v][0..2M) = neighbors[offsets[v] .. +2M)
__global__ void hnsw_level0_copy_kernel(const long long *offsets, const int32_t *neighbors, int32_t *nb0, long long v0,
                                        long long n, int L) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (n - v0) * L)
		return;
	const long long v = v0 + i / L;
	const int j = (int)(i % L);
	nb0[v * L + j] = neighbors[offsets[v] + j];
}

__global__ void hnsw_fill_empty_kernel(float *D, long long *I, long long n, float neutral) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		D[i] = neutral;
		I[i] = -1;
	}
}

// ------------------------------------------------------------------------------------------------ build kernel

struct BuildArgs {
	GraphDev g;
	int entry_point, max_level;
	int efC;
	const int32_t *order; // insertion order of this add() call
	int i0, i1;           // slice of `order` handled by this launch (all of level pt_level)
	int pt_level;
	int *counter;
	int *locks;
	int use_locks;
	int count_spins;
	unsigned char *clean; // [vertex] bit l: the level-l list is FULL and is the kept sequence of its last shrink (add_link's short cut)
	uint8_t *visited; // [grid][vstride]
	long long vstride;
	unsigned *vstamp; // [grid] rolling stamp, persists across launches
	unsigned long long *stats;
};

__device__ __forceinline__ void wave_lock(const BuildArgs &a, int v, int lane) {
	if (!a.use_locks)
		return;
	for (;;) {
		int got = 0;
		if (lane == 0) {
			int expected = 0;
			got = __hip_atomic_compare_exchange_strong(&a.locks[v], &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
			                                           __HIP_MEMORY_SCOPE_AGENT)
			          ? 1
			          : 0;
		}
		if (rfl(got))
			break;
		if (a.count_spins && lane == 0) // (MVS_INGEST_PROFILE: how long the build waits for vertex locks)
			atomicAdd(&a.stats[1], 1ull);
		__builtin_amdgcn_s_sleep(8);
	}
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ __forceinline__ void wave_unlock(const BuildArgs &a, int v, int lane) {
	if (!a.use_locks)
		return;
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
	__builtin_amdgcn_s_waitcnt(0); // every list store of this wave has been acknowledged
	if (lane == 0)
		__hip_atomic_store(&a.locks[v], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// HNSW.cpp shrink_neighbor_list on candidates sorted closest-first; result closest-first in out_id/out_d
template <int NI, bool IS_L2, int G>
__device__ __forceinline__ int shrink_select(const GraphDev &g, const u64 *keys, int n, int max_size, int *out_id,
                                             float *out_d, int lane, unsigned &ndis) {
	if (n < max_size) { // "if (input.size() < max_size) return;"
		for (int j = lane; j < n; j += 64) {
			const u64 kk = keys[j];
			out_id[j] = key_id(kk);
			out_d[j] = key_dis(kk);
		}
		wave_fence();
		return n;
	}
	int nout = 0;
	for (int c = 0; c < n; c++) {
		const u64 ck = rfl64(keys[c]);
		const int v1 = key_id(ck);
		const float d1 = key_dis(ck);
		bool good = true;
		if (nout > 0) {
			QV<NI> qc;
			load_row(qc, g.vecs + (size_t)v1 * g.dp4 * 4, g.dp4, lane);
			for (int j0 = 0; j0 < nout && good; j0 += 64) {
				const int j = j0 + lane;
				const int kid = j < nout ? out_id[j] : -1;
				u64 m = __builtin_amdgcn_ballot_w64(j < nout);
				// groups of G kept rows, stop at the first group holding a closer kept neighbour
				while (m && good) {
					u64 sub = 0;
#pragma unroll
					for (int t = 0; t < G; t++)
						if (m) {
							sub |= m & (~m + 1);
							m &= m - 1;
						}
					ndis += (unsigned)__popcll(sub);
					const float dd = eval_lanes<NI, IS_L2, G>(qc, g.vecs, g.dp4, kid, sub, lane);
					const bool closer = ((sub >> lane) & 1ull) && dd < d1;
					if (__builtin_amdgcn_ballot_w64(closer))
						good = false;
				}
			}
		}
		if (good) {
			if (lane == 0) {
				out_id[nout] = v1;
				out_d[nout] = d1;
			}
			wave_fence();
			nout++;
			if (nout >= max_size)
				break;
		}
	}
	return nout;
}

// HNSW.cpp add_link(src -> dest); srcq = row of src
// Round 6, the short cut for a list that is FULL and still the kept sequence k_1 .. k_L of its last shrink (flag `clean`, kept under the
// vertex's lock): shrinking {k_1 .. k_L, dest} needs no pairwise pass.  Sorted by distance to src, every k_i in front of dest meets the very
// kept set it met last time (k_1 .. k_{i-1}) and is kept again; dest is kept iff none of those is closer to it than src is; a k_i behind
// dest was good against k_1 .. k_{i-1} and stays good against any subset, so only dest can prune it (dist(k_i, dest) < dist(k_i, src)).
// <= 2L + 1 INDEPENDENT evaluations instead of ~L^2 / 4 dependent ones, the same list bit for bit (tests/test_hnsw_gpu.py: the single-wave
// build still reproduces the oracle's graph).  Hub vertices of high-dimensional rows sit on full lists for most of a build and every
// insertion near them queues on their lock: the short cut is what shortens that queue (profiles/r6_hnsw_build.txt).
template <int NI, bool IS_L2, int G>
__device__ __forceinline__ void add_link(const GraphDev &g, const QV<NI> &srcq, int src, int dest, int level, u64 *tkeys,
                                         int *out_id, float *out_d, int lane, unsigned &ndis, unsigned char *clean, unsigned long long *nshort) {
	const int L = nb_at(g, level);
	int32_t *list = g.neighbors + g.offsets[src] + cum_at(g, level);
	if (rfl(ld_nb<true>(list + L - 1)) == -1) { // room left: first free slot
		int cnt = 0;
		for (int c0 = 0; c0 < L; c0 += 64) {
			const int j = c0 + lane;
			const int nid = j < L ? ld_nb<true>(list + j) : -1;
			const u64 vmask = __builtin_amdgcn_ballot_w64(nid >= 0);
			// FAISS scans from the end for the last used slot; lists are packed, so that is the valid prefix
			const u64 inv = ~vmask;
			if (inv) {
				cnt = c0 + (int)__builtin_ctzll(inv);
				break;
			}
			cnt = c0 + 64;
		}
		if (lane == 0)
			st_nb(list + cnt, dest);
		return;
	}
	// full: shrink (current neighbours + dest) back to L
	int nt = 0;
	for (int c0 = 0; c0 < L; c0 += 64) {
		const int j = c0 + lane;
		const int nid = j < L ? ld_nb<true>(list + j) : -1;
		u64 m = __builtin_amdgcn_ballot_w64(j < L && nid >= 0);
		ndis += (unsigned)__popcll(m);
		const float mydd = eval_lanes<NI, IS_L2, G>(srcq, g.vecs, g.dp4, nid, m, lane);
		while (m) {
			const int l = (int)__builtin_ctzll(m);
			m &= m - 1;
			const float dd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydd), l));
			nt = sorted_insert(tkeys, nt, L + 1, mk_key(dd, __builtin_amdgcn_readlane(nid, l)), lane);
		}
	}
	{
		const float dd = wave_dist1<NI, IS_L2>(srcq, g.vecs, g.dp4, dest, lane);
		ndis++;
		nt = sorted_insert(tkeys, nt, L + 1, mk_key(dd, dest), lane);
	}
	const unsigned lbit = level < 8 ? 1u << level : 0u;
	unsigned cflag = 0u;
	if (clean && lbit) {
		if (lane == 0)
			cflag = __hip_atomic_load(clean + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		cflag = (unsigned)rfl((int)cflag);
	}
	int nout;
	if (cflag & lbit) {
		// ---- the short cut: tkeys = k_1 .. k_L and dest, closest first
		if (nshort && lane == 0)
			atomicAdd(nshort, 1ull);
		int r = 0;
		for (int j0 = 0; j0 < nt; j0 += 64) {
			const int j = j0 + lane;
			const u64 m = __builtin_amdgcn_ballot_w64(j < nt && key_id(tkeys[j]) == dest);
			if (m)
				r = j0 + (int)__builtin_ctzll(m);
		}
		const float d_dest = key_dis(rfl64(tkeys[r]));
		QV<NI> qd;
		load_row(qd, g.vecs + (size_t)dest * g.dp4 * 4, g.dp4, lane);
		bool good = true;
		for (int j0 = 0; j0 < r && good; j0 += 64) { // is a kept row in front of dest closer to dest than src is?
			const int j = j0 + lane;
			const int kid = j < r ? key_id(tkeys[j]) : -1;
			u64 m = __builtin_amdgcn_ballot_w64(j < r);
			while (m && good) {
				u64 sub = 0;
#pragma unroll
				for (int t = 0; t < G; t++)
					if (m) {
						sub |= m & (~m + 1);
						m &= m - 1;
					}
				ndis += (unsigned)__popcll(sub);
				const float dd = eval_lanes<NI, IS_L2, G>(qd, g.vecs, g.dp4, kid, sub, lane);
				if (__builtin_amdgcn_ballot_w64(((sub >> lane) & 1ull) && dd < d_dest))
					good = false;
			}
		}
		if (!good)
			return; // dest is pruned: the list stays what it is (and clean)
		for (int j = lane; j <= r; j += 64)
			out_id[j] = key_id(tkeys[j]);
		nout = r + 1;
		for (int j0 = 0; j0 < nt; j0 += 64) { // the rows behind dest: pruned when closer to dest than to src
			const int j = j0 + lane;
			const bool in = j > r && j < nt;
			const u64 kk = in ? tkeys[j] : 0ull;
			const int kid = in ? key_id(kk) : -1;
			const u64 m = __builtin_amdgcn_ballot_w64(in);
			if (!m)
				continue;
			ndis += (unsigned)__popcll(m);
			const float dd = eval_lanes<NI, IS_L2, G>(qd, g.vecs, g.dp4, kid, m, lane);
			const bool keep = in && !(dd < key_dis(kk));
			const u64 km = __builtin_amdgcn_ballot_w64(keep);
			const int pos = nout + (int)__popcll(km & ((1ull << lane) - 1ull));
			if (keep && pos < L)
				out_id[pos] = kid;
			nout += (int)__popcll(km);
		}
		nout = nout < L ? nout : L; // "if (output.size() >= max_size) return"
		wave_fence();
	} else {
		nout = shrink_select<NI, IS_L2, G>(g, tkeys, nt, L, out_id, out_d, lane, ndis);
	}
	// "while (resultSet.size()) neighbors[i++] = resultSet.top().id" : farthest first, then -1
	for (int c0 = 0; c0 < L; c0 += 64) {
		const int j = c0 + lane;
		if (j < L)
			st_nb(list + j, j < nout ? out_id[nout - 1 - j] : -1);
	}
	if (clean && lbit && lane == 0) {
		const unsigned nf = nout == L ? (cflag | lbit) : (cflag & ~lbit);
		if (nf != cflag)
			__hip_atomic_store(clean + src, (unsigned char)nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// One WORKGROUP of W wavefronts per inserted point (round 6).  Wave 0 does what HNSW::add_with_locks does up to the forward links: greedy
// descent, search_neighbors_to_add, shrink, add_link(pt -> neighbour).  The BACK links -- add_link(neighbour -> pt) for each of the <= 2M
// selected neighbours, each under that neighbour's lock, each a shrink of a full list: 2M + 1 candidates, ~2 000 distance evaluations of
// dependent row round trips -- are ~95 % of an insertion's evaluations and touch different vertices: the W waves share them out (t = wave,
// wave + W, ...).  Same lists as the sequential loop (a back link reads and writes its own vertex's list only), a W-th of the latency:
// profiles/r6_hnsw_build.txt -- the build is bound by the LATENCY of one insertion (the glue adds 2048 rows at a time, level bucket by level
// bucket: four launches per chunk, each as long as its slowest insertion), not by bandwidth or issue slots.
template <int NI, bool IS_L2, int G, int W>
__global__ __launch_bounds__(64 * W) void hnsw_build_kernel(const BuildArgs a) {
	extern __shared__ u64 smem[];
	const GraphDev &g = a.g;
	const int L0 = 2 * g.M;
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	u64 *rkeys = smem;                      // [efC] construction result set, bit 0 = already expanded (wave 0)
	int *sel_id = (int *)(rkeys + a.efC);   // [2M] link targets of the current level
	int *ctl = sel_id + L0;                 // [2] the point's position in `order` | link targets of this level
	u64 *tkeys = (u64 *)(ctl + 2) + (size_t)wave * (2 * (L0 + 1)); // per wave: [2M+1] add_link scratch
	int *out_id = (int *)(tkeys + L0 + 1);  // [2M+1]
	float *out_d = (float *)(out_id + L0 + 1);
	uint8_t *vis = a.visited + (size_t)blockIdx.x * a.vstride;
	unsigned stamp = wave == 0 ? a.vstamp[blockIdx.x] : 0u;
	unsigned ndis = 0;
	const int efC = a.efC;
	for (;;) {
		if (wave == 0) {
			int i = 0;
			if (lane == 0)
				i = atomicAdd(a.counter, 1);
			if (lane == 0)
				ctl[0] = i;
		}
		__syncthreads();
		const int i = rfl(ctl[0]) + a.i0;
		if (i >= a.i1)
			break;
		const int pt = a.order[i];
		QV<NI> q;
		int nearest = a.entry_point;
		float d_nearest = 0.f;
		if (wave == 0) {
			load_row(q, g.vecs + (size_t)pt * g.dp4 * 4, g.dp4, lane);
			wave_lock(a, pt, lane);
			d_nearest = wave_dist1<NI, IS_L2>(q, g.vecs, g.dp4, nearest, lane);
			ndis++;
			for (int level = a.max_level; level > a.pt_level; level--)
				greedy_update_nearest<NI, IS_L2, G, true>(g, q, level, nearest, d_nearest, lane, ndis);
		}
		for (int level = a.max_level < a.pt_level ? a.max_level : a.pt_level; level >= 0; level--) {
			const int L = nb_at(g, level);
			if (wave == 0) {
				// ---- search_neighbors_to_add: candidates = the not yet expanded entries of the result set
				if (++stamp == 256) {
					clear_table(vis, a.vstride / 16, lane);
					stamp = 1;
				}
				int nr = sorted_insert(rkeys, 0, efC, mk_key(d_nearest, nearest), lane);
				if (lane == 0)
					vis[nearest] = (uint8_t)stamp;
				for (;;) {
					int pos = -1;
					for (int base = 0; base < nr && pos < 0; base += 64) {
						const int i2 = base + lane;
						const bool open = i2 < nr && ((unsigned)rkeys[i2] & 1u) == 0u;
						const u64 m = __builtin_amdgcn_ballot_w64(open);
						if (m)
							pos = base + (int)__builtin_ctzll(m);
					}
					if (pos < 0)
						break;
					const u64 ck = rfl64(rkeys[pos]);
					if (lane == 0)
						rkeys[pos] = ck | 1ull;
					wave_fence();
					const int cur = key_id(ck);
					const long long base0 = g.offsets[cur] + cum_at(g, level);
					for (int c0 = 0; c0 < L; c0 += 64) {
						const int j = c0 + lane;
						const int nid = j < L ? ld_nb<true>(g.neighbors + base0 + j) : -1;
						const u64 vmask = __builtin_amdgcn_ballot_w64(nid >= 0);
						const u64 pm = valid_prefix(vmask);
						bool fresh = false;
						if ((pm >> lane) & 1ull) {
							fresh = vis[nid] != (uint8_t)stamp;
							if (fresh)
								vis[nid] = (uint8_t)stamp;
						}
						const u64 fmask = __builtin_amdgcn_ballot_w64(fresh);
						ndis += (unsigned)__popcll(fmask);
						const float mydd = eval_lanes<NI, IS_L2, G>(q, g.vecs, g.dp4, nid, fmask, lane);
						const float wmax = nr < efC ? FLT_MAX : key_dis(rfl64(rkeys[efC - 1]));
						u64 mm = __builtin_amdgcn_ballot_w64(fresh && (nr < efC || mydd < wmax));
						while (mm) {
							const int l = (int)__builtin_ctzll(mm);
							mm &= mm - 1;
							const float dd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydd), l));
							// "if (results.size() < efConstruction || results.top().d > dis)"
							if (nr < efC || key_dis(rfl64(rkeys[nr - 1])) > dd)
								nr = sorted_insert(rkeys, nr, efC, mk_key(dd, __builtin_amdgcn_readlane(nid, l)), lane);
						}
						if (~vmask)
							break;
					}
				}
				// ---- shrink to the level's capacity, then link both ways (add_links_starting_from)
				const int nsel = shrink_select<NI, IS_L2, G>(g, rkeys, nr, L, out_id, out_d, lane, ndis);
				for (int j = lane; j < nsel; j += 64)
					sel_id[j] = out_id[nsel - 1 - j]; // priority_queue pops the farthest first
				if (lane == 0)
					ctl[1] = nsel;
				wave_fence();
				for (int t = 0; t < nsel; t++)
					add_link<NI, IS_L2, G>(g, q, pt, rfl(sel_id[t]), level, tkeys, out_id, out_d, lane, ndis, a.clean, a.stats ? a.stats + 2 : nullptr);
				wave_unlock(a, pt, lane);
			}
			__syncthreads();
			const int nsel = rfl(ctl[1]);
			for (int t = wave; t < nsel; t += W) {
				const int other = rfl(sel_id[t]);
				wave_lock(a, other, lane);
				QV<NI> qo;
				load_row(qo, g.vecs + (size_t)other * g.dp4 * 4, g.dp4, lane);
				add_link<NI, IS_L2, G>(g, qo, other, pt, level, tkeys, out_id, out_d, lane, ndis, a.clean, a.stats ? a.stats + 2 : nullptr);
				wave_unlock(a, other, lane);
			}
			__syncthreads();
			if (wave == 0)
				wave_lock(a, pt, lane);
		}
		if (wave == 0)
			wave_unlock(a, pt, lane);
	}
	if (lane == 0) {
		if (wave == 0)
			a.vstamp[blockIdx.x] = stamp;
		if (a.stats)
			atomicAdd(&a.stats[0], (unsigned long long)ndis);
	}
}

// ------------------------------------------------------------------------------------------------ dispatch on d

// rows in flight per wave: the search kernel wants OCCUPANCY (tools/micro/gather_bw.hip: random 3 KB rows reach
// 6.6 TB/s with 32 waves/CU whatever G is, 5.8 with 16), so it keeps G small to stay under 80 VGPRs; the build
// kernel holds more state and keeps G = 4.
// rows in flight per wave (G) vs occupancy: measured at C5 (N=1M, d=768, efSearch=128) every point between
// {G=2, 24 waves/CU, HBM visited table} and {G=16, 8 waves/CU, LDS visited hash} lands at 16-18 ms (3.2-3.6 TB/s of
// row bytes): 70 % of a wave's time is the row round trip of a hop at ~14 us loaded latency.  Default = the LDS hash
// (no 1-byte HBM probes, no multi-GB visited tables) with G = 16; options hnsw_search_g / hnsw_visited_lds /
// hnsw_search_waves select the others.
template <template <int, bool, int> class F, int NI, int G, typename... A>
void dispatch_metric(bool is_l2, A &&...args) {
	if (is_l2)
		F<NI, true, G>::run(std::forward<A>(args)...);
	else
		F<NI, false, G>::run(std::forward<A>(args)...);
}
// gsel: rows in flight per wave for the search kernel (0 = default); the build kernel uses its own fixed G
template <template <int, bool, int> class F, bool SEARCH, typename... A>
void dispatch_ni(int dp4, bool is_l2, int gsel, A &&...args) {
	const int ni = (dp4 + 63) / 64;
#define MVS_NI_SMALL(NI)                                                                                               \
	if (ni <= NI) {                                                                                                    \
		if constexpr (!SEARCH)                                                                                         \
			dispatch_metric<F, NI, 4>(is_l2, std::forward<A>(args)...);                                                \
		else if (gsel == 2)                                                                                            \
			dispatch_metric<F, NI, 2>(is_l2, std::forward<A>(args)...);                                                \
		else if (gsel == 8)                                                                                            \
			dispatch_metric<F, NI, 8>(is_l2, std::forward<A>(args)...);                                                \
		else                                                                                                           \
			dispatch_metric<F, NI, 16>(is_l2, std::forward<A>(args)...);                                               \
		return;                                                                                                        \
	}
#define MVS_NI_CASE(NI, GS, GB)                                                                                        \
	if (ni <= NI) {                                                                                                    \
		dispatch_metric<F, NI, SEARCH ? GS : GB>(is_l2, std::forward<A>(args)...);                                     \
		return;                                                                                                        \
	}
	MVS_NI_SMALL(1)
	MVS_NI_SMALL(2)
	MVS_NI_SMALL(3)
	MVS_NI_CASE(4, 8, 2)
	MVS_NI_CASE(6, 8, 2)
	MVS_NI_CASE(8, 4, 1)
	MVS_NI_CASE(16, 2, 1)
#undef MVS_NI_CASE
#undef MVS_NI_SMALL
	throw_faiss("mvs::HNSWIndex", __FILE__, "dimension %d exceeds the supported maximum 4096", dp4 * 4);
}

// resident workgroups (= waves) per CU of the search kernel instance for this LDS size
template <int NI, bool IS_L2, int G>
struct SearchOccupancy {
	static void run(int *out, size_t lds) {
		int nb = 0;
		ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G>, lds);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)hnsw_search_kernel<NI, IS_L2, G>, 64, lds) !=
		        hipSuccess ||
		    nb <= 0)
			nb = 8;
		*out = nb;
	}
};

template <int NI, bool IS_L2, int G>
struct SearchLaunch {
	static void run(const SearchArgs &a, int grid, size_t lds, hipStream_t st) {
		ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G>, lds); // map lookup; raises the limit if needed
		hipLaunchKernelGGL((hnsw_search_kernel<NI, IS_L2, G>), dim3(grid), dim3(64), lds, st, a);
		MVS_HIP(hipGetLastError());
	}
};
template <int NI, bool IS_L2, int G>
struct SearchOccupancyBF {
	static void run(int *out, size_t lds) {
		int nb = 0;
		ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G, true>, lds);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)hnsw_search_kernel<NI, IS_L2, G, true>, 64, lds) !=
		        hipSuccess ||
		    nb <= 0)
			nb = 8;
		*out = nb;
	}
};
template <int NI, bool IS_L2, int G>
struct SearchLaunchBF {
	static void run(const SearchArgs &a, int grid, size_t lds, hipStream_t st) {
		ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G, true>, lds);
		hipLaunchKernelGGL((hnsw_search_kernel<NI, IS_L2, G, true>), dim3(grid), dim3(64), lds, st, a);
		MVS_HIP(hipGetLastError());
	}
};
// (bf16 first look + the two lists in registers: level 1 = ef <= 128, k <= 64; level 2 = ef <= 256, k <= 256)
#define MVS_HNSW_RL_STRUCTS(LV)                                                                                        \
	template <int NI, bool IS_L2, int G>                                                                               \
	struct SearchOccupancyBFRL##LV {                                                                                   \
		static void run(int *out, size_t lds) {                                                                        \
			int nb = 0;                                                                                                \
			ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G, true, LV>, lds);                         \
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)hnsw_search_kernel<NI, IS_L2, G, true, LV>, 64, \
			                                                 lds) != hipSuccess ||                                     \
			    nb <= 0)                                                                                               \
				nb = 8;                                                                                                \
			*out = nb;                                                                                                 \
		}                                                                                                              \
	};                                                                                                                 \
	template <int NI, bool IS_L2, int G>                                                                               \
	struct SearchLaunchBFRL##LV {                                                                                      \
		static void run(const SearchArgs &a, int grid, size_t lds, hipStream_t st) {                                   \
			ensure_dynamic_lds((const void *)hnsw_search_kernel<NI, IS_L2, G, true, LV>, lds);                         \
			hipLaunchKernelGGL((hnsw_search_kernel<NI, IS_L2, G, true, LV>), dim3(grid), dim3(64), lds, st, a);        \
			MVS_HIP(hipGetLastError());                                                                                \
		}                                                                                                              \
	};
MVS_HNSW_RL_STRUCTS(1)
MVS_HNSW_RL_STRUCTS(2)
MVS_HNSW_RL_STRUCTS(3)
#undef MVS_HNSW_RL_STRUCTS
template <int NI, bool IS_L2, int G>
struct BuildLaunch {
	template <int W>
	static void run_w(const BuildArgs &a, int grid, size_t lds, hipStream_t st) {
		ensure_dynamic_lds((const void *)hnsw_build_kernel<NI, IS_L2, G, W>, lds);
		hipLaunchKernelGGL((hnsw_build_kernel<NI, IS_L2, G, W>), dim3(grid), dim3(64 * W), lds, st, a);
		MVS_HIP(hipGetLastError());
	}
	// wg: wavefronts per inserted point (1 | 2 | 4 | 8)
	static void run(const BuildArgs &a, int grid, size_t lds, hipStream_t st, int wg) {
		if (wg >= 8)
			run_w<8>(a, grid, lds, st);
		else if (wg >= 4)
			run_w<4>(a, grid, lds, st);
		else if (wg >= 2)
			run_w<2>(a, grid, lds, st);
		else
			run_w<1>(a, grid, lds, st);
	}
};

// device buffer that keeps its contents when it grows
struct KeepBuf {
	void *p = nullptr;
	size_t cap = 0;
	void ensure(size_t need, size_t used, hipStream_t st, int fill = -2) {
		if (need <= cap)
			return;
		size_t nc = cap ? cap : 4096;
		while (nc < need)
			nc = nc + nc / 2 + 4096;
		void *np = nullptr;
		MVS_HIP(hipMalloc(&np, nc));
		if (fill != -2)
			MVS_HIP(hipMemsetAsync(np, fill, nc, st));
		if (used > 0 && p)
			MVS_HIP(hipMemcpyAsync(np, p, used, hipMemcpyDeviceToDevice, st));
		MVS_HIP(hipStreamSynchronize(st));
		if (p)
			MVS_HIP(hipFree(p));
		p = np;
		cap = nc;
	}
	void release() {
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

} // namespace

// =================================================================================================== HNSWIndex

class HNSWIndex : public IndexBase {
public:
	int M;
	int efConstruction = 40; // HNSW::efConstruction default
	int efSearch = 16;       // HNSW::efSearch default (the glue always passes SearchParametersHNSW, :693)
	int dp, dp4;
	int64_t build_waves = 0; // 0 = auto (concurrent, FAISS OpenMP semantics); 1 = deterministic order
	const bool profile_build = getenv("MVS_INGEST_PROFILE") != nullptr;
	int build_shortcut = 1;  // option hnsw_build_shortcut: add_link's short cut for full lists that are their last shrink's output (0: always the full pass)
	int build_wg = 4;        // option hnsw_build_wg: wavefronts that share one insertion's back links (1 | 2 | 4 | 8)
	int entry_point = -1, max_level = -1;
	double last_evals = 0, last_bf16_rows = 0, last_f32_rows_pub = 0; // counters of the last timed search (hnsw_walk_stats)

	HNSWIndex(int d_, int M_, int metric_) : IndexBase(MVS_KIND_HNSW, d_, metric_), M(M_), rng(12345) {
		if (metric != METRIC_L2 && metric != METRIC_IP)
			throw_faiss("mvs::HNSWIndex", __FILE__, "metric type %d is not implemented on the MI355X path", metric);
		if (M < 2 || M > 512)
			throw_faiss("mvs::HNSWIndex", __FILE__, "HNSW M = %d outside the supported range [2, 512]", M);
		dp = (d + 3) / 4 * 4;
		dp4 = dp / 4;
		if (dp4 > 64 * 16)
			throw_faiss("mvs::HNSWIndex", __FILE__, "dimension %d exceeds the supported maximum 4096", d);
		// HNSW::set_default_probas(M, 1 / log(M))
		const double mult = 1.0 / std::log((double)M);
		int nn = 0;
		cum_nn.push_back(0);
		for (int level = 0;; level++) {
			const double proba = std::exp(-level / mult) * (1 - std::exp(-1 / mult));
			if (proba < 1e-9)
				break;
			assign_probas.push_back(proba);
			nn += level == 0 ? M * 2 : M;
			cum_nn.push_back(nn);
		}
		offsets_h.push_back(0);
	}
	~HNSWIndex() override {
		(void)hipSetDevice(device);
		if (stream)
			(void)hipStreamSynchronize(stream);
		vecs.release();
		offsets.release();
		neighbors.release();
		locks.release();
		clean.release();
		nb0.release();
		vbf.release();
		ymax_dev.release();
		if (h_stats)
			(void)hipHostFree(h_stats);
	}

	// ---------------------------------------------------------------------------------------------- add
	int random_level() {
		double f = (double)((float)rng() / (float)rng.max()); // RandomGenerator::rand_float
		for (size_t level = 0; level < assign_probas.size(); level++) {
			if (f < assign_probas[level])
				return (int)level;
			f -= assign_probas[level];
		}
		return (int)assign_probas.size() - 1;
	}

	// dense level-0 adjacency for the search kernel, extended after every add
	void sync_level0() {
		if (nb0_rows >= ntotal)
			return;
		const int L = 2 * M;
		nb0.ensure((size_t)ntotal * L * 4, 0, stream);
		// reverse links touch old vertices too: refresh everything (N * 8M bytes, once per add batch)
		const long long tot = (long long)ntotal * L;
		hipLaunchKernelGGL(hnsw_level0_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
		                   (const long long *)offsets.p, (const int32_t *)neighbors.p, (int32_t *)nb0.p, 0ll,
		                   (long long)ntotal, L);
		MVS_HIP(hipGetLastError());
		nb0_rows = ntotal;
	}

	// bf16 copy of the rows for the first look of the search walk (option hnsw_bf16), extended after every add
	void sync_bf16() {
		if (vbf_rows >= ntotal)
			return;
		vbf.ensure((size_t)(vecs.cap / sizeof(float)) * sizeof(unsigned short), (size_t)vbf_rows * dp * sizeof(unsigned short), stream);
		if (!ymax_dev.p) {
			ymax_dev.ensure(64, 0, stream, 0);
		}
		const long long nr = ntotal - vbf_rows;
		hipLaunchKernelGGL(hnsw_rows_to_bf16_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, stream, (const float *)vecs.p,
		                   (long long)vbf_rows, (long long)ntotal, dp, (unsigned short *)vbf.p, (unsigned *)ymax_dev.p);
		MVS_HIP(hipGetLastError());
		vbf_rows = ntotal;
	}

	GraphDev graph_dev() const {
		GraphDev g;
		g.vecs = (const float *)vecs.p;
		g.dp4 = dp4;
		g.offsets = (const long long *)offsets.p;
		g.neighbors = (int32_t *)neighbors.p;
		g.M = M;
		return g;
	}

	void launch_build(const int32_t *d_order, int i0, int i1, int pt_level, int waves) {
		if (i1 <= i0)
			return;
		BuildArgs a;
		a.g = graph_dev();
		a.entry_point = entry_point;
		a.max_level = max_level;
		a.efC = efConstruction;
		a.order = d_order;
		a.i0 = i0;
		a.i1 = i1;
		a.pt_level = pt_level;
		a.counter = (int *)ws_counter.p;
		a.locks = (int *)locks.p;
		a.use_locks = waves > 1;
		a.count_spins = profile_build ? 1 : 0;
		a.clean = build_shortcut ? (unsigned char *)clean.p : nullptr;
		a.visited = (uint8_t *)bvis.p;
		a.vstride = bvis_stride;
		a.vstamp = (unsigned *)bstamp.p;
		a.stats = (unsigned long long *)ws_stats.p;
		MVS_HIP(hipMemsetAsync(ws_counter.p, 0, sizeof(int), stream));
		const int L0 = 2 * M;
		const int wg = build_wg >= 8 ? 8 : (build_wg >= 4 ? 4 : (build_wg >= 2 ? 2 : 1)); // wavefronts per inserted point (hnsw_build_kernel)
		const size_t lds = (size_t)efConstruction * 8 + (size_t)L0 * 4 + 8 + (size_t)wg * (L0 + 1) * 16 + 64;
		dispatch_ni<BuildLaunch, false>(dp4, metric == METRIC_L2, 0, a, waves, lds, stream, wg);
	}

	// d_x: [n][d] rows on device, ordered after everything enqueued on `stream`
	void add_core_device(int64_t n, const float *d_x) {
		if (n <= 0)
			return;
		if (ntotal + n > (int64_t)0x3fffffff)
			throw_faiss("mvs::HNSWIndex::add", __FILE__, "a single-device HNSW index holds at most 2^30 rows");
		if (efConstruction < 1 || efConstruction > 4096)
			throw_faiss("mvs::HNSWIndex::add", __FILE__, "efConstruction = %d outside the supported range [1, 4096]",
			            efConstruction);
		const int64_t n0 = ntotal, nt = ntotal + n;
		// storage->add(n, x)
		vecs.ensure((size_t)nt * dp * sizeof(float), (size_t)n0 * dp * sizeof(float), stream);
		launch_pad_rows(d_x, n, d, (float *)vecs.p + (size_t)n0 * dp, dp, stream);
		// prepare_level_tab
		levels_h.resize((size_t)nt);
		offsets_h.resize((size_t)nt + 1);
		int bucket_max = 0;
		for (int64_t i = 0; i < n; i++) {
			const int pt_level = random_level();
			levels_h[(size_t)(n0 + i)] = pt_level + 1;
			bucket_max = std::max(bucket_max, pt_level);
			offsets_h[(size_t)(n0 + i + 1)] = offsets_h[(size_t)(n0 + i)] + cum_nn[(size_t)pt_level + 1];
		}
		offsets.ensure((size_t)(nt + 1) * sizeof(int64_t), (size_t)(n0 + 1) * sizeof(int64_t), stream);
		MVS_HIP(hipMemcpyAsync((int64_t *)offsets.p + n0, &offsets_h[(size_t)n0], (size_t)(n + 1) * sizeof(int64_t),
		                       hipMemcpyHostToDevice, stream));
		const size_t nb_old = (size_t)offsets_h[(size_t)n0] * 4, nb_new = (size_t)offsets_h[(size_t)nt] * 4;
		neighbors.ensure(nb_new, nb_old, stream);
		MVS_HIP(hipMemsetAsync((char *)neighbors.p + nb_old, 0xFF, nb_new - nb_old, stream)); // -1 = empty slot
		locks.ensure((size_t)nt * sizeof(int), (size_t)n0 * sizeof(int), stream, 0);
		clean.ensure((size_t)nt, (size_t)n0, stream, 0);
		// hnsw_add_vertices: bucket sort by level, per bucket (highest level first) shuffle with rng2(789)
		std::vector<int> hist((size_t)bucket_max + 1, 0);
		for (int64_t i = 0; i < n; i++)
			hist[(size_t)levels_h[(size_t)(n0 + i)] - 1]++;
		std::vector<int> off((size_t)bucket_max + 2, 0);
		for (int l = 0; l <= bucket_max; l++)
			off[(size_t)l + 1] = off[(size_t)l] + hist[(size_t)l];
		std::vector<int32_t> order((size_t)n);
		{
			std::vector<int> cur(off.begin(), off.end() - 1);
			for (int64_t i = 0; i < n; i++)
				order[(size_t)cur[(size_t)levels_h[(size_t)(n0 + i)] - 1]++] = (int32_t)(n0 + i);
		}
		std::mt19937 rng2(789);
		{
			int i1 = (int)n;
			for (int pt_level = bucket_max; pt_level >= 0; pt_level--) {
				const int i0 = i1 - hist[(size_t)pt_level];
				for (int j = i0; j < i1; j++)
					std::swap(order[(size_t)j], order[(size_t)(j + (int)(rng2() % (uint32_t)(i1 - j)))]);
				i1 = i0;
			}
		}
		ws_order.reserve((size_t)n * sizeof(int32_t));
		MVS_HIP(hipMemcpyAsync(ws_order.p, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
		ws_counter.reserve(64);
		ws_stats.reserve(128);
		MVS_HIP(hipMemsetAsync(ws_stats.p, 0, 32, stream));
		// visited tables of the build waves
		int max_waves = build_waves > 0 ? (int)std::min<int64_t>(build_waves, 4096) : 1024;
		const size_t vcap = vecs.cap / ((size_t)dp * sizeof(float)); // rows the vector store can hold
		const size_t stride = (vcap + 15) / 16 * 16;
		max_waves = (int)std::max<size_t>(1, std::min<size_t>((size_t)max_waves, ((size_t)8 << 30) / stride));
		if (stride != (size_t)bvis_stride || (size_t)max_waves > (size_t)bvis_waves) {
			bvis.reserve(stride * (size_t)max_waves);
			bstamp.reserve((size_t)max_waves * sizeof(unsigned));
			MVS_HIP(hipMemsetAsync(bvis.p, 0, stride * (size_t)max_waves, stream));
			MVS_HIP(hipMemsetAsync(bstamp.p, 0, (size_t)max_waves * sizeof(unsigned), stream));
			bvis_stride = (int64_t)stride;
			bvis_waves = max_waves;
		}
		// insert, bucket by bucket
		ntotal = nt; // the kernels address rows up to nt
		int64_t inserted = n0;
		int i1 = (int)n;
		for (int pt_level = bucket_max; pt_level >= 0; pt_level--) {
			const int i0 = i1 - hist[(size_t)pt_level];
			int pos = i0;
			if (pos < i1 && entry_point < 0) { // very first vertex: becomes the entry point, no links
				entry_point = order[(size_t)pos];
				max_level = pt_level;
				pos++;
				inserted++;
			} else if (pos < i1 && pt_level > max_level) {
				// the first point above the current top level is linked alone, then becomes the entry point
				launch_build((const int32_t *)ws_order.p, pos, pos + 1, pt_level, 1);
				entry_point = order[(size_t)pos];
				max_level = pt_level;
				pos++;
				inserted++;
			}
			while (pos < i1) {
				int waves, seg;
				if (build_waves == 1) {
					waves = 1;
					seg = i1 - pos;
				} else {
					// concurrency grows with the graph so that concurrent inserts stay a small fraction of it
					waves = (int)std::max<int64_t>(1, std::min<int64_t>(max_waves, inserted / 32));
					seg = std::min(i1 - pos, waves * 8);
				}
				launch_build((const int32_t *)ws_order.p, pos, pos + seg, pt_level, std::min(waves, seg));
				pos += seg;
				inserted += seg;
			}
			i1 = i0;
		}
		MVS_HIP(hipStreamSynchronize(stream)); // `order` (pageable) and the caller's rows are done
		unsigned long long st[4] = {0, 0, 0, 0};
		MVS_HIP(hipMemcpy(st, ws_stats.p, sizeof st, hipMemcpyDeviceToHost));
		build_distances += st[0];
		build_shortcuts += st[2];
		if (profile_build)
			fprintf(stderr, "hnswprofile\tadd(%lld rows): %llu distance evaluations, %llu short cuts, %llu failed lock attempts (each followed by s_sleep 8)\n",
			        (long long)n, st[0], st[2], st[1]);
	}
	void add(int64_t n, const float *x) override {
		use_device();
		if (n <= 0)
			return;
		DevBuf dx;
		dx.reserve((size_t)n * d * sizeof(float));
		MVS_HIP(hipMemcpyAsync(dx.p, x, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice, stream));
		add_core_device(n, (const float *)dx.p);
	}
	void add_device(int64_t n, const float *d_x, hipStream_t st) override {
		use_device();
		if (n <= 0)
			return;
		stream_wait(stream, st);
		add_core_device(n, d_x);
		stream_wait(st, stream);
	}

	// ---------------------------------------------------------------------------------------------- search
	void search_mapped(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   const int64_t *d_idmap, hipStream_t st) override {
		use_device();
		if (k <= 0)
			throw_faiss("virtual void faiss::IndexHNSW::search(...) const", "faiss/IndexHNSW.cpp",
			            "Error: 'k > 0' failed");
		if (nq <= 0)
			return;
		const int64_t efs = params && params->efSearch > 0 ? params->efSearch : efSearch;
		const int64_t ef = std::max(efs, k);
		if (ef > 4096)
			throw_faiss("mvs::HNSWIndex::search", __FILE__, "max(efSearch, k) = %lld exceeds the supported maximum 4096",
			            (long long)ef);
		stream_wait(stream, st);
		memset(&kinfo, 0, sizeof kinfo);
		if (ntotal == 0 || entry_point < 0) {
			const long long tot = nq * k;
			hipLaunchKernelGGL(hnsw_fill_empty_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, d_D,
			                   (long long *)d_I, tot, metric == METRIC_L2 ? FLT_MAX : -FLT_MAX);
			MVS_HIP(hipGetLastError());
			stream_wait(st, stream);
			return;
		}
		ws_q.reserve((size_t)nq * dp * sizeof(float));
		launch_pad_rows(d_x, nq, d, (float *)ws_q.p, dp, stream);
		// visited hash: ~15 distance evaluations per unit of ef on clustered data; 4096 slots cover ef = 128 at < 1/2 load
		int hsize = 0;
		if (visited_lds) {
			hsize = 1024;
			while (hsize < 24 * ef && hsize < 8192)
				hsize *= 2;
			if (visited_lds >= 1024) // explicit slot count (power of two)
				hsize = visited_lds;
		}
		const bool use_bf = bf16_look != 0 && d >= 64;
		// the candidate / result lists in registers (csrc: "sorted lists in REGISTERS"): option hnsw_reg_lists, with the bf16 instances
		const int use_rl = !(use_bf && reg_lists != 0) ? 0
		                   : ((ef <= 128 && k <= 64) ? 1 : ((ef <= 256 && k <= 256) ? 2 : ((ef <= 512 && k <= 512) ? 3 : 0)));
		// (register lists: the LDS holds the visited hash only -- 16 KB at ef = 128: ten waves per CU instead of nine)
		const size_t lds = use_rl ? std::max<size_t>((size_t)hsize * 4, 64) : (size_t)(ef + k) * 8 + (size_t)hsize * 4 + 64;
		// rows in flight per wave (option hnsw_search_g): with the lists in registers the walk is no longer a chain of LDS round trips
		// and 8 rows in flight at three waves per SIMD beat 16 at two (C5: 9.5-10.0 vs 11.4-11.9 ms)
		const int search_g = this->search_g ? this->search_g : (use_rl ? 8 : 16);
		// one workgroup = one wave; fill every resident slot the kernel instance allows (VGPRs / LDS)
		if (cus <= 0) {
			int v = 0;
			cus = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0 ? v : 256;
		}
		if (occ_lds != lds || occ_g != search_g || occ_bf != (int)use_bf + 2 * use_rl) { // the occupancy query is not free: once per LDS size
			int v = 8;
			if (use_rl == 3)
				dispatch_ni<SearchOccupancyBFRL3, true>(dp4, metric == METRIC_L2, search_g, &v, lds);
			else if (use_rl == 2)
				dispatch_ni<SearchOccupancyBFRL2, true>(dp4, metric == METRIC_L2, search_g, &v, lds);
			else if (use_rl)
				dispatch_ni<SearchOccupancyBFRL1, true>(dp4, metric == METRIC_L2, search_g, &v, lds);
			else if (use_bf)
				dispatch_ni<SearchOccupancyBF, true>(dp4, metric == METRIC_L2, search_g, &v, lds);
			else
				dispatch_ni<SearchOccupancy, true>(dp4, metric == METRIC_L2, search_g, &v, lds);
			occ_bf = (int)use_bf + 2 * use_rl;
			occ_g = search_g;
			occ_waves = std::max(1, std::min(v, 32));
			occ_lds = lds;
		}
		const int per_cu = search_waves_per_cu > 0 ? std::min(search_waves_per_cu, occ_waves) : occ_waves;
		// visited tables (one byte per vertex per wave) persist across searches; the rolling stamp makes a fresh
		// table unnecessary, they are zeroed only when (re)allocated
		const size_t vcap = vecs.cap / ((size_t)dp * sizeof(float));
		const size_t stride = (vcap + 15) / 16 * 16;
		int grid = (int)std::min<int64_t>(nq, (int64_t)cus * per_cu);
		grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)grid, ((size_t)16 << 30) / stride));
		if ((int64_t)stride != svis_stride || grid > svis_waves) {
			const int nw = std::max(grid, svis_waves);
			ws_vis.reserve(stride * (size_t)nw);
			sstamp.reserve((size_t)nw * sizeof(unsigned));
			MVS_HIP(hipMemsetAsync(ws_vis.p, 0, stride * (size_t)nw, stream));
			MVS_HIP(hipMemsetAsync(sstamp.p, 0, (size_t)nw * sizeof(unsigned), stream));
			svis_stride = (int64_t)stride;
			svis_waves = nw;
		}
		ws_stats.reserve(128);
		MVS_HIP(hipMemsetAsync(ws_stats.p, 0, 128, stream));
		ws_counter.reserve(64);
		MVS_HIP(hipMemsetAsync(ws_counter.p, 0, sizeof(int), stream));
		SearchArgs a;
		a.g = graph_dev();
		a.entry_point = entry_point;
		a.max_level = max_level;
		a.xq = (const float *)ws_q.p;
		a.nq = nq;
		a.k = (int)k;
		a.ef = (int)ef;
		a.efSearch = (int)efs;
		a.hsize = hsize;
		sync_level0();
		a.nb0 = (const int32_t *)nb0.p;
		a.sel = selector.upload(params, stream);
		a.idmap = (const long long *)d_idmap;
		a.label_offset = label_offset;
		a.visited = (uint8_t *)ws_vis.p;
		a.vstride = (long long)stride;
		a.vstamp = (unsigned *)sstamp.p;
		a.counter = (int *)ws_counter.p;
		a.D = d_D;
		a.I = (long long *)d_I;
		a.stats = (unsigned long long *)ws_stats.p;
		a.vbf = nullptr;
		a.ymax_bits = nullptr;
		if (use_bf) {
			sync_bf16();
			a.vbf = (const unsigned short *)vbf.p;
			a.ymax_bits = (const unsigned *)ymax_dev.p;
		}
		begin_kernel_timing(stream);
		if (use_rl == 3)
			dispatch_ni<SearchLaunchBFRL3, true>(dp4, metric == METRIC_L2, search_g, a, grid, lds, stream);
		else if (use_rl == 2)
			dispatch_ni<SearchLaunchBFRL2, true>(dp4, metric == METRIC_L2, search_g, a, grid, lds, stream);
		else if (use_rl)
			dispatch_ni<SearchLaunchBFRL1, true>(dp4, metric == METRIC_L2, search_g, a, grid, lds, stream);
		else if (use_bf)
			dispatch_ni<SearchLaunchBF, true>(dp4, metric == METRIC_L2, search_g, a, grid, lds, stream);
		else
			dispatch_ni<SearchLaunch, true>(dp4, metric == METRIC_L2, search_g, a, grid, lds, stream);
		end_kernel_timing(stream);
		snprintf(kinfo.name, sizeof kinfo.name, "hnsw_search_kernel");
		kinfo.grid = grid;
		kinfo.block = 64;
		kinfo.lds_bytes = (int)lds;
		if (timing_enabled) {
			// the walk-length counters are only fetched when a bench asked for kernel timing: the plain search path
			// stays asynchronous on the caller's stream
			if (!h_stats)
				MVS_HIP(hipHostMalloc((void **)&h_stats, 128, hipHostMallocDefault));
			MVS_HIP(hipMemcpyAsync(h_stats, ws_stats.p, 128, hipMemcpyDeviceToHost, stream));
			MVS_HIP(hipStreamSynchronize(stream));
#ifdef MVS_HNSW_PROFILE
			fprintf(stderr,
			        "[hnsw profile] shader clocks: descent %.3g pop %.3g nbr %.3g visited %.3g eval %.3g insert %.3g | "
			        "total %.3g\n",
			        (double)h_stats[2], (double)h_stats[3], (double)h_stats[4], (double)h_stats[5], (double)h_stats[6],
			        (double)h_stats[7], (double)h_stats[8]);
#endif
			const unsigned long long nd = h_stats[0], ne = h_stats[1];
			last_f32_rows = use_bf ? (double)h_stats[9] : (double)nd;
			last_evals = (double)nd;
			last_f32_rows_pub = last_f32_rows;
			last_bf16_rows = use_bf ? (double)nd : 0.0; // (upper bound: before both lists are full a neighbour skips the first look)
			if (getenv("MVS_HNSW_STATS"))
				fprintf(stderr, "[hnsw] %llu distance evaluations, %.0f f32 rows fetched (%.1f %%), bf16 first look %s\n", nd, last_f32_rows,
				        nd ? 100.0 * last_f32_rows / (double)nd : 0.0, use_bf ? "on" : "off");
			kinfo.bytes = (double)nd * ((double)d * 4.0 + 4.0); // n_visited * (4d + 4), counted by the kernel
			kinfo.flops = (double)nd * d * (metric == METRIC_L2 ? 3.0 : 2.0);
			kinfo.nsplit = (int)(ne / (unsigned long long)std::max<int64_t>(nq, 1)); // mean expanded vertices per query
		}
		stream_wait(st, stream);
	}
	void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   hipStream_t st) override {
		search_mapped(nq, d_x, k, d_D, d_I, params, nullptr, st);
	}

	// ---------------------------------------------------------------------------------------------- placement
	void to_device(int new_device) override {
		if (new_device == device)
			return;
		throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp",
		            "moving an HNSW index between devices is not implemented on the MI355X path yet");
	}
	// deep copy through the host image (faiss::gpu::index_cpu_to_gpu also starts from host memory)
	IndexBase *clone(int on_device) override {
		int ndev = 0;
		MVS_HIP(hipGetDeviceCount(&ndev));
		if (on_device < 0 || on_device >= ndev)
			throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "Invalid GPU device %d", on_device);
		HostIndex h;
		to_host(h);
		HNSWIndex *c = static_cast<HNSWIndex *>(index_from_host(h, on_device));
		c->rng = rng; // a clone continues the level stream where the source stands (read_index restarts it, like FAISS)
		c->build_waves = build_waves;
		c->build_wg = build_wg;
		c->build_shortcut = build_shortcut;
		c->label_offset = label_offset;
		return c;
	}
	// IndexHNSWFlat image: struct HNSW + flat storage
	void to_host(HostIndex &out) override {
		use_device();
		MVS_HIP(hipStreamSynchronize(stream));
		out.kind = MVS_KIND_HNSW;
		out.d = d;
		out.metric = metric;
		out.ntotal = ntotal;
		out.is_trained = true;
		out.assign_probas = assign_probas;
		out.cum_nneighbor_per_level.assign(cum_nn.begin(), cum_nn.end());
		out.levels = levels_h;
		out.offsets.assign(offsets_h.begin(), offsets_h.end());
		out.neighbors.resize((size_t)graph_slots());
		if (graph_slots() > 0)
			MVS_HIP(hipMemcpy(out.neighbors.data(), neighbors.p, (size_t)graph_slots() * 4, hipMemcpyDeviceToHost));
		out.entry_point = entry_point;
		out.max_level = max_level;
		out.efConstruction = efConstruction;
		out.efSearch = efSearch;
		out.sub.reset(new HostIndex);
		HostIndex &st = *out.sub;
		st.kind = MVS_KIND_FLAT;
		st.d = d;
		st.metric = metric;
		st.ntotal = ntotal;
		st.rows.resize((size_t)ntotal * d);
		if (ntotal > 0) {
			std::vector<float> tmp((size_t)ntotal * dp);
			MVS_HIP(hipMemcpy(tmp.data(), vecs.p, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
			for (int64_t i = 0; i < ntotal; i++)
				memcpy(&st.rows[(size_t)i * d], &tmp[(size_t)i * dp], (size_t)d * sizeof(float));
		}
	}
	void adopt(const HostIndex &h) {
		use_device();
		const int64_t n = h.ntotal;
		if (!h.sub || h.sub->kind != MVS_KIND_FLAT || h.sub->ntotal != n || (int64_t)h.levels.size() != n ||
		    (int64_t)h.offsets.size() != n + 1 || h.neighbors.size() != (size_t)h.offsets[(size_t)n])
			throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
			            "inconsistent IndexHNSWFlat image (storage must be a Flat index of ntotal rows)");
		for (int64_t i = 0; i < n; i++) // the kernels derive the per-level ranges from M
			if (h.offsets[(size_t)i + 1] - h.offsets[(size_t)i] != (uint64_t)(h.levels[(size_t)i] + 1) * (uint64_t)M)
				throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
				            "HNSW image with a non-default neighbour layout is not supported on the MI355X path");
		// a truncated or crafted file must not make the walk kernels read out of bounds: every link is -1 or a vertex,
		// the entry point is a vertex that exists on the top level, no vertex is taller than max_level
		if (n == 0 ? (h.entry_point != -1) : (h.entry_point < 0 || h.entry_point >= n))
			throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
			            "HNSW image: entry point %d outside [0, %lld)", h.entry_point, (long long)n);
		int top = -1;
		for (int64_t i = 0; i < n; i++) {
			if (h.levels[(size_t)i] < 1 || h.levels[(size_t)i] > 64)
				throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
				            "HNSW image: vertex %lld has level count %d", (long long)i, h.levels[(size_t)i]);
			top = std::max(top, h.levels[(size_t)i] - 1);
		}
		if (h.max_level != top || (n > 0 && h.levels[(size_t)h.entry_point] - 1 != h.max_level))
			throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
			            "HNSW image: max_level %d does not match the level table (top %d)", h.max_level, top);
		for (size_t s = 0; s < h.neighbors.size(); s++)
			if (h.neighbors[s] < -1 || h.neighbors[s] >= n)
				throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
				            "HNSW image: neighbour slot %zu holds %d, outside [-1, %lld)", s, h.neighbors[s], (long long)n);
		efConstruction = h.efConstruction;
		efSearch = h.efSearch;
		entry_point = h.entry_point;
		max_level = h.max_level;
		levels_h = h.levels;
		offsets_h.assign(h.offsets.begin(), h.offsets.end());
		vecs.ensure(std::max<size_t>((size_t)n * dp * sizeof(float), 16), 0, stream);
		offsets.ensure((size_t)(n + 1) * sizeof(int64_t), 0, stream);
		neighbors.ensure(std::max<size_t>(h.neighbors.size() * 4, 16), 0, stream);
		locks.ensure(std::max<size_t>((size_t)n * 4, 16), 0, stream, 0);
		clean.ensure(std::max<size_t>((size_t)n, 16), 0, stream, 0);
		MVS_HIP(hipMemsetAsync(clean.p, 0, clean.cap, stream)); // (nothing is known about the lists of a loaded graph)
		if (n > 0) {
			std::vector<float> tmp((size_t)n * dp, 0.f);
			for (int64_t i = 0; i < n; i++)
				memcpy(&tmp[(size_t)i * dp], &h.sub->rows[(size_t)i * d], (size_t)d * sizeof(float));
			MVS_HIP(hipMemcpy(vecs.p, tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
			MVS_HIP(hipMemcpy(neighbors.p, h.neighbors.data(), h.neighbors.size() * 4, hipMemcpyHostToDevice));
		}
		MVS_HIP(hipMemcpy(offsets.p, offsets_h.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
		ntotal = n;
	}
	bool named_stat(const char *name, int64_t *v) override {
		if (!strcmp(name, "hnsw_build_distances"))
			*v = (int64_t)build_distances;
		else if (!strcmp(name, "hnsw_build_shortcuts"))
			*v = (int64_t)build_shortcuts;
		else
			return false;
		return true;
	}
	bool set_option(const char *key, int64_t v) override {
		if (!strcmp(key, "hnsw_build_waves")) {
			build_waves = v;
			return true;
		}
		if (!strcmp(key, "hnsw_build_shortcut")) {
			use_device();
			if (clean.p) { // (lists change unflagged while the short cut is off: nothing is known about them when it comes back)
				MVS_HIP(hipMemsetAsync(clean.p, 0, clean.cap, stream));
				MVS_HIP(hipStreamSynchronize(stream));
			}
			build_shortcut = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_build_wg")) {
			build_wg = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_ef_search")) {
			efSearch = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_reg_lists")) {
			reg_lists = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_bf16")) { // bf16 first look of the search walk (1) or every fresh neighbour's f32 row (0)
			bf16_look = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_search_g")) {
			search_g = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_visited_lds")) {
			visited_lds = (int)v;
			return true;
		}
		if (!strcmp(key, "hnsw_search_waves")) {
			search_waves_per_cu = (int)v;
			return true;
		}
		return false;
	}

	// introspection for parity tests (graph equality against the oracle)
	int64_t graph_slots() const {
		return offsets_h[(size_t)ntotal];
	}
	void get_graph(int32_t *levels_out, int64_t *offsets_out, int32_t *neighbors_out) {
		use_device();
		MVS_HIP(hipStreamSynchronize(stream));
		memcpy(levels_out, levels_h.data(), (size_t)ntotal * sizeof(int32_t));
		memcpy(offsets_out, offsets_h.data(), (size_t)(ntotal + 1) * sizeof(int64_t));
		if (graph_slots() > 0)
			MVS_HIP(hipMemcpy(neighbors_out, neighbors.p, (size_t)graph_slots() * 4, hipMemcpyDeviceToHost));
	}
	unsigned long long build_distances = 0, build_shortcuts = 0; // (mvs_index_get_stat: evaluations of all builds | add_link calls that took the short cut)

private:
	std::mt19937 rng; // RandomGenerator(12345)
	std::vector<double> assign_probas;
	std::vector<int> cum_nn;
	std::vector<int32_t> levels_h;
	std::vector<int64_t> offsets_h;
	KeepBuf vecs, offsets, neighbors, locks, clean, nb0, vbf, ymax_dev;
	int64_t nb0_rows = 0, vbf_rows = 0;
	int reg_lists = 1; // option hnsw_reg_lists: candidate / result lists in registers when ef <= 128 and k <= 64 (0: LDS arrays)
	int bf16_look = 1, occ_bf = -1; // option hnsw_bf16: bf16 first look of the search walk (0: every fresh neighbour's f32 row is fetched)
	double last_f32_rows = 0;
	DevBuf ws_order, ws_counter, ws_stats, ws_q, ws_vis, sstamp, bvis, bstamp;
	int cus = 0, occ_waves = 0, occ_g = -1;
	int search_g = 0, search_waves_per_cu = 0; // options hnsw_search_g / hnsw_search_waves (0 = default)
	int visited_lds = 1;                       // option hnsw_visited_lds
	size_t occ_lds = 0;
	unsigned long long *h_stats = nullptr; // pinned
	int64_t bvis_stride = 0, svis_stride = 0;
	int bvis_waves = 0, svis_waves = 0;
	SelectorHolder selector;
};

IndexBase *make_hnsw_index(int d, const std::string &desc, int metric) {
	if (desc.rfind("HNSW", 0) != 0)
		return nullptr;
	// index_factory.cpp: "HNSW<M>" and "HNSW<M>,Flat" -> IndexHNSWFlat(d, M, metric); bare "HNSW" -> M = 32
	char *end = nullptr;
	long M = strtol(desc.c_str() + 4, &end, 10);
	if (end == desc.c_str() + 4)
		M = 32;
	if (*end == 0 || !strcmp(end, ",Flat"))
		return new HNSWIndex(d, (int)M, metric);
	throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
	            "This index type is not implemented on the MI355X path yet: %s", desc.c_str());
}
IndexBase *hnsw_from_host(const HostIndex &h, int device) {
	CtorDevice scope(device);
	if (h.cum_nneighbor_per_level.size() < 2 || h.cum_nneighbor_per_level[1] % 2 != 0)
		throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp", "bad HNSW level table");
	auto *x = new HNSWIndex(h.d, h.cum_nneighbor_per_level[1] / 2, h.metric);
	try {
		x->adopt(h);
	} catch (...) {
		delete x;
		throw;
	}
	return x;
}
bool hnsw_set_ef_construction(IndexBase *ix, int v) {
	if (ix->kind != MVS_KIND_HNSW)
		return false;
	static_cast<HNSWIndex *>(ix)->efConstruction = v;
	return true;
}
int hnsw_get_ef_construction(IndexBase *ix) {
	return ix->kind == MVS_KIND_HNSW ? static_cast<HNSWIndex *>(ix)->efConstruction : -1;
}
int64_t hnsw_graph_info(IndexBase *ix, int *max_level, int *entry_point) {
	if (ix->kind != MVS_KIND_HNSW)
		return -1;
	auto *h = static_cast<HNSWIndex *>(ix);
	if (max_level)
		*max_level = h->max_level;
	if (entry_point)
		*entry_point = h->entry_point;
	return h->graph_slots();
}
bool hnsw_walk_stats(IndexBase *ix, double *evaluations, double *f32_rows, double *bf16_rows) {
	if (ix->kind != MVS_KIND_HNSW)
		return false;
	auto *h = static_cast<HNSWIndex *>(ix);
	if (evaluations)
		*evaluations = h->last_evals;
	if (f32_rows)
		*f32_rows = h->last_f32_rows_pub;
	if (bf16_rows)
		*bf16_rows = h->last_bf16_rows;
	return true;
}
bool hnsw_get_graph(IndexBase *ix, int32_t *levels, int64_t *offsets, int32_t *neighbors) {
	if (ix->kind != MVS_KIND_HNSW)
		return false;
	static_cast<HNSWIndex *>(ix)->get_graph(levels, offsets, neighbors);
	return true;
}

} // namespace mvs
