// csrc/ivf_scan.hip -- IVFFlat inverted-list scan (IVFFlatScanner::scan_codes of faiss/IndexIVFFlat.cpp), list-major.
//
// One workgroup (256 threads) per work item = (one inverted list, <= 20 of the queries that probe it), built by
// csrc/ivf.hip.  The list is streamed from HBM once per item:
//   * thread <-> row: a thread reads its own row straight from global memory, 16 floats (4 x dwordx4) at a time with
//     the next chunk in flight -- no LDS staging, no workgroup barrier in the scan loop; the 64 rows of a wave span
//     64 cache lines per load instruction and the following instructions hit the same lines in L1;
//   * the query values are wave-uniform: they arrive through the scalar cache as s_load_dwordx16 and feed the VALU
//     as SGPR operands.  The item's queries are first re-packed (ivf_pack_item_queries_kernel) so that the values
//     of TWO query slots for one dimension are adjacent: one v_pk_add_f32 (SGPR pair - row value broadcast by
//     op_sel) + one v_pk_fma_f32 advance two distance chains by one dimension -- packed fp32 doubles the VALU rate
//     (157 vs 79 TFLOP/s), and each component is the same IEEE fma as the scalar instruction;
//   * per-pair arithmetic in k order (fvec_L2sqr / fvec_inner_product restated as one fma chain per pair), so list
//     scans are bit-identical to oracle/orc_core.c ivf_search;
//   * 20 independent chains per thread give the ILP; ~80 VGPRs => 6 waves per SIMD hide the scalar-load latency;
//   * k-best: per-wave lists in LDS with a wave-cooperative insert (rows of a list arrive in ascending position, so
//     equal distances never displace), cross-item threshold sharing through the class slots of flat_mfma.hip;
//     one partial list per (item, query slot) goes to merge_items_kernel.
// HBM-bound by design (algorithmic bytes = list bytes per item); VALU work is 2*d ops per (row, query).
#include "common.h"

#include "../../include/mi355_faiss.h"

namespace mvs {

namespace {

constexpr int SQG = 20;    // query slots per work item
constexpr int STILE = 256; // rows per tile = threads per workgroup

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) const f32x16 cf32x16;

struct ScanArgs {
	const float *xq; // [nq][dp]
	const float *xi; // [item][SQG/2][dp/16][2][8][2]: per item, slot pairs interleaved per dimension
	const float *rows; // [n][dp], lists are contiguous row segments
	float *pd;
	int32_t *pi;
	long long n;
	int k, dp;
	SelectorDev sel;
	const long long *idmap;
	const int *nitems_dev;   // device-side item count (grid is an upper bound); null = every block is an item
	const int4 *items;       // {row_begin, row_end, qoff, nq_item}
	const int *qidx;         // query number of slot qoff + s
	const long long *rowids; // stored id of every row
	unsigned *gslot;         // [nq][slot_stride] shared threshold classes
	int slot_stride;
	// grid mode (items == null): the Flat per-pair path.  Block b = (row split b / ngroups, query group b % ngroups);
	// partial lists [nsplit][nq][k]; rows may be pair-interleaved (FlatGeom::pair_interleaved)
	int ngroups, nq;
	long long split_rows;
	// tiles between two refreshes of the shared bounds: every workgroup that scans for the same queries reads (and
	// publishes to) the same few cache lines, and with ~1000 row splits of ONE query group those agent-scope
	// accesses serialise in L2 (nq = 16, N = 10M: 6.6 ms with a refresh per tile)
	int refresh_every;
};

__device__ __forceinline__ bool sel_member_scan(const SelectorDev &s, long long id) {
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
__device__ __forceinline__ unsigned s_f2key(float f) {
	const unsigned b = __float_as_uint(f);
	return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float s_key2f(unsigned k) {
	return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
template <bool IS_L2>
__device__ __forceinline__ unsigned s_bkey(float v) { // "smaller is better"
	return IS_L2 ? s_f2key(v) : ~s_f2key(v);
}
template <bool IS_L2>
__device__ __forceinline__ bool s_lex_better(float v, int id, float tv, int tid) {
	if (IS_L2)
		return v < tv || (v == tv && id < tid);
	return v > tv || (v == tv && id < tid);
}
template <bool IS_L2>
__device__ __forceinline__ bool s_lex_worse(float v, int id, float tv, int tid) {
	if (IS_L2)
		return v > tv || (v == tv && id > tid);
	return v < tv || (v == tv && id > tid);
}
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

template <bool IS_L2, bool INTERLEAVED>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const ScanArgs a) {
	extern __shared__ __attribute__((aligned(16))) float smem[];
	if (a.nitems_dev && (int)blockIdx.x >= *a.nitems_dev)
		return;
	const int k = a.k;
	float *lv = smem;                       // [4][SQG][k] per-wave list values
	int *lid = (int *)(lv + 4 * SQG * k);   // [4][SQG][k] row positions
	float *wv = (float *)(lid + 4 * SQG * k); // [4][SQG] current worst of each list
	int *wid = (int *)(wv + 4 * SQG);
	int *wpos = wid + 4 * SQG;
	float *gb = (float *)(wpos + 4 * SQG);  // [4][SQG] shared bound per query slot (one copy per wave)
	int *qn_s = (int *)(gb + 4 * SQG);      // [SQG] query numbers (for lane-indexed access)

	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	long long r_begin, r_end;
	int qbase, nq_item, split = 0, xblock;
	if (a.items) {
		const int4 it = a.items[blockIdx.x];
		r_begin = it.x;
		r_end = it.y;
		qbase = it.z;
		nq_item = it.w;
		xblock = blockIdx.x;
	} else {
		split = blockIdx.x / a.ngroups;
		xblock = blockIdx.x - split * a.ngroups;
		r_begin = (long long)split * a.split_rows;
		r_end = r_begin + a.split_rows;
		qbase = xblock * SQG;
		nq_item = a.nq - qbase < SQG ? a.nq - qbase : SQG;
	}
	if (r_end > a.n)
		r_end = a.n;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;

	// wave-uniform query numbers (SGPRs); unused slots alias slot 0 and are never computed
	int qn[SQG];
#pragma unroll
	for (int qq = 0; qq < SQG; ++qq)
		qn[qq] = a.qidx ? __builtin_amdgcn_readfirstlane(a.qidx[qbase + (qq < nq_item ? qq : 0)])
		                : qbase + (qq < nq_item ? qq : 0);
	if (tid < SQG)
		qn_s[tid] = a.qidx ? a.qidx[qbase + (tid < nq_item ? tid : 0)] : qbase + (tid < nq_item ? tid : 0);
	for (int i = lane; i < SQG * k; i += 64) {
		lv[wave * SQG * k + i] = neutral;
		lid[wave * SQG * k + i] = -1;
	}
	if (lane < SQG) {
		wv[wave * SQG + lane] = neutral;
		wid[wave * SQG + lane] = -1;
		wpos[wave * SQG + lane] = 0;
	}
	__syncthreads();
	gb += wave * SQG;

	// gb[qq] = max over the k class slots of the query: a valid bound on its final k-th value (flat_mfma.hip
	// "threshold sharing").  All loads of a pass are issued before any is consumed.
	// The loads are issued at the top of a tile and consumed right before its k-best update, so their (agent-scope,
	// L2-miss) latency hides behind the tile's distance chains.  Slot rows wider than 16 classes take the slow loop.
	const bool fast_slots = a.gslot && a.slot_stride == 16;
	unsigned m[SQG / 4];
	auto refresh_issue = [&]() {
		if (fast_slots) {
#pragma unroll
			for (int p = 0; p < SQG / 4; ++p) { // 4 queries x 16 classes per pass
				const int qq = p * 4 + (lane >> 4);
				m[p] = qq < nq_item ? __hip_atomic_load(a.gslot + (size_t)qn_s[qq] * 16 + (lane & 15), __ATOMIC_RELAXED,
				                                        __HIP_MEMORY_SCOPE_AGENT)
				                    : 0u;
			}
		}
	};
	auto refresh_finish = [&]() {
		const unsigned nk = s_bkey<IS_L2>(neutral);
		if (!a.gslot) {
			if (lane < SQG)
				gb[lane] = neutral;
		} else if (fast_slots) {
#pragma unroll
			for (int p = 0; p < SQG / 4; ++p) {
				unsigned x = m[p];
#pragma unroll
				for (int off = 8; off >= 1; off >>= 1) {
					const unsigned o = (unsigned)__shfl_xor((int)x, off);
					x = o > x ? o : x;
				}
				if ((lane & 15) == 0) {
					const unsigned c = x < nk ? x : nk;
					gb[p * 4 + (lane >> 4)] = IS_L2 ? s_key2f(c) : s_key2f(~c);
				}
			}
		} else {
			for (int qq = 0; qq < nq_item; ++qq) {
				unsigned x = 0u;
				const unsigned *sl = a.gslot + (size_t)qn_s[qq] * a.slot_stride;
				for (int j = lane; j < a.slot_stride; j += 64) {
					const unsigned v = __hip_atomic_load(sl + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					x = v > x ? v : x;
				}
#pragma unroll
				for (int off = 32; off >= 1; off >>= 1) {
					const unsigned o = (unsigned)__shfl_xor((int)x, off);
					x = o > x ? o : x;
				}
				if (lane == 0) {
					const unsigned c = x < nk ? x : nk;
					gb[qq] = IS_L2 ? s_key2f(c) : s_key2f(~c);
				}
			}
		}
		wave_sync();
	};

	const int nchunk = a.dp / 16;
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin + STILE - 1) / STILE) : 0;
	for (int tile = 0; tile < ntiles; ++tile) {
		const long long row0 = r_begin + (long long)tile * STILE;
		const long long row = row0 + tid;
		const long long gr = row < a.n ? row : a.n - 1;
		const float4 *yp = reinterpret_cast<const float4 *>(a.rows + (size_t)gr * a.dp);
		float4 yc[4];
#pragma unroll
		for (int i = 0; i < 4; ++i)
			yc[i] = yp[i];
		const bool refresh = tile % a.refresh_every == 0;
		if (refresh)
			refresh_issue();
		f32x2 acc2[SQG / 2];
#pragma unroll
		for (int p = 0; p < SQG / 2; ++p)
			acc2[p] = (f32x2) {0.f, 0.f};
		const float *xitem = a.xi + (size_t)xblock * (SQG / 2) * nchunk * 32;
		const bool hi16 = (row >> 4) & 1; // pair-interleaved storage: [k0,k2,k1,k3] / [k1,k3,k0,k2] by bit 4 of the row
		for (int c = 0; c < nchunk; ++c) {
			float4 yn[4];
			const int cn = c + 1 < nchunk ? c + 1 : c;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				yn[i] = yp[cn * 4 + i];
			float y[16];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				if (INTERLEAVED) { // back to natural k order, once per 16 values (amortised over the 20 chains)
					y[4 * i + 0] = hi16 ? yc[i].z : yc[i].x;
					y[4 * i + 1] = hi16 ? yc[i].x : yc[i].z;
					y[4 * i + 2] = hi16 ? yc[i].w : yc[i].y;
					y[4 * i + 3] = hi16 ? yc[i].y : yc[i].w;
				} else {
					y[4 * i + 0] = yc[i].x;
					y[4 * i + 1] = yc[i].y;
					y[4 * i + 2] = yc[i].z;
					y[4 * i + 3] = yc[i].w;
				}
			}
			// one basic block per slot pair: 2 x s_load_dwordx16 (16 dimensions x 2 slots), then per dimension one
			// v_pk_add_f32 + one v_pk_fma_f32 (L2) / one v_pk_fma_f32 (IP).  The scalar-load latency is covered by
			// the other waves of the SIMD.  (Measured alternative: the packed block in LDS, read with wave-uniform
			// ds_read_b128 -- a broadcast read still costs the full 64-lane LDS time, the kernel ran 1.9x slower.)
#pragma unroll
			for (int p = 0; p < SQG / 2; ++p) {
				if (2 * p < nq_item) { // wave-uniform
					const float *xb = xitem + ((size_t)p * nchunk + c) * 32;
					const f32x16 x0 = *(cf32x16 *)xb;
					const f32x16 x1 = *(cf32x16 *)(xb + 16);
					f32x2 s2 = acc2[p];
#pragma unroll
					for (int kk = 0; kk < 16; ++kk) {
						const f32x2 xx = kk < 8 ? (f32x2) {x0[2 * kk], x0[2 * kk + 1]}
						                        : (f32x2) {x1[2 * (kk - 8)], x1[2 * (kk - 8) + 1]};
						const f32x2 yy = {y[kk], y[kk]};
						if (IS_L2) {
							const f32x2 t = xx - yy;
							s2 = __builtin_elementwise_fma(t, t, s2);
						} else {
							s2 = __builtin_elementwise_fma(xx, yy, s2);
						}
					}
					acc2[p] = s2;
				}
			}
#pragma unroll
			for (int i = 0; i < 4; ++i)
				yc[i] = yn[i];
		}
		float acc[SQG];
#pragma unroll
		for (int qq = 0; qq < SQG; ++qq)
			acc[qq] = (qq & 1) ? acc2[qq / 2].y : acc2[qq / 2].x;

		// ---- k-best update of the tile
		if (refresh)
			refresh_finish();
		bool valid = row < r_end;
		if (valid && a.sel.kind != MVS_SEL_NONE) {
			const long long lab = a.rowids ? a.rowids[row] : row;
			valid = sel_member_scan(a.sel, a.idmap ? a.idmap[lab] : lab);
		}
#pragma unroll
		for (int qq = 0; qq < SQG; ++qq) {
			if (qq >= nq_item)
				continue;
			const float v = acc[qq];
			const int slot = wave * SQG + qq;
			float tv = wv[slot];
			const float gbv = gb[qq];
			// rows arrive in ascending position, so an equal value never beats the stored worst
			const bool pass = valid && (IS_L2 ? (v < tv && v <= gbv) : (v > tv && v >= gbv));
			unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
			if (mask == 0ull)
				continue;
			int tpos = wpos[slot];
			float *mv = lv + slot * k;
			int *mi = lid + slot * k;
			while (mask) {
				const int l = __builtin_ctzll(mask);
				mask &= mask - 1;
				const float cv = __shfl(v, l);
				const int id = (int)(row0 + (tid - lane) + l);
				if (IS_L2 ? cv < tv : cv > tv) {
					if (lane == 0) {
						mv[tpos] = cv;
						mi[tpos] = id;
						if (a.gslot) // publish the best value of this row's class (fire and forget)
							__hip_atomic_fetch_min(a.gslot + (size_t)qn[qq] * a.slot_stride + (unsigned)id % (unsigned)k,
							                       s_bkey<IS_L2>(cv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					wave_sync();
					float bv = 0.f;
					int bi = 0, bp = -1;
					for (int j = lane; j < k; j += 64) {
						const float x = mv[j];
						const int xi = mi[j];
						if (bp < 0 || s_lex_worse<IS_L2>(x, xi, bv, bi)) {
							bv = x;
							bi = xi;
							bp = j;
						}
					}
#pragma unroll
					for (int off = 32; off >= 1; off >>= 1) {
						const float ov = __shfl_xor(bv, off);
						const int oi = __shfl_xor(bi, off);
						const int op = __shfl_xor(bp, off);
						if (op >= 0 && (bp < 0 || s_lex_worse<IS_L2>(ov, oi, bv, bi))) {
							bv = ov;
							bi = oi;
							bp = op;
						}
					}
					tv = bv;
					tpos = bp;
					if (lane == 0) {
						wv[slot] = bv;
						wid[slot] = bi;
						wpos[slot] = bp;
					}
				}
			}
			wave_sync();
		}
	}

	// ---- merge the 4 per-wave lists: ONE partial list per (item, query slot), [item][SQG][k]
	__syncthreads();
	for (int qq = wave; qq < nq_item; qq += 4) {
		const size_t base = a.items ? ((size_t)blockIdx.x * SQG + qq) * k : ((size_t)split * a.nq + (qbase + qq)) * k;
		for (int r = 0; r < k; ++r) {
			float bv = 0.f;
			int bi = 0x7fffffff, bp = -1;
			for (int c = lane; c < 4 * k; c += 64) {
				const int w2 = c / k, j = c - w2 * k, p2 = (w2 * SQG + qq) * k + j;
				const float v = lv[p2];
				const int id = lid[p2];
				if (id < 0)
					continue;
				if (bp < 0 || s_lex_better<IS_L2>(v, id, bv, bi)) {
					bv = v;
					bi = id;
					bp = p2;
				}
			}
#pragma unroll
			for (int off = 32; off >= 1; off >>= 1) {
				const float ov = __shfl_xor(bv, off);
				const int oi = __shfl_xor(bi, off);
				const int op = __shfl_xor(bp, off);
				if (op >= 0 && (bp < 0 || s_lex_better<IS_L2>(ov, oi, bv, bi))) {
					bv = ov;
					bi = oi;
					bp = op;
				}
			}
			if (lane == 0) {
				a.pd[base + r] = bp >= 0 ? bv : neutral;
				a.pi[base + r] = bp >= 0 ? bi : -1;
				if (bp >= 0)
					lid[bp] = -1; // consumed
			}
			wave_sync();
		}
	}
}

// xi[item][p][c][h][kk][j] = xq[qidx[qoff + 2p + j]][16c + 8h + kk]   (0 for unused slots)
__global__ void ivf_pack_item_queries_kernel(const float *xq, const int *qidx, const int4 *items, const int *nitems_dev,
                                             int dp, int nq, float *xi) {
	if (nitems_dev && (int)blockIdx.x >= *nitems_dev)
		return;
	int4 it;
	if (items) {
		it = items[blockIdx.x];
	} else { // grid mode: block = query group, slots = consecutive queries; nitems_dev is unused, it.w from nq
		it.z = blockIdx.x * SQG;
		it.w = nq - it.z < SQG ? nq - it.z : SQG;
	}
	const int nchunk = dp / 16;
	const int total = (SQG / 2) * nchunk * 32;
	float *dst = xi + (size_t)blockIdx.x * total;
	for (int e = threadIdx.x; e < total; e += blockDim.x) {
		const int j = e & 1, kk = (e >> 1) & 7, h = (e >> 4) & 1, pc = e >> 5;
		const int c = pc % nchunk, p = pc / nchunk;
		const int slot = 2 * p + j;
		float v = 0.f;
		if (slot < it.w)
			v = xq[(size_t)(qidx ? qidx[it.z + slot] : it.z + slot) * dp + c * 16 + h * 8 + kk];
		dst[e] = v;
	}
}

// ---- device-side grouping of the (query, probed list) pairs into work items (no host round trip) ---------------
// keys[q][p] = list probed by query q at rank p (-1: fewer than nprobe lists).  Any order of the pairs inside a list
// gives the same results (the per-query merge is order independent), so positions come from atomics.
// (key_stride > 1: pair i reads keys[i * key_stride] -- the nearest-list pre-pass groups column 0 of the [nq][nprobe] labels as a
// batch with ONE probe per query instead of masking the other columns and walking all nq * nprobe pairs)
__global__ void ivf_group_count_kernel(const long long *keys, int npairs, int *cnt, int key_stride) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npairs)
		return;
	const long long l = keys[(size_t)i * key_stride];
	if (l >= 0)
		atomicAdd(&cnt[l], 1);
}
// single workgroup: pair offsets and item offsets per list (exclusive scans), total item count
__global__ __launch_bounds__(1024) void ivf_group_scan_kernel(const int *cnt, int nlist, int G, int *pair_off,
                                                             int *item_off, int *cursor, int *nitems_out) {
	__shared__ int part_p[1024], part_i[1024];
	const int t = threadIdx.x;
	const int per = (nlist + 1023) / 1024;
	const int l0 = t * per, l1 = min(nlist, l0 + per);
	int sp = 0, si = 0;
	for (int l = l0; l < l1; ++l) {
		sp += cnt[l];
		si += (cnt[l] + G - 1) / G;
	}
	part_p[t] = sp;
	part_i[t] = si;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
		const int vp = t >= off ? part_p[t - off] : 0, vi = t >= off ? part_i[t - off] : 0;
		__syncthreads();
		part_p[t] += vp;
		part_i[t] += vi;
		__syncthreads();
	}
	int bp = part_p[t] - sp, bi = part_i[t] - si;
	for (int l = l0; l < l1; ++l) {
		pair_off[l] = bp;
		cursor[l] = bp;
		item_off[l] = bi;
		bp += cnt[l];
		bi += (cnt[l] + G - 1) / G;
	}
	if (t == 1023) {
		pair_off[nlist] = part_p[1023];
		item_off[nlist] = part_i[1023];
		*nitems_out = part_i[1023];
	}
}
__global__ void ivf_group_scatter_kernel(const long long *keys, int npairs, int nprobe, int G, int shift,
                                         const int *pair_off, const int *item_off, int *cursor, int *qidx, int *slots,
                                         int key_stride) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= npairs)
		return;
	const long long l = keys[(size_t)i * key_stride];
	if (l < 0) {
		if (slots)
			slots[i] = -1;
		return;
	}
	const int pos = atomicAdd(&cursor[l], 1);
	qidx[pos] = i / nprobe;
	if (slots) { // (the coarse-filter path merges per query from the candidate stream: it never reads the slot codes)
		const int rel = pos - pair_off[l];
		slots[i] = ((item_off[l] + rel / G) << shift) | (rel % G);
	}
}
// Round 4: the same two steps with a histogram per workgroup in LDS.  Clustered queries probe a few lists by the hundred or thousand
// (C3: 320 k pairs, 0.03 ms each for count and scatter -- the atomics on the hottest counters serialise in L2); a workgroup of 4 096
// pairs folds its pairs of one list into ONE global atomic (count) / one reservation of a run of positions (scatter).
constexpr int GROUP_CHUNK = 4096;
__global__ __launch_bounds__(1024) void ivf_group_count_lds_kernel(const long long *keys, int npairs, int *cnt, int key_stride, int nlist) {
	extern __shared__ int gh[]; // [nlist]
	for (int i = threadIdx.x; i < nlist; i += 1024)
		gh[i] = 0;
	__syncthreads();
	const int base = blockIdx.x * GROUP_CHUNK;
	for (int j = threadIdx.x; j < GROUP_CHUNK && base + j < npairs; j += 1024) {
		const long long l = keys[(size_t)(base + j) * key_stride];
		if (l >= 0)
			atomicAdd(&gh[l], 1);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nlist; i += 1024)
		if (gh[i])
			atomicAdd(&cnt[i], gh[i]);
}
__global__ __launch_bounds__(1024) void ivf_group_scatter_lds_kernel(const long long *keys, int npairs, int nprobe, int G, int shift,
                                                                     const int *pair_off, const int *item_off, int *cursor, int *qidx,
                                                                     int *slots, int key_stride, int nlist) {
	extern __shared__ int gh[]; // [nlist] pairs of this workgroup per list, then their running rank; [nlist] first position of the run
	int *gb = gh + nlist;
	for (int i = threadIdx.x; i < nlist; i += 1024)
		gh[i] = 0;
	__syncthreads();
	const int base = blockIdx.x * GROUP_CHUNK;
	long long mine[GROUP_CHUNK / 1024];
#pragma unroll
	for (int u = 0; u < GROUP_CHUNK / 1024; ++u) {
		const int j = threadIdx.x + 1024 * u;
		mine[u] = base + j < npairs ? keys[(size_t)(base + j) * key_stride] : -1;
		if (mine[u] >= 0)
			atomicAdd(&gh[mine[u]], 1);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nlist; i += 1024) {
		const int c = gh[i];
		if (c)
			gb[i] = atomicAdd(&cursor[i], c);
		gh[i] = 0;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < GROUP_CHUNK / 1024; ++u) {
		const int i = base + threadIdx.x + 1024 * u;
		if (i >= npairs)
			continue;
		const long long l = mine[u];
		if (l < 0) {
			if (slots)
				slots[i] = -1;
			continue;
		}
		const int pos = gb[l] + atomicAdd(&gh[l], 1);
		qidx[pos] = i / nprobe;
		if (slots) {
			const int rel = pos - pair_off[l];
			slots[i] = ((item_off[l] + rel / G) << shift) | (rel % G);
		}
	}
}
__global__ void ivf_group_items_kernel(const int *cnt, const int *pair_off, const int *item_off,
                                       const long long *list_begin, const long long *list_end, int nlist, int G,
                                       int4 *items) {
	const int l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlist)
		return;
	const int n = cnt[l];
	for (int g = 0; g * G < n; ++g)
		items[item_off[l] + g] = make_int4((int)list_begin[l], (int)list_end[l], pair_off[l] + g * G, min(G, n - g * G));
}

// ---- round 5: BOTH groupings of a coarse-filter search in three launches ------------------------------------------------------
// The IVF coarse filter (csrc/ivf_collect.hip) groups the batch twice: all nq x nprobe pairs for the main pass (set 1) and
// column 0 of the labels -- every query's nearest list -- for the publish-only pre-pass (set 0).  Rounds 3-4 ran the four
// kernels above once per set (eight launches of 5-11 us each, two memsets in front).  Here: one count kernel with two LDS
// histograms per workgroup, one scan kernel of two workgroups that also writes the work items, one scatter kernel for both
// sets -- which then ZEROES the counters for the next search (nobody reads them after the scan kernel).
struct Group2Set {
	int *cnt, *pair_off, *item_off, *cursor, *nitems; // ws ints of the set (launch_ivf_group's layout)
	int4 *items;
	int *qidx;
	int *slots; // slot codes per pair (may be null)
};
__global__ __launch_bounds__(1024) void ivf_group2_count_kernel(const long long *keys, int npairs, int nprobe, int nlist, int *cnt0,
                                                               int *cnt1, int use_lds) {
	extern __shared__ int gh[]; // [2][nlist]
	const int base = blockIdx.x * GROUP_CHUNK;
	if (!use_lds) {
		for (int j = threadIdx.x; j < GROUP_CHUNK && base + j < npairs; j += 1024) {
			const long long l = keys[base + j];
			if (l >= 0) {
				atomicAdd(&cnt1[l], 1);
				if ((base + j) % nprobe == 0)
					atomicAdd(&cnt0[l], 1);
			}
		}
		return;
	}
	for (int i = threadIdx.x; i < 2 * nlist; i += 1024)
		gh[i] = 0;
	__syncthreads();
	for (int j = threadIdx.x; j < GROUP_CHUNK && base + j < npairs; j += 1024) {
		const long long l = keys[base + j];
		if (l >= 0) {
			atomicAdd(&gh[nlist + l], 1);
			if ((base + j) % nprobe == 0)
				atomicAdd(&gh[l], 1);
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nlist; i += 1024) {
		if (gh[i])
			atomicAdd(&cnt0[i], gh[i]);
		if (gh[nlist + i])
			atomicAdd(&cnt1[i], gh[nlist + i]);
	}
}
// workgroup s scans set s (pair offsets, item offsets, cursors, item count) and writes the set's work items
__global__ __launch_bounds__(1024) void ivf_group2_scan_kernel(Group2Set s0, Group2Set s1, int nlist, int G, const long long *list_begin,
                                                              const long long *list_end) {
	const Group2Set s = blockIdx.x == 0 ? s0 : s1;
	__shared__ int part_p[1024], part_i[1024];
	const int t = threadIdx.x;
	const int per = (nlist + 1023) / 1024;
	const int l0 = t * per, l1 = min(nlist, l0 + per);
	int sp = 0, si = 0;
	for (int l = l0; l < l1; ++l) {
		sp += s.cnt[l];
		si += (s.cnt[l] + G - 1) / G;
	}
	part_p[t] = sp;
	part_i[t] = si;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
		const int vp = t >= off ? part_p[t - off] : 0, vi = t >= off ? part_i[t - off] : 0;
		__syncthreads();
		part_p[t] += vp;
		part_i[t] += vi;
		__syncthreads();
	}
	int bp = part_p[t] - sp, bi = part_i[t] - si;
	for (int l = l0; l < l1; ++l) {
		const int n = s.cnt[l];
		s.pair_off[l] = bp;
		s.cursor[l] = bp;
		s.item_off[l] = bi;
		for (int g = 0; g * G < n; ++g)
			s.items[bi + g] = make_int4((int)list_begin[l], (int)list_end[l], bp + g * G, min(G, n - g * G));
		bp += n;
		bi += (n + G - 1) / G;
	}
	if (t == 1023) {
		s.pair_off[nlist] = part_p[1023];
		s.item_off[nlist] = part_i[1023];
		*s.nitems = part_i[1023];
	}
}
__global__ __launch_bounds__(1024) void ivf_group2_scatter_kernel(const long long *keys, int npairs, int nprobe, int G, int shift,
                                                                 Group2Set s0, Group2Set s1, int nlist, int use_lds) {
	extern __shared__ int gh[]; // [2][nlist] pairs of this workgroup per (set, list), then their running rank; [2][nlist] first position of the run
	int *gb = gh + 2 * nlist;
	const int base = blockIdx.x * GROUP_CHUNK;
	long long mine[GROUP_CHUNK / 1024];
#pragma unroll
	for (int u = 0; u < GROUP_CHUNK / 1024; ++u) {
		const int j = threadIdx.x + 1024 * u;
		mine[u] = base + j < npairs ? keys[base + j] : -1;
	}
	if (use_lds) {
		for (int i = threadIdx.x; i < 2 * nlist; i += 1024)
			gh[i] = 0;
		__syncthreads();
#pragma unroll
		for (int u = 0; u < GROUP_CHUNK / 1024; ++u) {
			const int i = base + threadIdx.x + 1024 * u;
			if (mine[u] >= 0) {
				atomicAdd(&gh[nlist + mine[u]], 1);
				if (i % nprobe == 0)
					atomicAdd(&gh[mine[u]], 1);
			}
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nlist; i += 1024) {
			const int c0 = gh[i], c1 = gh[nlist + i];
			if (c0)
				gb[i] = atomicAdd(&s0.cursor[i], c0);
			if (c1)
				gb[nlist + i] = atomicAdd(&s1.cursor[i], c1);
			gh[i] = 0;
			gh[nlist + i] = 0;
		}
		__syncthreads();
	}
#pragma unroll
	for (int u = 0; u < GROUP_CHUNK / 1024; ++u) {
		const int i = base + threadIdx.x + 1024 * u;
		if (i >= npairs)
			continue;
		const long long l = mine[u];
		const bool first = i % nprobe == 0;
		if (l < 0) {
			if (s1.slots)
				s1.slots[i] = -1;
			if (first && s0.slots)
				s0.slots[i / nprobe] = -1;
			continue;
		}
		{
			const int pos = use_lds ? gb[nlist + l] + atomicAdd(&gh[nlist + l], 1) : atomicAdd(&s1.cursor[l], 1);
			s1.qidx[pos] = i / nprobe;
			if (s1.slots) {
				const int rel = pos - s1.pair_off[l];
				s1.slots[i] = ((s1.item_off[l] + rel / G) << shift) | (rel % G);
			}
		}
		if (first) {
			const int pos = use_lds ? gb[l] + atomicAdd(&gh[l], 1) : atomicAdd(&s0.cursor[l], 1);
			s0.qidx[pos] = i / nprobe;
			if (s0.slots) {
				const int rel = pos - s0.pair_off[l];
				s0.slots[i / nprobe] = ((s0.item_off[l] + rel / G) << shift) | (rel % G);
			}
		}
	}
	// the counters are not read after the scan kernel: zero for the next search (launch_ivf_group2's contract)
	for (int i = blockIdx.x * 1024 + threadIdx.x; i <= nlist; i += gridDim.x * 1024) {
		s0.cnt[i] = 0;
		s1.cnt[i] = 0;
	}
}

size_t scan_lds_bytes(int64_t k) {
	return (size_t)4 * SQG * k * 8 + (size_t)4 * SQG * 16 + SQG * 4 + 64;
}

} // namespace

bool ivf_scan_supported(int dp, int64_t k) {
	return dp % 16 == 0 && dp >= 16 && scan_lds_bytes(k) <= 150 * 1024;
}
size_t ivf_scan_lds_bytes(int64_t k) {
	return scan_lds_bytes(k);
}
static void launch_scan_kernel(int metric, bool interleaved, const ScanArgs &a, int grid, int64_t k, hipStream_t st) {
	const size_t lds = scan_lds_bytes(k);
#define MVS_SCAN(L2, IL)                                                                                               \
	{                                                                                                                  \
		auto kern = ivf_scan_kernel<L2, IL>;                                                                           \
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));        \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                                   \
	}
	if (metric == METRIC_IP) {
		if (interleaved)
			MVS_SCAN(false, true)
		else
			MVS_SCAN(false, false)
	} else {
		if (interleaved)
			MVS_SCAN(true, true)
		else
			MVS_SCAN(true, false)
	}
#undef MVS_SCAN
	MVS_HIP(hipGetLastError());
}

size_t ivf_scan_query_pack_bytes(int dp, int nitems) {
	return (size_t)nitems * SQG * dp * sizeof(float);
}

// ws_int: [4*(nlist+1) + 4] ints.  Outputs: d_items (<= max_items), d_qidx [npairs], d_slots [npairs], *d_nitems.
// group = query slots per item (20 for the per-pair scan, 128 for the MFMA variant); slot code = item << shift | slot.
int ivf_group_max_items(int64_t npairs, int64_t nlist, int group) {
	return (int)(npairs / group + nlist + 1);
}
size_t ivf_group_ws_ints(int64_t nlist) {
	return (size_t)4 * (nlist + 1) + 4;
}
void launch_ivf_group(const int64_t *d_keys, int64_t nq, int nprobe, int64_t nlist, int group, int shift,
                      const int64_t *d_list_begin, const int64_t *d_list_end, int *ws_int, void *d_items, int *d_qidx,
                      int *d_slots, int **d_nitems_out, int **d_cnt_out, hipStream_t st, int key_stride, bool counters_zeroed) {
	const int npairs = (int)(nq * nprobe);
	int *cnt = ws_int, *pair_off = cnt + (nlist + 1), *item_off = pair_off + (nlist + 1), *cursor = item_off + (nlist + 1);
	int *nitems = cursor + (nlist + 1);
	if (!counters_zeroed)
		MVS_HIP(hipMemsetAsync(cnt, 0, (size_t)(nlist + 1) * sizeof(int), st));
	const bool lds_hist = nlist <= 8192 && npairs >= 4 * GROUP_CHUNK;
	const unsigned chunks = (unsigned)((npairs + GROUP_CHUNK - 1) / GROUP_CHUNK);
	if (lds_hist)
		hipLaunchKernelGGL(ivf_group_count_lds_kernel, dim3(chunks), dim3(1024), (size_t)nlist * sizeof(int), st,
		                   (const long long *)d_keys, npairs, cnt, key_stride, (int)nlist);
	else
		hipLaunchKernelGGL(ivf_group_count_kernel, dim3((npairs + 255) / 256), dim3(256), 0, st, (const long long *)d_keys,
		                   npairs, cnt, key_stride);
	hipLaunchKernelGGL(ivf_group_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, (int)nlist, group, pair_off, item_off,
	                   cursor, nitems);
	if (lds_hist) {
		ensure_dynamic_lds((const void *)ivf_group_scatter_lds_kernel, (size_t)2 * nlist * sizeof(int));
		hipLaunchKernelGGL(ivf_group_scatter_lds_kernel, dim3(chunks), dim3(1024), (size_t)2 * nlist * sizeof(int), st,
		                   (const long long *)d_keys, npairs, nprobe, group, shift, pair_off, item_off, cursor, d_qidx, d_slots,
		                   key_stride, (int)nlist);
	} else
		hipLaunchKernelGGL(ivf_group_scatter_kernel, dim3((npairs + 255) / 256), dim3(256), 0, st, (const long long *)d_keys,
		                   npairs, nprobe, group, shift, pair_off, item_off, cursor, d_qidx, d_slots, key_stride);
	hipLaunchKernelGGL(ivf_group_items_kernel, dim3((unsigned)((nlist + 255) / 256)), dim3(256), 0, st, cnt, pair_off,
	                   item_off, (const long long *)d_list_begin, (const long long *)d_list_end, (int)nlist, group,
	                   (int4 *)d_items);
	MVS_HIP(hipGetLastError());
	*d_nitems_out = nitems;
	*d_cnt_out = cnt;
}

// Both groupings of one coarse-filter search (see ivf_group2_count_kernel).  ws0 / ws1: ivf_group_ws_ints(nlist) ints each, whose
// COUNTERS (the first nlist + 1 ints) are zero on entry and zero again on exit; set 0 = one probe per query (column 0 of the
// [nq][nprobe] labels), set 1 = all pairs.
void launch_ivf_group2(const int64_t *d_keys, int64_t nq, int nprobe, int64_t nlist, int group, int shift, const int64_t *d_list_begin,
                       const int64_t *d_list_end, int *ws0, int *ws1, void *d_items0, int *d_qidx0, int *d_slots0, void *d_items1,
                       int *d_qidx1, int *d_slots1, int **d_nitems0_out, int **d_nitems1_out, hipStream_t st) {
	const int npairs = (int)(nq * nprobe);
	auto mk = [&](int *ws, void *items, int *qidx, int *slots) {
		Group2Set s;
		s.cnt = ws, s.pair_off = s.cnt + (nlist + 1), s.item_off = s.pair_off + (nlist + 1), s.cursor = s.item_off + (nlist + 1);
		s.nitems = s.cursor + (nlist + 1);
		s.items = (int4 *)items, s.qidx = qidx, s.slots = slots;
		return s;
	};
	const Group2Set s0 = mk(ws0, d_items0, d_qidx0, d_slots0), s1 = mk(ws1, d_items1, d_qidx1, d_slots1);
	const int use_lds = nlist <= 8192 && npairs >= 4 * GROUP_CHUNK;
	const unsigned chunks = (unsigned)((npairs + GROUP_CHUNK - 1) / GROUP_CHUNK);
	const size_t lds_c = use_lds ? (size_t)2 * nlist * sizeof(int) : 0, lds_s = use_lds ? (size_t)4 * nlist * sizeof(int) : 0;
	if (use_lds) {
		ensure_dynamic_lds((const void *)ivf_group2_count_kernel, lds_c);
		ensure_dynamic_lds((const void *)ivf_group2_scatter_kernel, lds_s);
	}
	hipLaunchKernelGGL(ivf_group2_count_kernel, dim3(chunks), dim3(1024), lds_c, st, (const long long *)d_keys, npairs, nprobe, (int)nlist,
	                   s0.cnt, s1.cnt, use_lds);
	hipLaunchKernelGGL(ivf_group2_scan_kernel, dim3(2), dim3(1024), 0, st, s0, s1, (int)nlist, group, (const long long *)d_list_begin,
	                   (const long long *)d_list_end);
	hipLaunchKernelGGL(ivf_group2_scatter_kernel, dim3(chunks), dim3(1024), lds_s, st, (const long long *)d_keys, npairs, nprobe, group,
	                   shift, s0, s1, (int)nlist, use_lds);
	MVS_HIP(hipGetLastError());
	*d_nitems0_out = s0.nitems;
	*d_nitems1_out = s1.nitems;
}

// the items' queries in MFMA B-fragment order (util_kernels.hip pack_queries_kernel), gathered through qidx:
// qf[((((item*4 + w)*nch + ch)*(kc/8) + s4)*64 + lane)*4 + e] = x[qidx[qoff + 32w + (lane&31)]][ch*kc + 2*(4*s4+e) + (lane>>5)]
__global__ void ivf_pack_item_fragments_kernel(const float *x, int d, int kc, int nch, const int4 *items,
                                               const int *nitems_dev, const int *qidx, float *qf) {
	if ((int)blockIdx.x >= *nitems_dev)
		return;
	const int4 it = items[blockIdx.x];
	const int ks4 = kc / 8;
	const int per_item4 = 4 * nch * ks4 * 64;
	float4 *dst = reinterpret_cast<float4 *>(qf) + (size_t)blockIdx.x * per_item4;
	for (int i = threadIdx.x; i < per_item4; i += blockDim.x) {
		const int lane = i & 63;
		int t = i >> 6;
		const int s4 = t % ks4;
		t /= ks4;
		const int ch = t % nch, w = t / nch;
		const int slot = w * 32 + (lane & 31);
		float o[4] = {0.f, 0.f, 0.f, 0.f};
		if (slot < it.w) {
			const float *xr = x + (size_t)qidx[it.z + slot] * d;
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int kk = ch * kc + 2 * (4 * s4 + e) + (lane >> 5);
				o[e] = kk < d ? xr[kk] : 0.f;
			}
		}
		dst[i] = make_float4(o[0], o[1], o[2], o[3]);
	}
}
// L2 on RESIDUAL rows (csrc/ivf.hip build_lists_mf): the item's queries minus the centroid of the item's list
__global__ void ivf_pack_item_fragments_res_kernel(const float *x, int d, int kc, int nch, const int4 *items,
                                                   const int *nitems_dev, const int *qidx, float *qf, const float *cent,
                                                   const int *list_of_blk64, float *item_qn, unsigned *qmaxn_bits) {
	if ((int)blockIdx.x >= *nitems_dev)
		return;
	const int4 it = items[blockIdx.x];
	const float *c = cent + (size_t)list_of_blk64[it.x >> 6] * d; // every list starts at a multiple of 64 rows
	const int ks4 = kc / 8;
	const int per_item4 = 4 * nch * ks4 * 64;
	float4 *dst = reinterpret_cast<float4 *>(qf) + (size_t)blockIdx.x * per_item4;
	for (int i = threadIdx.x; i < per_item4; i += blockDim.x) {
		const int lane = i & 63;
		int t = i >> 6;
		const int s4 = t % ks4;
		t /= ks4;
		const int ch = t % nch, w = t / nch;
		const int slot = w * 32 + (lane & 31);
		float o[4] = {0.f, 0.f, 0.f, 0.f};
		if (slot < it.w) {
			const float *xr = x + (size_t)qidx[it.z + slot] * d;
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int kk = ch * kc + 2 * (4 * s4 + e) + (lane >> 5);
				o[e] = kk < d ? __fsub_rn(xr[kk], c[kk]) : 0.f;
			}
		}
		dst[i] = make_float4(o[0], o[1], o[2], o[3]);
	}
	for (int slot = threadIdx.x; slot < 128; slot += blockDim.x) {
		float acc = 0.f;
		if (slot < it.w) {
			const int q = qidx[it.z + slot];
			const float *xr = x + (size_t)q * d;
			for (int kk = 0; kk < d; ++kk) {
				const float r = __fsub_rn(xr[kk], c[kk]);
				acc = fmaf(r, r, acc);
			}
			atomicMax(qmaxn_bits + q, __float_as_uint(acc));
		}
		item_qn[(size_t)blockIdx.x * 128 + slot] = acc;
	}
}
void launch_ivf_pack_item_fragments_residual(const float *d_x, int d, int kc, int nch, const void *d_items,
                                             const int *d_nitems, int max_items, const int *d_qidx, float *d_qf,
                                             const float *d_centroids, const int *d_list_of_blk64, float *d_item_qn,
                                             unsigned *d_qmaxn_bits, hipStream_t st) {
	if (max_items <= 0)
		return;
	hipLaunchKernelGGL(ivf_pack_item_fragments_res_kernel, dim3(max_items), dim3(256), 0, st, d_x, d, kc, nch,
	                   (const int4 *)d_items, d_nitems, d_qidx, d_qf, d_centroids, d_list_of_blk64, d_item_qn, d_qmaxn_bits);
	MVS_HIP(hipGetLastError());
}

void launch_ivf_pack_item_fragments(const float *d_x, int d, int kc, int nch, const void *d_items, const int *d_nitems,
                                    int max_items, const int *d_qidx, float *d_qf, hipStream_t st) {
	if (max_items <= 0)
		return;
	hipLaunchKernelGGL(ivf_pack_item_fragments_kernel, dim3(max_items), dim3(256), 0, st, d_x, d, kc, nch,
	                   (const int4 *)d_items, d_nitems, d_qidx, d_qf);
	MVS_HIP(hipGetLastError());
}

void launch_ivf_scan(int dp, int metric, const float *d_xq, const float *d_rows, int64_t nrows, const int64_t *d_rowids,
                     int64_t k, const void *d_items, int nitems, const int *d_qidx, SelectorDev sel,
                     const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, float *d_xi,
                     const int *d_nitems, hipStream_t st) {
	if (nitems <= 0)
		return;
	hipLaunchKernelGGL(ivf_pack_item_queries_kernel, dim3(nitems), dim3(256), 0, st, d_xq, d_qidx, (const int4 *)d_items,
	                   d_nitems, dp, 0, d_xi);
	MVS_HIP(hipGetLastError());
	ScanArgs a;
	a.xq = d_xq;
	a.xi = d_xi;
	a.rows = d_rows;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = nrows;
	a.k = (int)k;
	a.dp = dp;
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.items = (const int4 *)d_items;
	a.nitems_dev = d_nitems;
	a.qidx = d_qidx;
	a.rowids = (const long long *)d_rowids;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	a.ngroups = 0;
	a.nq = 0;
	a.split_rows = 0;
	a.refresh_every = 1;
	launch_scan_kernel(metric, false, a, nitems, k, st);
}

// Flat per-pair path (nq < 20 or a selector; IndexFlat::search -> exhaustive_*_seq): regular grid over
// (row split, group of 20 queries).  d_xi: ceil(nq/20) packed query groups.
void launch_pair_scan(int dp, bool interleaved, int metric, const float *d_xq, int64_t nq, const float *d_rows,
                      int64_t nrows, int64_t k, int nsplit, int64_t split_rows, SelectorDev sel, const int64_t *d_idmap,
                      float *d_pd, int32_t *d_pi, unsigned *d_gslot, float *d_xi, hipStream_t st) {
	if (nq <= 0 || nrows <= 0)
		return;
	const int ngroups = (int)((nq + SQG - 1) / SQG);
	hipLaunchKernelGGL(ivf_pack_item_queries_kernel, dim3(ngroups), dim3(256), 0, st, d_xq, (const int *)nullptr,
	                   (const int4 *)nullptr, (const int *)nullptr, dp, (int)nq, d_xi);
	MVS_HIP(hipGetLastError());
	ScanArgs a;
	a.xq = d_xq;
	a.xi = d_xi;
	a.rows = d_rows;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = nrows;
	a.k = (int)k;
	a.dp = dp;
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.items = nullptr;
	a.nitems_dev = nullptr;
	a.qidx = nullptr;
	a.rowids = nullptr;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	a.ngroups = ngroups;
	a.nq = (int)nq;
	a.split_rows = split_rows;
	a.refresh_every = nsplit >= 64 ? nsplit / 32 : 1;
	launch_scan_kernel(metric, interleaved, a, nsplit * ngroups, k, st);
}

} // namespace mvs
