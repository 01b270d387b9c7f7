// csrc/flat_direct.hip -- per-pair brute-force search (HBM-bound streaming kernel).
//
// Replaces what the reference reaches through entry.index->search(...)
// (/root/reference/src/faiss_extension.cpp:631) when FAISS takes its NON-BLAS branch
// [UPSTREAM faiss/utils/distances.cpp exhaustive_L2sqr_seq / exhaustive_inner_product_seq]:
//   * nq < 20 (distance_compute_blas_threshold), e.g. the 1-query batches of the reference's Go harness
//     (go/benches_c.go:143-161), and
//   * any search with an IDSelector (faiss_search_filter / faiss_search_filter_set,
//     src/faiss_extension.cpp:959,1008): `if (!sel->is_member(j)) continue;`
// Arithmetic (bit-exact with oracle search_pair): L2 = k-ordered chain of fmaf(t,t,acc), t = x[k]-y[k];
// IP = k-ordered chain of fmaf(x[k],y[k],acc).  MODE_L2_FORMULA reproduces the BLAS-branch value
// (xn + yn) - 2 ip, used when k exceeds what the fused MFMA kernel's LDS lists can hold.
//
// Mapping: thread <-> database row (256-row tiles), queries are wave-uniform (scalar loads), the tile is
// staged through LDS only to turn coalesced HBM reads into per-row register vectors.  Each wave keeps a
// k-entry list per query in LDS, distributed over its lanes; insertion is wave-cooperative.
#include "common.h"

#include "../../include/mi355_faiss.h"

#include <cstring>

namespace mvs {

constexpr int MODE_IP = 0, MODE_L2_PAIR = 1, MODE_L2_FORMULA = 2;
// The other metrics the glue registers (src/faiss_extension.cpp:58-68) -> faiss::knn_extra_metrics: always per pair,
// VectorDistance<mt> of faiss/utils/extra_distances-inl.h as plain sequential float loops (oracle/orc_core.c
// extra_distance), one strict-insert heap per query; Jaccard is a similarity (kept: the largest), the others distances.
constexpr int MODE_L1 = 3, MODE_LINF = 4, MODE_LP = 5, MODE_CANBERRA = 6, MODE_BRAYCURTIS = 7, MODE_JS = 8, MODE_JACCARD = 9;
__host__ __device__ constexpr bool mode_two_sums(int m) {
	return m == MODE_BRAYCURTIS || m == MODE_JACCARD;
}
__host__ __device__ constexpr bool mode_needs_dim_guard(int m) { // zero padding is not neutral: 0^0, 0/0, 0 log(0/0)
	return m == MODE_LP || m == MODE_CANBERRA || m == MODE_JS;
}
constexpr int DTILE = 256;

struct DirectArgs {
	const float *xq; // [nq_pad][dp], rows >= nq are zero
	const float *xn; // query norms (formula mode)
	const float *yb;
	const float *yn;
	float *pd;
	int32_t *pi;
	long long n, split_rows;
	int nq, k, dp, nsplit, ngroups, interleaved;
	int d;            // logical dimension (extra metrics: padded dimensions are skipped)
	float metric_arg; // Lp exponent
	SelectorDev sel;
	const long long *idmap;
	// item mode (IVF list scan, csrc/ivf.hip): one workgroup per (row segment, <= QG queries) work item
	const int4 *items;        // {row_begin, row_end, qoff, nq_item}; null = regular (split x query-group) grid
	const int *qidx;          // query numbers of the items, indexed by qoff + slot
	const long long *rowids;  // stored id of every row (selector tests it); null = the row index
	// cross-workgroup threshold sharing (same scheme as flat_mfma.hip): [nq][slot_stride] keys, class = row mod k
	unsigned *gslot;
	int slot_stride;
};

__device__ __forceinline__ bool sel_member(const SelectorDev &s, long long id) {
	if (s.kind == MVS_SEL_BITMAP) {
		unsigned long long u = (unsigned long long)id;
		if ((u >> 3) >= (unsigned long long)s.nbytes)
			return false;
		return (s.bitmap[u >> 3] >> (u & 7)) & 1;
	}
	if (s.kind == MVS_SEL_BATCH) {
		long long lo = 0, hi = s.nids;
		while (lo < hi) {
			long long mid = (lo + hi) >> 1;
			if (s.sorted_ids[mid] < id)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo < s.nids && s.sorted_ids[lo] == id;
	}
	return true;
}

__device__ __forceinline__ unsigned d_f2key(float f) {
	const unsigned b = __float_as_uint(f);
	return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float d_key2f(unsigned k) {
	return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
template <bool IS_L2>
__device__ __forceinline__ unsigned d_bkey(float v) { // "smaller is better" key
	return IS_L2 ? d_f2key(v) : ~d_f2key(v);
}

template <bool IS_L2>
__device__ __forceinline__ bool lex_better(float v, int id, float tv, int tid) {
	if (IS_L2)
		return v < tv || (v == tv && id < tid);
	return v > tv || (v == tv && id < tid);
}
template <bool IS_L2>
__device__ __forceinline__ bool lex_worse(float v, int id, float tv, int tid) {
	if (IS_L2)
		return v > tv || (v == tv && id > tid);
	return v < tv || (v == tv && id > tid);
}

template <int KC, int QG, int MODE>
__global__ __launch_bounds__(256) void flat_direct_kernel(const DirectArgs a) {
	constexpr bool IS_L2 = MODE != MODE_IP && MODE != MODE_JACCARD; // "smaller is better"
	constexpr int LDA = KC + 1;
	constexpr int F4_PER_ROW = KC / 4, F4 = DTILE * F4_PER_ROW, NLD = F4 / 256;
	extern __shared__ __attribute__((aligned(16))) float smem[];
	float *tbuf = smem; // [2][256][LDA]
	const int k = a.k;
	// per wave, per query: list [k] values + [k] ids, then worst (v,id,pos)
	float *lv = smem + 2 * DTILE * LDA;
	int *lid = (int *)(lv + 4 * QG * k);
	float *wv = (float *)(lid + 4 * QG * k);
	int *wid = (int *)(wv + 4 * QG);
	int *wpos = wid + 4 * QG;

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	int split = 0, q0 = 0, qbase = 0, nq_item = 0;
	long long r_begin, r_end;
	if (a.items) {
		const int4 it = a.items[blockIdx.x];
		r_begin = it.x;
		r_end = it.y;
		qbase = it.z;
		nq_item = it.w;
	} else {
		split = blockIdx.x / a.ngroups;
		q0 = (blockIdx.x % a.ngroups) * QG;
		r_begin = (long long)split * a.split_rows;
		r_end = r_begin + a.split_rows;
		nq_item = a.nq - q0 < QG ? a.nq - q0 : QG;
	}
	if (r_end > a.n)
		r_end = a.n;
	// query number of slot qq (wave-uniform); slots >= nq_item read query 0 and are never recorded
	auto qnum = [&](int qq) -> int {
		return qq < nq_item ? (a.items ? __builtin_amdgcn_readfirstlane(a.qidx[qbase + qq]) : q0 + qq) : 0;
	};
	// shared bound of every query slot of this workgroup: gb[qq] = max over the k class slots of the query (a valid
	// bound on its final k-th value; see flat_mfma.hip "threshold sharing"), refreshed once per row tile
	float *gb = (float *)(wpos + 4 * QG); // [QG] (written redundantly by all waves with the same values... per wave copy)
	gb += wave * QG;
	auto refresh_bounds = [&]() {
		for (int qq = 0; qq < QG; ++qq) {
			unsigned m = 0u;
			if (qq < nq_item && a.gslot) {
				const unsigned *sl = a.gslot + (size_t)qnum(qq) * a.slot_stride;
				for (int j = lane; j < a.slot_stride; j += 64) {
					const unsigned x = __hip_atomic_load(sl + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					m = x > m ? x : m;
				}
#pragma unroll
				for (int off = 32; off >= 1; off >>= 1) {
					const unsigned o = (unsigned)__shfl_xor((int)m, off);
					m = o > m ? o : m;
				}
			} else {
				m = 0xFFFFFFFFu;
			}
			const unsigned nk = d_bkey<IS_L2>(IS_L2 ? FLT_MAX : -FLT_MAX);
			if (lane == 0)
				gb[qq] = IS_L2 ? d_key2f(m < nk ? m : nk) : d_key2f(~(m < nk ? m : nk));
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin + DTILE - 1) / DTILE) : 0;
	const int nch = a.dp / KC;
	const int total_units = ntiles * nch;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;

	for (int i = lane; i < QG * k; i += 64) {
		lv[wave * QG * k + i] = neutral;
		lid[wave * QG * k + i] = -1;
	}
	if (lane < QG) {
		wv[wave * QG + lane] = neutral;
		wid[wave * QG + lane] = -1;
		wpos[wave * QG + lane] = 0;
	}

	float4 stg[NLD];
	auto stage_load = [&](int u) {
		const int tile = u / nch, ch = u - tile * nch;
		const long long row0 = r_begin + (long long)tile * DTILE;
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			const int f = i * 256 + tid;
			const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
			long long gr = row0 + row;
			if (gr >= a.n)
				gr = a.n - 1;
			stg[i] = *(const float4 *)(a.yb + (size_t)gr * a.dp + ch * KC + c4 * 4);
		}
	};
	auto stage_store = [&](int u) {
		float *dst = tbuf + (u & 1) * DTILE * LDA;
#pragma unroll
		for (int i = 0; i < NLD; ++i) {
			const int f = i * 256 + tid;
			const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
			float *p = dst + row * LDA + c4 * 4;
			// back to natural k order (FlatGeom::pair_interleaved; tiles start at multiples of 256 rows, so bit 4
			// of the global row index is bit 4 of the tile row)
			if (!a.interleaved) {
				p[0] = stg[i].x;
				p[1] = stg[i].y;
				p[2] = stg[i].z;
				p[3] = stg[i].w;
			} else if (row & 16) { // stored [k1,k3,k0,k2]
				p[0] = stg[i].z;
				p[1] = stg[i].x;
				p[2] = stg[i].w;
				p[3] = stg[i].y;
			} else { // stored [k0,k2,k1,k3]
				p[0] = stg[i].x;
				p[1] = stg[i].z;
				p[2] = stg[i].y;
				p[3] = stg[i].w;
			}
		}
	};

	float acc[QG];
	float acc2[mode_two_sums(MODE) ? QG : 1]; // BrayCurtis / Jaccard: the denominator
	if (total_units > 0) {
		stage_load(0);
		stage_store(0);
	}
	__syncthreads();

	for (int u = 0; u < total_units; ++u) {
		const int tile = u / nch, ch = u - tile * nch;
		if (u + 1 < total_units)
			stage_load(u + 1);
		if (ch == 0) {
#pragma unroll
			for (int qq = 0; qq < QG; ++qq) {
				acc[qq] = 0.f;
				if (mode_two_sums(MODE))
					acc2[qq] = 0.f;
			}
			refresh_bounds();
		}
		float y[KC];
		const float *src = tbuf + (u & 1) * DTILE * LDA + tid * LDA;
#pragma unroll
		for (int kk = 0; kk < KC; ++kk)
			y[kk] = src[kk];
#pragma unroll
		for (int qq = 0; qq < QG; ++qq) {
			// wave-uniform address in the constant address space => s_load (SGPR operands of the VALU ops)
			typedef __attribute__((address_space(4))) const float cfloat;
			cfloat *xs = (cfloat *)(a.xq + (size_t)qnum(qq) * a.dp + ch * KC);
			float s = acc[qq];
			float s2 = mode_two_sums(MODE) ? acc2[qq] : 0.f;
#pragma unroll
			for (int kk = 0; kk < KC; ++kk) {
				if (MODE == MODE_L2_PAIR) {
					const float t = xs[kk] - y[kk];
					s = fmaf(t, t, s);
				} else if (MODE <= MODE_L2_FORMULA) {
					s = fmaf(xs[kk], y[kk], s);
				} else {
					if (mode_needs_dim_guard(MODE) && ch * KC + kk >= a.d)
						continue;
					const float xi = xs[kk], yi = y[kk];
					if (MODE == MODE_L1) {
						s = __fadd_rn(s, fabsf(__fsub_rn(xi, yi)));
					} else if (MODE == MODE_LINF) {
						s = fmaxf(s, fabsf(__fsub_rn(xi, yi)));
					} else if (MODE == MODE_LP) {
						s = __fadd_rn(s, powf(fabsf(__fsub_rn(xi, yi)), a.metric_arg));
					} else if (MODE == MODE_CANBERRA) {
						s = __fadd_rn(s, __fdiv_rn(fabsf(__fsub_rn(xi, yi)), __fadd_rn(fabsf(xi), fabsf(yi))));
					} else if (MODE == MODE_BRAYCURTIS) {
						s = __fadd_rn(s, fabsf(__fsub_rn(xi, yi)));
						s2 = __fadd_rn(s2, fabsf(__fadd_rn(xi, yi)));
					} else if (MODE == MODE_JS) {
						const float mi = __fmul_rn(0.5f, __fadd_rn(xi, yi));
						const float kl1 = __fmul_rn(-xi, logf(__fdiv_rn(mi, xi)));
						const float kl2 = __fmul_rn(-yi, logf(__fdiv_rn(mi, yi)));
						s = __fadd_rn(s, __fadd_rn(kl1, kl2));
					} else { // MODE_JACCARD
						s = __fadd_rn(s, fminf(xi, yi));
						s2 = __fadd_rn(s2, fmaxf(xi, yi));
					}
				}
			}
			acc[qq] = s;
			if (mode_two_sums(MODE))
				acc2[qq] = s2;
		}

		if (ch == nch - 1) {
			const long long row0 = r_begin + (long long)tile * DTILE;
			const long long row = row0 + tid;
			bool valid = row < r_end;
			if (valid && a.sel.kind != MVS_SEL_NONE) {
				long long lab = a.rowids ? a.rowids[row] : row;
				valid = sel_member(a.sel, a.idmap ? a.idmap[lab] : lab);
			}
			float ynr = 0.f;
			if (MODE == MODE_L2_FORMULA)
				ynr = a.yn[row < a.n ? row : a.n - 1];
#pragma unroll
			for (int qq = 0; qq < QG; ++qq) {
				float v = acc[qq];
				if (MODE == MODE_L2_FORMULA) {
					v = fmaf(-2.0f, v, a.xn[qnum(qq)] + ynr);
					v = v < 0.f ? 0.f : v;
				}
				if (mode_two_sums(MODE))
					v = __fdiv_rn(v, acc2[qq]);
				if (MODE == MODE_JS)
					v = __fmul_rn(0.5f, v);
				const int slot = wave * QG + qq;
				float tv = wv[slot];
				const float gbv = gb[qq];
				// rows arrive in ascending id order, so an equal value never beats the stored worst
				const bool pass = valid && (qq < nq_item) && (IS_L2 ? (v < tv && v <= gbv) : (v > tv && v >= gbv));
				unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
				if (mask != 0ull) {
					int tpos = wpos[slot];
					float *mv = lv + slot * k;
					int *mi = lid + slot * k;
					while (mask) {
						const int l = __builtin_ctzll(mask);
						mask &= mask - 1;
						const float cv = __shfl(v, l);
						const int id = (int)(row0 + (tid - lane) + l);
						if (IS_L2 ? cv < tv : cv > tv) {
							// replace the worst entry, then wave-wide search for the new worst
							if (lane == 0) {
								mv[tpos] = cv;
								mi[tpos] = id;
								if (a.gslot) // publish the best value of this row's class (fire and forget)
									__hip_atomic_fetch_min(a.gslot + (size_t)qnum(qq) * a.slot_stride + (unsigned)id % (unsigned)k,
									                       d_bkey<IS_L2>(cv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							}
							__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
							__builtin_amdgcn_wave_barrier();
							float bv = 0.f;
							int bi = 0, bp = -1;
							for (int j = lane; j < k; j += 64) {
								const float x = mv[j];
								const int xi = mi[j];
								if (bp < 0 || lex_worse<IS_L2>(x, xi, bv, bi)) {
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
								const bool take = op >= 0 && (bp < 0 || lex_worse<IS_L2>(ov, oi, bv, bi));
								if (take) {
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
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					__builtin_amdgcn_wave_barrier();
				}
			}
		}
		if (u + 1 < total_units)
			stage_store(u + 1);
		__syncthreads();
	}

	// ---- workgroup-level merge of the 4 per-wave lists, then ONE partial list per (workgroup, query) -----------
	// regular grid: [nsplit][nq][k]; item mode: [item][QG][k].  Wave w merges the query slots qq = w, w+4, ...
	__syncthreads();
	for (int qq = wave; qq < QG; qq += 4) {
		if (qq >= nq_item)
			break;
		size_t base;
		if (a.items)
			base = ((size_t)blockIdx.x * QG + qq) * k;
		else
			base = ((size_t)split * a.nq + (q0 + qq)) * k;
		for (int r = 0; r < k; ++r) { // k rounds of wave-wide lexicographic best over the 4k candidates
			float bv = 0.f;
			int bi = 0x7fffffff, bp = -1;
			for (int c = lane; c < 4 * k; c += 64) {
				const int w2 = c / k, j = c - w2 * k, p2 = (w2 * QG + qq) * k + j;
				const float v = lv[p2];
				const int id = lid[p2];
				if (id < 0)
					continue;
				if (bp < 0 || lex_better<IS_L2>(v, id, bv, bi)) {
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
				if (op >= 0 && (bp < 0 || lex_better<IS_L2>(ov, oi, bv, bi))) {
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
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
	}
}

// -------------------------------------------------------------------------------------------------------

static int direct_kc(const FlatGeom &g) {
	return g.dp == 8 ? 8 : (g.dp == 16 ? 16 : 32);
}
static size_t direct_lds(int kc, int qg, int64_t k) {
	return (size_t)2 * DTILE * (kc + 1) * 4 + (size_t)4 * qg * k * 8 + 4 * qg * 16;
}
int64_t flat_direct_max_k() {
	return (int64_t)((160 * 1024 - direct_lds(32, 1, 0)) / 32);
}
static int pick_qgroup(int64_t nq, int64_t k, int kc) {
	const int cands[] = {20, 4, 1};
	for (int qg : cands) {
		if (qg > 1 && nq <= (qg == 20 ? 4 : 1))
			continue; // small batches: do not pad to a wide group
		if (direct_lds(kc, qg, k) <= 150 * 1024)
			return qg;
	}
	return 1;
}

static DirectPlan plan_direct_with_group(const FlatGeom &g, int64_t nq, int64_t n, int64_t k, int qgroup);
DirectPlan plan_flat_direct(const FlatGeom &g, int64_t nq, int64_t n, int64_t k) {
	return plan_direct_with_group(g, nq, n, k, pick_qgroup(nq, k, direct_kc(g)));
}
// the extra-metric instances exist for 20 and 1 queries per workgroup only
DirectPlan plan_flat_direct_extra(const FlatGeom &g, int64_t nq, int64_t n, int64_t k) {
	const int qg = nq > 1 && direct_lds(direct_kc(g), 20, k) <= 150 * 1024 ? 20 : 1;
	return plan_direct_with_group(g, nq, n, k, qg);
}
static DirectPlan plan_direct_with_group(const FlatGeom &g, int64_t nq, int64_t n, int64_t k, int qgroup) {
	DirectPlan p;
	const int kc = direct_kc(g);
	p.qgroup = qgroup;
	const int ngroups = (int)((nq + p.qgroup - 1) / p.qgroup);
	const int64_t ntiles = (n + DTILE - 1) / DTILE;
	int64_t nsplit = 2048 / (ngroups > 0 ? ngroups : 1);
	if (nsplit > ntiles / 4)
		nsplit = ntiles / 4;
	if (nsplit < 1)
		nsplit = 1;
	// the merge kernel holds 4*nsplit*k candidates in LDS
	while (nsplit > 1 && (size_t)(nsplit + 1) * k * 8 > 150 * 1024)
		nsplit /= 2;
	const int64_t tiles_per_split = ntiles > 0 ? (ntiles + nsplit - 1) / nsplit : 1;
	p.split_rows = tiles_per_split * DTILE;
	p.nsplit = (int)nsplit;
	p.grid = p.nsplit * ngroups;
	p.lds_bytes = direct_lds(kc, p.qgroup, k);
	return p;
}

template <int KC, int QG>
static void launch_direct_inst(int mode, const DirectArgs &a, const DirectPlan &p, hipStream_t st) {
#define MVS_LAUNCH_DIRECT(M)                                                                                           \
	{                                                                                                                  \
		auto kern = flat_direct_kernel<KC, QG, M>;                                                                     \
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes)); \
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);                                          \
	}
	if (mode == MODE_IP)
		MVS_LAUNCH_DIRECT(MODE_IP)
	else if (mode == MODE_L2_PAIR)
		MVS_LAUNCH_DIRECT(MODE_L2_PAIR)
	else if (mode == MODE_L2_FORMULA)
		MVS_LAUNCH_DIRECT(MODE_L2_FORMULA)
	else if constexpr (QG != 4) { // the extra-metric instances: 20 or 1 queries per workgroup (plan_flat_direct_extra)
		if (mode == MODE_L1)
			MVS_LAUNCH_DIRECT(MODE_L1)
		else if (mode == MODE_LINF)
			MVS_LAUNCH_DIRECT(MODE_LINF)
		else if (mode == MODE_LP)
			MVS_LAUNCH_DIRECT(MODE_LP)
		else if (mode == MODE_CANBERRA)
			MVS_LAUNCH_DIRECT(MODE_CANBERRA)
		else if (mode == MODE_BRAYCURTIS)
			MVS_LAUNCH_DIRECT(MODE_BRAYCURTIS)
		else if (mode == MODE_JS)
			MVS_LAUNCH_DIRECT(MODE_JS)
		else
			MVS_LAUNCH_DIRECT(MODE_JACCARD)
	} else
		throw_faiss("mvs::launch_direct_inst", __FILE__, "no 4-query instance for metric mode %d", mode);
#undef MVS_LAUNCH_DIRECT
	MVS_HIP(hipGetLastError());
}

template <int KC>
static void launch_direct_kc(int mode, int qg, const DirectArgs &a, const DirectPlan &p, hipStream_t st) {
	if (qg == 20)
		launch_direct_inst<KC, 20>(mode, a, p, st);
	else if (qg == 4)
		launch_direct_inst<KC, 4>(mode, a, p, st);
	else
		launch_direct_inst<KC, 1>(mode, a, p, st);
}

// mode_formula: use the BLAS-branch L2 value (xn + yn - 2ip); d_xn must then be valid
void launch_flat_direct(const FlatGeom &g, const DirectPlan &p, int metric, const float *d_xq, int64_t nq, FlatDB db,
                        int64_t k, SelectorDev sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi,
                        hipStream_t st);

void launch_flat_direct_ex(const FlatGeom &g, const DirectPlan &p, int metric, bool formula, const float *d_xq,
                           const float *d_xn, int64_t nq, FlatDB db, int64_t k, SelectorDev sel,
                           const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, hipStream_t st) {
	if (nq <= 0)
		return;
	DirectArgs a;
	a.xq = d_xq;
	a.xn = d_xn;
	a.yb = db.vecs;
	a.yn = db.norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = db.n;
	a.split_rows = p.split_rows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.dp = g.dp;
	a.interleaved = g.pair_interleaved ? 1 : 0;
	a.nsplit = p.nsplit;
	a.ngroups = (int)((nq + p.qgroup - 1) / p.qgroup);
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.items = nullptr;
	a.qidx = nullptr;
	a.rowids = nullptr;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	const int mode = metric == METRIC_IP ? MODE_IP : (formula ? MODE_L2_FORMULA : MODE_L2_PAIR);
	const int kc = direct_kc(g);
	if (kc == 8)
		launch_direct_kc<8>(mode, p.qgroup, a, p, st);
	else if (kc == 16)
		launch_direct_kc<16>(mode, p.qgroup, a, p, st);
	else
		launch_direct_kc<32>(mode, p.qgroup, a, p, st);
}

static int extra_mode(int metric) {
	switch (metric) {
	case METRIC_L1:
		return MODE_L1;
	case METRIC_LINF:
		return MODE_LINF;
	case METRIC_LP:
		return MODE_LP;
	case METRIC_CANBERRA:
		return MODE_CANBERRA;
	case METRIC_BRAYCURTIS:
		return MODE_BRAYCURTIS;
	case METRIC_JENSENSHANNON:
		return MODE_JS;
	case METRIC_JACCARD:
		return MODE_JACCARD;
	}
	throw_faiss("mvs::extra_mode", __FILE__, "metric %d has no kernel instance", metric);
}
void launch_flat_direct_extra(const FlatGeom &g, const DirectPlan &p, int metric, float metric_arg, int d,
                              const float *d_xq, int64_t nq, FlatDB db, int64_t k, SelectorDev sel,
                              const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, hipStream_t st) {
	if (nq <= 0)
		return;
	DirectArgs a;
	memset(&a, 0, sizeof a);
	a.xq = d_xq;
	a.yb = db.vecs;
	a.yn = db.norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = db.n;
	a.split_rows = p.split_rows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.dp = g.dp;
	a.d = d;
	a.metric_arg = metric_arg;
	a.interleaved = g.pair_interleaved ? 1 : 0;
	a.nsplit = p.nsplit;
	a.ngroups = (int)((nq + p.qgroup - 1) / p.qgroup);
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	const int mode = extra_mode(metric);
	const int kc = direct_kc(g);
	if (kc == 8)
		launch_direct_kc<8>(mode, p.qgroup, a, p, st);
	else if (kc == 16)
		launch_direct_kc<16>(mode, p.qgroup, a, p, st);
	else
		launch_direct_kc<32>(mode, p.qgroup, a, p, st);
}

__global__ void init_direct_slots_kernel(unsigned *g, long long total, int stride, int k, int is_l2) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total)
		g[i] = (int)(i % stride) < k ? (is_l2 ? d_f2key(FLT_MAX) : ~d_f2key(-FLT_MAX)) : 0u;
}
// slots [0,k) = neutral key, padding = 0 (never the maximum); stride = k rounded up to 16
void launch_init_slots(unsigned *d_gslot, int64_t nq, int64_t k, int metric, hipStream_t st) {
	const int stride = (int)((k + 15) / 16 * 16);
	const long long total = (long long)nq * stride;
	if (total <= 0)
		return;
	hipLaunchKernelGGL(init_direct_slots_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_gslot, total,
	                   stride, (int)k, metric == METRIC_L2 ? 1 : 0);
	MVS_HIP(hipGetLastError());
}

// IVF list scan: nitems work items over a plain-layout row store (csrc/ivf.hip); QG fixed at 20 slots per item
size_t direct_items_lds_bytes(int dp, int64_t k) {
	return direct_lds(dp == 8 ? 8 : (dp == 16 ? 16 : 32), 20, k);
}
void launch_direct_items(int dp, int metric, const float *d_xq, int64_t nq, const float *d_rows, int64_t nrows,
                         const int64_t *d_rowids, int64_t k, const void *d_items, int nitems, const int *d_qidx,
                         SelectorDev sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot,
                         hipStream_t st) {
	if (nitems <= 0)
		return;
	DirectArgs a;
	memset(&a, 0, sizeof a);
	a.xq = d_xq;
	a.yb = d_rows;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = nrows;
	a.nq = (int)nq;
	a.k = (int)k;
	a.dp = dp;
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.items = (const int4 *)d_items;
	a.qidx = d_qidx;
	a.rowids = (const long long *)d_rowids;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	DirectPlan p;
	p.grid = nitems;
	p.qgroup = 20;
	p.lds_bytes = direct_items_lds_bytes(dp, k);
	const int mode = metric == METRIC_IP ? MODE_IP : MODE_L2_PAIR; // IVFFlatScanner: per-pair arithmetic
	const int kc = dp == 8 ? 8 : (dp == 16 ? 16 : 32);
	if (kc == 8)
		launch_direct_kc<8>(mode, 20, a, p, st);
	else if (kc == 16)
		launch_direct_kc<16>(mode, 20, a, p, st);
	else
		launch_direct_kc<32>(mode, 20, a, p, st);
}

void launch_flat_direct(const FlatGeom &g, const DirectPlan &p, int metric, const float *d_xq, int64_t nq, FlatDB db,
                        int64_t k, SelectorDev sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi,
                        hipStream_t st) {
	launch_flat_direct_ex(g, p, metric, false, d_xq, nullptr, nq, db, k, sel, d_idmap, d_pd, d_pi, nullptr, st);
}

} // namespace mvs
