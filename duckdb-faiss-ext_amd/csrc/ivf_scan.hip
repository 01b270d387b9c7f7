// csrc/ivf_scan.hip -- IVFFlat inverted-list scan (IVFFlatScanner::scan_codes of faiss/IndexIVFFlat.cpp), list-major.
//
// One workgroup (256 threads) per work item = (one inverted list, <= 20 of the queries that probe it), built by
// csrc/ivf.hip.  The list is streamed from HBM once per item:
//   * thread <-> row: a thread reads its own row straight from global memory, 16 floats (4 x dwordx4) at a time with
//     the next chunk in flight -- no LDS staging, no workgroup barrier in the scan loop; the 64 rows of a wave span
//     64 cache lines per load instruction and the following instructions hit the same lines in L1;
//   * the query values are wave-uniform: they arrive through the scalar cache as s_load_dwordx16 and feed the VALU
//     as SGPR operands, so a distance costs exactly v_sub + v_fmac per dimension (L2) / v_fmac (IP);
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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) const f32x16 cf32x16;

struct ScanArgs {
	const float *xq; // [nq][dp]
	const float *rows; // [n][dp], lists are contiguous row segments
	float *pd;
	int32_t *pi;
	long long n;
	int k, dp;
	SelectorDev sel;
	const long long *idmap;
	const int4 *items;       // {row_begin, row_end, qoff, nq_item}
	const int *qidx;         // query number of slot qoff + s
	const long long *rowids; // stored id of every row
	unsigned *gslot;         // [nq][slot_stride] shared threshold classes
	int slot_stride;
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

template <bool IS_L2>
__global__ __launch_bounds__(256) void ivf_scan_kernel(const ScanArgs a) {
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int k = a.k;
	float *lv = smem;                       // [4][SQG][k] per-wave list values
	int *lid = (int *)(lv + 4 * SQG * k);   // [4][SQG][k] row positions
	float *wv = (float *)(lid + 4 * SQG * k); // [4][SQG] current worst of each list
	int *wid = (int *)(wv + 4 * SQG);
	int *wpos = wid + 4 * SQG;
	float *gb = (float *)(wpos + 4 * SQG);  // [4][SQG] shared bound per query slot (one copy per wave)
	int *qn_s = (int *)(gb + 4 * SQG);      // [SQG] query numbers (for lane-indexed access)

	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int4 it = a.items[blockIdx.x];
	const long long r_begin = it.x;
	long long r_end = it.y;
	const int qbase = it.z, nq_item = it.w;
	if (r_end > a.n)
		r_end = a.n;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;

	// wave-uniform query numbers (SGPRs); unused slots alias slot 0 and are never computed
	int qn[SQG];
#pragma unroll
	for (int qq = 0; qq < SQG; ++qq)
		qn[qq] = __builtin_amdgcn_readfirstlane(a.qidx[qbase + (qq < nq_item ? qq : 0)]);
	if (tid < SQG)
		qn_s[tid] = a.qidx[qbase + (tid < nq_item ? tid : 0)];
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
	auto refresh_bounds = [&]() {
		const unsigned nk = s_bkey<IS_L2>(neutral);
		if (!a.gslot) {
			if (lane < SQG)
				gb[lane] = neutral;
		} else if (a.slot_stride == 16) {
			unsigned m[SQG / 4];
#pragma unroll
			for (int p = 0; p < SQG / 4; ++p) { // 4 queries x 16 classes per pass
				const int qq = p * 4 + (lane >> 4);
				m[p] = qq < nq_item ? __hip_atomic_load(a.gslot + (size_t)qn_s[qq] * 16 + (lane & 15), __ATOMIC_RELAXED,
				                                        __HIP_MEMORY_SCOPE_AGENT)
				                    : 0u;
			}
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
		refresh_bounds();
		float acc[SQG];
#pragma unroll
		for (int qq = 0; qq < SQG; ++qq)
			acc[qq] = 0.f;
		for (int c = 0; c < nchunk; ++c) {
			float4 yn[4];
			const int cn = c + 1 < nchunk ? c + 1 : c;
#pragma unroll
			for (int i = 0; i < 4; ++i)
				yn[i] = yp[cn * 4 + i];
			const float y[16] = {yc[0].x, yc[0].y, yc[0].z, yc[0].w, yc[1].x, yc[1].y, yc[1].z, yc[1].w,
			                     yc[2].x, yc[2].y, yc[2].z, yc[2].w, yc[3].x, yc[3].y, yc[3].z, yc[3].w};
			// one basic block per slot: s_load_dwordx16 of the slot's 16 query values, then 16 x (v_sub, v_fmac).  The
			// scalar-load latency is covered by the other waves of the SIMD (a hand-pipelined variant that issued
			// the next slot's load early made the compiler roll the loop with indexed VGPRs and ran 17 % slower).
#pragma unroll
			for (int qq = 0; qq < SQG; ++qq) {
				if (qq < nq_item) { // wave-uniform
					const f32x16 x = *(cf32x16 *)(a.xq + (size_t)qn[qq] * a.dp + c * 16);
					float s = acc[qq];
#pragma unroll
					for (int kk = 0; kk < 16; ++kk) {
						if (IS_L2) {
							const float t = x[kk] - y[kk];
							s = fmaf(t, t, s);
						} else {
							s = fmaf(x[kk], y[kk], s);
						}
					}
					acc[qq] = s;
				}
			}
#pragma unroll
			for (int i = 0; i < 4; ++i)
				yc[i] = yn[i];
		}

		// ---- k-best update of the tile
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
		const size_t base = ((size_t)blockIdx.x * SQG + qq) * k;
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

void launch_ivf_scan(int dp, int metric, const float *d_xq, const float *d_rows, int64_t nrows, const int64_t *d_rowids,
                     int64_t k, const void *d_items, int nitems, const int *d_qidx, SelectorDev sel,
                     const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, hipStream_t st) {
	if (nitems <= 0)
		return;
	ScanArgs a;
	a.xq = d_xq;
	a.rows = d_rows;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = nrows;
	a.k = (int)k;
	a.dp = dp;
	a.sel = sel;
	a.idmap = (const long long *)d_idmap;
	a.items = (const int4 *)d_items;
	a.qidx = d_qidx;
	a.rowids = (const long long *)d_rowids;
	a.gslot = d_gslot;
	a.slot_stride = d_gslot ? (int)((k + 15) / 16 * 16) : 0;
	const size_t lds = scan_lds_bytes(k);
	if (metric == METRIC_IP) {
		MVS_HIP(hipFuncSetAttribute((const void *)ivf_scan_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
		                            (int)lds));
		hipLaunchKernelGGL(ivf_scan_kernel<false>, dim3(nitems), dim3(256), lds, st, a);
	} else {
		MVS_HIP(hipFuncSetAttribute((const void *)ivf_scan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
		                            (int)lds));
		hipLaunchKernelGGL(ivf_scan_kernel<true>, dim3(nitems), dim3(256), lds, st, a);
	}
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
