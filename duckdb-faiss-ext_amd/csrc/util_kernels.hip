// csrc/util_kernels.hip -- K1 (row norms), query fragment packing, K4 (partial-list merge), synthetic data.
#include "common.h"

#include <algorithm>

namespace mvs {

// ---- K1: squared norms, k-ordered fma chain (bit-exact with oracle orc_norms) ------------------------
// One thread per row.  Used at add time (database) -- FAISS recomputes y norms per search
// (utils/distances.cpp exhaustive_L2sqr_blas); storing them next to the shard is the MI355X design.
__global__ void row_norms_kernel(const float *__restrict__ v, long long n, int dp, float *__restrict__ out) {
	long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n)
		return;
	const float4 *p = (const float4 *)(v + (size_t)r * dp);
	float acc = 0.f;
	for (int i = 0; i < dp / 4; ++i) {
		float4 x = p[i];
		acc = fmaf(x.x, x.x, acc);
		acc = fmaf(x.y, x.y, acc);
		acc = fmaf(x.z, x.z, acc);
		acc = fmaf(x.w, x.w, acc);
	}
	out[r] = acc;
}
void launch_row_norms(const float *d_vecs, int64_t n, int dp, float *d_norms, hipStream_t st) {
	if (n <= 0)
		return;
	hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_vecs, (long long)n, dp,
	                   d_norms);
	MVS_HIP(hipGetLastError());
}

// row-major [n][d] -> row-major [n][dp] with zero padding
__global__ void pad_rows_kernel(const float *__restrict__ src, long long n, int d, float *__restrict__ dst, int dp) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long total = n * dp;
	if (i >= total)
		return;
	long long r = i / dp;
	int c = (int)(i - r * dp);
	dst[i] = c < d ? src[r * d + c] : 0.f;
}
void launch_pad_rows(const float *d_src, int64_t n, int d, float *d_dst, int dp, hipStream_t st) {
	if (n <= 0)
		return;
	long long total = (long long)n * dp;
	hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_src, (long long)n, d,
	                   d_dst, dp);
	MVS_HIP(hipGetLastError());
}

// [n][d] row-major -> storage rows (FlatGeom::pair_interleaved), one thread per 16-byte chunk
__global__ void pack_rows_kernel(const float *__restrict__ src, long long n, int d, float *__restrict__ dst, int dp,
                                 long long row0, int interleave) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int cpr = dp / 4;
	if (i >= n * cpr)
		return;
	const long long r = i / cpr;
	const int cg = (int)(i - r * cpr);
	const float *s = src + r * d + cg * 4;
	float v[4];
#pragma unroll
	for (int e = 0; e < 4; ++e)
		v[e] = (cg * 4 + e < d) ? s[e] : 0.f;
	float4 o;
	if (!interleave)
		o = make_float4(v[0], v[1], v[2], v[3]);
	else if (((row0 + r) >> 4) & 1)
		o = make_float4(v[1], v[3], v[0], v[2]);
	else
		o = make_float4(v[0], v[2], v[1], v[3]);
	((float4 *)dst)[i] = o;
}
void launch_pack_rows(const FlatGeom &g, const float *d_src, int64_t n, float *d_dst, int64_t row0, hipStream_t st) {
	if (n <= 0)
		return;
	const long long total = (long long)n * (g.dp / 4);
	hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_src, (long long)n,
	                   g.d, d_dst, g.dp, (long long)row0, g.pair_interleaved ? 1 : 0);
	MVS_HIP(hipGetLastError());
}

// the inverse: storage rows row0 + i * stride (i < n) -> plain [n][d] rows (the shadow clustering of a Flat index is built and extended
// from the rows where they are, csrc/index.hip FlatIndex::shadow_search)
__global__ void unpack_rows_kernel(const float *__restrict__ src, int dp, int interleave, long long row0, long long stride, long long n, int d,
                                   float *__restrict__ dst) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * d)
		return;
	const long long j = i / d, r = row0 + j * stride;
	const int k = (int)(i - j * d);
	int sk = k;
	if (interleave) { // stored [k0,k2,k1,k3] (bit 4 of r clear) or [k1,k3,k0,k2]
		const int e = k & 3;
		const int pos = ((r >> 4) & 1) ? (e == 0 ? 2 : e == 1 ? 0 : e == 2 ? 3 : 1) : (e == 0 ? 0 : e == 1 ? 2 : e == 2 ? 1 : 3);
		sk = (k & ~3) + pos;
	}
	dst[i] = src[r * dp + sk];
}
void launch_unpack_rows(const FlatGeom &g, const float *d_rows, int64_t row0, int64_t stride, int64_t n, float *d_dst, hipStream_t st) {
	if (n <= 0)
		return;
	const long long total = (long long)n * g.d;
	hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_rows, g.dp, g.pair_interleaved ? 1 : 0,
	                   (long long)row0, (long long)stride, (long long)n, g.d, d_dst);
	MVS_HIP(hipGetLastError());
}

// ---- query packing: [nq][d] -> MFMA B-fragment order + norms -----------------------------------------
// qf[(((qblk32*nch + ch)*(KSTEPS/4) + s4)*64 + lane)*4 + e] = x[qblk32*32 + (lane&31)][ch*kc + 2*(4*s4+e) + (lane>>5)]
__global__ void pack_queries_kernel(const float *__restrict__ x, long long nq, int d, int kc, int nch,
                                    float *__restrict__ qf, long long total4) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // one float4 each
	if (i >= total4)
		return;
	const int ks4 = kc / 8; // KSTEPS/4
	int lane = (int)(i & 63);
	long long t = i >> 6;
	int s4 = (int)(t % ks4);
	t /= ks4;
	int ch = (int)(t % nch);
	long long qblk32 = t / nch;
	long long q = qblk32 * 32 + (lane & 31);
	float o[4];
#pragma unroll
	for (int e = 0; e < 4; ++e) {
		int kk = ch * kc + 2 * (4 * s4 + e) + (lane >> 5);
		o[e] = (q < nq && kk < d) ? x[q * d + kk] : 0.f;
	}
	((float4 *)qf)[i] = make_float4(o[0], o[1], o[2], o[3]);
}
// ||x_q||^2 as ONE k-ordered fma chain per query (the value FAISS's fvec_norm_L2sqr reference loop produces): one thread per
// query runs the chain; the rows reach it through LDS in coalesced 64-dim slabs (a thread walking its own row from global
// memory costs 3.6 ms for 10 000 x 768)
// (round 4: 16 rows per one-wave workgroup instead of 64 rows per 256 threads of which one wave ran the chains -- 10 000 queries
// are 625 workgroups, not 157 on a 256-CU device -- and the chain loop unrolled so that its LDS reads are batched: the search
// kernels wait for this one at the head of every step)
__global__ __launch_bounds__(64) void query_norms_kernel(const float *__restrict__ x, long long nq, int d, float *__restrict__ out) {
	// 16 rows per wavefront, 128 columns per pass: the 32 loads of a pass are in flight together (round 5; 64 columns per pass before:
	// two load -> chain round trips at d = 128), then lanes 0..15 run the k-ordered chains of their rows
	__shared__ float tile[16][129];
	const long long q0 = (long long)blockIdx.x * 16;
	const int t = threadIdx.x;
	float acc = 0.f;
	for (int c0 = 0; c0 < d; c0 += 128) {
		const int w = d - c0 < 128 ? d - c0 : 128;
		float v[32];
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const long long q = q0 + r < nq ? q0 + r : nq - 1; // (clamped: no branch between the requests; rows past the end are not written)
			v[2 * r] = x[q * d + c0 + (t < w ? t : 0)];
			v[2 * r + 1] = x[q * d + c0 + (64 + t < w ? 64 + t : 0)];
		}
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			tile[r][t] = t < w ? v[2 * r] : 0.f;
			tile[r][64 + t] = 64 + t < w ? v[2 * r + 1] : 0.f;
		}
		__syncthreads();
		if (t < 16) {
			if (w == 128) {
#pragma unroll 16
				for (int i = 0; i < 128; ++i)
					acc = fmaf(tile[t][i], tile[t][i], acc);
			} else {
				for (int i = 0; i < w; ++i)
					acc = fmaf(tile[t][i], tile[t][i], acc);
			}
		}
		__syncthreads();
	}
	if (t < 16 && q0 + t < nq)
		out[q0 + t] = acc;
}
void launch_query_norms(const float *d_x, int64_t n, int d, float *d_out, hipStream_t st) {
	if (n <= 0)
		return;
	hipLaunchKernelGGL(query_norms_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, st, d_x, (long long)n, d, d_out);
	MVS_HIP(hipGetLastError());
}
void launch_pack_queries(const FlatGeom &g, const float *d_x, int64_t nq, float *d_qf, float *d_qnorm,
                         hipStream_t st) {
	if (nq <= 0)
		return;
	long long total4 = (long long)(qfrag_floats(g, nq) / 4);
	hipLaunchKernelGGL(pack_queries_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, d_x,
	                   (long long)nq, g.d, g.kc, g.nch, d_qf, total4);
	if (d_qnorm)
		launch_query_norms(d_x, nq, g.d, d_qnorm, st);
	MVS_HIP(hipGetLastError());
}

// ---- K4: merge nsplit partial lists per query into the final k, FAISS order -------------------------
// One wave per query.  Candidates (value, row id) sit in LDS; k rounds of "wave-wide lexicographic best".
//   L2: ascending (dist, id)            [Heap.h heap_reorder over a CMax heap]
//   IP: descending score; membership prefers the smaller id, equal scores are PRINTED in descending id
//       order (heap_reorder over a CMin heap pops the smallest id of equal values to the back)
// kout <= k entries are written (tie detection merges k = kout + 1 candidates); flag != nullptr: see TieFlags.
template <bool IS_L2>
__global__ __launch_bounds__(64) void merge_partials_kernel(const float *__restrict__ pd, const int32_t *__restrict__ pi,
                                                           int nsplit, long long nq, int k,
                                                           const long long *__restrict__ idmap, long long label_offset,
                                                           float *__restrict__ D, long long *__restrict__ I, int kout,
                                                           TieFlags flag) {
	extern __shared__ __attribute__((aligned(16))) float sm[];
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const int C = nsplit * k;
	float *cv = sm;
	int *ci = (int *)(sm + C);
	float *ov = (float *)(ci + C);
	int *oi = (int *)(ov + k);
	for (int i = lane; i < C; i += 64) {
		int s = i / k, j = i - s * k;
		cv[i] = pd[((size_t)s * nq + q) * k + j];
		ci[i] = pi[((size_t)s * nq + q) * k + j];
	}
	__syncthreads();
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	for (int r = 0; r < k; ++r) {
		// lane-local best over its strided subset
		float bv = neutral;
		int bi = 0x7fffffff, bp = -1;
		for (int i = lane; i < C; i += 64) {
			float v = cv[i];
			int id = ci[i];
			if (id < 0)
				continue;
			bool better = IS_L2 ? (v < bv || (v == bv && id < bi)) : (v > bv || (v == bv && id < bi));
			if (bp < 0 || better) {
				bv = v;
				bi = id;
				bp = i;
			}
		}
		// wave reduction
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			float ov_ = __shfl_xor(bv, off);
			int oi_ = __shfl_xor(bi, off);
			int op_ = __shfl_xor(bp, off);
			bool take;
			if (op_ < 0)
				take = false;
			else if (bp < 0)
				take = true;
			else
				take = IS_L2 ? (ov_ < bv || (ov_ == bv && oi_ < bi)) : (ov_ > bv || (ov_ == bv && oi_ < bi));
			if (take) {
				bv = ov_;
				bi = oi_;
				bp = op_;
			}
		}
		if (lane == 0) {
			if (bp >= 0) {
				ov[r] = bv;
				oi[r] = bi;
				ci[bp] = -1; // consumed
			} else {
				ov[r] = neutral;
				oi[r] = -1;
			}
		}
		__syncthreads();
	}
	for (int j = lane; j < kout; j += 64) {
		int src = j;
		if (!IS_L2 && oi[j] >= 0) {
			// reverse each run of equal scores (print order: larger id first)
			int a = j, b = j;
			while (a > 0 && oi[a - 1] >= 0 && ov[a - 1] == ov[j])
				--a;
			while (b + 1 < kout && oi[b + 1] >= 0 && ov[b + 1] == ov[j])
				++b;
			src = a + (b - j);
		}
		int id = oi[src];
		long long label = id < 0 ? -1ll : (idmap ? idmap[id] : (long long)id + label_offset);
		D[q * kout + j] = ov[src];
		I[q * kout + j] = label;
	}
	// boundary tie: the kout-th and (kout+1)-th best scores are bit-equal -> which of the tied rows FAISS's heap keeps
	// depends on arrival order; hand the query to the tie pass with its raw candidates
	if (flag.count && kout < k && oi[kout] >= 0 && ov[kout] == ov[kout - 1]) {
		int slot = 0;
		if (lane == 0) {
			slot = atomicAdd(flag.count, 1);
			flag.query[slot] = (int)q;
		}
		slot = __shfl(slot, 0);
		for (int j = lane; j < k; j += 64) {
			flag.val[(size_t)slot * k + j] = ov[j];
			flag.row[(size_t)slot * k + j] = oi[j];
		}
	}
}

__global__ void gather_flagged_kernel(const float *__restrict__ x, int d, const int *__restrict__ fq,
                                      const float *__restrict__ fval, int nf, int k, int kout, float *__restrict__ xf,
                                      float *__restrict__ T) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (long long)nf * d)
		return;
	const int f = (int)(i / d), c = (int)(i - (long long)f * d);
	xf[i] = x[(long long)fq[f] * d + c];
	if (c == 0)
		T[f] = fval[(size_t)f * k + kout - 1];
}
void launch_gather_flagged(const float *d_x, int d, const TieFlags &f, int nf, int64_t k, int64_t kout, float *d_xf,
                           float *d_T, hipStream_t st) {
	if (nf <= 0)
		return;
	const long long total = (long long)nf * d;
	hipLaunchKernelGGL(gather_flagged_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_x, d, f.query,
	                   f.val, nf, (int)k, (int)kout, d_xf, d_T);
	MVS_HIP(hipGetLastError());
}

// FAISS's inner-product result under an exact tie at the k-th score (faiss/utils/Heap.h, CMin heap fed in ascending
// row order with the strict insert rule).  Let T = k-th best score, A = rows with score >= T in
// ascending row order, A_k = its first k entries.  Until A_k is complete the heap's root is below T, so every row of
// A_k is inserted and none of them evicted; afterwards rows equal to T are rejected and every later row ABOVE T
// evicts the root = the tied row with the smallest id.  Hence
//     result = {rows above T}  +  {tied rows of A_k, minus the G with the smallest ids},  G = #(rows above T not in A_k)
// printed by heap_reorder: score descending, equal scores in descending id order.
// One thread per flagged query: raw = merged top-(k+1) in the pure order (score desc, id asc), which holds every row
// above T; first = A_k from the tie pass (ascending ids).
__global__ void tie_resolve_kernel(TieFlags f, int nf, int kraw, int k, const long long *__restrict__ first,
                                   const long long *__restrict__ idmap, long long label_offset, float *__restrict__ D,
                                   long long *__restrict__ I) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nf)
		return;
	const long long q = f.query[t];
	const float *rv = f.val + (size_t)t * kraw;
	const int *rr = f.row + (size_t)t * kraw;
	const long long *ak = first + (size_t)t * k;
	const float T = rv[k - 1];
	int ngt = 0;
	while (ngt < k && rv[ngt] > T)
		++ngt;
	auto above = [&](long long row) {
		for (int j = 0; j < ngt; ++j)
			if (rr[j] == row)
				return true;
		return false;
	};
	int in_a = 0;
	for (int j = 0; j < k; ++j)
		if (ak[j] >= 0 && above(ak[j]))
			++in_a;
	const int G = ngt - in_a;
	float *Dq = D + q * k;
	long long *Iq = I + q * k;
	// rows above T: raw order is (score desc, id asc); print equal-score runs in descending id order
	for (int j = 0; j < ngt;) {
		int e = j;
		while (e + 1 < ngt && rv[e + 1] == rv[j])
			++e;
		for (int m = j; m <= e; ++m) {
			const int id = rr[j + (e - m)];
			Dq[m] = rv[j];
			Iq[m] = idmap ? idmap[id] : (long long)id + label_offset;
		}
		j = e + 1;
	}
	// tied rows of A_k in ascending id, the first G dropped; written back to front (descending id)
	int seen = 0, pos = k - 1;
	for (int j = 0; j < k; ++j) {
		const long long row = ak[j];
		if (row < 0 || above(row))
			continue;
		if (seen++ < G)
			continue;
		if (pos < ngt)
			break; // cannot happen (p - G = k - ngt); guards the write
		Dq[pos] = T;
		Iq[pos] = idmap ? idmap[row] : row + label_offset;
		--pos;
	}
	// the kept tied rows were written from the back in ASCENDING id, i.e. the slice reads descending from ngt on -- but
	// only if it is full; compact if fewer than k - ngt were found (k >= number of rows >= T cannot occur when flagged)
	if (pos >= ngt) {
		const int missing = pos - ngt + 1;
		for (int m = ngt; m + missing < k; ++m) {
			Dq[m] = Dq[m + missing];
			Iq[m] = Iq[m + missing];
		}
		for (int m = k - missing; m < k; ++m) {
			Dq[m] = -FLT_MAX;
			Iq[m] = -1;
		}
	}
}
void launch_tie_resolve(const TieFlags &f, int nf, int64_t k, int64_t kout, const int64_t *d_first_ids,
                        const int64_t *d_idmap, int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st) {
	if (nf <= 0)
		return;
	hipLaunchKernelGGL(tie_resolve_kernel, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, st, f, nf, (int)k, (int)kout,
	                   (const long long *)d_first_ids, (const long long *)d_idmap, (long long)label_offset, d_D,
	                   (long long *)d_I);
	MVS_HIP(hipGetLastError());
}

// One list per query that is ALREADY in FAISS's L2 order -- (distance, row) ascending, missing entries (FLT_MAX, -1) at the end: what
// collect_select_kernel leaves -- only needs its labels: D / I [nq][kout] <- the first kout of k entries.  (merge_partials_kernel
// with one split re-derives the order in k wave-wide rounds: 30 us per 10 000 queries at k = 10, a hundredth of a shard's step)
__global__ void emit_sorted_kernel(const float *__restrict__ pd, const int32_t *__restrict__ pi, int k, int kout, long long total,
                                   const long long *__restrict__ idmap, long long label_offset, float *__restrict__ D,
                                   long long *__restrict__ I) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total)
		return;
	const long long q = i / kout;
	const int j = (int)(i - q * kout);
	const int id = pi[q * k + j];
	D[i] = id < 0 ? FLT_MAX : pd[q * k + j];
	I[i] = id < 0 ? -1ll : (idmap ? idmap[id] : (long long)id + label_offset);
}
void launch_emit_sorted(const float *d_pd, const int32_t *d_pi, int64_t nq, int64_t k, int64_t kout, const int64_t *d_idmap,
                        int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st) {
	const long long total = (long long)nq * kout;
	if (total <= 0)
		return;
	hipLaunchKernelGGL(emit_sorted_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_pd, d_pi, (int)k, (int)kout, total,
	                   (const long long *)d_idmap, (long long)label_offset, d_D, (long long *)d_I);
	MVS_HIP(hipGetLastError());
}

// ONE list per query that is ALREADY in the pure order (L2: (distance, id) ascending; inner product: score descending, id ascending) --
// what the coarse filter's selection leaves: FAISS's print order and the boundary flag without the k selection rounds above
// (k = 1001: 2.9 ms per 2 048 queries in merge_partials_kernel, whose rounds each walk the whole list; round 6).
template <bool IS_L2>
__global__ __launch_bounds__(64) void emit_presorted_kernel(const float *__restrict__ pd, const int32_t *__restrict__ pi, long long nq, int k,
                                                           const long long *__restrict__ idmap, long long label_offset,
                                                           float *__restrict__ D, long long *__restrict__ I, int kout, TieFlags flag) {
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const float *ov = pd + (size_t)q * k;
	const int32_t *oi = pi + (size_t)q * k;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	for (int j = lane; j < kout; j += 64) {
		int src = j;
		if (!IS_L2 && oi[j] >= 0) { // reverse each run of equal scores (print order: larger id first)
			int a = j, b = j;
			while (a > 0 && oi[a - 1] >= 0 && ov[a - 1] == ov[j])
				--a;
			while (b + 1 < kout && oi[b + 1] >= 0 && ov[b + 1] == ov[j])
				++b;
			src = a + (b - j);
		}
		const int id = oi[src];
		D[q * kout + j] = id < 0 ? neutral : ov[src];
		I[q * kout + j] = id < 0 ? -1ll : (idmap ? idmap[id] : (long long)id + label_offset);
	}
	if (flag.count && kout < k && oi[kout] >= 0 && ov[kout] == ov[kout - 1]) {
		int slot = 0;
		if (lane == 0) {
			slot = atomicAdd(flag.count, 1);
			flag.query[slot] = (int)q;
		}
		slot = __shfl(slot, 0);
		for (int j = lane; j < k; j += 64) {
			flag.val[(size_t)slot * k + j] = oi[j] >= 0 ? ov[j] : neutral;
			flag.row[(size_t)slot * k + j] = oi[j];
		}
	}
}
void launch_merge_partials(int metric, const float *d_pd, const int32_t *d_pi, int nsplit, int64_t nq, int64_t k,
                           const int64_t *d_idmap, int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st,
                           int64_t kout_, const TieFlags *flags, bool presorted) {
	if (nq <= 0)
		return;
	const int kout = (int)(kout_ < 0 ? k : kout_);
	TieFlags fl = {nullptr, nullptr, nullptr, nullptr};
	if (flags)
		fl = *flags;
	if (presorted && nsplit == 1) {
		if (metric_order(metric) == METRIC_L2)
			hipLaunchKernelGGL(emit_presorted_kernel<true>, dim3((unsigned)nq), dim3(64), 0, st, d_pd, d_pi, (long long)nq, (int)k,
			                   (const long long *)d_idmap, (long long)label_offset, d_D, (long long *)d_I, kout, fl);
		else
			hipLaunchKernelGGL(emit_presorted_kernel<false>, dim3((unsigned)nq), dim3(64), 0, st, d_pd, d_pi, (long long)nq, (int)k,
			                   (const long long *)d_idmap, (long long)label_offset, d_D, (long long *)d_I, kout, fl);
		MVS_HIP(hipGetLastError());
		return;
	}
	size_t lds = ((size_t)nsplit * k + k) * 8;
	if (lds > 160 * 1024)
		throw_faiss(__func__, __FILE__, "merge: nsplit*k = %lld too large", (long long)nsplit * k);
	if (metric_order(metric) == METRIC_L2) {
		auto kern = merge_partials_kernel<true>;
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, d_pd, d_pi, nsplit, (long long)nq, (int)k,
		                   (const long long *)d_idmap, (long long)label_offset, d_D, (long long *)d_I, kout, fl);
	} else {
		auto kern = merge_partials_kernel<false>;
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, d_pd, d_pi, nsplit, (long long)nq, (int)k,
		                   (const long long *)d_idmap, (long long)label_offset, d_D, (long long *)d_I, kout, fl);
	}
	MVS_HIP(hipGetLastError());
}

// ---- IVF: merge the per-(probe item, wave) partial lists of one query ---------------------------------
// slots[q*nprobe + p] = (item << 5 | slot) of probe p, or -1; partial lists live at [item][20][k].
// Order: L2 (dist asc, row position asc); IP (score desc, row position asc), equal scores printed in descending
// label order is NOT attempted here: FAISS's own IVF tie order depends on probe order (DESIGN.md "ties").
// Probes are merged in chunks of `pchunk` lists: LDS holds the running best k (front) plus one chunk of candidates, so
// any nprobe (up to nlist) fits.
template <bool IS_L2>
__global__ __launch_bounds__(64) void merge_items_kernel(const float *__restrict__ pd, const int32_t *__restrict__ pi,
                                                        const int *__restrict__ slots, int nprobe, int pchunk, int k,
                                                        int group, int shift, const long long *__restrict__ rowids,
                                                        const long long *__restrict__ idmap, float *__restrict__ D,
                                                        long long *__restrict__ I) {
	extern __shared__ __attribute__((aligned(16))) float sm[];
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const int CAP = (pchunk + 1) * k; // [0,k): running best, [k, k + chunk*k): candidates of the current chunk
	float *cv = sm;
	int *ci = (int *)(sm + CAP);
	float *ov = (float *)(ci + CAP); // selection output of a pass
	int *oi = (int *)(ov + k);
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	for (int i = lane; i < k; i += 64) {
		cv[i] = neutral;
		ci[i] = -1;
	}
	for (int p0 = 0; p0 < nprobe; p0 += pchunk) {
		const int np = nprobe - p0 < pchunk ? nprobe - p0 : pchunk;
		const int C = (np + 1) * k;
		for (int i = k + lane; i < C; i += 64) {
			const int p = p0 + (i - k) / k, j = (i - k) % k;
			const int s = slots[q * nprobe + p];
			if (s < 0) {
				cv[i] = 0.f;
				ci[i] = -1;
			} else {
				const size_t base = ((size_t)(s >> shift) * group + (s & ((1 << shift) - 1))) * k + j;
				cv[i] = pd[base];
				ci[i] = pi[base];
			}
		}
		__syncthreads();
		for (int r = 0; r < k; ++r) {
			float bv = neutral;
			int bi = 0x7fffffff, bp = -1;
			for (int i = lane; i < C; i += 64) {
				const float v = cv[i];
				const int id = ci[i];
				if (id < 0)
					continue;
				const bool better = IS_L2 ? (v < bv || (v == bv && id < bi)) : (v > bv || (v == bv && id < bi));
				if (bp < 0 || better) {
					bv = v;
					bi = id;
					bp = i;
				}
			}
#pragma unroll
			for (int off = 32; off >= 1; off >>= 1) {
				const float ov_ = __shfl_xor(bv, off);
				const int oi_ = __shfl_xor(bi, off);
				const int op_ = __shfl_xor(bp, off);
				bool take;
				if (op_ < 0)
					take = false;
				else if (bp < 0)
					take = true;
				else
					take = IS_L2 ? (ov_ < bv || (ov_ == bv && oi_ < bi)) : (ov_ > bv || (ov_ == bv && oi_ < bi));
				if (take) {
					bv = ov_;
					bi = oi_;
					bp = op_;
				}
			}
			if (lane == 0) {
				if (bp >= 0)
					ci[bp] = -1;
				ov[r] = bp >= 0 ? bv : neutral;
				oi[r] = bp >= 0 ? bi : -1;
			}
			__syncthreads();
		}
		for (int i = lane; i < k; i += 64) { // the pass result becomes the running best
			cv[i] = ov[i];
			ci[i] = oi[i];
		}
		__syncthreads();
	}
	for (int r = lane; r < k; r += 64) {
		const int bi = ci[r];
		long long label = -1;
		if (bi >= 0) {
			label = rowids ? rowids[bi] : (long long)bi; // rowids == nullptr: the caller wants row positions
			if (idmap)
				label = idmap[label];
		}
		D[q * k + r] = bi >= 0 ? cv[r] : neutral;
		I[q * k + r] = label;
	}
}
void launch_merge_items(int metric, const float *d_pd, const int32_t *d_pi, const int *d_slots, int nprobe, int64_t nq,
                        int64_t k, const int64_t *d_rowids, const int64_t *d_idmap, float *d_D, int64_t *d_I,
                        hipStream_t st, int group, int shift) {
	if (nq <= 0)
		return;
	// candidates of one pass: at most ~8k entries (64 KB of LDS)
	int pchunk = (int)std::max<int64_t>(1, std::min<int64_t>(nprobe, 8192 / k));
	const size_t lds = ((size_t)(pchunk + 1) * k + k) * 8;
	if (lds > 150 * 1024)
		throw_faiss(__func__, __FILE__, "IVF merge: k = %lld too large", (long long)k);
	if (metric_order(metric) == METRIC_L2) {
		auto kern = merge_items_kernel<true>;
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, d_pd, d_pi, d_slots, nprobe, pchunk, (int)k, group,
		                   shift, (const long long *)d_rowids, (const long long *)d_idmap, d_D, (long long *)d_I);
	} else {
		auto kern = merge_items_kernel<false>;
		ensure_dynamic_lds((const void *)kern, (size_t)(lds));
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, d_pd, d_pi, d_slots, nprobe, pchunk, (int)k, group,
		                   shift, (const long long *)d_rowids, (const long long *)d_idmap, d_D, (long long *)d_I);
	}
	MVS_HIP(hipGetLastError());
}

// ---- cross-shard merge ON THE DEVICE (one process per GPU: pyhost/sharded.py after the RCCL all-gather) -------------------
// rec: [nshard][nq][kk][2] int64 records {value bits (low 32), global label} exactly as gathered; one wave per query keeps
// the kout best of the nshard * kk candidates under the pure order (L2: value asc, label asc; inner product: value desc,
// label asc) and, unless `raw`, prints inner-product runs of equal scores in descending label order (heap_reorder over a
// CMin heap; csrc/merge_host.hip merge_shards_host is the host twin).  Entries with label < 0 are empty.
template <bool IS_L2>
__global__ __launch_bounds__(64) void merge_records_kernel(const long long *__restrict__ rec, int nshard, long long nq, int kk,
                                                          int kout, int raw, float *__restrict__ D,
                                                          long long *__restrict__ I) {
	extern __shared__ __attribute__((aligned(16))) long long mr_sm[];
	const long long q = blockIdx.x;
	const int lane = threadIdx.x, c = nshard * kk;
	long long *lab = mr_sm;              // [c]
	float *val = (float *)(lab + c);     // [c]
	float *ov = val + c;                 // [kout]
	long long *oi = (long long *)(mr_sm + c + (c + kout + 1) / 2); // [kout] (8-byte aligned)
	for (int i = lane; i < c; i += 64) {
		const int s = i / kk, j = i - s * kk;
		const long long *r = rec + (((size_t)s * nq + q) * kk + j) * 2;
		val[i] = __int_as_float((int)r[0]);
		lab[i] = r[1];
	}
	__syncthreads();
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	for (int o = 0; o < kout; ++o) {
		float bv = 0.f;
		long long bl = -1;
		int bi = -1;
		for (int i = lane; i < c; i += 64) {
			const long long l = lab[i];
			if (l < 0)
				continue;
			const float v = val[i];
			const bool better = bi < 0 || (IS_L2 ? (v < bv || (v == bv && l < bl)) : (v > bv || (v == bv && l < bl)));
			if (better) {
				bv = v;
				bl = l;
				bi = i;
			}
		}
		for (int off = 32; off >= 1; off >>= 1) {
			const float v2 = __shfl_xor(bv, off);
			const long long l2 = __shfl_xor(bl, off);
			const int i2 = __shfl_xor(bi, off);
			const bool take = i2 >= 0 && (bi < 0 || (IS_L2 ? (v2 < bv || (v2 == bv && l2 < bl)) : (v2 > bv || (v2 == bv && l2 < bl))));
			if (take) {
				bv = v2;
				bl = l2;
				bi = i2;
			}
		}
		if (lane == 0) {
			ov[o] = bi >= 0 ? bv : neutral;
			oi[o] = bi >= 0 ? bl : -1;
			if (bi >= 0)
				lab[bi] = -1; // consumed
		}
		__syncthreads();
	}
	for (int o = lane; o < kout; o += 64) {
		int src = o;
		if (!IS_L2 && !raw && oi[o] >= 0) { // the run of equal scores around o, printed back to front
			int a = o, b = o + 1;
			while (a > 0 && oi[a - 1] >= 0 && ov[a - 1] == ov[o])
				--a;
			while (b < kout && oi[b] >= 0 && ov[b] == ov[o])
				++b;
			src = a + b - 1 - o;
		}
		D[q * kout + o] = ov[src];
		I[q * kout + o] = oi[src];
	}
}
void launch_merge_records(int metric, const int64_t *d_rec, int nshard, int64_t nq, int kk, int kout, bool raw, float *d_D,
                          int64_t *d_I, hipStream_t st) {
	if (nq <= 0)
		return;
	const int c = nshard * kk;
	const size_t lds = (size_t)c * 8 + ((size_t)(c + kout + 1) / 2) * 8 + (size_t)kout * 8 + 16;
	if (lds > 150 * 1024) {
		// k in the thousands on several shards: the candidates of one query no longer fit a workgroup's LDS.  A cold shape
		// (a single GPU serves such k through flat_direct / the all-distances IVF path, not through k-lists either): the
		// records go to the host, merge_records_host (csrc/merge_host.hip) keeps the same kout best in the same order.
		std::vector<int64_t> h_rec((size_t)nshard * nq * kk * 2);
		std::vector<float> h_D((size_t)nq * kout);
		std::vector<int64_t> h_I((size_t)nq * kout);
		MVS_HIP(hipMemcpyAsync(h_rec.data(), d_rec, h_rec.size() * 8, hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		merge_records_host(metric, h_rec.data(), nshard, nq, kk, kout, raw, h_D.data(), h_I.data());
		MVS_HIP(hipMemcpyAsync(d_D, h_D.data(), h_D.size() * 4, hipMemcpyHostToDevice, st));
		MVS_HIP(hipMemcpyAsync(d_I, h_I.data(), h_I.size() * 8, hipMemcpyHostToDevice, st));
		MVS_HIP(hipStreamSynchronize(st));
		return;
	}
	if (metric_order(metric) == METRIC_L2) {
		auto kern = merge_records_kernel<true>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, (const long long *)d_rec, nshard, (long long)nq, kk, kout,
		                   raw ? 1 : 0, d_D, (long long *)d_I);
	} else {
		auto kern = merge_records_kernel<false>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3((unsigned)nq), dim3(64), lds, st, (const long long *)d_rec, nshard, (long long)nq, kk, kout,
		                   raw ? 1 : 0, d_D, (long long *)d_I);
	}
	MVS_HIP(hipGetLastError());
}

// rows gathered by a permutation: dst[i] = src[perm[i]] (16-byte chunks; dp % 4 == 0)
__global__ void gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ perm, long long n, int dp,
                                   float *__restrict__ dst) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int cpr = dp / 4;
	if (i >= n * cpr)
		return;
	const long long r = i / cpr;
	const int c = (int)(i - r * cpr);
	((float4 *)dst)[i] = ((const float4 *)src)[(long long)perm[r] * cpr + c];
}
void launch_gather_rows(const float *d_src, const int *d_perm, int64_t n, int dp, float *d_dst, hipStream_t st) {
	if (n <= 0)
		return;
	const long long total = (long long)n * (dp / 4);
	hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_src, d_perm,
	                   (long long)n, dp, d_dst);
	MVS_HIP(hipGetLastError());
}

// ---- synthetic data: same integer arithmetic as oracle/orc_core.c orc_synth_* -------------------------
__device__ __forceinline__ unsigned long long splitmix(unsigned long long z) {
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(unsigned long long seed, unsigned long long ctr) {
	unsigned long long z = splitmix(seed + (ctr + 1) * 0x9E3779B97F4A7C15ull);
	return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__global__ void synth_uniform_kernel(float *out, long long total, unsigned long long seed, long long ctr0) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long stride = (long long)gridDim.x * blockDim.x;
	for (; i < total; i += stride)
		out[i] = u01(seed, (unsigned long long)(ctr0 + i));
}
void launch_synth_uniform(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, hipStream_t st) {
	long long total = (long long)n_rows * d;
	if (total <= 0)
		return;
	long long blocks = (total + 255) / 256;
	if (blocks > 256 * 32)
		blocks = 256 * 32;
	hipLaunchKernelGGL(synth_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_out, total,
	                   (unsigned long long)seed, (long long)row0 * d);
	MVS_HIP(hipGetLastError());
}
__device__ __forceinline__ float ih4(unsigned long long seed, unsigned long long ctr) {
	float a = u01(seed, 4 * ctr), b = u01(seed, 4 * ctr + 1), c = u01(seed, 4 * ctr + 2), e = u01(seed, 4 * ctr + 3);
	return __fmul_rn(__fsub_rn(__fadd_rn(__fadd_rn(a, b), __fadd_rn(c, e)), 2.0f), 1.7320508f);
}
__global__ void synth_clustered_kernel(float *out, long long n_rows, int d, unsigned long long seed, long long row0,
                                       int n_centers, float sigma) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	long long stride = (long long)gridDim.x * blockDim.x;
	long long total = n_rows * d;
	for (; i < total; i += stride) {
		long long r = i / d;
		int col = (int)(i - r * d);
		unsigned long long row = (unsigned long long)(row0 + r);
		unsigned long long c =
		    splitmix(seed ^ (row * 0xD1B54A32D192ED03ull + 0x5851F42D4C957F2Dull)) % (unsigned long long)n_centers;
		float centre = ih4(0xC0FFEEull, c * (unsigned long long)d + (unsigned long long)col);
		float g = ih4(seed, row * (unsigned long long)d + (unsigned long long)col);
		out[i] = fmaf(sigma, g, centre);
	}
}
void launch_synth_clustered(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers,
                            float sigma, hipStream_t st) {
	long long total = (long long)n_rows * d;
	if (total <= 0)
		return;
	long long blocks = (total + 255) / 256;
	if (blocks > 256 * 32)
		blocks = 256 * 32;
	hipLaunchKernelGGL(synth_clustered_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_out, (long long)n_rows, d,
	                   (unsigned long long)seed, (long long)row0, n_centers, sigma);
	MVS_HIP(hipGetLastError());
}


// ---- diagnostics: ONE v_mfma_f32_16x16x32_bf16 per wavefront on caller-supplied tiles (round 5) ------------------------------------
// The coarse filters' error bound has one MODELLED term: what the bf16 MFMA's internal accumulation of 32 products + C can deviate
// from the exact sum (csrc/flat_collect.hip: 4 ulp-units of the magnitudes per instruction x 1.25).  tests/test_mfma_model_gpu.py
// feeds this kernel adversarial tiles and compares with the exact sum: the instruction itself, nothing around it.
// A: [ntiles][16 rows][32 k] bf16 bits, Bt: [ntiles][16 columns][32 k] bf16 bits, C / D: [ntiles][16 rows][16 columns] f32.
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void mfma_bf16_probe_kernel(const unsigned short *__restrict__ A, const unsigned short *__restrict__ Bt,
                                                            const float *__restrict__ C, float *__restrict__ D) {
	const size_t t = blockIdx.x;
	const int l = threadIdx.x, rc = l & 15, g = l >> 4;
	const probe_bf16x8 a = *(const probe_bf16x8 *)(A + t * 512 + rc * 32 + 8 * g);  // row rc, k = 8 g .. 8 g + 7
	const probe_bf16x8 b = *(const probe_bf16x8 *)(Bt + t * 512 + rc * 32 + 8 * g); // column rc, the same k
	probe_f32x4 c;
#pragma unroll
	for (int r = 0; r < 4; ++r)
		c[r] = C[t * 256 + (4 * g + r) * 16 + rc]; // rows 4 g + r, column rc
	const probe_f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; ++r)
		D[t * 256 + (4 * g + r) * 16 + rc] = d[r];
}
void launch_mfma_bf16_probe(const unsigned short *d_A, const unsigned short *d_Bt, const float *d_C, float *d_D, int64_t ntiles, hipStream_t st) {
	if (ntiles <= 0)
		return;
	hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3((unsigned)ntiles), dim3(64), 0, st, d_A, d_Bt, d_C, d_D);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
