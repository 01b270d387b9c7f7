// csrc/ivf_ties.hip -- FAISS's heap outcome for IndexIVF::search under EXACT distance ties.
//
// IVFFlatScanner::scan_codes (faiss/IndexIVFFlat.cpp, reached from /root/reference/src/faiss_extension.cpp:631 with the
// SearchParametersIVF of :675-689) feeds one heap per query in ARRIVAL order = probe rank, then position in the list, with
// the strict insert rule, the heap comparing (value, stored id) pairs (faiss/utils/Heap.h, ordered_key_value.h cmp2);
// heap_reorder prints L2 as (distance asc, id asc) and inner product as (score desc, id desc).  oracle/orc_core.c
// ivf_search replays exactly that.  The scan kernels keep the PURE order (value, position in the list-sorted store), which
// differs in two places, both only where float values are bit-equal:
//   (a) inside the result, runs of equal values are printed by stored id, not by position;
//   (b) at the k-th value T: with A = the rows not worse than T in arrival order and A_k its first k entries, every row of
//       A_k enters the heap and none is evicted until A_k is complete (the root is worse than T until then); afterwards
//       rows equal to T are rejected and every later row BETTER than T evicts the root = the tied row with the largest
//       stored id (CMax, L2) / the smallest (CMin, inner product).  Hence
//           result = {rows better than T} + {tied rows of A_k minus the G extreme ids},  G = #(rows better than T outside A_k).
//       (csrc/util_kernels.hip tie_resolve_kernel is the Flat case, where arrival order = id order.)
// Every IVF search path therefore emits one entry more than asked for, as (value, position) in the pure order;
// ivf_finish_kernel applies (a), flags the queries whose k-th and (k+1)-th values are bit-equal, and ivf_tie_pass_kernel
// replays (b) for those: it walks the query's probed lists in probe order with the scanner's arithmetic (the same k-ordered
// chain every scan kernel and the exact re-scoring use, so "== T" is meaningful) until A_k is complete.
#include "common.h"

#include <algorithm>
#include "../../include/mi355_faiss.h"

namespace mvs {

namespace {

__device__ __forceinline__ bool tie_sel_member(const SelectorDev &s, long long id) {
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

// pd / pi: [nq][kx] pure lists (value, position in the list-sorted store; -1 = empty, empties last); D / I: [nq][k]
template <bool IS_L2>
__global__ void ivf_finish_kernel(const float *__restrict__ pd, const long long *__restrict__ pi, int kx, int k, long long nq,
                                  const long long *__restrict__ rowids, const long long *__restrict__ idmap,
                                  float *__restrict__ D, long long *__restrict__ I, int *__restrict__ flag_cnt,
                                  int *__restrict__ flag_q) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq * k)
		return;
	const long long q = i / k;
	const int j = (int)(i - q * k);
	const float *v = pd + q * kx;
	const long long *p = pi + q * kx;
	if (p[j] < 0) {
		D[i] = IS_L2 ? FLT_MAX : -FLT_MAX;
		I[i] = -1;
		return;
	}
	const float val = v[j];
	const long long id = rowids[p[j]];
	int a = j, b = j;
	while (a > 0 && v[a - 1] == val)
		--a;
	while (b + 1 < k && p[b + 1] >= 0 && v[b + 1] == val)
		++b;
	int rank = 0;
	for (int m = a; m <= b && b > a; ++m) {
		const long long idm = rowids[p[m]];
		rank += IS_L2 ? (idm < id || (idm == id && m < j)) : (idm > id || (idm == id && m < j));
	}
	const long long o = q * k + a + rank;
	D[o] = val;
	I[o] = idmap ? idmap[id] : id;
	if (j == k - 1 && kx > k && p[k] >= 0 && v[k] == val)
		flag_q[atomicAdd(flag_cnt, 1)] = (int)q;
}

// one workgroup per flagged query (persistent: block f, f + grid, ...); thread <-> row of a probed list
template <bool IS_L2>
__global__ __launch_bounds__(256) void ivf_tie_pass_kernel(const int *__restrict__ flag_cnt, const int *__restrict__ flag_q,
                                                          const float *__restrict__ x, int d, const float *__restrict__ pd,
                                                          int kx, int k, const long long *__restrict__ coarse, int np,
                                                          const long long *__restrict__ list_off,
                                                          const float *__restrict__ codes, int dp,
                                                          const long long *__restrict__ rowids, SelectorDev sel,
                                                          const long long *__restrict__ idmap_sel,
                                                          const long long *__restrict__ idmap_out, float *__restrict__ D,
                                                          long long *__restrict__ I, const float *__restrict__ T_ext,
                                                          float *__restrict__ emit_v, long long *__restrict__ emit_id,
                                                          int *__restrict__ emit_p) {
	// EMIT mode (emit_v != null; row shards, csrc/sharded.hip): T comes from the CROSS-SHARD merge (T_ext[f]) and the kernel only
	// reports this shard's A_k -- value, stored id (= global row) and probe rank of its first k rows not worse than T in arrival
	// order -- for the host to merge by (probe rank, global row) and to apply the closed form on.
	extern __shared__ __attribute__((aligned(16))) float tp_sm[];
	const int dpad = (d + 3) & ~3, kpad = (k + 1) & ~1;
	float *xs = tp_sm;                          // [dpad]
	float *av = xs + dpad;                      // [kpad]  values of A_k, arrival order
	long long *aid = (long long *)(av + kpad);  // [k]     stored ids of A_k
	int *ctl = (int *)(aid + k);                // [0] |A| so far, [1..4] per-wave counts, [5] rows better than T in the pure list,
	                                            // [6] in A_k, [7] tied rows in A_k
	int *ap = ctl + 8;                          // [k]     probe rank of the entries of A_k (EMIT mode)
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int nflag = *flag_cnt;
	for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
		const long long q = flag_q[f];
		__syncthreads();
		for (int t = tid; t < d; t += 256)
			xs[t] = x[q * d + t];
		if (tid < 8)
			ctl[tid] = 0;
		const float T = T_ext ? T_ext[f] : pd[q * kx + k - 1];
		__syncthreads();
		// rows strictly better than T in the pure top-k (all of them are there: fewer than k exist)
		for (int t = tid; t < k && !emit_v; t += 256)
			if (IS_L2 ? pd[q * kx + t] < T : pd[q * kx + t] > T)
				atomicAdd(&ctl[5], 1);
		bool full = false;
		for (int p = 0; p < np && !full; ++p) {
			const long long l = coarse[q * np + p];
			if (l < 0)
				continue; // fewer than nprobe centroids
			const long long rb = list_off[l], re = list_off[l + 1];
			for (long long r0 = rb; r0 < re && !full; r0 += 256) {
				const long long r = r0 + tid;
				bool ok = false;
				float val = 0.f;
				long long id = -1;
				if (r < re) {
					id = rowids[r];
					if (sel.kind == MVS_SEL_NONE || tie_sel_member(sel, idmap_sel ? idmap_sel[id] : id)) {
						const float *y = codes + (size_t)r * dp;
						float acc = 0.f;
						int kk = 0;
						for (; kk + 4 <= d; kk += 4) {
							const float4 yv = *(const float4 *)(y + kk);
							const float4 xv = *(const float4 *)(xs + kk);
							const float ys[4] = {yv.x, yv.y, yv.z, yv.w}, xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
							for (int e = 0; e < 4; ++e) {
								if (IS_L2) {
									const float t = __fsub_rn(xq[e], ys[e]);
									acc = fmaf(t, t, acc);
								} else {
									acc = fmaf(xq[e], ys[e], acc);
								}
							}
						}
						for (; kk < d; ++kk) {
							if (IS_L2) {
								const float t = __fsub_rn(xs[kk], y[kk]);
								acc = fmaf(t, t, acc);
							} else {
								acc = fmaf(xs[kk], y[kk], acc);
							}
						}
						val = acc;
						// what the heap can hold at all (the scan kernels drop the rest as EMPTY), not worse than T
						ok = IS_L2 ? (acc < FLT_MAX && acc <= T) : (acc > -FLT_MAX && acc >= T);
					}
				}
				const unsigned long long bal = __builtin_amdgcn_ballot_w64(ok);
				if (lane == 0)
					ctl[1 + wave] = __popcll(bal);
				__syncthreads();
				int base = ctl[0];
				for (int w = 0; w < wave; ++w)
					base += ctl[1 + w];
				const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
				if (ok && pos < k) {
					av[pos] = val;
					aid[pos] = id;
					if (emit_v)
						ap[pos] = p;
				}
				const int tot = ctl[0] + ctl[1] + ctl[2] + ctl[3] + ctl[4];
				__syncthreads();
				if (tid == 0)
					ctl[0] = tot;
				full = tot >= k;
			}
		}
		__syncthreads();
		const int na = ctl[0] < k ? ctl[0] : k;
		if (emit_v) {
			for (int t = tid; t < k; t += 256) {
				emit_v[(size_t)f * k + t] = t < na ? av[t] : 0.f;
				emit_id[(size_t)f * k + t] = t < na ? aid[t] : -1ll;
				emit_p[(size_t)f * k + t] = t < na ? ap[t] : -1;
			}
			continue;
		}
		for (int t = tid; t < na; t += 256) {
			if (av[t] == T)
				atomicAdd(&ctl[7], 1);
			else
				atomicAdd(&ctl[6], 1);
		}
		__syncthreads();
		const int nbetter = ctl[5], ntied = ctl[7];
		const int G = nbetter - ctl[6]; // rows better than T that arrived after A_k was complete
		// slots [0, nbetter) of D / I already hold the rows better than T in print order (ivf_finish_kernel); the tied rows
		// of A_k follow by stored id, without the G evicted ones
		for (int t = tid; t < na; t += 256) {
			if (av[t] != T)
				continue;
			const long long id = aid[t];
			int rnk = 0;
			for (int m = 0; m < na; ++m)
				rnk += av[m] == T && (aid[m] < id || (aid[m] == id && m < t));
			int slot;
			if (IS_L2) { // the G largest ids were evicted; ascending id
				if (rnk >= ntied - G)
					continue;
				slot = nbetter + rnk;
			} else { // the G smallest ids were evicted; descending id
				if (rnk < G)
					continue;
				slot = nbetter + (ntied - 1 - rnk);
			}
			if (slot < k) {
				D[q * k + slot] = T;
				I[q * k + slot] = idmap_out ? idmap_out[id] : id;
			}
		}
	}
}

__global__ void ivf_mf_to_csr_kernel(long long *I, long long total, const int *__restrict__ perm_mf) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total && I[i] >= 0)
		I[i] = perm_mf[I[i]];
}

} // namespace

// positions in the padded MFMA list store -> positions in the list-sorted (CSR) store
void launch_ivf_mf_to_csr(int64_t *d_I, int64_t total, const int *d_perm_mf, hipStream_t st) {
	if (total <= 0)
		return;
	hipLaunchKernelGGL(ivf_mf_to_csr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (long long *)d_I,
	                   (long long)total, d_perm_mf);
	MVS_HIP(hipGetLastError());
}

// d_pd / d_pi: [nq][kx] pure lists with kx = k + 1 (kx = k: no boundary detection); d_flag: [1 + nq] ints {count, queries...}
void launch_ivf_finish(int metric, const float *d_pd, const int64_t *d_pi, int64_t nq, int kx, int k, const int64_t *d_rowids,
                       const int64_t *d_idmap_out, float *d_D, int64_t *d_I, int *d_flag, hipStream_t st) {
	if (nq <= 0 || k <= 0)
		return;
	MVS_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
	const long long tot = (long long)nq * k;
	const dim3 grid((unsigned)((tot + 255) / 256));
	if (metric_order(metric) == METRIC_L2)
		hipLaunchKernelGGL(ivf_finish_kernel<true>, grid, dim3(256), 0, st, d_pd, (const long long *)d_pi, kx, k, (long long)nq,
		                   (const long long *)d_rowids, (const long long *)d_idmap_out, d_D, (long long *)d_I, d_flag, d_flag + 1);
	else
		hipLaunchKernelGGL(ivf_finish_kernel<false>, grid, dim3(256), 0, st, d_pd, (const long long *)d_pi, kx, k, (long long)nq,
		                   (const long long *)d_rowids, (const long long *)d_idmap_out, d_D, (long long *)d_I, d_flag, d_flag + 1);
	MVS_HIP(hipGetLastError());
}

// the tie pass keeps the query (d floats) and A_k (k values + k ids) of a flagged query in LDS
static size_t ivf_tie_pass_lds(int d, int k) {
	const int dpad = (d + 3) & ~3, kpad = (k + 1) & ~1;
	return (size_t)(dpad + kpad) * 4 + (size_t)k * 8 + 64 + (size_t)k * 4;
}
bool ivf_tie_pass_fits(int d, int64_t k) {
	return k < ((int64_t)1 << 24) && ivf_tie_pass_lds(d, (int)k) <= 150 * 1024;
}
void launch_ivf_tie_pass(int metric, const int *d_flag, int64_t nq, const float *d_x, int d, const float *d_pd, int kx, int k,
                         const int64_t *d_coarse, int np, const int64_t *d_list_off, const float *d_codes, int dp,
                         const int64_t *d_rowids, SelectorDev sel, const int64_t *d_idmap_sel, const int64_t *d_idmap_out,
                         float *d_D, int64_t *d_I, hipStream_t st) {
	if (nq <= 0 || kx <= k)
		return;
	const size_t lds = ivf_tie_pass_lds(d, k);
	if (lds > 150 * 1024) // (callers gate on ivf_tie_pass_fits: such k keep the pure order)
		throw_faiss(__func__, __FILE__, "IVF tie pass: k = %d too large", k);
	// no host round trip: a fixed grid of persistent workgroups, each takes flagged queries f, f + grid, ... (usually none)
	const unsigned grid = (unsigned)std::min<int64_t>(nq, 2048);
	if (metric_order(metric) == METRIC_L2) {
		auto kern = ivf_tie_pass_kernel<true>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, d_flag, d_flag + 1, d_x, d, d_pd, kx, k,
		                   (const long long *)d_coarse, np, (const long long *)d_list_off, d_codes, dp,
		                   (const long long *)d_rowids, sel, (const long long *)d_idmap_sel, (const long long *)d_idmap_out, d_D,
		                   (long long *)d_I, nullptr, nullptr, nullptr, nullptr);
	} else {
		auto kern = ivf_tie_pass_kernel<false>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, d_flag, d_flag + 1, d_x, d, d_pd, kx, k,
		                   (const long long *)d_coarse, np, (const long long *)d_list_off, d_codes, dp,
		                   (const long long *)d_rowids, sel, (const long long *)d_idmap_sel, (const long long *)d_idmap_out, d_D,
		                   (long long *)d_I, nullptr, nullptr, nullptr, nullptr);
	}
	MVS_HIP(hipGetLastError());
}

// EMIT mode for a row shard: d_flag = {count, queries...} of the flagged queries, d_T their boundary values from the cross-shard
// merge; out: [nf][k] value / stored id / probe rank of this shard's first k rows not worse than T in arrival order (-1 padded)
void launch_ivf_tie_emit(int metric, const int *d_flag, int nf, const float *d_x, int d, const float *d_T, int k,
                         const int64_t *d_coarse, int np, const int64_t *d_list_off, const float *d_codes, int dp,
                         const int64_t *d_rowids, SelectorDev sel, const int64_t *d_idmap_sel, float *d_emit_v, int64_t *d_emit_id,
                         int *d_emit_p, hipStream_t st) {
	if (nf <= 0)
		return;
	const size_t lds = ivf_tie_pass_lds(d, k);
	if (lds > 150 * 1024)
		throw_faiss(__func__, __FILE__, "IVF tie pass: k = %d too large", k);
	const unsigned grid = (unsigned)std::min<int>(nf, 2048);
	if (metric_order(metric) == METRIC_L2) {
		auto kern = ivf_tie_pass_kernel<true>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, d_flag, d_flag + 1, d_x, d, (const float *)nullptr, k + 1, k,
		                   (const long long *)d_coarse, np, (const long long *)d_list_off, d_codes, dp, (const long long *)d_rowids, sel,
		                   (const long long *)d_idmap_sel, (const long long *)nullptr, (float *)nullptr, (long long *)nullptr, d_T,
		                   d_emit_v, (long long *)d_emit_id, d_emit_p);
	} else {
		auto kern = ivf_tie_pass_kernel<false>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, d_flag, d_flag + 1, d_x, d, (const float *)nullptr, k + 1, k,
		                   (const long long *)d_coarse, np, (const long long *)d_list_off, d_codes, dp, (const long long *)d_rowids, sel,
		                   (const long long *)d_idmap_sel, (const long long *)nullptr, (float *)nullptr, (long long *)nullptr, d_T,
		                   d_emit_v, (long long *)d_emit_id, d_emit_p);
	}
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
