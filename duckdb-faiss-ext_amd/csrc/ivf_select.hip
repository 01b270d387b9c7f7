// csrc/ivf_select.hip -- IVF list scan for k beyond the LDS k-lists of the scan kernels (k > 256).
// The reference's harness reaches this: its post-filter runs ask an IVF2048 index for the ~2 000 nearest rows so that 10
// survive a 1 % filter (go/main_test.go:17-45 requiredN; src/faiss_extension.cpp:631 passes k straight through).
// FAISS [IndexIVF::search_preassigned + IVFFlatScanner::scan_codes, faiss/IndexIVFFlat.cpp] pushes every row of the
// probed lists through one k-heap per query.  With k in the thousands a heap (or any per-lane list) is the wrong device
// structure: here every (query, probed list) pair writes ALL its distances -- the scanner's per-pair arithmetic, the same
// chains as ivf_scan_kernel / flat_direct_kernel MODE_L2_PAIR -- as 64-bit keys (order-preserving distance bits : row
// position), one rocPRIM segmented radix sort orders every query's candidates, and the first k are decoded.  The order
// is the one the k-list kernels + merge_items_kernel produce: (distance asc | score desc, then position in the list-sorted
// row store asc).  HBM-bound on the list rows (read once per probing query) + 2 x 8 B per candidate per sort pass.
#include "index.h"

#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace mvs {

namespace {

__device__ __forceinline__ bool sel_member_sel(const SelectorDev &s, long long id) {
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
__device__ __forceinline__ unsigned f2key(float f) { // unsigned order == float order
	const unsigned b = __float_as_uint(f);
	return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
	return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

constexpr unsigned long long EMPTY_KEY = ~0ull; // rejected by the selector: sorts behind every candidate

template <bool IS_L2>
__global__ __launch_bounds__(256) void ivf_all_distances_kernel(const float *__restrict__ xq, int dp,
                                                                const float *__restrict__ rows,
                                                                const long long *__restrict__ rowids,
                                                                const IvfSelectPair *__restrict__ pairs, SelectorDev sel,
                                                                const long long *__restrict__ idmap,
                                                                unsigned long long *__restrict__ keys) {
	extern __shared__ __attribute__((aligned(16))) float xs[];
	const IvfSelectPair p = pairs[blockIdx.x];
	for (int i = threadIdx.x; i < dp; i += 256)
		xs[i] = xq[(size_t)p.q * dp + i];
	__syncthreads();
	const int nc = dp >> 2;
	for (int r = threadIdx.x; r < p.len; r += 256) {
		const long long row = (long long)p.row_begin + r;
		const float4 *y = (const float4 *)(rows + (size_t)row * dp);
		float s = 0.f;
		for (int c = 0; c < nc; ++c) {
			const float4 v = y[c];
			const float4 x = ((const float4 *)xs)[c];
			if (IS_L2) { // IVFFlatScanner -> fvec_L2sqr: the k-ordered chain of (x-y)^2 (oracle l2_chain)
				float t = x.x - v.x;
				s = fmaf(t, t, s);
				t = x.y - v.y;
				s = fmaf(t, t, s);
				t = x.z - v.z;
				s = fmaf(t, t, s);
				t = x.w - v.w;
				s = fmaf(t, t, s);
			} else {
				s = fmaf(x.x, v.x, s);
				s = fmaf(x.y, v.y, s);
				s = fmaf(x.z, v.z, s);
				s = fmaf(x.w, v.w, s);
			}
		}
		bool ok = true;
		if (sel.kind != MVS_SEL_NONE) {
			const long long lab = rowids[row];
			ok = sel_member_sel(sel, idmap ? idmap[lab] : lab);
		}
		const unsigned vk = IS_L2 ? f2key(s) : ~f2key(s);
		keys[p.out + r] = ok ? ((unsigned long long)vk << 32) | (unsigned)row : EMPTY_KEY;
	}
}

template <bool IS_L2>
__global__ __launch_bounds__(256) void ivf_select_out_kernel(const unsigned long long *__restrict__ sorted,
                                                             const int *__restrict__ seg, int k,
                                                             const long long *__restrict__ rowids,
                                                             const long long *__restrict__ idmap, float *__restrict__ D,
                                                             long long *__restrict__ I) {
	const long long q = blockIdx.x;
	const int b = seg[q], e = seg[q + 1];
	for (int j = threadIdx.x; j < k; j += 256) {
		const unsigned long long key = j < e - b ? sorted[(size_t)b + j] : EMPTY_KEY;
		float v = IS_L2 ? FLT_MAX : -FLT_MAX;
		long long lab = -1;
		if (key != EMPTY_KEY) {
			const unsigned vk = (unsigned)(key >> 32);
			v = key2f(IS_L2 ? vk : ~vk);
			lab = rowids ? rowids[(unsigned)key] : (long long)(unsigned)key; // rowids == nullptr: positions
			if (idmap)
				lab = idmap[lab];
		}
		D[q * k + j] = v;
		I[q * k + j] = lab;
	}
}

} // namespace

size_t ivf_select_temp_bytes(int64_t total, int64_t nseg) {
	size_t bytes = 0;
	MVS_HIP(rocprim::segmented_radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
	                                           (unsigned)total, (unsigned)nseg, (const int *)nullptr, (const int *)nullptr, 0,
	                                           64, (hipStream_t) nullptr));
	return bytes;
}

// pairs: device array of npairs (query, list) work items whose `out` offsets tile [0,total); seg: [nseg+1] offsets of the
// queries' candidate ranges; keys_a / keys_b: total keys each; D / I: [nseg][k] (already offset to the chunk's first query)
void launch_ivf_select(int metric, const float *d_xq, int dp, const float *d_rows, const int64_t *d_rowids,
                       const IvfSelectPair *d_pairs, int npairs, const int *d_seg, int64_t nseg, int64_t total, int64_t k,
                       SelectorDev sel, const int64_t *d_idmap_sel, const int64_t *d_idmap_out, unsigned long long *keys_a,
                       unsigned long long *keys_b, void *d_temp, size_t temp_bytes, float *d_D, int64_t *d_I,
                       hipStream_t st, bool raw_positions) {
	const bool is_l2 = metric_order(metric) == METRIC_L2;
	if (npairs > 0 && total > 0) {
		const size_t lds = (size_t)dp * sizeof(float);
		if (is_l2)
			hipLaunchKernelGGL(ivf_all_distances_kernel<true>, dim3((unsigned)npairs), dim3(256), lds, st, d_xq, dp, d_rows,
			                   (const long long *)d_rowids, d_pairs, sel, (const long long *)d_idmap_sel, keys_a);
		else
			hipLaunchKernelGGL(ivf_all_distances_kernel<false>, dim3((unsigned)npairs), dim3(256), lds, st, d_xq, dp, d_rows,
			                   (const long long *)d_rowids, d_pairs, sel, (const long long *)d_idmap_sel, keys_a);
		MVS_HIP(hipGetLastError());
		MVS_HIP(rocprim::segmented_radix_sort_keys(d_temp, temp_bytes, keys_a, keys_b, (unsigned)total, (unsigned)nseg, d_seg,
		                                           d_seg + 1, 0, 64, st));
	}
	if (is_l2)
		hipLaunchKernelGGL(ivf_select_out_kernel<true>, dim3((unsigned)nseg), dim3(256), 0, st, keys_b, d_seg, (int)k,
		                   raw_positions ? nullptr : (const long long *)d_rowids, (const long long *)d_idmap_out, d_D, (long long *)d_I);
	else
		hipLaunchKernelGGL(ivf_select_out_kernel<false>, dim3((unsigned)nseg), dim3(256), 0, st, keys_b, d_seg, (int)k,
		                   raw_positions ? nullptr : (const long long *)d_rowids, (const long long *)d_idmap_out, d_D, (long long *)d_I);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
