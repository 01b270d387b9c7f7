// csrc/kmeans_update.hip -- Clustering::train's compute_centroids step on device (faiss/Clustering.cpp).
//
// FAISS sums, per centroid, the points assigned to it IN INPUT ORDER (each OpenMP thread owns a range of centroids
// and walks all points front to back), counts them in a float and multiplies by 1/count.  fp32 addition is not
// associative, so the device keeps that order exactly: a STABLE radix sort of (assignment, point index) groups the
// points per centroid in input order (rocPRIM), then one thread per (centroid, dimension) adds its segment
// sequentially -- neighbouring threads read neighbouring dimensions of the same point, so the reads coalesce.
// Centroids come out bit-identical to oracle/orc_core.c kmeans_train.
#include "common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace mvs {

namespace {

// A label outside [0,k) (the k = 1 assignment search returns -1 when no finite distance exists, e.g. an overflow to
// inf) is filed under the extra key k: it sorts behind every real cluster and is not counted, so the caller sees
// sum(hassign) < nx and raises FAISS's "ci >= 0 && ci < k" assertion instead of writing out of bounds.
__global__ void km_keys_kernel(const long long *lab, int n, int k, int *keys, int *vals, int *cnt) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const long long c = lab[i];
	vals[i] = i;
	if (c < 0 || c >= k) {
		keys[i] = k;
		return;
	}
	keys[i] = (int)c;
	atomicAdd(&cnt[c], 1);
}
// exclusive scan of cnt[0..k) into off[0..k] (single workgroup; k is the number of centroids)
__global__ __launch_bounds__(1024) void km_scan_kernel(const int *cnt, int k, int *off) {
	__shared__ int part[1024];
	const int t = threadIdx.x;
	const int per = (k + 1023) / 1024;
	const int c0 = t * per, c1 = min(k, c0 + per);
	int s = 0;
	for (int c = c0; c < c1; ++c)
		s += cnt[c];
	part[t] = s;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		const int v = t >= o ? part[t - o] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	int b = part[t] - s;
	for (int c = c0; c < c1; ++c) {
		off[c] = b;
		b += cnt[c];
	}
	if (t == 1023)
		off[k] = part[1023];
}
__global__ void km_sum_kernel(const float *x, const int *order, const int *off, int k, int d, float *cent, float *hassign) {
	const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= (long long)k * d)
		return;
	const int c = (int)(g / d), j = (int)(g - (long long)c * d);
	const int b = off[c], e = off[c + 1];
	float s = 0.f;
	for (int p = b; p < e; ++p)
		s += x[(size_t)order[p] * d + j];
	const float n = (float)(e - b); // hassign[ci] += 1.0f per point: exact below 2^24
	if (e > b)
		s *= 1 / n;
	cent[g] = s;
	if (j == 0)
		hassign[c] = n;
}

} // namespace

size_t kmeans_update_ws_bytes(int64_t nx, int64_t k) {
	size_t temp = 0;
	(void)rocprim::radix_sort_pairs(nullptr, temp, (int *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr,
	                                (size_t)nx, 0, 32, nullptr);
	return (size_t)nx * 4 * sizeof(int) + (size_t)(2 * k + 2) * sizeof(int) + temp + 256;
}

// d_assign: [nx] labels in [0,k).  Outputs d_cent [k][d] and d_hassign [k] (float counts, 0 = empty cluster).
void launch_kmeans_update(const float *d_x, int64_t nx, int d, const int64_t *d_assign, int64_t k, float *d_cent,
                          float *d_hassign, void *ws, size_t ws_bytes, hipStream_t st) {
	int *keys = (int *)ws, *vals = keys + nx, *keys_s = vals + nx, *vals_s = keys_s + nx;
	int *cnt = vals_s + nx, *off = cnt + k;
	char *temp = (char *)(off + k + 2);
	temp += (256 - ((uintptr_t)temp & 255)) & 255;
	size_t temp_bytes = ws_bytes - (size_t)(temp - (char *)ws);
	MVS_HIP(hipMemsetAsync(cnt, 0, (size_t)k * sizeof(int), st));
	hipLaunchKernelGGL(km_keys_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, st, (const long long *)d_assign,
	                   (int)nx, (int)k, keys, vals, cnt);
	int bits = 1;
	while (((int64_t)1 << bits) <= k) // keys run over [0, k]: k = "label out of range"
		bits++;
	MVS_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_s, vals, vals_s, (size_t)nx, 0, bits, st));
	hipLaunchKernelGGL(km_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, (int)k, off);
	const long long tot = (long long)k * d;
	hipLaunchKernelGGL(km_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_x, vals_s, off, (int)k, d,
	                   d_cent, d_hassign);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
