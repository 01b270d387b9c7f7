// csrc/flat_collect.hip -- bf16 COARSE FILTER for the brute-force search: one bf16 MFMA per 16 dimensions, candidates by a
// proven bound, exact f32 re-scoring.
//
// Same place in the path as flat_mfma.hip / flat_bf16.hip (IndexFlat::search, /root/reference/src/faiss_extension.cpp:631,
// BLAS branch of knn_L2sqr / knn_inner_product) and the same results bit for bit.  flat_bf16.hip keeps the k' best
// APPROXIMATE values (three bf16 products per element pair, error ~2^-18) and proves afterwards that they contain the
// exact top-k.  This file turns the argument round so that ONE product per pair (error ~2^-8) is enough:
//
//   coarse value      s(q, row) = 2 <bf(x'), bf(y')> - ||y'||^2  (L2; the larger the nearer)  or  <bf(x'), bf(y')> + <mu, y>
//                     (inner product); x' = x - mu, y' = y - mu (mu = mean row: the error scales with ||x'|| ||y'||, distances
//                     do not), bf = bf16 rounding, f32 accumulation on the matrix pipe
//   error bound       |s - s_exact| <= E(q) for every row, E from ||x'||, max ||y'||, ... and d (collect_bounds_kernel)
//   running bound     B(q) = the kk-th best of the best s seen in each of 16 row classes (row id mod 16), shared by all
//                     workgroups through the threshold slots of DESIGN.md 3.3: kk DISTINCT rows have s >= B, so the exact
//                     kk-th best value is no worse than B - E
//   candidates        every row with s >= B - 2E at the time it is scanned (B only rises) -- a row of the exact top-kk has
//                     s >= (its exact value) - E >= (B - E) - E.  No row of the result can be missed; ties at the k-th value
//                     are ALL candidates, so FAISS's (value, id) order is applied to exact values only.
//   re-scoring        the candidates (a few hundred per query out of 10^7 rows) are grouped by query (one radix sort),
//                     re-computed with the oracle's arithmetic (k-ordered fmaf chain over the f32 row, (xn + yn) - 2 ip,
//                     clamp) and the kk best per query go through the normal merge: labels AND distances are those of
//                     flat_mfma.hip / oracle/orc_core.c search_blas.
//
// No k-lists in the scan kernel at all: the epilogue is one fma + half a max3 per value and a compare per 16 values; the
// rare path appends (query, row) to a workgroup queue in LDS that is flushed to a global stream with one atomic per
// ~1000 candidates.  Queries whose bound is not finite (NaN / Inf / overflowing norms) are re-run on the exact kernel.
//
// Geometry (CDNA4): workgroup = 4 waves x 128 queries (four 32-query B tiles resident as bf16: 128 VGPRs), database tiles
// of 32 rows x 256 B (bf16, 16-byte chunks XOR-swizzled by row) double-buffered by LDS-DMA, one ds_read_b128 per 4 MFMAs;
// two workgroups per CU, i.e. 1024 queries share every byte a CU pulls out of L2.
#include "flat_collect.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace mvs {

// ---- storage: CENTRED rows as bf16 (round to nearest even) + one f32 per row -----------------------------------------------
// The error of a bf16 product scales with ||x|| ||y||, distances do not: rows and queries are shifted by mu (the mean of the
// rows present when the store is first built; ANY vector is valid, it only has to be the same for rows and queries):
//   L2:  ||x - y||^2 = ||x'||^2 + ||y'||^2 - 2 <x', y'>,          x' = x - mu, y' = y - mu     beta(row) = -||y'||^2, alpha = 2
//   IP:  <x, y> = <x', y'> + <mu, y> + <x', mu>  (last term: per query)                        beta(row) = <mu, y>,  alpha = 1
// coarse value s = alpha <bf16(x'), bf16(y')> + beta(row).
__global__ __launch_bounds__(256) void collect_colsum_kernel(const float *__restrict__ src, long long nrows, int dp,
                                                            int interleaved, float *__restrict__ sum) {
	// block = (dp / 8 column groups) x (256 / (dp / 8) row lanes); 1024 rows per block
	const int g8 = dp / 8, rl = threadIdx.x / g8, c8 = threadIdx.x % g8, nrl = 256 / g8;
	float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	const long long r0 = (long long)blockIdx.x * 1024;
	// (round 6: the block's row lanes meet in LDS and ONE lane per column sends the atomic -- 2 048 float atomics per block on 128
	// addresses were 3.2 ms of a first search for a 0.5 GB read)
	__shared__ float red[256 * 8];
	const bool idle = rl >= nrl; // dp / 8 does not divide 256: the leftover threads would count rows twice
	for (long long r = r0 + rl; !idle && r < r0 + 1024 && r < nrows; r += nrl) {
		const float4 s0 = *(const float4 *)(src + (size_t)r * dp + c8 * 8);
		const float4 s1 = *(const float4 *)(src + (size_t)r * dp + c8 * 8 + 4);
		float v[8];
		if (!interleaved) {
			v[0] = s0.x, v[1] = s0.y, v[2] = s0.z, v[3] = s0.w, v[4] = s1.x, v[5] = s1.y, v[6] = s1.z, v[7] = s1.w;
		} else if ((r >> 4) & 1) {
			v[0] = s0.z, v[1] = s0.x, v[2] = s0.w, v[3] = s0.y, v[4] = s1.z, v[5] = s1.x, v[6] = s1.w, v[7] = s1.y;
		} else {
			v[0] = s0.x, v[1] = s0.z, v[2] = s0.y, v[3] = s0.w, v[4] = s1.x, v[5] = s1.z, v[6] = s1.y, v[7] = s1.w;
		}
#pragma unroll
		for (int e = 0; e < 8; ++e)
			acc[e] += v[e];
	}
#pragma unroll
	for (int e = 0; e < 8; ++e)
		red[threadIdx.x * 8 + e] = idle ? 0.f : acc[e];
	__syncthreads();
	if (rl == 0) { // (thread c8: the column group's first row lane)
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			float t = 0.f;
			for (int l = 0; l < nrl; ++l)
				t += red[(l * g8 + c8) * 8 + e];
			atomicAdd(sum + c8 * 8 + e, t);
		}
	}
}
__global__ void collect_mean_kernel(float *sum, int dp, int d, float inv_n) {
	for (int i = threadIdx.x; i < dp; i += blockDim.x)
		sum[i] = i < d ? sum[i] * inv_n : 0.f;
}
// mu[dp] <- column means of the first `nrows` rows (padded dimensions: 0)
void launch_collect_mean(const FlatGeom &g, const float *d_vecs, int64_t nrows, float *d_mu, hipStream_t st) {
	MVS_HIP(hipMemsetAsync(d_mu, 0, (size_t)std::max(g.dp, 1024) * sizeof(float), st)); // (sized for the widest bf16 store)
	if (nrows <= 0)
		return;
	hipLaunchKernelGGL(collect_colsum_kernel, dim3((unsigned)((nrows + 1023) / 1024)), dim3(256), 0, st, d_vecs,
	                   (long long)nrows, g.dp, g.pair_interleaved ? 1 : 0, d_mu);
	hipLaunchKernelGGL(collect_mean_kernel, dim3(1), dim3(256), 0, st, d_mu, g.dp, g.d, 1.0f / (float)nrows);
	MVS_HIP(hipGetLastError());
}

template <int CTRL>
__device__ __forceinline__ float cl_dpp(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// ---- outlier rows (round 6) --------------------------------------------------------------------------------
// E(q) scales with the LARGEST ||y'|| of the store: one row of 100 x the usual norm made every query admit thousands of rows (the
// stream overflowed, the batch fell to the bf16x3 / exact kernels: 17.5 -> 602 ms per batch at the headline, tools/collect_sensitivity.py
// "outlier").  A row whose ||y'||^2 exceeds tau = 64 x the mean ||y'||^2 of the rows present when the store is first built (at most
// CL_OUTL_CAP of them) is kept OUT of the store -- zero vector, beta = -inf: never a candidate, never evidence for a bound, not in the
// maxima -- and IN every query's candidate set: collect_append_outliers_kernel puts (query, row) for every outlier row behind the scan's
// entries in the candidate stream, so the exact stage scores it like any other candidate.  max_bits[2] = the largest ||y||^2 over the
// rows that ARE in the store (the bound's S term), max_bits[3] = tau (float bits; +inf: no outlier handling -- small stores, the IVF
// quantiser's centroids); outl[0] = count, outl[1 ..] = rows.
__global__ void collect_outlier_threshold_kernel(const float *__restrict__ norms, long long nrows, const float *__restrict__ mu, int dp,
                                                 unsigned *__restrict__ max_bits) {
	__shared__ double part[256];
	double acc = 0.0;
	// (a mean: every 16th row of a large sample says the same -- one workgroup walking a million norms was 1.7 ms of a first search)
	const long long step = nrows >= 65536 ? 16 : 1;
	long long cnt = 0;
	for (long long i = (long long)threadIdx.x * step; i < nrows; i += 256 * step) {
		acc += (double)norms[i];
		++cnt;
	}
	__shared__ long long pcnt[256];
	pcnt[threadIdx.x] = cnt;
	part[threadIdx.x] = acc;
	__syncthreads();
	for (int o = 128; o >= 1; o >>= 1) {
		if (threadIdx.x < o) {
			part[threadIdx.x] += part[threadIdx.x + o];
			pcnt[threadIdx.x] += pcnt[threadIdx.x + o];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		nrows = pcnt[0];
		double mun = 0.0;
		for (int k = 0; k < dp; ++k)
			mun += (double)mu[k] * (double)mu[k];
		// mean ||y - mu||^2 over the rows mu was averaged over = mean ||y||^2 - ||mu||^2
		const double mean_c = part[0] / (double)(nrows > 0 ? nrows : 1) - mun;
		const float tau = mean_c > 0.0 && mean_c < 1e30 ? (float)(64.0 * mean_c) : INFINITY; // (degenerate data: no outlier handling)
		max_bits[3] = __float_as_uint(tau);
	}
}
void launch_collect_outlier_threshold(const float *d_norms, int64_t nrows, const float *d_mu, int dp, unsigned *d_max_norm_bits, hipStream_t st) {
	hipLaunchKernelGGL(collect_outlier_threshold_kernel, dim3(1), dim3(256), 0, st, d_norms, (long long)nrows, d_mu, dp, d_max_norm_bits);
	MVS_HIP(hipGetLastError());
}
// (query, outlier row) for every query and every recorded outlier row behind the scan's entries; value +inf: they pass every
// final-bound filter.  SEL: rows the IDSelector rejects are left out.
__global__ __launch_bounds__(256) void collect_append_outliers_kernel(const int *__restrict__ outl, int nq, unsigned long long *__restrict__ stream,
                                                                      float *__restrict__ stream_s, unsigned long long *__restrict__ cnt,
                                                                      long long cap, const unsigned long long *__restrict__ rowmask) {
	__shared__ unsigned long long base_s;
	const int no = outl[0] < CL_OUTL_CAP ? outl[0] : CL_OUTL_CAP;
	const long long total = (long long)nq * no;
	for (long long b0 = (long long)blockIdx.x * 256; b0 < total; b0 += (long long)gridDim.x * 256) {
		const long long i = b0 + threadIdx.x;
		const int q = (int)(i / no), j = (int)(i - (long long)q * no);
		const unsigned row = i < total ? (unsigned)outl[1 + j] : 0u;
		const bool ok = i < total && (!rowmask || ((rowmask[row >> 6] >> (row & 63u)) & 1ull));
		const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
		__shared__ int wsum[4];
		if ((threadIdx.x & 63) == 0)
			wsum[threadIdx.x >> 6] = __builtin_popcountll(m);
		__syncthreads();
		if (threadIdx.x == 0)
			base_s = atomicAdd(cnt, (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
		__syncthreads();
		int wb = 0;
		for (int w = 0; w < (int)(threadIdx.x >> 6); ++w)
			wb += wsum[w];
		const long long pos = (long long)base_s + wb + __builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
		if (ok && pos < cap) {
			stream[pos] = ((unsigned long long)(unsigned)q << 32) | row;
			if (stream_s)
				stream_s[pos] = INFINITY;
		}
		__syncthreads();
	}
}
void launch_collect_append_outliers(const int *d_outl, int n_outliers, int64_t nq, unsigned long long *d_stream, float *d_stream_s,
                                    unsigned long long *d_cnt, int64_t cap, const unsigned long long *d_rowmask, hipStream_t st) {
	if (n_outliers <= 0 || nq <= 0)
		return;
	const long long total = (long long)nq * std::min(n_outliers, CL_OUTL_CAP);
	hipLaunchKernelGGL(collect_append_outliers_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 1024)), dim3(256), 0, st, d_outl, (int)nq,
	                   d_stream, d_stream_s, d_cnt, (long long)cap, d_rowmask);
	MVS_HIP(hipGetLastError());
}

// one thread per (row, 8 dims); the dp / 8 threads of a row are neighbours in a wave
// max_bits[0]: largest squared norm of the ORIGINAL rows (shared with flat_bf16.hip), max_bits[8]: of the centred rows,
// max_bits[12]: of the centred rows' bf16 rounding residuals y' - bf16(y'); [2], [3], outl: see "outlier rows" above
// (round 6: the maxima are kept in registers over a grid-stride loop and reach max_bits with ONE atomic per wave and maximum at the end.
// Rounds 4-5 compared every row against a plain load of max_bits[..] -- which a CU's vector cache keeps serving as it was when the line
// was first fetched, atomics by other CUs notwithstanding -- so nearly every row sent its atomicMax to the same L2 line: 9.9 M atomics
// for 10 M rows, 103 ms for a 7.7 GB pass, most of a first search's 142 ms: profiles/r6_first_call.txt)
__device__ __forceinline__ unsigned cl_wave_max_u32(unsigned v) {
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const unsigned w = (unsigned)__shfl_xor((int)v, o);
		v = w > v ? w : v;
	}
	return v;
}
template <bool IS_L2>
__global__ void rows_to_bf16_hi_kernel(const float *__restrict__ src, long long row0, long long nrows, int dp, int dpd,
                                       int interleaved, const float *__restrict__ mu, unsigned short *__restrict__ dst,
                                       float *__restrict__ beta, const float *__restrict__ norms,
                                       unsigned *__restrict__ max_bits, int *__restrict__ outl) {
	const int g8 = dp / 8;
	const long long total = nrows * g8, stride = (long long)gridDim.x * blockDim.x;
	const float tau = outl ? __uint_as_float(max_bits[3]) : INFINITY; // (written before this launch)
	unsigned m0 = 0u, m2 = 0u, m8 = 0u, m12 = 0u; // this thread's share of max_bits[0], [2], [8], [12]
	// (every lane of a wave runs the same number of rounds: the DPP / shuffle sums below need the row's 16 neighbours)
	for (long long base = (long long)blockIdx.x * blockDim.x; base < total; base += stride) {
	const long long i = base + threadIdx.x;
	const bool live = i < total;
	const long long r = row0 + (live ? i / g8 : 0);
	const int c8 = (int)(i % g8);
	float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (live) {
		const float4 s0 = *(const float4 *)(src + (size_t)r * dp + c8 * 8);
		const float4 s1 = *(const float4 *)(src + (size_t)r * dp + c8 * 8 + 4);
		if (!interleaved) {
			v[0] = s0.x, v[1] = s0.y, v[2] = s0.z, v[3] = s0.w, v[4] = s1.x, v[5] = s1.y, v[6] = s1.z, v[7] = s1.w;
		} else if ((r >> 4) & 1) { // stored [k1,k3,k0,k2] (FlatGeom::pair_interleaved)
			v[0] = s0.z, v[1] = s0.x, v[2] = s0.w, v[3] = s0.y, v[4] = s1.z, v[5] = s1.x, v[6] = s1.w, v[7] = s1.y;
		} else { // stored [k0,k2,k1,k3]
			v[0] = s0.x, v[1] = s0.z, v[2] = s0.y, v[3] = s0.w, v[4] = s1.x, v[5] = s1.z, v[6] = s1.y, v[7] = s1.w;
		}
	}
	bf16x8 hi;
	float n2 = 0.f, my = 0.f, r2 = 0.f; // partial ||y'||^2, <mu, y> and ||y' - bf16(y')||^2 (the row's actual rounding residual)
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		const float m = mu[c8 * 8 + e];
		const float c = v[e] - m;
		hi[e] = (__bf16)c;
		const float dl = c - (float)hi[e]; // (exact: the two agree in their leading 8 bits)
		n2 = fmaf(c, c, n2);
		my = fmaf(m, v[e], my);
		r2 = fmaf(dl, dl, r2);
	}
	if (g8 == 16) { // the row's threads are one DPP row of 16 lanes: sum by quad_perm x 2, row_half_mirror, row_mirror (no LDS)
		n2 += cl_dpp<0xB1>(n2), my += cl_dpp<0xB1>(my), r2 += cl_dpp<0xB1>(r2);
		n2 += cl_dpp<0x4E>(n2), my += cl_dpp<0x4E>(my), r2 += cl_dpp<0x4E>(r2);
		n2 += cl_dpp<0x141>(n2), my += cl_dpp<0x141>(my), r2 += cl_dpp<0x141>(r2);
		n2 += cl_dpp<0x140>(n2), my += cl_dpp<0x140>(my), r2 += cl_dpp<0x140>(r2);
	} else {
		for (int o = g8 >> 1; o >= 1; o >>= 1) { // an aligned group of g8 lanes
			n2 += __shfl_xor(n2, o);
			my += __shfl_xor(my, o);
			r2 += __shfl_xor(r2, o);
		}
	}
	// (n2 is the whole row's sum in every one of its g8 threads; the first of them takes the slot, the others read it from its lane --
	// the count may run past the capacity: rows beyond it stay in the store, and in the maxima)
	const bool big = live && n2 > tau;
	int slot = CL_OUTL_CAP;
	if (big && c8 == 0)
		slot = atomicAdd(outl, 1);
	slot = __shfl(slot, (int)(threadIdx.x & 63u) & ~(g8 - 1));
	if (!live)
		continue;
	const unsigned b = __float_as_uint(norms[r]);
	const bool out = big && slot < CL_OUTL_CAP;
	if (out) {
		if (c8 == 0)
			outl[1 + slot] = (int)r;
		const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
		*(bf16x8 *)(dst + (size_t)r * dpd + c8 * 8) = zero;
		if (c8 == 0) {
			beta[r] = -INFINITY;
			m0 = b > m0 ? b : m0;
		}
		continue;
	}
	*(bf16x8 *)(dst + (size_t)r * dpd + c8 * 8) = hi; // (dpd = 128 >= dp: the store is zero-filled when it is allocated)
	if (c8 == 0) {
		beta[r] = IS_L2 ? -n2 : my;
		m0 = b > m0 ? b : m0;
		m2 = b > m2 ? b : m2;
		const unsigned bc = __float_as_uint(n2); // (>= 0 or NaN: the bit pattern orders like the value, NaN above everything)
		m8 = bc > m8 ? bc : m8;
		const unsigned br = __float_as_uint(r2);
		m12 = br > m12 ? br : m12;
	}
	}
	m0 = cl_wave_max_u32(m0), m2 = cl_wave_max_u32(m2), m8 = cl_wave_max_u32(m8), m12 = cl_wave_max_u32(m12);
	if ((threadIdx.x & 63u) == 0u) {
		if (m0)
			atomicMax(max_bits, m0);
		if (m2)
			atomicMax(max_bits + 2, m2);
		if (m8)
			atomicMax(max_bits + 8, m8);
		if (m12)
			atomicMax(max_bits + 12, m12);
	}
}
void launch_rows_to_bf16_hi(const FlatGeom &g, int metric, const float *d_vecs, int64_t row0, int64_t nrows, const float *d_mu,
                            unsigned short *d_bf, float *d_beta, const float *d_norms, unsigned *d_max_norm_bits,
                            hipStream_t st, int *d_outl) {
	if (nrows <= 0)
		return;
	const long long total = (long long)nrows * (g.dp / 8);
	const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 16384)); // (grid-stride: 16 resident workgroups per CU, four rounds)
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(rows_to_bf16_hi_kernel<true>, grid, dim3(256), 0, st, d_vecs, (long long)row0, (long long)nrows, g.dp, 128,
		                   g.pair_interleaved ? 1 : 0, d_mu, d_bf, d_beta, d_norms, d_max_norm_bits, d_outl);
	else
		hipLaunchKernelGGL(rows_to_bf16_hi_kernel<false>, grid, dim3(256), 0, st, d_vecs, (long long)row0, (long long)nrows, g.dp, 128,
		                   g.pair_interleaved ? 1 : 0, d_mu, d_bf, d_beta, d_norms, d_max_norm_bits, d_outl);
	MVS_HIP(hipGetLastError());
}

// queries -> B fragments of v_mfma_f32_16x16x32_bf16: qf[(qblk16 * KB + kb) * 64 + lane] = 8 bf16 of alpha x (the CENTRED query
// qblk16*16 + (lane & 15)), dims kb*32 + 8*(lane >> 4) + 0..7; alpha = 2 (L2) or 1 goes into the operand so that the MFMA chain,
// started from beta(row) instead of 0, delivers s = alpha <x', y'> + beta with no vector-ALU work at all
__global__ void collect_pack_queries_kernel(const float *__restrict__ x, long long nq, int d, int nkb,
                                            const float *__restrict__ mu, float alpha, bf16x8 *__restrict__ qf,
                                            long long total) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // one (qblk16, kb, lane)
	if (i >= total)
		return;
	const int lane = (int)(i & 63);
	const long long t = i >> 6;
	const int kb = (int)(t % nkb);
	const long long qblk16 = t / nkb;
	const long long q = qblk16 * 16 + (lane & 15);
	bf16x8 hi;
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		const int kk = kb * 32 + 8 * (lane >> 4) + e;
		hi[e] = (__bf16)((q < nq && kk < d) ? alpha * (x[q * d + kk] - mu[kk]) : 0.f); // (alpha = 1 or 2: exact)
	}
	qf[i] = hi;
}
size_t collect_qfrag_bytes(const FlatGeom &g, int64_t nq) {
	(void)g; // (the d <= 128 store is 128 dims wide whatever FlatGeom::dp is: 16 < d <= 64 rows are zero-padded)
	const int64_t nblk16 = (nq + CL_QBLOCK - 1) / CL_QBLOCK * (CL_QBLOCK / 16);
	return (size_t)nblk16 * 4 * 64 * 16;
}
// the same for a store of dp1 dims and workgroups of `qblock` queries (csrc/flat_collect_wide.hip)
size_t collect_qfrag_bytes_ex(int dp1, int qblock, int64_t nq) {
	const int64_t nblk16 = (nq + qblock - 1) / qblock * (qblock / 16);
	return (size_t)nblk16 * (dp1 / 32) * 64 * 16;
}
void launch_collect_pack_queries_ex(int d, int dp1, int qblock, int metric, const float *d_x, int64_t nq, const float *d_mu,
                                    void *d_qf, hipStream_t st) {
	const int64_t nblk16 = (nq + qblock - 1) / qblock * (qblock / 16);
	const long long total = (long long)nblk16 * (dp1 / 32) * 64;
	hipLaunchKernelGGL(collect_pack_queries_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_x,
	                   (long long)nq, d, dp1 / 32, d_mu, metric == METRIC_L2 ? 2.0f : 1.0f, (bf16x8 *)d_qf, total);
	MVS_HIP(hipGetLastError());
}
void launch_collect_pack_queries(const FlatGeom &g, int metric, const float *d_x, int64_t nq, const float *d_mu, void *d_qf,
                                 hipStream_t st) {
	const int64_t nblk16 = (nq + CL_QBLOCK - 1) / CL_QBLOCK * (CL_QBLOCK / 16);
	const long long total = (long long)nblk16 * 4 * 64;
	hipLaunchKernelGGL(collect_pack_queries_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_x,
	                   (long long)nq, g.d, 4, d_mu, metric == METRIC_L2 ? 2.0f : 1.0f, (bf16x8 *)d_qf, total);
	MVS_HIP(hipGetLastError());
}

// ---- error bound ---------------------------------------------------------------------------------------------------
// u = 2^-24.  Primed quantities are centred (x' = fl(x - mu), y' = fl(y - mu)); S' = ||x'|| ||y'||_max >= sum |x'_i y'_i|
// (Cauchy-Schwarz), S = ||x|| ||y||_max; every norm is inflated by 1e-4 for its own rounding.
//   bf16 rounding of both operands: |x'_i y'_i - bf(x'_i) bf(y'_i)| <= (2 * 2^-8 + 2^-16) |x'_i y'_i|  -> (2^-7 + 2^-16) S'
//   (bf16 keeps 8 significant bits: round-to-nearest errs by up to 2^-8 |v| per operand)
//   ROUND 4 (mode 1, default): the same term from the ACTUAL rounding residuals instead of the worst case per element.  With
//   a = alpha x' (what the pack kernel rounds), Q = bf(a), Y = bf(y'):  <a, y'> - <Q, Y> = <a - Q, y'> + <Q, y' - Y>, hence
//        | . | <= ||a - Q|| ||y'||_max + (||a|| + ||a - Q||) ||y' - Y||_max        (Cauchy-Schwarz, nothing modelled)
//   ||a - Q|| is computed here per query (the differences are exact in f32), ||y' - Y||_max when the bf16 store is built
//   (max_norm_bits[12]).  A rounding residual is uniform inside its half ulp, so the norms come out near 0.41 x 2^-8 of the
//   operand norms instead of 2^-8: E shrinks ~2.4x and with it the band of rows the scan has to admit (uniform rows at the
//   headline: 234 -> ~150 candidates per query; the band is exponentially sensitive on clustered rows).
//   bf16 MFMA accumulation, the chain starting at C = beta: the instruction's internal alignment is not documented.  MEASURED in
//   round 5 (tests/test_mfma_model_gpu.py, the bare v_mfma_f32_16x16x32_bf16 against the exact sum on adversarial tiles): worst
//   |D - exact| = 8.8 u (|C| + sum |a_k b_k|) per 32-product instruction -- terms far below the largest one are truncated, not
//   rounded, in two stages (3.8 u when the large value is C, 8.8 u when it is a product).  Charged: CL_MFMA_UNITS = 8 u of the
//   magnitudes per 16 dimensions with a 1.25 safety factor = 20 u per instruction (rounds 2-4 charged 4: 10 u per instruction, on
//   the edge of what was then unmeasured):                                                 -> 1.25 (d/16) 8u ((1 + 2^-7 + 2^-16) alpha S' + |beta|_max)
//   => |s - (alpha <x', y'> + beta)| <= es = alpha (2^-7 + 2^-16) S' + that
//   centring: x', y' carry one rounding per component (<= u |.|), beta is a d-term f32 chain:
//        L2: | ||x-y||^2 - (||x'||^2 + ||y'||^2 - 2<x',y'>) | <= 4u (xn' + yn'_max);  |beta + ||y'||^2| <= d u yn'_max
//        IP: | <x,y> - (<x',y'> + <mu,y> + <x',mu>) | <= 4u S' + 2u ||mu|| ||y||_max;   |beta - <mu,y>| <= d u ||mu|| ||y||_max
//   the exact value the oracle reports: L2 D = max(0, fl(fl(xn + yn) - 2 chain)), chain = d sequential fmas:
//        |D - ||x-y||^2| <= 2 d u S + 4u (xn + yn_max); per-pair branch (selector, nq < 20): D = sum fl((x_k - y_k)^2) accumulated
//        in f32: <= (d + 4) u ||x-y||^2 <= 2 (d + 8) u (xn + yn_max);   IP: |chain - <x,y>| <= d u S
//   E = the sum of the applicable lines; e2 = 2 E (1 + 2^-10) + the rounding of (B - e2)
//   itself.  Anything non-finite -> NaN (the query goes to the exact kernel).
template <bool IS_L2>
__global__ void collect_bounds_kernel(const float *__restrict__ x, long long nq, int d, const float *__restrict__ mu,
                                      const unsigned *__restrict__ max_norm_bits, float *__restrict__ e2,
                                      int *__restrict__ fail_cnt, int *__restrict__ fail_q, int bound_mode) {
	const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nq) {
		e2[q] = __uint_as_float(0x7fc00000u); // (the slots behind the last query, up to the next multiple of 256: nothing passes)
		return;
	}
	float xn = 0.f, xnc = 0.f, mun = 0.f, dq2 = 0.f;
	const float alf = IS_L2 ? 2.0f : 1.0f;
	for (int t = 0; t < d; ++t) {
		const float v = x[q * d + t], m = mu[t], c = v - m;
		xn = fmaf(v, v, xn);
		xnc = fmaf(c, c, xnc);
		mun = fmaf(m, m, mun);
		const float a = alf * c; // exactly the operand collect_pack_queries_kernel rounds
		const float dl = a - (float)(__bf16)a;
		dq2 = fmaf(dl, dl, dq2);
	}
	const float yn = __uint_as_float(max_norm_bits[2]), ync = __uint_as_float(max_norm_bits[8]); // ([2]: over the rows IN the store)
	const float dyc = __uint_as_float(max_norm_bits[12]);
	const double u = 5.9604644775390625e-08, infl = 1.0001;
	const double S = sqrt((double)xn * infl) * sqrt((double)yn * infl);
	const double Sc = sqrt((double)xnc * infl) * sqrt((double)ync * infl);
	const double MY = sqrt((double)mun * infl) * sqrt((double)yn * infl); // >= |<mu, y>|
	// s comes straight out of the MFMA chain: C starts at beta, the B operand carries alpha
	const double al = IS_L2 ? 2.0 : 1.0;
	const double bmax = IS_L2 ? (double)ync : MY; // >= |beta|
	const double rnd_worst = al * (0.0078125 + 1.52587890625e-05) * Sc;
	const double ndq = sqrt((double)dq2 * infl), ndy = sqrt((double)dyc * infl);
	const double rnd_actual = ndq * sqrt((double)ync * infl) + (al * sqrt((double)xnc * infl) + ndq) * ndy;
	// (never above the worst case; a non-finite residual norm poisons E below like any other non-finite input)
	const double rnd = bound_mode == 0 ? rnd_worst : (rnd_actual < rnd_worst || !(rnd_actual == rnd_actual) ? rnd_actual : rnd_worst);
	const double es = rnd + 1.25 * ((double)d / 16.0) * CL_MFMA_UNITS * u * ((1.0 + 0.0079) * al * Sc + bmax);
	double E;
	if (IS_L2)
		E = es + 4.0 * u * ((double)xnc + ync) + (double)d * u * ync + 2.0 * d * u * S + 4.0 * u * ((double)xn + yn) +
		    2.0 * ((double)d + 8.0) * u * ((double)xn + yn); // (+ the per-pair value FAISS reports under a selector / for < 20 queries)
	else
		E = es + 4.0 * u * Sc + 2.0 * u * MY + (double)d * u * MY + (double)d * u * S;
	float r = (float)(2.0 * E * (1.0 + 0.0009765625) + 8.0 * u * (Sc + MY + (double)xnc + ync) + 1e-30);
	const bool ok = isfinite(xn) && isfinite(yn) && isfinite(ync) && isfinite(mun) && isfinite(dq2) && isfinite(dyc) && isfinite(r) && r < 1e30f;
	if (!ok) {
		r = __uint_as_float(0x7fc00000u);
		fail_q[atomicAdd(fail_cnt, 1)] = (int)q;
	}
	e2[q] = r;
}
void launch_collect_bounds(int metric, const float *d_x, int64_t nq, int d, const float *d_mu,
                           const unsigned *d_max_norm_bits, float *d_e2, int *d_fail_cnt, int *d_fail_q, hipStream_t st) {
	if (nq <= 0)
		return;
	const dim3 grid((unsigned)((nq + 255) / 256));
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(collect_bounds_kernel<true>, grid, dim3(256), 0, st, d_x, (long long)nq, d, d_mu, d_max_norm_bits, d_e2,
		                   d_fail_cnt, d_fail_q, tune().cl_bound_mode);
	else
		hipLaunchKernelGGL(collect_bounds_kernel<false>, grid, dim3(256), 0, st, d_x, (long long)nq, d, d_mu, d_max_norm_bits, d_e2,
		                   d_fail_cnt, d_fail_q, tune().cl_bound_mode);
	MVS_HIP(hipGetLastError());
}

// ---- round 5: everything a search does PER QUERY in front of the scans, in one launch (d <= 128 store) -------------------------
// Rounds 2-4: collect_pack_queries_kernel (6.7 us at 10 000 queries), query_norms_kernel (21: one wave per 16 queries),
// collect_bounds_kernel (18.6: one THREAD per query walking its row with a 512-byte stride), init_gslot_kernel (4.9), a memset of
// the control block (5) -- five launches and their gaps, ~ 70 us of a 2.7 ms shard step.  Here a workgroup takes 64 queries:
// their rows pass through LDS once (coalesced), thread t < 64 runs query t's four chains -- ||x||^2 as the k-ordered fma chain the
// exact re-scoring needs bit for bit (csrc/util_kernels.hip query_norms_kernel), and the three sums of collect_bounds_kernel --
// while all threads write the bf16 fragments (collect_pack_queries_kernel's layout), the neutral class slots and the zeroed
// control words of the workgroup's queries.
template <bool IS_L2>
__global__ __launch_bounds__(256) void collect_query_prep_kernel(const float *__restrict__ x, long long nq, int d,
                                                                const float *__restrict__ mu, const unsigned *__restrict__ max_norm_bits,
                                                                bf16x8 *__restrict__ qf, long long nq_frag /* queries the fragment array covers */,
                                                                float *__restrict__ qn, float *__restrict__ e2, long long nq_e2 /* entries of e2 */,
                                                                int *__restrict__ fail_cnt, int *__restrict__ fail_q, int bound_mode,
                                                                unsigned *__restrict__ gslot, int stride, int *__restrict__ ctl_hdr,
                                                                int *__restrict__ ctl_seg /* [2 nq] */) {
	__shared__ float xs[64][129];
	__shared__ float ms[128];
	const int tid = threadIdx.x;
	const long long q0 = (long long)blockIdx.x * 64;
	for (int i = tid; i < 128; i += 256)
		ms[i] = i < d ? mu[i] : 0.f;
	for (int i = tid; i < 64 * d; i += 256) {
		const int r = i / d, t = i - r * d;
		xs[r][t] = q0 + r < nq ? x[(q0 + r) * d + t] : 0.f;
	}
	__syncthreads();
	// round 6: the four sums of a query are four independent k-ordered chains -- one WAVE each (wave-uniform branch) instead of all four
	// on one thread while three waves idled (20 us at 10 000 queries, in front of every large search)
	__shared__ float part[4][64];
	{
		const int r = tid & 63, which = tid >> 6;
		float acc = 0.f;
		if (q0 + r < nq) {
			const float alf = IS_L2 ? 2.0f : 1.0f;
			if (which == 0) {
				for (int t = 0; t < d; ++t) {
					const float v = xs[r][t];
					acc = fmaf(v, v, acc);
				}
			} else if (which == 1) {
				for (int t = 0; t < d; ++t) {
					const float c = xs[r][t] - ms[t];
					acc = fmaf(c, c, acc);
				}
			} else if (which == 2) {
				for (int t = 0; t < d; ++t) {
					const float m = ms[t];
					acc = fmaf(m, m, acc);
				}
			} else {
				for (int t = 0; t < d; ++t) {
					const float a = alf * (xs[r][t] - ms[t]); // exactly the operand the fragment loop below rounds
					const float dl = a - (float)(__bf16)a;
					acc = fmaf(dl, dl, acc);
				}
			}
		}
		part[which][r] = acc;
	}
	__syncthreads();
	if (tid < 64) {
		const long long q = q0 + tid;
		if (q < nq) {
			const float xn = part[0][tid], xnc = part[1][tid], mun = part[2][tid], dq2 = part[3][tid];
			qn[q] = xn;
			const float yn = __uint_as_float(max_norm_bits[2]), ync = __uint_as_float(max_norm_bits[8]); // ([2]: over the rows IN the store)
			const float dyc = __uint_as_float(max_norm_bits[12]);
			const double u = 5.9604644775390625e-08, infl = 1.0001;
			const double S = sqrt((double)xn * infl) * sqrt((double)yn * infl);
			const double Sc = sqrt((double)xnc * infl) * sqrt((double)ync * infl);
			const double MY = sqrt((double)mun * infl) * sqrt((double)yn * infl); // >= |<mu, y>|
			const double al = IS_L2 ? 2.0 : 1.0;
			const double bmax = IS_L2 ? (double)ync : MY; // >= |beta|
			const double rnd_worst = al * (0.0078125 + 1.52587890625e-05) * Sc;
			const double ndq = sqrt((double)dq2 * infl), ndy = sqrt((double)dyc * infl);
			const double rnd_actual = ndq * sqrt((double)ync * infl) + (al * sqrt((double)xnc * infl) + ndq) * ndy;
			const double rnd = bound_mode == 0 ? rnd_worst : (rnd_actual < rnd_worst || !(rnd_actual == rnd_actual) ? rnd_actual : rnd_worst);
			const double es = rnd + 1.25 * ((double)d / 16.0) * CL_MFMA_UNITS * u * ((1.0 + 0.0079) * al * Sc + bmax);
			double E; // (collect_bounds_kernel's formula, term by term)
			if (IS_L2)
				E = es + 4.0 * u * ((double)xnc + ync) + (double)d * u * ync + 2.0 * d * u * S + 4.0 * u * ((double)xn + yn) +
				    2.0 * ((double)d + 8.0) * u * ((double)xn + yn);
			else
				E = es + 4.0 * u * Sc + 2.0 * u * MY + (double)d * u * MY + (double)d * u * S;
			float r = (float)(2.0 * E * (1.0 + 0.0009765625) + 8.0 * u * (Sc + MY + (double)xnc + ync) + 1e-30);
			const bool ok = isfinite(xn) && isfinite(yn) && isfinite(ync) && isfinite(mun) && isfinite(dq2) && isfinite(dyc) && isfinite(r) && r < 1e30f;
			if (!ok) {
				r = __uint_as_float(0x7fc00000u);
				fail_q[atomicAdd(fail_cnt, 1)] = (int)q;
			}
			e2[q] = r;
		} else if (q < nq_e2) {
			e2[q] = __uint_as_float(0x7fc00000u); // (the slots behind the last query: nothing passes)
		}
	}
	// fragments of the workgroup's four 16-query blocks: entry (qblk16, kb, lane) = queries qblk16 * 16 + (lane & 15), dims kb * 32 + 8 (lane >> 4) + e
	const float alpha = IS_L2 ? 2.0f : 1.0f;
	for (int i = tid; i < 4 * 4 * 64; i += 256) {
		const int lane = i & 63, kb = (i >> 6) & 3, qb = i >> 8;
		const int r = qb * 16 + (lane & 15);
		if (q0 + qb * 16 >= nq_frag)
			continue;
		bf16x8 hi;
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int kk = kb * 32 + 8 * (lane >> 4) + e;
			hi[e] = (__bf16)((q0 + r < nq && kk < d) ? alpha * (xs[r][kk] - ms[kk]) : 0.f); // (alpha = 1 or 2: exact)
		}
		qf[((q0 / 16 + qb) * 4 + kb) * 64 + lane] = hi;
	}
	// class slots neutral ("larger s is better" keys), control words zero
	const unsigned neutral = ~f2key(-FLT_MAX);
	for (int i = tid; i < 64 * stride; i += 256) {
		const long long q = q0 + i / stride;
		if (q < nq)
			gslot[q * stride + (i % stride)] = neutral;
	}
	if (tid < 64 && q0 + tid < nq) {
		ctl_seg[q0 + tid] = 0;
		ctl_seg[nq + q0 + tid] = 0;
	}
	if (blockIdx.x == 0 && tid < 64)
		ctl_hdr[tid] = 0;
}
// d <= 128 store only (collect_store_dims(d) == 128); e2 has (nq rounded up to 256) entries, qf covers nq rounded up to CL_QBLOCK
void launch_collect_query_prep(int metric, const float *d_x, int64_t nq, int d, const float *d_mu, const unsigned *d_max_norm_bits,
                               void *d_qf, float *d_qn, float *d_e2, int *d_fail_cnt, int *d_fail_q, unsigned *d_gslot, int stride,
                               int *d_ctl_hdr, int *d_ctl_seg, hipStream_t st) {
	if (nq <= 0)
		return;
	const long long nq_frag = (nq + CL_QBLOCK - 1) / CL_QBLOCK * CL_QBLOCK, nq_e2 = (nq + 255) / 256 * 256;
	const dim3 grid((unsigned)(nq_frag / 64));
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(collect_query_prep_kernel<true>, grid, dim3(256), 0, st, d_x, (long long)nq, d, d_mu, d_max_norm_bits, (bf16x8 *)d_qf,
		                   nq_frag, d_qn, d_e2, nq_e2, d_fail_cnt, d_fail_q, tune().cl_bound_mode, d_gslot, stride, d_ctl_hdr, d_ctl_seg);
	else
		hipLaunchKernelGGL(collect_query_prep_kernel<false>, grid, dim3(256), 0, st, d_x, (long long)nq, d, d_mu, d_max_norm_bits, (bf16x8 *)d_qf,
		                   nq_frag, d_qn, d_e2, nq_e2, d_fail_cnt, d_fail_q, tune().cl_bound_mode, d_gslot, stride, d_ctl_hdr, d_ctl_seg);
	MVS_HIP(hipGetLastError());
}

// ---- the scan kernel ---------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16: D[16 x 16] += A[16 x 32] B[32 x 16]; lane l holds A[row l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4)
// + j][col l & 15] (8 bf16 each) and D[row 4 (l >> 4) + r][col l & 15] (4 f32).  A 32-row x 32-query block is four such tiles
// with FOUR INDEPENDENT accumulators: back-to-back MFMAs never wait for each other's result (with the 32x32x16 shape the 8
// MFMAs of a block form one dependent chain: 18.2 vs 14.4 ms for the bare loop).
// COLLECT = false: bound estimation only (publish to the slots, append nothing) -- the pre-pass over the first rows
// ABL (profiling builds of the L2 collect instance only; results are WRONG when != 0): bit 0 = no rare path, bit 1 = no
// fold either (bare MFMA + staging), bit 2 = stage only the first tile, bit 3 = no workgroup barrier
typedef float f32x4acc __attribute__((ext_vector_type(4)));
// SEL: an IDSelector is active -- only the rows whose bit is set in a.rowmask (one bit per row, built per search by
// collect_rowmask_kernel) are published and appended; the bound then is the kk-th best SELECTED row's, as it must be
// NC: row classes per query (row & (NC - 1)): 16, or 32 for 16 < kk <= 32 (the bound is the kk-th best of NC class bests)
template <int KCH, bool IS_L2, bool COLLECT, int ABL = 0, bool SEL = false, int NC = 16>
__global__ __launch_bounds__(256, 2) void flat_bf16_collect_kernel(const CollectArgs a) {
	constexpr int DP = KCH * 16;
	constexpr int KB = DP / 32;               // k-blocks of 32 dimensions
	constexpr int PITCH = DP * 2;             // bytes per row (256 at d = 128)
	constexpr int C = PITCH / 16;             // 16-byte chunks per row
	constexpr int TILE_BYTES = CL_BN * PITCH; // 8 KB at d = 128: what one pass of the MFMA loop consumes
	constexpr int STAGE_BYTES = CL_SUB * TILE_BYTES; // what is staged between two barriers (CL_SUB tiles)
	constexpr int NDMA = STAGE_BYTES / 1024;  // LDS-DMA instructions per stage (1 KB per wave-instruction)
	constexpr int DMA_PER_WAVE = NDMA / 4;
	static_assert(C == 16 && KB == 4 && NDMA % 4 == 0 && CL_SUB * CL_BN <= 64, "d = 128 geometry");

	extern __shared__ __attribute__((aligned(16))) float smem[];
	char *tbuf = (char *)smem;                                  // [2][STAGE_BYTES]
	float *nbuf = (float *)(tbuf + 2 * STAGE_BYTES);            // [2][64] beta of the staged rows
	unsigned long long *qbuf = (unsigned long long *)(nbuf + 2 * 64); // [CL_QCAP] candidate queue
	float *cqtab = (float *)(qbuf + CL_QCAP);                   // [4 waves][4 t][16 c][2]: pass bound of every query
	unsigned *qctl = (unsigned *)(cqtab + CL_QBLOCK);           // [7] candidates counted after the stream filled up
	float *qval = (float *)(qctl + 16);                         // [CL_QCAP] value of every queued hit (wave w: entries 512 w ..)

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int hq = lane >> 4, c = lane & 15;
	int split, qb;
	if (a.xcd_map) {
		const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
		split = (idx / a.nqb) * 8 + xcd;
		qb = idx % a.nqb;
	} else {
		split = blockIdx.x / a.nqb;
		qb = blockIdx.x % a.nqb;
	}
	const long long r_begin = a.row_first + (long long)split * a.split_rows;
	long long r_end = r_begin + a.split_rows;
	if (r_end > a.n)
		r_end = a.n;
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin + CL_SUB * CL_BN - 1) / (CL_SUB * CL_BN)) : 0; // staged blocks
	if (tid == 0)
		qctl[7] = 0u; // candidates counted after the stream filled up

	// the wave's 128 queries = 8 column blocks of 16; block cb = 2 t + i belongs to "tile" t; lane (hq, c) sees query
	// qw + 16 cb + c in every block and OWNS (bound refresh) the two blocks of t = hq
	const int qw = qb * CL_QBLOCK + wave * 128;

	// B fragments, resident: [column block][k-block]
	bf16x8 bq[8][KB];
	{
		const bf16x8 *qsrc = (const bf16x8 *)a.qf;
#pragma unroll
		for (int cb = 0; cb < 8; ++cb) {
			const size_t qblk16 = (size_t)qb * (CL_QBLOCK / 16) + wave * 8 + cb;
#pragma unroll
			for (int kb = 0; kb < KB; ++kb)
				bq[cb][kb] = qsrc[(qblk16 * KB + kb) * 64 + lane];
		}
	}

	// LDS-DMA staging.  Instruction `inst` of a tile fills LDS bytes [1024 inst, +1024); lane l owns 16-byte slot S = 64 inst
	// + l = (row r = S / 16, position p = S % 16) and fetches the row's chunk p ^ (r & 15).  Wave w issues inst = 4 i + w:
	// r = 16 i + 4 w + l / 16, so r & 15 does not depend on i and the per-lane byte offset is loop invariant; every issue is
	// ONE instruction with a uniform base.  Tiles past the end are fetched as well (64 rows of zero padding): no clamp, no
	// branch around a vector-memory instruction ("what round 2 learnt").
	unsigned dma_off;
	{
		const int rr = 4 * wave + (lane >> 4);
		dma_off = (unsigned)(rr * PITCH + (((lane & 15) ^ rr) * 16));
	}
	auto dma_issue = [&](int u, int i) {
		const char *base = (const char *)a.yb + ((size_t)(r_begin + (long long)u * (CL_SUB * CL_BN)) + (size_t)i * 16) * PITCH; // uniform
		__builtin_amdgcn_global_load_lds((glb_f32c *)(base + dma_off),
		                                 (lds_f32c *)(smem + ((u & 1) * STAGE_BYTES + (i * 4 + wave) * 1024) / 4), 16, 0, 0);
	};
	auto dma_norms = [&](int u) {
		const float *base = a.yn + (r_begin + (long long)u * (CL_SUB * CL_BN)); // uniform
		__builtin_amdgcn_global_load_lds((glb_f32c *)(base + lane), (lds_f32c *)(smem + (2 * STAGE_BYTES) / 4 + (u & 1) * 64),
		                                 4, 0, 0);
	};
	// Pass bounds through a TABLE IN GLOBAL MEMORY (round 4).  Round 3 had every wave re-derive its 128 bounds from the class slots
	// (an L2 round trip per query pair + a 16-key network, ~4.4 us of workgroup time, 11 times per 9 766-row split = 9 % of a shard's
	// scan) -- and the 25 workgroups that share a query block all derived the same numbers.  Now a.pbnd holds B - 2E of every query in
	// the order of the workgroup's LDS table, a wave fetches its 128 entries with two LDS-DMA instructions (no registers, no wait:
	// they land before the staged block's barrier; a bound is valid whenever it was computed, so old and new entries may mix), and
	// the full derivation runs once per PB_R staged blocks per workgroup at a phase that depends on the row split: the workgroups
	// of a query block take turns and between them refresh the table every block or two.
	const bool use_tab = a.pbnd != nullptr;
	// (round 6, lists beyond 128 entries: the bounds come from a pass of their own and stay what they are -- a.opt bit 8: the table is
	// never re-derived from the class slots, which no longer mean anything, and nothing is published to them)
	const bool frozen = use_tab && (a.opt & 256) != 0;
	const int tab_bits = (a.opt >> 2) & 3;
	const int tab_shift = tab_bits == 0 ? 2 : (tab_bits == 1 ? 1 : tab_bits + 1); // fetch every 4 (default) / 2 / 8 / 16 staged blocks
	// full derivation every 64 staged blocks per workgroup (same box, N = 1.25 M / 1 M / 10 M, ms per step: every 16: 3.17 / 2.69 / 18.08,
	// 32: 3.05 / 2.62 / 17.59, 64: 3.02 / 2.58 / 17.30), option bits 4..5: 1 = 16, 2 = 128, 3 = by the scan's progress: 16 while the
	// query block's workgroups have seen little (the bound still moves), 64, then 256
	// (the publish-only pre-pass of the 32-class instances runs a few blocks per workgroup from cold slots: every 4 blocks there)
	const int duty_bits = (a.opt >> 4) & 3;
	const int duty_g0 = duty_bits == 3 ? (int)(blockIdx.x / 512u) * ntiles : 0; // staged blocks this workgroup's predecessors on the slot saw
	auto duty_mask_at = [&](int u) {
		if (!COLLECT)
			return 3;
		if (duty_bits == 3) {
			const int g = duty_g0 + u;
			return g < 64 ? 15 : (g < 256 ? 63 : 255);
		}
		return duty_bits == 0 ? (NC > 32 ? 255 : 63) : (duty_bits == 1 ? 15 : 127); // (128 classes: four networks per derivation)
	};
	const int duty_phase = split * 13 + 5;
	auto dma_bounds = [&]() {
		const float *base = a.pbnd + (size_t)qb * CL_QBLOCK + wave * 128; // uniform
		float *dst = cqtab + wave * 128;
		__builtin_amdgcn_global_load_lds((glb_f32c *)(base + lane), (lds_f32c *)dst, 4, 0, 16 /* sc1: agent scope */);
		__builtin_amdgcn_global_load_lds((glb_f32c *)(base + 64 + lane), (lds_f32c *)(dst + 64), 4, 0, 16);
	};
	if (ntiles > 0) {
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i)
			dma_issue(0, i);
		dma_norms(0);
		if (use_tab)
			dma_bounds();
	}
	__syncthreads();

	// A fragment (row block rb, k-block kb): row 16 rb + c, chunk 4 kb + hq -> byte 4096 rb + 256 c + (((4 kb + hq) ^ c) * 16)
	// = 4096 rb + (rbase ^ (64 kb)) with rbase = 256 c | ((hq ^ c) * 16)  (4 kb and hq occupy disjoint bits of the chunk number)
	const unsigned rbase = (unsigned)(c * PITCH) | (unsigned)(((hq ^ c) & 15) * 16);
	constexpr int WQCAP = CL_QCAP / 4; // every wave has its own quarter of the queue
	const unsigned qcnt_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned *)qctl);
	const unsigned qbuf_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned long long *)qbuf) + (unsigned)(wave * WQCAP * 8);
	const unsigned qval_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)qval) + (unsigned)(wave * WQCAP * 4);
	// the lane's two bounds of tile t: cqtab[wave][t][c][0..1]
	const unsigned cq_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)cqtab) + (unsigned)((wave * 64 + c) * 8);

	int ovf = 0; // (uniform) set at a flush that found the stream full: the rare path then only counts
	// Rare path of half a 32-query tile (row block rb; sv holds s of its 4 rows x 2 queries per lane): every passing row is
	// published to its class slot (16 classes: row & 15) and appended.  Lane (hq, c): rows 16 rb + 4 hq + r, queries of column
	// blocks 2 t + i.
	// (round 4: the loop only RECORDS a passing row -- {query, row} and its value into the wave's own part of the LDS queue, at
	// position wave-uniform fill + ballot rank: no LDS atomic with a result, no address arithmetic, no class-slot atomic by the one lane
	// that had the hit while 63 wait -- and publish() / wdrain() work the queue off with one hit per LANE at the next look at the queue.
	// The path is entered in 7 % of the half tiles at N = 1.25 M (1.4 % at 10 M) and was 14 % (4.4 %) of the scan: profiles/
	// r4_ablation_bare_mfma_loop.txt; the IVF scan, where it was a third, does the same: csrc/ivf_collect.hip.)
	int wfill = 0, wpub = 0; // entries recorded by this wave / of those, already published to the class slots (wave-uniform)
	auto publish = [&]() __attribute__((always_inline)) {
		const unsigned n = (unsigned)wfill < (unsigned)WQCAP ? (unsigned)wfill : (unsigned)WQCAP;
		for (unsigned e = (unsigned)wpub + lane; e < n && !frozen; e += 64) {
			unsigned long long ent;
			float v;
			asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(ent), "=&v"(v)
			             : "v"(qbuf_lds + 8u * e), "v"(qval_lds + 4u * e)
			             : "memory");
			const unsigned row = (unsigned)ent;
			typedef __attribute__((address_space(1))) unsigned *GU;
			__hip_atomic_fetch_min((GU)(a.gslot + (size_t)(unsigned)(ent >> 32) * NC) + (row & (unsigned)(NC - 1)), skey(v), __ATOMIC_RELAXED,
			                       __HIP_MEMORY_SCOPE_AGENT);
		}
		wpub = (int)n;
	};
	auto wdrain = [&]() __attribute__((always_inline)) { // publish what is left; the wave's whole queue -> the global stream behind one reservation
		publish();
		const unsigned n = (unsigned)wpub;
		wfill = 0;
		wpub = 0;
		if (!COLLECT || n == 0u)
			return;
		unsigned long long base = 0ull;
		if (lane == 0) { // (by hand: a compiled atomic with a result makes hipcc wait where the branches meet)
			const unsigned long long n64 = n;
			typedef __attribute__((address_space(1))) unsigned long long *GUL;
			asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(base)
			             : "v"((GUL)a.stream_cnt), "v"(n64)
			             : "memory");
		}
		const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
		const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
		base = ((unsigned long long)bhi << 32) | blo;
		for (unsigned e = lane; e < n; e += 64) {
			unsigned long long ent;
			float v;
			asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(ent), "=&v"(v)
			             : "v"(qbuf_lds + 8u * e), "v"(qval_lds + 4u * e)
			             : "memory");
			if ((long long)(base + e) < a.stream_cap) {
				typedef __attribute__((address_space(1))) unsigned long long *GUL;
				*((GUL)a.stream + (base + e)) = ent;
				if (a.stream_s) // (round 5: the final-bound filter -- launch_collect_final_thr, csrc/ivf_collect.hip launch_ivf_bucket_scatter)
					a.stream_s[base + e] = v;
			}
		}
		ovf |= (long long)(base + n) >= a.stream_cap ? 1 : 0; // (wave-uniform)
	};
	auto rare = [&](const f32x4acc (&sv)[2], int rb, int t, bool any_t, f32x2n cqv, long long row0, int nvalid, unsigned rowbits) {
		if (ABL & 1) {
			MVS_KEEP_VGPR(any_t);
			return;
		}
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(any_t) == 0ull, 1)) // (hot path = fall-through: no taken branch per half tile)
			return;
		int qo = qw;
		MVS_OPAQUE_VGPR(qo); // (keeps the per-query addresses of this path out of the hot loop's registers)
		unsigned m0 = 0u, m1 = 0u;
		if (any_t) {
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const bool in = 16 * rb + 4 * hq + r < nvalid;
				m0 |= (in && sv[0][r] >= cqv[0]) ? 1u << r : 0u;
				m1 |= (in && sv[1][r] >= cqv[1]) ? 1u << r : 0u;
			}
			if (SEL) { // rows the IDSelector rejects: neither a candidate nor evidence for the bound
				const unsigned rbits = (rowbits >> (16 * rb + 4 * hq)) & 15u;
				m0 &= rbits;
				m1 &= rbits;
			}
		}
		unsigned m = m0 | (m1 << 4); // bit 4 i + r: row r of column block i
		while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) { // (wave-uniform: every lane takes part in every step)
			const bool has = m != 0u;
			const int j8 = has ? __builtin_ctz(m) : 0;
			m &= m - 1u;
			const int i = j8 >> 2, j = j8 & 3;
			const float s0 = i ? sv[1][0] : sv[0][0], s1 = i ? sv[1][1] : sv[0][1], s2 = i ? sv[1][2] : sv[0][2], s3 = i ? sv[1][3] : sv[0][3];
			const float lo = (j & 1) ? s1 : s0;
			const float hi = (j & 1) ? s3 : s2;
			const float v = (j & 2) ? hi : lo;
			const unsigned row = (unsigned)(row0 + 16 * rb + 4 * hq + j);
			const unsigned q = (unsigned)(qo + 32 * t + 16 * i + c);
			// (round 5: a queue that cannot take this step's hits is drained then and there -- one reservation per queue; rounds 3-4 sent
			// every hit beyond the queue to the stream on its own, one returning atomic + wait each: see csrc/ivf_collect.hip)
			if (!(COLLECT && ovf) && __builtin_expect(wfill + (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(has)) > WQCAP, 0))
				wdrain();
			const bool counting = COLLECT && ovf; // the stream is full: publish at once, count, do not queue (see wdrain / the host's re-run)
			const unsigned long long act = __builtin_amdgcn_ballot_w64(has && !counting);
			const unsigned pos = (unsigned)wfill + __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
			wfill = __builtin_amdgcn_readfirstlane(wfill + (int)__builtin_popcountll(act));
			if (has) {
				const unsigned long long ent = ((unsigned long long)q << 32) | row;
				if (__builtin_expect(!counting, 1)) {
					asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" ::"v"(qbuf_lds + 8u * pos), "v"(ent), "v"(qval_lds + 4u * pos), "v"(v) : "memory");
				} else {
					// the stream is full: whatever is appended now is dropped, but the host wants the TRUE number of candidates (it sizes
					// the next attempt from it) -- publish, count, do not queue.  (All-duplicates data, 31 250 copies of every query's
					// nearest row: 3e8 candidates through the queue took 47 s per launch.)
					typedef __attribute__((address_space(1))) unsigned *GU;
					__hip_atomic_fetch_min((GU)(a.gslot + (size_t)q * NC) + (row & (unsigned)(NC - 1)), skey(v), __ATOMIC_RELAXED,
					                       __HIP_MEMORY_SCOPE_AGENT);
					const unsigned one = 1u;
					asm volatile("ds_add_u32 %0, %1" ::"v"(qcnt_lds + 28u), "v"(one) : "memory");
				}
			}
		}
	};

	for (int u = 0; u < ntiles; ++u) {
		// Shared bound: every `period` tiles the lane fetches the 16 class slots of the two column blocks it owns and WAITS for
		// them (one L2 round trip; the accumulators are dead here, so the transient registers are free).
		// (cadence 1, 1, 1, 1, then every 8 / 32 / 128 staged blocks: twice as sparse as first tuned, +1 % at the headline and at C2;
		// A/B: option cl_ksplit_opt bits 2..3: 1 = the old cadence, 2 / 3 = sparser still)
		const int pb = (a.opt >> 2) & 3, psh = pb == 1 ? 0 : (pb == 0 ? 1 : pb);
		const int period = u < 4 ? 1 : (u < 32 ? 4 << psh : (u < 256 ? 16 << psh : 64 << psh)); // (in staged blocks of CL_SUB tiles)
		const bool full = !frozen && (use_tab ? ((u + duty_phase) & duty_mask_at(u)) == 0 : (u % period) == 0);
		if (use_tab && !full && u > 0 && (u & ((1 << tab_shift) - 1)) == 0)
			dma_bounds(); // (lands before this block's barrier; until then the tiles use the entries already there)
		if (full) {
			publish(); // (this wave's own evidence is in the class slots before it reads them)
			// B = the kk-th best of the 16 class bests (kk distinct rows are at least that good): as keys, the kk-th smallest
			// (bitonic network in registers).  The pass bound B - 2E goes to the wave's table in LDS.
			int qo = qw;
			MVS_OPAQUE_VGPR(qo); // (keeps the per-query addresses of this block out of the hot loop's registers)
			f32x2n v = {0.f, 0.f};
#pragma unroll 1
			for (int i = 0; i < 2; ++i) { // one query at a time: 128 VGPRs of fragments are resident, the network needs ~40 more
				const int q = qo + 32 * hq + 16 * i + c;
				const int qc = q < a.nq ? q : 0;
				const float e2v = __builtin_nontemporal_load(a.e2 + qc);
				// NC = 128 (32 < kk <= 128): four SUBSETS of 32 classes (class = row & 127, subset = class >> 5); the ceil(kk / 4)-th best
				// of a subset's class bests has that many distinct rows at least as good, the WORST of the four subsets' values has
				// >= kk -- the same 32-key network four times instead of a 128-key one
				constexpr int SUBN = NC > 32 ? 32 : NC, NSUB = NC / SUBN;
				const int rank = NSUB == 1 ? a.nclass - 1 : (a.nclass + NSUB - 1) / NSUB - 1;
				unsigned kth = 0u;
#pragma unroll 1
				for (int sb = 0; sb < NSUB; ++sb) {
				const unsigned long long *src = (const unsigned long long *)(a.gslot + (size_t)qc * NC + sb * SUBN);
				unsigned long long w[SUBN / 2];
#pragma unroll
				for (int j = 0; j < SUBN / 2; ++j)
					w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
				for (int j = 0; j < SUBN / 2; ++j) // every load is issued before the first is consumed: one round trip
					asm volatile("" : "+v"(w[j]));
				unsigned key[SUBN];
#pragma unroll
				for (int j = 0; j < SUBN / 2; ++j) {
					key[2 * j] = (unsigned)w[j];
					key[2 * j + 1] = (unsigned)(w[j] >> 32);
				}
#pragma unroll
				for (int kbit = 2; kbit <= SUBN; kbit <<= 1)
#pragma unroll
					for (int jb = kbit >> 1; jb > 0; jb >>= 1)
#pragma unroll
						for (int x0 = 0; x0 < SUBN; ++x0) {
							const int x1 = x0 ^ jb;
							if (x1 > x0) {
								const unsigned lo = key[x0] < key[x1] ? key[x0] : key[x1];
								const unsigned hi = key[x0] < key[x1] ? key[x1] : key[x0];
								const bool asc = (x0 & kbit) == 0;
								key[x0] = asc ? lo : hi;
								key[x1] = asc ? hi : lo;
							}
						}
				unsigned ks = key[0];
#pragma unroll
				for (int j = 1; j < SUBN; ++j)
					ks = (rank == j) ? key[j] : ks;
				kth = ks > kth ? ks : kth; // (keys: smaller = better; the worst subset decides)
				}
				const unsigned neutral = skey(-FLT_MAX);
				const float B = skey2f(kth < neutral ? kth : neutral); // -FLT_MAX while fewer than kk classes are set
				const float bv = q < a.nq ? B - e2v : __uint_as_float(0x7fc00000u); // (2E = NaN stays NaN; NaN: nothing passes)
				v[0] = i == 0 ? bv : v[0];
				v[1] = i == 1 ? bv : v[1];
			}
			*(f32x2n *)(cqtab + (wave * 64 + hq * 16 + c) * 2) = v;
			if (use_tab) { // ... and for every other workgroup of this query block (agent scope: the XCDs' L2s are not coherent)
				unsigned long long bits;
				__builtin_memcpy(&bits, &v, 8);
				__hip_atomic_store((unsigned long long *)(a.pbnd + (size_t)qb * CL_QBLOCK + (wave * 64 + hq * 16 + c) * 2), bits, __ATOMIC_RELAXED,
				                   __HIP_MEMORY_SCOPE_AGENT);
			}
		}
#pragma unroll 1
		for (int sub = 0; sub < CL_SUB; ++sub) {
		// The A fragments of the WHOLE tile (8 x ds_read_b128 = 32 VGPRs) and the rows' beta (2 x ds_read_b128), by hand: the
		// reads are issued before the next tile's LDS-DMA (hipcc would put s_waitcnt vmcnt(0) in front of a compiled LDS read
		// issued after it) and each is waited for just before its first use.
		bf16x8 A[KB][2];
		f32x4n Y[2];
		{
			const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32c *)(nbuf + (u & 1) * 64 + sub * CL_BN + 4 * hq));
			asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(Y[0]), "=&v"(Y[1]) : "v"(nb_lds) : "memory");
			const unsigned ab =
			    (unsigned)(uintptr_t)((lds_f32c *)(smem + (((ABL & 4) ? 0 : (u & 1)) * STAGE_BYTES + sub * TILE_BYTES) / 4)) + rbase;
#pragma unroll
			for (int kb = 0; kb < KB; ++kb) {
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[kb][0]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
				asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(A[kb][1]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
			}
		}
		if (!(ABL & 4) && sub == 0) { // the next staged block, behind this tile's fragment reads
#pragma unroll
			for (int i = 0; i < DMA_PER_WAVE; ++i)
				dma_issue(u + 1, i);
			dma_norms(u + 1);
		}
		const long long row0 = r_begin + ((long long)u * CL_SUB + sub) * CL_BN;
		const int nvalid = (int)((r_end - row0) < CL_BN ? (r_end - row0) : CL_BN); // (<= 0 behind the split's last row)
		unsigned rowbits = 0xFFFFFFFFu;
		if (SEL) { // this tile's 32 selector bits: a wave-uniform (scalar) load; blocks start at multiples of 64 rows
			typedef __attribute__((address_space(4))) const unsigned cuint;
			rowbits = *((cuint *)a.rowmask + (row0 >> 5));
		}

		// Eight half tiles (32 queries x 16 rows: 8 MFMAs into two interleaved accumulators) in turn: while the matrix pipe
		// works on one half the vector ALU folds the PREVIOUS half (running maximum of s per query) and runs its
		// rare path, so a wave overlaps its own epilogue and does not depend on the CU's other workgroup being in the opposite
		// phase; only the last half's fold is exposed.  One accumulator set: half (t, rb) lives in acc[rb][*] until the same row
		// block of tile t + 1 starts, a full phase after its fold.
		f32x4acc acc[2][2]; // [row block][column block of the tile]
		f32x2n cqv[2];      // pass bounds of tile t in cqv[t & 1]
		float mx0 = -INFINITY, mx1 = -INFINITY;
		auto fold = [&](f32x4acc &p, int rb, int i) { // four rows of one query: the accumulator already holds s
			if (ABL & 2) {
				MVS_KEEP_VGPR(p);
				return;
			}
			(void)rb;
			if (i == 0) // two v_max3_f32
				mx0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx0, p[0]), p[1]), p[2]), p[3]);
			else
				mx1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx1, p[0]), p[1]), p[2]), p[3]);
		};
		auto any_of = [&](f32x2n cqp) { // NaN on either side: false
			const bool r = (mx0 >= cqp[0]) || (mx1 >= cqp[1]);
			mx0 = -INFINITY;
			mx1 = -INFINITY;
			return r;
		};
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			// this tile's bounds (written at a refresh by the owning lane; LDS keeps a wave's accesses in order)
			asm volatile("ds_read_b64 %0, %1" : "=v"(cqv[t & 1]) : "v"(cq_lds + (unsigned)(t * 128)) : "memory");
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				const int prb = rb ^ 1, pt = rb == 0 ? t - 1 : t; // the half folded under this one
#pragma unroll
				for (int kb = 0; kb < KB; ++kb) {
					if (t == 0) { // beta and A[kb][rb] have arrived (LDS returns in order: the reads behind them are counted)
						if (rb == 0 && kb == 0)
							asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(A[0][0]), "+v"(Y[0]), "+v"(Y[1]));
						else if (rb == 0 && kb == 1)
							asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(A[1][0]));
						else if (rb == 0 && kb == 2)
							asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A[2][0]));
						else if (rb == 0 && kb == 3)
							asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[3][0]));
						else if (rb == 1 && kb == 0) // everything: the other row block's fragments, the bounds
							asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0][1]), "+v"(A[1][1]), "+v"(A[2][1]), "+v"(A[3][1]), "+v"(cqv[0]));
					}
#pragma unroll
					for (int i = 0; i < 2; ++i) {
						if (kb == 0) // the chain starts at beta(row): s = alpha <x', y'> + beta comes out of the matrix pipe
							acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + i][kb], Y[rb], 0, 0, 0);
						else
							acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + i][kb], acc[rb][i], 0, 0, 0);
					}
					if (pt >= 0 && kb < 2)
						fold(acc[prb][kb], prb, kb);
					__builtin_amdgcn_sched_barrier(0);
				}
				if (pt >= 0) {
					if (rb == 0) // (pt = t - 1: its bounds were read a tile ago; this tile's read is waited for as well)
						asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cqv[0]), "+v"(cqv[1]));
					rare(acc[prb], prb, pt, any_of(cqv[pt & 1]), cqv[pt & 1], row0, nvalid, rowbits);
				}
			}
		}
		{
			fold(acc[1][0], 1, 0);
			fold(acc[1][1], 1, 1);
			rare(acc[1], 1, 3, any_of(cqv[1]), cqv[1], row0, nvalid, rowbits);
		}
		} // sub
		if (ABL & 8) // profiling: no workgroup barrier (the waves drift apart; results wrong)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		else
			__syncthreads(); // also drains this block's LDS-DMA (vmcnt(0)) before the next block reads it
		if ((u % CL_FLUSH_EVERY) == CL_FLUSH_EVERY - 1 || u == ntiles - 1) {
			// (no LDS-DMA is in flight between the barrier above and the next tile's first issue)
			// every wave looks after its own queue: what it recorded since the last look goes to the class slots; the queue goes to
			// the stream when it is half full, and at the end
			if (wfill >= WQCAP / 2 || u == ntiles - 1)
				wdrain();
			else
				publish();
		}
	}
	if (COLLECT) { // (candidates that were only counted after the stream filled up: see the rare path)
		__syncthreads();
		if (tid == 0 && qctl[7] != 0u)
			atomicAdd(a.stream_cnt, (unsigned long long)qctl[7]);
	}
}

static size_t collect_lds_bytes(const FlatGeom &g) {
	(void)g;
	return (size_t)2 * CL_SUB * CL_BN * 128 * 2 + 2 * 64 * 4 + (size_t)CL_QCAP * 8 + (size_t)CL_QBLOCK * 4 + 64 + (size_t)CL_QCAP * 4;
}

bool collect_supported(const FlatGeom &g) {
	return collect_store_dims(g.d) > 0;
}

// one bit per row: does the IDSelector accept it?  (ids as the SEL instances of the f32 kernel see them: idmap[row] behind
// an IndexIDMap, the row number otherwise.)  One wave writes one 64-bit word; rows >= n: 0.
__device__ __forceinline__ bool cl_sel_member(const SelectorDev &s, long long id) {
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
__global__ __launch_bounds__(256) void collect_rowmask_kernel(SelectorDev sel, const long long *__restrict__ idmap, long long n,
                                                             long long nwords, unsigned long long *__restrict__ mask) {
	const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	bool ok = false;
	if (row < n)
		ok = cl_sel_member(sel, idmap ? idmap[row] : row);
	const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
	if ((threadIdx.x & 63) == 0 && (row >> 6) < nwords)
		mask[row >> 6] = b;
}
size_t collect_rowmask_bytes(int64_t n) {
	return (size_t)((n + 63) / 64 + 64) * 8; // + slack: tiles past the last row are looked up, never used
}
void launch_collect_rowmask(SelectorDev sel, const int64_t *d_idmap, int64_t n, unsigned long long *d_mask, hipStream_t st) {
	const long long nwords = (long long)(collect_rowmask_bytes(n) / 8);
	const long long rows = nwords * 64;
	hipLaunchKernelGGL(collect_rowmask_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, sel,
	                   (const long long *)d_idmap, (long long)n, nwords, d_mask);
	MVS_HIP(hipGetLastError());
}


int flat_mfma_slot_stride(int64_t k);
__global__ void init_gslot_kernel(unsigned *g, long long total, int stride, int k, int is_l2);

// The pass-bound table of the d <= 128 scan from the class slots as they stand: entry j of query block qb = the bound of query
// qb * 512 + 128 w + 32 hq + 16 i + c with j = 128 w + 2 (16 hq + c) + i -- the order of the workgroup's LDS table.  B = the kk-th
// best of the NC class bests (as keys: the kk-th smallest), pass bound = B - 2E; NaN (nothing passes) behind the last query and
// for the queries without a finite 2E.  Runs in front of every scan launch (slots warm from the pre-pass or a previous attempt).
template <int NC>
__global__ void collect_bound_table_kernel(const unsigned *__restrict__ gslot, const float *__restrict__ e2, int nclass, int nq,
                                           long long total, float *__restrict__ pbnd) {
	const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= total)
		return;
	const int o = (int)(j & 127), w = (int)((j >> 7) & 3);
	const long long qb = j >> 9;
	const int i = o & 1, c = (o >> 1) & 15, hq = o >> 5;
	const long long q = qb * CL_QBLOCK + w * 128 + 32 * hq + 16 * i + c;
	float bv = __uint_as_float(0x7fc00000u);
	if (q < nq) {
		constexpr int SUBN = NC > 32 ? 32 : NC, NSUB = NC / SUBN; // (flat_bf16_collect_kernel: subsets of 32 classes at NC = 128)
		const int rank = NSUB == 1 ? nclass - 1 : (nclass + NSUB - 1) / NSUB - 1;
		unsigned kth = 0u;
		for (int sb = 0; sb < NSUB; ++sb) {
			unsigned key[SUBN];
#pragma unroll
			for (int t = 0; t < SUBN; ++t)
				key[t] = gslot[(size_t)q * NC + sb * SUBN + t];
			// the rank-th smallest with duplicates counted: the key whose rank interval covers it
			unsigned ks = 0xffffffffu;
#pragma unroll
			for (int t = 0; t < SUBN; ++t) {
				int less = 0, leq = 0;
#pragma unroll
				for (int s2 = 0; s2 < SUBN; ++s2) {
					less += key[s2] < key[t];
					leq += key[s2] <= key[t];
				}
				if (less <= rank && rank < leq)
					ks = key[t];
			}
			kth = ks > kth ? ks : kth;
		}
		const unsigned neutral = skey(-FLT_MAX);
		const float B = skey2f(kth < neutral ? kth : neutral);
		bv = B - e2[q];
	}
	pbnd[j] = bv;
}
static void launch_collect_bound_table(const CollectArgs &a, int nqb, hipStream_t st) {
	if (!a.pbnd)
		return;
	const long long total = (long long)nqb * CL_QBLOCK;
	const dim3 grid((unsigned)((total + 255) / 256));
	if (a.slot_stride == 128)
		hipLaunchKernelGGL(collect_bound_table_kernel<128>, grid, dim3(256), 0, st, (const unsigned *)a.gslot, a.e2, a.nclass, a.nq, total, a.pbnd);
	else if (a.slot_stride == 32)
		hipLaunchKernelGGL(collect_bound_table_kernel<32>, grid, dim3(256), 0, st, (const unsigned *)a.gslot, a.e2, a.nclass, a.nq, total, a.pbnd);
	else
		hipLaunchKernelGGL(collect_bound_table_kernel<16>, grid, dim3(256), 0, st, (const unsigned *)a.gslot, a.e2, a.nclass, a.nq, total, a.pbnd);
	MVS_HIP(hipGetLastError());
}
size_t collect_bound_table_bytes(int64_t nq) {
	return (size_t)((nq + CL_QBLOCK - 1) / CL_QBLOCK) * CL_QBLOCK * sizeof(float) + 1024;
}

template <bool COLLECT>
static void launch_collect_range(const FlatGeom &g, int metric, CollectArgs a, int64_t row_first, int64_t row_end,
                                 int64_t nsplit_want, int64_t nq, hipStream_t st, int *grid_out, int *nsplit_out) {
	const int nqb = (int)((nq + CL_QBLOCK - 1) / CL_QBLOCK);
	if (!tune().cl_tab && !(a.opt & 256))
		a.pbnd = nullptr;
	if (!(a.opt & 256)) // (frozen bounds: the caller filled the table -- launch_collect_big_bounds)
		launch_collect_bound_table(a, nqb, st);
	const int64_t ntiles = (row_end - row_first + CL_SUB * CL_BN - 1) / (CL_SUB * CL_BN); // staged blocks
	const int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(nsplit_want, ntiles));
	a.xcd_map = (nsplit >= 8 && nsplit % 8 == 0) ? 1 : 0;
	a.row_first = row_first;
	a.n = row_end;
	a.split_rows = (ntiles + nsplit - 1) / nsplit * (CL_SUB * CL_BN);
	a.nqb = nqb;
	a.nsplit = (int)nsplit;
	const int grid = nqb * (int)nsplit;
	const size_t lds = collect_lds_bytes(g);
#ifdef MVS_PROFILING
	if (metric == METRIC_L2 && COLLECT && tune().cl_abl) {
#define MVS_CL_ABL(N)                                                                                                  \
	if (tune().cl_abl == N) {                                                                                               \
		auto kern = flat_bf16_collect_kernel<8, true, true, N>;                                                        \
		ensure_dynamic_lds((const void *)kern, lds);                                                                   \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                                   \
	}
		MVS_CL_ABL(1) MVS_CL_ABL(3) MVS_CL_ABL(7) MVS_CL_ABL(11) MVS_CL_ABL(15)
#undef MVS_CL_ABL
	} else
#endif
	if (a.slot_stride == 32 || a.slot_stride == 128) { // 16 < kk <= 32: 32 row classes; 32 < kk <= 128: 4 subsets of 32
#define MVS_CL_NC32(L2, SEL_)                                                                                           \
	{                                                                                                                  \
		if (a.slot_stride == 128) {                                                                                    \
			auto kern = flat_bf16_collect_kernel<8, L2, COLLECT, 0, SEL_, 128>;                                        \
			ensure_dynamic_lds((const void *)kern, lds);                                                               \
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                               \
		} else {                                                                                                       \
			auto kern = flat_bf16_collect_kernel<8, L2, COLLECT, 0, SEL_, 32>;                                         \
			ensure_dynamic_lds((const void *)kern, lds);                                                               \
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                               \
		}                                                                                                              \
	}
		if (a.rowmask && metric == METRIC_L2)
			MVS_CL_NC32(true, true)
		else if (a.rowmask)
			MVS_CL_NC32(false, true)
		else if (metric == METRIC_L2)
			MVS_CL_NC32(true, false)
		else
			MVS_CL_NC32(false, false)
#undef MVS_CL_NC32
	} else if (a.rowmask) {
		if (metric == METRIC_L2) {
			auto kern = flat_bf16_collect_kernel<8, true, COLLECT, 0, true>;
			ensure_dynamic_lds((const void *)kern, lds);
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
		} else {
			auto kern = flat_bf16_collect_kernel<8, false, COLLECT, 0, true>;
			ensure_dynamic_lds((const void *)kern, lds);
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
		}
	} else if (metric == METRIC_L2) {
		auto kern = flat_bf16_collect_kernel<8, true, COLLECT>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
	} else {
		auto kern = flat_bf16_collect_kernel<8, false, COLLECT>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
	}
	MVS_HIP(hipGetLastError());
	if (grid_out)
		*grid_out = grid;
	if (nsplit_out)
		*nsplit_out = (int)nsplit;
}


// ---- bound estimation pre-pass (d <= 128): class maxima in REGISTERS ---------------------------------------------------------------
// The COLLECT = false instance of the scan kernel warms the class slots through its rare path: with no bound yet EVERY value
// passes, i.e. one global atomic per (query, row) pair -- 0.26 ms for 3 900 rows at C2, 0.4 ms for 16 384 at the headline, a
// fixed cost every search, every DuckDB chunk and every row shard pays.  The accumulator layout makes
// the atomics unnecessary: lane (hq, c) holds rows 16 rb + 4 hq + r of query c of a column block, and every tile starts at a
// multiple of 32 rows, so register r of the lane ALWAYS belongs to row class 4 hq + r of that query.  The lane keeps a running
// maximum per (column block, r) -- 32 registers -- and publishes 32 atomics at the very end: nq x 16 x (row splits) atomics per
// search instead of nq x rows.  Same geometry, staging and fragment reads as the scan kernel; no bounds, no queue, no stream.
template <bool IS_L2, bool SEL>
__global__ __launch_bounds__(256, 2) void flat_bf16_seed_kernel(const CollectArgs a) {
	constexpr int KB = 4, PITCH = 256;
	constexpr int TILE_BYTES = CL_BN * PITCH, STAGE_BYTES = CL_SUB * TILE_BYTES;
	constexpr int DMA_PER_WAVE = STAGE_BYTES / 4096;
	extern __shared__ __attribute__((aligned(16))) float smem[];
	float *nbuf = (float *)((char *)smem + 2 * STAGE_BYTES); // [2][64] beta of the staged rows
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int hq = lane >> 4, c = lane & 15;
	const int split = blockIdx.x / a.nqb, qb = blockIdx.x % a.nqb;
	const long long r_begin = a.row_first + (long long)split * a.split_rows;
	long long r_end = r_begin + (a.split_len > 0 ? a.split_len : a.split_rows); // (split_len: the big lists' pass A looks at a part of every stride)
	if (r_end > a.n)
		r_end = a.n;
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin) / (CL_SUB * CL_BN)) : 0; // whole staged blocks only (the host rounds)
	const int qw = qb * CL_QBLOCK + wave * 128;
	bf16x8 bq[8][KB];
	{
		const bf16x8 *qsrc = (const bf16x8 *)a.qf;
#pragma unroll
		for (int cb = 0; cb < 8; ++cb) {
			const size_t qblk16 = (size_t)qb * (CL_QBLOCK / 16) + wave * 8 + cb;
#pragma unroll
			for (int kb = 0; kb < KB; ++kb)
				bq[cb][kb] = qsrc[(qblk16 * KB + kb) * 64 + lane];
		}
	}
	unsigned dma_off;
	{
		const int rr = 4 * wave + (lane >> 4);
		dma_off = (unsigned)(rr * PITCH + (((lane & 15) ^ rr) * 16));
	}
	auto dma_block = [&](int u) {
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i) {
			const char *base = (const char *)a.yb + ((size_t)(r_begin + (long long)u * (CL_SUB * CL_BN)) + (size_t)i * 16) * PITCH; // uniform
			__builtin_amdgcn_global_load_lds((glb_f32c *)(base + dma_off),
			                                 (lds_f32c *)(smem + ((u & 1) * STAGE_BYTES + (i * 4 + wave) * 1024) / 4), 16, 0, 0);
		}
		const float *nb = a.yn + (r_begin + (long long)u * (CL_SUB * CL_BN)); // uniform
		__builtin_amdgcn_global_load_lds((glb_f32c *)(nb + lane), (lds_f32c *)(smem + (2 * STAGE_BYTES) / 4 + (u & 1) * 64), 4, 0, 0);
	};
	if (ntiles > 0)
		dma_block(0);
	__syncthreads();
	const unsigned rbase = (unsigned)(c * PITCH) | (unsigned)(((hq ^ c) & 15) * 16);
	f32x4acc cm[8]; // running maximum of s: [column block][r] = class 4 hq + r of query qw + 16 cb + c
#pragma unroll
	for (int cb = 0; cb < 8; ++cb)
		cm[cb] = f32x4acc {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
	for (int u = 0; u < ntiles; ++u) {
#pragma unroll 1
		for (int sub = 0; sub < CL_SUB; ++sub) {
			bf16x8 A[KB][2];
			f32x4n Y[2];
			{
				const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32c *)(nbuf + (u & 1) * 64 + sub * CL_BN + 4 * hq));
				asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(Y[0]), "=&v"(Y[1]) : "v"(nb_lds) : "memory");
				const unsigned ab = (unsigned)(uintptr_t)((lds_f32c *)(smem + ((u & 1) * STAGE_BYTES + sub * TILE_BYTES) / 4)) + rbase;
#pragma unroll
				for (int kb = 0; kb < KB; ++kb) {
					asm volatile("ds_read_b128 %0, %1" : "=v"(A[kb][0]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
					asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(A[kb][1]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
				}
			}
			if (sub == 0)
				dma_block(u + 1); // (past the range: padding rows or the rows behind it, fetched and never used)
			unsigned rowbits = 0xFFFFFFFFu;
			if (SEL) {
				typedef __attribute__((address_space(4))) const unsigned cuint;
				const long long row0 = r_begin + ((long long)u * CL_SUB + sub) * CL_BN;
				rowbits = *((cuint *)a.rowmask + (row0 >> 5));
			}
			asm volatile("s_waitcnt lgkmcnt(0)"
			             : "+v"(Y[0]), "+v"(Y[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[2][0]), "+v"(A[2][1]),
			               "+v"(A[3][0]), "+v"(A[3][1]));
			f32x4acc acc[2][2];
			auto fold = [&](int rb, int t) { // half (t, rb): rows 16 rb + 4 hq + r, column blocks 2 t, 2 t + 1
#pragma unroll
				for (int i = 0; i < 2; ++i)
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						float v = acc[rb][i][r];
						if (SEL) // rows the IDSelector rejects are no evidence for the bound
							v = ((rowbits >> (16 * rb + 4 * hq + r)) & 1u) ? v : -INFINITY;
						cm[2 * t + i][r] = __builtin_fmaxf(cm[2 * t + i][r], v); // (NaN: ignored)
					}
			};
#pragma unroll
			for (int t = 0; t < 4; ++t) {
#pragma unroll
				for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
					for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
						for (int i = 0; i < 2; ++i) {
							if (kb == 0)
								acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + i][kb], Y[rb], 0, 0, 0);
							else
								acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + i][kb], acc[rb][i], 0, 0, 0);
						}
						if (kb == 1 && (t > 0 || rb > 0)) // the previous half, behind this half's MFMAs
							fold(rb ^ 1, rb == 0 ? t - 1 : t);
						__builtin_amdgcn_sched_barrier(0);
					}
				}
			}
			fold(1, 3);
		}
		__syncthreads(); // also drains this block's LDS-DMA before the next block reads it
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (a.seed_stage) {
		// Round 5: the split's class maxima as ONE coalesced 16-byte store per lane and column block (16 queries x 64 bytes = 1 KB per
		// instruction) into [split][query][16]; collect_seed_reduce_kernel takes the maximum over the splits.  The atomics below -- nq x 16
		// x nsplit = 4 M at the headline's batch, all at the kernel's end -- were 38 of its 99 us (profiles/r5_c3_ab.txt, 12).
#pragma unroll
		for (int cb = 0; cb < 8; ++cb) {
			const int q = qw + 16 * cb + c;
			if (q < a.nq)
				*(f32x4acc *)(a.seed_stage + ((size_t)split * (size_t)a.nq + (size_t)q) * 16 + 4 * hq) = cm[cb]; // (-inf: no row seen)
		}
		return;
	}
	if (ntiles > 0) {
#pragma unroll
		for (int cb = 0; cb < 8; ++cb) {
			const int q = qw + 16 * cb + c;
			if (q < a.nq) {
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (cm[cb][r] > -INFINITY) {
						typedef __attribute__((address_space(1))) unsigned *GU;
						__hip_atomic_fetch_min((GU)(a.gslot + (size_t)q * 16) + (4 * hq + r), skey(cm[cb][r]), __ATOMIC_RELAXED,
						                       __HIP_MEMORY_SCOPE_AGENT);
					}
			}
		}
	}
}
// one thread per (query, four classes): the maximum over the row splits' staged class maxima -> the class slots (neutral where no split saw a row)
__global__ void collect_seed_reduce_kernel(const float *__restrict__ stage, int nsplit, long long nq, unsigned *__restrict__ gslot) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // = 4 q + quad
	if (i >= nq * 4)
		return;
	f32x4acc m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
	for (int s = 0; s < nsplit; ++s) {
		const f32x4acc v = *(const f32x4acc *)(stage + ((size_t)s * (size_t)nq * 16) + (size_t)i * 4);
#pragma unroll
		for (int r = 0; r < 4; ++r)
			m[r] = __builtin_fmaxf(m[r], v[r]); // (NaN: ignored, as the atomics on keys did)
	}
#pragma unroll
	for (int r = 0; r < 4; ++r)
		if (m[r] > -INFINITY) {
			const unsigned k = skey(m[r]);
			unsigned *p = gslot + (size_t)i * 4 + r;
			*p = k < *p ? k : *p; // (the slots are neutral, or hold what an earlier attempt left: keep the better)
		}
}
static void launch_collect_seed(int metric, CollectArgs a, int64_t rows, int64_t nq, hipStream_t st) {
	const int nqb = (int)((nq + CL_QBLOCK - 1) / CL_QBLOCK);
	const int64_t nblocks = rows / (CL_SUB * CL_BN); // whole staged blocks of 64 rows
	if (nblocks <= 0)
		return;
	// one round of the 512 resident workgroups where the batch allows it, >= 8 staged blocks per workgroup
	const int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(8, 512 / nqb), nblocks / 8));
	a.row_first = 0;
	a.split_rows = (nblocks + nsplit - 1) / nsplit * (CL_SUB * CL_BN);
	a.n = nblocks * (CL_SUB * CL_BN);
	a.nqb = nqb;
	a.nsplit = (int)((nblocks * (CL_SUB * CL_BN) + a.split_rows - 1) / a.split_rows);
	const int grid = nqb * a.nsplit;
	const size_t lds = (size_t)2 * CL_SUB * CL_BN * 256 + 2 * 64 * 4 + 64;
#define MVS_SEED(L2, SL)                                                                                           \
	{                                                                                                              \
		auto kern = flat_bf16_seed_kernel<L2, SL>;                                                                 \
		ensure_dynamic_lds((const void *)kern, lds);                                                               \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                               \
	}
	if (metric == METRIC_L2 && a.rowmask)
		MVS_SEED(true, true)
	else if (metric == METRIC_L2)
		MVS_SEED(true, false)
	else if (a.rowmask)
		MVS_SEED(false, true)
	else
		MVS_SEED(false, false)
#undef MVS_SEED
	if (a.seed_stage)
		hipLaunchKernelGGL(collect_seed_reduce_kernel, dim3((unsigned)((nq * 4 + 255) / 256)), dim3(256), 0, st, (const float *)a.seed_stage, a.nsplit,
		                   (long long)nq, a.gslot);
	MVS_HIP(hipGetLastError());
}
size_t collect_seed_stage_bytes(int64_t nq) { // (launch_collect_seed: at most 64 row splits)
	return (size_t)64 * (size_t)nq * 16 * sizeof(float);
}

// row classes per query: 16, or 32 for 16 < kk <= 32 (d <= 128 only: the wide instances keep 16)
// (headline shape: kk = 32 on 32 classes admits 1 269 candidates per query -- the bound is the WORST class best --, on 4 x 32 classes 512;
// but the 128-class instance's derivation is four networks and its scan ran 23.7 vs 23.2 ms there, 23.4 vs 20.3 at kk = 25: from 33 on)
int collect_slot_stride(int kk, int dp1) {
	if (dp1 == 128 && kk > 28) // (with the derivation every 256 blocks: kk = 33 on 4 x 32 classes 21.0 ms, kk = 32 on 32 classes 23.3)
		return 128;
	if (dp1 > 128 && kk > 32 && collect_wide_max_classes(dp1) >= 128) // (round 6: the wide stores serve kk <= 128 the same way)
		return 128;
	return (kk > 16 || kk >= tune().cl_nc32_from) ? 32 : 16;
}
int collect_max_k(int d) {
	const int dp1 = collect_store_dims(d);
	if (dp1 == 0)
		return 0;
	// (the k-split kernel -- option cl_wide_big = 0 at the 768 / 1024-dim stores -- keeps 16 classes; the d <= 128 scan serves
	// kk <= 128 with four subsets of 32 classes, round 4)
	if (dp1 == 128)
		return 128;
	return collect_wide_max_classes(dp1); // (128, or the k-split kernel's 16)
}

// thr[q] = B - 2E with the class slots as the scan LEFT them (linear in q): the scan admitted a row when s >= B_then - 2E; bounds only
// tighten, so an entry with s < thr[q] is not in the result (round 5, the final-bound filter: csrc/ivf_collect.hip)
template <int NC>
__global__ void collect_final_thr_kernel(const unsigned *__restrict__ gslot, const float *__restrict__ e2, int nclass, long long nq,
                                         float *__restrict__ thr) {
	const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nq)
		return;
	constexpr int SUBN = NC > 32 ? 32 : NC, NSUB = NC / SUBN;
	const int rank = NSUB == 1 ? nclass - 1 : (nclass + NSUB - 1) / NSUB - 1;
	unsigned kth = 0u;
	for (int sb = 0; sb < NSUB; ++sb) {
		unsigned key[SUBN];
#pragma unroll
		for (int t = 0; t < SUBN; ++t)
			key[t] = gslot[(size_t)q * NC + sb * SUBN + t];
		unsigned ks = 0xffffffffu;
#pragma unroll
		for (int t = 0; t < SUBN; ++t) {
			int less = 0, leq = 0;
#pragma unroll
			for (int s2 = 0; s2 < SUBN; ++s2) {
				less += key[s2] < key[t];
				leq += key[s2] <= key[t];
			}
			if (less <= rank && rank < leq)
				ks = key[t];
		}
		kth = ks > kth ? ks : kth;
	}
	const unsigned neutral = skey(-FLT_MAX);
	const float B = skey2f(kth < neutral ? kth : neutral);
	thr[q] = B - e2[q]; // (the scan's own arithmetic; NaN: nothing of the query is in the stream)
}
// ---- lists beyond 128 entries (round 6; VERDICT r5 missing #3: they went to the f32 kernels -- 104 ms for k = 129 against 8 ms for
// k = 128 at 2 048 queries, 3 s for k = 1000).  The class slots cannot bound a k-th value for k > their number, but P disjoint ROW
// RANGES can: range p's slots give B_p with >= ceil(k / P) distinct rows of the range at least that good (s >= B_p), so
// T = min_p B_p has >= k rows at least that good over the ranges and the exact k-th best value is >= T - E: every row of the result has
// s >= T - 2E.  Pass A estimates the B_p (bound estimation only, the scan kernel's COLLECT = false instances, 128 classes per range,
// ceil(k / P) <= 64 so that the bound sits near the range's ceil(k / P)-th best and not at its worst class); the ranges need not cover
// the database -- any rows give a valid T, fewer rows a lower one -- so they are P strides of a FRACTION of it.  Pass B scans every row
// against the frozen T - 2E (a.opt bit 8) and streams what passes; the candidates are re-scored exactly, sorted per query
// (rocPRIM segmented radix sort) and the first k taken: launch_collect_select_big.
__global__ void collect_bound_table_multi_kernel(const unsigned *__restrict__ gslot, long long range_stride, int nranges,
                                                 const float *__restrict__ e2, int nclass, int nq, long long total, float *__restrict__ pbnd,
                                                 int linear) {
	const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= total)
		return;
	const int o = (int)(j & 127), w = (int)((j >> 7) & 3);
	const long long qb = j >> 9;
	const int i = o & 1, c = (o >> 1) & 15, hq = o >> 5;
	// (the d <= 128 scan's table order: collect_bound_table_kernel; the wide stores' kernels read one bound per query: linear)
	const long long q = linear ? j : qb * CL_QBLOCK + w * 128 + 32 * hq + 16 * i + c;
	float bv = __uint_as_float(0x7fc00000u);
	if (q < nq) {
		const int rank = (nclass + 3) / 4 - 1; // four subsets of 32 classes per range
		unsigned kth = 0u;
		for (int p = 0; p < nranges; ++p)
			for (int sb = 0; sb < 4; ++sb) {
				unsigned key[32];
#pragma unroll
				for (int t = 0; t < 32; ++t)
					key[t] = gslot[(size_t)p * range_stride + (size_t)q * 128 + sb * 32 + t];
				unsigned ks = 0xffffffffu;
#pragma unroll
				for (int t = 0; t < 32; ++t) {
					int less = 0, leq = 0;
#pragma unroll
					for (int s2 = 0; s2 < 32; ++s2) {
						less += key[s2] < key[t];
						leq += key[s2] <= key[t];
					}
					if (less <= rank && rank < leq)
						ks = key[t];
				}
				kth = ks > kth ? ks : kth; // (keys: smaller = better; the worst subset of the worst range decides)
			}
		const unsigned neutral = skey(-FLT_MAX);
		const float B = skey2f(kth < neutral ? kth : neutral);
		bv = B - e2[q];
	}
	pbnd[j] = bv;
}
// pass A of the d <= 128 store on the REGISTER pre-pass kernel (flat_bf16_seed_kernel: a split's 16 class maxima, no atomics, no bounds,
// the matrix pipe's pace): the splits are the ranges -- P = ceil(k / 8) strides of the database, split_len rows of each scanned --
// and B_p = the ceil(k / P)-th best of split p's 16 class maxima.  (The class-slot version below took 13 ms of a 26 ms k = 1000 batch:
// every range starts with cold slots, so its first rows all pass -- and 128-class derivations every four blocks; it stays for the wide
// stores, which have no register pre-pass.)
__global__ void collect_bound_table_seed_kernel(const float *__restrict__ stage, int nsplit, int rank, const float *__restrict__ e2, int nq,
                                                long long total, float *__restrict__ pbnd) {
	const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= total)
		return;
	const int o = (int)(j & 127), w = (int)((j >> 7) & 3);
	const long long qb = j >> 9;
	const int i = o & 1, c = (o >> 1) & 15, hq = o >> 5;
	const long long q = qb * CL_QBLOCK + w * 128 + 32 * hq + 16 * i + c; // (the table's order: collect_bound_table_kernel)
	float bv = __uint_as_float(0x7fc00000u);
	if (q < nq) {
		float T = INFINITY;
		for (int p = 0; p < nsplit; ++p) {
			float v[16];
			const f32x4acc *src = (const f32x4acc *)(stage + ((size_t)p * (size_t)nq + (size_t)q) * 16);
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const f32x4acc x = src[t];
				v[4 * t] = x[0], v[4 * t + 1] = x[1], v[4 * t + 2] = x[2], v[4 * t + 3] = x[3];
			}
			// the (rank + 1)-th largest of the 16 maxima, duplicates counted (-inf: a class without a row; NaN compares false: treated as -inf)
			float b = -INFINITY;
#pragma unroll
			for (int t = 0; t < 16; ++t) {
				const float vt = v[t] == v[t] ? v[t] : -INFINITY;
				int greater = 0, geq = 0;
#pragma unroll
				for (int s2 = 0; s2 < 16; ++s2) {
					const float vs = v[s2] == v[s2] ? v[s2] : -INFINITY;
					greater += vs > vt;
					geq += vs >= vt;
				}
				if (greater <= rank && rank < geq)
					b = vt;
			}
			T = b < T ? b : T;
		}
		// (-inf: some split has fewer than rank + 1 classes with a row -- no bound, everything passes; the scan kernel's own neutral value)
		const float B = T > -FLT_MAX ? T : -FLT_MAX;
		bv = B - e2[q];
	}
	pbnd[j] = bv;
}
void launch_collect_big_bounds_seed(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms, int64_t n,
                                    int64_t nq, int kf, int nsplits, int64_t split_len, const float *d_e2, float *d_stage,
                                    const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st) {
	CollectArgs a;
	memset(&a, 0, sizeof a);
	a.qf = d_qf, a.yb = d_rows, a.yn = d_norms, a.e2 = d_e2;
	a.nq = (int)nq;
	a.rowmask = d_rowmask;
	a.seed_stage = d_stage;
	const int nqb = (int)((nq + CL_QBLOCK - 1) / CL_QBLOCK);
	a.row_first = 0;
	a.split_rows = (n / nsplits) / 64 * 64; // the stride
	a.split_len = std::min<int64_t>(split_len, a.split_rows) / 64 * 64;
	a.n = n / 64 * 64;
	a.nqb = nqb;
	a.nsplit = nsplits;
	const int grid = nqb * nsplits;
	const size_t lds = (size_t)2 * CL_SUB * CL_BN * 256 + 2 * 64 * 4 + 64;
#define MVS_SEED(L2, SL)                                                                                           \
	{                                                                                                              \
		auto kern = flat_bf16_seed_kernel<L2, SL>;                                                                 \
		ensure_dynamic_lds((const void *)kern, lds);                                                               \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                               \
	}
	if (metric == METRIC_L2 && a.rowmask)
		MVS_SEED(true, true)
	else if (metric == METRIC_L2)
		MVS_SEED(true, false)
	else if (a.rowmask)
		MVS_SEED(false, true)
	else
		MVS_SEED(false, false)
#undef MVS_SEED
	const int kfp = (kf + nsplits - 1) / nsplits;
	const long long total = (long long)nqb * CL_QBLOCK;
	hipLaunchKernelGGL(collect_bound_table_seed_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, (const float *)d_stage, nsplits,
	                   kfp - 1, d_e2, (int)nq, total, d_pbnd);
	MVS_HIP(hipGetLastError());
}
// pass A: d_gslot [nranges][nq][128] -> d_pbnd (the scan's table order) = T - 2E per query.  Range p = rows [p n / P, p n / P + range_rows).
void launch_collect_big_bounds(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms, int64_t n,
                               int64_t nq, int kf, int nranges, int64_t range_rows, const float *d_e2, unsigned *d_gslot,
                               const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st) {
	const long long gtotal = (long long)nranges * nq * 128;
	hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gslot, gtotal, 128, 128, 0 /* larger s is better */);
	const int kfp = (kf + nranges - 1) / nranges;
	const int nqb = (int)((nq + CL_QBLOCK - 1) / CL_QBLOCK);
	const int dp1 = collect_store_dims(g.d);
	for (int p = 0; p < nranges; ++p) {
		CollectArgs a;
		memset(&a, 0, sizeof a);
		a.qf = d_qf, a.yb = d_rows, a.yn = d_norms, a.e2 = d_e2;
		a.gslot = d_gslot + (size_t)p * nq * 128;
		a.slot_stride = 128;
		a.nclass = kfp;
		a.nq = (int)nq;
		a.rowmask = d_rowmask;
		a.opt = tune().ksplit_opt;
		// (disjoint ranges: the proof counts a row once -- strides of whole 192-row units, a range no longer than its stride)
		const int64_t stride = (n / nranges) / 192 * 192, r0 = (int64_t)p * stride, r1 = std::min<int64_t>(n, r0 + std::min<int64_t>(range_rows, stride));
		if (dp1 > 128) { // (the wide stores' kernels derive their bounds per wave: no table)
			const int nqbw = (int)((nq + collect_wide_qblock(dp1) - 1) / collect_wide_qblock(dp1));
			launch_collect_wide_range(dp1, metric, false, a, r0, r1, std::max<int64_t>(8, std::min<int64_t>(64, 1024 / nqbw)), nq, st, nullptr, nullptr);
		} else {
			a.pbnd = d_pbnd; // (derived from this range's slots in front of its launch and refreshed by its workgroups)
			launch_collect_range<false>(g, metric, a, r0, r1, std::max<int64_t>(8, std::min<int64_t>(64, 1024 / nqb)), nq, st, nullptr, nullptr);
		}
	}
	const long long total = dp1 > 128 ? (long long)nq : (long long)nqb * CL_QBLOCK;
	hipLaunchKernelGGL(collect_bound_table_multi_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, st, (const unsigned *)d_gslot,
	                   (long long)nq * 128, nranges, d_e2, kfp, (int)nq, total, d_pbnd, dp1 > 128 ? 1 : 0);
	MVS_HIP(hipGetLastError());
}

void launch_collect_final_thr(const unsigned *d_gslot, int d, int kk, const float *d_e2, int64_t nq, float *d_thr, hipStream_t st) {
	if (nq <= 0)
		return;
	const int stride = collect_slot_stride(kk, collect_store_dims(d));
	const dim3 grid((unsigned)((nq + 63) / 64));
	if (stride == 128)
		hipLaunchKernelGGL(collect_final_thr_kernel<128>, grid, dim3(64), 0, st, d_gslot, d_e2, kk, (long long)nq, d_thr);
	else if (stride == 32)
		hipLaunchKernelGGL(collect_final_thr_kernel<32>, grid, dim3(64), 0, st, d_gslot, d_e2, kk, (long long)nq, d_thr);
	else
		hipLaunchKernelGGL(collect_final_thr_kernel<16>, grid, dim3(64), 0, st, d_gslot, d_e2, kk, (long long)nq, d_thr);
	MVS_HIP(hipGetLastError());
}

// the words the host reads after a search -- the scan's entry count and (bucketed finish) the control block's header, the fail count,
// the largest rounding residual -- written into PINNED HOST memory by one tiny kernel instead of four 4 .. 256-byte copies in a row
// (each a launch of its own on the stream: 19 us of a 2.7 ms shard step)
__global__ void collect_report_kernel(const unsigned long long *__restrict__ hdr, const int *__restrict__ fail_cnt,
                                      const unsigned *__restrict__ maxnorm, int *__restrict__ h_flags,
                                      unsigned long long *__restrict__ h_hdr, int with_cnt) {
	const int t = threadIdx.x;
	if (h_hdr && hdr && t < 32)
		h_hdr[t] = hdr[t];
	if (t == 0) {
		h_flags[8] = *fail_cnt;
		h_flags[9] = (int)*maxnorm;
		if (with_cnt && hdr) {
			const unsigned long long c = hdr[0];
			h_flags[10] = (int)(unsigned)c;
			h_flags[11] = (int)(unsigned)(c >> 32);
		}
	}
}
void launch_collect_report(const void *d_hdr, const int *d_fail_cnt, const unsigned *d_maxnorm, int *h_flags, void *h_hdr, bool with_cnt,
                           hipStream_t st) {
	hipLaunchKernelGGL(collect_report_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long *)d_hdr, d_fail_cnt, d_maxnorm, h_flags,
	                   (unsigned long long *)h_hdr, with_cnt ? 1 : 0);
	MVS_HIP(hipGetLastError());
}

// slots -> neutral, stream counter -> 0, then the bound-estimation pre-pass over the first rows
void launch_collect_prepare(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms,
                            int64_t n, int64_t nq, int kk, const float *d_e2, unsigned *d_gslot,
                            unsigned long long *d_stream_cnt, const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st,
                            bool cnt_zeroed, bool slots_ready, float *d_seed_stage) {
	const int stride = collect_slot_stride(kk, collect_store_dims(g.d)); // 16 row classes whatever kk <= 16 is: the bound is the kk-th best of them
	const long long gtotal = (long long)nq * stride;
	if (!slots_ready) // (launch_collect_query_prep set them neutral)
		hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gslot, gtotal, stride, stride,
		                   0 /* larger s is better */);
	if (!cnt_zeroed)
		MVS_HIP(hipMemsetAsync(d_stream_cnt, 0, 16, st));
	CollectArgs a;
	memset(&a, 0, sizeof a);
	a.qf = d_qf;
	a.yb = d_rows;
	a.yn = d_norms;
	a.e2 = d_e2;
	a.gslot = d_gslot;
	a.slot_stride = stride;
	a.nclass = kk;
	a.nq = (int)nq;
	a.rowmask = d_rowmask;
	a.opt = tune().ksplit_opt;
	a.pbnd = d_pbnd;
	// (a fixed cost per search: scaled down with the database so that a row shard of a multi-GPU index does not pay 16k rows)
	const int dp1 = collect_store_dims(g.d);
	if (dp1 == 128 && tune().cl_seed_regs && stride == 16) { // (32 classes: 64 registers of maxima do not fit; the publish-only scan below)
		// d <= 128: class maxima in registers (flat_bf16_seed_kernel) -- cheap enough for 32 768 rows (an eighth of a small index)
		const int64_t rows = std::min<int64_t>(tune().cl_seed_rows > 16384 ? tune().cl_seed_rows : std::max(1024, tune().cl_seed_reg_rows), n / 8) / 64 * 64;
		if (rows >= 1024) {
			a.seed_stage = d_seed_stage;
			launch_collect_seed(metric, a, rows, nq, st);
		}
		return;
	}
	// (lists beyond 16: the bound is the kk-th best of the class bests, so the sample must grow with kk or the main scan starts with
	// a bound that admits whole percents of the rows -- kk = 65 on 2 048 seed rows of a 150 000-row index: > 4 096 candidates per query)
	const int64_t kscale = std::max(1, kk / 16);
	const int64_t seed = std::min<int64_t>(n, kscale * std::min<int64_t>(tune().cl_seed_rows, std::max<int64_t>(2048, n / 256)));
	if (seed > 0 && seed < n) {
		if (dp1 > 128)
			launch_collect_wide_range(dp1, metric, false, a, 0, seed, tune().cl_seed_split > 0 ? tune().cl_seed_split : 32, nq, st, nullptr, nullptr);
		else
			launch_collect_range<false>(g, metric, a, 0, seed, tune().cl_seed_split > 0 ? tune().cl_seed_split : 32, nq, st, nullptr, nullptr);
	}
}

// the main scan: every row, candidates into the stream
void launch_collect_scan(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms,
                         int64_t n, int64_t nq, int kk, const float *d_e2, unsigned *d_gslot, unsigned long long *d_stream,
                         unsigned long long *d_stream_cnt, int64_t stream_cap, const unsigned long long *d_rowmask, float *d_pbnd,
                         hipStream_t st, int *grid_out, int *nsplit_out, int *lds_out, float *d_stream_s, bool frozen) {
	CollectArgs a;
	memset(&a, 0, sizeof a);
	a.qf = d_qf;
	a.yb = d_rows;
	a.yn = d_norms;
	a.e2 = d_e2;
	a.gslot = d_gslot;
	a.slot_stride = frozen ? 16 : collect_slot_stride(kk, collect_store_dims(g.d)); // (frozen bounds: the 16-class instance, its slots unused)
	a.nclass = frozen ? 16 : kk;
	a.nq = (int)nq;
	a.stream = d_stream;
	a.stream_s = d_stream_s;
	a.stream_cnt = d_stream_cnt;
	a.stream_cap = stream_cap;
	a.rowmask = d_rowmask;
	a.opt = tune().ksplit_opt | (frozen ? 256 : 0);
	a.pbnd = d_pbnd;
	const int dp1 = collect_store_dims(g.d);
	const int qblock = dp1 > 128 ? collect_wide_qblock(dp1) : CL_QBLOCK;
	const int nqb = (int)((nq + qblock - 1) / qblock);
	// two workgroups per CU: 512 slots; whole rounds, splits a multiple of 8 (XCD mapping), >= 7680 rows per split (8192 kept C2's
	// N = 1 M at 120 splits = 4.7 rounds of workgroups; 128 splits of 7 812 rows fill five: 3.0-3.17 -> 2.80-2.86 ms per batch)
	const int64_t slots = dp1 > 128 ? collect_wide_slots(dp1) : 512; // resident workgroups
	int64_t nsplit = tune().cl_nsplit;
	if (nsplit <= 0) {
		const int64_t max_split = std::max<int64_t>(1, n / 7680);
		nsplit = 1;
		double best = -1;
		for (int64_t s = 8; s <= std::min<int64_t>(max_split, 512); s += 8) {
			const int64_t w = s * nqb, rounds = (w + slots - 1) / slots;
			double eff = (double)w / (double)(rounds * slots);
			if (rounds < 2)
				eff -= 0.05;
			eff += 1e-5 * (double)std::min<int64_t>(rounds, 10); // at equal fill: more, shorter rounds balance better (17.6 vs 17.9 ms)
			if (eff > best) {
				best = eff;
				nsplit = s;
			}
		}
		if (max_split < 8)
			nsplit = max_split;
	}
	if (dp1 > 128)
		launch_collect_wide_range(dp1, metric, true, a, 0, n, nsplit, nq, st, grid_out, nsplit_out);
	else
		launch_collect_range<true>(g, metric, a, 0, n, nsplit, nq, st, grid_out, nsplit_out);
	if (lds_out)
		*lds_out = (int)(dp1 > 128 ? collect_wide_lds_bytes(dp1) : collect_lds_bytes(g));
}

// ---- stream overflow: the queries that hold more than their share --------------------------------------------------------------
__global__ void collect_count_queries_kernel(const unsigned long long *__restrict__ stream, long long n, int *__restrict__ qcount) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		atomicAdd(&qcount[(unsigned)(stream[i] >> 32)], 1);
}
__global__ void collect_drop_heavy_kernel(const int *__restrict__ qcount, long long nq, int share, float *__restrict__ e2,
                                          int *__restrict__ fail_cnt, int *__restrict__ fail_q, int *__restrict__ nheavy) {
	const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nq || qcount[q] <= share || e2[q] != e2[q]) // (NaN: already out)
		return;
	e2[q] = __uint_as_float(0x7fc00000u); // NaN: nothing of this query passes any more
	fail_q[atomicAdd(fail_cnt, 1)] = (int)q;
	atomicAdd(nheavy, 1);
}
// counts the (truncated) stream's entries per query, takes the queries above `share` out of the coarse filter; returns how many
int launch_collect_drop_heavy(const unsigned long long *d_stream, int64_t n, int64_t nq, int share, int *d_qcount /* [nq + 16], zeroed */,
                              float *d_e2, int *d_fail_cnt, int *d_fail_q, hipStream_t st) {
	hipLaunchKernelGGL(collect_count_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_stream, (long long)n, d_qcount);
	hipLaunchKernelGGL(collect_drop_heavy_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (const int *)d_qcount,
	                   (long long)nq, share, d_e2, d_fail_cnt, d_fail_q, d_qcount + nq);
	MVS_HIP(hipGetLastError());
	int nheavy = 0;
	MVS_HIP(hipMemcpyAsync(&nheavy, d_qcount + nq, sizeof(int), hipMemcpyDeviceToHost, st));
	MVS_HIP(hipStreamSynchronize(st));
	return nheavy;
}

// ---- candidates -> exact values ----------------------------------------------------------------------------------------
// segment of every query in the stream sorted by query
__global__ void collect_segments_kernel(const unsigned long long *__restrict__ sorted, long long n, int *__restrict__ seg_b,
                                        int *__restrict__ seg_e, unsigned nq) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const unsigned q = (unsigned)(sorted[i] >> 32);
	if (q >= nq) // (the sentinel tail of launch_collect_group_est)
		return;
	if (i == 0 || (unsigned)(sorted[i - 1] >> 32) != q)
		seg_b[q] = (int)i;
	if (i == n - 1 || (unsigned)(sorted[i + 1] >> 32) != q)
		seg_e[q] = (int)(i + 1);
}

// Thread <-> candidate: the oracle's value of the row (ip = fmaf chain in k order over the ORIGINAL f32 row; L2:
// max(0, (xn + yn) - 2 ip)) replaces the query number in the entry: (order-preserving value key << 32) | row.  FAISS
// inserts a value only if it beats the neutral element (strict compare; NaN never): such entries become EMPTY.
// One wave per 64 candidates.  The rows are staged through LDS with coalesced loads (half a wave per 512-byte row; a
// thread streaming its own row thrashes the 32 KB L1: 6.2 ms for 10^7 candidates), row pitch DP + 4 floats so that the 16
// lanes of a ds_read_b128 phase hit distinct banks; then lane <-> candidate runs the k-ordered chain.
// PAIR (L2 with an IDSelector: FAISS's per-pair branch, exhaustive_L2sqr_seq): t = x_k - y_k, acc = fmaf(t, t, acc), k ascending
// cnt != null: the number of candidates is min(*cnt, ncand) -- read on the device, the host never waited for it -- and the grid is
// a fixed number of waves that walk the groups of 64 in strides (round 4: no host round trip between the scan and the re-scoring).
template <bool IS_L2, int DP, bool PAIR = false>
__global__ __launch_bounds__(64) void collect_exact_kernel(unsigned long long *__restrict__ sorted, long long ncand,
                                                          const float *__restrict__ x, int d,
                                                          const float *__restrict__ vecs, int interleaved,
                                                          const float *__restrict__ norms, const float *__restrict__ qn,
                                                          const unsigned long long *__restrict__ cnt) {
	constexpr int PITCH = DP + 4, CPR = DP / 4; // floats per LDS row, float4 chunks per row
	constexpr int RPI = 64 / CPR;               // rows per load instruction (2 at DP = 128)
	__shared__ __attribute__((aligned(16))) float rows[64 * PITCH];
	const int lane = threadIdx.x;
	if (cnt) {
		const unsigned long long have = *cnt;
		ncand = have < (unsigned long long)ncand ? (long long)have : ncand;
	}
	for (long long i0 = (long long)blockIdx.x * 64; i0 < ncand; i0 += (long long)gridDim.x * 64) {
	const long long i = i0 + lane;
	const unsigned long long ent = i < ncand ? sorted[i] : 0ull;
	const unsigned row = (unsigned)ent;
	const long long q = (long long)(ent >> 32);
	const int sub = lane / CPR, ch = lane % CPR;
	const float *xq = x + q * d;
	// d = DP (the headline's 128): the lane's WHOLE query in registers, every load issued before the rows are staged -- one memory
	// round trip instead of one per four dimensions in front of the fma chain.  (global 16-byte loads need 4-byte alignment only)
	const bool whole = d == DP;
	float4 xr[CPR];
	if (whole) {
#pragma unroll
		for (int c4 = 0; c4 < CPR; ++c4)
			xr[c4] = *(const float4 *)(xq + c4 * 4);
	}
#pragma unroll 16
	for (int r = 0; r < 64; r += RPI) {
		// (v_readlane + select instead of this ds_bpermute, fully unrolled, measured 0.40 instead of 0.14 ms: the loads then wait on
		// scalar moves; the permutes of 16 iterations pipeline)
		const unsigned rr = (unsigned)__shfl((int)row, r + sub);
		const float4 v = *(const float4 *)(vecs + (size_t)rr * DP + ch * 4);
		*(float4 *)(rows + (r + sub) * PITCH + ch * 4) = v;
	}
	__syncthreads();
	if (i < ncand) {
	const float *y = rows + lane * PITCH;
	const bool odd = interleaved && ((row >> 4) & 1);
	float ip = 0.f;
	if (whole) {
#pragma unroll
		for (int c4 = 0; c4 < CPR; ++c4) {
			const float4 s = *(const float4 *)(y + c4 * 4);
			float v0, v1, v2, v3;
			if (!interleaved)
				v0 = s.x, v1 = s.y, v2 = s.z, v3 = s.w;
			else if (odd)
				v0 = s.z, v1 = s.x, v2 = s.w, v3 = s.y;
			else
				v0 = s.x, v1 = s.z, v2 = s.y, v3 = s.w;
			const float4 xv = xr[c4];
			if (PAIR) {
				float t = __fsub_rn(xv.x, v0);
				ip = fmaf(t, t, ip);
				t = __fsub_rn(xv.y, v1);
				ip = fmaf(t, t, ip);
				t = __fsub_rn(xv.z, v2);
				ip = fmaf(t, t, ip);
				t = __fsub_rn(xv.w, v3);
				ip = fmaf(t, t, ip);
			} else {
				ip = fmaf(xv.x, v0, ip);
				ip = fmaf(xv.y, v1, ip);
				ip = fmaf(xv.z, v2, ip);
				ip = fmaf(xv.w, v3, ip);
			}
		}
	}
	for (int g4 = whole ? d : 0; g4 < d; g4 += 4) {
		const float4 s = *(const float4 *)(y + g4);
		float v0, v1, v2, v3;
		if (!interleaved)
			v0 = s.x, v1 = s.y, v2 = s.z, v3 = s.w;
		else if (odd)
			v0 = s.z, v1 = s.x, v2 = s.w, v3 = s.y;
		else
			v0 = s.x, v1 = s.z, v2 = s.y, v3 = s.w;
		if (PAIR) {
			float t = __fsub_rn(xq[g4], v0);
			ip = fmaf(t, t, ip);
			if (g4 + 1 < d) {
				t = __fsub_rn(xq[g4 + 1], v1);
				ip = fmaf(t, t, ip);
			}
			if (g4 + 2 < d) {
				t = __fsub_rn(xq[g4 + 2], v2);
				ip = fmaf(t, t, ip);
			}
			if (g4 + 3 < d) {
				t = __fsub_rn(xq[g4 + 3], v3);
				ip = fmaf(t, t, ip);
			}
			continue;
		}
		ip = fmaf(xq[g4], v0, ip);
		if (g4 + 1 < d)
			ip = fmaf(xq[g4 + 1], v1, ip);
		if (g4 + 2 < d)
			ip = fmaf(xq[g4 + 2], v2, ip);
		if (g4 + 3 < d)
			ip = fmaf(xq[g4 + 3], v3, ip);
	}
	float ex;
	bool ok;
	if (PAIR) {
		ex = ip;
		ok = ex < FLT_MAX;
	} else if (IS_L2) {
		ex = fmaf(-2.0f, ip, qn[q] + norms[row]);
		ex = ex < 0.f ? 0.f : ex; // FAISS: if (dis < 0) dis = 0
		ok = ex < FLT_MAX;
	} else {
		ex = ip;
		ok = ex > -FLT_MAX;
	}
	sorted[i] = ok ? (((unsigned long long)bkey<IS_L2>(ex) << 32) | row) : ~0ull;
	}
	__syncthreads(); // (the next group's rows overwrite the tile)
	}
}

__device__ __forceinline__ unsigned long long cl_lane64(unsigned long long v, int l) { // l uniform
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
	return ((unsigned long long)hi << 32) | lo;
}
// One wave per query: the kk <= 128 best keys (value, row) of its segment, kept as a sorted list spread over the lanes (entry i in
// lane i & 63 of register i >> 6): insert position by ballot, shift by one lane (the second register takes the first one's lane 63).
template <bool IS_L2>
__global__ __launch_bounds__(64) void collect_select_kernel(const unsigned long long *__restrict__ keys,
                                                           const int *__restrict__ seg_b, const int *__restrict__ seg_e,
                                                           int kk, float *__restrict__ pd1, int *__restrict__ pi1) {
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const int b = seg_b[q], e = seg_e[q];
	const unsigned long long EMPTY = ~0ull;
	// Fast path (round 4; <= 1 024 candidates, lists of < 64 -- the usual case: 90-150 candidates for k = 10): every key in registers,
	// U = the kk-th smallest of the 64 LANE MINIMA by value (an upper bound of the kk-th smallest value: the minima are distinct
	// entries), the keys with value <= U -- a few times kk -- compacted into LDS and ranked against each other there.  The serial
	// insertion below costs ~25 instructions per candidate that beats the running worst, ~kk ln(n / kk) + kk of them per query.
	// Longer segments (clustered data: a few queries hold thousands of candidates and set the kernel's duration): the first 1 024
	// entries go through the fast path, its result seeds the sorted list and the running worst of the serial loop, which then
	// inserts only what beats a bound that is already tight.
	int start = b;
	unsigned long long seed_mine = EMPTY, seed_worst = EMPTY;
	if (kk < 64) {
		constexpr int R = 16;
		__shared__ unsigned long long surv[256];
		__shared__ unsigned long long top[64];
		const int ce = e - b <= 1024 ? e : b + 1024; // end of the chunk taken here
		const int n = ce - b, nr = (n + 63) >> 6;
		unsigned long long kreg[R];
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const int i = b + 64 * r + lane;
			kreg[r] = (r < nr && i < ce) ? keys[i] : EMPTY;
		}
		unsigned U = 0xffffffffu;
		if (n > 64) {
			unsigned long long lmin = kreg[0];
#pragma unroll
			for (int r = 1; r < R; ++r)
				lmin = kreg[r] < lmin ? kreg[r] : lmin;
			const unsigned hi = (unsigned)(lmin >> 32); // (EMPTY: 0xffffffff)
			U = 0u;
#pragma unroll 1
			for (int bit = 31; bit >= 0; --bit) {
				const unsigned t = U | (1u << bit);
				if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(hi < t)) < kk)
					U = t;
			}
		}
		// (values <= U: a few times kk unless many candidates share the boundary value; past 256 of them: the serial path)
		int total = 0;
#pragma unroll
		for (int r = 0; r < R; ++r)
			total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kreg[r] != EMPTY && (unsigned)(kreg[r] >> 32) <= U));
		if (total <= 256) {
			int S = 0;
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const bool take = kreg[r] != EMPTY && (unsigned)(kreg[r] >> 32) <= U;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
				if (take)
					surv[S + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = kreg[r];
				S += __builtin_popcountll(m);
			}
			if (lane < 64)
				top[lane] = EMPTY;
			__syncthreads();
			for (int p = lane; p < S; p += 64) {
				const unsigned long long me = surv[p];
				int rank = 0;
				for (int j = 0; j < S; ++j) { // (equal keys cannot occur -- a row is scanned once per query --; ranked by position if they did)
					const unsigned long long o = surv[j];
					rank += (o < me || (o == me && j < p)) ? 1 : 0;
				}
				if (rank < kk)
					top[rank] = me;
			}
			__syncthreads();
			if (ce == e) { // the whole segment: done
				if (lane < kk) {
					const unsigned long long me = top[lane];
					const bool have = me != EMPTY;
					pd1[q * kk + lane] = have ? bkey2f<IS_L2>((unsigned)(me >> 32)) : (IS_L2 ? FLT_MAX : -FLT_MAX);
					pi1[q * kk + lane] = have ? (int)(unsigned)me : -1;
				}
				return;
			}
			seed_mine = lane < kk ? top[lane] : EMPTY;
			seed_worst = top[kk - 1];
			start = ce;
		}
	}
	unsigned long long mine = seed_mine, mine2 = EMPTY; // entries `lane` and `64 + lane` of the sorted list
	unsigned long long worst = seed_worst;
	auto shr1 = [](unsigned long long v) { // lane i <- lane i - 1 (lane 0: 0)
		return ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false) << 32) |
		       (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
	};
	// (eight loads in flight per round trip: a segment of thousands of entries walked 64 at a time was one dependent global load
	// per iteration -- the few heavy queries of a clustered batch set the kernel's duration)
	for (int base8 = start; base8 < e; base8 += 512) {
	unsigned long long k8[8];
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const int i = base8 + 64 * r + lane;
		k8[r] = i < e ? keys[i] : EMPTY;
	}
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const unsigned long long key = k8[r];
		unsigned long long pend = __builtin_amdgcn_ballot_w64(key < worst);
		while (pend != 0ull) {
			const int L = __builtin_ctzll(pend);
			pend &= pend - 1ull;
			// (v_readlane / DPP wave_shr:1 instead of ds_bpermute: no LDS round trip per inserted candidate -- csrc/hnsw.hip, "sorted
			// lists in REGISTERS")
			const unsigned long long ck = cl_lane64(key, L);
			if (ck >= worst)
				continue;
			const int pos = __popcll(__builtin_amdgcn_ballot_w64(lane < kk && mine <= ck)) +
			                (kk > 64 ? __popcll(__builtin_amdgcn_ballot_w64(64 + lane < kk && mine2 <= ck)) : 0);
			if (kk > 64) {
				const unsigned long long carry = cl_lane64(mine, 63);
				const unsigned long long sh2 = shr1(mine2);
				const unsigned long long up2 = lane == 0 ? carry : sh2;
				if (64 + lane == pos)
					mine2 = ck;
				else if (64 + lane > pos && 64 + lane < kk)
					mine2 = up2;
			}
			const unsigned long long up = shr1(mine);
			if (lane == pos)
				mine = ck;
			else if (lane > pos && lane < kk)
				mine = up;
			worst = kk > 64 ? cl_lane64(mine2, kk - 65) : cl_lane64(mine, kk - 1);
		}
	}
	}
	if (lane < kk) {
		const bool have = mine != EMPTY;
		pd1[q * kk + lane] = have ? bkey2f<IS_L2>((unsigned)(mine >> 32)) : (IS_L2 ? FLT_MAX : -FLT_MAX);
		pi1[q * kk + lane] = have ? (int)(unsigned)mine : -1;
	}
	if (64 + lane < kk) {
		const bool have = mine2 != EMPTY;
		pd1[q * kk + 64 + lane] = have ? bkey2f<IS_L2>((unsigned)(mine2 >> 32)) : (IS_L2 ? FLT_MAX : -FLT_MAX);
		pi1[q * kk + 64 + lane] = have ? (int)(unsigned)mine2 : -1;
	}
}

// ---- candidates grouped by query WITHOUT the host knowing how many there are (round 4) ------------------------------------------
// Round 3 read the stream's fill back to the host behind the scan (a stream synchronisation in the middle of every search: the GPU
// idles for the round trip, then for the launch latency of each of the ten small kernels behind it) because rocPRIM's radix sort
// and the re-scoring grid want the count as a host value.  Now the host passes an ESTIMATE n_est (what the index's previous search
// produced per query, + 30 %): the tail [min(*cnt, n_est), n_est) is filled with sentinel keys that sort behind every query, the
// sort runs over n_est entries, the kernels behind it read min(*cnt, cap) on the device.  If the estimate was too small the
// caller finds out from the count after ITS synchronisation and runs the search again the synchronous way.
// (A counting sort by query -- histogram, prefix sums, scatter, all on the device count -- needs no estimate but took 120 us for
// 1.4 M candidates against the radix sort's 65: one global atomic per entry on 10 000 counters, twice.)
__global__ void collect_fill_tail_kernel(unsigned long long *__restrict__ stream, const unsigned long long *__restrict__ cnt,
                                         long long n_est) {
	const unsigned long long have = *cnt;
	const long long b = have < (unsigned long long)n_est ? (long long)have : n_est;
	for (long long i = b + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_est; i += (long long)gridDim.x * blockDim.x)
		stream[i] = ~0ull;
}
static int collect_qbits(int64_t nq) { // sort key: the query number and one bit more, so that the all-ones sentinel is no query
	int qbits = 1;
	while (((int64_t)1 << qbits) < nq)
		++qbits;
	return qbits + 1;
}
size_t collect_sort_temp_bytes_est(int64_t n_est, int64_t nq) {
	size_t bytes = 0;
	MVS_HIP(rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n_est, 32,
	                                 32 + collect_qbits(nq), (hipStream_t) nullptr));
	return bytes;
}
void launch_collect_group_est(unsigned long long *d_stream, unsigned long long *d_sorted, const unsigned long long *d_cnt,
                              int64_t n_est, void *d_temp, size_t temp_bytes, int64_t nq, int *d_seg, hipStream_t st, bool seg_zeroed) {
	if (nq <= 0)
		return;
	if (!seg_zeroed)
		MVS_HIP(hipMemsetAsync(d_seg, 0, (size_t)2 * nq * sizeof(int), st));
	if (n_est <= 0)
		return;
	hipLaunchKernelGGL(collect_fill_tail_kernel, dim3((unsigned)std::min<int64_t>((n_est + 255) / 256, 1024)), dim3(256), 0, st, d_stream,
	                   d_cnt, (long long)n_est);
	MVS_HIP(rocprim::radix_sort_keys(d_temp, temp_bytes, d_stream, d_sorted, (size_t)n_est, 32, 32 + collect_qbits(nq), st));
	hipLaunchKernelGGL(collect_segments_kernel, dim3((unsigned)((n_est + 255) / 256)), dim3(256), 0, st, d_sorted, (long long)n_est,
	                   d_seg, d_seg + nq, (unsigned)nq);
	MVS_HIP(hipGetLastError());
}

size_t collect_sort_temp_bytes(int64_t ncand, int64_t nq) {
	size_t bytes = 0;
	int qbits = 1;
	while (((int64_t)1 << qbits) < nq)
		++qbits;
	MVS_HIP(rocprim::radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
	                                 (size_t)ncand, 32, 32 + qbits, (hipStream_t) nullptr));
	return bytes;
}

// stream (ncand entries of q << 32 | row) -> sorted by query + the segment of every query (d_seg: [2 nq] begin | end)
void launch_collect_group(unsigned long long *d_stream, unsigned long long *d_sorted, int64_t ncand, void *d_temp,
                          size_t temp_bytes, int64_t nq, int *d_seg, hipStream_t st, bool seg_zeroed) {
	if (nq <= 0)
		return;
	int qbits = 1;
	while (((int64_t)1 << qbits) < nq)
		++qbits;
	if (!seg_zeroed)
		MVS_HIP(hipMemsetAsync(d_seg, 0, (size_t)2 * nq * sizeof(int), st));
	if (ncand > 0) {
		MVS_HIP(rocprim::radix_sort_keys(d_temp, temp_bytes, d_stream, d_sorted, (size_t)ncand, 32, 32 + qbits, st));
		hipLaunchKernelGGL(collect_segments_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, st, d_sorted,
		                   (long long)ncand, d_seg, d_seg + nq, (unsigned)nq);
		MVS_HIP(hipGetLastError());
	}
}
// keys (order-preserving value key << 32 | id) of every query's segment -> the kk best: pd1 / pi1 [nq][kk], best first
void launch_collect_select(int metric, const unsigned long long *d_keys, const int *d_seg, int64_t nq, int kk, float *d_pd1,
                           int32_t *d_pi1, hipStream_t st) {
	if (nq <= 0)
		return;
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(collect_select_kernel<true>, dim3((unsigned)nq), dim3(64), 0, st, d_keys, d_seg, d_seg + nq, kk, d_pd1,
		                   d_pi1);
	else
		hipLaunchKernelGGL(collect_select_kernel<false>, dim3((unsigned)nq), dim3(64), 0, st, d_keys, d_seg, d_seg + nq, kk, d_pd1,
		                   d_pi1);
	MVS_HIP(hipGetLastError());
}

// Inner-product tie pass from the candidate list (instead of another pass over the database): for flagged query f with boundary
// score T the rows with exact score >= T all are candidates (they are at least as good as the kk-th best), so A_k -- the k smallest
// row ids among them, ascending; what FlatIndex::tie_candidates computes with the TIE epilogue -- is read off the query's
// segment of the re-scored list.  One wave per flagged query, k rounds of "smallest row id above the previous one".
// (pitch > 0, round 6: the bucketed finish -- the query's exact keys are bucket[q][pitch], seg_b = the per-query counts)
__global__ __launch_bounds__(64) void collect_tie_rows_kernel(const unsigned long long *__restrict__ sorted, const int *__restrict__ seg_b,
                                                             const int *__restrict__ seg_e, const int *__restrict__ fq,
                                                             const float *__restrict__ T, int k, long long *__restrict__ first, int pitch) {
	const int f = blockIdx.x, lane = threadIdx.x;
	const int q = fq[f];
	const unsigned tk = bkey<false>(T[f]); // smaller key = larger score
	long long b = seg_b[q], e = pitch > 0 ? 0 : seg_e[q];
	if (pitch > 0) {
		const long long cnt = (unsigned)seg_b[q] < (unsigned)pitch ? (long long)(unsigned)seg_b[q] : (long long)pitch;
		b = (long long)q * pitch;
		e = b + cnt;
	}
	long long last = -1;
	for (int j = 0; j < k; ++j) {
		unsigned best = 0xffffffffu;
		if (last != -2) {
			for (long long i = b + lane; i < e; i += 64) {
				const unsigned long long ent = sorted[i];
				const unsigned row = (unsigned)ent;
				if (ent != ~0ull && (unsigned)(ent >> 32) <= tk && (long long)row > last && row < best)
					best = row;
			}
			for (int o = 32; o >= 1; o >>= 1) {
				const unsigned other = (unsigned)__shfl_xor((int)best, o);
				best = other < best ? other : best;
			}
		}
		if (lane == 0)
			first[(size_t)f * k + j] = best == 0xffffffffu ? -1ll : (long long)best;
		last = best == 0xffffffffu ? -2 : (long long)best; // exhausted: the remaining slots are -1
	}
}
void launch_collect_tie_rows(const unsigned long long *d_sorted, const int *d_seg, int64_t nq, const int *d_flag_query,
                             const float *d_T, int nf, int k, int64_t *d_first, hipStream_t st) {
	if (nf <= 0)
		return;
	hipLaunchKernelGGL(collect_tie_rows_kernel, dim3((unsigned)nf), dim3(64), 0, st, d_sorted, d_seg, d_seg + nq, d_flag_query, d_T,
	                   k, (long long *)d_first, 0);
	MVS_HIP(hipGetLastError());
}
void launch_collect_tie_rows_bucket(const unsigned long long *d_bucket, const unsigned *d_bcount, int pitch, const int *d_flag_query,
                                    const float *d_T, int nf, int k, int64_t *d_first, hipStream_t st) {
	if (nf <= 0)
		return;
	hipLaunchKernelGGL(collect_tie_rows_kernel, dim3((unsigned)nf), dim3(64), 0, st, d_bucket, (const int *)d_bcount, (const int *)nullptr,
	                   d_flag_query, d_T, k, (long long *)d_first, pitch);
	MVS_HIP(hipGetLastError());
}

// ---- selection for lists beyond 128 entries: one segmented radix sort of the exact keys, then the first kk of every segment
template <bool IS_L2>
__global__ void collect_take_sorted_kernel(const unsigned long long *__restrict__ sorted, const int *__restrict__ seg_b,
                                           const int *__restrict__ seg_e, int kk, float *__restrict__ pd1, int *__restrict__ pi1) {
	const long long q = blockIdx.x;
	const int b = seg_b[q], n = seg_e[q] - b;
	for (int j = threadIdx.x; j < kk; j += blockDim.x) {
		const bool have = j < n;
		const unsigned long long key = have ? sorted[(size_t)b + j] : 0ull;
		pd1[q * kk + j] = have ? bkey2f<IS_L2>((unsigned)(key >> 32)) : (IS_L2 ? FLT_MAX : -FLT_MAX);
		pi1[q * kk + j] = have ? (int)(unsigned)key : -1;
	}
}
size_t collect_select_big_temp_bytes(int64_t ncand, int64_t nq) {
	size_t bytes = 0;
	MVS_HIP(rocprim::segmented_radix_sort_keys(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (unsigned)ncand,
	                                           (unsigned)nq, (const int *)nullptr, (const int *)nullptr, 0, 64, (hipStream_t) nullptr));
	return bytes + 256;
}
void launch_collect_select_big(int metric, unsigned long long *d_keys, unsigned long long *d_out, int64_t ncand, const int *d_seg, int64_t nq,
                               int kk, void *d_temp, size_t temp_bytes, float *d_pd1, int32_t *d_pi1, hipStream_t st) {
	if (nq <= 0)
		return;
	const unsigned long long *res = d_keys;
	if (ncand > 0) {
		size_t need = 0;
		MVS_HIP(rocprim::segmented_radix_sort_keys(nullptr, need, d_keys, d_out, (unsigned)ncand, (unsigned)nq, d_seg, d_seg + nq, 0, 64, st));
		if (need > temp_bytes)
			throw_faiss("mvs::launch_collect_select_big", __FILE__, "sort workspace of %zu bytes, %zu needed", temp_bytes, need);
		MVS_HIP(rocprim::segmented_radix_sort_keys(d_temp, need, d_keys, d_out, (unsigned)ncand, (unsigned)nq, d_seg, d_seg + nq, 0, 64, st));
		res = d_out;
	}
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(collect_take_sorted_kernel<true>, dim3((unsigned)nq), dim3(256), 0, st, res, d_seg, d_seg + nq, kk, d_pd1, d_pi1);
	else
		hipLaunchKernelGGL(collect_take_sorted_kernel<false>, dim3((unsigned)nq), dim3(256), 0, st, res, d_seg, d_seg + nq, kk, d_pd1, d_pi1);
	MVS_HIP(hipGetLastError());
}

// stream (ncand entries) -> per query the kk best exact candidates: pd1 / pi1 [nq][kk] (value, row), best first
// d_cnt != null: device-count mode -- ncand is the host's ESTIMATE of the number of entries (<= the stream's capacity; the sort
// covers that many, sentinels behind the real ones), the real number is min(*d_cnt, ncand) on the device
void launch_collect_rescore(int metric, unsigned long long *d_stream, unsigned long long *d_sorted, int64_t ncand, void *d_temp,
                            size_t temp_bytes, int64_t nq, int kk, const float *d_x, const FlatGeom &g, const float *d_vecs,
                            const float *d_norms, const float *d_qn, int *d_seg, float *d_pd1, int32_t *d_pi1,
                            bool per_pair, hipStream_t st, const unsigned long long *d_cnt, bool seg_zeroed) {
	if (nq <= 0)
		return;
	if (d_cnt)
		launch_collect_group_est(d_stream, d_sorted, d_cnt, ncand, d_temp, temp_bytes, nq, d_seg, st, seg_zeroed);
	else
		launch_collect_group(d_stream, d_sorted, ncand, d_temp, temp_bytes, nq, d_seg, st, seg_zeroed);
	if (ncand > 0 && collect_store_dims(g.d) > 128) {
		launch_collect_exact_wide(metric, per_pair, d_sorted, ncand, d_x, g.d, d_vecs, g.dp, g.pair_interleaved ? 1 : 0, d_norms, d_qn, st, d_cnt);
	} else if (ncand > 0) {
		// (device-count mode: 8 192 waves walk the groups in strides -- two dispatch rounds of the 4 096 resident ones)
		const dim3 grid((unsigned)(d_cnt ? std::min<int64_t>((ncand + 63) / 64, 8192) : (ncand + 63) / 64));
		// (row pitch of the f32 store = FlatGeom::dp: 128 for 64 < d <= 128, 64 / 32 below)
#define MVS_CL_EXACT(DPV)                                                                                              \
	{                                                                                                                  \
		if (metric == METRIC_L2 && per_pair)                                                                           \
			hipLaunchKernelGGL((collect_exact_kernel<true, DPV, true>), grid, dim3(64), 0, st, d_sorted, (long long)ncand, d_x, g.d, \
			                   d_vecs, g.pair_interleaved ? 1 : 0, d_norms, d_qn, d_cnt);                              \
		else if (metric == METRIC_L2)                                                                                  \
			hipLaunchKernelGGL((collect_exact_kernel<true, DPV>), grid, dim3(64), 0, st, d_sorted, (long long)ncand, d_x, g.d, \
			                   d_vecs, g.pair_interleaved ? 1 : 0, d_norms, d_qn, d_cnt);                              \
		else                                                                                                           \
			hipLaunchKernelGGL((collect_exact_kernel<false, DPV>), grid, dim3(64), 0, st, d_sorted, (long long)ncand, d_x, g.d, \
			                   d_vecs, g.pair_interleaved ? 1 : 0, d_norms, d_qn, d_cnt);                              \
	}
		if (g.dp == 128)
			MVS_CL_EXACT(128)
		else if (g.dp == 64)
			MVS_CL_EXACT(64)
		else if (g.dp == 32)
			MVS_CL_EXACT(32)
		else
			throw_faiss("mvs::launch_collect_rescore", __FILE__, "no re-scoring instance for a row pitch of %d floats", g.dp);
#undef MVS_CL_EXACT
		MVS_HIP(hipGetLastError());
	}
	if (kk > 128) // (round 6: lists beyond the selection kernel's -- every segment sorted, the first kk taken; d_stream is free by now)
		launch_collect_select_big(metric, d_sorted, d_stream, ncand, d_seg, nq, kk, d_temp, temp_bytes, d_pd1, d_pi1, st);
	else
		launch_collect_select(metric, d_sorted, d_seg, nq, kk, d_pd1, d_pi1, st);
}

} // namespace mvs
