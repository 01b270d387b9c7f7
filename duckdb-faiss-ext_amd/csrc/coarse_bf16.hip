// csrc/coarse_bf16.hip -- IVF coarse quantisation on the bf16 matrix pipe (round 6; VERDICT r5 #1a).
//
// Same place in the path as csrc/coarse_select.hip (IndexIVF::search -> quantizer->search(n, x, nprobe), faiss/IndexIVF.cpp; reference
// call site src/faiss_extension.cpp:631 through IndexIVFFlat) and the same output bit for bit: per query the nprobe nearest centroids in
// FAISS's order (dis ascending, id ascending) with dis = max(0, fmaf(-2, ip, ||x||^2 + ||c||^2)), ip = one k-ordered fma chain
// (exhaustive_L2sqr_blas as csrc/coarse_select.hip and oracle/orc_core.c restate it).
//
// coarse_select.hip writes the whole [nq][nlist] distance matrix with the f32 matrix pipe (10.5 GFLOP at 157 TFLOP/s peak: 122 us at
// C3's 10 000 x 4 096 x 128) and reads it back for the selection (164 MB each way, 48 us).  A coarse quantiser IS a Flat search of a
// small database with k = nprobe, and the Flat index answers those with ONE bf16 product per pair, a proven error bound and exact
// re-scoring of the few rows that pass (csrc/flat_collect.hip).  Its scan kernel is built for N >> 10^5 (class slots shared by
// workgroups through HBM, a candidate stream, a seed pass, ~ 190 us of fixed cost); here all of a query's centroids fit ONE workgroup:
//
//   coarse_bf16_filter_kernel   32 queries x all centroids per workgroup (4 waves, tiles of 16 centroids dealt round-robin).  The operands
//                               are the Flat index's own: query fragments bf16(2 x'), ||x||^2 and 2E(q) from
//                               collect_query_prep_kernel, the centred bf16 store + beta = -||c'||^2 from ensure_h1_rows; the MFMA
//                               chain starts at beta, so s(q, c) is flat_bf16_collect_kernel's coarse value and |s - s_exact| <= E(q)
//                               is that kernel's bound (csrc/flat_collect.hip collect_bounds_kernel; nothing new is modelled).
//                               pass 1 (all tiles): class maxima -- 128 classes per query = (wave, lane group, register, tile parity)
//                               -- and T(q) = the np-th largest of them: np DISTINCT centroids have s >= T, so the np-th best exact
//                               value is no worse than T - E and every centroid of the result (ties at the np-th value included)
//                               has s >= T - 2E.  pass 2 (all tiles): centroids with s >= T - 2E -> the query's candidate list.
//   coarse_bf16_exact_kernel    one wavefront per query: the candidates' exact distances (the k-ordered chain on the f32 centroid rows,
//                               two chains per lane), the np smallest (dis, id) keys (csrc/collect_bucket.h cb_select_wave), printed
//                               in order.  A query whose list overflowed or whose bound is not finite computes ALL nlist exact
//                               distances here (chunks of 448 through the same selection): no host round trip, no other path.
//
// HBM traffic: the queries once, the candidate ids (2 bytes each) -- the matrix never exists.  Bound: the bf16 pipe (2 x 10.5 GFLOP at
// C3) + the exact stage's L2-resident row gathers.
#include "collect_bucket.h"
#include "flat_collect.h"
#include "index.h"

#include <algorithm>
#include <cfloat>
#include <cstring>

namespace mvs {

constexpr int CB16_QB = 32;    // queries per workgroup of the filter
constexpr int CB16_CAP = 512;  // candidate ids per query (more: the exact kernel computes every centroid for that query)
constexpr int CB16_CHUNK = 448; // exact-all: new keys per selection round (CAP - 64 kept)

struct CoarseBf16Args {
	const bf16x8 *qf;         // [(qblk16 * 4 + kb) * 64 + lane]: bf16(2 x'), csrc/flat_collect.hip collect_query_prep_kernel
	const unsigned short *yb; // [nlist (+ pad)][128] bf16 centred centroids
	const float *beta;        // [nlist] -||c'||^2
	const float *e2;          // [nq rounded up to 256] 2E(q); NaN: the bound is not finite
	int nq, nlist, np;
	unsigned short *cand;     // [nq][CB16_CAP]
	int *ccount;              // [nq] candidates of the query; > CB16_CAP: overflowed; -1: no finite bound
};

__global__ __launch_bounds__(256, 2) void coarse_bf16_filter_kernel(const CoarseBf16Args a) {
	__shared__ float cls[CB16_QB][128];
	__shared__ float thr_s[CB16_QB];
	__shared__ int cnt_s[CB16_QB];
	__shared__ __attribute__((aligned(16))) unsigned short cand_s[CB16_QB][CB16_CAP];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int col = lane & 15, rg = lane >> 4;
	const long long q0 = (long long)blockIdx.x * CB16_QB;
	if (tid < CB16_QB)
		cnt_s[tid] = 0;
	// the wave's B operands: 2 blocks of 16 queries x 4 k-blocks (resident: 32 VGPRs)
	bf16x8 bq[2][4];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int kb = 0; kb < 4; ++kb)
			bq[i][kb] = a.qf[((q0 / 16 + i) * 4 + kb) * 64 + lane];
	const int ntile = a.nlist >> 4;          // (nlist % 16 == 0: coarse_bf16_supported)
	const int ntw = (ntile - wave + 3) >> 2; // tiles t = wave + 4 j, j < ntw
	// A fragment of tile t: lane holds row 16 t + (lane & 15), bytes [64 kb + 16 (lane >> 4), + 16) of its 256-byte row.  The store is
	// L2-resident (nlist x 256 bytes) but an L2 round trip is ~ 15 tiles' worth of MFMAs: FOUR tiles of a wave are in flight (v1 had
	// one: 89 us for 1.25 passes, the wave waited ~ 2 600 cycles per tile)
	auto load_tile = [&](int t, bf16x8 (&A)[4], f32x4n &Y) {
		const unsigned short *row = a.yb + ((size_t)(16 * t + col) << 7);
#pragma unroll
		for (int kb = 0; kb < 4; ++kb)
			A[kb] = *(const bf16x8 *)(row + 32 * kb + 8 * rg);
		Y = *(const f32x4n *)(a.beta + 16 * t + 4 * rg); // C rows 4 rg + r of the tile
	};
	auto mfma_tile = [&](const bf16x8 (&A)[4], const f32x4n &Y, f32x4n (&acc)[2]) {
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], bq[i][0], Y, 0, 0, 0);
#pragma unroll
			for (int kb = 1; kb < 4; ++kb)
				acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb], bq[i][kb], acc[i], 0, 0, 0);
		}
	};
	bf16x8 A[4][4];
	f32x4n Y[4];
	// ---- pass 1: class maxima over ALL the wave's tiles; class of (query 16 i + col) = (wave, rg, r, j & 1): 128 classes per query
	float cm[2][4][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int r = 0; r < 4; ++r)
			cm[i][r][0] = cm[i][r][1] = -FLT_MAX;
#pragma unroll
	for (int s = 0; s < 4; ++s)
		if (s < ntw)
			load_tile(wave + 4 * s, A[s], Y[s]);
	for (int j0 = 0; j0 < ntw; j0 += 4) {
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			const int j = j0 + s;
			if (j < ntw) { // (wave-uniform)
				f32x4n acc[2];
				mfma_tile(A[s], Y[s], acc);
				if (j + 4 < ntw)
					load_tile(wave + 4 * (j + 4), A[s], Y[s]);
#pragma unroll
				for (int i = 0; i < 2; ++i)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						cm[i][r][s & 1] = acc[i][r] > cm[i][r][s & 1] ? acc[i][r] : cm[i][r][s & 1]; // (NaN never wins; j & 1 == s & 1)
			}
		}
	}
	// (pass 2 starts with the same four tiles: requested now, they arrive under the threshold search)
#pragma unroll
	for (int s = 0; s < 4; ++s)
		if (s < ntw)
			load_tile(wave + 4 * s, A[s], Y[s]);
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int p = 0; p < 2; ++p)
				cls[16 * i + col][64 * p + 16 * wave + 4 * rg + r] = cm[i][r][p];
	__syncthreads();
	// T(q) = the np-th largest of the 128 class maxima (bitwise search on "smaller is better" keys); thr = T - 2E as the Flat scan forms it
	for (int qi = wave; qi < CB16_QB; qi += 4) {
		const unsigned k0 = skey(cls[qi][lane]), k1 = skey(cls[qi][64 + lane]);
		unsigned U = 0u; // the np-th smallest key: the largest U with #(key < U) < np
#pragma unroll 1
		for (int b = 31; b >= 0; --b) {
			const unsigned t = U | (1u << b);
			if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(k0 < t)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(k1 < t)) < a.np)
				U = t;
		}
		if (lane == 0) {
			const long long q = q0 + qi;
			const float e2 = q < a.nq ? a.e2[q] : __uint_as_float(0x7fc00000u);
			thr_s[qi] = skey2f(U) - e2; // (NaN: nothing passes; fewer than np classes set: -FLT_MAX - e2, everything passes)
		}
	}
	__syncthreads();
	const float th0 = thr_s[col], th1 = thr_s[16 + col];
	// ---- pass 2: every tile again; a centroid with s >= thr joins its query's list
	for (int j0 = 0; j0 < ntw; j0 += 4) {
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			const int j = j0 + s;
			if (j < ntw) {
				const int t = wave + 4 * j;
				f32x4n acc[2];
				mfma_tile(A[s], Y[s], acc);
				if (j + 4 < ntw)
					load_tile(t + 16, A[s], Y[s]);
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					const float th = i ? th1 : th0;
					const float mx = __builtin_fmaxf(__builtin_fmaxf(acc[i][0], acc[i][1]), __builtin_fmaxf(acc[i][2], acc[i][3]));
					if (mx >= th) {
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							if (acc[i][r] >= th) {
								const int p = atomicAdd(&cnt_s[16 * i + col], 1);
								if (p < CB16_CAP)
									cand_s[16 * i + col][p] = (unsigned short)(16 * t + 4 * rg + r);
							}
						}
					}
				}
			}
		}
	}
	__syncthreads();
	// the lists -> global (8 queries per wave; 16-byte stores)
	for (int qi = wave; qi < CB16_QB; qi += 4) {
		const long long q = q0 + qi;
		if (q >= a.nq)
			continue;
		const int n = cnt_s[qi];
		const bool finite = thr_s[qi] == thr_s[qi];
		if (lane == 0)
			a.ccount[q] = finite ? n : -1;
		if (!finite || n > CB16_CAP)
			continue;
		const uint4 *src = (const uint4 *)cand_s[qi];
		uint4 *dst = (uint4 *)(a.cand + (size_t)q * CB16_CAP);
		for (int c8 = lane; c8 * 8 < n; c8 += 64)
			dst[c8] = src[c8];
	}
}

struct CoarseExactArgs {
	const float *x; // [nq][d]
	int d, nq, nlist, np;
	const float *cent; // f32 centroid rows, pitch sdp, FlatGeom::pair_interleaved if `interleaved`
	int sdp, interleaved;
	const float *qn, *cn; // ||x||^2, ||c||^2: k-ordered chains
	const unsigned short *cand;
	const int *ccount;
	float *outD;      // [nq][np]
	long long *outI;  // [nq][np], id + label_offset
	long long label_offset;
	unsigned long long *stats; // [0] queries that computed every centroid (one atomic each: rare)
};

__device__ __forceinline__ unsigned long long cb16_key(float dis, int c) {
	// (dis >= 0 or NaN: the bit pattern orders the finite values; a candidate iff dis < FLT_MAX, the heap's strict compare -- NaN never enters)
	return dis < FLT_MAX ? (((unsigned long long)__float_as_uint(dis) << 32) | (unsigned)c) : CB_EMPTY;
}

// One wavefront per query.  32 candidate rows at a time come in with COALESCED 16-byte loads (a row of DP floats = DP / 4 neighbouring
// lanes; v1 let every lane walk its own row: 64 cache lines per load instruction, 256 us at C3), are de-interleaved and transposed
// through LDS, and lanes 0 .. 31 run the k-ordered chains -- fmaf(x_k, y_k, acc) from k = 0, then fmaf(-2, acc, xn + cn[c]), clamped
// at 0: csrc/coarse_select.hip coarse_dist_kernel's value bit for bit.  The next 32 rows are requested before the chains start.
template <int DP>
__global__ __launch_bounds__(64) void coarse_bf16_exact_kernel(const CoarseExactArgs a) {
	constexpr int CPR = DP / 4, PITCH = DP + 4, NIT = 32 * CPR / 64; // 16-byte chunks per row; floats per LDS row; loads per lane and tile
	__shared__ __attribute__((aligned(16))) float yrows[32 * PITCH];
	__shared__ __attribute__((aligned(16))) float xs[DP];
	__shared__ unsigned long long keys[CB16_CAP];
	__shared__ unsigned long long surv[256];
	__shared__ unsigned long long top[64];
	__shared__ __attribute__((aligned(16))) unsigned short cl[CB16_CAP];
	const int lane = threadIdx.x;
	const long long q = blockIdx.x;
	for (int i = lane; i < DP; i += 64)
		xs[i] = i < a.d ? a.x[q * a.d + i] : 0.f;
	const int n = a.ccount[q];
	const float xn = a.qn[q];
	const bool all = n < 0 || n > CB16_CAP;
	const int nch = (a.d + 3) >> 2;
	// rows [b0, b0 + 32) of the current list (ids from `idof`) -> registers, coalesced; invalid slots repeat a valid row
	f32x4n ry[NIT];
	auto fetch = [&](auto idof, int b0, int m) {
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int idx = it * 64 + lane, r = idx / CPR, ch = idx - r * CPR;
			const int c = idof(b0 + r < m ? b0 + r : (m > 0 ? m - 1 : 0));
			const f32x4n v = *(const f32x4n *)(a.cent + (size_t)c * a.sdp + 4 * ch);
			f32x4n o = v;
			if (a.interleaved) { // stored [k0,k2,k1,k3] (bit 4 of the row clear) or [k1,k3,k0,k2]
				const bool f = (c >> 4) & 1;
				o[0] = f ? v[2] : v[0], o[1] = f ? v[0] : v[2], o[2] = f ? v[3] : v[1], o[3] = f ? v[1] : v[3];
			}
			ry[it] = o;
		}
	};
	auto spill = [&]() { // registers -> the LDS tile
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int idx = it * 64 + lane, r = idx / CPR, ch = idx - r * CPR;
			*(f32x4n *)(yrows + r * PITCH + 4 * ch) = ry[it];
		}
		asm volatile("" ::: "memory"); // (one wavefront: LDS keeps its accesses in order)
	};
	auto chain = [&](int c) -> float { // lane < 32: candidate `lane` of the tile
		const float *y = yrows + lane * PITCH;
		float acc = 0.f;
		for (int ch = 0; ch < nch; ++ch) {
			const f32x4n xv = *(const f32x4n *)(xs + 4 * ch);
			const f32x4n yv = *(const f32x4n *)(y + 4 * ch);
#pragma unroll
			for (int e = 0; e < 4; ++e)
				acc = fmaf(xv[e], yv[e], acc);
		}
		float dis = fmaf(-2.0f, acc, xn + a.cn[c]);
		return dis < 0.f ? 0.f : dis; // FAISS: if (dis < 0) dis = 0  (NaN stays NaN)
	};
	unsigned long long mine = CB_EMPTY;
	if (!all) {
		{
			const uint4 *src = (const uint4 *)(a.cand + (size_t)q * CB16_CAP);
			for (int c8 = lane; c8 * 8 < n; c8 += 64)
				((uint4 *)cl)[c8] = src[c8];
		}
		asm volatile("" ::: "memory");
		auto idof = [&](int i) { return (int)cl[i]; };
		if (n > 0)
			fetch(idof, 0, n);
		for (int b0 = 0; b0 < n; b0 += 32) {
			spill();
			if (b0 + 32 < n)
				fetch(idof, b0 + 32, n);
			if (lane < 32 && b0 + lane < n) {
				const int c = (int)cl[b0 + lane];
				keys[b0 + lane] = cb16_key(chain(c), c);
			}
			asm volatile("" ::: "memory"); // (the next tile overwrites yrows)
		}
		if (n > 0)
			mine = cb_select_wave<false>(keys, n, a.np, lane, surv, top);
	} else {
		// every centroid: keys[0, np) = the best so far, keys[np, np + m) = the next m <= CB16_CHUNK exact keys
		int have = 0;
		for (int c00 = 0; c00 < a.nlist; c00 += CB16_CHUNK) {
			const int m = a.nlist - c00 < CB16_CHUNK ? a.nlist - c00 : CB16_CHUNK;
			if (lane < have)
				keys[lane] = mine;
			auto idof = [&](int i) { return c00 + i; };
			fetch(idof, 0, m);
			for (int b0 = 0; b0 < m; b0 += 32) {
				spill();
				if (b0 + 32 < m)
					fetch(idof, b0 + 32, m);
				if (lane < 32 && b0 + lane < m) {
					const int c = c00 + b0 + lane;
					keys[have + b0 + lane] = cb16_key(chain(c), c);
				}
				asm volatile("" ::: "memory");
			}
			mine = cb_select_wave<false>(keys, have + m, a.np, lane, surv, top);
			have = a.np;
		}
		if (lane == 0 && a.stats)
			atomicAdd(a.stats, 1ull);
	}
	if (lane < a.np) {
		const bool hv = mine != CB_EMPTY;
		a.outD[q * a.np + lane] = hv ? __uint_as_float((unsigned)(mine >> 32)) : FLT_MAX;
		a.outI[q * a.np + lane] = hv ? (long long)(unsigned)mine + a.label_offset : -1ll;
	}
}

bool coarse_bf16_supported(int d, int64_t nlist, int64_t np) { // (16 < d <= 128: the f32 rows have a pitch of 32, 64 or 128 floats)
	return collect_store_dims(d) == 128 && nlist >= 256 && nlist <= 65536 && nlist % 16 == 0 && np >= 1 && np <= 64 && np < nlist;
}
size_t coarse_bf16_cand_bytes(int64_t nq) {
	return (size_t)nq * CB16_CAP * sizeof(unsigned short);
}

// d_qf / d_qn / d_e2: what launch_collect_query_prep left for these nq queries; d_yb / d_beta: the quantizer's centred bf16 store
void launch_coarse_bf16(const float *d_x, int64_t nq, int d, const void *d_qf, const float *d_qn, const float *d_e2, const unsigned short *d_yb,
                        const float *d_beta, const float *d_cent, int sdp, int interleaved, const float *d_cn, int64_t nlist, int64_t np,
                        unsigned short *d_cand, int *d_ccount, float *d_outD, int64_t *d_outI, int64_t label_offset,
                        unsigned long long *d_stats, hipStream_t st) {
	if (nq <= 0)
		return;
	CoarseBf16Args f;
	memset(&f, 0, sizeof f);
	f.qf = (const bf16x8 *)d_qf, f.yb = d_yb, f.beta = d_beta, f.e2 = d_e2;
	f.nq = (int)nq, f.nlist = (int)nlist, f.np = (int)np, f.cand = d_cand, f.ccount = d_ccount;
	hipLaunchKernelGGL(coarse_bf16_filter_kernel, dim3((unsigned)((nq + CB16_QB - 1) / CB16_QB)), dim3(256), 0, st, f);
	CoarseExactArgs e;
	memset(&e, 0, sizeof e);
	e.x = d_x, e.d = d, e.nq = (int)nq, e.nlist = (int)nlist, e.np = (int)np, e.cent = d_cent, e.sdp = sdp, e.interleaved = interleaved;
	e.qn = d_qn, e.cn = d_cn, e.cand = d_cand, e.ccount = d_ccount, e.outD = d_outD, e.outI = (long long *)d_outI, e.label_offset = label_offset;
	e.stats = d_stats;
	if (sdp == 128)
		hipLaunchKernelGGL(coarse_bf16_exact_kernel<128>, dim3((unsigned)nq), dim3(64), 0, st, e);
	else if (sdp == 64)
		hipLaunchKernelGGL(coarse_bf16_exact_kernel<64>, dim3((unsigned)nq), dim3(64), 0, st, e);
	else
		hipLaunchKernelGGL(coarse_bf16_exact_kernel<32>, dim3((unsigned)nq), dim3(64), 0, st, e);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
