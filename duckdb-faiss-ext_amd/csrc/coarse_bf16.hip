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

constexpr int CB16_QB = 128;    // queries per workgroup of the filter (32 per wave)
constexpr int CB16_CAP = 512;   // candidate ids per query (more: the exact kernel computes every centroid for that query)
constexpr int CB16_CHUNK = 448; // exact-all: new keys per selection round (CAP - 64 kept)
constexpr int CB16_LCAP = 64;   // candidate ids per query and SLICE in the filter's LDS list
constexpr int CB16_ROWB = 272;  // bytes per staged centroid row in LDS (256 + 16: a fragment read's 8 lanes hit 8 x 16 distinct bytes of a 128-byte bank row)
constexpr int CB16_CPS = 32;    // row classes per query and slice: (lane group, register, tile parity)

// Geometry (v3).  v1 / v2 gave a workgroup 32 queries and let its four waves stream disjoint quarters of the bf16 store straight from
// L2: every 32 queries pulled the whole store (1 MB at C3) through the L2 -> CU fabric twice -- 626 MB per search, ~ 5 TB/s, 130 us: bound
// by L2 bandwidth at 160 TFLOP/s.  Now a workgroup is 128 queries x ONE SLICE of the centroids (grid = query blocks x slices): the four
// waves hold 32 queries each and share every 64-row chunk through LDS (loaded once per workgroup, double-buffered in registers -> LDS),
// so the store crosses the fabric once per 128 queries and pass.  The price: a query's class maxima come from several workgroups --
// pass 1 and pass 2 are two launches with the maxima in global memory between them.
struct CoarseBf16Args {
	const bf16x8 *qf;         // [(qblk16 * 4 + kb) * 64 + lane]: bf16(2 x'), csrc/flat_collect.hip collect_query_prep_kernel
	const unsigned short *yb; // [nlist (+ pad)][128] bf16 centred centroids
	const float *beta;        // [nlist] -||c'||^2
	const float *e2;          // [nq rounded up to 256] 2E(q); NaN: the bound is not finite
	int nq, nlist, np;
	int nslice, tiles_per_slice; // (tiles_per_slice: a multiple of 4)
	float *cls;               // [nq][nslice * CB16_CPS] class maxima (pass 1 -> threshold kernel)
	float *thr;               // [nq] T - 2E (threshold kernel -> pass 2)
	unsigned short *cand;     // [nq][CB16_CAP]
	int *ccount;              // [nq] candidates of the query; > CB16_CAP: overflowed; < 0: no finite bound
};

// PASS 1: class maxima of this workgroup's (128 queries, slice) -> a.cls; PASS 2: thresholds from a.cls, then the candidates
template <int PASS>
__global__ __launch_bounds__(256, 2) void coarse_bf16_filter_kernel(const CoarseBf16Args a) {
	__shared__ __attribute__((aligned(16))) unsigned char tile_s[2][64 * CB16_ROWB];
	__shared__ __attribute__((aligned(16))) float beta_s[2][64];
	__shared__ float thr_s[CB16_QB];
	__shared__ int cnt_s[CB16_QB], base_s[CB16_QB];
	__shared__ unsigned short cand_s[PASS == 2 ? CB16_QB : 1][CB16_LCAP];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int col = lane & 15, rg = lane >> 4;
	const int slice = blockIdx.x % a.nslice;
	const long long q0 = (long long)(blockIdx.x / a.nslice) * CB16_QB, qw = q0 + 32 * wave;
	const int ntile = a.nlist >> 4; // (nlist % 16 == 0: coarse_bf16_supported)
	const int t0 = slice * a.tiles_per_slice;
	const int t1 = t0 + a.tiles_per_slice < ntile ? t0 + a.tiles_per_slice : ntile;
	const int nchunk = t1 > t0 ? (t1 - t0 + 3) >> 2 : 0;
	// the wave's B operands: 2 blocks of 16 queries x 4 k-blocks (resident: 32 VGPRs)
	bf16x8 bq[2][4];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int kb = 0; kb < 4; ++kb)
			bq[i][kb] = a.qf[((qw / 16 + i) * 4 + kb) * 64 + lane];
	if (PASS == 2 && tid < CB16_QB) {
		thr_s[tid] = q0 + tid < a.nq ? a.thr[q0 + tid] : __uint_as_float(0x7fc00000u); // (coarse_bf16_threshold_kernel)
	}
	// chunk c of the slice = tiles t0 + 4 c .. + 3 = 64 rows x 256 bytes: thread tid fetches 16-byte pieces idx = 256 i + tid (row idx >> 4,
	// piece idx & 15 -- a wave reads 1 KB contiguous), rows past the slice's end are zero
	uint4 g[4];
	float4 gb = {0.f, 0.f, 0.f, 0.f};
	auto fetch = [&](int c) {
		const int row0 = 16 * (t0 + 4 * c), rend = 16 * t1;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int idx = 256 * i + tid, row = row0 + (idx >> 4);
			g[i] = row < rend ? *(const uint4 *)(a.yb + ((size_t)row << 7) + 8 * (idx & 15)) : uint4{0u, 0u, 0u, 0u};
		}
		if (tid < 16)
			gb = row0 + 4 * tid < rend ? *(const float4 *)(a.beta + row0 + 4 * tid) : float4{0.f, 0.f, 0.f, 0.f};
	};
	auto stage = [&](int buf) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int idx = 256 * i + tid;
			*(uint4 *)(tile_s[buf] + (idx >> 4) * CB16_ROWB + 16 * (idx & 15)) = g[i];
		}
		if (tid < 16)
			*(float4 *)(beta_s[buf] + 4 * tid) = gb;
	};
	float cm[2][4][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int r = 0; r < 4; ++r)
			cm[i][r][0] = cm[i][r][1] = -FLT_MAX;
	if (nchunk > 0) {
		fetch(0);
		stage(0);
	}
	__syncthreads(); // (chunk 0 staged; PASS 2: thr_s / cnt_s set)
	float th[2] = {0.f, 0.f};
	int hc[2] = {0, 0}; // candidates of query (i, col) so far: the same number in the query's four lane groups
	if (PASS == 2)
		th[0] = thr_s[32 * wave + col], th[1] = thr_s[32 * wave + 16 + col];
	for (int c = 0; c < nchunk; ++c) {
		const int buf = c & 1;
		if (c + 1 < nchunk)
			fetch(c + 1); // (in flight under this chunk's MFMAs)
#pragma unroll
		for (int tt = 0; tt < 4; ++tt) {
			const int t = t0 + 4 * c + tt;
			if (t < t1) { // (workgroup-uniform)
				const unsigned char *rowp = tile_s[buf] + (16 * tt + col) * CB16_ROWB + 16 * rg;
				bf16x8 A[4];
#pragma unroll
				for (int kb = 0; kb < 4; ++kb)
					A[kb] = *(const bf16x8 *)(rowp + 64 * kb);
				const f32x4n Y = *(const f32x4n *)(beta_s[buf] + 16 * tt + 4 * rg); // C rows 4 rg + r of the tile
				f32x4n acc[2];
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], bq[i][0], Y, 0, 0, 0);
#pragma unroll
					for (int kb = 1; kb < 4; ++kb)
						acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb], bq[i][kb], acc[i], 0, 0, 0);
				}
				if (PASS == 1) {
#pragma unroll
					for (int i = 0; i < 2; ++i)
#pragma unroll
						for (int r = 0; r < 4; ++r)
							cm[i][r][tt & 1] = acc[i][r] > cm[i][r][tt & 1] ? acc[i][r] : cm[i][r][tt & 1]; // (NaN never wins)
				} else {
					// A query (column block i, column col) belongs to THIS wave alone; its four lane groups (rg) each keep the same
					// count in a register and take their list positions from ballots -- no LDS atomic, no wait (v3a: one returning
					// ds_add per hit, ~ 200 cycles each: pass 2 took twice pass 1's time)
#pragma unroll
					for (int i = 0; i < 2; ++i) {
						const float mx = __builtin_fmaxf(__builtin_fmaxf(acc[i][0], acc[i][1]), __builtin_fmaxf(acc[i][2], acc[i][3]));
						if (__builtin_amdgcn_ballot_w64(mx >= th[i]) != 0ull) { // (wave-uniform)
#pragma unroll
							for (int r = 0; r < 4; ++r) {
								const bool hit = acc[i][r] >= th[i];
								const unsigned long long b = __builtin_amdgcn_ballot_w64(hit) >> col; // bits 0 / 16 / 32 / 48: this column's lane groups
								const unsigned long long same = b & 0x0001000100010001ull;
								const int before = __builtin_popcountll(same & ((1ull << (16 * rg)) - 1ull));
								if (hit && hc[i] + before < CB16_LCAP)
									cand_s[32 * wave + 16 * i + col][hc[i] + before] = (unsigned short)(16 * t + 4 * rg + r);
								hc[i] += __builtin_popcountll(same);
							}
						}
					}
				}
			}
		}
		if (c + 1 < nchunk)
			stage(buf ^ 1); // (its last readers passed the barrier at the end of chunk c - 1)
		__syncthreads();
	}
	if (PASS == 1) {
		// query 16 i + col of the wave: classes slice * 32 + 8 rg + 2 r + parity (8 consecutive floats per lane and query)
		const int nc = a.nslice * CB16_CPS;
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const long long q = qw + 16 * i + col;
			if (q < a.nq) {
				float *dst = a.cls + (size_t)q * nc + slice * CB16_CPS + 8 * rg;
				*(float4 *)dst = float4{cm[i][0][0], cm[i][0][1], cm[i][1][0], cm[i][1][1]};
				*(float4 *)(dst + 4) = float4{cm[i][2][0], cm[i][2][1], cm[i][3][0], cm[i][3][1]};
			}
		}
		if (slice == 0 && tid < CB16_QB && q0 + tid < a.nq)
			a.ccount[q0 + tid] = 0; // (pass 2's workgroups add to it)
		return;
	}
	if (rg == 0) // (the counts of the wave's own 32 queries)
		cnt_s[32 * wave + col] = hc[0], cnt_s[32 * wave + 16 + col] = hc[1];
	__syncthreads();
	// the slice's lists -> the queries' global lists: ONE returning atomic per (query, slice) that has candidates -- lane l < 32 reserves
	// for the wave's query l, all 32 atomics in flight at once (v3b walked the 32 queries one by one behind 32 round trips: 25 us)
	{
		const int qi = 32 * wave + (lane & 31);
		const long long q = q0 + qi;
		const int n = cnt_s[qi];
		const bool finite = thr_s[qi] == thr_s[qi];
		int base = 0;
		if (lane < 32 && q < a.nq) {
			if (!finite) {
				if (slice == 0)
					a.ccount[q] = -(1 << 30); // (nothing passed anywhere: no other workgroup touches the counter)
			} else if (n > 0) { // (an overflowing slice pushes the count past CAP: the exact kernel then computes every centroid of the query)
				base = atomicAdd(a.ccount + q, n > CB16_LCAP ? CB16_CAP + 1 : n);
			}
			base_s[qi] = base;
		}
	}
	asm volatile("" ::: "memory"); // (base_s of the wave's own queries: written and read by this wave only)
	for (int qi = 32 * wave; qi < 32 * wave + 32; ++qi) {
		const long long q = q0 + qi;
		const int n = cnt_s[qi];
		if (q >= a.nq || n == 0 || n > CB16_LCAP || !(thr_s[qi] == thr_s[qi]))
			continue;
		const int base = base_s[qi];
		if (lane < n && base + lane < CB16_CAP)
			a.cand[(size_t)q * CB16_CAP + base + lane] = cand_s[qi][lane];
	}
}

// T(q) = the np-th largest of the query's nslice * 32 class maxima (bitwise search on "smaller is better" keys); thr = T - 2E as the
// Flat scan forms it.  One wavefront per query (v3a derived the 128 thresholds of a query block in EVERY slice's workgroup: 166 us).
__global__ __launch_bounds__(64) void coarse_bf16_threshold_kernel(const CoarseBf16Args a) {
	const int lane = threadIdx.x;
	const long long q = blockIdx.x;
	const int nc = a.nslice * CB16_CPS;
	unsigned key[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int c = 64 * j + lane;
		key[j] = c < nc ? skey(a.cls[(size_t)q * nc + c]) : 0xffffffffu;
	}
	// the np-th smallest key from above: the largest U with #(key < U) < np, searched on the top 22 bits and completed with ones -- any
	// key >= the true one is a valid (looser) bound; the lost 10 bits are 2^-13 of |s|, a hundredth of 2E
	unsigned U = 0u;
#pragma unroll 1
	for (int b = 31; b >= 10; --b) {
		const unsigned t = U | (1u << b);
		int cnt = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j)
			cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] < t));
		if (cnt < a.np)
			U = t;
	}
	U |= 0x3ffu;
	if (lane == 0)
		a.thr[q] = skey2f(U) - a.e2[q]; // (NaN: nothing passes; fewer than np classes set: -FLT_MAX - e2, everything passes)
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

// One wavefront per query, 64 candidates per round -- lane <-> candidate.  A round walks the rows in STEPS of 32 dims: the step's
// 64 x 128 bytes come in with coalesced 16-byte loads (8 neighbouring lanes = one row's cache line; v1 let every lane walk its own row:
// 64 cache lines per load instruction, 256 us at C3), are de-interleaved and transposed through a 9 KB LDS tile, and every lane continues
// ITS candidate's k-ordered chain -- fmaf(x_k, y_k, acc) from k = 0 -- across the steps; then fmaf(-2, acc, xn + cn[c]), clamped at 0:
// csrc/coarse_select.hip coarse_dist_kernel's value bit for bit.  The next step's loads are in flight under the chains.
template <int DP>
__global__ __launch_bounds__(64) void coarse_bf16_exact_kernel(const CoarseExactArgs a) {
	constexpr int SD = 32, NSTEP = DP / SD, PITCH = SD + 4; // dims per step; steps per round; floats per LDS row
	__shared__ __attribute__((aligned(16))) float yt[64 * PITCH];
	__shared__ __attribute__((aligned(16))) float xs[DP];
	__shared__ unsigned long long keys[CB16_CAP];
	__shared__ unsigned long long surv[256];
	__shared__ unsigned long long top[64];
	__shared__ __attribute__((aligned(16))) unsigned short cl[CB16_CAP];
	const int lane = threadIdx.x;
	const long long q = blockIdx.x;
	for (int i = lane; i < DP; i += 64)
		xs[i] = i < a.d ? a.x[q * a.d + i] : 0.f;
	// (the first 64 candidate ids are requested WITH the count, not behind it: one round trip less in front of the row gathers)
	const unsigned short first_id = a.cand[(size_t)q * CB16_CAP + lane];
	const int n = a.ccount[q];
	const float xn = a.qn[q];
	const bool all = n < 0 || n > CB16_CAP;
	// step `st` of the 64 rows idof(b0 + r): piece idx = 64 it + lane -> row r = idx >> 3, 16-byte piece idx & 7 of the step's 128 bytes
	f32x4n ry[8];
	auto fetch = [&](auto idof, int b0, int m, int st) {
#pragma unroll
		for (int it = 0; it < 8; ++it) {
			const int idx = 64 * it + lane, r = idx >> 3, pc = idx & 7;
			const int c = idof(b0 + r < m ? b0 + r : m - 1); // (slots past the list repeat its last row)
			const f32x4n v = *(const f32x4n *)(a.cent + (size_t)c * a.sdp + SD * st + 4 * pc);
			f32x4n o = v;
			if (a.interleaved) { // stored [k0,k2,k1,k3] (bit 4 of the row clear) or [k1,k3,k0,k2]
				const bool f = (c >> 4) & 1;
				o[0] = f ? v[2] : v[0], o[1] = f ? v[0] : v[2], o[2] = f ? v[3] : v[1], o[3] = f ? v[1] : v[3];
			}
			ry[it] = o;
		}
	};
	// one round: the chains of candidates b0 .. b0 + 63 (ids from idof; m = length of the list) -> dis of lane's candidate
	auto round64 = [&](auto idof, int b0, int m) -> float {
		float acc = 0.f;
		fetch(idof, b0, m, 0);
#pragma unroll 1 // (unrolled, the compiler hoists every step's loads: 294 registers, one wave per SIMD)
		for (int st = 0; st < NSTEP; ++st) {
#pragma unroll
			for (int it = 0; it < 8; ++it) {
				const int idx = 64 * it + lane;
				*(f32x4n *)(yt + (idx >> 3) * PITCH + 4 * (idx & 7)) = ry[it];
			}
			asm volatile("" ::: "memory"); // (one wavefront: LDS keeps its accesses in order)
			if (st + 1 < NSTEP)
				fetch(idof, b0, m, st + 1);
			const float *y = yt + lane * PITCH;
#pragma unroll
			for (int ch = 0; ch < SD / 4; ++ch) { // (dims past d: zero on both sides, fmaf(0, 0, acc) = acc)
				const f32x4n xv = *(const f32x4n *)(xs + SD * st + 4 * ch);
				const f32x4n yv = *(const f32x4n *)(y + 4 * ch);
#pragma unroll
				for (int e = 0; e < 4; ++e)
					acc = fmaf(xv[e], yv[e], acc);
			}
			asm volatile("" ::: "memory"); // (the next step overwrites the tile)
		}
		const int c = idof(b0 + lane < m ? b0 + lane : m - 1);
		const float dis = fmaf(-2.0f, acc, xn + a.cn[c]);
		return dis < 0.f ? 0.f : dis; // FAISS: if (dis < 0) dis = 0  (NaN stays NaN)
	};
	unsigned long long mine = CB_EMPTY;
	if (!all) {
		cl[lane] = first_id;
		if (n > 64) {
			const uint4 *src = (const uint4 *)(a.cand + (size_t)q * CB16_CAP);
			for (int c8 = 8 + lane; c8 * 8 < n; c8 += 64)
				((uint4 *)cl)[c8] = src[c8];
		}
		asm volatile("" ::: "memory");
		auto idof = [&](int i) { return (int)cl[i]; };
		for (int b0 = 0; b0 < n; b0 += 64) {
			const float dis = round64(idof, b0, n);
			if (b0 + lane < n)
				keys[b0 + lane] = cb16_key(dis, (int)cl[b0 + lane]);
		}
		asm volatile("" ::: "memory");
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
			for (int b0 = 0; b0 < m; b0 += 64) {
				const float dis = round64(idof, b0, m);
				if (b0 + lane < m)
					keys[have + b0 + lane] = cb16_key(dis, c00 + b0 + lane);
			}
			asm volatile("" ::: "memory");
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
// slices of the centroid range per query block: enough workgroups for the device (~ 2 per CU), at least 2 np classes per query
// (np-th largest of nslice * 32 class maxima: the more classes, the closer T comes to the true np-th value), tiles in chunks of 4
int coarse_bf16_slices(int64_t nq, int64_t nlist, int64_t np) {
	const int64_t nqb = (nq + CB16_QB - 1) / CB16_QB, ntile = nlist >> 4;
	int ns = 1;
	while (ns < 8 && (nqb * ns < 512 || ns * CB16_CPS < 2 * np) && (ntile + 2 * ns - 1) / (2 * ns) >= 4)
		ns *= 2;
	return ns;
}
size_t coarse_bf16_cls_bytes(int64_t nq, int64_t nlist, int64_t np) {
	return (size_t)nq * (coarse_bf16_slices(nq, nlist, np) * CB16_CPS + 1) * sizeof(float); // class maxima + one threshold per query
}

// d_qf / d_qn / d_e2: what launch_collect_query_prep left for these nq queries; d_yb / d_beta: the quantizer's centred bf16 store
void launch_coarse_bf16(const float *d_x, int64_t nq, int d, const void *d_qf, const float *d_qn, const float *d_e2, const unsigned short *d_yb,
                        const float *d_beta, const float *d_cent, int sdp, int interleaved, const float *d_cn, int64_t nlist, int64_t np,
                        unsigned short *d_cand, int *d_ccount, float *d_cls, float *d_outD, int64_t *d_outI, int64_t label_offset,
                        unsigned long long *d_stats, hipStream_t st) {
	if (nq <= 0)
		return;
	CoarseBf16Args f;
	memset(&f, 0, sizeof f);
	f.qf = (const bf16x8 *)d_qf, f.yb = d_yb, f.beta = d_beta, f.e2 = d_e2;
	f.nq = (int)nq, f.nlist = (int)nlist, f.np = (int)np, f.cand = d_cand, f.ccount = d_ccount, f.cls = d_cls;
	f.nslice = coarse_bf16_slices(nq, nlist, np);
	const int ntile = (int)(nlist >> 4);
	f.tiles_per_slice = ((ntile + f.nslice - 1) / f.nslice + 3) / 4 * 4;
	const dim3 grid((unsigned)((nq + CB16_QB - 1) / CB16_QB) * (unsigned)f.nslice);
	f.thr = d_cls + (size_t)nq * f.nslice * CB16_CPS; // (behind the class maxima: coarse_bf16_cls_bytes counts it)
	hipLaunchKernelGGL(coarse_bf16_filter_kernel<1>, grid, dim3(256), 0, st, f);
	hipLaunchKernelGGL(coarse_bf16_threshold_kernel, dim3((unsigned)nq), dim3(64), 0, st, f);
	hipLaunchKernelGGL(coarse_bf16_filter_kernel<2>, grid, dim3(256), 0, st, f);
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
