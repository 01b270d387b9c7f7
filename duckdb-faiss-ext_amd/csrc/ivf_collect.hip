// csrc/ivf_collect.hip -- bf16 COARSE FILTER for the IVFFlat list scan (L2): the flat_collect.hip argument on inverted lists.
//
// Same place in the path as ivf_scan.hip (IndexIVF::search_preassigned + IVFFlatScanner::scan_codes behind
// /root/reference/src/faiss_extension.cpp:631) and the same results: labels and distances are those of the scanner kernel
// and of oracle/orc_core.c's IVF restatement, bit for bit away from exact distance ties.
//
//   rows        list by list (every list padded to a multiple of 64 rows), as RESIDUALS y' = y - c_list in bf16, with
//               beta(row) = -||y'||^2 in f32; the residual against the list's own centroid is the natural centring: the error
//               of a bf16 product scales with ||x'|| ||y'|| = (distance to the centroid)^2-sized numbers, not with ||x|| ||y||
//   work item   (list, <= 128 of the queries that probe it) = ONE wavefront; its queries enter as bf16(2 (x - c_list)) with
//               gamma(slot) = -||x - c_list||^2, and the MFMA chain starts at C = beta(row) + gamma(slot):
//               s = 2 <x', y'> - ||y'||^2 - ||x'||^2 = -||x - y||^2 (approximately), comparable across the lists of a query
//   bound       |s - s_exact| <= E(slot) from ||x'||, the list's largest ||y'|| and d (ivf_collect_pack_kernel).  E differs
//               between the lists a query probes, so the class slots (16 row classes per query, shared by all its items) hold
//               LOWER bounds s - E(list) of distinct rows' exact values; B(q) = their kk-th best <= the exact kk-th value; a
//               row of list l with s >= B - E(l) is a candidate (flat_collect.hip, "candidates", with per-list E)
//   re-scoring  the candidates are grouped by query, recomputed with the scanner's arithmetic (t = x_k - y_k, acc = fmaf(t, t,
//               acc), k ascending, on the ORIGINAL f32 rows) and the k best by (value, position in the list-sorted store)
//               are kept -- the order of ivf_scan_kernel + merge_items_kernel.
// A list is walked in segments of 512 rows, one wavefront each (long lists do not set the pace).  A pre-pass over the first 256
// rows of every query's nearest list (publish only) warms the bounds.
#include "flat_fused.h"
#include "collect_bucket.h"
#include "flat_collect.h" // CL_MFMA_UNITS: the modelled bf16-MFMA term of the bound

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mvs {

typedef __bf16 bf16x8i __attribute__((ext_vector_type(8)));
typedef float f32x4i __attribute__((ext_vector_type(4)));
typedef float f32x2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32i;
typedef __attribute__((address_space(1))) const float glb_f32i;

constexpr int IC_BN = 32;    // rows per tile
constexpr int IC_QCAP = 160; // hit queue of a work item: 8 bytes {value, row} + 1 byte {slot} per entry (20 384 bytes of LDS per wavefront = EIGHT per CU)

struct IvfCollectArgs {
	const int4 *items;         // {row_begin (multiple of 64, padded row space), row_end, qoff, nq_item <= 128}
	const int *nitems_dev;     // device-side item count; the grid is an upper bound
	const int *qidx;           // query number of slot qoff + s
	const void *xi;            // [item][8 column blocks][4 k-blocks][64 lanes] x 16 bytes: bf16(2 (x - c)) fragments
	const float *igamma;       // [item][128] -||x - c||^2
	const float *ie2;          // [item][128] 2E (NaN: the slot is empty / the query is not served here)
	const unsigned short *yb;  // bf16 residual rows [nrows_mf + 192][128]
	const float *beta;         // [nrows_mf + 192] -||y'||^2
	unsigned *gslot;           // [nq][16] class slots
	unsigned long long *stream; // candidates (q << 32 | padded row)
	unsigned long long *stream_cnt;
	float *stream_u;           // (may be null) per entry: an UPPER bound s + E of the row's exact value -- the final-bound filter's input
	long long stream_cap;
	int kk;
	int seg_rows; // rows per block: grid.y walks a list in segments (one wavefront per segment: long lists do not set the pace)
	int collect;  // 0: bound estimation only (publish to the slots, append nothing)
	int refresh;  // tiles between two refreshes of the bounds after the first (option ivf_cl_refresh; 0: 1, 1, 1, 1, 4, 4 ... 16)
	const unsigned *rowmask; // IDSelector active: bit r of word w = padded row 32 w + r is accepted (nullptr: no selector)
	const float *bfix;       // (round 6, lists beyond 32 entries) [nq] FROZEN B(q): a lower bound of the exact kk-th best value found elsewhere -- the class slots are not consulted
	int abl;     // profiling library only (option ivf_cl_abl)
	int nseg;    // segments per item (xcd_map >= 2 decodes the segment from blockIdx.x)
	int gx8;     // workgroups of one segment round (a multiple of 8)
	int xcd_map; // 2: as 1, and the segments of an item are consecutive workgroups of its XCD; 1: XCD j (= blockIdx.x & 7) takes the contiguous item range [j n/8, (j+1) n/8) (option ivf_cl_xcd)
};

__device__ __forceinline__ unsigned ic_skey(float s) { // "larger s is better" as a smaller-is-better key
	return ~f2key(s);
}
__device__ __forceinline__ float ic_skey2f(unsigned k) {
	return key2f(~k);
}
template <int CTRL>
__device__ __forceinline__ float ic_dpp(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// ---- storage: residual rows [n][d] f32 (padding rows are zero) -> bf16 [n][128] + beta + the largest ||y'||^2 of every list
__global__ void ivf_rows_to_bf16_kernel(const float *__restrict__ src, long long nrows, int d,
                                        const int *__restrict__ list_of_blk64, unsigned short *__restrict__ dst,
                                        float *__restrict__ beta, unsigned *__restrict__ list_max_bits, int nlist) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // one (row, 8 dims); 16 neighbours share a row
	const bool live = i < nrows * 16;
	const long long r = live ? i >> 4 : 0;
	const int c8 = (int)(i & 15);
	bf16x8i hi;
	float n2 = 0.f;
	float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (live && (d & 7) == 0) { // 16-byte aligned: two vector loads
		if (c8 * 8 < d) {
			const float4 s0 = *(const float4 *)(src + (size_t)r * d + c8 * 8);
			const float4 s1 = *(const float4 *)(src + (size_t)r * d + c8 * 8 + 4);
			v8[0] = s0.x, v8[1] = s0.y, v8[2] = s0.z, v8[3] = s0.w, v8[4] = s1.x, v8[5] = s1.y, v8[6] = s1.z, v8[7] = s1.w;
		}
	} else if (live) {
#pragma unroll
		for (int e = 0; e < 8; ++e)
			if (c8 * 8 + e < d)
				v8[e] = src[(size_t)r * d + c8 * 8 + e];
	}
	float r2 = 0.f; // ||y' - bf16(y')||^2: the row's actual rounding residual (csrc/flat_collect.hip, "ROUND 4")
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		hi[e] = (__bf16)v8[e];
		const float dl = v8[e] - (float)hi[e];
		n2 = fmaf(v8[e], v8[e], n2);
		r2 = fmaf(dl, dl, r2);
	}
	n2 += ic_dpp<0xB1>(n2), r2 += ic_dpp<0xB1>(r2);
	n2 += ic_dpp<0x4E>(n2), r2 += ic_dpp<0x4E>(r2);
	n2 += ic_dpp<0x141>(n2), r2 += ic_dpp<0x141>(r2);
	n2 += ic_dpp<0x140>(n2), r2 += ic_dpp<0x140>(r2);
	if (!live)
		return;
	*(bf16x8i *)(dst + (size_t)r * 128 + c8 * 8) = hi;
	if (c8 == 0) {
		beta[r] = -n2;
		unsigned *m = list_max_bits + list_of_blk64[r >> 6];
		const unsigned b = __float_as_uint(n2); // (>= 0 or NaN: the bit pattern orders like the value)
		// (agent-scope loads: a plain load is served from this CU's vector cache as the line was first fetched -- every row would then send
		// its atomic: csrc/flat_collect.hip rows_to_bf16_hi_kernel, round 6)
		if (b > __hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			atomicMax(m, b);
		const unsigned br = __float_as_uint(r2);
		if (br > __hip_atomic_load(m + nlist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			atomicMax(m + nlist, br);
	}
}
void launch_ivf_rows_to_bf16(const float *d_res, int64_t nrows, int d, const int *d_list_of_blk64, unsigned short *d_bf,
                             float *d_beta, unsigned *d_list_max_bits, int64_t nlist, hipStream_t st) {
	if (nrows <= 0)
		return;
	const long long total = (long long)nrows * 16;
	hipLaunchKernelGGL(ivf_rows_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_res,
	                   (long long)nrows, d, d_list_of_blk64, d_bf, d_beta, d_list_max_bits, (int)nlist);
	MVS_HIP(hipGetLastError());
}

// ---- per item: query fragments, gamma, 2E ------------------------------------------------------------------------------
// Error bound, in "s" units (s_exact = -D_oracle, the scanner's value), u = 2^-24, S' = ||x'|| ||y'||_max(list), norms inflated
// by 1e-4 for their own rounding:
//   bf16 rounding of both operands (the query operand carries the exact factor 2):        2 (2^-7 + 2^-16) S'
//   MFMA accumulation from C = beta + gamma (CL_MFMA_UNITS = 8 ulp-units of the magnitudes per 16 dimensions -- measured: csrc/flat_collect.hip --,
//   1.25 safety factor as in flat_collect.hip):                                            1.25 (d/16) 8u ((1 + 2^-7) 2 S' + xn' + yn'_max)
//   C = fl(beta + gamma), beta and gamma d-term f32 chains:                               (d + 1) u (xn' + yn'_max)
//   the scanner's value: D = sum fl((x_k - y_k)^2) accumulated in f32, on x' - y' = x - y up to one rounding per component of
//   each residual:                                                                        (d + 8) u (||x'|| + ||y'||_max)^2
//   E = the sum; e2 = 2 E (1 + 2^-10) + slack.  Non-finite -> NaN (the query is re-run on the scanner kernel).
// Inner product (IS_L2 = false): the query enters as bf16(x) (not centred: <x, y> = <x, y'> + <x, c>), gamma = <x, c_list>, beta = 0;
//   E = (2^-7 + 2^-16) S + 1.25 (d/16) 8u ((1 + 2^-7 + 2^-16) S + |gamma|) + d u ||x|| ||c|| (gamma's chain) + u S (y' rounding)
//       + d u ||x|| (||c|| + ||y'||_max) (the scanner's chain over the original row),   S = ||x|| ||y'||_max(list)
// 2 E of one (query, list) pair from the four sums of ivf_collect_pack_kernel (NaN: not finite -> the query is re-run on the scanner)
template <bool IS_L2>
__device__ __forceinline__ float ivf_slot_e2(float xn, float cn, float dq2, float yn, float dyn, int d, int bound_mode) {
	const double u = 5.9604644775390625e-08, infl = 1.0001;
	const double nx = sqrt((double)xn * infl), ny = sqrt((double)yn * infl), nc = sqrt((double)cn * infl), S = nx * ny;
	// bf16 rounding of both operands: the worst case per element, or (round 4, csrc/flat_collect.hip collect_bounds_kernel) from
	// the ACTUAL residual norms: |<a, y'> - <bf(a), bf(y')>| <= ||a - bf(a)|| ||y'||_max + (||a|| + ||a - bf(a)||) ||y' - bf(y')||_max
	const double al = IS_L2 ? 2.0 : 1.0;
	const double ndq = sqrt((double)dq2 * infl), ndy = sqrt((double)dyn * infl);
	const double rnd_worst = al * (0.0078125 + 1.52587890625e-05) * S, rnd_actual = ndq * ny + (al * nx + ndq) * ndy;
	const double rnd = bound_mode == 0 ? rnd_worst : (rnd_actual < rnd_worst || !(rnd_actual == rnd_actual) ? rnd_actual : rnd_worst);
	double E;
	if (IS_L2)
		E = rnd + 1.25 * ((double)d / 16.0) * CL_MFMA_UNITS * u * ((1.0 + 0.0079) * 2.0 * S + (double)xn + yn) +
		    ((double)d + 1.0) * u * ((double)xn + yn) + ((double)d + 8.0) * u * (nx + ny) * (nx + ny);
	else
		E = rnd + 1.25 * ((double)d / 16.0) * CL_MFMA_UNITS * u * ((1.0 + 0.0079) * S + nx * nc) + (double)d * u * nx * nc + u * S +
		    ((double)d + 2.0) * u * nx * (nc + ny);
	const float r = (float)(2.0 * E * (1.0 + 0.0009765625) + 8.0 * u * (S + (double)xn + yn + nx * nc) + 1e-30);
	if (isfinite(xn) && isfinite(yn) && isfinite(dq2) && isfinite(dyn) && isfinite(r) && r < 1e30f)
		return r;
	return __uint_as_float(0x7fc00000u);
}
template <bool IS_L2>
__device__ __forceinline__ void ivf_collect_pack_body(unsigned bid, const float *__restrict__ x, int d, const int4 *__restrict__ items,
                                                      const int *__restrict__ nitems_dev, const int *__restrict__ qidx,
                                                      const float *__restrict__ cent,
                                                      const int *__restrict__ list_of_blk64,
                                                      const unsigned *__restrict__ list_max_bits,
                                                      bf16x8i *__restrict__ xi, float *__restrict__ igamma,
                                                      float *__restrict__ ie2, int *__restrict__ qfail,
                                                      const long long *__restrict__ coarse, int np, float *__restrict__ ie2_pre,
                                                      int nlist, int bound_mode) {
	// (round 4: one workgroup per HALF item -- slots 64 h .. 64 h + 63 -- with half the LDS: four workgroups per CU instead of two,
	// and an item of <= 64 queries costs one workgroup's worth of work; the kernel waits on memory more than it computes)
	const int item = (int)(bid >> 1), s0 = 64 * (int)(bid & 1u);
	if (item >= *nitems_dev)
		return;
	const int4 it = items[item];
	if (s0 >= it.w && s0 > 0)
		return; // (the first half also writes the NaN bounds of the unused slots 64 .. 127)
	const int nsl = it.w - s0 < 64 ? (it.w - s0 > 0 ? it.w - s0 : 0) : 64; // slots of this half that hold a query
	const int l = list_of_blk64[it.x >> 6]; // every list starts at a multiple of 64 rows
	// the item's query rows and its centroid pass through LDS once (coalesced 16-byte loads when d % 4 == 0): both halves below
	// walk them element by element -- from global memory that was one dependent, uncoalesced load per element
	extern __shared__ __attribute__((aligned(16))) float pk_lds[]; // [64][d + 1] query rows, then [d] the centroid
	const int xp = d + 1;
	float *xs = pk_lds, *c = pk_lds + 64 * xp;
	if ((d & 3) == 0) {
		const int cpr = d >> 2;
		for (int i = threadIdx.x; i < nsl * cpr; i += 256) {
			const int sl = i / cpr, ch = i - sl * cpr;
			const float4 v = *(const float4 *)(x + (size_t)qidx[it.z + s0 + sl] * d + 4 * ch);
			float *o = xs + sl * xp + 4 * ch;
			o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
		}
	} else {
		for (int i = threadIdx.x; i < nsl * d; i += 256) {
			const int sl = i / d, kk = i - sl * d;
			xs[sl * xp + kk] = x[(size_t)qidx[it.z + s0 + sl] * d + kk];
		}
	}
	for (int i = threadIdx.x; i < d; i += 256)
		c[i] = cent[(size_t)l * d + i];
	__syncthreads();
	bf16x8i *dst = xi + (size_t)item * (8 * 4 * 64) + (size_t)(s0 >> 4) * (4 * 64);
	// (only the column blocks that hold a slot: the scan kernel does not fetch the others)
	for (int i = threadIdx.x; i < ((nsl + 15) >> 4) * 4 * 64; i += 256) { // (column block of this half, k-block, lane)
		const int lane = i & 63, kb = (i >> 6) & 3, cb = i >> 8;
		const int sl = cb * 16 + (lane & 15);
		bf16x8i v;
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int kk = kb * 32 + 8 * (lane >> 4) + e;
			float o = 0.f;
			if (sl < nsl && kk < d) {
				const float xv = xs[sl * xp + kk];
				o = IS_L2 ? 2.0f * __fsub_rn(xv, c[kk]) : xv;
			}
			v[e] = (__bf16)o;
		}
		dst[i] = v;
	}
	// slot bounds: this half's 64 slots; the first half of an item with <= 64 queries also writes the NaN entries of slots 64 .. 127
	// The four sums of a slot -- ||x'||^2 (L2) or ||x||^2 (IP); ||c||^2; <x, c>; ||a - bf16(a)||^2 of the operand a -- as four partial
	// chains over the dimensions k = j (mod 4), one per wave (a single 128-step chain per slot was the kernel's critical path: 8 us
	// per workgroup whatever the number of slots; any summation order is inside the d u sum|terms| the bound allows for these sums)
	__shared__ float part[4][64][4];
	{
		const int sl = threadIdx.x & 63, j = threadIdx.x >> 6;
		float xn = 0.f, cn = 0.f, xc = 0.f, dq2 = 0.f;
		if (sl < nsl) {
#pragma unroll 4
			for (int kk = j; kk < d; kk += 4) {
				const float xv = xs[sl * xp + kk];
				const float r = IS_L2 ? __fsub_rn(xv, c[kk]) : xv;
				xn = fmaf(r, r, xn);
				cn = fmaf(c[kk], c[kk], cn);
				xc = fmaf(xv, c[kk], xc);
				const float a = IS_L2 ? 2.0f * r : r; // exactly what the loop above rounded to bf16
				const float dl = a - (float)(__bf16)a;
				dq2 = fmaf(dl, dl, dq2);
			}
		}
		part[j][sl][0] = xn, part[j][sl][1] = cn, part[j][sl][2] = xc, part[j][sl][3] = dq2;
	}
	__syncthreads();
	const int nout = (s0 == 0 && it.w <= 64) ? 128 : 64;
	if ((int)threadIdx.x < nout) {
		const int sl = threadIdx.x, slot = s0 + sl;
		float g = 0.f, e2 = __uint_as_float(0x7fc00000u);
		if (slot < it.w) {
			const int q = qidx[it.z + slot];
			const float xn = ((part[0][sl][0] + part[1][sl][0]) + part[2][sl][0]) + part[3][sl][0];
			const float cn = ((part[0][sl][1] + part[1][sl][1]) + part[2][sl][1]) + part[3][sl][1];
			const float xc = ((part[0][sl][2] + part[1][sl][2]) + part[2][sl][2]) + part[3][sl][2];
			const float dq2 = ((part[0][sl][3] + part[1][sl][3]) + part[2][sl][3]) + part[3][sl][3];
			g = IS_L2 ? -xn : xc;
			const float yn = __uint_as_float(list_max_bits[l]), dyn = __uint_as_float(list_max_bits[nlist + l]);
			e2 = ivf_slot_e2<IS_L2>(xn, cn, dq2, yn, dyn, d, bound_mode);
			if (!(e2 == e2))
				qfail[q] = 1;
		}
		igamma[(size_t)item * 128 + slot] = g;
		ie2[(size_t)item * 128 + slot] = e2;
		// the pre-pass over THESE items (option ivf_cl_prepass_shared): only the slots whose list is their query's nearest take
		// part -- E = NaN switches a slot off (nothing of it passes, nothing is published)
		if (ie2_pre) {
			const bool nearest = slot < it.w && coarse[(size_t)qidx[it.z + slot] * np] == (long long)l;
			ie2_pre[(size_t)item * 128 + slot] = nearest ? e2 : __uint_as_float(0x7fc00000u);
		}
	}
}
// The nearest-list pre-pass has ONE pair per query, spread over ~ nlist items of a few slots each: the item-wise kernel above
// then spends a workgroup, and four dependent global round trips, on two or three queries (70 us at C3, as long as the main
// pass's packing of 32 times the pairs).  Here one wave per QUERY: slot code (ivf_group_scatter_*: item << 7 | slot) -> item ->
// list -> centroid; lanes 0..15 hold the sixteen 8-element pieces of the slot's fragment column, the four sums are reduced over
// them.  Slots of an item that no query owns are not written: the scan gives them a NaN bound whatever is stored (own_q < 0).
template <bool IS_L2>
__device__ __forceinline__ void ivf_collect_pack_pairs_body(unsigned bid, const float *__restrict__ x, int d, long long npairs, int np,
                                                            const int *__restrict__ slots, const int4 *__restrict__ items,
                                                            const float *__restrict__ cent, const int *__restrict__ list_of_blk64,
                                                            const unsigned *__restrict__ list_max_bits, bf16x8i *__restrict__ xi,
                                                            float *__restrict__ igamma, float *__restrict__ ie2,
                                                            int *__restrict__ qfail, int nlist, int bound_mode) {
	// sixteen lanes per (query, list) pair -- the sixteen 8-element pieces of the slot's fragment column --, sixteen pairs per workgroup
	const long long p = (long long)bid * 16 + (threadIdx.x >> 4);
	const int l16 = threadIdx.x & 15;
	const int code = p < npairs ? slots[p] : -1;
	float xn = 0.f, cn = 0.f, xc = 0.f, dq2 = 0.f;
	int item = 0, slot = 0, l = 0;
	const long long q = p / np;
	if (code >= 0) {
		item = code >> 7, slot = code & 127;
		const int4 it = items[item];
		l = list_of_blk64[it.x >> 6];
		const int kb = l16 >> 2, hq = l16 & 3;
		bf16x8i v;
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int kk = kb * 32 + 8 * hq + e;
			const bool in = kk < d;
			const float xv = in ? x[q * d + kk] : 0.f, cv = in ? cent[(size_t)l * d + kk] : 0.f;
			const float r = IS_L2 ? __fsub_rn(xv, cv) : xv;
			const float a = IS_L2 ? 2.0f * r : r;
			v[e] = (__bf16)a;
			xn = fmaf(r, r, xn);
			cn = fmaf(cv, cv, cn);
			xc = fmaf(xv, cv, xc);
			const float dl = a - (float)(__bf16)a;
			dq2 = fmaf(dl, dl, dq2);
		}
		xi[(size_t)item * (8 * 4 * 64) + (size_t)(((slot >> 4) * 4 + kb) * 64 + hq * 16 + (slot & 15))] = v;
	}
#pragma unroll
	for (int o = 8; o >= 1; o >>= 1) { // (within the pair's sixteen lanes)
		xn += __shfl_xor(xn, o);
		cn += __shfl_xor(cn, o);
		xc += __shfl_xor(xc, o);
		dq2 += __shfl_xor(dq2, o);
	}
	if (code >= 0 && l16 == 0) {
		const float yn = __uint_as_float(list_max_bits[l]), dyn = __uint_as_float(list_max_bits[nlist + l]);
		const float e2 = ivf_slot_e2<IS_L2>(xn, cn, dq2, yn, dyn, d, bound_mode);
		if (!(e2 == e2))
			qfail[q] = 1;
		igamma[(size_t)item * 128 + slot] = IS_L2 ? -xn : xc;
		ie2[(size_t)item * 128 + slot] = e2;
	}
}
// Round 5: the packing of BOTH passes in one launch -- the first nb0 workgroups pack the nearest-list pre-pass pair by pair (set 0:
// its own items / fragments / bounds), the others the main pass item by item (set 1).
struct IvfPack2Args {
	const float *x;
	int d, nlist, bound_mode;
	unsigned nb0;
	long long nq;
	const float *cent;
	const int *list_of_blk64;
	const unsigned *list_max_bits;
	int *qfail;
	// set 0 (pairs)
	const int *slots0;
	const int4 *items0;
	bf16x8i *xi0;
	float *igamma0, *ie20;
	// set 1 (items)
	const int4 *items1;
	const int *nitems1;
	const int *qidx1;
	bf16x8i *xi1;
	float *igamma1, *ie21;
	unsigned *gslot; // [nq][nclass] class slots -> neutral
	int nclass;
	int *ctl_hdr;    // the control block's 64 header ints -> 0
	int *flag_cnt;   // -> 0
};
template <bool IS_L2>
__global__ __launch_bounds__(256) void ivf_collect_pack2_kernel(const IvfPack2Args a) {
	if (blockIdx.x < a.nb0) {
		// (what launch_init_slots and the control block's memset did in launches of their own)
		const long long q = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
		const int l16 = threadIdx.x & 15;
		if (q < a.nq)
			for (int c = l16; c < a.nclass; c += 16)
				a.gslot[q * a.nclass + c] = ic_skey(-FLT_MAX);
		if (blockIdx.x == 0 && threadIdx.x < 64) {
			a.ctl_hdr[threadIdx.x] = 0;
			if (threadIdx.x == 0)
				*a.flag_cnt = 0;
		}
	}
	if (blockIdx.x < a.nb0)
		ivf_collect_pack_pairs_body<IS_L2>(blockIdx.x, a.x, a.d, a.nq, 1, a.slots0, a.items0, a.cent, a.list_of_blk64, a.list_max_bits, a.xi0,
		                                   a.igamma0, a.ie20, a.qfail, a.nlist, a.bound_mode);
	else
		ivf_collect_pack_body<IS_L2>(blockIdx.x - a.nb0, a.x, a.d, a.items1, a.nitems1, a.qidx1, a.cent, a.list_of_blk64, a.list_max_bits,
		                             a.xi1, a.igamma1, a.ie21, a.qfail, nullptr, 0, nullptr, a.nlist, a.bound_mode);
}
void launch_ivf_collect_pack2(int metric, const float *d_x, int d, int64_t nq, const int *d_slots0, const void *d_items0, void *d_xi0,
                              float *d_igamma0, float *d_ie20, const void *d_items1, const int *d_nitems1, int max_items1,
                              const int *d_qidx1, void *d_xi1, float *d_igamma1, float *d_ie21, const float *d_cent,
                              const int *d_list_of_blk64, const unsigned *d_list_max_bits, int *d_qfail, int64_t nlist, unsigned *d_gslot,
                              int nclass, int *d_ctl_hdr, int *d_flag_cnt, hipStream_t st) {
	if (nq <= 0 || max_items1 <= 0)
		return;
	IvfPack2Args a;
	memset(&a, 0, sizeof a);
	a.gslot = d_gslot, a.nclass = nclass, a.ctl_hdr = d_ctl_hdr, a.flag_cnt = d_flag_cnt;
	a.x = d_x, a.d = d, a.nlist = (int)nlist, a.bound_mode = tune().cl_bound_mode, a.nb0 = (unsigned)((nq + 15) / 16), a.nq = nq;
	a.cent = d_cent, a.list_of_blk64 = d_list_of_blk64, a.list_max_bits = d_list_max_bits, a.qfail = d_qfail;
	a.slots0 = d_slots0, a.items0 = (const int4 *)d_items0, a.xi0 = (bf16x8i *)d_xi0, a.igamma0 = d_igamma0, a.ie20 = d_ie20;
	a.items1 = (const int4 *)d_items1, a.nitems1 = d_nitems1, a.qidx1 = d_qidx1, a.xi1 = (bf16x8i *)d_xi1, a.igamma1 = d_igamma1, a.ie21 = d_ie21;
	const size_t lds = ((size_t)64 * (d + 1) + d) * sizeof(float);
	const dim3 grid(a.nb0 + 2u * (unsigned)max_items1);
	if (metric == METRIC_L2) {
		auto kern = ivf_collect_pack2_kernel<true>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
	} else {
		auto kern = ivf_collect_pack2_kernel<false>;
		ensure_dynamic_lds((const void *)kern, lds);
		hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
	}
	MVS_HIP(hipGetLastError());
}
size_t ivf_collect_xi_bytes(int max_items) {
	return (size_t)max_items * 8 * 4 * 64 * 16;
}
// ---- the scan kernel: one wavefront per work item -----------------------------------------------------------------------
// MFMA geometry, LDS layout of a tile, half-tile pipeline and rare path as flat_bf16_collect_kernel (csrc/flat_collect.hip);
// what differs: the wave stages its own tiles (8 LDS-DMA instructions per 32-row tile), the chain starts at beta + gamma, the
// bounds of the item's 128 slots live in an LDS table {B - E, gamma} that the wave refreshes itself.
// NC: row classes per query (16, or 32 for 16 < kk <= 32 -- csrc/flat_collect.hip)
typedef float f32x4a __attribute__((ext_vector_type(4)));
template <int NC>
__global__ __launch_bounds__(64, 2) void ivf_bf16_collect_kernel(const IvfCollectArgs a) {
	constexpr int KB = 4, PITCH = 256, TILE_BYTES = IC_BN * PITCH;
	__shared__ __attribute__((aligned(16))) float smem[(2 * TILE_BYTES + 2 * 64 * 4 + IC_QCAP * 8 + 128 * 8 + 128 * 8 + IC_QCAP) / 4];
	char *tbuf = (char *)smem;                                        // [2][TILE_BYTES]
	float *nbuf = (float *)(tbuf + 2 * TILE_BYTES);                   // [2][64] beta of the tile's rows
	unsigned long long *qbuf = (unsigned long long *)(nbuf + 2 * 64); // [IC_QCAP] hit queue {value bits | row << 32}
	float *ctab = (float *)(qbuf + IC_QCAP);                          // [4 t][16 c][2 i]{B_lower - E, gamma}
	int *qtab = (int *)(ctab + 128 * 2);                              // [128]{query number, E (float bits)} of every slot
	unsigned char *qslot = (unsigned char *)(qtab + 128 * 2);         // [IC_QCAP] slot of every queued hit

	// Items of one list are neighbours in the item table and stream the SAME rows; the dispatcher deals consecutive workgroups
	// round-robin to the eight XCDs, so with item = blockIdx.x every XCD's L2 fetched the list for itself (PMC, C3's main pass:
	// 5.08 GB on the memory side for 2.59 GB of list rows).  XCD j now takes the contiguous item range [j n/8, (j+1) n/8):
	// neighbours run on one XCD at the same time and the second one finds the rows in that L2.
	// (xcd_map = 2, round 4: the SEGMENTS of an item are neighbours too -- a one-dimensional grid, segment fastest inside the XCD's
	// item range -- so the five or so waves of an item fetch its 32 KB of query fragments while they are in that L2, and the next
	// item of the same list walks the same segment a few waves later.  With the segment in blockIdx.y the waves of one item were
	// max_items launches apart: every one of them fetched the fragments from memory.)
	const int nitems = *a.nitems_dev;
	int item = (int)blockIdx.x, seg = (int)blockIdx.y;
	if (a.xcd_map >= 2) {
		// (3: segment 0 of EVERY item first -- every query sees the head of each of its lists before anything else, as with the
		// two-dimensional grid: walked item by item the bounds tighten late and the candidate count grows several times)
		const int per = (nitems + 7) >> 3;
		unsigned b = blockIdx.x;
		int nsg = a.nseg, s0 = 0;
		bool head = false;
		if (a.xcd_map == 3 && a.nseg > 1) {
			head = b < (unsigned)a.gx8;
			if (!head)
				b -= (unsigned)a.gx8, nsg = a.nseg - 1, s0 = 1;
		}
		const int idx = (int)(b >> 3);
		const int local = head ? idx : idx / nsg;
		seg = head ? 0 : s0 + (idx - local * nsg);
		item = (int)(b & 7u) * per + local;
		if (local >= per)
			return;
	} else if (a.xcd_map) {
		const int per = (nitems + 7) >> 3, idx = (int)(blockIdx.x >> 3);
		item = (int)(blockIdx.x & 7u) * per + idx;
		if (idx >= per)
			return;
	}
	if (item >= nitems)
		return;
	const int4 it = a.items[item];
	const int lane = threadIdx.x;
	const int hq = lane >> 4, c = lane & 15;
	const long long r_begin = (long long)it.x + (long long)seg * a.seg_rows;
	if (r_begin >= it.y)
		return;
	const long long r_end = r_begin + a.seg_rows < it.y ? r_begin + a.seg_rows : it.y;
	const int ntiles = (int)((r_end - r_begin + IC_BN - 1) / IC_BN);

	// the lane OWNS (bound refresh) slots 32 hq + 16 i + c, i = 0, 1, i.e. column blocks 2 hq + i = the two blocks of tile t = hq
	int own_q[2];
	float own_e2[2];
	// B fragments, resident: [column block][k-block]
	bf16x8i bq[8][KB];
	{
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int slot = 32 * hq + 16 * i + c;
			own_q[i] = slot < it.w ? a.qidx[it.z + slot] : -1;
			own_e2[i] = 0.5f * a.ie2[(size_t)item * 128 + slot]; // E of THIS (query, list) pair (inflated, with slack)
			qtab[2 * slot] = own_q[i];
			qtab[2 * slot + 1] = __float_as_int(own_e2[i]);
			ctab[((hq * 16 + c) * 2 + i) * 2 + 1] = a.igamma[(size_t)item * 128 + slot];
		}
		const bf16x8i *qsrc = (const bf16x8i *)a.xi + (size_t)item * (8 * 4 * 64);
#pragma unroll
		for (int cb = 0; cb < 8; ++cb)
#pragma unroll
			for (int kb = 0; kb < KB; ++kb) // (column blocks past the item's last slot: zeros, not fetched -- their slots' bounds are NaN)
				bq[cb][kb] = cb * 16 < it.w ? qsrc[(cb * 4 + kb) * 64 + lane] : bf16x8i{0, 0, 0, 0, 0, 0, 0, 0};
	}

	// LDS-DMA staging by this one wave: instruction i of a tile fills LDS bytes [1024 i, +1024) = rows 4 i + (l >> 4), position
	// l & 15, and fetches the row's chunk (l & 15) ^ (row & 15): four loop-invariant lane offsets (i & 3), + 4096 for i >= 4
	unsigned dma_off[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int rr = 4 * i + (lane >> 4);
		dma_off[i] = (unsigned)(rr * PITCH + (((lane & 15) ^ rr) * 16));
	}
	auto dma_tile = [&](int u) {
		const char *base = (const char *)a.yb + (size_t)(r_begin + (long long)u * IC_BN) * PITCH; // uniform
#pragma unroll
		for (int i = 0; i < 8; ++i)
			__builtin_amdgcn_global_load_lds((glb_f32i *)(base + (i >> 2) * 4096 + dma_off[i & 3]),
			                                 (lds_f32i *)(smem + ((u & 1) * TILE_BYTES + i * 1024) / 4), 16, 0, 0);
		const float *bb = a.beta + (r_begin + (long long)u * IC_BN); // uniform
		__builtin_amdgcn_global_load_lds((glb_f32i *)(bb + lane), (lds_f32i *)(smem + (2 * TILE_BYTES) / 4 + (u & 1) * 64), 4, 0, 0);
	};
	if (ntiles > 0)
		dma_tile(0);
	__syncthreads();

	const unsigned rbase = (unsigned)(c * PITCH) | (unsigned)(((hq ^ c) & 15) * 16);
	const unsigned qbuf_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned long long *)qbuf);
	const unsigned qtab_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) int *)qtab);
	const unsigned ct_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)ctab) + (unsigned)(c * 16);

	const unsigned qslot_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char *)qslot);
	// Rare path.  At C3 it is not rare: 149 candidates per query over 32 x 2 441 rows are 1.9e-3 per (row, query) pair, a call covers
	// 16 rows x 32 slots, so six calls in ten have a hit, and a hit used to be worked off then and there by the ONE lane that had it
	// (slot -> query and E from LDS, address arithmetic, class-slot atomic, queue position by an LDS atomic with result) while 63
	// lanes waited: ~150 instructions per call, a third of the kernel (profiles/r4_ivf_scan_rare_path.txt: 0.93 ms with, 0.64 without).
	// Now the loop only RECORDS a hit -- {value, row} and the slot into an LDS queue, position = wave-uniform fill + ballot rank --
	// and drain() works the queue off with one hit per LANE: before every refresh of the wave's bounds (its own evidence is
	// published first), when the queue is nearly full, and at the end.
	int qfill = 0; // entries recorded (all lanes hold the same value); entries past IC_QCAP took the immediate path below
	int qpub = 0;  // ... of which the first qpub have been published to the class slots already
	// publish(): the recorded hits not yet published -> class slots (fire-and-forget atomics: nothing to wait for)
	auto publish = [&]() __attribute__((always_inline)) {
		const unsigned n = (unsigned)qfill < (unsigned)IC_QCAP ? (unsigned)qfill : (unsigned)IC_QCAP;
		for (unsigned e = (unsigned)qpub + lane; e < n; e += 64) {
			unsigned long long ent;
			unsigned sl;
			asm volatile("ds_read_b64 %0, %2\n\tds_read_u8 %1, %3\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(ent), "=&v"(sl)
			             : "v"(qbuf_lds + 8u * e), "v"(qslot_lds + e)
			             : "memory");
			int2 qe;
			asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(qe) : "v"(qtab_lds + sl * 8u) : "memory");
			const float v = __uint_as_float((unsigned)ent);
			const unsigned row = (unsigned)(ent >> 32);
			typedef __attribute__((address_space(1))) unsigned *GU;
			// the slots hold LOWER bounds s - E(list): E differs between the lists a query probes
			__hip_atomic_fetch_min((GU)(a.gslot + (size_t)qe.x * NC) + (row & (unsigned)(NC - 1)), ic_skey(v - __int_as_float(qe.y)),
			                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		qpub = (int)n;
	};
	// drain(): publish what is left, then the whole queue -> the global stream behind ONE reservation (the only wait in here)
	auto drain = [&]() __attribute__((always_inline)) { // (forced: past ~ 10 call sites the inliner leaves a CALL with the closure -- and the kernel's argument block -- in scratch memory)
		publish();
		const unsigned n = (unsigned)qpub;
		qfill = 0;
		qpub = 0;
		if (n == 0u || !a.collect)
			return;
		unsigned long long base = 0ull;
		if (lane == 0) { // (by hand: no compiled atomic with a result in the loop)
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
			unsigned sl;
			asm volatile("ds_read_b64 %0, %2\n\tds_read_u8 %1, %3\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(ent), "=&v"(sl)
			             : "v"(qbuf_lds + 8u * e), "v"(qslot_lds + e)
			             : "memory");
			int2 qe;
			asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(qe) : "v"(qtab_lds + sl * 8u) : "memory");
			if ((long long)(base + e) < a.stream_cap) {
				typedef __attribute__((address_space(1))) unsigned long long *GUL;
				*((GUL)a.stream + (base + e)) = ((unsigned long long)(unsigned)qe.x << 32) | (unsigned)(ent >> 32);
				if (a.stream_u) { // s + E(this pair), rounded up: exact <= s + E (ivf_final_bound_kernel / ivf_refilter_kernel)
					const float ub = __uint_as_float((unsigned)ent) + __int_as_float(qe.y);
					a.stream_u[base + e] = fmaf(fabsf(ub), 1.1920929e-07f, ub);
				}
			}
		}
	};
	auto rare = [&](const f32x4a (&sv)[2], int rb, int t, bool any_t, f32x4i cg, long long row0, int nvalid, unsigned rowbits) {
#ifdef MVS_PROFILING
		if (a.abl & 1) { // (profiling library only, option ivf_cl_abl: no rare path -- results are wrong)
			asm volatile("" ::"v"(any_t));
			return;
		}
#endif
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(any_t) == 0ull, 1)) // (hot path = fall-through: no taken branch per half tile)
			return;
		unsigned m0 = 0u, m1 = 0u;
		if (any_t) {
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const bool in = 16 * rb + 4 * hq + r < nvalid;
				m0 |= (in && sv[0][r] >= cg[0]) ? 1u << r : 0u;
				m1 |= (in && sv[1][r] >= cg[2]) ? 1u << r : 0u;
			}
			const unsigned rbits = (rowbits >> (16 * rb + 4 * hq)) & 15u; // rows the IDSelector rejects: no candidate, no evidence for the bound
			m0 &= rbits;
			m1 &= rbits;
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
			const unsigned sl = (unsigned)(32 * t + 16 * i + c);
			const unsigned long long act = __builtin_amdgcn_ballot_w64(has);
			const int nact = (int)__builtin_popcountll(act);
			// (round 5: a queue that cannot take this step's hits is drained THEN AND THERE -- one reservation per <= 160 hits.  Rounds
			// 3-4 sent every hit beyond the queue straight to the stream, one returning atomic + wait each: a list whose own queries all
			// pass most of its rows -- a giant list covering several clusters, E of its pairs is large -- took 1.3 ms in that branch
			// while the rest of the launch had long finished: profiles/r5_c3_ab.txt, 6)
			if (__builtin_expect(qfill + nact > IC_QCAP, 0)) {
				if (a.collect && lane == 0) // census: forced drains (control block header, byte 128; mvs_index_ivf_probe_stats)
					__hip_atomic_fetch_add((unsigned *)(a.stream_cnt + 16), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				drain();
			}
			const unsigned pos = (unsigned)qfill + __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
			qfill = __builtin_amdgcn_readfirstlane(qfill + nact);
			if (has) {
				const unsigned long long ent = ((unsigned long long)row << 32) | __float_as_uint(v);
				asm volatile("ds_write_b64 %0, %1\n\tds_write_b8 %2, %3" ::"v"(qbuf_lds + 8u * pos), "v"(ent), "v"(qslot_lds + pos), "v"(sl) : "memory");
			}
		}
	};

	for (int u = 0; u < ntiles; ++u) {
		// bounds of the 128 slots: B = the kk-th best of the query's 16 class bests (bitonic network) -- each a LOWER bound
		// s - E(its list) of a distinct row's exact value, so the exact kk-th best is >= B; a row of the result in THIS list has
		// s >= exact - E >= B - E: table entry = B - E (E of this item's list).  With one E for all lists this is the B - 2E of
		// csrc/flat_collect.hip.
		const int period = a.refresh > 0 ? a.refresh : (u < 4 ? 1 : (u < 32 ? 4 : 16));
		if (qfill > IC_QCAP / 2)
			drain(); // (a queue more than half full is emptied)
		if (a.bfix != nullptr) { // frozen bounds: set once, nothing derived, nothing published that anyone reads
			if (u == 0) {
#pragma unroll
				for (int i = 0; i < 2; ++i) {
					const float B = own_q[i] >= 0 ? a.bfix[own_q[i]] : 0.f;
					ctab[((hq * 16 + c) * 2 + i) * 2 + 0] = own_q[i] >= 0 ? B - own_e2[i] : __uint_as_float(0x7fc00000u);
				}
			}
		} else if ((u % period) == 0) {
			publish(); // (this wave's own evidence is in the class slots before it reads them)
			// (NC = 16: both queries' slots in one round trip; NC = 32: one query at a time -- 32 keys + the network's temporaries)
			unsigned long long w[NC == 16 ? 2 : 1][NC / 2];
			auto fetch = [&](int i, int wi) {
				const int qc = own_q[i] >= 0 ? own_q[i] : 0;
				const unsigned long long *src = (const unsigned long long *)(a.gslot + (size_t)qc * NC);
#pragma unroll
				for (int j = 0; j < NC / 2; ++j)
					w[wi][j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			};
			if (NC == 16) {
				fetch(0, 0);
				fetch(1, NC == 16 ? 1 : 0);
#pragma unroll
				for (int i = 0; i < (NC == 16 ? 2 : 1); ++i)
#pragma unroll
					for (int j = 0; j < NC / 2; ++j)
						asm volatile("" : "+v"(w[i][j]));
			}
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const int wi = NC == 16 ? i : 0;
				if (NC != 16) {
					fetch(i, 0);
#pragma unroll
					for (int j = 0; j < NC / 2; ++j)
						asm volatile("" : "+v"(w[0][j]));
				}
				unsigned key[NC];
#pragma unroll
				for (int j = 0; j < NC / 2; ++j) {
					key[2 * j] = (unsigned)w[wi][j];
					key[2 * j + 1] = (unsigned)(w[wi][j] >> 32);
				}
#pragma unroll
				for (int kbit = 2; kbit <= NC; kbit <<= 1)
#pragma unroll
					for (int jb = kbit >> 1; jb > 0; jb >>= 1)
#pragma unroll
						for (int x0 = 0; x0 < NC; ++x0) {
							const int x1 = x0 ^ jb;
							if (x1 > x0) {
								const unsigned lo = key[x0] < key[x1] ? key[x0] : key[x1];
								const unsigned hi = key[x0] < key[x1] ? key[x1] : key[x0];
								const bool asc = (x0 & kbit) == 0;
								key[x0] = asc ? lo : hi;
								key[x1] = asc ? hi : lo;
							}
						}
				unsigned kth = key[0];
#pragma unroll
				for (int j = 1; j < NC; ++j)
					kth = (a.kk - 1 == j) ? key[j] : kth;
				const unsigned neutral = ic_skey(-FLT_MAX);
				const float B = ic_skey2f(kth < neutral ? kth : neutral); // -FLT_MAX while fewer than kk classes are set
				const float v = own_q[i] >= 0 ? B - own_e2[i] : __uint_as_float(0x7fc00000u); // NaN: nothing passes
				ctab[((hq * 16 + c) * 2 + i) * 2 + 0] = v;
			}
		}
		bf16x8i A[KB][2];
		f32x4i Y[2];
		{
			const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32i *)(nbuf + (u & 1) * 64 + 4 * hq));
			asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(Y[0]), "=&v"(Y[1]) : "v"(nb_lds) : "memory");
			const unsigned ab = (unsigned)(uintptr_t)((lds_f32i *)(smem + ((u & 1) * TILE_BYTES) / 4)) + rbase;
#pragma unroll
			for (int kb = 0; kb < KB; ++kb) {
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[kb][0]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
				asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(A[kb][1]) : "v"(ab ^ (unsigned)(kb * 64)) : "memory");
			}
		}
		dma_tile(u + 1);
		const long long row0 = r_begin + (long long)u * IC_BN;
		const int nvalid = (int)((r_end - row0) < IC_BN ? (r_end - row0) : IC_BN);
		unsigned rowbits = 0xFFFFFFFFu;
		if (a.rowmask) { // this tile's 32 selector bits: one wave-uniform (scalar) load (tiles start at multiples of 32 rows)
			typedef __attribute__((address_space(4))) const unsigned cuint;
			rowbits = *((cuint *)a.rowmask + (row0 >> 5));
		}

		f32x4a acc[2][2]; // [row block][column block of the tile]
		f32x4i cg[2];     // {B - E, gamma} x 2 column blocks of tile t in cg[t & 1]
		float mx0 = -INFINITY, mx1 = -INFINITY;
		auto fold = [&](const f32x4a &p, int i) {
			if (i == 0)
				mx0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx0, p[0]), p[1]), p[2]), p[3]);
			else
				mx1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(mx1, p[0]), p[1]), p[2]), p[3]);
		};
		auto any_of = [&](f32x4i cgp) { // NaN on either side: false
			const bool r = (mx0 >= cgp[0]) || (mx1 >= cgp[2]);
			mx0 = -INFINITY;
			mx1 = -INFINITY;
			return r;
		};
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			asm volatile("ds_read_b128 %0, %1" : "=v"(cg[t & 1]) : "v"(ct_lds + (unsigned)(t * 256)) : "memory");
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				const int prb = rb ^ 1, pt = rb == 0 ? t - 1 : t; // the half folded under this one
				if (rb == 0) // the tile's table entry (and at t = 0: beta, every fragment) has arrived
					asm volatile("s_waitcnt lgkmcnt(0)"
					             : "+v"(cg[0]), "+v"(cg[1]), "+v"(Y[0]), "+v"(Y[1]), "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]),
					               "+v"(A[1][1]), "+v"(A[2][0]), "+v"(A[2][1]), "+v"(A[3][0]), "+v"(A[3][1]));
				// the chain starts at beta(row) + gamma(slot): s = 2 <x', y'> - ||y'||^2 - ||x'||^2 comes out of the matrix pipe
				f32x4a c0v = Y[rb] + cg[t & 1][1], c1v = Y[rb] + cg[t & 1][3];
#pragma unroll
				for (int kb = 0; kb < KB; ++kb) {
					acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + 0][kb], kb == 0 ? c0v : acc[rb][0], 0, 0, 0);
					acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb][rb], bq[2 * t + 1][kb], kb == 0 ? c1v : acc[rb][1], 0, 0, 0);
					if (pt >= 0 && kb < 2)
						fold(acc[prb][kb], kb);
					__builtin_amdgcn_sched_barrier(0);
				}
				if (pt >= 0)
					rare(acc[prb], prb, pt, any_of(cg[pt & 1]), cg[pt & 1], row0, nvalid, rowbits);
			}
		}
		fold(acc[1][0], 0);
		fold(acc[1][1], 1);
		rare(acc[1], 1, 3, any_of(cg[1]), cg[1], row0, nvalid, rowbits);
		__syncthreads(); // one wave: drains this tile's LDS-DMA (vmcnt(0)) before the next tile reads it
		if (u == ntiles - 1)
			drain();
	}
}

void launch_ivf_collect_scan(const void *d_items, const int *d_nitems, int max_items, const int *d_qidx, const void *d_xi,
                             const float *d_igamma, const float *d_ie2, const unsigned short *d_rows_bf, const float *d_beta,
                             unsigned *d_gslot, unsigned long long *d_stream, unsigned long long *d_stream_cnt,
                             int64_t stream_cap, int kk, int seg_rows, int nseg, int collect, const unsigned *d_rowmask,
                             hipStream_t st, float *d_stream_u, const float *d_bfix) {
	if (max_items <= 0 || nseg <= 0)
		return;
	IvfCollectArgs a;
	memset(&a, 0, sizeof a);
	a.bfix = d_bfix;
	if (d_bfix)
		kk = 16; // (frozen bounds: the 16-class instance, its slots unused)
	a.items = (const int4 *)d_items;
	a.nitems_dev = d_nitems;
	a.qidx = d_qidx;
	a.xi = d_xi;
	a.igamma = d_igamma;
	a.ie2 = d_ie2;
	a.yb = d_rows_bf;
	a.beta = d_beta;
	a.gslot = d_gslot;
	a.stream = d_stream;
	a.stream_cnt = d_stream_cnt;
	a.stream_u = d_stream_u;
	a.stream_cap = stream_cap;
	a.kk = kk;
	a.seg_rows = seg_rows;
	a.collect = collect;
	a.refresh = tune().ivf_cl_refresh;
	a.rowmask = d_rowmask;
	a.xcd_map = (tune().ivf_cl_xcd && max_items >= 64) ? tune().ivf_cl_xcd : 0; // (the Flat small-batch path has one or two items: nothing to place)
	a.nseg = nseg;
	a.abl = tune().ivf_cl_abl;
	unsigned gx = (unsigned)max_items + (a.xcd_map ? 8u : 0u);
	if (a.xcd_map >= 2) {
		gx = (gx + 7u) & ~7u;
		if ((uint64_t)gx * (uint64_t)nseg >= ((uint64_t)1 << 31))
			a.xcd_map = 1;
	}
	a.gx8 = (int)gx;
	const dim3 grid = a.xcd_map >= 2 ? dim3(gx * (unsigned)nseg) : dim3(gx, nseg);
	if (kk > 16) // 32 row classes: the caller sized and initialised 32 slots per query (ivf_collect_slot_stride)
		hipLaunchKernelGGL(ivf_bf16_collect_kernel<32>, grid, dim3(64), (size_t)tune().ivf_cl_lds_pad, st, a);
	else
		hipLaunchKernelGGL(ivf_bf16_collect_kernel<16>, grid, dim3(64), (size_t)tune().ivf_cl_lds_pad, st, a);
	MVS_HIP(hipGetLastError());
}

// ---- round 5: stream -> exact values INTO PER-QUERY BUCKETS -> the kk best -> the search's output (csrc/collect_bucket.h) -----
// Kernel A (ivf_exact_bucket_kernel) = ivf_collect_exact_kernel on the UNSORTED stream: the lane's entry (q << 32 | padded row)
// is re-scored with the scanner's arithmetic as before, and the key (order-preserving value key << 32 | position in the
// list-sorted store) goes to entry bcount[q]++ of bucket[q][bpitch].  The atomic that hands out the position is issued BEFORE the
// rows are staged and its result is used after the 128-step chain: its round trip costs nothing.  (Taking the position in the
// SCAN's drain instead -- measured: + 60 us on the scan, whose drain then waits for 64 round trips instead of one.)
// Kernel B (ivf_bucket_select_kernel): one wavefront per query selects the kk best of its bucket and writes
//   pd / pi [nq][kk]   the pure list (value, position | label), what collect_select_kernel + ivf_emit_sorted_kernel produced, and
//   D / I [nq][k]      (inside the exact-tie wrapper) FAISS's print order + the boundary-tie flags: csrc/ivf_ties.hip
//                      ivf_finish_kernel's rule applied to the list while it is still in registers;
// it also compacts the per-query fail flags (ivf_compact_flags_kernel's job) and sums the bucket statistics.
struct EbMeta {
	int q, pos;
	unsigned slot;
};
// stream entry of lane `lane` of the group at `base`: query, bucket slot (the atomic's result is used behind the chains), position
__device__ __forceinline__ EbMeta eb_load_meta(const unsigned long long *__restrict__ strm, long long base, int lane, long long ncand,
                                               unsigned *__restrict__ bcount, const int *__restrict__ perm) {
	EbMeta m;
	const long long i = base + lane;
	const bool in = i < ncand;
	const unsigned long long ent = in ? strm[i] : 0ull;
	m.q = (int)(ent >> 32); // (query numbers fit 31 bits: nq * nprobe < 2^26)
	m.slot = 0xffffffffu;
	if (in)
		m.slot = __hip_atomic_fetch_add(bcount + m.q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	m.pos = in ? perm[(unsigned)ent] : 0;
	return m;
}
// one half of a group: its 32 database rows and 32 query rows from registers into LDS, the 32 chains on lanes 0 .. 31, every
// lane gets the value of lane (lane & 31).  (No barrier: one wavefront per workgroup, a wave's LDS operations execute in order.)
// ARITH: 0 = the scanner's L2 (t = x_k - y_k, acc = fmaf(t, t, acc)), 1 = the inner-product chain, 2 = the chain for FAISS's BLAS-branch
// L2 formula (Flat shadow, csrc/index.hip: the caller turns it into max(0, (xn + yn) - 2 ip))
template <int ARITH>
__device__ __forceinline__ float eb_half(float *yrows, float *xrows, const f32x4i (&ry)[16], const f32x4i (&rx)[16], int lane, int sub,
                                         int ch) {
	// (f32x4i, the native vector type: whole-value copies of HIP's float4 STRUCT become memcpy calls that keep the buffers in
	// scratch memory -- every load then waits for its own store)
#pragma unroll
	for (int it = 0; it < 16; ++it) {
		const int r = 2 * it + sub;
		*(f32x4i *)(yrows + r * 132 + ch * 4) = ry[it];
		*(f32x4i *)(xrows + r * 132 + ch * 4) = rx[it];
	}
	asm volatile("" ::: "memory");
	float a2 = 0.f;
	if (sub == 0) {
		const float *y = yrows + lane * 132, *xx = xrows + lane * 132;
#pragma unroll
		for (int c4 = 0; c4 < 32; ++c4) {
			const float4 xv = *(const float4 *)(xx + c4 * 4);
			const float4 yv = *(const float4 *)(y + c4 * 4);
			const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				if (ARITH == 0) {
					const float t = __fsub_rn(xs[e], ys[e]);
					a2 = fmaf(t, t, a2);
				} else {
					a2 = fmaf(xs[e], ys[e], a2);
				}
			}
		}
	}
	const float got = __shfl(a2, lane & 31);
	asm volatile("" ::: "memory");
	return got;
}
template <int ARITH>
__device__ __forceinline__ unsigned long long eb_key(float acc, int pos, long long q, const IvfFlatArith &fa) {
	constexpr bool L2KEY = ARITH != 1;
	float v = acc;
	unsigned low = (unsigned)pos;
	if (ARITH == 2 && pos >= 0) {
		v = fmaf(-2.0f, acc, fa.qn[q] + fa.yn[pos]);
		v = v < 0.f ? 0.f : v; // FAISS: if (dis < 0) dis = 0
		low = fa.rowids ? (unsigned)fa.rowids[pos] : (unsigned)pos;
	}
	const bool ok = pos >= 0 && (L2KEY ? v < FLT_MAX : v > -FLT_MAX);
	return ok ? (((unsigned long long)bkey<L2KEY>(v) << 32) | low) : CB_EMPTY;
}
template <int ARITH>
__global__ __launch_bounds__(64) void ivf_exact_bucket_kernel(const unsigned long long *__restrict__ strm, long long ncand,
                                                             const unsigned long long *__restrict__ cnt,
                                                             const float *__restrict__ x, int d,
                                                             const float *__restrict__ rows_csr, int dp,
                                                             const int *__restrict__ perm, unsigned long long *__restrict__ bucket,
                                                             unsigned *__restrict__ bcount, int bpitch, const IvfFlatArith fa) {
	// d = dp = 128: the 64 entries of a group in two halves of 32; a half's 32 database rows AND its 32 query rows are staged
	// through LDS with coalesced 512-byte loads (the stream is not sorted by query: a lane loading ITS query on its own touches 64
	// cache lines per instruction), lanes 0 .. 31 run the chains of the half.  Both halves' 64 KB are requested at once into
	// registers, the next group's stream entries, positions and bucket slots a whole group ahead.  (One wavefront per workgroup:
	// LDS traffic of a wave is ordered, no barrier -- __syncthreads() would also wait for the loads in flight.)  Other shapes: rows
	// staged, the query read per lane, as ivf_collect_exact_kernel.
	__shared__ __attribute__((aligned(16))) float rows[64 * (128 + 4)];
	__shared__ int mpos[64], mq[64];
	const int pitch = dp + 4, cpr = dp / 4; // floats per LDS row (bank spread), float4 chunks per row
	const int lane = threadIdx.x;
	{
		const unsigned long long have = *cnt;
		ncand = have < (unsigned long long)ncand ? (long long)have : ncand;
	}
	const long long stride = (long long)gridDim.x * 64;
	if (d == 128 && dp == 128) {
		float *yrows = rows, *xrows = rows + 32 * 132;
		const int sub = lane >> 5, ch = lane & 31;
		long long i0 = (long long)blockIdx.x * 64;
		if (i0 >= ncand)
			return;
		EbMeta cur = eb_load_meta(strm, i0, lane, ncand, bcount, perm);
		while (i0 < ncand) {
			const long long i1 = i0 + stride;
			// BOTH halves' rows and queries are requested at once -- 64 KB in flight per wave, in registers (a wave alone on its SIMD
			// owns 512) -- and the next group's stream entries, positions and bucket slots behind them: one gather round trip per group
			f32x4i ry0[16], rx0[16], ry1[16], rx1[16];
			// (positions and queries of the other lanes through LDS, not by __shfl: a loop around a cross-lane builtin is unrolled
			// too late for the register buffers to be split into registers -- they landed in scratch memory, every load waited for)
			mpos[lane] = cur.pos;
			mq[lane] = cur.q;
			asm volatile("" ::: "memory");
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int r = 2 * it + sub;
				const int pp = mpos[r], qq = mq[r];
				ry0[it] = *(const f32x4i *)(rows_csr + (size_t)(pp < 0 ? 0 : pp) * 128 + ch * 4);
				rx0[it] = *(const f32x4i *)(x + (size_t)qq * 128 + ch * 4);
			}
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int r = 32 + 2 * it + sub;
				const int pp = mpos[r], qq = mq[r];
				ry1[it] = *(const f32x4i *)(rows_csr + (size_t)(pp < 0 ? 0 : pp) * 128 + ch * 4);
				rx1[it] = *(const f32x4i *)(x + (size_t)qq * 128 + ch * 4);
			}
			const EbMeta nxt = eb_load_meta(strm, i1, lane, ncand, bcount, perm); // (past the end: q = 0, pos = 0, no slot)
			// (the half's 32 results sit in lanes 0 .. 31) lane 32 h + j takes the value lane j computed in half h
			const float got0 = eb_half<ARITH>(yrows, xrows, ry0, rx0, lane, sub, ch);
			const float got1 = eb_half<ARITH>(yrows, xrows, ry1, rx1, lane, sub, ch);
			const float acc = sub == 0 ? got0 : got1;
			if (i0 + lane < ncand) {
				if (cur.slot < (unsigned)bpitch) // (entries past the bucket are only counted: the host grows the pitch and repeats the pass)
					bucket[(size_t)cur.q * (size_t)bpitch + cur.slot] = eb_key<ARITH>(acc, cur.pos, cur.q, fa);
			}
			cur = nxt;
			i0 = i1;
		}
		return;
	}
	for (long long i0 = (long long)blockIdx.x * 64; i0 < ncand; i0 += stride) {
		const long long i = i0 + lane;
		const unsigned long long ent = i < ncand ? strm[i] : 0ull;
		const long long q = (long long)(ent >> 32);
		unsigned slot = 0xffffffffu;
		if (i < ncand)
			slot = __hip_atomic_fetch_add(bcount + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (used behind the chain)
		const int pos = i < ncand ? perm[(unsigned)ent] : 0;
		float acc = 0.f;
		const float *xq = x + q * d;
		for (int it = 0; it < cpr; ++it) { // 64 consecutive float4 of the 64 x cpr block per step
			const int idx = it * 64 + lane, r = idx / cpr, ch = idx - r * cpr;
			const int pp = __shfl(pos, r);
			const float4 v = *(const float4 *)(rows_csr + (size_t)(pp < 0 ? 0 : pp) * dp + ch * 4);
			*(float4 *)(rows + r * pitch + ch * 4) = v;
		}
		__syncthreads();
		if (i < ncand) {
			const float *y = rows + lane * pitch;
			int kd = 0;
			if ((d & 3) == 0) { // 16-byte loads of the query and of the staged row; the chain keeps its k order
				for (; kd < d; kd += 4) {
					const float4 xv = *(const float4 *)(xq + kd);
					const float4 yv = *(const float4 *)(y + kd);
					const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						if (ARITH == 0) {
							const float t = __fsub_rn(xs[e], ys[e]);
							acc = fmaf(t, t, acc);
						} else {
							acc = fmaf(xs[e], ys[e], acc);
						}
					}
				}
			}
			for (; kd < d; ++kd) {
				if (ARITH == 0) {
					const float t = __fsub_rn(xq[kd], y[kd]);
					acc = fmaf(t, t, acc);
				} else {
					acc = fmaf(xq[kd], y[kd], acc); // fvec_inner_product: the k-ordered chain
				}
			}
			if (slot < (unsigned)bpitch)
				bucket[(size_t)q * (size_t)bpitch + slot] = eb_key<ARITH>(acc, pos, q, fa);
		}
		__syncthreads(); // (the next group's rows overwrite the tile)
	}
}
struct IvfBucketSelectArgs {
	const unsigned long long *bucket;
	unsigned *bcount; // [nq]
	int bpitch;
	long long nq;
	int kk;
	float *pd;
	long long *pi;
	const long long *rowids; // labels of the pure list (nullptr: the keys' low words as they are), then through idmap if given
	const long long *idmap;
	long long label_offset;  // ... or + this (Flat shadow: label_offset of the Flat index)
	int k; // fin (D != nullptr): k < kk
	float *D;
	long long *I;
	const long long *fin_rowids;
	const long long *fin_idmap;
	int *flag_cnt;
	int *flag_q;
	unsigned long long *stats; // [0] sum of the bucket counts, [1] the largest
	int *qfail;                // per-query fail flags -> fail_q[fail_cnt++]
	int *fail_cnt;
	int *fail_q;
	int reset; // leave bcount[q] and qfail[q] zero for the next search (prep2: no memset in front of a search)
	const int *kept_blk;          // (may be null) per-workgroup survivor counts of the scatter kernel, nkept_blk of them
	int nkept_blk;
	unsigned long long *kept_out; // ... their sum goes here (the query-0 wavefront adds them up)
	IpFlatEmit ipf;               // (D != nullptr) the Flat index's inner-product output: see index.h
};
template <bool IS_L2>
__global__ __launch_bounds__(64) void ivf_bucket_select_kernel(const IvfBucketSelectArgs a) {
	__shared__ unsigned long long surv[256];
	__shared__ unsigned long long top[64];
	__shared__ float fv[64];
	__shared__ long long fid[64];
	__shared__ int fp[64];
	const int lane = threadIdx.x, kk = a.kk;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	const long long q = blockIdx.x;
	const unsigned have = a.bcount[q];
	const int n = (int)(have < (unsigned)a.bpitch ? have : (unsigned)a.bpitch);
	if (q == 0 && a.kept_blk) { // the scatter kernel's per-workgroup survivor counts -> one number for the host
		int t = 0;
		for (int i = lane; i < a.nkept_blk; i += 64)
			t += a.kept_blk[i];
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
			t += __shfl_xor(t, off);
		if (lane == 0)
			*a.kept_out = (unsigned long long)t;
	}
	if (lane == 0) {
		// (one atomic per QUERY on one address -- sum and maximum of the counts -- serialised in L2: 250 us for 10 000 queries.  The
		// sum is the stream's own count; the maximum matters only when a bucket was too small)
		if (have > (unsigned)a.bpitch)
			atomicMax(a.stats + 1, (unsigned long long)have);
		if (a.qfail && a.qfail[q]) // (null: the caller keeps its own fail list -- the Flat index's bucketed finish)
			a.fail_q[atomicAdd(a.fail_cnt, 1)] = (int)q;
		if (a.reset) {
			a.bcount[q] = 0u;
			if (a.qfail)
				a.qfail[q] = 0;
		}
	}
	const unsigned long long mine = n > 0 ? cb_select_wave<false>(a.bucket + (size_t)q * (size_t)a.bpitch, n, kk, lane, surv, top) : CB_EMPTY;
	const bool hv = mine != CB_EMPTY;
	const float val = hv ? bkey2f<IS_L2>((unsigned)(mine >> 32)) : neutral;
	const int ps = hv ? (int)(unsigned)mine : -1;
	if (a.ipf.D) { // Flat inner product (csrc/util_kernels.hip merge_partials_kernel<false>'s print rule and tie flags; entry j in lane j)
		const int kout = a.ipf.kout;
		fv[lane] = val;
		fp[lane] = lane < kk ? ps : -1;
		__syncthreads();
		if (lane < kout) {
			int src = lane;
			if (fp[lane] >= 0) { // a run of equal scores is printed with the larger row first
				int lo = lane, hi = lane;
				while (lo > 0 && fp[lo - 1] >= 0 && fv[lo - 1] == val)
					--lo;
				while (hi + 1 < kout && fp[hi + 1] >= 0 && fv[hi + 1] == val)
					++hi;
				src = lo + (hi - lane);
			}
			const int id = fp[src];
			a.ipf.D[q * kout + lane] = fv[src];
			a.ipf.I[q * kout + lane] = id < 0 ? -1ll : (a.ipf.idmap ? a.ipf.idmap[id] : (long long)id + a.ipf.label_offset);
		}
		if (a.ipf.flags.count && kout < kk && fp[kout] >= 0 && fv[kout] == fv[kout - 1]) {
			int slot = 0;
			if (lane == 0) {
				slot = atomicAdd(a.ipf.flags.count, 1);
				a.ipf.flags.query[slot] = (int)q;
			}
			slot = __shfl(slot, 0);
			if (lane < kk) {
				a.ipf.flags.val[(size_t)slot * kk + lane] = fv[lane];
				a.ipf.flags.row[(size_t)slot * kk + lane] = fp[lane];
			}
		}
	}
	if (a.pd && lane < kk) {
		a.pd[q * kk + lane] = val;
		long long lab = ps;
		if (ps >= 0) {
			if (a.rowids)
				lab = a.rowids[ps];
			lab = a.idmap ? a.idmap[lab] : lab + a.label_offset;
		}
		a.pi[q * kk + lane] = lab;
	}
	if (a.D) { // csrc/ivf_ties.hip ivf_finish_kernel, entry j of the pure list in lane j
		const int k = a.k;
		fv[lane] = val;
		fp[lane] = lane < kk ? ps : -1;
		fid[lane] = (lane < kk && ps >= 0) ? a.fin_rowids[ps] : -1;
		__syncthreads();
		if (lane < k) {
			const int j = lane;
			if (fp[j] < 0) {
				a.D[q * k + j] = neutral;
				a.I[q * k + j] = -1;
			} else {
				const long long id = fid[j];
				int lo = j, hi = j;
				while (lo > 0 && fv[lo - 1] == val)
					--lo;
				while (hi + 1 < k && fp[hi + 1] >= 0 && fv[hi + 1] == val)
					++hi;
				int rank = 0;
				for (int m = lo; m <= hi && hi > lo; ++m) {
					const long long idm = fid[m];
					rank += IS_L2 ? (idm < id || (idm == id && m < j)) : (idm > id || (idm == id && m < j));
				}
				const long long o = q * k + lo + rank;
				a.D[o] = val;
				a.I[o] = a.fin_idmap ? a.fin_idmap[id] : id;
				if (j == k - 1 && kk > k && fp[k] >= 0 && fv[k] == val)
					a.flag_q[atomicAdd(a.flag_cnt, 1)] = (int)q;
			}
		}
	}
}
// ---- final-bound filter (round 5) ---------------------------------------------------------------------------------------------
// The scan admits a row when its coarse value passes the bound of THAT MOMENT: s >= B_now - E.  Bounds only tighten, so most of the
// ~ 134 candidates per query at C3 were admitted early and would not pass the bound the scan ENDS with.  The stream therefore carries
// u = s + E per entry (an upper bound of the row's exact value, rounded up), and before the exact stage
//   ivf_final_bound_kernel  Bf[q] = the kf-th best of the query's class slots at the end of the scan: kf distinct rows have exact
//                           values >= Bf (every slot is a LOWER bound s' - E' of a row's exact value), so the kf-th best exact value
//                           of the result is >= Bf;  -FLT_MAX while fewer than kf classes are set
//   ivf_refilter_kernel     keeps an entry iff u >= Bf[q] (a row of the result has u >= exact >= Bf; ties included; NaN: kept) and
//                           writes the survivors, compacted per workgroup (one reservation each), to a second stream
// -- the same argument as the scan's own test with the last bound instead of the running one.  The exact stage reads 4-5 x fewer
// rows (profiles/r5_c3_ab.txt, 7).
template <int NC>
__global__ __launch_bounds__(64) void ivf_final_bound_kernel(const unsigned *__restrict__ gslot, int kf, long long nq, float *__restrict__ bf) {
	const long long q = (long long)blockIdx.x * 64 + threadIdx.x;
	if (q >= nq)
		return;
	unsigned key[NC];
	const uint4 *src = (const uint4 *)(gslot + q * NC); // (the scan has finished: plain loads)
#pragma unroll
	for (int j = 0; j < NC / 4; ++j) {
		const uint4 v = src[j];
		key[4 * j] = v.x, key[4 * j + 1] = v.y, key[4 * j + 2] = v.z, key[4 * j + 3] = v.w;
	}
	// the kf-th smallest key (keys: smaller = better) by rank counting; equal keys are ordered by position
	unsigned kth = 0xffffffffu;
#pragma unroll
	for (int j = 0; j < NC; ++j) {
		int r = 0;
#pragma unroll
		for (int i = 0; i < NC; ++i)
			r += (key[i] < key[j] || (key[i] == key[j] && i < j)) ? 1 : 0;
		kth = (r == kf - 1) ? key[j] : kth;
	}
	const unsigned neutral = ic_skey(-FLT_MAX);
	bf[q] = ic_skey2f(kth < neutral ? kth : neutral);
}
constexpr int RF_PER = 2; // entries per thread (a latency-bound gather of Bf[q]: many small workgroups)
__global__ __launch_bounds__(256) void ivf_refilter_kernel(const unsigned long long *__restrict__ strm, const float *__restrict__ su,
                                                          long long cap, const unsigned long long *__restrict__ cnt,
                                                          const float *__restrict__ bf, unsigned long long *__restrict__ out,
                                                          unsigned long long *__restrict__ out_cnt) {
	__shared__ int wsum[4];
	__shared__ unsigned long long gbase;
	const unsigned long long have = *cnt;
	const long long n = have < (unsigned long long)cap ? (long long)have : cap;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (long long b0 = (long long)blockIdx.x * (256 * RF_PER); b0 < n; b0 += (long long)gridDim.x * (256 * RF_PER)) {
		unsigned long long ent[RF_PER];
		bool keep[RF_PER];
		int mine = 0;
#pragma unroll
		for (int e = 0; e < RF_PER; ++e) {
			const long long i = b0 + e * 256 + threadIdx.x;
			ent[e] = i < n ? strm[i] : 0ull;
			const float u = i < n ? su[i] : 0.f;
			const float b = bf[i < n ? (unsigned)(ent[e] >> 32) : 0u];
			keep[e] = i < n && !(u < b);
			mine += keep[e] ? 1 : 0;
		}
		// exclusive prefix over the workgroup: lanes by DPP-free shuffles, waves through LDS, ONE global reservation per workgroup
		int inc = mine;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const int v = __shfl_up(inc, off);
			inc += lane >= off ? v : 0;
		}
		if (lane == 63)
			wsum[wave] = inc;
		__syncthreads();
		int wbase = 0, total = 0;
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			wbase += w < wave ? wsum[w] : 0;
			total += wsum[w];
		}
		if (threadIdx.x == 0)
			gbase = total ? atomicAdd(out_cnt, (unsigned long long)total) : 0ull;
		__syncthreads();
		unsigned long long pos = gbase + (unsigned long long)(wbase + inc - mine);
#pragma unroll
		for (int e = 0; e < RF_PER; ++e)
			if (keep[e])
				out[pos++] = ent[e];
		__syncthreads(); // (wsum / gbase are rewritten by the next round)
	}
}
// the compaction alone, thresholds given (the Flat index's wide stores: csrc/index.hip collect_candidates)
void launch_stream_refilter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt, const float *d_thr,
                            unsigned long long *d_out, unsigned long long *d_out_cnt, hipStream_t st) {
	if (cap <= 0)
		return;
	const unsigned blocks = (unsigned)std::min<int64_t>((cap + 256 * RF_PER - 1) / (256 * RF_PER), 8192);
	hipLaunchKernelGGL(ivf_refilter_kernel, dim3(blocks), dim3(256), 0, st, d_strm, d_su, (long long)cap, d_cnt, d_thr, d_out, d_out_cnt);
	MVS_HIP(hipGetLastError());
}
void launch_ivf_refilter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt,
                         const unsigned *d_gslot, int nclass, int kf, int64_t nq, float *d_bf, unsigned long long *d_out,
                         unsigned long long *d_out_cnt, hipStream_t st) {
	if (nq <= 0 || cap <= 0)
		return;
	if (nclass == 32)
		hipLaunchKernelGGL(ivf_final_bound_kernel<32>, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, d_gslot, kf, (long long)nq, d_bf);
	else
		hipLaunchKernelGGL(ivf_final_bound_kernel<16>, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, d_gslot, kf, (long long)nq, d_bf);
	const unsigned blocks = (unsigned)std::min<int64_t>((cap + 256 * RF_PER - 1) / (256 * RF_PER), 8192);
	hipLaunchKernelGGL(ivf_refilter_kernel, dim3(blocks), dim3(256), 0, st, d_strm, d_su, (long long)cap, d_cnt, d_bf, d_out, d_out_cnt);
	MVS_HIP(hipGetLastError());
}

// ---- bucketed exact stage (round 5, second cut): the survivors of the final-bound test go to their QUERY's bucket first, as (padded)
// row numbers; then one wavefront per query re-scores its bucket -- the query row is fetched once (LDS, broadcast reads) instead of once
// per candidate (half the gather traffic of ivf_exact_bucket_kernel, whose stream is not grouped by query), a query's ~ 36 rows are one
// or two half groups of 32.  Keys go to the key bucket ivf_bucket_select_kernel reads; bcount[q] counts every survivor (also those
// past the pitch: the select kernel reports the maximum, the host grows the pitch and repeats the pass).
constexpr int BEX_CHUNK = 128, BEX_EXTRA = 4096; // candidates of a query per unit of the exact stage; extra wavefronts that walk the unit list
// (a query's bucket counter takes ONE global atomic per workgroup round that has survivors of it: the round's 512 entries are counted
// per query in an LDS hash table first.  One returning atomic per survivor -- the first version -- serialises on the counters of the
// few queries that own thousands of survivors: 109 us at C3 against 16 for the plain compaction)
constexpr int BSC_PER = 2, BSC_HASH = 1024; // entries per thread and round; LDS hash slots (a power of two > 256 * BSC_PER)
__global__ __launch_bounds__(256) void ivf_bucket_scatter_kernel(const unsigned long long *__restrict__ strm, const float *__restrict__ su,
                                                                long long cap, const unsigned long long *__restrict__ cnt,
                                                                const float *__restrict__ bf, unsigned *__restrict__ brow,
                                                                unsigned *__restrict__ bcount, int bpitch,
                                                                int *__restrict__ kept_blk,
                                                                unsigned long long *__restrict__ units, unsigned *__restrict__ unit_cnt) {
	__shared__ unsigned hkey[BSC_HASH], hcnt[BSC_HASH], hbase[BSC_HASH];
	__shared__ int wsum[4];
	const unsigned long long have = *cnt;
	const long long n = have < (unsigned long long)cap ? (long long)have : cap;
	int mine = 0;
	for (long long b0 = (long long)blockIdx.x * (256 * BSC_PER); b0 < n; b0 += (long long)gridDim.x * (256 * BSC_PER)) {
		for (int h = threadIdx.x; h < BSC_HASH; h += 256)
			hkey[h] = 0xffffffffu, hcnt[h] = 0u;
		__syncthreads();
		unsigned row[BSC_PER], q[BSC_PER], hs[BSC_PER], rk[BSC_PER];
		bool keep[BSC_PER];
#pragma unroll
		for (int e = 0; e < BSC_PER; ++e) {
			const long long i = b0 + e * 256 + threadIdx.x;
			const unsigned long long ent = i < n ? strm[i] : 0ull;
			row[e] = (unsigned)ent, q[e] = (unsigned)(ent >> 32);
			keep[e] = i < n && !(su[i] < bf[q[e]]); // (NaN on either side: kept)
			hs[e] = 0u, rk[e] = 0u;
			if (keep[e]) {
				unsigned h = (q[e] * 2654435761u) >> 22; // (10 bits)
				for (;;) {
					const unsigned old = atomicCAS(&hkey[h], 0xffffffffu, q[e]);
					if (old == 0xffffffffu || old == q[e])
						break;
					h = (h + 1u) & (unsigned)(BSC_HASH - 1);
				}
				hs[e] = h;
				rk[e] = atomicAdd(&hcnt[h], 1u);
				++mine;
			}
		}
		__syncthreads();
		for (int h = threadIdx.x; h < BSC_HASH; h += 256)
			if (hcnt[h])
				hbase[h] = __hip_atomic_fetch_add(bcount + hkey[h], hcnt[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
#pragma unroll
		for (int e = 0; e < BSC_PER; ++e) {
			if (!keep[e])
				continue;
			const unsigned slot = hbase[hs[e]] + rk[e];
			if (slot < (unsigned)bpitch) {
				brow[(size_t)q[e] * (size_t)bpitch + slot] = row[e];
				// a query's first BEX_CHUNK rows belong to its own wavefront; every further chunk becomes a unit of the extra waves
				// (exactly one entry has slot = a given multiple of the chunk) -- a query sitting on thousands of candidates is not ONE
				// wave's serial work (C3: 36 queries of a giant list, ~ 3 500 each, were 0.45 ms of tail)
				if (slot >= (unsigned)BEX_CHUNK && (slot & (unsigned)(BEX_CHUNK - 1)) == 0u)
					units[__hip_atomic_fetch_add(unit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] =
					    ((unsigned long long)q[e] << 32) | (slot / (unsigned)BEX_CHUNK);
			}
		}
		__syncthreads(); // (the tables are cleared for the next round)
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		mine += __shfl_xor(mine, off);
	if ((threadIdx.x & 63) == 0)
		wsum[threadIdx.x >> 6] = mine;
	__syncthreads();
	// (the survivors of this workgroup: a plain store -- one atomic per workgroup on ONE counter serialised 2 700 of them behind the
	// kernel's last workgroups, 54 of its 69 us; the selection kernel's first wavefront adds the entries up)
	if (threadIdx.x == 0 && kept_blk)
		kept_blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// ARITH as ivf_exact_bucket_kernel; IL: the rows are a Flat index's pair-interleaved f32 store (csrc/common.h FlatGeom: inside every
// four consecutive k the floats sit as [k0,k2,k1,k3] in rows with bit 4 clear, [k1,k3,k0,k2] in rows with bit 4 set).  d = dp = 128.
template <int ARITH, bool IL>
__global__ __launch_bounds__(64) void ivf_bucket_exact_kernel(const unsigned *__restrict__ brow, const unsigned *__restrict__ bcount,
                                                             int bpitch, const float *__restrict__ x, const float *__restrict__ rows,
                                                             const int *__restrict__ perm, unsigned long long *__restrict__ bucket,
                                                             const IvfFlatArith fa, long long nq, const unsigned long long *__restrict__ units,
                                                             const unsigned *__restrict__ unit_cnt) {
	__shared__ __attribute__((aligned(16))) float yrows[32 * 132];
	__shared__ __attribute__((aligned(16))) float xq[128];
	__shared__ int mpos[32];
	const int lane = threadIdx.x, sub = lane >> 5, ch = lane & 31;
	// wavefront b < nq: the first BEX_CHUNK candidates of query b; the BEX_EXTRA wavefronts behind them walk the unit list
	const bool extra = (long long)blockIdx.x >= nq;
	const unsigned nunits = extra ? *unit_cnt : 1u;
	for (unsigned un = extra ? (unsigned)((long long)blockIdx.x - nq) : 0u; un < nunits; un += (unsigned)BEX_EXTRA) {
	long long q = blockIdx.x;
	int begin = 0;
	if (extra) {
		const unsigned long long ue = units[un];
		q = (long long)(ue >> 32);
		begin = (int)(unsigned)ue * BEX_CHUNK;
	}
	const unsigned have = bcount[q];
	int n = have < (unsigned)bpitch ? (int)have : bpitch;
	n = n < begin + BEX_CHUNK ? n : begin + BEX_CHUNK;
	if (n <= begin) {
		if (!extra)
			return;
		continue;
	}
	asm volatile("" ::: "memory");
	if (lane < 32)
		*(f32x4i *)(xq + 4 * lane) = *(const f32x4i *)(x + (size_t)q * 128 + 4 * lane);
	const unsigned *mine = brow + (size_t)q * (size_t)bpitch;
	unsigned long long *keys = bucket + (size_t)q * (size_t)bpitch;
	int pos = -1;
	if (lane < 32 && begin + lane < n) {
		const unsigned r = mine[begin + lane];
		pos = perm ? perm[r] : (int)r;
	}
	for (int g0 = begin; g0 < n; g0 += 32) {
		if (lane < 32)
			mpos[lane] = pos;
		asm volatile("" ::: "memory"); // (one wavefront: LDS keeps its accesses in order)
		f32x4i ry[16];
#pragma unroll
		for (int it = 0; it < 16; ++it) {
			const int pp = mpos[2 * it + sub];
			ry[it] = pp >= 0 ? *(const f32x4i *)(rows + (size_t)pp * 128 + ch * 4) : f32x4i{0.f, 0.f, 0.f, 0.f};
		}
		const int cur = pos; // (this half group's position of lanes 0 .. 31)
		pos = -1;
		if (lane < 32 && g0 + 32 + lane < n) { // the next half group's rows, behind this one's loads
			const unsigned r = mine[g0 + 32 + lane];
			pos = perm ? perm[r] : (int)r;
		}
#pragma unroll
		for (int it = 0; it < 16; ++it)
			*(f32x4i *)(yrows + (2 * it + sub) * 132 + ch * 4) = ry[it];
		asm volatile("" ::: "memory");
		if (lane < 32 && cur >= 0) {
			const float *y = yrows + lane * 132;
			const bool flip = IL && ((cur >> 4) & 1);
			float acc = 0.f;
			// (eight chunks per trip: fully unrolled, the ARITH = 2 instance hoists all 64 float4 of x and y -- 256 VGPRs + 20 AGPRs,
			// ONE wave per SIMD where LDS allows two or three)
#pragma unroll 8
			for (int c4 = 0; c4 < 32; ++c4) {
				const float4 xv = *(const float4 *)(xq + c4 * 4);
				const float4 yv = *(const float4 *)(y + c4 * 4);
				const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
				float ys[4] = {yv.x, yv.y, yv.z, yv.w};
				if (IL) {
					ys[0] = flip ? yv.z : yv.x, ys[1] = flip ? yv.x : yv.z, ys[2] = flip ? yv.w : yv.y, ys[3] = flip ? yv.y : yv.w;
				}
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					if (ARITH == 0) {
						const float t = __fsub_rn(xs[e], ys[e]);
						acc = fmaf(t, t, acc);
					} else {
						acc = fmaf(xs[e], ys[e], acc);
					}
				}
			}
			keys[g0 + lane] = eb_key<ARITH>(acc, cur, q, fa);
		}
		asm volatile("" ::: "memory"); // (the next half group's rows overwrite the tile)
	}
	if (!extra)
		return;
	} // unit
}
unsigned ivf_bucket_scatter_blocks(int64_t cap_entries) { // workgroups of the scatter kernel = entries of its per-workgroup survivor counts
	return (unsigned)std::min<int64_t>((cap_entries + 256 * BSC_PER - 1) / (256 * BSC_PER), 4096);
}
void launch_ivf_bucket_scatter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt,
                               const unsigned *d_gslot, int nclass, int kf, int64_t nq, float *d_bf, unsigned *d_brow, unsigned *d_bcount,
                               int bpitch, int *d_kept_blk, unsigned long long *d_units, unsigned *d_unit_cnt, hipStream_t st) {
	if (nq <= 0 || cap <= 0)
		return;
	if (!d_gslot)
		; // (d_bf holds the thresholds already: the Flat index, launch_collect_final_thr)
	else if (nclass == 32)
		hipLaunchKernelGGL(ivf_final_bound_kernel<32>, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, d_gslot, kf, (long long)nq, d_bf);
	else
		hipLaunchKernelGGL(ivf_final_bound_kernel<16>, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, d_gslot, kf, (long long)nq, d_bf);
	const unsigned blocks = ivf_bucket_scatter_blocks(cap);
	hipLaunchKernelGGL(ivf_bucket_scatter_kernel, dim3(blocks), dim3(256), 0, st, d_strm, d_su, (long long)cap, d_cnt, d_bf, d_brow, d_bcount,
	                   bpitch, d_kept_blk, d_units, d_unit_cnt);
	MVS_HIP(hipGetLastError());
}
size_t ivf_bucket_units_bytes(int64_t cap_entries) { // (every unit stands for BEX_CHUNK candidates of one query)
	return ((size_t)(cap_entries / BEX_CHUNK) + 64) * sizeof(unsigned long long);
}

// d_strm: the scan's candidate stream (ncand = its capacity or the host's count; the real number is min(*d_cnt, ncand));
// d_bucket [nq][bpitch] keys, d_bcount [nq] (zeroed); outputs as ivf_bucket_select_kernel describes
// fa != nullptr: the Flat shadow's arithmetic (see IvfFlatArith) -- the lists come out in FAISS's Flat L2 order, labels = row + label_offset
void launch_ivf_bucket_finish(int metric, const unsigned long long *d_strm, int64_t ncand, const unsigned long long *d_cnt,
                              unsigned long long *d_bucket, unsigned *d_bcount, int bpitch, int64_t nq, const float *d_x, int d,
                              const float *d_rows_csr, int dp_csr, const int *d_perm, int kk, float *d_pd, int64_t *d_pi,
                              const int64_t *d_rowids, const int64_t *d_idmap, int k, float *d_D, int64_t *d_I,
                              const int64_t *d_fin_rowids, const int64_t *d_fin_idmap, int *d_flag, unsigned long long *d_stats,
                              int *d_qfail, int *d_fail_cnt, int *d_fail_q, bool reset, hipStream_t st, const IvfFlatArith *fa,
                              int64_t label_offset, const unsigned *d_brow, int rows_interleaved, const unsigned long long *d_units,
                              const unsigned *d_unit_cnt, const int *d_kept_blk, int nkept_blk, unsigned long long *d_kept_out,
                              const IpFlatEmit *ipf) {
	if (nq <= 0)
		return;
	if (dp_csr % 4 != 0 || dp_csr > 128 || kk > 64 || kk < 1)
		throw_faiss("mvs::launch_ivf_bucket_finish", __FILE__, "row pitch %d / k %d is not served", dp_csr, kk);
	const bool l2 = metric_order(metric) == METRIC_L2;
	IvfFlatArith nofa;
	memset(&nofa, 0, sizeof nofa);
	if (d_brow) { // the candidates sit in their queries' buckets already (launch_ivf_bucket_scatter): one wavefront per query
		if (d != 128 || dp_csr != 128)
			throw_faiss("mvs::launch_ivf_bucket_finish", __FILE__, "the bucketed exact stage serves d = 128 (got d = %d, pitch %d)", d, dp_csr);
		const dim3 grid((unsigned)nq + (unsigned)BEX_EXTRA);
#define MVS_BEX(AR, ILV) hipLaunchKernelGGL((ivf_bucket_exact_kernel<AR, ILV>), grid, dim3(64), 0, st, d_brow, d_bcount, bpitch, d_x, d_rows_csr, d_perm, d_bucket, fa ? *fa : nofa, (long long)nq, d_units, d_unit_cnt)
		if (fa && fa->qn) {
			if (rows_interleaved)
				MVS_BEX(2, true);
			else
				MVS_BEX(2, false);
		} else if (l2) {
			if (rows_interleaved)
				MVS_BEX(0, true);
			else
				MVS_BEX(0, false);
		} else {
			if (rows_interleaved)
				MVS_BEX(1, true);
			else
				MVS_BEX(1, false);
		}
#undef MVS_BEX
	} else if (ncand > 0) { // a fixed grid walks the stream in strides (its length is on the device): two dispatch rounds of the 4 096 resident waves
		const dim3 grid((unsigned)std::min<int64_t>((ncand + 63) / 64, 8192));
		if (fa)
			hipLaunchKernelGGL(ivf_exact_bucket_kernel<2>, grid, dim3(64), 0, st, d_strm, (long long)ncand, d_cnt, d_x, d, d_rows_csr, dp_csr,
			                   d_perm, d_bucket, d_bcount, bpitch, *fa);
		else if (l2)
			hipLaunchKernelGGL(ivf_exact_bucket_kernel<0>, grid, dim3(64), 0, st, d_strm, (long long)ncand, d_cnt, d_x, d, d_rows_csr, dp_csr,
			                   d_perm, d_bucket, d_bcount, bpitch, nofa);
		else
			hipLaunchKernelGGL(ivf_exact_bucket_kernel<1>, grid, dim3(64), 0, st, d_strm, (long long)ncand, d_cnt, d_x, d, d_rows_csr, dp_csr,
			                   d_perm, d_bucket, d_bcount, bpitch, nofa);
	}
	IvfBucketSelectArgs a;
	memset(&a, 0, sizeof a);
	a.bucket = d_bucket, a.bcount = d_bcount, a.bpitch = bpitch, a.nq = nq, a.kk = kk;
	a.pd = d_pd, a.pi = (long long *)d_pi, a.rowids = (const long long *)d_rowids, a.idmap = (const long long *)d_idmap;
	a.label_offset = label_offset;
	a.k = k, a.D = d_D, a.I = (long long *)d_I, a.fin_rowids = (const long long *)d_fin_rowids, a.fin_idmap = (const long long *)d_fin_idmap;
	a.flag_cnt = d_flag, a.flag_q = d_flag ? d_flag + 1 : nullptr, a.stats = d_stats;
	a.qfail = d_qfail, a.fail_cnt = d_fail_cnt, a.fail_q = d_fail_q, a.reset = reset ? 1 : 0;
	a.kept_blk = d_kept_blk, a.nkept_blk = nkept_blk, a.kept_out = d_kept_out;
	if (ipf)
		a.ipf = *ipf;
	if (l2 || fa)
		hipLaunchKernelGGL(ivf_bucket_select_kernel<true>, dim3((unsigned)nq), dim3(64), 0, st, a);
	else
		hipLaunchKernelGGL(ivf_bucket_select_kernel<false>, dim3((unsigned)nq), dim3(64), 0, st, a);
	MVS_HIP(hipGetLastError());
}

// ---- Flat shadow (round 5): is the probed set PROVABLY enough? ------------------------------------------------------------------
// A Flat L2 index whose rows are clustered keeps this IVF index as a shadow (csrc/index.hip FlatIndex::shadow_*): a search probes
// the nprobe nearest lists exactly as C3 does, re-scores in the Flat arithmetic, and this kernel then checks, per query, that no
// UNPROBED list can hold a row that belongs into the result.  Every row y of list j satisfies (triangle inequality)
//        ||x - y|| >= ||x - c_j|| - ||y - c_j|| >= ||x - c_j|| - r_j,           r_j = the list's largest residual norm,
// so list j is out as soon as  (sqrt(cd_j - e_c) - r_j (1 + 1e-4))^2 (1 - 1e-6) > D_k + e_f  with
//   cd_j  the computed coarse distance ((xn + cn) - 2 ip, the matrix of csrc/coarse_select.hip), e_c = 2 (d + 2) u (||x|| + ||c_j||)^2
//         what that formula can be off by (Higham: d + 2 roundings of magnitudes <= (||x|| + ||c||)^2; factor 2 for slack),
//   D_k   the k-th COMPUTED distance found so far, e_f = 2 (d + 2) u (||x|| + ||y||_max)^2 what a row's computed Flat distance can be
//         below its true one -- a row with computed distance <= D_k (ties included) has true distance <= D_k + e_f.
// All in double.  A query with fewer than k results, a non-finite anything, or ONE list that cannot be excluded joins the fail list:
// the caller re-runs it on the Flat kernels.  One wavefront per query.
struct IvfShadowVerifyArgs {
	const float *cmat;     // [nq][nlist] computed coarse distances
	const float *cD;       // [nq][np] the probed lists' distances, ascending
	const long long *cI;   // [nq][np] the probed lists
	int nlist, np, d, k;
	const float *qn;       // [nq] ||x||^2
	const float *cn;       // [nlist] ||c||^2
	const unsigned *list_max; // [nlist] largest ||y - c||^2 of every list (bit pattern)
	const long long *lb, *le; // padded row range of every list (empty: lb == le)
	const float *D;        // [nq][k] the results
	const long long *I;
	const unsigned *ymax_bits; // the Flat index's largest ||y||^2 (bit pattern)
	int *fail_cnt;
	int *fail_q;
};
__global__ __launch_bounds__(64) void ivf_shadow_verify_kernel(const IvfShadowVerifyArgs a) {
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const double u = 5.9604644775390625e-08;
	const float Dk = a.D[q * a.k + a.k - 1];
	const bool have = a.I[q * a.k + a.k - 1] >= 0;
	const double xn = (double)a.qn[q], nx = sqrt(xn > 0 ? xn : 0.0);
	const double ny = sqrt((double)__uint_as_float(*a.ymax_bits));
	const double ef = 2.0 * (a.d + 2.0) * u * (nx + ny) * (nx + ny);
	const double thr = (double)Dk + ef;
	const float cdl = a.cD[q * a.np + a.np - 1];
	bool bad = !have || !(thr == thr) || !(xn == xn) || !(Dk < FLT_MAX);
	for (int j = lane; j < a.nlist && !bad; j += 64) {
		if (a.lb[j] == a.le[j])
			continue; // an empty list holds nothing
		const float cd = a.cmat[q * a.nlist + j];
		// (cI may be the PRUNED probe list of ivf_probe_prune_kernel: a pruned list counts as unprobed and is proven here, in this
		// index's arithmetic)
		bool probed = false;
		if (cd <= cdl) {
			for (int p = 0; p < a.np; ++p)
				probed |= a.cI[q * a.np + p] == (long long)j;
		}
		if (probed)
			continue;
		const double cnj = (double)a.cn[j], nc = sqrt(cnj > 0 ? cnj : 0.0);
		const double ec = 2.0 * (a.d + 2.0) * u * (nx + nc) * (nx + nc);
		const double lo = (double)cd - ec;
		const double A = lo > 0 ? sqrt(lo) : 0.0;
		const double R = sqrt((double)__uint_as_float(a.list_max[j])) * 1.0001;
		const double gap = A - R;
		const bool out = gap > 0 && gap * gap * (1.0 - 1e-6) > thr; // (NaN anywhere: false)
		bad |= !out;
	}
	if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0)
		a.fail_q[atomicAdd(a.fail_cnt, 1)] = (int)q;
}
void launch_ivf_shadow_verify(const float *d_cmat, const float *d_cD, const int64_t *d_cI, int64_t nq, int nlist, int np, int d, int k,
                              const float *d_qn, const float *d_cn, const unsigned *d_list_max, const int64_t *d_lb, const int64_t *d_le,
                              const float *d_D, const int64_t *d_I, const unsigned *d_ymax_bits, int *d_fail_cnt, int *d_fail_q,
                              hipStream_t st) {
	if (nq <= 0)
		return;
	IvfShadowVerifyArgs a;
	memset(&a, 0, sizeof a);
	a.cmat = d_cmat, a.cD = d_cD, a.cI = (const long long *)d_cI, a.nlist = nlist, a.np = np, a.d = d, a.k = k, a.qn = d_qn, a.cn = d_cn;
	a.list_max = d_list_max, a.lb = (const long long *)d_lb, a.le = (const long long *)d_le, a.D = d_D, a.I = (const long long *)d_I;
	a.ymax_bits = d_ymax_bits, a.fail_cnt = d_fail_cnt, a.fail_q = d_fail_q;
	hipLaunchKernelGGL(ivf_shadow_verify_kernel, dim3((unsigned)nq), dim3(64), 0, st, a);
	MVS_HIP(hipGetLastError());
}

// ---- probe pruning (round 5): probed lists that PROVABLY hold none of a query's k nearest rows are not scanned -------------------
// IndexIVF::search scans the nprobe nearest lists whatever they hold.  A list j whose every row is farther from x than k rows of the
// query's nearest lists are contributes nothing to the result -- not even under ties, the inequality is strict -- so leaving it out
// changes no label and no distance.  Witnesses: the first m probes (rank order) whose lists hold >= k rows together; every row y of
// list p has  ||x - y|| <= ||x - c_p|| + r_p  (r_p = the list's largest residual norm, list_max).  With cd = the COMPUTED coarse
// distance, e_c = 2 (d + 2) u (||x|| + ||c||)^2 what FAISS's (xn + yn) - 2 ip formula can be off by, eps = 2 (d + 2) u the relative
// error of the scanner's sum of squares:  W = max_{p < m} (sqrt(cd_p + e_c) + r_p)^2 (1 + eps) bounds the k-th COMPUTED result from
// above, (sqrt(cd_j - e_c) - r_j)^2 (1 - eps) bounds every computed distance of list j from below; j is pruned when the second
// exceeds the first.  All in double; r inflated by 1e-4 (its f32 chain); anything non-finite: nothing is pruned.  L2 only, no
// IDSelector (the witnesses must be selectable rows).  out[q][p] = the list or -1; the probe list itself (ties, tie pass) stays whole.
struct IvfProbePruneArgs {
	const float *x;        // [nq][d]
	const float *cD;       // [nq][np] computed coarse distances, ascending
	const long long *cI;   // [nq][np]
	long long *out;        // [nq][np]
	int *kept;             // [nq] probes kept (statistics), may be null
	int nq, np, d, k;
	const float *cn;       // [nlist] ||c||^2
	const unsigned *list_max;
	const long long *list_off; // [nlist + 1] rows of every list
};
__global__ __launch_bounds__(256) void ivf_probe_prune_kernel(const IvfProbePruneArgs a) {
	__shared__ double sU[4][256];
	__shared__ int sN[4][256];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long long q = (long long)blockIdx.x * 4 + w;
	if (q >= a.nq)
		return; // (no workgroup barrier below: a wave works on its own rows of the tables)
	const double u = 5.9604644775390625e-08, eps = 2.0 * (a.d + 2.0) * u;
	double xn = 0.0;
	for (int i = lane; i < a.d; i += 64) {
		const double v = (double)a.x[q * a.d + i];
		xn += v * v;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		xn += __shfl_xor(xn, off);
	const double nx = sqrt(xn) * (1.0 + 1e-9);
	double lo2[4]; // (np <= 256: four entries per lane) lower bound of the list's computed distances; < 0: cannot be pruned
	long long id[4];
#pragma unroll
	for (int e = 0; e < 4; ++e) {
		const int p = lane + 64 * e;
		lo2[e] = -1.0;
		id[e] = -1;
		if (p >= a.np)
			continue;
		const long long j = a.cI[q * a.np + p];
		id[e] = j;
		double U = INFINITY;
		int n = 0;
		if (j >= 0) {
			const double cd = (double)a.cD[q * a.np + p];
			const double cnj = (double)a.cn[j], nc = sqrt(cnj > 0 ? cnj : 0.0) * (1.0 + 1e-7);
			const double ec = 2.0 * (a.d + 2.0) * u * (nx + nc) * (nx + nc);
			const double R = sqrt((double)__uint_as_float(a.list_max[j])) * 1.0001;
			const double hi = sqrt(cd + ec) + R;
			U = hi * hi * (1.0 + eps);
			const double l = cd - ec, A = l > 0 ? sqrt(l) : 0.0, gap = A - R;
			if (gap > 0)
				lo2[e] = gap * gap * (1.0 - eps);
			n = (int)(a.list_off[j + 1] - a.list_off[j]);
			if (!(U == U) || !(cd == cd))
				U = INFINITY, lo2[e] = -1.0;
		}
		sU[w][p] = U;
		sN[w][p] = n;
	}
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (single wave: LDS keeps its accesses in order)
	double W = 0.0;
	int m = 0, cum = 0;
	for (; m < a.np && cum < a.k; ++m) {
		cum += sN[w][m];
		W = sU[w][m] > W ? sU[w][m] : W;
	}
	const bool can = cum >= a.k && W < INFINITY;
	int mine = 0;
#pragma unroll
	for (int e = 0; e < 4; ++e) {
		const int p = lane + 64 * e;
		if (p >= a.np)
			continue;
		const bool prune = can && p >= m && lo2[e] > W; // (strict; NaN: false)
		a.out[q * a.np + p] = prune ? -1ll : id[e];
		mine += (!prune && id[e] >= 0) ? 1 : 0;
	}
	if (a.kept) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
			mine += __shfl_xor(mine, off);
		if (lane == 0)
			a.kept[q] = mine;
	}
}
void launch_ivf_probe_prune(const float *d_x, int64_t nq, int d, const float *d_cD, const int64_t *d_cI, int np, int k, const float *d_cn,
                            const unsigned *d_list_max, const int64_t *d_list_off, int64_t *d_out, int *d_kept, hipStream_t st) {
	if (nq <= 0)
		return;
	IvfProbePruneArgs a;
	memset(&a, 0, sizeof a);
	a.x = d_x, a.cD = d_cD, a.cI = (const long long *)d_cI, a.out = (long long *)d_out, a.kept = d_kept, a.nq = (int)nq, a.np = np, a.d = d;
	a.k = k, a.cn = d_cn, a.list_max = d_list_max, a.list_off = (const long long *)d_list_off;
	hipLaunchKernelGGL(ivf_probe_prune_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, a);
	MVS_HIP(hipGetLastError());
}

// one bit per padded row: does the IDSelector accept the row's stored id (through the id map of an IndexIDMap wrapper)?
__device__ __forceinline__ bool ic_sel_member(const SelectorDev &s, long long id) {
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
__global__ __launch_bounds__(256) void ivf_rowmask_kernel(SelectorDev sel, const long long *__restrict__ rowids_mf,
                                                         const int *__restrict__ perm, const long long *__restrict__ idmap,
                                                         long long nrows, long long nwords, unsigned long long *__restrict__ mask) {
	const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	bool ok = false;
	if (row < nrows && perm[row] >= 0) {
		const long long lab = rowids_mf[row];
		ok = ic_sel_member(sel, idmap ? idmap[lab] : lab);
	}
	const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
	if ((threadIdx.x & 63) == 0 && (row >> 6) < nwords)
		mask[row >> 6] = b;
}
size_t ivf_rowmask_bytes(int64_t nrows_mf) {
	return (size_t)((nrows_mf + 63) / 64 + 64) * 8;
}
void launch_ivf_rowmask(SelectorDev sel, const int64_t *d_rowids_mf, const int *d_perm, const int64_t *d_idmap, int64_t nrows_mf,
                        void *d_mask, hipStream_t st) {
	const long long nwords = (long long)(ivf_rowmask_bytes(nrows_mf) / 8);
	const long long rows = nwords * 64;
	hipLaunchKernelGGL(ivf_rowmask_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, sel,
	                   (const long long *)d_rowids_mf, d_perm, (const long long *)d_idmap, (long long)nrows_mf, nwords,
	                   (unsigned long long *)d_mask);
	MVS_HIP(hipGetLastError());
}

// Flat small batches ride this kernel too (csrc/index.hip collect_candidates): the whole database is ONE list, every group of
// <= 128 queries one work item, identity query map.  With one wavefront per 2048-row segment, seven or eight of them per
// CU keep 8 KB each in flight, against two 16 KB blocks per CU for flat_bf16_collect_kernel.
__global__ void collect_flat_items_kernel(int4 *items, int *nitems, int *qidx, long long nq, long long n) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int ni = (int)((nq + 127) / 128);
	if (i < ni)
		items[i] = make_int4(0, (int)n, (int)(i * 128), (int)((nq - i * 128) < 128 ? (nq - i * 128) : 128));
	if (i < nq)
		qidx[i] = (int)i;
	if (i == 0)
		*nitems = ni;
}
void launch_collect_flat_items(void *d_items, int *d_nitems, int *d_qidx, int64_t nq, int64_t n, hipStream_t st) {
	const long long t = std::max<long long>(nq, (nq + 127) / 128);
	hipLaunchKernelGGL(collect_flat_items_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, (int4 *)d_items, d_nitems,
	                   d_qidx, (long long)nq, (long long)n);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
