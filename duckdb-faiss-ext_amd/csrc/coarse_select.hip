// csrc/coarse_select.hip -- IVF coarse quantisation (IndexIVF::search -> quantizer->search(n, x, nprobe), faiss/IndexIVF.cpp;
// reference call site src/faiss_extension.cpp:631 through IndexIVFFlat) for a Flat L2 quantizer of a few thousand centroids.
//
// The k-list kernels are built for k << N: with k = nprobe = 32 lists over 4096 rows every (query, row split) pair starts cold
// and a fifth of all rows is inserted somewhere (1.35 ms at C3, a third of the step).  Here the whole [nq][nlist] distance
// matrix is written once -- the SAME arithmetic as the BLAS branch (exhaustive_L2sqr_blas): ip = one k-ordered fma chain,
// dis = max(0, fmaf(-2, ip, ||x||^2 + ||y||^2)) -- and one wavefront per query selects the nprobe smallest (dis, id) pairs
// with two bitwise binary searches (on the distance bits, then on the ids among the rows tied at the nprobe-th distance).
// merge_partials_kernel then orders the list as always.  HBM-bound on the matrix: nq * nlist * 8 bytes.
#include "common.h"
#include "flat_fused.h"
#include "index.h"

#include <algorithm>
#include <cfloat>

namespace mvs {

// ---- D[q][c]: 128 queries x 128 centroids per workgroup, 8 x 8 chains per thread, operands through LDS in slabs of 16 dims --------
// (round 3: 64 x 128 with 4 x 8 chains read 12 LDS floats per 32 fmas -- 0.225 ms at C3's 10 000 x 4 096 x 128; 8 x 8 chains read 16
// per 64 with four ds_read_b128)
// centroid rows: pitch sdp, FlatGeom::pair_interleaved (every 4 floats stored [k0,k2,k1,k3] or, bit 4 of the row set,
// [k1,k3,k0,k2]); dimensions >= d are zero on both sides (fma(0, 0, acc) = acc)
__global__ __launch_bounds__(256) void coarse_dist_kernel(const float *__restrict__ x, long long nq, int d,
                                                         const float *__restrict__ cent, int sdp, int interleaved, int nlist,
                                                         const float *__restrict__ qn, const float *__restrict__ cn,
                                                         int is_l2, float *__restrict__ D) {
	__shared__ __attribute__((aligned(16))) float xs[16][128 + 4];
	__shared__ __attribute__((aligned(16))) float ys[16][128 + 4];
	const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
	const long long q0 = (long long)blockIdx.y * 128;
	const int c0 = blockIdx.x * 128;
	float acc[8][8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 8; ++j)
			acc[i][j] = 0.f;
	// the NEXT slab's operands travel from global memory into registers while the current slab is multiplied (without this the two
	// barriers of a slab exposed a global round trip per 16 dims: 0.224 ms at C3's shape whatever the tile)
	float xr[8];
	float4 yr[2];
	const long long qrow = q0 + (tid >> 1);
	const int crow = c0 + (tid >> 1);
	const bool flip = interleaved && ((crow >> 4) & 1);
	auto fetch = [&](int k0) {
		const int kk = k0 + 8 * (tid & 1);
#pragma unroll
		for (int e = 0; e < 8; ++e)
			xr[e] = (qrow < nq && kk + e < d) ? x[qrow * d + kk + e] : 0.f;
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			yr[g] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (crow < nlist && kk + 4 * g < sdp)
				yr[g] = *(const float4 *)(cent + (size_t)crow * sdp + kk + 4 * g);
		}
	};
	fetch(0);
	for (int k0 = 0; k0 < d; k0 += 16) {
#pragma unroll
		for (int e = 0; e < 8; ++e)
			xs[8 * (tid & 1) + e][tid >> 1] = xr[e];
#pragma unroll
		for (int g = 0; g < 2; ++g) {
			const float4 v = yr[g];
			float y0 = v.x, y1 = v.y, y2 = v.z, y3 = v.w;
			if (interleaved) {
				y0 = flip ? v.z : v.x, y1 = flip ? v.x : v.z, y2 = flip ? v.w : v.y, y3 = flip ? v.y : v.w;
			}
			const int kb = 8 * (tid & 1) + 4 * g;
			ys[kb + 0][tid >> 1] = y0;
			ys[kb + 1][tid >> 1] = y1;
			ys[kb + 2][tid >> 1] = y2;
			ys[kb + 3][tid >> 1] = y3;
		}
		__syncthreads();
		if (k0 + 16 < d)
			fetch(k0 + 16);
#pragma unroll
		for (int k = 0; k < 16; ++k) { // k ascending: every accumulator is ONE k-ordered chain
			const float4 xa = *(const float4 *)&xs[k][8 * ty], xb = *(const float4 *)&xs[k][8 * ty + 4];
			const float4 ya = *(const float4 *)&ys[k][8 * tx], yb = *(const float4 *)&ys[k][8 * tx + 4];
			const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
			// two chains per v_pk_fma_f32 (each half is an ordinary fma: same bits as the scalar chain, half the issue slots -- the
			// kernel is bound by the vector ALU's issue rate: 0.224 ms whatever the tile or the staging)
			typedef float f32x2c __attribute__((ext_vector_type(2)));
			const f32x2c yp[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}};
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const f32x2c xx = {xv[i], xv[i]};
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					f32x2c a2 = {acc[i][2 * j], acc[i][2 * j + 1]};
					a2 = __builtin_elementwise_fma(xx, yp[j], a2);
					acc[i][2 * j] = a2.x;
					acc[i][2 * j + 1] = a2.y;
				}
			}
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const long long q = q0 + 8 * ty + i;
		if (q >= nq)
			continue;
		const float xn = is_l2 ? qn[q] : 0.f;
		float out[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int c = c0 + 8 * tx + j;
			float dis = fmaf(-2.0f, acc[i][j], xn + (c < nlist ? cn[c] : 0.f));
			dis = dis < 0.f ? 0.f : dis; // FAISS: if (dis < 0) dis = 0  (NaN stays NaN)
			out[j] = is_l2 ? dis : acc[i][j]; // inner product: the chain itself
		}
		if (c0 + 8 * tx + 7 < nlist) {
			*(float4 *)(D + q * nlist + c0 + 8 * tx) = make_float4(out[0], out[1], out[2], out[3]);
			*(float4 *)(D + q * nlist + c0 + 8 * tx + 4) = make_float4(out[4], out[5], out[6], out[7]);
		} else {
#pragma unroll
			for (int j = 0; j < 8; ++j)
				if (c0 + 8 * tx + j < nlist)
					D[q * nlist + c0 + 8 * tx + j] = out[j];
		}
	}
}

// ---- the same matrix on the f32 matrix pipe (round 4) ---------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 accumulates D = A B + C over its two k-steps in k order with one rounding per product -- the k-ordered
// fma chain of the vector-ALU kernel above, bit for bit (csrc/flat_mfma.hip rests on the same property) -- at 4x the vector ALU's
// issue-bound rate.  A = QUERIES (M), B = CENTROIDS (N): lane l holds A[q = l & 31][k = l >> 5], B[k = l >> 5][c = l & 31] and
// D[q = 8 g + 4 (l >> 5) + e][c = l & 31] in register 4 g + e, so lanes 0..31 of a register write 32 CONSECUTIVE centroids of one
// query: 128-byte runs, no transpose.  Workgroup = 128 queries x 128 centroids, wave w = centroid block 32 w .. 32 w + 31 against
// all four query blocks (64 accumulator registers); operands through LDS in slabs of 16 dims exactly as above.
typedef float f32x16c __attribute__((ext_vector_type(16)));
// Round 4, second cut (C3: 170 -> see DESIGN 3.2): slabs of 32 dims in TWO LDS buffers -- one barrier per slab, and the next slab's
// operands are in flight (registers) during the 64 MFMAs of the current one (with 16-dim slabs the 32 MFMAs of a slab were shorter
// than a global round trip and the pipe sat idle 60 % of the time); float4 loads of the query rows when d % 4 == 0; the query norms
// of the workgroup in LDS for the epilogue.
constexpr int CM_K = 32, CM_P = 128 + 4;
// The (tile, slab) loop is flattened over a workgroup's tiles (tile += gridDim.x): launched with one workgroup per 128 x 128 tile
// (default) it is the plain kernel; launched PERSISTENT (option ivf_coarse_persistent = 1: two workgroups per CU, the next tile's
// first slab fetched under the current tile's last MFMAs, the epilogue's stores draining under the next tile) it measured SLOWER at
// C3 -- 226 vs 176 us, every part of it (staging alone 73 vs 51, + stores 144 vs 93, + MFMAs without stores 169 vs 146:
// profiles/r4_coarse_kernel_ablation.txt).  What that file shows for the default launch: staging (51 us), MFMAs (95) and stores
// (30) add up instead of overlapping -- the two workgroups of a CU start together and stay in step.
__global__ __launch_bounds__(256) void coarse_dist_mfma_kernel(const float *__restrict__ x, long long nq, int d,
                                                              const float *__restrict__ cent, int sdp, int interleaved, int nlist,
                                                              const float *__restrict__ qn, const float *__restrict__ cn,
                                                              int is_l2, float *__restrict__ D, int abl) {
	extern __shared__ __attribute__((aligned(16))) float cm_lds[]; // [2]{xs[CM_K][CM_P], ys[CM_K][CM_P]}, qs[2][128]
	float *qs = cm_lds + 4 * CM_K * CM_P;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
	const int ntx = (nlist + 127) / 128;
	const long long ntiles = (long long)ntx * ((nq + 127) / 128);
	f32x16c acc[4];
#pragma unroll
	for (int t = 0; t < 4; ++t)
#pragma unroll
		for (int r = 0; r < 16; ++r)
			acc[t][r] = 0.f;
	float4 xr[4], yr[4]; // this thread's 16 dims (half = tid & 1) of query row / centroid row tid >> 1 of the slab in flight
	float qv = 0.f;      // ... and, with a tile's first slab, the norm of query row tid (tid < 128)
	bool flip = false;
	const bool x4 = (d & 3) == 0;
	auto fetch = [&](long long tile, int k0) {
		const long long q0 = (tile / ntx) * 128;
		const int c0 = (int)(tile % ntx) * 128;
		const long long qrow = q0 + (tid >> 1);
		const int crow = c0 + (tid >> 1);
		flip = interleaved && ((crow >> 4) & 1);
		const int kk = k0 + 16 * (tid & 1);
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			xr[g] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (qrow < nq) {
				const float *src = x + qrow * d + kk + 4 * g;
				if (x4) {
					if (kk + 4 * g < d)
						xr[g] = *(const float4 *)src;
				} else {
					xr[g].x = kk + 4 * g + 0 < d ? src[0] : 0.f;
					xr[g].y = kk + 4 * g + 1 < d ? src[1] : 0.f;
					xr[g].z = kk + 4 * g + 2 < d ? src[2] : 0.f;
					xr[g].w = kk + 4 * g + 3 < d ? src[3] : 0.f;
				}
			}
			yr[g] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (crow < nlist && kk + 4 * g < sdp)
				yr[g] = *(const float4 *)(cent + (size_t)crow * sdp + kk + 4 * g);
		}
		if (k0 == 0 && tid < 128)
			qv = (is_l2 && q0 + tid < nq) ? qn[q0 + tid] : 0.f;
	};
	auto stage = [&](int buf) {
		float *xs = cm_lds + buf * 2 * CM_K * CM_P, *ys = xs + CM_K * CM_P;
		const int col = tid >> 1;
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int kb = 16 * (tid & 1) + 4 * g;
			xs[(kb + 0) * CM_P + col] = xr[g].x;
			xs[(kb + 1) * CM_P + col] = xr[g].y;
			xs[(kb + 2) * CM_P + col] = xr[g].z;
			xs[(kb + 3) * CM_P + col] = xr[g].w;
			const float4 v = yr[g];
			float y0 = v.x, y1 = v.y, y2 = v.z, y3 = v.w;
			if (interleaved) {
				y0 = flip ? v.z : v.x, y1 = flip ? v.x : v.z, y2 = flip ? v.w : v.y, y3 = flip ? v.y : v.w;
			}
			ys[(kb + 0) * CM_P + col] = y0;
			ys[(kb + 1) * CM_P + col] = y1;
			ys[(kb + 2) * CM_P + col] = y2;
			ys[(kb + 3) * CM_P + col] = y3;
		}
	};
	long long tile = blockIdx.x;
	if (tile < ntiles)
		fetch(tile, 0);
	int buf = 0, tp = 0;
	for (; tile < ntiles; tile += gridDim.x, tp ^= 1) {
		const long long q0 = (tile / ntx) * 128;
		const int c0 = (int)(tile % ntx) * 128;
		for (int k0 = 0; k0 < d; k0 += CM_K, buf ^= 1) {
			stage(buf);
			if (k0 == 0 && tid < 128)
				qs[tp * 128 + tid] = qv;
			// (the buffer written here was last read two slabs ago -- every wave has passed the barrier of the slab between; the same
			// holds for qs[tp], read in the epilogue two tiles ago)
			__syncthreads();
			if (k0 + CM_K < d)
				fetch(tile, k0 + CM_K);
			else if (tile + gridDim.x < ntiles)
				fetch(tile + gridDim.x, 0);
			const float *xs = cm_lds + buf * 2 * CM_K * CM_P, *ys = xs + CM_K * CM_P;
#pragma unroll
			for (int k = 0; k < CM_K; k += 2) { // k ascending: every accumulator element is ONE k-ordered chain
#ifdef MVS_PROFILING
				if (abl & 2)
					break;
#endif
				const float b = ys[(k + h) * CM_P + 32 * wave + ln];
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const float a = xs[(k + h) * CM_P + 32 * t + ln];
					acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
				}
			}
		}
#ifdef MVS_PROFILING
		if ((abl & 1) && acc[0][0] != 12345.678f) // (profiling library: no matrix written -- results are wrong)
			continue;
#endif
		const int c = c0 + 32 * wave + ln;
		const float cnv = (is_l2 && c < nlist) ? cn[c] : 0.f;
#pragma unroll
		for (int t = 0; t < 4; ++t) {
#pragma unroll
			for (int g = 0; g < 4; ++g)
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					const int ql = 32 * t + 8 * g + 4 * h + e;
					const long long q = q0 + ql;
					if (q >= nq || c >= nlist)
						continue;
					const float ip = acc[t][4 * g + e];
					float dis = fmaf(-2.0f, ip, qs[tp * 128 + ql] + cnv);
					dis = dis < 0.f ? 0.f : dis; // FAISS: if (dis < 0) dis = 0  (NaN stays NaN)
					D[q * nlist + c] = is_l2 ? dis : ip;
				}
#pragma unroll
			for (int r = 0; r < 16; ++r)
				acc[t][r] = 0.f;
		}
	}
}
// Round 5: the same tile, restaged, and the matrix written UNDER the next tile's MFMAs.
// (1) The ISA of the kernel above holds one reason its three parts ADD UP (staging 51 + MFMAs 95 + stores 30 = 176 us,
// profiles/r4_coarse_kernel_ablation.txt): the eight loads of a slab sit behind branches with an s_waitcnt vmcnt(0) between them -- four
// serial round trips per slab -- and each touches 32 cache lines for 16 bytes.  Here: FOUR lanes per row and slab (32 bytes each, two
// float4: an instruction reads 64 bytes of 16 rows' lines, its neighbour the other 64), every load unconditional from a clamped
// address (zeroed by a select when a CUT slab is staged), issued back to back a whole slab ahead of its LDS write; the tile in LDS
// ROW-major ([256 rows][32 dims + 4]) with every block of 8 dims de-interleaved ([k0 k2 k4 k6 | k1 k3 k5 k7]): eight ds_write_b128 per
// thread instead of 32 transposing ds_write_b32, and lane (ln, h) of the MFMA loop reads the four k-steps of its half with ONE
// conflict-free ds_read_b128 (20 per slab against 64 MFMAs of 64 cycles), requested one block of 16 MFMAs ahead.
// (2) The other reason: every tile costs the same, so all workgroups of the chip reach their epilogue TOGETHER -- a 33 MB burst of
// stores per round with the matrix pipe idle, then MFMAs with HBM idle (one workgroup per tile, (1) alone: 170 -> 136 us at C3;
// with (2): 123 .. 131; what is left and what was tried on top: profiles/r5_coarse_kernel.txt).
// Persistent workgroups (two per CU, tile += gridDim.x; the (tile, slab) loop flattened so that the next tile's first slabs are
// fetched and staged under the current tile's last MFMAs) keep a finished tile's 64 distances per lane in registers and store them
// sixteen at a time behind the MFMA blocks of the NEXT tile's first slab.
// Same instruction, same k order, same epilogue arithmetic: same bits.  d % 4 == 0 (16-byte query rows), else the kernel above.
constexpr int C2_K = 32, C2_P = C2_K + 4;
template <bool IL, bool L2>
__global__ __launch_bounds__(256, 2) void coarse_dist_mfma2_kernel(const float *__restrict__ x, long long nq, int d,
                                                                  const float *__restrict__ cent, int sdp, int nlist,
                                                                  const float *__restrict__ qn, const float *__restrict__ cn,
                                                                  float *__restrict__ D, int abl) {
	extern __shared__ __attribute__((aligned(16))) float c2_lds[]; // [2][256][C2_P]: rows 0..127 queries, 128..255 centroids; qs[2][128]
	float *qs = c2_lds + 2 * 256 * C2_P;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, ln = lane & 31;
	const int ntx = (nlist + 127) / 128;
	const int ntiles = ntx * (int)((nq + 127) / 128);
	const int S = (d + C2_K - 1) / C2_K; // slabs per tile
	const int r4 = tid >> 2, c8 = tid & 3; // staging: tile rows r4 + 64 i (i < 2 queries, i >= 2 centroids), dims 8 c8 .. 8 c8 + 7 of the slab
	if ((int)blockIdx.x >= ntiles)
		return;
	const int total = ((ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1) * S; // slabs of this workgroup
	float4 rg[8];
	float qv = 0.f;
	unsigned fl = 0u; // bit i: centroid row r4 + 64 i of the slab in flight is a flipped row of the pair-interleaved store
	// (32-bit running positions: a 64-bit division per slab costs more instructions than the slab's address arithmetic)
	int f_tile = blockIdx.x, f_sl = 0, f_c0 = (int)(blockIdx.x % (unsigned)ntx) * 128; // the slab the next fetch() requests
	long long f_q0 = (long long)(blockIdx.x / (unsigned)ntx) * 128;
	auto fetch = [&]() __attribute__((always_inline)) {
		const int kk = f_sl * C2_K + 8 * c8;
		// (dims past the row: a valid address, zeroed when the slab is staged)
		const int kx0 = kk < d ? kk : 0, kx1 = kk + 4 < d ? kk + 4 : 0, kc0 = kk < sdp ? kk : 0, kc1 = kk + 4 < sdp ? kk + 4 : 0;
		fl = 0u;
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const long long qrow = f_q0 + r4 + 64 * i;
			const int crow = f_c0 + r4 + 64 * i;
			const float *xr = x + (qrow < nq ? qrow : nq - 1) * d; // (rows past the end: results never stored)
			const float *cr = cent + (size_t)(crow < nlist ? crow : nlist - 1) * sdp;
			rg[2 * i] = *(const float4 *)(xr + kx0);
			rg[2 * i + 1] = *(const float4 *)(xr + kx1);
			rg[4 + 2 * i] = *(const float4 *)(cr + kc0);
			rg[4 + 2 * i + 1] = *(const float4 *)(cr + kc1);
			fl |= (IL && ((crow >> 4) & 1)) ? (1u << i) : 0u;
		}
		if (f_sl == 0 && tid < 128)
			qv = (L2 && f_q0 + tid < nq) ? qn[f_q0 + tid] : 0.f;
		if (++f_sl == S) {
			f_sl = 0;
			f_tile += gridDim.x;
			f_q0 = (long long)((unsigned)f_tile / (unsigned)ntx) * 128;
			f_c0 = (int)((unsigned)f_tile % (unsigned)ntx) * 128;
		}
	};
	int s_sl = 0, s_par = 0; // the slab the next stage() writes: its position in the tile, its tile's parity
	auto stage = [&](int buf) __attribute__((always_inline)) {
		// LDS rows of 32 dims + 4: every block of 8 dims as [k0 k2 k4 k6 | k1 k3 k5 k7] -- lane (ln, h) of the MFMA loop reads the four
		// k-steps of its half (dims 8 b + 2 j + h) with ONE ds_read_b128 (16 lanes of a group on 16 different 16-byte slots of the
		// 256-byte bank row: conflict-free, where ds_read_b32 of one dim across 32 rows is four-way whatever 16-byte-aligned pitch)
		float *base = c2_lds + buf * 256 * C2_P + r4 * C2_P + 8 * c8;
		const int kk = s_sl * C2_K + 8 * c8;
		const bool okx0 = kk < d, okx1 = kk + 4 < d, okc0 = kk < sdp, okc1 = kk + 4 < sdp;
		const bool cut = (s_sl + 1) * C2_K > (d < sdp ? d : sdp); // (uniform: only a row's last slab can reach past its end)
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			float4 a = rg[2 * i], b = rg[2 * i + 1], u = rg[4 + 2 * i], v = rg[4 + 2 * i + 1]; // (values: see fetch())
			if (cut) {
				a.x = okx0 ? a.x : 0.f, a.y = okx0 ? a.y : 0.f, a.z = okx0 ? a.z : 0.f, a.w = okx0 ? a.w : 0.f;
				b.x = okx1 ? b.x : 0.f, b.y = okx1 ? b.y : 0.f, b.z = okx1 ? b.z : 0.f, b.w = okx1 ? b.w : 0.f;
				u.x = okc0 ? u.x : 0.f, u.y = okc0 ? u.y : 0.f, u.z = okc0 ? u.z : 0.f, u.w = okc0 ? u.w : 0.f;
				v.x = okc1 ? v.x : 0.f, v.y = okc1 ? v.y : 0.f, v.z = okc1 ? v.z : 0.f, v.w = okc1 ? v.w : 0.f;
			}
			*(float4 *)(base + 64 * i * C2_P) = make_float4(a.x, a.z, b.x, b.z);
			*(float4 *)(base + 64 * i * C2_P + 4) = make_float4(a.y, a.w, b.y, b.w);
			float4 ev = make_float4(u.x, u.z, v.x, v.z), od = make_float4(u.y, u.w, v.y, v.w);
			if (IL) { // (csrc/common.h FlatGeom: four dims as [k0,k2,k1,k3] in rows with bit 4 clear, [k1,k3,k0,k2] with it set)
				const bool f = (fl >> i) & 1u;
				const float4 lo = make_float4(u.x, u.y, v.x, v.y), hi = make_float4(u.z, u.w, v.z, v.w);
				ev.x = f ? hi.x : lo.x, ev.y = f ? hi.y : lo.y, ev.z = f ? hi.z : lo.z, ev.w = f ? hi.w : lo.w;
				od.x = f ? lo.x : hi.x, od.y = f ? lo.y : hi.y, od.z = f ? lo.z : hi.z, od.w = f ? lo.w : hi.w;
			}
			*(float4 *)(base + (128 + 64 * i) * C2_P) = ev;
			*(float4 *)(base + (128 + 64 * i) * C2_P + 4) = od;
		}
		if (s_sl == 0 && tid < 128)
			qs[s_par * 128 + tid] = qv; // (read at the end of its tile; the tile two back was finished a barrier ago)
		if (++s_sl == S)
			s_sl = 0, s_par ^= 1;
	};
	f32x16c acc[4];
#pragma unroll
	for (int t = 0; t < 4; ++t)
#pragma unroll
		for (int r = 0; r < 16; ++r)
			acc[t][r] = 0.f;
	// the finished tile waiting to be stored: out[16 t + 4 g + e] = its distance for query row 32 t + 8 g + 4 h + e, centroid column ln
	float out[64];
	float *outp = D;       // D + (q0 + 4 h) * nlist + c of that tile
	int qlim = 0;          // rows r = 32 t + 8 g + e with r < qlim are inside the matrix (0: the column is not, or nothing is pending)
	bool pending = false, whole = false; // whole: every row and column of that tile is inside the matrix
	// (fetch() is UNCONDITIONAL, also past the workgroup's last slab -- its addresses are clamped into the matrices anyway: behind a
	// branch the loaded registers are merged with their old values, and the copies that merge needs wait for the loads on the spot)
	fetch();
	stage(0);
	fetch();
	__syncthreads();
	int buf = 0, c_sl = 0, c_par = 0, c_tile = blockIdx.x; // the slab under the MFMAs
	for (int it = 0; it < total; ++it, buf ^= 1) {
#ifdef MVS_PROFILING
		if (!(abl & 8)) // (8: MFMAs on whatever the first slab left in LDS)
#endif
		{
			if (it + 1 < total)
				stage(buf ^ 1); // (last read one slab ago: every wave has passed the barrier since)
			fetch();
		}
		const float *xs = c2_lds + buf * 256 * C2_P + ln * C2_P + 4 * h, *ys = xs + (128 + 32 * wave) * C2_P;
		// the operands of a block of 8 dims (four k-steps x four query blocks + the centroid block: five ds_read_b128), requested one
		// block AHEAD of the 16 MFMAs that consume them
		float4 a[2][4], b[2];
#pragma unroll
		for (int t = 0; t < 4; ++t)
			a[0][t] = *(const float4 *)(xs + 32 * t * C2_P);
		b[0] = *(const float4 *)ys;
		const bool flush = pending; // (uniform)
#pragma unroll
		for (int g = 0; g < C2_K / 8; ++g) { // k ascending: every accumulator element is ONE k-ordered chain
			if (g + 1 < C2_K / 8) {
#pragma unroll
				for (int t = 0; t < 4; ++t)
					a[(g + 1) & 1][t] = *(const float4 *)(xs + 32 * t * C2_P + 8 * g + 8);
				b[(g + 1) & 1] = *(const float4 *)(ys + 8 * g + 8);
			}
			__builtin_amdgcn_sched_barrier(0); // (left alone the scheduler sinks the requests behind the MFMAs they are meant to hide under)
#ifdef MVS_PROFILING
			if (abl & 2)
				continue;
#endif
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const float bv = j == 0 ? b[g & 1].x : j == 1 ? b[g & 1].y : j == 2 ? b[g & 1].z : b[g & 1].w;
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const float4 av = a[g & 1][t];
					acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w, bv, acc[t], 0, 0, 0);
				}
			}
			if (flush) { // sixteen of the previous tile's 64 stores behind this block
				if (whole) { // (uniform: the tile lies inside the matrix -- no test per store)
#pragma unroll
					for (int j = 0; j < 16; ++j) {
						const int r = 16 * g + j, row = 32 * (r >> 4) + 8 * ((r >> 2) & 3) + (r & 3);
						outp[(long long)row * nlist] = out[r];
					}
				} else {
#pragma unroll
					for (int j = 0; j < 16; ++j) {
						const int r = 16 * g + j, row = 32 * (r >> 4) + 8 * ((r >> 2) & 3) + (r & 3);
						if (row < qlim)
							outp[(long long)row * nlist] = out[r];
					}
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		pending = false;
#ifdef MVS_PROFILING
		if (abl & 32) // (32: no epilogue at all)
			c_sl = -1000000;
#endif
		if (++c_sl == S) { // the tile is complete: its distances into out[], the accumulators cleared
			const long long q0 = (long long)((unsigned)c_tile / (unsigned)ntx) * 128;
			const int c0 = (int)((unsigned)c_tile % (unsigned)ntx) * 128, c = c0 + 32 * wave + ln;
			const float cnv = (L2 && c < nlist) ? cn[c] : 0.f;
			const float *qt = qs + c_par * 128 + 4 * h;
			c_sl = 0, c_par ^= 1, c_tile += gridDim.x;
#pragma unroll
			for (int t = 0; t < 4; ++t) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					const float4 q4 = *(const float4 *)(qt + 32 * t + 8 * g); // the norms of rows 32 t + 8 g + 4 h + 0 .. 3
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const float ip = acc[t][4 * g + e];
						float dis = fmaf(-2.0f, ip, (e == 0 ? q4.x : e == 1 ? q4.y : e == 2 ? q4.z : q4.w) + cnv);
						dis = dis < 0.f ? 0.f : dis; // FAISS: if (dis < 0) dis = 0  (NaN stays NaN)
						out[16 * t + 4 * g + e] = L2 ? dis : ip;
					}
				}
#pragma unroll
				for (int r = 0; r < 16; ++r)
					acc[t][r] = 0.f;
			}
			const long long left = nq - q0 - 4 * h;
			qlim = c < nlist ? (int)(left < 128 ? left : 128) : 0;
			whole = q0 + 128 <= nq && c0 + 128 <= nlist;
			outp = D + (q0 + 4 * h) * nlist + (c < nlist ? c : 0);
#ifdef MVS_PROFILING
			if ((abl & 1) && out[0] != 12345.678f) // (profiling library: no matrix written -- results are wrong)
				qlim = 0;
			if (abl & 4) // ... or every workgroup writes one and the same tile: the store instructions without their HBM traffic
				outp = D + 4 * h * nlist + 32 * wave + ln;
#endif
			pending = true;
		}
#ifdef MVS_PROFILING
		if (!(abl & 16)) // (16: no barrier between the slabs)
#endif
			__syncthreads();
	}
	if (pending) { // the workgroup's last tile
#pragma unroll
		for (int r = 0; r < 64; ++r) {
			const int row = 32 * (r >> 4) + 8 * ((r >> 2) & 3) + (r & 3);
			if (row < qlim)
				outp[(long long)row * nlist] = out[r];
		}
	}
}
static size_t coarse_mfma2_lds_bytes() {
	return ((size_t)2 * 256 * C2_P + 256) * sizeof(float);
}
static int device_cu_count() { // compute units of the current device (256 on MI355X)
	static int cached[64];
	int dev = 0;
	MVS_HIP(hipGetDevice(&dev));
	if (dev < 0 || dev >= 64)
		return 256;
	if (!cached[dev]) {
		int n = 0;
		MVS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
		cached[dev] = n > 0 ? n : 256;
	}
	return cached[dev];
}
static size_t coarse_mfma_lds_bytes() {
	return ((size_t)4 * CM_K * CM_P + 256) * sizeof(float);
}

// ---- one wavefront per query: the np smallest (dis, id) of its row of D -> pd / pi [nq][np] (any order; missing: FLT_MAX, -1) --
// PL = values per lane (nlist <= 64 PL, a multiple of 4).  A row is a candidate iff dis < FLT_MAX / score > -FLT_MAX (the heap's
// strict compare against its neutral value; NaN never enters).  Inner product: the pure order (score descending, id ascending);
// the caller asks for one entry more than nprobe and lets the merge flag boundary ties (FlatIndex::resolve_ip_ties).
template <int PL, bool IS_L2>
// outD != nullptr (L2, round 4): the list comes out ORDERED (dis ascending, id ascending -- what merge_partials_kernel made of it in a
// launch of its own) straight into the caller's [nq][np] distances / labels (id + label_offset); pd / pi are not written.
__global__ __launch_bounds__(64) void coarse_select_kernel(const float *__restrict__ D, int nlist, int np,
                                                          float *__restrict__ pd, int *__restrict__ pi,
                                                          float *__restrict__ outD, long long *__restrict__ outI, long long label_offset) {
	const long long q = blockIdx.x;
	const int lane = threadIdx.x;
	const float *row = D + q * nlist;
	unsigned key[PL]; // lane holds ids (4 (64 it + lane) + e), it < PL / 4
#pragma unroll
	for (int it = 0; it < PL / 4; ++it) {
		const int c = 4 * (64 * it + lane);
		// (nlist is a multiple of 4: a group of four is inside the row or past its end.  The address is clamped so that the loads of
		// a lane are issued back to back -- behind branches they were 16 serial round trips; past the end: NaN, never a candidate)
		const bool in = c < nlist;
		float4 v = *(const float4 *)(row + (in ? c : 0));
		const float pad = __uint_as_float(0x7fc00000u);
		v.x = in ? v.x : pad, v.y = in ? v.y : pad, v.z = in ? v.z : pad, v.w = in ? v.w : pad;
		const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			// L2: dis >= 0 or NaN, the bit pattern orders the finite values; inner product: larger score = smaller key
			const unsigned b = IS_L2 ? __float_as_uint(f[e]) : ~f2key(f[e]);
			key[4 * it + e] = (IS_L2 ? f[e] < FLT_MAX : f[e] > -FLT_MAX) ? b : 0xffffffffu;
		}
	}
	// Fast path (np <= 64).  The np-th smallest key is <= U, the np-th smallest of the 64 LANE MINIMA (64 distinct elements of the
	// row).  The keys <= U -- a few times np of them unless the row is adversarially ordered -- are compacted into LDS as
	// (key << 32 | id): the np smallest (dis, id) pairs are the np smallest of these 64-bit values, found by one bitwise binary
	// search over <= 8 entries per lane instead of two over PL (the full search below: 49 steps x PL compares, 0.26 ms at C3).
	__shared__ unsigned long long cand[512];
	if (np <= 64) {
		unsigned lmin = key[0];
#pragma unroll
		for (int j = 1; j < PL; ++j)
			lmin = key[j] < lmin ? key[j] : lmin;
		unsigned U = 0u;
#pragma unroll 1
		for (int b = 31; b >= 0; --b) {
			const unsigned t = U | (1u << b);
			if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(lmin < t)) < np)
				U = t;
		}
		if (U != 0xffffffffu) { // (else: fewer than np lanes hold a candidate at all)
			int mine = 0;
#pragma unroll
			for (int j = 0; j < PL; ++j)
				mine += key[j] <= U ? 1 : 0;
			// (exclusive prefix over the lanes in six shuffle steps; round 4 walked the 63 lanes with v_readlane -- with the bit searches
			// below this kernel is bound by its own instruction latency, not by the 16 KB it reads per query: 60 us at C3)
			int inc = mine;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const int v = __shfl_up(inc, off);
				inc += lane >= off ? v : 0;
			}
			int pos = inc - mine;
			const int total = __builtin_amdgcn_readlane(inc, 63);
			if (total <= 512) {
#pragma unroll
				for (int j = 0; j < PL; ++j) {
					const unsigned id = 4u * (64u * (unsigned)(j >> 2) + (unsigned)lane) + (unsigned)(j & 3);
					if (key[j] <= U)
						cand[pos++] = ((unsigned long long)key[j] << 32) | id;
				}
				__syncthreads();
				unsigned long long e[8];
#pragma unroll
				for (int i = 0; i < 8; ++i)
					e[i] = 64 * i + lane < total ? cand[64 * i + lane] : ~0ull;
				// V = the np-th smallest entry (total >= np: the np lane minima <= U are among them; entries are distinct)
				unsigned long long V = 0ull;
#pragma unroll 1
				for (int b = 63; b >= 0; --b) {
					if (b == 31)
						b = 16; // bits 31..17 are zero in every entry (ids < 2^17)
					const unsigned long long t = V | (1ull << b);
					int cnt = 0;
					if (total <= 128) { // (the usual case: a few times np keys survived -- two registers per lane hold them all)
						cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(e[0] < t)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(e[1] < t));
					} else {
#pragma unroll
						for (int i = 0; i < 8; ++i)
							cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(e[i] < t));
					}
					if (cnt < np)
						V = t;
				}
				int base = 0;
				if (outD)
					__syncthreads(); // (cand is reused for the compacted list: every lane has read its entries)
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const bool take = e[i] <= V;
					const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
					if (take) {
						const int p = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
						const unsigned kv = (unsigned)(e[i] >> 32);
						if (outD) {
							cand[p] = e[i];
						} else {
							pd[q * np + p] = IS_L2 ? __uint_as_float(kv) : key2f(~kv);
							pi[q * np + p] = (int)(unsigned)e[i];
						}
					}
					base += __builtin_popcountll(m);
				}
				if (outD) { // exactly np distinct entries (key << 32 | id): lane p's rank = the entries below its own
					__syncthreads();
					if (lane < np) {
						const unsigned long long mine = cand[lane];
						int r = 0;
						for (int j = 0; j < np; ++j)
							r += cand[j] < mine ? 1 : 0;
						const unsigned kv = (unsigned)(mine >> 32);
						outD[q * np + r] = IS_L2 ? __uint_as_float(kv) : key2f(~kv);
						outI[q * np + r] = (long long)(unsigned)mine + label_offset;
					}
				}
				return;
			}
		}
	}
	// counts over the wave: one ballot + scalar population count per register slot (no cross-lane data movement)
	// T = the np-th smallest key: the largest T with #(key < T) < np  (bit by bit)
	unsigned T = 0u;
#pragma unroll 1
	for (int b = 31; b >= 0; --b) {
		const unsigned t = T | (1u << b);
		int cnt = 0;
#pragma unroll
		for (int j = 0; j < PL; ++j)
			cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] < t));
		if (cnt < np)
			T = t;
	}
	int less = 0;
#pragma unroll
	for (int j = 0; j < PL; ++j)
		less += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] < T));
	// rows tied at T: the np - less smallest ids (T = 0xffffffff: fewer than np candidates exist, nothing tied is taken)
	const int need = T == 0xffffffffu ? 0 : np - less;
	unsigned Tid = 0u; // ids < Tid among the tied rows are taken: the largest Tid with #(tied, id < Tid) <= need, found bitwise
#pragma unroll 1
	for (int b = 16; b >= 0; --b) {
		const unsigned t = Tid | (1u << b);
		int cnt = 0;
#pragma unroll
		for (int j = 0; j < PL; ++j) {
			const unsigned id = 4u * (64u * (unsigned)(j >> 2) + (unsigned)lane) + (unsigned)(j & 3);
			cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] == T && id < t));
		}
		if (cnt <= need)
			Tid = t;
	}
	// emit
	int base = 0;
#pragma unroll
	for (int j = 0; j < PL; ++j) {
		const unsigned id = 4u * (64u * (unsigned)(j >> 2) + (unsigned)lane) + (unsigned)(j & 3);
		const bool take = key[j] < T || (need > 0 && key[j] == T && id < Tid);
		const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
		if (take) {
			const int pos = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
			if (pos < np) {
				if (outD) {
					cand[pos] = ((unsigned long long)key[j] << 32) | id; // (np <= 256 entries: ordered below)
				} else {
					pd[q * np + pos] = IS_L2 ? __uint_as_float(key[j]) : key2f(~key[j]);
					pi[q * np + pos] = (int)id;
				}
			}
		}
		base += __builtin_popcountll(m);
	}
	if (outD) {
		const int have = base < np ? base : np;
		__syncthreads();
		for (int p = lane; p < have; p += 64) {
			const unsigned long long mine = cand[p];
			int r = 0;
			for (int j2 = 0; j2 < have; ++j2)
				r += cand[j2] < mine ? 1 : 0;
			const unsigned kv = (unsigned)(mine >> 32);
			outD[q * np + r] = IS_L2 ? __uint_as_float(kv) : key2f(~kv);
			outI[q * np + r] = (long long)(unsigned)mine + label_offset;
		}
		for (int pos = have + lane; pos < np; pos += 64) {
			outD[q * np + pos] = IS_L2 ? FLT_MAX : -FLT_MAX;
			outI[q * np + pos] = -1;
		}
		return;
	}
	for (int pos = base + lane; pos < np; pos += 64) {
		pd[q * np + pos] = IS_L2 ? FLT_MAX : -FLT_MAX;
		pi[q * np + pos] = -1;
	}
}

bool coarse_select_supported(int64_t nlist, int64_t np) {
	return nlist >= 256 && nlist <= 8192 && nlist % 4 == 0 && np >= 1 && np <= 256 && np < nlist; // (16-byte rows of D)
}
size_t coarse_select_matrix_bytes(int64_t nq, int64_t nlist) {
	return (size_t)nq * nlist * sizeof(float);
}

// D (scratch, [nq][nlist]) <- distances; pd / pi [nq][np] <- the np nearest centroids of every query, unordered
void launch_coarse_select(const float *d_x, int64_t nq, int d, const float *d_cent, int sdp, int interleaved, int64_t nlist,
                          const float *d_qn, const float *d_cn, int64_t np, int is_l2, float *d_D, float *d_pd, int32_t *d_pi,
                          hipStream_t st, float *d_outD, int64_t *d_outI, int64_t label_offset) {
	if (nq <= 0)
		return;
	const dim3 grid((unsigned)((nlist + 127) / 128), (unsigned)((nq + 127) / 128));
	if (tune().coarse_mfma >= 2 && (d & 3) == 0 && (sdp & 3) == 0) {
		// (option ivf_coarse_mfma = 3: one workgroup per tile -- the same kernel without its store/MFMA overlap, for A/B)
		const long long ntiles = (long long)grid.x * grid.y;
		const unsigned wgs = (unsigned)std::min<long long>(ntiles, tune().coarse_mfma == 3 ? ntiles : (tune().coarse_abl & 64 ? 1 : 2) * device_cu_count());
#define MVS_CM2(ILV, L2V)                                                                                                          \
	{                                                                                                                              \
		ensure_dynamic_lds((const void *)coarse_dist_mfma2_kernel<ILV, L2V>, coarse_mfma2_lds_bytes());                            \
		hipLaunchKernelGGL((coarse_dist_mfma2_kernel<ILV, L2V>), dim3(wgs), dim3(256), coarse_mfma2_lds_bytes(), st, d_x, (long long)nq, d, \
		                   d_cent, sdp, (int)nlist, d_qn, d_cn, d_D, tune().coarse_abl);                                           \
	}
		if (interleaved && is_l2)
			MVS_CM2(true, true)
		else if (interleaved)
			MVS_CM2(true, false)
		else if (is_l2)
			MVS_CM2(false, true)
		else
			MVS_CM2(false, false)
#undef MVS_CM2
	} else if (tune().coarse_mfma) {
		ensure_dynamic_lds((const void *)coarse_dist_mfma_kernel, coarse_mfma_lds_bytes());
		const long long ntiles = (long long)grid.x * grid.y;
		hipLaunchKernelGGL(coarse_dist_mfma_kernel, dim3((unsigned)std::min<long long>(ntiles, tune().coarse_persistent ? 2 * device_cu_count() : ntiles)), dim3(256), coarse_mfma_lds_bytes(), st, d_x, (long long)nq, d, d_cent, sdp,
		                   interleaved, (int)nlist, d_qn, d_cn, is_l2, d_D, tune().coarse_abl);
	}
	else
		hipLaunchKernelGGL(coarse_dist_kernel, grid, dim3(256), 0, st, d_x, (long long)nq, d, d_cent, sdp, interleaved, (int)nlist, d_qn,
		                   d_cn, is_l2, d_D);
	MVS_HIP(hipGetLastError());
#define MVS_CSEL(PL)                                                                                                               \
	{                                                                                                                              \
		if (is_l2)                                                                                                                 \
			hipLaunchKernelGGL((coarse_select_kernel<PL, true>), dim3((unsigned)nq), dim3(64), 0, st, d_D, (int)nlist, (int)np, d_pd, d_pi, d_outD, (long long *)d_outI, (long long)label_offset); \
		else                                                                                                                       \
			hipLaunchKernelGGL((coarse_select_kernel<PL, false>), dim3((unsigned)nq), dim3(64), 0, st, d_D, (int)nlist, (int)np, d_pd, d_pi, d_outD, (long long *)d_outI, (long long)label_offset); \
	}
	if (nlist <= 1024)
		MVS_CSEL(16)
	else if (nlist <= 2048)
		MVS_CSEL(32)
	else if (nlist <= 4096)
		MVS_CSEL(64)
	else
		MVS_CSEL(128)
#undef MVS_CSEL
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
