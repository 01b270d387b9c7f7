// csrc/flat_bf16.hip -- bf16x3 PREFILTER for the brute-force search + exact f32 re-scoring of its candidates.
//
// Same place in the path as flat_mfma.hip (IndexFlat::search, /root/reference/src/faiss_extension.cpp:631, BLAS branch of
// knn_L2sqr / knn_inner_product), same results bit for bit -- but the N x nq contraction, 99 % of the work, runs on the
// bf16 matrix pipe (v_mfma_f32_32x32x16_bf16: 16x the f32 MFMA rate per instruction-cycle) instead of the f32 one:
//
//   1. every f32 value is split x = hi + lo + r, hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-16 |x| (rows once, at the first
//      search after an add; queries per call);  <x,y> ~ <xh,yh> + <xh,yl> + <xl,yh>: three bf16 MFMAs per 16 dimensions,
//      f32 accumulation.  |approx - exact chain| <= E = c(d) ||x|| ||y||, c(d) derived in prefilter_cerr() below.
//   2. the fused epilogue of flat_mfma.hip (shared code, flat_fused.h) keeps, per query, the k' = k + margin best
//      APPROXIMATE values; partial lists are merged as usual.
//   3. rescore_verify_kernel recomputes the k' candidates with the oracle's arithmetic (k-ordered fmaf chain over the
//      original f32 rows, (xn + yn) - 2 ip) and PROVES per query that the candidate set contains the exact top-k:
//      if a_(k') is beyond a_(k) by more than 2E, every row whose exact value can reach the exact k-th is in the set
//      (proof at rescore_verify_kernel).  Queries that cannot be proven (heavy ties / duplicates, non-finite input, huge
//      norms) are re-run on the exact f32 kernel by the caller (FlatIndex::search_flat) -- correctness never depends
//      on the error model being tight, only on it being an upper bound.
//   4. the exact values go through the normal merge (FAISS order, inner-product tie detection), so labels AND distances
//      are those of flat_mfma.hip / oracle/orc_core.c search_blas.
//
// Kernel geometry (CDNA4): workgroup = 4 waves, wave = 64 queries (two 32-query B tiles resident as bf16 hi/lo: 128
// VGPRs), database tiles of 32 rows staged by LDS-DMA (16 KB at d = 128: [row][hi | lo], 16-byte chunks XOR-swizzled
// by row so that ds_read_b128 of one k-chunk across 32 rows is conflict free), every A fragment (hi, lo: one
// ds_read_b128 each) feeds 6 MFMAs, i.e. 1/3 LDS read per MFMA; two workgroups per CU alternate on the SIMDs so that one
// wave's epilogue VALU work overlaps the other's MFMAs.
#include "flat_fused.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) float lds_f32b;
typedef __attribute__((address_space(1))) const float glb_f32b;

constexpr int PF_QBLOCK = 256; // queries per workgroup
constexpr int PF_BN = 32;      // rows per staged tile
constexpr int PF_SLOT_PERIOD = 32; // tiles between two reads of the shared threshold slots

// ---- error model ---------------------------------------------------------------------------------------------------
// With S = sum |x_i y_i| <= ||x|| ||y|| (Cauchy-Schwarz) and u = 2^-24:
//   exact chain (oracle / f32 MFMA, d sequential fmas):      |chain - <x,y>| <= d u S
//   split x = xh + xl + xr, xh = bf16(x), xl = bf16(x - xh):  |x - xh| <= 2^-8 |x|, |xr| <= 2^-16 |x| (x - xh is exact in f32);
//        dropped terms xl yl + xr y + x yr - xr yr:           <= 3.01 * 2^-16 S   (bf16 = 8 significant bits: unit roundoff 2^-8)
//   bf16 MFMA: the 16 products of an instruction are exact in f32 (8 x 8 significant bits); the instruction returns
//        C + their sum in f32.  Its internal alignment / rounding is not documented; measured in round 5 (tests/
//        test_mfma_model_gpu.py: small terms are truncated in two stages, worst 8.8 u of the magnitudes per 32-product instruction)
//        and charged as 8 ulp-units of the magnitudes involved per 16 dimensions (rounds 2-4: 4):
//        3 d / 16 * 8 u * S (1 + 2^-8)
//   c(d) = 1.25 x (sum of the three).  The device reports the largest |approx - exact| / (||x|| ||y||) it sees among the
//   re-scored candidates (mvs_index_prefilter_stats); tests assert it stays >= 10x below c(d).  The proof in
//   rescore_verify_kernel needs an upper bound, not a tight one: a query it cannot prove is re-run on the exact kernel.
float prefilter_cerr(int d) {
	const double u = std::ldexp(1.0, -24);
	const double split = 3.01 * std::ldexp(1.0, -16);
	const double mfma = (3.0 * d / 16.0) * 8.0 * u * (1.0 + 1.0 / 256); // (8: measured, csrc/flat_collect.hip CL_MFMA_UNITS)
	const double chain = (double)d * u;
	return (float)(1.25 * (split + mfma + chain));
}

// ---- storage: rows as [hi(dp) | lo(dp)] bf16 ---------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo) {
	hi = (__bf16)x; // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
	lo = (__bf16)(x - (float)hi);
}

// src: the index's f32 row store ([n][dp], pair-interleaved when dp <= 128, see FlatGeom); one thread per (row, 8 dims)
__global__ void rows_to_bf16_kernel(const float *__restrict__ src, long long row0, long long nrows, int dp,
                                    int interleaved, unsigned short *__restrict__ dst, const float *__restrict__ norms,
                                    unsigned *__restrict__ max_norm_bits) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int g8 = dp / 8;
	if (i >= nrows * g8)
		return;
	const long long r = row0 + i / g8;
	const int c8 = (int)(i % g8);
	const float4 s0 = *(const float4 *)(src + (size_t)r * dp + c8 * 8);
	const float4 s1 = *(const float4 *)(src + (size_t)r * dp + c8 * 8 + 4);
	float v[8];
	if (!interleaved) {
		v[0] = s0.x, v[1] = s0.y, v[2] = s0.z, v[3] = s0.w, v[4] = s1.x, v[5] = s1.y, v[6] = s1.z, v[7] = s1.w;
	} else if ((r >> 4) & 1) { // stored [k1,k3,k0,k2]
		v[0] = s0.z, v[1] = s0.x, v[2] = s0.w, v[3] = s0.y, v[4] = s1.z, v[5] = s1.x, v[6] = s1.w, v[7] = s1.y;
	} else { // stored [k0,k2,k1,k3]
		v[0] = s0.x, v[1] = s0.z, v[2] = s0.y, v[3] = s0.w, v[4] = s1.x, v[5] = s1.z, v[6] = s1.y, v[7] = s1.w;
	}
	bf16x8 hi, lo;
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		__bf16 h, l;
		split_bf16(v[e], h, l);
		hi[e] = h;
		lo[e] = l;
	}
	unsigned short *row = dst + (size_t)r * 2 * dp;
	*(bf16x8 *)(row + c8 * 8) = hi;
	*(bf16x8 *)(row + dp + c8 * 8) = lo;
	if (c8 == 0) { // largest squared row norm (>= 0: the bit pattern orders like the value; NaN sorts above everything)
		// (agent-scope load: a plain one is served from this CU's vector cache as first fetched and every row would send its atomic)
		const unsigned b = __float_as_uint(norms[r]);
		if (b > __hip_atomic_load(max_norm_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			atomicMax(max_norm_bits, b);
	}
}
void launch_rows_to_bf16(const FlatGeom &g, const float *d_vecs, int64_t row0, int64_t nrows, unsigned short *d_bf,
                         const float *d_norms, unsigned *d_max_norm_bits, hipStream_t st) {
	if (nrows <= 0)
		return;
	const long long total = (long long)nrows * (g.dp / 8);
	hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_vecs, (long long)row0,
	                   (long long)nrows, g.dp, g.pair_interleaved ? 1 : 0, d_bf, d_norms, d_max_norm_bits);
	MVS_HIP(hipGetLastError());
}

// queries -> B fragments: qf[((qblk32 * KCH + ch) * 2 + part) * 64 + lane] = 8 bf16 of query qblk32*32 + (lane & 31),
// dims ch*16 + 8*(lane >> 5) + 0..7  (v_mfma_f32_32x32x16_bf16 B operand: lane l holds B[k = 8(l>>5) + j][col l & 31])
__global__ void pack_queries_bf16_kernel(const float *__restrict__ x, long long nq, int d, int kch,
                                         bf16x8 *__restrict__ qf, long long total) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; // one (qblk32, ch, lane)
	if (i >= total)
		return;
	const int lane = (int)(i & 63);
	const long long t = i >> 6;
	const int ch = (int)(t % kch);
	const long long qblk32 = t / kch;
	const long long q = qblk32 * 32 + (lane & 31);
	bf16x8 hi, lo;
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		const int kk = ch * 16 + 8 * (lane >> 5) + e;
		const float v = (q < nq && kk < d) ? x[q * d + kk] : 0.f;
		__bf16 h, l;
		split_bf16(v, h, l);
		hi[e] = h;
		lo[e] = l;
	}
	qf[((qblk32 * kch + ch) * 2 + 0) * 64 + lane] = hi;
	qf[((qblk32 * kch + ch) * 2 + 1) * 64 + lane] = lo;
}
size_t prefilter_qfrag_bytes(const FlatGeom &g, int64_t nq) {
	const int64_t nblk32 = (nq + PF_QBLOCK - 1) / PF_QBLOCK * (PF_QBLOCK / 32);
	return (size_t)nblk32 * (g.dp / 16) * 2 * 64 * 16;
}
void launch_pack_queries_bf16(const FlatGeom &g, const float *d_x, int64_t nq, void *d_qf, hipStream_t st) {
	const int64_t nblk32 = (nq + PF_QBLOCK - 1) / PF_QBLOCK * (PF_QBLOCK / 32);
	const long long total = (long long)nblk32 * (g.dp / 16) * 64;
	hipLaunchKernelGGL(pack_queries_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_x, (long long)nq,
	                   g.d, g.dp / 16, (bf16x8 *)d_qf, total);
	MVS_HIP(hipGetLastError());
}

// ---- the prefilter kernel --------------------------------------------------------------------------------------------
// MfmaArgs as in flat_mfma.hip; a.yb = bf16 row store, a.qf = bf16 query fragments, a.nqb counts 256-query blocks.
// ABL (profiling builds only, results are WRONG when != 0): bit 0 = no epilogue, bit 1 = stage only the first tile,
// bit 2 = no per-tile slot loads, bit 3 = no LDS fragment reads after the first chunk, bit 4 = epilogue fast path only
template <int KCH, bool IS_L2, bool GL, int ABL = 0>
__global__ __launch_bounds__(256, 2) void flat_bf16x3_kernel(const MfmaArgs a) {
	constexpr int DP = KCH * 16;
	constexpr int PITCH = DP * 4;                // bytes per row: hi block + lo block
	constexpr int C = PITCH / 16;                // 16-byte chunks per row (32 at d = 128)
	constexpr int TILE_BYTES = PF_BN * PITCH;    // 16 KB at d = 128
	constexpr int NDMA = TILE_BYTES / 1024;      // LDS-DMA instructions per tile (1 KB per wave-instruction)
	constexpr int DMA_PER_WAVE = NDMA / 4;
	static_assert(NDMA % 4 == 0 && DMA_PER_WAVE <= KCH, "one LDS-DMA instruction per k-chunk at most");

	extern __shared__ __attribute__((aligned(16))) float smem[];
	char *tbuf = (char *)smem;                              // [2][TILE_BYTES]
	float *nbuf = (float *)(tbuf + 2 * TILE_BYTES);         // [2][64]
	float *ld = nbuf + 2 * 64;                              // [256][k]  (absent with GL); nbuf holds 64 norms per buffer
	int *li = (int *)(ld + (GL ? 0 : PF_QBLOCK * a.k));
	float *lthr = (float *)(li + (GL ? 0 : PF_QBLOCK * a.k));
	int *lthrid = (int *)(lthr + PF_QBLOCK);
	int *lpos = lthrid + PF_QBLOCK;

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int h = lane >> 5, c = lane & 31;
	const int k = a.k;
#ifdef MVS_COUNT_EVENTS
	const unsigned long long t_kernel0 = __builtin_amdgcn_s_memtime();
#endif
	int split, qb;
	if (a.xcd_map) {
		const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
		split = (idx / a.nqb) * 8 + xcd;
		qb = idx % a.nqb;
	} else {
		split = blockIdx.x / a.nqb;
		qb = blockIdx.x % a.nqb;
	}
	const long long r_begin = (long long)split * a.split_rows;
	long long r_end = r_begin + a.split_rows;
	if (r_end > a.n)
		r_end = a.n;
	const int ntiles = r_end > r_begin ? (int)((r_end - r_begin + PF_BN - 1) / PF_BN) : 0;
	const int nwin = a.slot_stride >> 4;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;

	// the lane's two queries (one per 32-query B tile of the wave)
	int ql[2], q[2];
	bool qvalid[2];
	float thr[2], xnq[2];
	float *ldq[2];
	int *liq[2];
#pragma unroll
	for (int t = 0; t < 2; ++t) {
		ql[t] = wave * 64 + t * 32 + c;
		q[t] = qb * PF_QBLOCK + ql[t];
		qvalid[t] = q[t] < a.nq;
		thr[t] = qvalid[t] ? neutral : (IS_L2 ? -INFINITY : INFINITY);
		const size_t ob = ((size_t)split * a.nq + (qvalid[t] ? q[t] : 0)) * k;
		ldq[t] = GL ? a.pd + ob : ld + ql[t] * k;
		liq[t] = GL ? (int *)(a.pi + ob) : li + ql[t] * k;
		if (h == 0) {
			if (!GL || qvalid[t])
				for (int j = 0; j < k; ++j) {
					ldq[t][j] = neutral;
					liq[t][j] = -1;
				}
			lthr[ql[t]] = thr[t];
			lthrid[ql[t]] = -1;
			lpos[ql[t]] = 0;
		}
		xnq[t] = (IS_L2 && qvalid[t]) ? a.qn[q[t]] : 0.f;
	}

	// Two workgroups share a CU (two waves per SIMD).  Left alone they phase-lock: both in their MFMA phase (alternating
	// issue, each at half speed), then both in their epilogue + barrier with the matrix pipe idle.  The wave slot this wave
	// occupies on its SIMD (HW_ID.WAVE_ID) tells the two apart: the odd slot gets the lower issue priority (it then fills
	// the gaps the even slot's epilogues leave) and / or starts half a tile late.
	if (a.sched) {
		const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u; // HW_REG_HW_ID, WAVE_ID[3:0]
		if (a.sched & 1) {
			if (slot)
				__builtin_amdgcn_s_setprio(0);
			else
				__builtin_amdgcn_s_setprio(2);
		}
		if ((a.sched & 2) && slot)
			__builtin_amdgcn_s_sleep(24); // 24 x 64 clocks ~ one tile's MFMA phase
	}

	// B fragments, resident: [query tile][k-chunk][hi | lo]
	bf16x8 bq[2][KCH][2];
	{
		const bf16x8 *qsrc = (const bf16x8 *)a.qf;
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			const size_t qblk32 = (size_t)qb * (PF_QBLOCK / 32) + wave * 2 + t;
#pragma unroll
			for (int ch = 0; ch < KCH; ++ch) {
				bq[t][ch][0] = qsrc[((qblk32 * KCH + ch) * 2 + 0) * 64 + lane];
				bq[t][ch][1] = qsrc[((qblk32 * KCH + ch) * 2 + 1) * 64 + lane];
			}
		}
	}

	// LDS-DMA: instruction `inst` of a tile fills LDS bytes [1024 inst, +1024); lane l owns 16-byte slot 64 inst + l =
	// (row = slot / C, position p = slot % C) and fetches the row's chunk p ^ (row & 15)
	// LDS-DMA staging.  Instruction `inst` of a tile fills LDS bytes [1024 inst, +1024); lane l owns 16-byte slot
	// S = 64 inst + l = (row r = S / C, position p = S % C) and fetches the row's chunk p ^ (r & 15).  Wave w issues
	// inst = 4 i + w, so its rows are r = i * RPI + r0 (RPI = 256 / C): the per-lane byte offset is loop invariant up to
	// the parity of i (for C = 32 the row's bit 3 flips) -> two VGPRs, and every issue is ONE instruction with a uniform
	// base (SGPR pair).  Tiles past the split's / database's end are fetched as well (the stores own 64 rows of zeroed
	// padding, FlatIndex::ensure_bf16_rows): no clamp, no branch around a vector-memory instruction.
	constexpr int RPI = 256 / C;
	unsigned dma_off[2];
	{
		const int r0 = wave * (64 / C) + lane / C, p = lane % C;
		dma_off[0] = (unsigned)(r0 * PITCH + ((p ^ (r0 & 15)) * 16));
		dma_off[1] = (unsigned)(r0 * PITCH + ((p ^ ((r0 + RPI) & 15)) * 16));
	}
	auto dma_issue = [&](int u, int i) {
		const char *base = (const char *)a.yb + ((size_t)(r_begin + (long long)u * PF_BN) + (size_t)i * RPI) * PITCH; // uniform
		__builtin_amdgcn_global_load_lds((glb_f32b *)(base + dma_off[i & 1]),
		                                 (lds_f32b *)(smem + ((u & 1) * TILE_BYTES + (i * 4 + wave) * 1024) / 4), 16, 0, 0);
	};
	// norms of rows row0 .. row0 + 63 (every wave writes the same 64 floats; the norm array is padded as well)
	auto dma_norms = [&](int u) {
		if (IS_L2) {
			const float *base = a.yn + (r_begin + (long long)u * PF_BN); // uniform
			__builtin_amdgcn_global_load_lds((glb_f32b *)(base + lane), (lds_f32b *)(smem + (2 * TILE_BYTES) / 4 + (u & 1) * 64), 4,
			                                 0, 0);
		}
	};

	SlotBound sbound[2];
	unsigned gkey[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}; // no shared bound yet
	if (ntiles > 0) {
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i)
			dma_issue(0, i);
		dma_norms(0);
	}
	__syncthreads();

	// read address of (row c, chunk index ci = part * 2 KCH + 2 ch + h): c * PITCH + ((ci ^ (c & 15)) * 16).  h is bit 0 of
	// ci and (part, ch) only touch bits >= 1, and the XOR is bitwise, so with rbase = c * PITCH | (((c & 15) ^ h) * 16) the
	// address is rbase ^ ((part * 2 KCH + 2 ch) * 16): ONE v_xor with a constant per read
	const unsigned rbase = (unsigned)(c * PITCH) | (unsigned)((((c & 15) ^ h) & 15) * 16);

	for (int u = 0; u < ntiles; ++u) {
		f32x16 acc[2][1];
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int r = 0; r < 16; ++r)
				acc[t][0][r] = 0.f;
		const char *Abase = tbuf + ((ABL & 2) ? 0 : (u & 1)) * TILE_BYTES;
		bf16x8 af[2][2]; // [ring][hi | lo]
		auto read_a = [&](int ch, int slot) {
			unsigned rb = rbase;
			MVS_OPAQUE_VGPR(rb); // recompute the two addresses per read instead of keeping 2 KCH of them live
			const unsigned o0 = rb ^ (unsigned)((2 * ch) * 16);
			const unsigned o1 = rb ^ (unsigned)((2 * KCH + 2 * ch) * 16);
			af[slot][0] = *(const bf16x8 *)(Abase + o0);
			af[slot][1] = *(const bf16x8 *)(Abase + o1);
		};
		// Shared threshold slots: every PF_SLOT_PERIOD tiles (a tile is 5x shorter than the f32 kernel's) the wave fetches the
		// current 16-slot window of its 2 x 32 queries and WAITS for it (one L2 round trip; the other workgroup's wave keeps
		// the SIMD busy).  Keeping the words in flight across the tile instead costs 16 VGPRs the kernel does not have (at the
		// 256-register limit hipcc serialises the loads through one register pair), and a wait for them later in the tile is
		// a vmcnt(0) that also covers the next tile's LDS-DMA.
		// (most insertions of a (query, split) pair happen in its first few hundred rows: refresh every other tile there)
		const int period = u < a.k ? 2 : (u < 256 ? 8 : PF_SLOT_PERIOD);
		if (a.nclass == 32 && ((ABL & 4) ? u == 0 : (u % period) == 0)) {
			// 32 row classes for k' <= 16 lists: the bound is the k'-th SMALLEST of the 32 per-class minima (k' distinct rows
			// at least that good exist), ~the 1.4 k'-th best row seen so far by anyone; the maximum over k' classes that the
			// generic scheme uses is ~the 3 k'-th best (coupon collecting), i.e. ~2.4x more rows pass the filter and each of
			// them stalls the workgroup's other waves at the tile barrier.  Lanes l and l + 32 hold 16 keys each; ten
			// bisection steps on the order-preserving keys find the k'-th smallest to 1/1024 of the key range (the upper end
			// of the final interval is returned: still a valid bound).
#pragma unroll 1
			for (int t = 0; t < 2; ++t) {
				const unsigned long long *src =
				    (const unsigned long long *)(a.gslot + (size_t)(qvalid[t] ? q[t] : 0) * a.slot_stride + 16 * h);
				unsigned long long w[8];
#pragma unroll
				for (int j = 0; j < 8; ++j)
					w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				unsigned key[16];
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					key[2 * j] = (unsigned)w[j];
					key[2 * j + 1] = (unsigned)(w[j] >> 32);
				}
				unsigned lo = key[0], hi = key[0];
#pragma unroll
				for (int j = 1; j < 16; ++j) {
					lo = lo < key[j] ? lo : key[j];
					hi = hi > key[j] ? hi : key[j];
				}
				{
					const unsigned olo = (unsigned)__shfl_xor((int)lo, 32), ohi = (unsigned)__shfl_xor((int)hi, 32);
					lo = lo < olo ? lo : olo;
					hi = hi > ohi ? hi : ohi;
				}
#pragma unroll 1
				for (int it = 0; it < 10 && lo < hi; ++it) {
					const unsigned mid = lo + ((hi - lo) >> 1);
					int cnt = 0;
#pragma unroll
					for (int j = 0; j < 16; ++j)
						cnt += key[j] <= mid ? 1 : 0;
					cnt += __shfl_xor(cnt, 32);
					if (cnt >= k)
						hi = mid;
					else
						lo = mid + 1;
				}
				gkey[t] = hi;
			}
		} else if ((ABL & 4) ? u == 0 : (u % period) == 0) {
			const int window = (u / period) % nwin;
			SlotRegs sr[2];
#pragma unroll
			for (int t = 0; t < 2; ++t)
				slots_prefetch(sr[t], a.gslot + (size_t)(qvalid[t] ? q[t] : 0) * a.slot_stride, window, h);
#pragma unroll
			for (int t = 0; t < 2; ++t) // all eight loads are issued before the first is consumed
				asm volatile("" : "+v"(sr[t].w[0]), "+v"(sr[t].w[1]), "+v"(sr[t].w[2]), "+v"(sr[t].w[3]));
#pragma unroll
			for (int t = 0; t < 2; ++t)
				gkey[t] = slots_update(sbound[t], slots_reduce(sr[t]), window, nwin);
		}
		read_a(0, 0);
#pragma unroll
		for (int ch = 0; ch < KCH; ++ch) {
			__builtin_amdgcn_sched_barrier(0);
			if (ch + 1 < KCH) {
				if (ABL & 8) {
					af[(ch + 1) & 1][0] = af[ch & 1][0];
					af[(ch + 1) & 1][1] = af[ch & 1][1];
				} else {
					read_a(ch + 1, (ch + 1) & 1);
				}
			}
			if (!(ABL & 2)) {
				if (ch < DMA_PER_WAVE)
					dma_issue(u + 1, ch);
				if (ch == 0)
					dma_norms(u + 1);
			}
			__builtin_amdgcn_sched_barrier(0);
			const bf16x8 ah = af[ch & 1][0], al = af[ch & 1][1];
#pragma unroll
			for (int t = 0; t < 2; ++t) {
				acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[t][ch][0], acc[t][0], 0, 0, 0);
				acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[t][ch][1], acc[t][0], 0, 0, 0);
				acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[t][ch][0], acc[t][0], 0, 0, 0);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		const long long row0 = r_begin + (long long)u * PF_BN;
		const int nvalid = (int)((r_end - row0) < PF_BN ? (r_end - row0) : PF_BN);
		// Row norms of this tile, LDS -> registers by hand: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it
		// compiles while an LDS-DMA is in flight (it cannot tell the targets apart), i.e. it would wait here for the NEXT
		// tile's staging.  These norms were staged a tile ago and are ordered by the tile-end vmcnt(0) + barrier.
		float4 yn4[4];
		if (IS_L2 && !(ABL & 1)) {
			const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32b *)(nbuf + ((ABL & 2) ? 0 : (u & 1)) * 64 + 4 * h));
			asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
			             "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(yn4[0]), "=&v"(yn4[1]), "=&v"(yn4[2]), "=&v"(yn4[3])
			             : "v"(nb_lds)
			             : "memory");
		}
#pragma unroll
		for (int t = 0; t < 2; ++t) {
			if (ABL & 1) {
				MVS_KEEP_VGPR(acc[t][0]);
				continue;
			}
			tile_epilogue<1, IS_L2, (ABL & 16) != 0, false, false, true, GL ? 2 : 1, false>(acc[t], nullptr, row0, nvalid, xnq[t], thr[t], qvalid[t], gkey[t],
			                                                   a.gslot + (size_t)(qvalid[t] ? q[t] : 0) * a.slot_stride, ldq[t],
			                                                   liq[t], k, lthr + ql[t], lthrid + ql[t], lpos + ql[t], h, nullptr,
			                                                   yn4, a.nclass);
		}
		__syncthreads(); // also drains this tile's LDS-DMA (vmcnt(0)) before the next tile reads it
	}

#ifdef MVS_COUNT_EVENTS
	if (lane == 0)
		atomicAdd(&g_dbg_counters[2], __builtin_amdgcn_s_memtime() - t_kernel0);
#endif
	if (!GL && h == 0) {
#pragma unroll
		for (int t = 0; t < 2; ++t)
			if (qvalid[t]) {
				const size_t ob = ((size_t)split * a.nq + q[t]) * k;
				for (int j = 0; j < k; ++j) {
					a.pd[ob + j] = ldq[t][j];
					a.pi[ob + j] = liq[t][j];
				}
			}
	}
}

static size_t pf_lds_bytes(const FlatGeom &g, int64_t k, bool gl) {
	return (size_t)2 * PF_BN * g.dp * 4 + 2 * 64 * 4 + (gl ? 0 : (size_t)PF_QBLOCK * k * 8) + PF_QBLOCK * 12;
}

bool prefilter_supported(const FlatGeom &g) {
	return g.nch == 1 && (g.dp == 64 || g.dp == 128);
}

#ifdef MVS_COUNT_EVENTS
extern "C" void mvs_debug_counters(unsigned long long *out, int reset) {
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_counters), 32);
	if (reset) {
		unsigned long long z[4] = {0, 0, 0, 0};
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_counters), z, 32);
	}
}
#endif

FlatSearchPlan plan_prefilter(const FlatGeom &g, int64_t nq, int64_t n, int64_t kp) {
	FlatSearchPlan p;
	p.nqb = (int)((nq + PF_QBLOCK - 1) / PF_QBLOCK);
	const int64_t ntiles = (n + PF_BN - 1) / PF_BN;
	const int64_t slots = 2 * 256;
	int64_t max_split = std::max<int64_t>(1, ntiles / 256); // >= 8192 rows per split: the cold start of a (query, split)
	                                                       // pair costs as much as ~100 tiles here
	max_split = std::min<int64_t>(max_split, (int64_t)(150 * 1024 / 8) / std::max<int64_t>(kp, 1) - 1);
	max_split = std::min<int64_t>(max_split, 512);
	int64_t nsplit = 1;
	p.xcd_map = false;
	if (tune().pf_nsplit > 0) {
		nsplit = tune().pf_nsplit;
	} else if (max_split >= 8) {
		double best = -1;
		for (int64_t s = 8; s <= max_split; s += 8) {
			const int64_t w = s * p.nqb;
			if (w < slots && s + 8 <= max_split)
				continue;
			const int64_t rounds = (w + slots - 1) / slots;
			double eff = (double)w / (double)(rounds * slots);
			if (rounds < 3)
				eff -= 0.03 * (3 - rounds);
			eff -= 2e-4 * s; // at equal fill prefer fewer splits (fewer cold starts)
			if (eff > best) {
				best = eff;
				nsplit = s;
			}
		}
	} else {
		nsplit = max_split;
	}
	if (nsplit >= 8 && nsplit % 8 == 0)
		p.xcd_map = true;
	const int64_t tiles_per_split = (ntiles + nsplit - 1) / nsplit;
	p.split_rows = tiles_per_split * PF_BN;
	p.nsplit = (int)nsplit;
	p.grid = p.nqb * p.nsplit;
	p.global_lists = kp > 20; // 256 queries x 20 x 8 B = 40 KB of LDS lists still leaves two workgroups per CU
	p.lds_bytes = pf_lds_bytes(g, kp, p.global_lists);
	return p;
}

int flat_mfma_slot_stride(int64_t k);
__global__ void init_gslot_kernel(unsigned *g, long long total, int stride, int k, int is_l2);

template <int KCH>
static void launch_pf_inst(int metric, bool gl, const MfmaArgs &a, const FlatSearchPlan &p, hipStream_t st) {
#define MVS_PF(L2, GLV)                                                                                                \
	{                                                                                                                  \
		auto kern = flat_bf16x3_kernel<KCH, L2, GLV>;                                                                  \
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));                                                 \
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);                                         \
	}
	if (metric == METRIC_L2) {
		if (gl)
			MVS_PF(true, true)
		else
			MVS_PF(true, false)
	} else {
		if (gl)
			MVS_PF(false, true)
		else
			MVS_PF(false, false)
	}
#undef MVS_PF
	MVS_HIP(hipGetLastError());
}

void launch_prefilter(const FlatGeom &g, const FlatSearchPlan &p, int metric, const void *d_qf, const float *d_qnorm,
                      int64_t nq, const unsigned short *d_rows_bf, const float *d_norms, int64_t n, int64_t kp, float *d_pd,
                      int32_t *d_pi, unsigned *d_gthr, hipStream_t st) {
	if (nq <= 0)
		return;
	const int nclass = (kp <= 16 && tune().pf_classes32) ? 32 : (int)kp; // 32 classes + k'-th smallest for the common small-k case
	const int stride = flat_mfma_slot_stride(nclass);
	const long long gtotal = (long long)nq * stride;
	hipLaunchKernelGGL(init_gslot_kernel, dim3((unsigned)((gtotal + 255) / 256)), dim3(256), 0, st, d_gthr, gtotal, stride,
	                   nclass, metric == METRIC_L2 ? 1 : 0);
	MfmaArgs a;
	memset(&a, 0, sizeof a);
	a.gslot = d_gthr;
	a.slot_stride = stride;
	a.nclass = nclass;
	a.sched = tune().pf_sched;
	a.qf = (const float *)d_qf;
	a.qn = d_qnorm;
	a.yb = (const float *)d_rows_bf;
	a.yn = d_norms;
	a.pd = d_pd;
	a.pi = d_pi;
	a.n = n;
	a.split_rows = p.split_rows;
	a.nq = (int)nq;
	a.k = (int)kp;
	a.nqb = p.nqb;
	a.nsplit = p.nsplit;
	a.dp = g.dp;
	a.nch = 1;
	a.xcd_map = p.xcd_map ? 1 : 0;
	// Seeding pre-pass: the same kernel over the first rows only (8 splits x >= 2048 rows).  It leaves, in the shared
	// class slots, the k'-th best of a ~16k-row sample for every query, so that the 512 workgroups of the main launch's
	// FIRST round do not all start with no bound at all (their cold-start insertions are what the other three waves of
	// a workgroup wait for at the tile barrier).  Its partial lists are overwritten by the main launch.
	if (tune().pf_seed > 0 && n >= (int64_t)64 * tune().pf_seed && !tune().pf_abl) {
		MfmaArgs s = a;
		const int64_t rows = std::max<int64_t>(PF_BN, (int64_t)tune().pf_seed / 8 / PF_BN * PF_BN);
		s.n = rows * 8;
		s.split_rows = rows;
		s.nsplit = 8;
		s.xcd_map = 1;
		FlatSearchPlan ps = p;
		ps.nsplit = 8;
		ps.grid = p.nqb * 8;
		if (g.dp == 128)
			launch_pf_inst<8>(metric, p.global_lists, s, ps, st);
		else
			launch_pf_inst<4>(metric, p.global_lists, s, ps, st);
	}
#ifdef MVS_PROFILING
	if (g.dp == 128 && metric == METRIC_L2 && !p.global_lists && tune().pf_abl) {
#define MVS_PF_ABL(N)                                                                                                  \
	if (tune().pf_abl == N) {                                                                                               \
		auto kern = flat_bf16x3_kernel<8, true, false, N>;                                                             \
		ensure_dynamic_lds((const void *)kern, (size_t)(p.lds_bytes));                                                 \
		hipLaunchKernelGGL(kern, dim3(p.grid), dim3(256), p.lds_bytes, st, a);                                         \
	}
		MVS_PF_ABL(1) MVS_PF_ABL(2) MVS_PF_ABL(3) MVS_PF_ABL(7) MVS_PF_ABL(15) MVS_PF_ABL(16)
#undef MVS_PF_ABL
		MVS_HIP(hipGetLastError());
		return;
	}
#endif
	if (g.dp == 128)
		launch_pf_inst<8>(metric, p.global_lists, a, p, st);
	else
		launch_pf_inst<4>(metric, p.global_lists, a, p, st);
}

// ---- exact re-scoring + proof ------------------------------------------------------------------------------------------
// One wave per query, lane j <-> candidate j (kp <= 64).  ca / ci: the merged approximate top-kp (value, row), best first.
//   exact_j = the oracle's value of row ci[j]: ip = fmaf chain in k order over the ORIGINAL f32 row; L2: max(0, (xn+yn) - 2 ip)
// Proof obligation (kk = k, or k + 1 when the caller needs the (k+1)-th value for tie detection): with a = approximate,
// D = exact, |a - D| <= E for every row of this query:
//   the kk rows with the best a have D within E of their a, so the exact kk-th best value T satisfies T <= a_(kk) + E
//   (L2; mirrored for IP).  A row of the exact top-kk has D <= T, hence a <= D + E <= a_(kk) + 2E.  If a_(kp) > a_(kk) + 2E
//   such a row ranks before position kp in the approximate order, i.e. it IS one of the kp candidates.
// If the list is not full, every comparable row of the database is a candidate and nothing needs proving.
// Queries that fail are appended to fail_q (the caller re-runs them on the exact kernel).
template <bool IS_L2>
__global__ __launch_bounds__(64) void rescore_verify_kernel(const float *__restrict__ ca, const long long *__restrict__ ci,
                                                           int kp, int kk, const float *__restrict__ x, int d,
                                                           const float *__restrict__ vecs, int dp, int interleaved,
                                                           const float *__restrict__ norms, const float *__restrict__ qn,
                                                           const unsigned *__restrict__ max_norm_bits, float cerr,
                                                           float *__restrict__ pd1, int *__restrict__ pi1,
                                                           int *__restrict__ fail_cnt, int *__restrict__ fail_q,
                                                           unsigned *__restrict__ max_rel_err_bits) {
	const long long q = blockIdx.x;
	const int j = threadIdx.x;
	const float neutral = IS_L2 ? FLT_MAX : -FLT_MAX;
	long long row = -1;
	float av = neutral;
	if (j < kp) {
		row = ci[q * kp + j];
		av = ca[q * kp + j];
	}
	float ex = neutral;
	if (row >= 0) {
		const float *y = vecs + (size_t)row * dp;
		const float *xq = x + q * d;
		const bool odd = interleaved && ((row >> 4) & 1);
		float ip = 0.f;
		for (int g4 = 0; g4 < d; g4 += 4) {
			const float4 s = *(const float4 *)(y + g4);
			float v0, v1, v2, v3;
			if (!interleaved)
				v0 = s.x, v1 = s.y, v2 = s.z, v3 = s.w;
			else if (odd)
				v0 = s.z, v1 = s.x, v2 = s.w, v3 = s.y;
			else
				v0 = s.x, v1 = s.z, v2 = s.y, v3 = s.w;
			ip = fmaf(xq[g4], v0, ip);
			if (g4 + 1 < d)
				ip = fmaf(xq[g4 + 1], v1, ip);
			if (g4 + 2 < d)
				ip = fmaf(xq[g4 + 2], v2, ip);
			if (g4 + 3 < d)
				ip = fmaf(xq[g4 + 3], v3, ip);
		}
		if (IS_L2) {
			ex = fmaf(-2.0f, ip, qn[q] + norms[row]);
			ex = ex < 0.f ? 0.f : ex; // FAISS: if (dis < 0) dis = 0
		} else {
			ex = ip;
		}
	}
	if (j < kp) {
		pd1[q * kp + j] = ex;
		pi1[q * kp + j] = (int)row;
	}
	{ // diagnostics: observed |approx - exact| of the inner product, relative to ||x|| ||y|| (the quantity c(d) bounds)
		float rel = 0.f;
		if (row >= 0 && isfinite(av) && isfinite(ex)) {
			float xx = IS_L2 ? qn[q] : 0.f;
			if (!IS_L2) {
				for (int t = 0; t < d; ++t)
					xx = fmaf(x[q * d + t], x[q * d + t], xx);
			}
			const float den = sqrtf(xx) * sqrtf(norms[row]);
			if (den > 0.f)
				rel = fabsf(av - ex) * (IS_L2 ? 0.5f : 1.f) / den;
		}
		for (int o = 32; o >= 1; o >>= 1)
			rel = fmaxf(rel, __shfl_xor(rel, o));
		if (j == 0 && rel > 0.f && __float_as_uint(rel) > __hip_atomic_load(max_rel_err_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			atomicMax(max_rel_err_bits, __float_as_uint(rel));
	}
	// proof
	const unsigned long long have = __builtin_amdgcn_ballot_w64(row >= 0);
	const int navail = __popcll(have);
	if (navail >= kp && kk < kp) {
		const float a_kk = __shfl(av, kk - 1), a_kp = __shfl(av, kp - 1);
		const float xn = IS_L2 ? qn[q] : 0.f;
		float qnorm2 = 0.f;
		if (!IS_L2) { // inner product: the query norm is not an input of the search; one lane-strided chain is enough
			float s = 0.f;
			for (int t = j; t < d; t += 64)
				s = fmaf(x[q * d + t], x[q * d + t], s);
			for (int o = 32; o >= 1; o >>= 1)
				s += __shfl_xor(s, o);
			qnorm2 = s * 1.0001f;
		} else {
			qnorm2 = xn;
		}
		const float ymax2 = __uint_as_float(*max_norm_bits);
		const float e_ip = cerr * sqrtf(qnorm2) * sqrtf(ymax2) + 1e-35f;
		bool ok;
		if (IS_L2) {
			const float e = 2.f * e_ip + 4.8e-7f * (xn + ymax2); // + the roundings of (xn + yn) - 2 ip on both sides
			ok = a_kp > a_kk + 2.f * e;
		} else {
			ok = a_kp < a_kk - 2.f * e_ip;
		}
		if (!ok && j == 0) // (false for NaN / inf bounds as well)
			fail_q[atomicAdd(fail_cnt, 1)] = (int)q;
	} else if (navail >= kp && j == 0) {
		fail_q[atomicAdd(fail_cnt, 1)] = (int)q; // kp == kk: no margin at all (never configured that way)
	}
}

void launch_rescore_verify(int metric, const float *d_ca, const int64_t *d_ci, int64_t nq, int kp, int kk, const float *d_x,
                           const FlatGeom &g, const float *d_vecs, const float *d_norms, const float *d_qn,
                           const unsigned *d_max_norm_bits, float *d_pd1, int32_t *d_pi1, int *d_fail_cnt, int *d_fail_q,
                           unsigned *d_max_rel_err_bits, hipStream_t st) {
	if (nq <= 0)
		return;
	const float cerr = prefilter_cerr(g.d);
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(rescore_verify_kernel<true>, dim3((unsigned)nq), dim3(64), 0, st, d_ca, (const long long *)d_ci, kp,
		                   kk, d_x, g.d, d_vecs, g.dp, g.pair_interleaved ? 1 : 0, d_norms, d_qn, d_max_norm_bits, cerr, d_pd1,
		                   d_pi1, d_fail_cnt, d_fail_q, d_max_rel_err_bits);
	else
		hipLaunchKernelGGL(rescore_verify_kernel<false>, dim3((unsigned)nq), dim3(64), 0, st, d_ca, (const long long *)d_ci,
		                   kp, kk, d_x, g.d, d_vecs, g.dp, g.pair_interleaved ? 1 : 0, d_norms, d_qn, d_max_norm_bits, cerr,
		                   d_pd1, d_pi1, d_fail_cnt, d_fail_q, d_max_rel_err_bits);
	MVS_HIP(hipGetLastError());
}

// results of the re-run queries back into their rows: D[fq[f]] = Df[f]
__global__ void scatter_rows_kernel(const int *__restrict__ fq, int nf, int k, const float *__restrict__ Df,
                                    const long long *__restrict__ If, float *__restrict__ D, long long *__restrict__ I) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nf * k)
		return;
	const int f = i / k, j = i - f * k;
	D[(long long)fq[f] * k + j] = Df[i];
	I[(long long)fq[f] * k + j] = If[i];
}
__global__ void gather_query_rows_kernel(const float *__restrict__ x, int d, const int *__restrict__ fq, int nf,
                                         float *__restrict__ xf) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (long long)nf * d)
		return;
	const int f = (int)(i / d), c = (int)(i - (long long)f * d);
	xf[i] = x[(long long)fq[f] * d + c];
}
void launch_gather_query_rows(const float *d_x, int d, const int *d_fq, int nf, float *d_xf, hipStream_t st) {
	if (nf <= 0)
		return;
	const long long total = (long long)nf * d;
	hipLaunchKernelGGL(gather_query_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_x, d, d_fq, nf,
	                   d_xf);
	MVS_HIP(hipGetLastError());
}
void launch_scatter_rows(const int *d_fq, int nf, int64_t k, const float *d_Df, const int64_t *d_If, float *d_D,
                         int64_t *d_I, hipStream_t st) {
	if (nf <= 0)
		return;
	const long long total = (long long)nf * k;
	hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_fq, nf, (int)k, d_Df,
	                   (const long long *)d_If, d_D, (long long *)d_I);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
