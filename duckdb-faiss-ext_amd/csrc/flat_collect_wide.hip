// csrc/flat_collect_wide.hip -- the bf16 coarse filter of flat_collect.hip for 128 < d <= 1024.
//
// Same argument, same bound, same candidate stream and re-scoring (see flat_collect.hip); what changes is the geometry.  The
// query fragments of a wave must stay in registers for the whole scan (re-streaming them costs more L2 bandwidth than the
// matrix pipe saves), and 128 queries x 512 dims do not fit: a wave keeps 32 * QT queries (QT = 2 up to d = 256, 1 beyond) x
// all KB k-blocks = 8 * QT * KB VGPRs (128 at d = 256 and 512), a workgroup 128 * QT queries, so the database is re-read by
// nq / (128 QT) query blocks instead of nq / 512 -- the scan becomes bound by L2 -> LDS traffic rather than by the matrix
// pipe, and still several times faster than the f32 kernel (which serves d > 128 otherwise).  512 < d <= 768: 32 queries x
// 768 dims = 192 VGPRs of fragments spill -- flat_bf16_ksplit_kernel below splits the k dimension over a wave pair.
//   rows      16 per tile (one MFMA row block), staged 1-3 tiles per barrier (<= 24 KB) by LDS-DMA, chunks XOR-swizzled by
//             row within aligned groups of 16 chunks
//   A         fragments read on demand, two k-blocks ahead (hand-written ds_read_b128, a ring of 4 VGPR quads)
//   per tile  QT passes of 2 * KB MFMAs (two accumulators, the chain starting at C = beta(row)), then the running maxima and
//             the rare path of that pass (not overlapped: 10 vector instructions per 2 * KB MFMAs)
#include "flat_collect.h"

#include <algorithm>
#include <cstring>

namespace mvs {

typedef float f32x4w __attribute__((ext_vector_type(4)));

// NC: row classes per query (16; 32 for 16 < kk <= 32 -- csrc/flat_collect.hip)
template <int KB, int QT, int NCBP, bool IS_L2, bool COLLECT, int NC = 16>
__global__ __launch_bounds__(256, 2) void flat_bf16_wide_kernel(const CollectArgs a) {
	constexpr int PITCH = 64 * KB;            // bytes per row
	constexpr int C = 4 * KB;                 // 16-byte chunks per row (a multiple of 16)
	constexpr int RT = 16;                    // rows per tile
	constexpr int TILE_BYTES = RT * PITCH;    // 1 KB * KB
	constexpr int WSUB = KB >= 16 ? 1 : (KB >= 12 ? 2 : 3); // tiles per staged block (16 - 24 KB)
	constexpr int STAGE_BYTES = WSUB * TILE_BYTES;
	constexpr int DMA_PER_WAVE = STAGE_BYTES / 4096; // 1 KB per wave-instruction, four waves
	constexpr int QW = 16 * NCBP * QT, QB = 4 * QW; // queries per wave (QT passes of NCBP column blocks) / workgroup
	static_assert(NCBP == 2 || NCBP == 3, "column blocks per pass");
	static_assert(KB % 4 == 0 && STAGE_BYTES % 4096 == 0 && WSUB * RT <= 64, "geometry");

	extern __shared__ __attribute__((aligned(16))) float smem[];
	char *tbuf = (char *)smem;                                        // [2][STAGE_BYTES]
	float *nbuf = (float *)(tbuf + 2 * STAGE_BYTES);                  // [2][64] beta of the staged rows
	unsigned long long *qbuf = (unsigned long long *)(nbuf + 2 * 64); // [CL_QCAP] candidate queue
	float *cqtab = (float *)(qbuf + CL_QCAP);                         // [4 waves][QT][16 c][4]: pass bound of every query (NCBP used)
	unsigned *qctl = (unsigned *)(cqtab + 4 * QT * 16 * 4);                        // [0] queue fill, [2..3] flush base

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
	const int nblocks = r_end > r_begin ? (int)((r_end - r_begin + WSUB * RT - 1) / (WSUB * RT)) : 0; // staged blocks
	if (tid == 0)
		qctl[0] = 0u;
	const int qw = qb * QB + wave * QW;

	// B fragments, resident: [column block][k-block]
	bf16x8 bq[NCBP * QT][KB];
	{
		const bf16x8 *qsrc = (const bf16x8 *)a.qf;
#pragma unroll
		for (int cb = 0; cb < NCBP * QT; ++cb) {
			const size_t qblk16 = (size_t)qb * (QB / 16) + wave * (NCBP * QT) + cb;
#pragma unroll
			for (int kb = 0; kb < KB; ++kb)
				bq[cb][kb] = qsrc[(qblk16 * KB + kb) * 64 + lane];
		}
	}

	// LDS-DMA: instruction inst = 4 i + wave of a staged block fills LDS bytes [1024 inst, +1024); lane l owns 16-byte slot
	// S = 64 inst + l = (row r = S / C, position p = S % C) and fetches the row's chunk (p & ~15) | ((p & 15) ^ (r & 15))
	auto dma_block = [&](int u) {
		const char *base = (const char *)a.yb + (size_t)(r_begin + (long long)u * (WSUB * RT)) * PITCH; // uniform
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i) {
			const int inst = 4 * i + wave;
			const int S = 64 * inst + lane, r = S / C, p = S - r * C;
			const unsigned off = (unsigned)(r * PITCH + (((p & ~15) | ((p & 15) ^ (r & 15))) * 16));
			__builtin_amdgcn_global_load_lds((glb_f32c *)(base + off),
			                                 (lds_f32c *)(smem + ((u & 1) * STAGE_BYTES + inst * 1024) / 4), 16, 0, 0);
		}
		const float *bb = a.yn + (r_begin + (long long)u * (WSUB * RT)); // uniform
		__builtin_amdgcn_global_load_lds((glb_f32c *)(bb + lane), (lds_f32c *)(smem + (2 * STAGE_BYTES) / 4 + (u & 1) * 64), 4, 0, 0);
	};
	if (nblocks > 0)
		dma_block(0);
	__syncthreads();

	// A fragment (k-block kb) of the tile's 16 rows: row c, chunk 4 kb + hq -> byte c * PITCH + 256 (kb >> 2) + (rb16 ^ (64 (kb & 3)))
	// with rb16 = ((hq ^ c) & 15) * 16  (4 (kb & 3) and hq occupy disjoint bits of the chunk number's low nibble)
	const unsigned rbase = (unsigned)(c * PITCH) + (unsigned)(((hq ^ c) & 15) * 16);
	const unsigned qcnt_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned *)qctl);
	const unsigned qbuf_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned long long *)qbuf);
	const unsigned cq_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)cqtab) + (unsigned)((wave * QT * 16 + c) * 16);

	auto rare = [&](const f32x4w (&sv)[NCBP], int t, bool any_t, f32x4n cqv, long long row0, int nvalid) {
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(any_t) == 0ull, 1)) // (hot path = fall-through: no taken branch per half tile)
			return;
		int qo = qw;
		MVS_OPAQUE_VGPR(qo); // (keeps the per-query addresses of this path out of the hot loop's registers)
#pragma unroll
		for (int i = 0; i < NCBP; ++i) {
			const int q = qo + 16 * NCBP * t + 16 * i + c;
			const float c0 = cqv[i];
			unsigned m = 0u;
			if (any_t) {
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (4 * hq + r < nvalid && sv[i][r] >= c0)
						m |= 1u << r;
			}
			if (a.rowmask && m != 0u) { // IDSelector: rejected rows are neither candidates nor evidence for the bound
				const unsigned long long rr = (unsigned long long)(row0 + 4 * hq);
				m &= (unsigned)(((const unsigned *)a.rowmask)[rr >> 5] >> (rr & 31u));
			}
			while (m != 0u) {
				const int j = __builtin_ctz(m);
				m &= m - 1u;
				const float lo = (j & 1) ? sv[i][1] : sv[i][0];
				const float hi = (j & 1) ? sv[i][3] : sv[i][2];
				const float v = (j & 2) ? hi : lo;
				const unsigned row = (unsigned)(row0 + 4 * hq + j);
				typedef __attribute__((address_space(1))) unsigned *GU;
				__hip_atomic_fetch_min((GU)(a.gslot + (size_t)q * NC) + (row & (unsigned)(NC - 1)), skey(v), __ATOMIC_RELAXED,
				                       __HIP_MEMORY_SCOPE_AGENT);
				if (COLLECT) {
					unsigned pos;
					const unsigned one = 1u;
					asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(qcnt_lds), "v"(one) : "memory");
					const unsigned long long ent = ((unsigned long long)(unsigned)q << 32) | row;
					if (pos < (unsigned)CL_QCAP) {
						asm volatile("ds_write_b64 %0, %1" ::"v"(qbuf_lds + 8u * pos), "v"(ent) : "memory");
					} else { // a burst beyond the queue: straight to the stream (by hand, wait included: flat_collect.hip)
						unsigned long long gp;
						const unsigned long long one64 = 1ull;
						typedef __attribute__((address_space(1))) unsigned long long *GUL;
						asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
						             : "=&v"(gp)
						             : "v"((GUL)a.stream_cnt), "v"(one64)
						             : "memory");
						if ((long long)gp < a.stream_cap)
							*((GUL)a.stream + gp) = ent;
					}
				}
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	};

	for (int u = 0; u < nblocks; ++u) {
		// (the cadence of flat_collect.hip -- every 64, 256, 1024, 4096 rows -- in staged blocks of WSUB * 16 rows)
		constexpr int PS = 64 / (WSUB * RT) > 0 ? 64 / (WSUB * RT) : 1;
		const int pb = (a.opt >> 2) & 3, psh = pb == 1 ? 0 : (pb == 0 ? 1 : pb); // (as in flat_collect.hip: bits 2..3 of cl_ksplit_opt)
		const int period = a.opt & 2 ? (u < 4 ? 1 : (u < 32 ? 4 : (u < 256 ? 16 : 64)))
		                             : (u < 4 * PS ? PS : (u < 32 * PS ? (4 * PS) << psh : (u < 256 * PS ? (16 * PS) << psh : (64 * PS) << psh)));
		if ((u % period) == 0) {
			// B = the kk-th best of the 16 class bests (bitonic network in registers); lane (hq, c) owns the two column blocks of
			// query tile t = hq (hq < QT); the pass bound B - 2E goes to the wave's table in LDS
			if (hq < QT) {
				int qo = qw;
				MVS_OPAQUE_VGPR(qo);
				f32x4n v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
				for (int i = 0; i < NCBP; ++i) { // (one query at a time: the resident fragments leave few registers)
					const int q = qo + 16 * NCBP * hq + 16 * i + c;
					const int qc = q < a.nq ? q : 0;
					if (a.pbnd != nullptr && (a.opt & 256)) { // lists beyond 128 entries: frozen bounds, one per query (csrc/flat_collect.hip)
						const float bvf = q < a.nq ? a.pbnd[qc] : __uint_as_float(0x7fc00000u);
						v[0] = i == 0 ? bvf : v[0];
						v[1] = i == 1 ? bvf : v[1];
						v[2] = i == 2 ? bvf : v[2];
						continue;
					}
					const float e2v = __builtin_nontemporal_load(a.e2 + qc);
					// NC = 128 (32 < kk <= 128, round 6): four SUBSETS of 32 classes (class = row & 127, subset = class >> 5); the worst of the
					// subsets' ceil(kk / 4)-th best class values has >= kk distinct rows at least as good (csrc/flat_collect.hip)
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
						for (int j = 0; j < SUBN / 2; ++j)
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
					v[2] = i == 2 ? bv : v[2];
				}
				*(f32x4n *)(cqtab + (wave * QT * 16 + hq * 16 + c) * 4) = v;
			}
		}
		dma_block(u + 1); // the next staged block streams in under this one's MFMAs (every LDS read below is hand-written)
#pragma unroll 1
		for (int sub = 0; sub < WSUB; ++sub) {
			const unsigned tb = (unsigned)(uintptr_t)((lds_f32c *)(smem + ((u & 1) * STAGE_BYTES + sub * TILE_BYTES) / 4)) + rbase;
			const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32c *)(nbuf + (u & 1) * 64 + sub * RT + 4 * hq));
			const long long row0 = r_begin + ((long long)u * WSUB + sub) * RT;
			const int nvalid = (int)((r_end - row0) < RT ? (r_end - row0) : RT); // (<= 0 behind the split's last row)
			f32x4n Y;
			asm volatile("ds_read_b128 %0, %1" : "=v"(Y) : "v"(nb_lds) : "memory");
#pragma unroll
			for (int t = 0; t < QT; ++t) {
				f32x4n cqv;
				asm volatile("ds_read_b128 %0, %1" : "=v"(cqv) : "v"(cq_lds + (unsigned)(t * 256)) : "memory");
				bf16x8 A[4]; // ring: k-block kb lives in A[kb & 3]; two k-blocks are read ahead
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[0]) : "v"(tb) : "memory");
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[1]) : "v"(tb ^ 64u) : "memory");
				f32x4w acc[NCBP];
#pragma unroll
				for (int g = 0; g < KB / 2; ++g) { // groups of two k-blocks
					if (g + 1 < KB / 2) {
						const int k2 = 2 * g + 2, k3 = 2 * g + 3;
						asm volatile("ds_read_b128 %0, %1" : "=v"(A[k2 & 3]) : "v"((tb ^ (unsigned)((k2 & 3) * 64)) + (unsigned)((k2 >> 2) * 256)) : "memory");
						asm volatile("ds_read_b128 %0, %1" : "=v"(A[k3 & 3]) : "v"((tb ^ (unsigned)((k3 & 3) * 64)) + (unsigned)((k3 >> 2) * 256)) : "memory");
						// this group's two fragments (and, the first time, beta and the bounds) have arrived: LDS returns in order
						asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[(2 * g) & 3]), "+v"(A[(2 * g + 1) & 3]), "+v"(Y), "+v"(cqv));
					} else {
						asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[(2 * g) & 3]), "+v"(A[(2 * g + 1) & 3]), "+v"(Y), "+v"(cqv));
					}
#pragma unroll
					for (int kk2 = 0; kk2 < 2; ++kk2) {
						const int kb = 2 * g + kk2;
#pragma unroll
						for (int i = 0; i < NCBP; ++i) {
							if (kb == 0) // the chain starts at beta(row): s comes out of the matrix pipe
								acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb & 3], bq[NCBP * t + i][kb], Y, 0, 0, 0);
							else
								acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb & 3], bq[NCBP * t + i][kb], acc[i], 0, 0, 0);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				const float mx0 = __builtin_fmaxf(__builtin_fmaxf(acc[0][0], acc[0][1]), __builtin_fmaxf(acc[0][2], acc[0][3]));
				const float mx1 = __builtin_fmaxf(__builtin_fmaxf(acc[1][0], acc[1][1]), __builtin_fmaxf(acc[1][2], acc[1][3]));
				bool any_t = (mx0 >= cqv[0]) || (mx1 >= cqv[1]); // NaN on either side: false
				if (NCBP == 3) {
					const float mx2 = __builtin_fmaxf(__builtin_fmaxf(acc[2][0], acc[2][1]), __builtin_fmaxf(acc[2][2], acc[2][3]));
					any_t = any_t || (mx2 >= cqv[2]);
				}
				rare(acc, t, any_t, cqv, row0, nvalid);
			}
		}
		__syncthreads(); // also drains this block's LDS-DMA (vmcnt(0)) before the next block reads it
		if (COLLECT && ((u % CL_FLUSH_EVERY) == CL_FLUSH_EVERY - 1 || u == nblocks - 1)) {
			const unsigned fill = qctl[0];
			__syncthreads(); // everybody has read the same fill before anyone appends again
			const unsigned n = fill < (unsigned)CL_QCAP ? fill : (unsigned)CL_QCAP;
			if (n >= (unsigned)CL_QCAP / 2 || (u == nblocks - 1 && n > 0)) {
				if (tid == 0) {
					*(unsigned long long *)(qctl + 2) = atomicAdd(a.stream_cnt, (unsigned long long)n);
					qctl[0] = 0u;
				}
				__syncthreads();
				const unsigned long long base = *(const unsigned long long *)(qctl + 2);
				for (unsigned i = tid; i < n; i += 256)
					if ((long long)(base + i) < a.stream_cap)
						a.stream[base + i] = qbuf[i];
				__syncthreads();
			}
		}
	}
}

// ---- 512 < d <= 1024: the k dimension split over a wave pair ---------------------------------------------------------------------
// 32 queries x 24 k-blocks are 192 VGPRs of query fragments -- too many for one wave.  Waves 2 g and 2 g + 1 share the 32 queries
// of group g and hold 12 k-blocks each (96 VGPRs); both run their half of the chain over the same 16-row tile, hand the partial
// sums of the OTHER wave's column block over through LDS (1 KB each way) and finish their own 16 queries: s = (beta + half) + half.
// (One more f32 addition than the single chain; the bound counts d / 16 accumulation steps where d / 32 + 1 happen.)
// A workgroup serves 64 queries, a staged block is one tile (24 KB), the workgroup barrier of the hand-over is the staging barrier.
template <bool IS_L2, bool COLLECT, int NW, int NST, int NCB, int KBT>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void flat_bf16_ksplit_kernel(const CollectArgs a) {
	constexpr int KH = KBT / 2; // k-blocks per wave: 12 (store of 768 dims) or 16 (1024)
	static_assert(KH % 4 == 0 && 8 * NCB * KH <= 576, "k-blocks per wave");
	constexpr int PITCH = 64 * KBT, C = 4 * KBT, RT = 16;
	constexpr int STAGE_BYTES = RT * PITCH; // 24 KB (32 KB)
	constexpr int DMA_PER_WAVE = STAGE_BYTES / (1024 * NW);
	static_assert(STAGE_BYTES % (1024 * NW) == 0, "staging");
	constexpr int QP = 16 * NCB;        // queries of a wave pair: NCB column blocks (the third one finished by the two waves in turn)
	constexpr int QB = (NW / 2) * QP;
	constexpr int FLUSH_EVERY = NST == 3 ? 16 : 8;
	constexpr int QCAP = (NW == 4 && NCB == 3) ? CL_QCAP / 2 : CL_QCAP; // (two workgroups per CU must fit 160 KB)
	static_assert(NCB == 2 || NCB == 3, "column blocks per wave pair");
	static_assert(NST == 2 || NST == 3, "two stages (one tile ahead, __syncthreads) or a ring of three (two tiles ahead)");

	extern __shared__ __attribute__((aligned(16))) float smem[];
	char *tbuf = (char *)smem;                                        // [NST][STAGE_BYTES]
	float *nbuf = (float *)(tbuf + NST * STAGE_BYTES);                // [NST][64] beta of the staged rows (16 used)
	unsigned long long *qbuf = (unsigned long long *)(nbuf + NST * 64); // [QCAP] candidate queue
	f32x4w *xbuf = (f32x4w *)(qbuf + QCAP);                        // [2][NW waves][NCB - 1][64 lanes] partial sums for the partner wave
	float *cqtab = (float *)(xbuf + 2 * NW * (NCB - 1) * 64);         // [NW / 2 pairs][QP]: pass bound of every query
	unsigned *qctl = (unsigned *)(cqtab + QB);                        // [0] queue fill, [2..3] flush base

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int kh = wave & 1, qg = wave >> 1;
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
	const int nblocks = r_end > r_begin ? (int)((r_end - r_begin + RT - 1) / RT) : 0;
	if (tid == 0)
		qctl[0] = 0u;
	const int qpair = qb * QB + qg * QP; // the pair's queries; this wave finishes column block kh (and block 2 on tiles u & 1 == kh)

	bf16x8 bq[NCB][KH]; // [column block of the group][k-block of this wave's half]
	{
		const bf16x8 *qsrc = (const bf16x8 *)a.qf;
#pragma unroll
		for (int cb = 0; cb < NCB; ++cb) {
			const size_t qblk16 = (size_t)qb * (QB / 16) + qg * NCB + cb;
#pragma unroll
			for (int kb = 0; kb < KH; ++kb)
				bq[cb][kb] = qsrc[(qblk16 * KBT + kh * KH + kb) * 64 + lane];
		}
	}

	auto dma_block = [&](int u, int stg) {
		const char *base = (const char *)a.yb + (size_t)(r_begin + (long long)u * RT) * PITCH; // uniform
#pragma unroll
		for (int i = 0; i < DMA_PER_WAVE; ++i) {
			const int inst = NW * i + wave;
			const int S = 64 * inst + lane, r = S / C, p = S - r * C;
			const unsigned off = (unsigned)(r * PITCH + (((p & ~15) | ((p & 15) ^ (r & 15))) * 16));
			__builtin_amdgcn_global_load_lds((glb_f32c *)(base + off),
			                                 (lds_f32c *)(smem + (stg * STAGE_BYTES + inst * 1024) / 4), 16, 0, 0);
		}
		if (wave == 0) {
			const float *bb = a.yn + (r_begin + (long long)u * RT); // uniform
			__builtin_amdgcn_global_load_lds((glb_f32c *)(bb + lane), (lds_f32c *)(smem + (NST * STAGE_BYTES) / 4 + stg * 64), 4, 0, 0);
		}
	};
	if (nblocks > 0) {
		dma_block(0, 0);
		if (NST == 3)
			dma_block(1, 1);
	}
	__syncthreads();

	const unsigned rbase = (unsigned)(c * PITCH) + (unsigned)(((hq ^ c) & 15) * 16) + (unsigned)(kh * (KH / 4) * 256);
	const unsigned qcnt_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned *)qctl);
	const unsigned qbuf_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned long long *)qbuf);
	const unsigned cq_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)cqtab) + (unsigned)((qg * QP + kh * 16 + c) * 4);
	const unsigned cq2_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float *)cqtab) + (unsigned)((qg * QP + 32 + c) * 4);
	const unsigned xb_lds = (unsigned)(uintptr_t)((__attribute__((address_space(3))) f32x4w *)xbuf) + (unsigned)(lane * 16);
	int stg = 0; // stage of block u
	// the bound test of tile u - 1 runs behind the MFMAs of tile u (the partner's sums then are one barrier old): what it needs
	f32x4w pmine = {0.f, 0.f, 0.f, 0.f}, pmine2 = {0.f, 0.f, 0.f, 0.f};
	float pcq = 0.f, pcq2 = 0.f;
	if (NW == 8 && (a.opt & 1)) // the two waves of a SIMD take the matrix pipe one after the other (measured: 30.4 vs 29.x ms without)
	{
		if (wave < 4)
			__builtin_amdgcn_s_setprio(1);
		else
			__builtin_amdgcn_s_setprio(0);
	}

	auto finish = [&](const f32x4w sv, const float cqv, const int qoff, const long long row0, const int nvalid) {
		const float mx = __builtin_fmaxf(__builtin_fmaxf(sv[0], sv[1]), __builtin_fmaxf(sv[2], sv[3]));
		const bool any_t = mx >= cqv; // NaN on either side: false
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(any_t) == 0ull, 1)) // (hot path = fall-through: no taken branch per half tile)
			return;
		int qo = qpair;
		MVS_OPAQUE_VGPR(qo);
		const int q = qo + qoff + c;
		unsigned m = 0u;
		if (any_t) {
#pragma unroll
			for (int r = 0; r < 4; ++r)
				if (4 * hq + r < nvalid && sv[r] >= cqv)
					m |= 1u << r;
		}
		if (a.rowmask && m != 0u) { // IDSelector: rejected rows are neither candidates nor evidence for the bound
			const unsigned long long rr = (unsigned long long)(row0 + 4 * hq);
			m &= (unsigned)(((const unsigned *)a.rowmask)[rr >> 5] >> (rr & 31u));
		}
		while (m != 0u) {
			const int j = __builtin_ctz(m);
			m &= m - 1u;
			const float lo = (j & 1) ? sv[1] : sv[0];
			const float hi = (j & 1) ? sv[3] : sv[2];
			const float v = (j & 2) ? hi : lo;
			const unsigned row = (unsigned)(row0 + 4 * hq + j);
			typedef __attribute__((address_space(1))) unsigned *GU;
			__hip_atomic_fetch_min((GU)(a.gslot + (size_t)q * 16) + (row & 15u), skey(v), __ATOMIC_RELAXED,
			                       __HIP_MEMORY_SCOPE_AGENT);
			if (COLLECT) {
				unsigned pos;
				const unsigned one = 1u;
				asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(qcnt_lds), "v"(one) : "memory");
				const unsigned long long ent = ((unsigned long long)(unsigned)q << 32) | row;
				if (pos < (unsigned)QCAP) {
					asm volatile("ds_write_b64 %0, %1" ::"v"(qbuf_lds + 8u * pos), "v"(ent) : "memory");
				} else {
					unsigned long long gp;
					const unsigned long long one64 = 1ull;
					typedef __attribute__((address_space(1))) unsigned long long *GUL;
					asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)"
					             : "=&v"(gp)
					             : "v"((GUL)a.stream_cnt), "v"(one64)
					             : "memory");
					if ((long long)gp < a.stream_cap)
						*((GUL)a.stream + gp) = ent;
				}
			}
		}
		if (NST == 3) // (the slot and stream updates are done before the next LDS-DMA is issued: the barrier counts loads only)
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		else
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	};
	// tile t (t = u - 1 inside the loop): the partner's sums from xbuf[t & 1] + this wave's, then the bound test
	auto epilogue = [&](int t) {
		const bool m2 = NCB == 3 && (t & 1) == kh;
		f32x4w other, other2;
		asm volatile("ds_read_b128 %0, %1" : "=v"(other) : "v"(xb_lds + (unsigned)((((t & 1) * NW + (wave ^ 1)) * (NCB - 1)) * 1024)) : "memory");
		if (NCB == 3 && m2)
			asm volatile("ds_read_b128 %0, %1" : "=v"(other2) : "v"(xb_lds + (unsigned)((((t & 1) * NW + (wave ^ 1)) * (NCB - 1) + 1) * 1024)) : "memory");
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(other), "+v"(other2));
		const long long row0 = r_begin + (long long)t * RT;
		const int nvalid = (int)((r_end - row0) < RT ? (r_end - row0) : RT);
		finish(pmine + other, pcq, kh * 16, row0, nvalid);
		if (NCB == 3 && m2)
			finish(pmine2 + other2, pcq2, 32, row0, nvalid);
	};

	for (int u = 0; u < nblocks; ++u) {
		const int period = u < 8 ? 1 : (u < 64 ? 8 : (u < 512 ? 32 : 128));
		if ((u % period) == 0 && hq < NCB - 1) {
			// B = the kk-th best of the 16 class bests of the lane's query (bitonic network in registers): lanes hq = 0 own the wave's
			// column block, lanes hq = 1 the pair's third one (both waves write it: either value is a valid bound)
			const int qoff = hq == 0 ? kh * 16 : 32;
			int qo = qpair;
			MVS_OPAQUE_VGPR(qo);
			const int q = qo + qoff + c;
			const int qc = q < a.nq ? q : 0;
			unsigned long long w[8];
			const unsigned long long *src = (const unsigned long long *)(a.gslot + (size_t)qc * 16);
#pragma unroll
			for (int j = 0; j < 8; ++j)
				w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const float e2v = __builtin_nontemporal_load(a.e2 + qc);
#pragma unroll
			for (int j = 0; j < 8; ++j)
				asm volatile("" : "+v"(w[j]));
			unsigned key[16];
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				key[2 * j] = (unsigned)w[j];
				key[2 * j + 1] = (unsigned)(w[j] >> 32);
			}
#pragma unroll
			for (int kbit = 2; kbit <= 16; kbit <<= 1)
#pragma unroll
				for (int jb = kbit >> 1; jb > 0; jb >>= 1)
#pragma unroll
					for (int x0 = 0; x0 < 16; ++x0) {
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
			for (int j = 1; j < 16; ++j)
				kth = (a.nclass - 1 == j) ? key[j] : kth;
			const unsigned neutral = skey(-FLT_MAX);
			const float B = skey2f(kth < neutral ? kth : neutral);
			cqtab[qg * QP + qoff + c] = q < a.nq ? B - e2v : __uint_as_float(0x7fc00000u);
		}
		// NST = 3: block u + 2 goes to the stage block u - 1 left at the last barrier; NST = 2: block u + 1
		dma_block(u + NST - 1, stg == 0 ? NST - 1 : stg - 1);
		const unsigned tb = (unsigned)(uintptr_t)((lds_f32c *)(smem + (stg * STAGE_BYTES) / 4)) + rbase;
		const unsigned nb_lds = (unsigned)(uintptr_t)((lds_f32c *)(nbuf + stg * 64 + 4 * hq));
		const long long row0 = r_begin + (long long)u * RT;
		const int nvalid = (int)((r_end - row0) < RT ? (r_end - row0) : RT);
		const bool mine2 = NCB == 3 && (u & 1) == kh; // this wave finishes the third column block of this tile
		f32x4n Y;
		float cq, cq2 = 0.f;
		asm volatile("ds_read_b128 %0, %1" : "=v"(Y) : "v"(nb_lds) : "memory");
		asm volatile("ds_read_b32 %0, %1" : "=v"(cq) : "v"(cq_lds) : "memory");
		if (NCB == 3)
			asm volatile("ds_read_b32 %0, %1" : "=v"(cq2) : "v"(cq2_lds) : "memory");
		bf16x8 A[4];
		asm volatile("ds_read_b128 %0, %1" : "=v"(A[0]) : "v"(tb) : "memory");
		asm volatile("ds_read_b128 %0, %1" : "=v"(A[1]) : "v"(tb ^ 64u) : "memory");
		f32x4w acc[NCB];
#pragma unroll
		for (int g = 0; g < KH / 2; ++g) {
			if (g + 1 < KH / 2) {
				const int k2 = 2 * g + 2, k3 = 2 * g + 3;
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[k2 & 3]) : "v"((tb ^ (unsigned)((k2 & 3) * 64)) + (unsigned)((k2 >> 2) * 256)) : "memory");
				asm volatile("ds_read_b128 %0, %1" : "=v"(A[k3 & 3]) : "v"((tb ^ (unsigned)((k3 & 3) * 64)) + (unsigned)((k3 >> 2) * 256)) : "memory");
				asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[(2 * g) & 3]), "+v"(A[(2 * g + 1) & 3]), "+v"(Y), "+v"(cq), "+v"(cq2));
			} else {
				asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[(2 * g) & 3]), "+v"(A[(2 * g + 1) & 3]), "+v"(Y), "+v"(cq), "+v"(cq2));
			}
#pragma unroll
			for (int kk2 = 0; kk2 < 2; ++kk2) {
				const int kb = 2 * g + kk2;
#pragma unroll
				for (int i = 0; i < NCB; ++i) {
					if (kb == 0) { // beta(row) enters the chain of the column block(s) this wave finishes, once
						const bool fin = i == 2 ? mine2 : i == kh;
						f32x4w y0;
#pragma unroll
						for (int r = 0; r < 4; ++r)
							y0[r] = fin ? Y[r] : 0.f;
						acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb & 3], bq[i][kb], y0, 0, 0, 0);
					} else {
						acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kb & 3], bq[i][kb], acc[i], 0, 0, 0);
					}
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		{
			const f32x4w theirs = kh ? acc[0] : acc[1];
			asm volatile("ds_write_b128 %0, %1" ::"v"(xb_lds + (unsigned)((((u & 1) * NW + wave) * (NCB - 1)) * 1024)), "v"(theirs) : "memory");
			if (NCB == 3 && !mine2)
				asm volatile("ds_write_b128 %0, %1" ::"v"(xb_lds + (unsigned)((((u & 1) * NW + wave) * (NCB - 1) + 1) * 1024)), "v"(acc[NCB - 1]) : "memory");
		}
		if (u > 0)
			epilogue(u - 1);
		pmine = kh ? acc[1] : acc[0];
		if (NCB == 3)
			pmine2 = acc[NCB - 1];
		pcq = cq;
		pcq2 = cq2;
		// the partner's half is there; block u + 1 has landed; this stage is free again.  NST = 3: the newest block (the last
		// DMA_PER_WAVE (+ 1: beta) loads of this wave, nothing else is in flight: loads return in order) stays in flight
		if (NST == 3) {
			if (wave == 0)
				asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_WAVE + 1) : "memory");
			else
				asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_WAVE) : "memory");
		} else {
			__syncthreads();
		}
		stg = stg + 1 == NST ? 0 : stg + 1;
		if (COLLECT && (u % FLUSH_EVERY) == FLUSH_EVERY - 1 && u != nblocks - 1) {
			__syncthreads();
			const unsigned fill = qctl[0];
			__syncthreads();
			const unsigned n = fill < (unsigned)QCAP ? fill : (unsigned)QCAP;
			if (n >= (unsigned)QCAP / 2) {
				if (tid == 0) {
					*(unsigned long long *)(qctl + 2) = atomicAdd(a.stream_cnt, (unsigned long long)n);
					qctl[0] = 0u;
				}
				__syncthreads();
				const unsigned long long base = *(const unsigned long long *)(qctl + 2);
				for (unsigned i = tid; i < n; i += 64 * NW)
					if ((long long)(base + i) < a.stream_cap)
						a.stream[base + i] = qbuf[i];
				__syncthreads();
			}
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the blocks fetched past the split's end)
	if (nblocks > 0)
		epilogue(nblocks - 1);
	if (COLLECT) {
		__syncthreads(); // every wave's appends are in
		const unsigned fill = qctl[0];
		const unsigned n = fill < (unsigned)QCAP ? fill : (unsigned)QCAP;
		if (n > 0) {
			if (tid == 0)
				*(unsigned long long *)(qctl + 2) = atomicAdd(a.stream_cnt, (unsigned long long)n);
			__syncthreads();
			const unsigned long long base = *(const unsigned long long *)(qctl + 2);
			for (unsigned i = tid; i < n; i += 64 * NW)
				if ((long long)(base + i) < a.stream_cap)
					a.stream[base + i] = qbuf[i];
		}
	}
}

// ---- storage: one wave per row (plain f32 rows of pitch sdp, d logical dims) -> centred bf16 [dp1] + beta ------------------------
template <bool IS_L2>
__global__ __launch_bounds__(256) void rows_to_bf16_wide_kernel(const float *__restrict__ src, int sdp, int d, int dp1,
                                                               int interleaved, long long row0, long long nrows, const float *__restrict__ mu,
                                                               unsigned short *__restrict__ dst, float *__restrict__ beta,
                                                               const float *__restrict__ norms, unsigned *__restrict__ max_bits,
                                                               int *__restrict__ outl) {
	const int lane = threadIdx.x & 63;
	// (round 6: grid-stride over the rows, the maxima in registers, one atomic per wave and maximum at the end -- a plain load of
	// max_bits[..] is served from the CU's vector cache as it was first fetched, so rounds 4-5 sent four atomics PER ROW to one L2 line:
	// csrc/flat_collect.hip rows_to_bf16_hi_kernel)
	const float tau = outl ? __uint_as_float(max_bits[3]) : INFINITY; // (written before this launch)
	unsigned m0 = 0u, m2 = 0u, m8 = 0u, m12 = 0u;
	for (long long r = row0 + (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < row0 + nrows; r += (long long)gridDim.x * 4) {
	// FlatGeom::pair_interleaved: every 4 floats stored [k0,k2,k1,k3] (bit 4 of the row clear) or [k1,k3,k0,k2]
	const int flip = interleaved ? (((r >> 4) & 1) ? 2 : 0) : 0;
	float n2 = 0.f, my = 0.f, r2 = 0.f; // ||y'||^2, <mu, y>, ||y' - bf16(y')||^2 (flat_collect.hip, "ROUND 4")
	for (int c8 = lane; c8 < dp1 / 8; c8 += 64) {
		bf16x8 hi;
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int kk = c8 * 8 + e;
			const int j = kk & 3, sk = interleaved ? ((kk & ~3) + ((((j & 1) << 1) | (j >> 1)) ^ flip)) : kk;
			const float v = kk < d ? src[(size_t)r * sdp + sk] : 0.f;
			const float m = kk < d ? mu[kk] : 0.f;
			const float cv = v - m;
			hi[e] = (__bf16)cv;
			const float dl = cv - (float)hi[e];
			n2 = fmaf(cv, cv, n2);
			my = fmaf(m, v, my);
			r2 = fmaf(dl, dl, r2);
		}
		*(bf16x8 *)(dst + (size_t)r * dp1 + c8 * 8) = hi;
	}
	for (int o = 32; o >= 1; o >>= 1) {
		n2 += __shfl_xor(n2, o);
		my += __shfl_xor(my, o);
		r2 += __shfl_xor(r2, o);
	}
	const unsigned b = __float_as_uint(norms[r]);
	// outlier rows stay out of the store (csrc/flat_collect.hip "outlier rows"): zero vector, beta = -inf, not in the maxima
	{
		int slot = CL_OUTL_CAP;
		if (n2 > tau && outl) {
			if (lane == 0)
				slot = atomicAdd(outl, 1);
			slot = __shfl(slot, 0);
		}
		if (slot < CL_OUTL_CAP) {
			const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
			for (int c8 = lane; c8 < dp1 / 8; c8 += 64)
				*(bf16x8 *)(dst + (size_t)r * dp1 + c8 * 8) = zero;
			if (lane == 0) {
				outl[1 + slot] = (int)r;
				beta[r] = -INFINITY;
			}
			m0 = b > m0 ? b : m0;
			continue;
		}
	}
	if (lane == 0)
		beta[r] = IS_L2 ? -n2 : my;
	m0 = b > m0 ? b : m0;
	m2 = b > m2 ? b : m2;
	const unsigned bc = __float_as_uint(n2);
	m8 = bc > m8 ? bc : m8;
	const unsigned br = __float_as_uint(r2);
	m12 = br > m12 ? br : m12;
	}
	if (lane == 0) { // (the wave's values are uniform: every lane saw the same sums)
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
void launch_rows_to_bf16_wide(int metric, const float *d_vecs, int sdp, int interleaved, int d, int dp1, int64_t row0, int64_t nrows,
                              const float *d_mu, unsigned short *d_bf, float *d_beta, const float *d_norms,
                              unsigned *d_max_norm_bits, hipStream_t st, int *d_outl) {
	if (nrows <= 0)
		return;
	const dim3 grid((unsigned)std::min<int64_t>((nrows + 3) / 4, 16384)); // (grid-stride)
	if (metric == METRIC_L2)
		hipLaunchKernelGGL(rows_to_bf16_wide_kernel<true>, grid, dim3(256), 0, st, d_vecs, sdp, d, dp1, interleaved, (long long)row0,
		                   (long long)nrows, d_mu, d_bf, d_beta, d_norms, d_max_norm_bits, d_outl);
	else
		hipLaunchKernelGGL(rows_to_bf16_wide_kernel<false>, grid, dim3(256), 0, st, d_vecs, sdp, d, dp1, interleaved, (long long)row0,
		                   (long long)nrows, d_mu, d_bf, d_beta, d_norms, d_max_norm_bits, d_outl);
	MVS_HIP(hipGetLastError());
}

// ---- candidates -> exact values: one wave per 64 candidates, one k-ordered chain per lane (PAIR: FAISS's per-pair branch for L2,
// see flat_collect.hip).  The 64 rows (f32, pitch sdp, FlatGeom::pair_interleaved or plain) pass through LDS in slabs of 64
// dimensions fetched with coalesced 16-byte loads; the candidates are sorted by query, so the query values are broadcast loads.
template <bool IS_L2, bool PAIR>
__global__ __launch_bounds__(64) void collect_exact_wide_kernel(unsigned long long *__restrict__ sorted, long long ncand,
                                                               const float *__restrict__ x, int d,
                                                               const float *__restrict__ vecs, int sdp, int interleaved,
                                                               const float *__restrict__ norms, const float *__restrict__ qn,
                                                               const unsigned long long *__restrict__ cnt) {
	__shared__ float tile[64][65];
	const int lane = threadIdx.x;
	if (cnt) { // device-count mode (flat_collect.hip, "candidates grouped by query WITHOUT the host knowing how many there are")
		const unsigned long long have = *cnt;
		ncand = have < (unsigned long long)ncand ? (long long)have : ncand;
	}
	for (long long i0 = (long long)blockIdx.x * 64; i0 < ncand; i0 += (long long)gridDim.x * 64) {
	const long long i = i0 + lane;
	const unsigned long long ent = i < ncand ? sorted[i] : 0ull; // (idle lanes: row 0 of query 0, computed and dropped)
	const unsigned row = (unsigned)ent;
	const long long q = (long long)(ent >> 32);
	const bool flip = interleaved && ((row >> 4) & 1); // stored [k1,k3,k0,k2] instead of [k0,k2,k1,k3]
	const float *xq = x + q * d;
	const bool x16 = (d & 3) == 0;
	float acc = 0.f;
	for (int c0 = 0; c0 < sdp; c0 += 64) {
#pragma unroll 4
		for (int it = 0; it < 16; ++it) { // 4 rows x 16 float4 per step
			const int r = it * 4 + (lane >> 4), ch = lane & 15;
			const unsigned rr = (unsigned)__shfl((int)row, r);
			const float4 v = *(const float4 *)(vecs + (size_t)rr * sdp + c0 + ch * 4);
			tile[r][ch * 4 + 0] = v.x;
			tile[r][ch * 4 + 1] = v.y;
			tile[r][ch * 4 + 2] = v.z;
			tile[r][ch * 4 + 3] = v.w;
		}
		__syncthreads();
		const int w = d - c0 < 64 ? d - c0 : 64; // (<= 0 in the padding)
		for (int g = 0; g * 4 < w; ++g) {
			const float s0 = tile[lane][g * 4], s1 = tile[lane][g * 4 + 1], s2 = tile[lane][g * 4 + 2], s3 = tile[lane][g * 4 + 3];
			float y[4];
			if (interleaved) {
				y[0] = flip ? s2 : s0, y[1] = flip ? s0 : s2, y[2] = flip ? s3 : s1, y[3] = flip ? s1 : s3;
			} else {
				y[0] = s0, y[1] = s1, y[2] = s2, y[3] = s3;
			}
			float xv[4] = {0.f, 0.f, 0.f, 0.f};
			if (x16) {
				const float4 t4 = *(const float4 *)(xq + c0 + g * 4);
				xv[0] = t4.x, xv[1] = t4.y, xv[2] = t4.z, xv[3] = t4.w;
			} else {
#pragma unroll
				for (int e = 0; e < 4; ++e)
					if (g * 4 + e < w)
						xv[e] = xq[c0 + g * 4 + e];
			}
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				if (g * 4 + e < w) {
					if (PAIR) {
						const float t = __fsub_rn(xv[e], y[e]);
						acc = fmaf(t, t, acc);
					} else {
						acc = fmaf(xv[e], y[e], acc);
					}
				}
			}
		}
		__syncthreads();
	}
	if (i < ncand) {
	float ex;
	bool ok;
	if (PAIR) {
		ex = acc;
		ok = ex < FLT_MAX;
	} else if (IS_L2) {
		ex = fmaf(-2.0f, acc, qn[q] + norms[row]);
		ex = ex < 0.f ? 0.f : ex; // FAISS: if (dis < 0) dis = 0
		ok = ex < FLT_MAX;
	} else {
		ex = acc;
		ok = ex > -FLT_MAX;
	}
	sorted[i] = ok ? (((unsigned long long)bkey<IS_L2>(ex) << 32) | row) : ~0ull;
	}
	}
}
void launch_collect_exact_wide(int metric, bool per_pair, unsigned long long *d_sorted, int64_t ncand, const float *d_x, int d,
                               const float *d_vecs, int sdp, int interleaved, const float *d_norms, const float *d_qn, hipStream_t st,
                               const unsigned long long *d_cnt) {
	if (ncand <= 0)
		return;
	const dim3 grid((unsigned)(d_cnt ? std::min<int64_t>((ncand + 63) / 64, 8192) : (ncand + 63) / 64));
#define MVS_EXW(L2, PR)                                                                                                \
	{                                                                                                                  \
		auto kern = collect_exact_wide_kernel<L2, PR>;                                                                 \
		hipLaunchKernelGGL(kern, grid, dim3(64), 0, st, d_sorted, (long long)ncand, d_x, d, d_vecs, sdp, interleaved, d_norms, d_qn, d_cnt); \
	}
	if (metric == METRIC_L2 && per_pair)
		MVS_EXW(true, true)
	else if (metric == METRIC_L2)
		MVS_EXW(true, false)
	else
		MVS_EXW(false, false)
#undef MVS_EXW
	MVS_HIP(hipGetLastError());
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// row pitch (dims) of the bf16 store for a logical dimension: 128 (flat_collect.hip), 256, 384, 512 or 768 (k-split); 0 = not served
int collect_store_dims(int d) {
	// (d <= 16: the f32 kernel's contraction is 8-16 dims deep and wins against a 128-dim bf16 product)
	return d <= 16 ? 0 : (d <= 128 ? 128 : (d <= 256 ? 256 : (d <= 384 ? 384 : (d <= 512 ? 512 : (d <= 768 ? 768 : (d <= 1024 ? 1024 : (d <= 1536 ? 1536 : 0)))))));
}
static int ksplit_ncb() {
	return tune().ksplit_ncb == 3 ? 3 : 2;
}
// flat_bf16_big_kernel (csrc/flat_collect_big.hip): the 768 / 1024-dim stores by option, the 1536-dim store always (its only kernel)
static bool wide_on_big(int dp1) {
	return dp1 == 1536 || (tune().wide_big && (dp1 == 768 || dp1 == 1024));
}
// the label last_kernel_info() reports for a store pitch (bench.py's roofline line and the traffic table key on it)
const char *collect_wide_kernel_name(int dp1) {
	return wide_on_big(dp1) ? "flat_bf16_big_kernel" : "flat_bf16_wide_kernel";
}
// row classes the kernel serving a wide store can keep per query: the k-split kernel (A/B options) has its 16 hard-wired; the wide and
// big kernels are instantiated for 16, 32 and 4 x 32
int collect_wide_max_classes(int dp1) {
	if (wide_on_big(dp1))
		return 128;
	return (dp1 == 768 || dp1 == 1024 || (dp1 == 512 && tune().wide512_ksplit)) ? 16 : 128;
}
static int wide_qt(int dp1) {
	return dp1 <= 256 ? 2 : 1;
}
int collect_wide_qblock(int dp1) {
	if (wide_on_big(dp1)) // flat_bf16_big_kernel: one wave per SIMD, all of k resident
		return collect_big_qblock(dp1);
	if (dp1 == 1024) // 8 waves, two column blocks per pair (2 x 16 k-blocks = 128 VGPRs of fragments)
		return 128;
	if (dp1 == 384)
		return tune().wide384_ncb == 3 ? 192 : 128;
	if (dp1 == 512 && tune().wide512_ksplit)
		return 96;
	return dp1 == 768 ? (tune().ksplit_waves / 2) * 16 * ksplit_ncb() : 128 * wide_qt(dp1);
}
static int wide_wsub(int dp1) {
	const int KB = dp1 / 32;
	return KB >= 16 ? 1 : (KB >= 12 ? 2 : 3);
}
// resident workgroups of the scan kernel on the device (256 CUs)
int collect_wide_slots(int dp1) {
	if (wide_on_big(dp1))
		return 256; // one workgroup per CU
	return (dp1 == 1024 || (dp1 == 768 && tune().ksplit_waves == 8)) ? 256 : 512;
}
size_t collect_wide_lds_bytes(int dp1) {
	if (wide_on_big(dp1))
		return collect_big_lds_bytes(dp1);
	if (dp1 == 1024) // flat_bf16_ksplit_kernel<8, 2, 2, 32>: two 32 KB stages, beta, queue, hand-over buffers, bounds, control
		return (size_t)2 * (16 * 1024 * 2 + 64 * 4) + (size_t)CL_QCAP * 8 + (size_t)2 * 8 * 64 * 16 + 128 * 4 + 64;
	if (dp1 == 512 && tune().wide512_ksplit) // flat_bf16_ksplit_kernel<4, 2, 3, 16>
		return (size_t)2 * (16 * 512 * 2 + 64 * 4) + (size_t)(CL_QCAP / 2) * 8 + (size_t)2 * 4 * 2 * 64 * 16 + 96 * 4 + 64;
	if (dp1 == 768) // flat_bf16_ksplit_kernel: two 24 KB stages, beta, queue, hand-over buffers, bounds, control
		return (size_t)(tune().ksplit_waves == 8 ? 3 : 2) * (16 * 768 * 2 + 64 * 4) +
		       (size_t)(tune().ksplit_waves == 4 && ksplit_ncb() == 3 ? CL_QCAP / 2 : CL_QCAP) * 8 +
		       (size_t)2 * tune().ksplit_waves * (ksplit_ncb() - 1) * 64 * 16 + (size_t)collect_wide_qblock(768) * 4 + 64;
	return (size_t)2 * wide_wsub(dp1) * 16 * dp1 * 2 + 2 * 64 * 4 + (size_t)CL_QCAP * 8 + (size_t)4 * wide_qt(dp1) * 16 * 4 * 4 + 64;
}
int collect_wide_block_rows(int dp1) {
	if (wide_on_big(dp1))
		return 16;
	return 16 * wide_wsub(dp1);
}

template <int KB, int QT, int NCBP, bool COLLECT>
static void launch_wide_inst(int metric, const CollectArgs &a, int grid, size_t lds, hipStream_t st) {
#define MVS_WIDE1(L2, NCV)                                                                                        \
	{                                                                                                             \
		auto kern = flat_bf16_wide_kernel<KB, QT, NCBP, L2, COLLECT, NCV>;                                        \
		ensure_dynamic_lds((const void *)kern, lds);                                                              \
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);                                              \
	}
	if (a.slot_stride == 128) { // 32 < kk <= 128: four subsets of 32 row classes (round 6)
		if (metric == METRIC_L2)
			MVS_WIDE1(true, 128)
		else
			MVS_WIDE1(false, 128)
	} else if (a.slot_stride == 32) { // 16 < kk <= 32: 32 row classes per query
		if (metric == METRIC_L2)
			MVS_WIDE1(true, 32)
		else
			MVS_WIDE1(false, 32)
	} else if (metric == METRIC_L2)
		MVS_WIDE1(true, 16)
	else
		MVS_WIDE1(false, 16)
#undef MVS_WIDE1
	MVS_HIP(hipGetLastError());
}

// one launch over rows [row_first, row_end): collect = false -> bound estimation only
void launch_collect_wide_range(int dp1, int metric, bool collect, CollectArgs a, int64_t row_first, int64_t row_end,
                               int64_t nsplit_want, int64_t nq, hipStream_t st, int *grid_out, int *nsplit_out) {
	const int QB = collect_wide_qblock(dp1), BR = collect_wide_block_rows(dp1);
	const int nqb = (int)((nq + QB - 1) / QB);
	const int64_t nblocks = (row_end - row_first + BR - 1) / BR;
	const int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(nsplit_want, nblocks));
	a.xcd_map = (nsplit >= 8 && nsplit % 8 == 0) ? 1 : 0;
	a.row_first = row_first;
	a.n = row_end;
	a.split_rows = (nblocks + nsplit - 1) / nsplit * BR;
	a.nqb = nqb;
	a.nsplit = (int)nsplit;
	a.opt = tune().ksplit_opt | (a.opt & 256); // (bit 8: frozen bounds, the caller's)
	const int grid = nqb * (int)nsplit;
	const size_t lds = collect_wide_lds_bytes(dp1);
	if (wide_on_big(dp1)) {
		launch_collect_big(dp1, metric, collect, a, grid, st);
	} else if (dp1 == 256) {
		if (collect)
			launch_wide_inst<8, 2, 2, true>(metric, a, grid, lds, st);
		else
			launch_wide_inst<8, 2, 2, false>(metric, a, grid, lds, st);
	} else if (dp1 == 384) {
		if (tune().wide384_ncb == 3) {
			if (collect)
				launch_wide_inst<12, 1, 3, true>(metric, a, grid, lds, st);
			else
				launch_wide_inst<12, 1, 3, false>(metric, a, grid, lds, st);
		} else {
			if (collect)
				launch_wide_inst<12, 1, 2, true>(metric, a, grid, lds, st);
			else
				launch_wide_inst<12, 1, 2, false>(metric, a, grid, lds, st);
		}
	} else if (dp1 == 512 && tune().wide512_ksplit) {
#define MVS_KSP5(L2, CO)                                                                                        \
	{                                                                                                           \
		auto kern = flat_bf16_ksplit_kernel<L2, CO, 4, 2, 3, 16>;                                               \
		ensure_dynamic_lds((const void *)kern, lds);                                                            \
		hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);                                  \
	}
		if (metric == METRIC_L2 && collect)
			MVS_KSP5(true, true)
		else if (metric == METRIC_L2)
			MVS_KSP5(true, false)
		else if (collect)
			MVS_KSP5(false, true)
		else
			MVS_KSP5(false, false)
#undef MVS_KSP5
		MVS_HIP(hipGetLastError());
	} else if (dp1 == 512) {
		if (collect)
			launch_wide_inst<16, 1, 2, true>(metric, a, grid, lds, st);
		else
			launch_wide_inst<16, 1, 2, false>(metric, a, grid, lds, st);
	} else if (dp1 == 768) {
#define MVS_KSP(L2, CO)                                                                                         \
	{                                                                                                           \
		if (tune().ksplit_waves == 8 && ksplit_ncb() == 3) {                                                         \
			auto kern = flat_bf16_ksplit_kernel<L2, CO, 8, 3, 3, 24>;                                               \
			ensure_dynamic_lds((const void *)kern, lds);                                                        \
			hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);                              \
		} else if (tune().ksplit_waves == 8) {                                                                       \
			auto kern = flat_bf16_ksplit_kernel<L2, CO, 8, 3, 2, 24>;                                               \
			ensure_dynamic_lds((const void *)kern, lds);                                                        \
			hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);                              \
		} else if (ksplit_ncb() == 3) {                                                                         \
			auto kern = flat_bf16_ksplit_kernel<L2, CO, 4, 2, 3, 24>;                                               \
			ensure_dynamic_lds((const void *)kern, lds);                                                        \
			hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);                              \
		} else {                                                                                                \
			auto kern = flat_bf16_ksplit_kernel<L2, CO, 4, 2, 2, 24>;                                               \
			ensure_dynamic_lds((const void *)kern, lds);                                                        \
			hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);                              \
		}                                                                                                       \
	}
		if (metric == METRIC_L2 && collect)
			MVS_KSP(true, true)
		else if (metric == METRIC_L2)
			MVS_KSP(true, false)
		else if (collect)
			MVS_KSP(false, true)
		else
			MVS_KSP(false, false)
#undef MVS_KSP
		MVS_HIP(hipGetLastError());
	} else if (dp1 == 1024) {
#define MVS_KSP1(L2, CO)                                                                                        \
	{                                                                                                           \
		auto kern = flat_bf16_ksplit_kernel<L2, CO, 8, 2, 2, 32>;                                               \
		ensure_dynamic_lds((const void *)kern, lds);                                                            \
		hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);                                  \
	}
		if (metric == METRIC_L2 && collect)
			MVS_KSP1(true, true)
		else if (metric == METRIC_L2)
			MVS_KSP1(true, false)
		else if (collect)
			MVS_KSP1(false, true)
		else
			MVS_KSP1(false, false)
#undef MVS_KSP1
		MVS_HIP(hipGetLastError());
	} else {
		throw_faiss("mvs::launch_collect_wide_range", __FILE__, "no instance for a %d-dim store", dp1);
	}
	if (grid_out)
		*grid_out = grid;
	if (nsplit_out)
		*nsplit_out = (int)nsplit;
}

} // namespace mvs
