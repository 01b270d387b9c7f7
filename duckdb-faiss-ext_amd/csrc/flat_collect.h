// csrc/flat_collect.h -- shared by the coarse-filter scan kernels (flat_collect.hip: d <= 128; flat_collect_wide.hip: 128 < d <= 1024)
#pragma once
#include "flat_fused.h"

namespace mvs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f32c;
typedef __attribute__((address_space(1))) const float glb_f32c;

constexpr int CL_QBLOCK = 512;  // queries per workgroup
constexpr int CL_BN = 32;       // rows per tile (one pass of the MFMA loop)
constexpr double CL_MFMA_UNITS = 8.0; // modelled bf16-MFMA accumulation error, ulp-units (2^-24) of the magnitudes per 16 dimensions (csrc/flat_collect.hip)
constexpr int CL_SUB = 2;       // tiles per staged block (one barrier per CL_SUB tiles)
constexpr int CL_QCAP = 2048;   // candidate queue of a workgroup (entries of 8 bytes)
constexpr int CL_FLUSH_EVERY = 2; // staged blocks between two looks at the queue

struct CollectArgs {
	const void *qf;            // query fragments (bf16), [qblk32][ch][lane] x 16 bytes
	const unsigned short *yb;  // bf16 rows [n + 64][dp]
	const float *yn;           // beta(row): -||y'||^2 (L2) or <mu, y> (inner product), f32, padded by 64
	const float *e2;           // [nq] 2E(q) (NaN: the query is not served here)
	unsigned *gslot;           // [nq][slot_stride] class slots: keys of the best s per row class (smaller key = better)
	unsigned long long *stream; // candidates (q << 32 | row)
	unsigned long long *stream_cnt; // [0] entries appended
	float *stream_s;           // (may be null; d <= 128 scan) the coarse value s of every entry: the final-bound filter's input
	float *seed_stage;         // (may be null; flat_bf16_seed_kernel) [nsplit][nq][16] class maxima of every row split instead of atomics
	const unsigned long long *rowmask; // SEL instances: bit r of word b = row 64 b + r passes the IDSelector
	long long stream_cap;
	int slot_stride, nclass; // 16 class slots per query (row & 15); nclass = kk, the rank of the bound among them
	long long n, row_first, split_rows;
	long long split_len; // flat_bf16_seed_kernel: rows of a split actually scanned (0: split_rows) -- pass A of the big lists strides over the database
	int nq, nqb, nsplit, xcd_map;
	float *pbnd; // d <= 128 scan: [nqb][512] pass bounds B - 2E in the order of a workgroup's LDS table (flat_collect.hip), or null
	int opt; // A/B bits (option cl_ksplit_opt): 0 = k-split kernel with 8 waves: s_setprio skew between the two waves of a SIMD;
	         // 1 = wide kernels: bound refresh cadence counted in staged blocks instead of rows; 2..3 = d <= 128 kernel: refresh cadence
	         // (0: every 8 / 32 / 128 staged blocks, 1: 4 / 16 / 64, 2: 16 / 64 / 256, 3: 32 / 128 / 512); with the pass-bound table
	         // (pbnd != null): bits 2..3 = table fetch period in staged blocks (0: 4, 1: 2, 2: 8, 3: 16), bits 4..5: full
	         // derivation every 64 (0) / 16 (1) / 128 (2) staged blocks per workgroup, 3 = by the scan's progress (16 / 64 / 256)
};

// csrc/flat_collect_wide.hip
int collect_store_dims(int d); // row pitch (dims) of the bf16 store: 128, 256, 384, 512, 768, 1024; 0 = the coarse filter does not serve d
int collect_wide_qblock(int dp1);
int collect_wide_slots(int dp1);
int collect_wide_max_classes(int dp1); // row classes per query the wide store's kernel can keep: 128 (wide / big kernels: 16 | 32 | 4 x 32) or 16 (k-split)
size_t collect_wide_lds_bytes(int dp1);
int collect_wide_block_rows(int dp1);
void launch_collect_wide_range(int dp1, int metric, bool collect, CollectArgs a, int64_t row_first, int64_t row_end,
                               int64_t nsplit_want, int64_t nq, hipStream_t st, int *grid_out, int *nsplit_out);
// csrc/flat_collect_big.hip: 512 < d <= 1024, one wave per SIMD with all of k resident (512 registers per wave)
int collect_big_qblock(int dp1);
size_t collect_big_lds_bytes(int dp1);
void launch_collect_big(int dp1, int metric, bool collect, const CollectArgs &a, int grid, hipStream_t st);
void launch_rows_to_bf16_wide(int metric, const float *d_vecs, int sdp, int interleaved, int d, int dp1, int64_t row0, int64_t nrows,
                              const float *d_mu, unsigned short *d_bf, float *d_beta, const float *d_norms,
                              unsigned *d_max_norm_bits, hipStream_t st, int *d_outl);
void launch_collect_exact_wide(int metric, bool per_pair, unsigned long long *d_sorted, int64_t ncand, const float *d_x, int d,
                               const float *d_vecs, int sdp, int interleaved, const float *d_norms, const float *d_qn, hipStream_t st,
                               const unsigned long long *d_cnt = nullptr);

__device__ __forceinline__ unsigned skey(float s) { // "larger s is better" as a smaller-is-better key
	return ~f2key(s);
}
__device__ __forceinline__ float skey2f(unsigned k) {
	return key2f(~k);
}

} // namespace mvs
