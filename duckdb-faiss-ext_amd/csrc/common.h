// csrc/common.h -- shared declarations of the MI355X (gfx950) vector-search library.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace mvs {

// Error carrying FAISS's message text (the reference greps substrings of it; include/mi355_faiss.h)
struct Error : std::runtime_error {
	explicit Error(const std::string &m) : std::runtime_error(m) {
	}
};

[[noreturn]] void throw_faiss(const char *func, const char *file, const char *fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-kernel, process-wide setting: different indexes (different k / ef)
// launch the same kernel instance from different host threads, so the limit is only ever RAISED (a thread that lowered
// it between another thread's "set" and "launch" would make that launch fail).  Per device.
void ensure_dynamic_lds(const void *kernel, size_t bytes);

#define MVS_HIP(expr)                                                                                                  \
	do {                                                                                                               \
		hipError_t e_ = (expr);                                                                                        \
		if (e_ != hipSuccess)                                                                                          \
			::mvs::throw_faiss(__func__, __FILE__, "HIP error %d (%s) in %s", (int)e_, hipGetErrorString(e_), #expr);  \
	} while (0)

constexpr int METRIC_IP = 0;
constexpr int METRIC_L2 = 1;
// the other MetricType values the glue registers (src/faiss_extension.cpp:58-68; faiss/MetricType.h)
constexpr int METRIC_L1 = 2, METRIC_LINF = 3, METRIC_LP = 4, METRIC_CANBERRA = 20, METRIC_BRAYCURTIS = 21,
              METRIC_JENSENSHANNON = 22, METRIC_JACCARD = 23;
inline bool metric_is_extra(int m) {
	return (m >= METRIC_L1 && m <= METRIC_LP) || (m >= METRIC_CANBERRA && m <= METRIC_JACCARD);
}
// list order of a metric's results: similarity metrics (is_similarity_metric) keep the largest values, like inner
// product; every other one the smallest, like L2 -- merges, neutral values and threshold keys only need this
inline int metric_order(int m) {
	return (m == METRIC_IP || m == METRIC_JACCARD) ? METRIC_IP : METRIC_L2;
}

// ---------------------------------------------------------------------------------------------
// Flat brute-force search geometry (DESIGN.md "K2/K3")
//   workgroup = 256 threads = 4 waves; wave w owns queries [32w, 32w+32) of a 128-query block and
//   keeps them as the B operand of v_mfma_f32_32x32x2_f32 (query on the lane => top-k is lane local)
// ---------------------------------------------------------------------------------------------
constexpr int QBLOCK = 128; // queries per workgroup
constexpr int WAVE_Q = 32;  // queries per wave (= MFMA N)

// Database storage: row-major, rows padded to dp floats; dp = kc * nch
struct FlatGeom {
	int d;      // logical dimension
	int dp;     // padded row length (floats)
	int kc;     // k extent of one LDS staging unit
	int nch;    // staging units per row (1 => queries stay resident in registers)
	int ntile;  // 32-row MFMA tiles per wave per row tile (2 resident, 8 streaming)
	// HBM row format.  false: plain row-major.  true ("pair-interleaved", d <= 128): inside every group of four
	// consecutive k the floats are stored as [k0,k2,k1,k3] for rows whose index has bit 4 clear and [k1,k3,k0,k2]
	// for rows with bit 4 set -- the two k-steps one MFMA lane half needs are then ONE aligned 8-byte word, and
	// the 32 lanes of a ds_read_b64 hit 32 distinct bank pairs (DESIGN.md "HBM layout").
	bool pair_interleaved;
	int bn() const {
		return ntile * 32;
	}
};
FlatGeom flat_geom_for(int d);

struct FlatSearchPlan {
	int nqb;             // query blocks of 128
	int nsplit;          // database splits (partial top-k lists per query)
	int64_t split_rows;  // rows per split (multiple of bn)
	int grid;
	size_t lds_bytes;
	bool xcd_map;
	bool global_lists; // k-lists in the partial-result buffers (global memory) instead of LDS
};

// device views -------------------------------------------------------------------------------
struct FlatDB {
	const float *vecs;  // [n][dp]
	const float *norms; // [n]   (sum of squares, k-ordered fma chain)
	int64_t n;
};

struct SelectorDev {
	int kind;              // MVS_SEL_*
	const uint8_t *bitmap; // device
	int64_t nbytes;
	const int64_t *sorted_ids; // device, ascending
	int64_t nids;
};

// kernels / launchers (flat_mfma.hip, flat_direct.hip, merge.hip, util_kernels.hip) ------------
size_t qfrag_floats(const FlatGeom &g, int64_t nq);
void launch_pack_queries(const FlatGeom &g, const float *d_x, int64_t nq, float *d_qf, float *d_qnorm,
                         hipStream_t st);
void launch_row_norms(const float *d_vecs, int64_t n, int dp, float *d_norms, hipStream_t st);
void launch_pad_rows(const float *d_src, int64_t n, int d, float *d_dst, int dp, hipStream_t st);
// [n][d] row-major -> storage rows [n][dp] (zero padded; pair-interleaved if g says so); row0 = index of the first row
void launch_pack_rows(const FlatGeom &g, const float *d_src, int64_t n, float *d_dst, int64_t row0, hipStream_t st);
void launch_unpack_rows(const FlatGeom &g, const float *d_rows, int64_t row0, int64_t stride, int64_t n, float *d_dst, hipStream_t st);
void launch_query_norms(const float *d_x, int64_t n, int d, float *d_out, hipStream_t st);

FlatSearchPlan plan_flat_mfma(const FlatGeom &g, int64_t nq, int64_t n, int64_t k);
// partial lists: pd [nsplit][nq][k] f32, pi [nsplit][nq][k] i32
void launch_flat_mfma(const FlatGeom &g, const FlatSearchPlan &p_in, int metric, const float *d_qf, const float *d_qnorm,
                      int64_t nq, FlatDB db, int64_t k, float *d_pd, int32_t *d_pi, unsigned *d_gthr, hipStream_t st,
                      const SelectorDev *sel = nullptr, const int64_t *d_idmap = nullptr); // sel: inner product only
// tie pass of an inner-product search (FlatIndex::search_flat): per query the k smallest row ids with score >= d_T[q]
void launch_flat_mfma_tie(const FlatGeom &g, const FlatSearchPlan &p_in, const float *d_qf, const float *d_T, int64_t nq,
                          FlatDB db, int64_t k, float *d_pd, int32_t *d_pi, unsigned *d_gthr, hipStream_t st,
                          const SelectorDev *sel, const int64_t *d_idmap);
int64_t flat_mfma_max_k(const FlatGeom &g);
int64_t flat_mfma_max_k_lds(const FlatGeom &g); // selector / IVF-item instances keep their k-lists in LDS
// IVF list scan as a segmented variant of the fused kernel (csrc/flat_mfma.hip, ITEMS instances)
bool flat_mfma_items_supported(const FlatGeom &g, int64_t k);
size_t flat_mfma_item_query_floats(const FlatGeom &g, int max_items);
int flat_mfma_item_slots(); // query slots per work item (128)
void launch_flat_mfma_items(const FlatGeom &g, int metric, const float *d_qf, const float *d_qnorm, int64_t nq,
                            const float *d_rows, const float *d_norms, int64_t nrows, int64_t k, const void *d_items,
                            const int *d_nitems, int max_items, const int *d_qidx, const int64_t *d_rowids,
                            const SelectorDev *sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gthr,
                            hipStream_t st, const float *d_item_qn = nullptr /* L2 on residual rows: per item slot */);
// residual variant of the item query packing (L2): fragments of (query - centroid of the item's list), their squared norms
// per item slot and, per query, the largest of them (float bits, atomicMax; the caller zeroes it)
void launch_ivf_pack_item_fragments_residual(const float *d_x, int d, int kc, int nch, const void *d_items,
                                             const int *d_nitems, int max_items, const int *d_qidx, float *d_qf,
                                             const float *d_centroids, const int *d_list_of_blk64, float *d_item_qn,
                                             unsigned *d_qmaxn_bits, hipStream_t st);
// bf16x3 prefilter + exact f32 re-scoring (csrc/flat_bf16.hip)
bool prefilter_supported(const FlatGeom &g);
float prefilter_cerr(int d);
FlatSearchPlan plan_prefilter(const FlatGeom &g, int64_t nq, int64_t n, int64_t kp);
size_t prefilter_qfrag_bytes(const FlatGeom &g, int64_t nq);
void launch_rows_to_bf16(const FlatGeom &g, const float *d_vecs, int64_t row0, int64_t nrows, unsigned short *d_bf,
                         const float *d_norms, unsigned *d_max_norm_bits, hipStream_t st);
void launch_pack_queries_bf16(const FlatGeom &g, const float *d_x, int64_t nq, void *d_qf, hipStream_t st);
void launch_prefilter(const FlatGeom &g, const FlatSearchPlan &p, int metric, const void *d_qf, const float *d_qnorm,
                      int64_t nq, const unsigned short *d_rows_bf, const float *d_norms, int64_t n, int64_t kp, float *d_pd,
                      int32_t *d_pi, unsigned *d_gthr, hipStream_t st);
void launch_rescore_verify(int metric, const float *d_ca, const int64_t *d_ci, int64_t nq, int kp, int kk, const float *d_x,
                           const FlatGeom &g, const float *d_vecs, const float *d_norms, const float *d_qn,
                           const unsigned *d_max_norm_bits, float *d_pd1, int32_t *d_pi1, int *d_fail_cnt, int *d_fail_q,
                           unsigned *d_max_rel_err_bits, hipStream_t st);
void launch_gather_query_rows(const float *d_x, int d, const int *d_fq, int nf, float *d_xf, hipStream_t st);
void launch_scatter_rows(const int *d_fq, int nf, int64_t k, const float *d_Df, const int64_t *d_If, float *d_D,
                         int64_t *d_I, hipStream_t st);

// Flat shadow (ARITH = 2): query norms, row norms (k-ordered chains of the ORIGINAL rows, by position in the list-sorted store) and
// the rows' numbers in the Flat index -- the key's low word is the ROW NUMBER, so that equal values order by id as FAISS's L2 heap does
constexpr int CL_OUTL_CAP = 256; // outlier rows kept out of a Flat index's bf16 store and in every query's candidate set (csrc/flat_collect.hip)
struct IvfFlatArith {
	const float *qn;
	const float *yn;
	const long long *rowids;
};

// ---- per-index tuning (round 5): every knob that used to be a process-wide `g_*` int --------------------------------------------
// An index owns one Tuning (IndexBase::tune_); IndexBase::use_device() -- the first statement of every entry point -- makes it the
// calling thread's current one, and the launch functions read tune().x.  Two indexes searched from two threads neither share
// nor race on these values (SURVEY 8b "Threading"; reference src/faiss_extension.cpp:629 takes only a per-index lock).
struct Tuning {
	// csrc/coarse_select.hip
	int coarse_persistent = 0; // option ivf_coarse_persistent (measured slower, see coarse_dist_mfma_kernel)
	int coarse_abl = 0;        // (profiling library only, option coarse_abl: 1 = no matrix written, 2 = no MFMA loop -- results wrong)
	int coarse_mfma = 2;       // option ivf_coarse_mfma: the distance matrix on the f32 matrix pipe (2: round 5's staging, 1: round 4's) or on the vector ALU (0)
	int coarse_select = 1;     // option ivf_coarse_select: 0 = the IVF coarse quantiser runs on the k-list kernels
	int coarse_bf16 = 1;       // option ivf_coarse_bf16: L2 coarse quantiser as a bf16 filter + exact re-scoring, no distance matrix (csrc/coarse_bf16.hip; 0: round 5's matrix + selection)
	// csrc/flat_bf16.hip
	int pf_nsplit = 0;
	int pf_sched = 0;     // option pf_sched (see the kernel)
	int pf_classes32 = 0; // option pf_classes32 = 1: 32 classes + k-th smallest (measured slower: 67.4 vs 65.5 ms, same box)
	int pf_seed = 0;      // rows of the seeding pre-pass (0 = off: measured no gain)
	int pf_abl = 0;       // profiling: ablation instance of the d = 128 L2 kernel (results wrong)
	// csrc/flat_collect.hip
	int cl_bound_mode = 1;    // option cl_bound_mode: bf16 rounding term from the actual residual norms (1) or the worst case per element (0)
	int cl_abl = 0;           // option cl_abl: profiling ablation of the L2 scan (results wrong)
	int cl_nsplit = 0;        // option cl_nsplit: row splits of the main scan (0 = planned)
	int cl_seed_split = 0;    // option cl_seed_split: row splits of the pre-pass (0 = 32)
	int cl_seed_rows = 16384; // option cl_seed_rows: rows of the bound-estimation pre-pass
	int cl_seed_reg_rows = 32768; // option cl_seed_reg_rows: ... of the register pre-pass of the d <= 128 store (flat_bf16_seed_kernel)
	int cl_seed_regs = 1;     // option cl_seed_regs: d <= 128 pre-pass with class maxima in registers (0: through the scan kernel's rare path)
	int cl_tab = 1;           // option cl_tab: pass bounds through the global table (1) or every wave derives its own (0, round 3)
	int cl_nc32_from = 17;    // option cl_nc32_from
	// csrc/flat_collect_wide.hip, flat_collect_big.hip
	int wide_big = 1;       // option cl_wide_big: 512 < d <= 1024 on flat_bf16_big_kernel (1) or on the k-split kernel (0)
	int big_mode = 3;       // option cl_big_mode (see MODE)
	int ksplit_waves = 4;   // waves per workgroup of flat_bf16_ksplit_kernel (option cl_ksplit_waves: 4 or 8)
	int wide512_ksplit = 0; // option cl_wide512_ksplit: the 512-dim store on the k-split kernel
	int wide384_ncb = 3;    // option cl_wide384_ncb: column blocks per wave of the 384-dim instance (2 | 3)
	int ksplit_opt = 0;     // option cl_ksplit_opt: bit 0 = s_setprio skew
	int ksplit_ncb = 3;     // column blocks per wave pair (option cl_ksplit_ncb: 2, or 3 with 8 waves)
	// csrc/flat_mfma.hip
	int mfma_nsplit = 0;       // 0 = heuristic; > 0 forces the split count (tuning / tests)
	int mfma_warm = 0;         // > 1: warm-up pre-pass over n / mfma_warm rows (experiment)
	int mfma_global_lists = 1; // 1: k-lists in global memory for every k > 12
	int mfma_variant = 2;      // 1 = register-staged generic kernel, 2 = LDS-DMA + A-ring resident kernel
	// csrc/ivf_collect.hip
	int ivf_cl_abl = 0;      // (profiling library only: 1 = the scan without its rare path -- results wrong)
	int ivf_cl_lds_pad = 0;  // (experiment: unused dynamic LDS per workgroup = fewer wavefronts per CU)
	int ivf_cl_xcd = 1;      // option ivf_cl_xcd: items of one list on one XCD (1), their segments next to each other too (2), or dealt round-robin (0)
	int ivf_cl_refresh = 16; // option ivf_cl_refresh (see IvfCollectArgs::refresh)
};
const Tuning &tune();                      // the calling thread's current tuning (the defaults when no index call is in progress)
void set_current_tuning(const Tuning *t);
void forget_current_tuning(const Tuning *t); // (an index is going away: the calling thread must not keep reading its knobs)

// ---- roctx ranges (round 6; SURVEY 5 "tracing"): stage / search / exchange / merge show up as named ranges in a
// rocprofv3 --marker-trace run.  The roctx library is bound at run time and only when asked for (env MVS_ROCTX=1) or when a
// rocprofiler tool is attached to the process; otherwise a TraceRange is two predictable branches.
void trace_push(const char *name);
void trace_pop();
struct TraceRange {
	explicit TraceRange(const char *name) {
		trace_push(name);
	}
	~TraceRange() {
		trace_pop();
	}
	TraceRange(const TraceRange &) = delete;
	TraceRange &operator=(const TraceRange &) = delete;
};

// direct (per-pair) path: nq < 20 or selector present -- FAISS exhaustive_*_seq arithmetic
struct DirectPlan {
	int nsplit;
	int64_t split_rows;
	int grid;
	size_t lds_bytes;
	int qgroup;
};
DirectPlan plan_flat_direct(const FlatGeom &g, int64_t nq, int64_t n, int64_t k);
void launch_flat_direct(const FlatGeom &g, const DirectPlan &p, int metric, const float *d_xq /*[nq][dp]*/, int64_t nq,
                        FlatDB db, int64_t k, SelectorDev sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi,
                        hipStream_t st);
int64_t flat_direct_max_k();

// merge partial lists -> final (FAISS order), translate labels
// The lists hold k entries and the first kout are written out.  Tie detection (inner product, kout = k - 1): a query
// whose kout-th and (kout+1)-th scores are bit-equal is appended to the flag buffers {count, query, raw top-k values,
// raw top-k row ids} for the tie pass.
struct TieFlags {
	int *count;  // [1]
	int *query;  // [nq]
	float *val;  // [nq][k]   merged candidates in the pure order (score desc, row id asc)
	int *row;    // [nq][k]
};
// Flat inner product behind the bucketed finish (round 6): the select kernel prints FAISS's order (score descending, equal scores in
// descending row order) for the first kout of its kk = kout + 1 entries and records the boundary ties as merge_partials_kernel does
struct IpFlatEmit {
	TieFlags flags; // (count == nullptr: no tie detection)
	int kout;
	float *D;
	long long *I;
	const long long *idmap;
	long long label_offset;
};
void launch_emit_sorted(const float *d_pd, const int32_t *d_pi, int64_t nq, int64_t k, int64_t kout, const int64_t *d_idmap,
                        int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st);
void launch_merge_partials(int metric, const float *d_pd, const int32_t *d_pi, int nsplit, int64_t nq, int64_t k,
                           const int64_t *d_idmap, int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st,
                           int64_t kout = -1, const TieFlags *flags = nullptr, bool presorted = false); // presorted: ONE list per query, already in the pure order
// flagged queries -> contiguous [nf][d] query rows and their boundary scores T
void launch_gather_flagged(const float *d_x, int d, const TieFlags &f, int nf, int64_t k, int64_t kout, float *d_xf,
                           float *d_T, hipStream_t st);
// FAISS's CMin-heap outcome for the flagged queries from (raw top-k, the kout smallest row ids with score >= T)
void launch_tie_resolve(const TieFlags &f, int nf, int64_t k, int64_t kout, const int64_t *d_first_ids /*[nf][kout]*/,
                        const int64_t *d_idmap, int64_t label_offset, float *d_D, int64_t *d_I, hipStream_t st);

// inner product, k >= 100, exact tie at the k-th score: FAISS's ReservoirTopN outcome for the flagged queries (csrc/flat_reservoir.hip)
int64_t reservoir_replay_max_k();
void launch_reservoir_replay(const float *d_xf, int nf, int d, const float *d_vecs, int dp, int interleaved, int64_t n, int64_t k,
                             SelectorDev sel, const int64_t *d_idmap, const float *d_T, float *d_scores, float *d_out_v,
                             int32_t *d_out_r, hipStream_t st);
void launch_merge_records(int metric, const int64_t *d_rec, int nshard, int64_t nq, int kk, int kout, bool raw, float *d_D,
                          int64_t *d_I, hipStream_t st);
// host twin (csrc/merge_host.hip) for nshard * kk beyond a workgroup's LDS; rec on the host
void merge_records_host(int metric, const int64_t *rec, int nshard, int64_t nq, int kk, int kout, bool raw, float *D_out,
                        int64_t *I_out);
// IVF (csrc/ivf.hip)
size_t direct_items_lds_bytes(int dp, int64_t k);
void launch_direct_items(int dp, int metric, const float *d_xq, int64_t nq, const float *d_rows, int64_t nrows,
                         const int64_t *d_rowids, int64_t k, const void *d_items, int nitems, const int *d_qidx,
                         SelectorDev sel, const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot,
                         hipStream_t st);
void launch_init_slots(unsigned *d_gslot, int64_t nq, int64_t k, int metric, hipStream_t st);
// IVF with k beyond the k-list kernels (csrc/ivf_select.hip): all distances of the probed lists + segmented sort
struct IvfSelectPair {
	long long out; // first key slot of this (query, list) pair
	int row_begin, len, q, pad;
};
size_t ivf_select_temp_bytes(int64_t total, int64_t nseg);
void launch_ivf_select(int metric, const float *d_xq, int dp, const float *d_rows, const int64_t *d_rowids,
                       const IvfSelectPair *d_pairs, int npairs, const int *d_seg, int64_t nseg, int64_t total, int64_t k,
                       SelectorDev sel, const int64_t *d_idmap_sel, const int64_t *d_idmap_out, unsigned long long *keys_a,
                       unsigned long long *keys_b, void *d_temp, size_t temp_bytes, float *d_D, int64_t *d_I,
                       hipStream_t st, bool raw_positions = false /* I = positions in d_rows instead of stored ids */);
// csrc/ivf_ties.hip: FAISS's heap outcome under exact distance ties (arrival order = probe rank, then list position)
void launch_ivf_mf_to_csr(int64_t *d_I, int64_t total, const int *d_perm_mf, hipStream_t st);
void launch_ivf_finish(int metric, const float *d_pd, const int64_t *d_pi, int64_t nq, int kx, int k, const int64_t *d_rowids,
                       const int64_t *d_idmap_out, float *d_D, int64_t *d_I, int *d_flag /* [1 + nq] */, hipStream_t st);
bool ivf_tie_pass_fits(int d, int64_t k); // A_k of a flagged query fits the tie pass's LDS
void launch_ivf_tie_emit(int metric, const int *d_flag, int nf, const float *d_x, int d, const float *d_T, int k,
                         const int64_t *d_coarse, int np, const int64_t *d_list_off, const float *d_codes, int dp,
                         const int64_t *d_rowids, SelectorDev sel, const int64_t *d_idmap_sel, float *d_emit_v, int64_t *d_emit_id,
                         int *d_emit_p, hipStream_t st);
void launch_ivf_tie_pass(int metric, const int *d_flag, int64_t nq, const float *d_x, int d, const float *d_pd, int kx, int k,
                         const int64_t *d_coarse, int np, const int64_t *d_list_off, const float *d_codes, int dp,
                         const int64_t *d_rowids, SelectorDev sel, const int64_t *d_idmap_sel, const int64_t *d_idmap_out,
                         float *d_D, int64_t *d_I, hipStream_t st);
// csrc/ivf_scan.hip: list-major scan without LDS staging (dp multiple of 16); same item / partial-list formats
bool ivf_scan_supported(int dp, int64_t k);
size_t ivf_scan_lds_bytes(int64_t k);
size_t ivf_scan_query_pack_bytes(int dp, int nitems); // workspace d_xi: the items' queries, slot pairs interleaved
void launch_ivf_scan(int dp, int metric, const float *d_xq, const float *d_rows, int64_t nrows, const int64_t *d_rowids,
                     int64_t k, const void *d_items, int nitems, const int *d_qidx, SelectorDev sel,
                     const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, float *d_xi,
                     const int *d_nitems /* device item count, grid = upper bound; may be null */, hipStream_t st);
// csrc/kmeans_update.hip: compute_centroids of Clustering::train on device, FAISS's summation order
size_t kmeans_update_ws_bytes(int64_t nx, int64_t k);
void launch_kmeans_update(const float *d_x, int64_t nx, int d, const int64_t *d_assign, int64_t k, float *d_cent,
                          float *d_hassign, void *ws, size_t ws_bytes, hipStream_t st);
// Flat per-pair path on the same kernel (regular grid over row splits x groups of 20 queries); partials [nsplit][nq][k]
void launch_pair_scan(int dp, bool interleaved, int metric, const float *d_xq, int64_t nq, const float *d_rows,
                      int64_t nrows, int64_t k, int nsplit, int64_t split_rows, SelectorDev sel, const int64_t *d_idmap,
                      float *d_pd, int32_t *d_pi, unsigned *d_gslot, float *d_xi, hipStream_t st);
// device-side construction of the work items from the coarse-search labels (no host round trip)
int ivf_group_max_items(int64_t npairs, int64_t nlist, int group);
size_t ivf_group_ws_ints(int64_t nlist);
void launch_ivf_group(const int64_t *d_keys, int64_t nq, int nprobe, int64_t nlist, int group, int shift,
                      const int64_t *d_list_begin, const int64_t *d_list_end, int *ws_int, void *d_items, int *d_qidx,
                      int *d_slots, int **d_nitems_out, int **d_cnt_out, hipStream_t st, int key_stride = 1,
                      bool counters_zeroed = false);
void launch_ivf_group2(const int64_t *d_keys, int64_t nq, int nprobe, int64_t nlist, int group, int shift, const int64_t *d_list_begin,
                       const int64_t *d_list_end, int *ws0, int *ws1, void *d_items0, int *d_qidx0, int *d_slots0, void *d_items1,
                       int *d_qidx1, int *d_slots1, int **d_nitems0_out, int **d_nitems1_out, hipStream_t st);
void launch_ivf_pack_item_fragments(const float *d_x, int d, int kc, int nch, const void *d_items, const int *d_nitems,
                                    int max_items, const int *d_qidx, float *d_qf, hipStream_t st);
void launch_merge_items(int metric, const float *d_pd, const int32_t *d_pi, const int *d_slots, int nprobe, int64_t nq,
                        int64_t k, const int64_t *d_rowids, const int64_t *d_idmap, float *d_D, int64_t *d_I,
                        hipStream_t st, int group = 20, int shift = 5);
void launch_gather_rows(const float *d_src, const int *d_perm, int64_t n, int dp, float *d_dst, hipStream_t st);

void launch_synth_uniform(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, hipStream_t st);
void launch_synth_clustered(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers,
                            float sigma, hipStream_t st);

} // namespace mvs
