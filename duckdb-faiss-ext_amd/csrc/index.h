// csrc/index.h -- device-native index classes behind the C ABI (include/mi355_faiss.h).
#pragma once
#include "common.h"

#include "../../include/mi355_faiss.h"

#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace mvs {

struct DevBuf { // grow-only device scratch (contents NOT preserved on growth)
	void *p = nullptr;
	size_t cap = 0;
	void reserve(size_t bytes);
	void release();
	~DevBuf() {
		release();
	}
};

// ring of pinned host buffers: the caller's rows are copied here so add()/search() can return /
// overlap while hipMemcpyAsync drains the slot (SURVEY.md 8f-1 ingest staging)
struct PinnedRing {
	static constexpr int NB = 4;
	static constexpr size_t SLOT_BYTES = 8u << 20;
	void *buf[NB] = {nullptr, nullptr, nullptr, nullptr};
	size_t cap[NB] = {0, 0, 0, 0};
	hipEvent_t ev[NB] = {nullptr, nullptr, nullptr, nullptr};
	int next = 0;
	int acquire(size_t bytes);
	void release(int slot, hipStream_t st);
	void drop_events();
	~PinnedRing();
};

struct SelectorHolder {
	DevBuf buf;
	std::vector<int64_t> sorted;
	SelectorDev upload(const mvs_search_params *p, hipStream_t st);
};

// Host-side image of an index: what faiss::write_index stores (impl/index_write.cpp).  Used by index_io.hip and by
// cross-device clones (faiss::gpu::index_cpu_to_gpu), both of which go through host memory.
struct HostIndex {
	int kind = 0, d = 0, metric = 0;
	float metric_arg = 0.f;
	int64_t ntotal = 0;
	bool is_trained = true;
	std::vector<float> rows;       // Flat: [ntotal][d]
	std::unique_ptr<HostIndex> sub; // IDMap: wrapped index; IVF: quantizer; HNSW: storage (a Flat image)
	std::vector<int64_t> ids;      // IDMap: id_map
	bool idmap2 = false;
	int64_t nlist = 0, nprobe = 1; // IVF
	std::vector<std::vector<int64_t>> list_ids;
	std::vector<std::vector<float>> list_codes; // per list [n][d]
	// HNSW (struct HNSW fields in FAISS's order)
	std::vector<double> assign_probas;
	std::vector<int32_t> cum_nneighbor_per_level, levels, neighbors;
	std::vector<uint64_t> offsets;
	int32_t entry_point = -1;
	int max_level = -1, efConstruction = 40, efSearch = 16;
};

// indexes constructed while one of these is alive (same thread) live on `dev` instead of MVS_DEVICE
struct CtorDevice {
	int prev;
	explicit CtorDevice(int dev);
	~CtorDevice();
};

class IndexBase {
public:
	int kind, d, metric;
	int64_t ntotal = 0;
	bool is_trained = true;
	int device = 0;
	hipStream_t stream = nullptr;
	int64_t label_offset = 0;
	float metric_arg = 0.f; // faiss::Index::metric_arg (Lp exponent); the glue leaves it at 0
	mvs_kernel_info kinfo {};

	IndexBase(int kind, int d, int metric);
	virtual ~IndexBase();
	void use_device() const;
	Tuning tune_;                                  // this index's tuning knobs (csrc/common.h); current for the thread after use_device()
	bool set_tuning(const char *key, int64_t v); // the option keys that are tuning knobs (csrc/index.hip)
	// a copy / shard set / shadow of an index keeps its knobs (they reverted to the defaults silently); wrappers pass it on
	virtual void adopt_tuning(const Tuning &t) {
		tune_ = t;
	}

	// faiss::Index virtuals the glue calls (src/faiss_extension.cpp:396,510,512,583,607,609,631)
	virtual void train(int64_t n, const float *x);
	virtual void add(int64_t n, const float *x) = 0;
	virtual void add_with_ids(int64_t n, const float *x, const int64_t *ids);
	virtual void search(int64_t nq, const float *x, int64_t k, float *D, int64_t *I, const mvs_search_params *params);
	// device-resident variants
	virtual void add_device(int64_t n, const float *d_x, hipStream_t st) = 0;
	virtual void add_with_ids_device(int64_t n, const float *d_x, const int64_t *d_ids, hipStream_t st);
	virtual void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                           const mvs_search_params *params, hipStream_t st) = 0;
	// search on behalf of an IndexIDMap wrapper: selector tests d_idmap[internal], labels = d_idmap[internal]
	virtual void search_mapped(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                           const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st) = 0;
	// IVF row shard of a ShardedIndex: for the flagged queries of the LAST search (d_flag = {count, queries...}; the coarse assignment
	// of that search is still in place) this shard's first k rows not worse than T[f] in arrival order -- value, stored id, probe
	// rank, -1 padded (csrc/ivf_ties.hip EMIT mode)
	virtual void tie_emit(const int *d_flag, int nf, const float *d_x, const float *d_T, int64_t k, const mvs_search_params *params,
	                      const int64_t *d_idmap_sel, float *d_v, int64_t *d_id, int *d_p, hipStream_t st);
	// an IVF index as the internal clustering of a Flat L2 index (csrc/ivf.hip flat_shadow_search).  0: run; 1: this call's shape is
	// not served (the caller takes its normal path for the batch, the shadow stays); 2: this DATA is not served (stream / bucket limits)
	virtual int flat_shadow_search(int64_t, const float *, int64_t, const float *, float *, int64_t *, const int64_t *, int64_t,
	                               const unsigned *, int *, int *, int, hipStream_t) {
		return 1;
	}
	virtual int64_t flat_shadow_max_queries(int) { // queries one flat_shadow_search call takes (distance matrix, pair count)
		return 0;
	}
	virtual size_t device_bytes() const { // HBM held by the index's row stores (approximate; mvs_index_shadow_stats)
		return 0;
	}
	virtual void to_device(int new_device) = 0;
	virtual IndexBase *clone(int on_device) = 0; // deep copy living on `on_device`
	virtual void to_host(HostIndex &out) = 0;    // host image (write_index, cross-device clone)
	virtual void set_label_offset(int64_t off) {
		label_offset = off;
	}
	virtual const float *last_batch_ptr() const { // IVF: the device pointer of the batch whose coarse assignment the index still holds (tie_emit)
		return nullptr;
	}
	virtual bool named_stat(const char *, int64_t *) { // index-specific counters of mvs_index_get_stat (HNSW: hnsw_build_distances | hnsw_build_shortcuts)
		return false;
	}
	virtual bool probe_stats(int64_t *, int64_t *, int64_t *, int64_t *) { // IVF: (query, list) pairs of the last search / of those, scanned (mvs_index_ivf_probe_stats)
		return false;
	}
	virtual bool collect_stats(int64_t *, int64_t *, int64_t *) { // coarse-filter census of an IVF index (mvs_index_collect_stats)
		return false;
	}
	virtual bool set_option(const char *, int64_t) {
		return false;
	}

	// HIP-event timing of the dominant kernel (bench.py roofline)
	virtual void set_timing(bool on) {
		timing_enabled = on;
	}
	virtual void resolve_timing(int *count, double *total_ms) {
		resolve_kernel_timing(count, total_ms);
	}

protected:
	PinnedRing pinned;
	DevBuf ws_hx, ws_hD, ws_hI;
	bool timing_enabled = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_events;
	int timing_count = 0;
	double timing_total_ms = 0;
	void begin_kernel_timing(hipStream_t st);
	void end_kernel_timing(hipStream_t st);
	void resolve_kernel_timing(int *count, double *total_ms);
};

class FlatIndex : public IndexBase {
public:
	FlatGeom geom;
	float *vecs = nullptr;  // [cap][dp]
	float *norms = nullptr; // [cap]
	int64_t cap = 0;
	bool force_direct = false; // test hook: per-pair kernel for any nq
	bool force_staged = false; // test hook: LDS-staged flat_direct kernel instead of the packed scan kernel

	FlatIndex(int d, int metric);
	~FlatIndex() override;
	void reset();
	void copy_rows_to_host(float *out); // logical [ntotal][d] rows (storage format undone)
	void add(int64_t n, const float *x) override;
	void add_device(int64_t n, const float *d_x, hipStream_t st) override;
	void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                   const mvs_search_params *params, hipStream_t st) override;
	void search_mapped(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                   const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st) override;
	void to_device(int new_device) override;
	IndexBase *clone(int on_device) override;
	void to_host(HostIndex &out) override;
	bool set_option(const char *key, int64_t v) override;
	void search_flat(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                 const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st);
	void search_extra_metric(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                         const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st);

	DevBuf ws_q, ws_qn, ws_pd, ws_pi, ws_gthr, ws_add, ws_xi, ws_pbnd;
	DevBuf ws_flag, ws_tie; // inner-product boundary ties: flagged queries + tie-pass scratch
	// bf16x3 prefilter (csrc/flat_bf16.hip): rows as bf16 hi/lo, derived lazily from `vecs` before a search
	unsigned short *vecs_bf = nullptr; // [bf_cap][2 dp]
	int64_t bf_cap = 0, bf_rows = 0;
	unsigned *d_max_norm_bits = nullptr; // largest squared row norm among the first bf_rows rows (float bits)
	int prefilter_mode = -1;             // option "prefilter": -1 auto, 0 off, 1 whenever the kernel supports the shape
	int pf_margin = 6;                   // spare candidate ranks beyond k (option "pf_margin")
	bool pf_suppressed = false;          // set while the queries the proof rejected are re-run on the exact kernel
	bool pf_pair_branch = false;         // ... and whether their batch was one FAISS sends down its per-pair branch (nq < 20)
	int64_t pf_last_fallback = 0;        // diagnostics: queries of the last search that were re-run
	int64_t pf_queries_total = 0, pf_fallback_total = 0;
	float pf_max_rel_err = 0.f; // largest observed |approx - exact| / (||x|| ||y||) among re-scored candidates
	DevBuf ws_pfq, ws_cand, ws_ex, ws_fail, ws_fb;
	// bf16 coarse filter (csrc/flat_collect.hip): rows as bf16 only, candidate stream, per-query bounds
	unsigned short *vecs_h1 = nullptr; // [h1_cap + 192][dp] centred rows as bf16
	float *beta_h1 = nullptr;          // [h1_cap + 192] -||y - mu||^2 (L2) or <mu, y> (inner product)
	float *mu_h1 = nullptr;            // [dp] the centre (mean of the rows present at the first build)
	int64_t h1_cap = 0, h1_rows = 0;
	int64_t cl_queries_total = 0, cl_candidates_total = 0, cl_overflows = 0, cl_last_candidates = 0, cl_heavy_total = 0;
	int cl_stream_cap_per_query = 0; // option cl_stream_cap (0: 4096 entries per query, at least 2^20)
	int64_t cl_cap_hint = 0;         // entries per query the last overflow asked for (the next search starts there)
	const unsigned long long *cl_sorted = nullptr;  // the re-scored candidate list of the last coarse-filter batch (in ws_stream)
	const unsigned long long *tie_sorted = nullptr; // != nullptr: resolve_ip_ties reads A_k off that list instead of scanning again
	const unsigned long long *tie_bucket = nullptr; // ... or off the bucketed finish's per-query key buckets (round 6)
	const unsigned long long *cl_fb_keys = nullptr; // (the key buckets of the last inner-product bucketed finish)
	const unsigned *tie_bcount = nullptr;
	int tie_bpitch = 0;
	bool tie_from_candidates = true; // option tie_from_candidates = 0: inner-product ties re-scan the database (A/B, tests)
	bool cl_k32 = true;       // 16 < k <= 32 at d <= 128 on the coarse filter with 32 row classes (option cl_k32; 0: bf16x3 / f32 as before)
	bool cl_small_path = true; // batches of <= 256 queries on the one-wavefront-per-segment kernel (option cl_small_path)
	DevBuf ws_e2, ws_stream, ws_sorttmp, ws_seg, ws_rowmask, ws_items1, ws_qcount;
	void ensure_bf16_rows(hipStream_t st);
	void ensure_h1_rows(hipStream_t st);
	int *d_outl = nullptr; // [1 + CL_OUTL_CAP] outlier rows of the coarse-filter store: count, rows (csrc/flat_collect.hip "outlier rows")
	int h1_outliers = 0;   // ... their number as the host last read it (after a conversion)
	int64_t outl_total = 0; // (diagnostics: mvs_index_get_stat "flat_outlier_rows")
	bool outlier_rows = true; // option outlier_rows
	// IVF coarse quantisation: the np nearest rows (L2, FAISS order) by distance matrix + selection (csrc/coarse_select.hip);
	// false: shape not served, the caller uses search_device
	// need_matrix: the caller reads coarse_matrix() afterwards (the Flat shadow's proof); otherwise L2 quantisers of 16 < d <= 128 take the
	// bf16 filter + exact re-scoring of csrc/coarse_bf16.hip (round 6: no distance matrix at all, same output bit for bit)
	bool coarse_topk(int64_t nq, const float *d_x, int64_t np, float *d_D, int64_t *d_I, hipStream_t st, bool need_matrix = true);
	DevBuf ws_cb16;
	int64_t cb16_queries = 0; // queries served by csrc/coarse_bf16.hip (diagnostics)
	size_t cb16_stats_off = 0; // byte offset of the exhaustive-query counter in ws_cb16
	int64_t cb16_last_nq = 0;  // queries / offset of the per-query candidate counts of the last call (diagnostics)
	size_t cb16_ccount_off = 0;
	// what the last coarse_topk left behind (csrc/ivf.hip flat_shadow_search): the [nq][ntotal] distance matrix -- whole only when
	// the batch fitted one chunk --, and the rows' squared norms
	const float *coarse_matrix() const {
		return (const float *)ws_q.p;
	}
	bool coarse_matrix_covers(int64_t nq) const {
		return ntotal > 0 && nq <= std::max<int64_t>(64, ((int64_t)512 << 20) / (ntotal * 4) / 64 * 64);
	}
	const float *row_norms() const {
		return norms;
	}
	// defer_count: nothing waits for the candidate count between the scan and the re-scoring (device-count mode); the count is
	// copied to h_flag_count[10..11] asynchronously and the caller checks it against cl_deferred_cap after ITS stream
	// synchronisation -- on an overflow it runs the search again with defer_count = false (the synchronous overflow handling)
	bool collect_candidates(int64_t nq, const float *d_x, int kk, float **pd1, int32_t **pi1, int *fail_cnt, int *fail_q,
	                        const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st, bool defer_count = false, int kf = 0);
	int64_t cl_deferred_cap = 0;
	bool cl_defer = true; // option cl_defer_count
	// Round 5, bucketed finish of the d = 128 L2 coarse filter: final-bound filter -> the survivors into per-query row buckets -> one
	// wavefront per query re-scores them -> one wavefront per query selects and prints (csrc/ivf_collect.hip, shared with the IVF path):
	// no radix sort, no segments, ~ 4 x fewer rows re-scored.  Set by search_prefilter_pass, consumed by collect_candidates.
	bool cl_seed_stage = true;   // option cl_seed_stage: the register pre-pass stages its class maxima per row split, one reduce kernel publishes them
	bool cl_wide_refilter = true; // option cl_wide_refilter: the final-bound filter in front of the sorted pipeline of the 512 < d <= 1536 stores (flat_bf16_big_kernel)
	bool cl_fbucket = true;      // option cl_fbucket
	bool cl_fbucket_off = false; // a query's bucket overflowed on this index's data: the sorted pipeline from then on
	int cl_bigk_whole = -1, cl_bigk_per = 0; // options (A/B of the big lists' pass A: rows looked at, rows per split that decide)
	bool cl_bigk = true; // option cl_bigk: lists of 129 .. 2048 entries on the coarse filter (bounds from row ranges, frozen scan, segmented sort); 0: the exact kernels
	int cl_fpitch = 256;         // bucket entries per query
	float *cl_out_D = nullptr;
	int64_t *cl_out_I = nullptr;
	const int64_t *cl_out_map = nullptr;
	int64_t cl_out_off = 0;
	const TieFlags *cl_out_flags = nullptr; // (inner product: the boundary-tie flags of the search, or null)
	int cl_out_kout = 0;                    // (inner product: entries printed per query; the selection carries kk >= kout)
	bool cl_emitted = false;     // collect_candidates wrote the final lists itself (the caller skips its emission)
	int64_t cl_last_rescored = -1, cl_rescored_total = 0, cl_rescored_queries = 0, cl_admitted_in_fb = 0; // bucketed finish: survivors of the final-bound filter
	bool cl_wrf_used = false;    // the last collect_candidates compacted the stream with the final-bound filter (wide stores)
	bool cl_report_cnt = false;  // the scan's entry count (and the bucket header) still has to reach the host: launch_collect_report does it
	unsigned long long *h_cl_hdr = nullptr; // pinned copy of the control block's header (bucket statistics)
	DevBuf ws_fbk, ws_fbr, ws_seed;
	bool cl_prep1 = true;        // option cl_prep1: one fused per-query preparation kernel in front of the d <= 128 coarse filter
	double cl_est_per_query = 0; // candidates per query of the last search: sizes the next search's sort (collect_sort_estimate)
	int cl_skip = 0, cl_skip_len = 0; // searches that bypass the coarse filter after it gave up on this index's data (doubling, <= 64)
	// ---- shadow clustering (round 5): a Flat L2 index whose rows CLUSTER keeps an IVF index of the same rows and answers large
	// batches through it -- nprobe nearest lists by the coarse filter with per-list centring, then a proof per query that no other
	// list can matter; what cannot be proven is re-run on the Flat kernels.  Built lazily when the global-centring filter admits
	// thousands of candidates per query (clustered rows with large norms: 9 764 at the C3 mixture, 45.6 ms per batch).
	IndexBase *shadow = nullptr;
	int64_t shadow_rows = -1;  // ntotal the shadow holds
	int shadow_state = 0;      // 0 not wanted yet, 1 wanted / in use, -1 given up on this data (too many queries could not be proven)
	int shadow_mode = -1;      // option flat_shadow: -1 auto, 0 never, 1 from the first large search on
	int shadow_nprobe = 32;    // option flat_shadow_nprobe
	int64_t shadow_queries = 0, shadow_unproven = 0;
	int64_t shadow_trained_rows = 0; // ntotal when the clustering was trained (re-trained once the index has doubled)
	uint64_t mut_gen = 0, shadow_gen = 0; // generation of the rows (bumped by reset()) / the one the shadow was built from
	double shadow_build_seconds = 0;      // time spent building / extending the shadow (inside searches), mvs_index_shadow_stats
	int64_t shadow_builds = 0, shadow_extends = 0;
	bool shadow_sync(hipStream_t st);
	bool shadow_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   const int64_t *d_idmap, const int64_t *out_map, int64_t out_off, hipStream_t st);
	void drop_shadow();
	void set_timing(bool on) override {
		timing_enabled = on;
		if (shadow)
			shadow->set_timing(on);
	}
	void resolve_timing(int *count, double *total_ms) override { // (the dominant kernel of a shadow search is timed by the shadow index)
		resolve_kernel_timing(count, total_ms);
		if (shadow) {
			int c2 = 0;
			double t2 = 0;
			shadow->resolve_timing(&c2, &t2);
			if (count)
				*count += c2;
			if (total_ms)
				*total_ms += t2;
		}
	}
	void drop_bf16_rows();
	bool search_prefilter(int64_t nq, const float *d_x, int64_t k_user, int64_t kk, float *d_D, int64_t *d_I,
	                      const mvs_search_params *params, const int64_t *d_idmap, const int64_t *out_map, int64_t out_off,
	                      const TieFlags *flp, hipStream_t st);
	bool search_prefilter_pass(int64_t nq, const float *d_x, int64_t k_user, int64_t kk, float *d_D, int64_t *d_I,
	                           const mvs_search_params *params, const int64_t *d_idmap, const int64_t *out_map, int64_t out_off,
	                           const TieFlags *flp, hipStream_t st, bool defer, bool *overflow);
	int *h_flag_count = nullptr; // pinned
	// row shard of a ShardedIndex (csrc/sharded.hip): results are the shard's ROW numbers in the pure order, no tie pass
	// (the sharded index resolves ties across shards); an id map passed to search_flat then only feeds the selector
	bool raw_rows = false;
	// tie pass for `nf` flagged queries: out[f][0..k) = the k smallest row ids with score >= d_T[f] (ascending, -1 padded)
	void tie_candidates(int64_t nf, const float *d_xf, const float *d_T, int64_t k, int64_t *d_rows_out, SelectorDev sel,
	                    const int64_t *d_selmap, hipStream_t st);
	void offset_rows(int64_t *d_rows, int64_t total, hipStream_t st); // += label_offset on valid entries
	SelectorDev upload_selector(const mvs_search_params *p, hipStream_t st) {
		return selector.upload(p, st);
	}
	bool ip_exact_ties = true;   // option "ip_exact_ties" = 0: keep the pure (score desc, id asc) order (raw shard lists)
	void resolve_ip_ties(int64_t nq, const float *d_x, int64_t k, const TieFlags &fl, SelectorDev sel,
	                     const int64_t *d_idmap, float *d_D, int64_t *d_I, hipStream_t st, int64_t kraw = 0);
	SelectorHolder selector;
	hipStream_t last_search_stream = nullptr;
	bool have_last_search = false;
	void grow(int64_t need, hipStream_t st);
	// ---- ingest staging (round 6): DataChunk-sized adds are collected in a pinned slot and sent to the device once per slot ----
	PinnedRing add_ring;
	int pend_slot = -1, add_flip = 0;
	size_t pend_bytes = 0;
	int64_t pend_rows = 0, pend_row0 = 0; // rows of ntotal that are staged, not yet on the device / the first of them
	bool lazy_adds = true;                // option lazy_adds = 0: every add() goes to the device at once (round 5)
	int64_t add_flushes = 0;
	bool flush_adds();                    // true: rows were sent (on `stream`)
	struct Retired {
		void *a, *b;
		hipEvent_t done;
	};
	std::vector<Retired> retired; // row stores replaced by a growth: freed once the copy out of them has finished
	void retire_buffers(hipStream_t st, void *a, void *b);
	void reap_retired(bool wait);
};

class IDMapIndex : public IndexBase {
public:
	IndexBase *sub; // owned
	int64_t *ids = nullptr;
	int64_t idcap = 0;
	PinnedRing id_ring; // (round 6) the ids of DataChunk-sized adds are staged like their rows
	int idp_slot = -1;
	size_t idp_bytes = 0;
	int64_t idp_row0 = 0;
	void flush_ids();

	explicit IDMapIndex(IndexBase *sub);
	~IDMapIndex() override;
	void adopt_tuning(const Tuning &t) override {
		tune_ = t;
		sub->adopt_tuning(t);
	}
	void train(int64_t n, const float *x) override;
	void add(int64_t n, const float *x) override;
	void add_with_ids(int64_t n, const float *x, const int64_t *ids) override;
	void add_device(int64_t n, const float *d_x, hipStream_t st) override;
	void add_with_ids_device(int64_t n, const float *d_x, const int64_t *d_ids, hipStream_t st) override;
	void search(int64_t nq, const float *x, int64_t k, float *D, int64_t *I, const mvs_search_params *params) override;
	void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                   const mvs_search_params *params, hipStream_t st) override;
	void search_mapped(int64_t, const float *, int64_t, float *, int64_t *, const mvs_search_params *, const int64_t *,
	                   hipStream_t) override {
		throw_faiss("mvs::IDMapIndex::search_mapped", __FILE__, "nested IDMap is not supported");
	}
	void to_device(int new_device) override;
	IndexBase *clone(int on_device) override;
	void to_host(HostIndex &out) override;
	void set_label_offset(int64_t) override {
	}
	bool set_option(const char *key, int64_t v) override {
		return sub->set_option(key, v);
	}
	void set_timing(bool on) override {
		sub->set_timing(on);
	}
	void resolve_timing(int *count, double *total_ms) override {
		sub->resolve_timing(count, total_ms);
		kinfo = sub->kinfo;
	}

	void adopt_ids(const int64_t *xids, int64_t n); // image load: rows are already in `sub`

private:
	void grow_ids(int64_t need, hipStream_t st);
};

IndexBase *index_factory(int d, const char *description, int metric);
// rebuild a device index from its host image on the CURRENT default device (MVS_DEVICE / 0), or on `device` if >= 0
IndexBase *index_from_host(const HostIndex &h, int device = -1);
void stream_wait(hipStream_t waiter, hipStream_t signal);

// csrc/ivf.hip
IndexBase *make_ivf_index(int d, const std::string &desc, int metric); // nullptr if desc is not an IVF string
IndexBase *ivf_quantizer_of(IndexBase *ix);
int64_t ivf_nlist_of(IndexBase *ix);
bool ivf_get_centroids(IndexBase *ix, float *out);
bool ivf_set_centroids(IndexBase *ix, const float *c);
IndexBase *ivf_from_host(const HostIndex &h, int device);
// csrc/hnsw.hip
IndexBase *make_hnsw_index(int d, const std::string &desc, int metric);
bool hnsw_set_ef_construction(IndexBase *ix, int v);
int hnsw_get_ef_construction(IndexBase *ix); // -1 if not HNSW
int64_t hnsw_graph_info(IndexBase *ix, int *max_level, int *entry_point); // neighbour slots, -1 if not HNSW
bool hnsw_walk_stats(IndexBase *ix, double *evaluations, double *f32_rows, double *bf16_rows); // counters of the last timed search
bool hnsw_get_graph(IndexBase *ix, int32_t *levels, int64_t *offsets, int32_t *neighbors);
IndexBase *hnsw_from_host(const HostIndex &h, int device);
// csrc/io.cpp-ish (index_io.hip)
void write_index_file(IndexBase *ix, const char *filename);
IndexBase *read_index_file(const char *filename);
// csrc/sharded.hip: one index over several devices behind the same surface
std::vector<int> shard_devices_from_env(); // env MVS_DEVICES="0,1,...,7" (empty / unset: no sharding)
IndexBase *make_sharded_index(int d, const char *desc, int metric, const std::vector<int> &devices);
IndexBase *shard_from_host(const HostIndex &h, const std::vector<int> &devices);
bool is_sharded(const IndexBase *ix);
IndexBase *sharded_inner_view(IndexBase *ix); // the index the glue's dynamic_casts should see (ix itself if unsharded)
void sharded_for_each(IndexBase *ix, const std::function<void(IndexBase *)> &f);
int sharded_info(const IndexBase *ix, int *devices, int max_devices, int64_t *rows_per_shard, int64_t *last_flagged);
// csrc/merge_host.hip
void merge_shards_raw_host(int metric, int64_t n, int64_t kk, int nshard, const float *D, const int64_t *I, float *D_out,
                           int64_t *I_out);
void finish_ip_ties_host(int64_t n, int64_t k, int64_t kk, const float *raw_v, const int64_t *raw_g, int64_t nf,
                         const int64_t *fq, const int64_t *first, float *D_out, int64_t *I_out);
void merge_shards_host(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out,
                       int64_t *I_out);

// csrc/flat_collect.hip
bool collect_supported(const FlatGeom &g);
void launch_collect_mean(const FlatGeom &g, const float *d_vecs, int64_t nrows, float *d_mu, hipStream_t st);
void launch_rows_to_bf16_hi(const FlatGeom &g, int metric, const float *d_vecs, int64_t row0, int64_t nrows, const float *d_mu,
                            unsigned short *d_bf, float *d_beta, const float *d_norms, unsigned *d_max_norm_bits,
                            hipStream_t st, int *d_outl = nullptr);
// outlier rows of the coarse-filter store (csrc/flat_collect.hip "outlier rows")
void launch_collect_outlier_threshold(const float *d_norms, int64_t nrows, const float *d_mu, int dp, unsigned *d_max_norm_bits, hipStream_t st);
void launch_collect_append_outliers(const int *d_outl, int n_outliers, int64_t nq, unsigned long long *d_stream, float *d_stream_s,
                                    unsigned long long *d_cnt, int64_t cap, const unsigned long long *d_rowmask, hipStream_t st);
size_t collect_qfrag_bytes(const FlatGeom &g, int64_t nq);
int collect_wide_max_classes(int dp1); // 128: the wide store's kernel has the 4 x 32-class instance (csrc/flat_collect_wide.hip)
const char *collect_wide_kernel_name(int dp1); // "flat_bf16_big_kernel" / "flat_bf16_wide_kernel" for a store pitch > 128
int collect_store_dims(int d); // 128 / 256 / 384 / 512: row pitch of the bf16 store; 0: d is not served (csrc/flat_collect_wide.hip)
int collect_wide_qblock(int dp1);
void launch_rows_to_bf16_wide(int metric, const float *d_vecs, int sdp, int interleaved, int d, int dp1, int64_t row0, int64_t nrows,
                              const float *d_mu, unsigned short *d_bf, float *d_beta, const float *d_norms,
                              unsigned *d_max_norm_bits, hipStream_t st, int *d_outl);
size_t collect_qfrag_bytes_ex(int dp1, int qblock, int64_t nq);
void launch_collect_pack_queries_ex(int d, int dp1, int qblock, int metric, const float *d_x, int64_t nq, const float *d_mu,
                                    void *d_qf, hipStream_t st);
void launch_collect_pack_queries(const FlatGeom &g, int metric, const float *d_x, int64_t nq, const float *d_mu, void *d_qf,
                                 hipStream_t st);
void launch_collect_bounds(int metric, const float *d_x, int64_t nq, int d, const float *d_mu,
                           const unsigned *d_max_norm_bits, float *d_e2, int *d_fail_cnt, int *d_fail_q, hipStream_t st);
int collect_slot_stride(int kk, int dp1 = 128); // class slots per query: 16 | 32 | 128 (kk > 32; wide stores: where the kernel has the instance)
int collect_max_k(int d); // largest k (+1 with tie detection) the coarse filter serves at this d: 128, 16 (k-split kernels), or 0
int launch_collect_drop_heavy(const unsigned long long *d_stream, int64_t n, int64_t nq, int share, int *d_qcount, float *d_e2,
                              int *d_fail_cnt, int *d_fail_q, hipStream_t st);
void launch_collect_prepare(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms,
                            int64_t n, int64_t nq, int kk, const float *d_e2, unsigned *d_gslot,
                            unsigned long long *d_stream_cnt, const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st, bool cnt_zeroed = false, bool slots_ready = false,
                            float *d_seed_stage = nullptr);
size_t collect_seed_stage_bytes(int64_t nq); // [64][nq][16] floats: the register pre-pass's class maxima per row split (csrc/flat_collect.hip)
void launch_collect_query_prep(int metric, const float *d_x, int64_t nq, int d, const float *d_mu, const unsigned *d_max_norm_bits,
                               void *d_qf, float *d_qn, float *d_e2, int *d_fail_cnt, int *d_fail_q, unsigned *d_gslot, int stride,
                               int *d_ctl_hdr, int *d_ctl_seg, hipStream_t st);
size_t collect_bound_table_bytes(int64_t nq);
size_t collect_rowmask_bytes(int64_t n);
void launch_collect_rowmask(SelectorDev sel, const int64_t *d_idmap, int64_t n, unsigned long long *d_mask, hipStream_t st);
void launch_collect_scan(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms,
                         int64_t n, int64_t nq, int kk, const float *d_e2, unsigned *d_gslot, unsigned long long *d_stream,
                         unsigned long long *d_stream_cnt, int64_t stream_cap, const unsigned long long *d_rowmask, float *d_pbnd,
                         hipStream_t st, int *grid_out, int *nsplit_out, int *lds_out, float *d_stream_s = nullptr, bool frozen = false);
// lists beyond 128 entries (d <= 128 store): pass A -- T - 2E per query from nranges row ranges' class slots, into the scan's bound table;
// the scan then runs with frozen = true (csrc/flat_collect.hip "lists beyond 128 entries")
void launch_collect_big_bounds(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms, int64_t n,
                               int64_t nq, int kf, int nranges, int64_t range_rows, const float *d_e2, unsigned *d_gslot,
                               const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st);
// (d <= 128 store: pass A on the register pre-pass kernel -- nsplits = ceil(k / 8) strides, 16 class maxima each, d_stage [nsplits][nq][16])
void launch_collect_big_bounds_seed(const FlatGeom &g, int metric, const void *d_qf, const unsigned short *d_rows, const float *d_norms, int64_t n,
                                    int64_t nq, int kf, int nsplits, int64_t split_len, const float *d_e2, float *d_stage,
                                    const unsigned long long *d_rowmask, float *d_pbnd, hipStream_t st);
size_t collect_select_big_temp_bytes(int64_t ncand, int64_t nq);
void launch_collect_select_big(int metric, unsigned long long *d_keys, unsigned long long *d_out, int64_t ncand, const int *d_seg, int64_t nq,
                               int kk, void *d_temp, size_t temp_bytes, float *d_pd1, int32_t *d_pi1, hipStream_t st);
// thr[q] = B - 2E from the class slots as the scan left them (csrc/flat_collect.hip): the final-bound filter of the bucketed finish
void launch_mfma_bf16_probe(const unsigned short *d_A, const unsigned short *d_Bt, const float *d_C, float *d_D, int64_t ntiles, hipStream_t st);
void launch_collect_report(const void *d_hdr, const int *d_fail_cnt, const unsigned *d_maxnorm, int *h_flags, void *h_hdr, bool with_cnt,
                           hipStream_t st);
void launch_stream_refilter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt, const float *d_thr,
                            unsigned long long *d_out, unsigned long long *d_out_cnt, hipStream_t st);
void launch_collect_final_thr(const unsigned *d_gslot, int d, int kk, const float *d_e2, int64_t nq, float *d_thr, hipStream_t st);
// Entries the deferred sort of a search is launched with, from the candidates per query c of the index's previous search: the
// margin shrinks with the batch (the mean of nq heavy-tailed per-query counts), 17 % + 16 per query at 10 000 queries, 40 % at 64
// (round 4, first cut: 30 % + 64 per query whatever the batch -- at C3's 149 per query the sort ran over 75 % more entries than it had)
static inline int64_t collect_sort_estimate(double c, int64_t nq) {
	const double per = c * (1.15 + 2.0 / sqrt((double)std::max<int64_t>(nq, 1))) + 16.0;
	return ((int64_t)(per * (double)nq) + 65535) / 65536 * 65536;
}
size_t collect_sort_temp_bytes(int64_t ncand, int64_t nq);
size_t collect_sort_temp_bytes_est(int64_t n_est, int64_t nq);
void launch_collect_group_est(unsigned long long *d_stream, unsigned long long *d_sorted, const unsigned long long *d_cnt,
                              int64_t n_est, void *d_temp, size_t temp_bytes, int64_t nq, int *d_seg, hipStream_t st,
                              bool seg_zeroed = false);
void launch_collect_rescore(int metric, unsigned long long *d_stream, unsigned long long *d_sorted, int64_t ncand, void *d_temp,
                            size_t temp_bytes, int64_t nq, int kk, const float *d_x, const FlatGeom &g, const float *d_vecs,
                            const float *d_norms, const float *d_qn, int *d_seg, float *d_pd1, int32_t *d_pi1,
                            bool per_pair, hipStream_t st, const unsigned long long *d_cnt = nullptr, bool seg_zeroed = false);
void launch_collect_tie_rows_bucket(const unsigned long long *d_bucket, const unsigned *d_bcount, int pitch, const int *d_flag_query,
                                    const float *d_T, int nf, int k, int64_t *d_first, hipStream_t st);
void launch_collect_tie_rows(const unsigned long long *d_sorted, const int *d_seg, int64_t nq, const int *d_flag_query,
                             const float *d_T, int nf, int k, int64_t *d_first, hipStream_t st);
bool coarse_select_supported(int64_t nlist, int64_t np);
void launch_coarse_select(const float *d_x, int64_t nq, int d, const float *d_cent, int sdp, int interleaved, int64_t nlist,
                          const float *d_qn, const float *d_cn, int64_t np, int is_l2, float *d_D, float *d_pd, int32_t *d_pi,
                          hipStream_t st, float *d_outD = nullptr, int64_t *d_outI = nullptr, int64_t label_offset = 0);
void launch_collect_group(unsigned long long *d_stream, unsigned long long *d_sorted, int64_t ncand, void *d_temp,
                          size_t temp_bytes, int64_t nq, int *d_seg, hipStream_t st, bool seg_zeroed = false);
void launch_collect_select(int metric, const unsigned long long *d_keys, const int *d_seg, int64_t nq, int kk, float *d_pd1,
                           int32_t *d_pi1, hipStream_t st);
// csrc/ivf_collect.hip
void launch_ivf_rows_to_bf16(const float *d_res, int64_t nrows, int d, const int *d_list_of_blk64, unsigned short *d_bf,
                             float *d_beta, unsigned *d_list_max_bits /* [2 nlist] */, int64_t nlist, hipStream_t st);
size_t ivf_collect_xi_bytes(int max_items);
void launch_ivf_collect_pack2(int metric, const float *d_x, int d, int64_t nq, const int *d_slots0, const void *d_items0, void *d_xi0,
                              float *d_igamma0, float *d_ie20, const void *d_items1, const int *d_nitems1, int max_items1,
                              const int *d_qidx1, void *d_xi1, float *d_igamma1, float *d_ie21, const float *d_cent,
                              const int *d_list_of_blk64, const unsigned *d_list_max_bits, int *d_qfail, int64_t nlist, unsigned *d_gslot,
                              int nclass, int *d_ctl_hdr, int *d_flag_cnt, hipStream_t st);
void launch_ivf_collect_scan(const void *d_items, const int *d_nitems, int max_items, const int *d_qidx, const void *d_xi,
                             const float *d_igamma, const float *d_ie2, const unsigned short *d_rows_bf, const float *d_beta,
                             unsigned *d_gslot, unsigned long long *d_stream, unsigned long long *d_stream_cnt,
                             int64_t stream_cap, int kk, int seg_rows, int nseg, int collect, const unsigned *d_rowmask,
                             hipStream_t st, float *d_stream_u = nullptr, const float *d_bfix = nullptr);
// final-bound filter between the scan and the exact stage (csrc/ivf_collect.hip): entries whose s + E is below the bound the scan ended with are dropped
// csrc/coarse_bf16.hip: the coarse quantiser as a bf16 filter + exact re-scoring inside one workgroup per 32 queries
bool coarse_bf16_supported(int d, int64_t nlist, int64_t np);
size_t coarse_bf16_cand_bytes(int64_t nq);
size_t coarse_bf16_cls_bytes(int64_t nq, int64_t nlist, int64_t np);
int coarse_bf16_slices(int64_t nq, int64_t nlist, int64_t np);
void launch_coarse_bf16(const float *d_x, int64_t nq, int d, const void *d_qf, const float *d_qn, const float *d_e2, const unsigned short *d_yb,
                        const float *d_beta, const float *d_cent, int sdp, int interleaved, const float *d_cn, int64_t nlist, int64_t np,
                        unsigned short *d_cand, int *d_ccount, float *d_cls, float *d_outD, int64_t *d_outI, int64_t label_offset,
                        unsigned long long *d_stats, hipStream_t st);
void launch_ivf_refilter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt,
                         const unsigned *d_gslot, int nclass, int kf, int64_t nq, float *d_bf, unsigned long long *d_out,
                         unsigned long long *d_out_cnt, hipStream_t st);
void launch_ivf_bucket_finish(int metric, const unsigned long long *d_strm, int64_t ncand, const unsigned long long *d_cnt,
                              unsigned long long *d_bucket, unsigned *d_bcount, int bpitch, int64_t nq, const float *d_x, int d,
                              const float *d_rows_csr, int dp_csr, const int *d_perm, int kk, float *d_pd, int64_t *d_pi,
                              const int64_t *d_rowids, const int64_t *d_idmap, int k, float *d_D, int64_t *d_I,
                              const int64_t *d_fin_rowids, const int64_t *d_fin_idmap, int *d_flag, unsigned long long *d_stats,
                              int *d_qfail, int *d_fail_cnt, int *d_fail_q, bool reset, hipStream_t st, const IvfFlatArith *fa = nullptr,
                              int64_t label_offset = 0, const unsigned *d_brow = nullptr, int rows_interleaved = 0,
                              const unsigned long long *d_units = nullptr, const unsigned *d_unit_cnt = nullptr,
                              const int *d_kept_blk = nullptr, int nkept_blk = 0, unsigned long long *d_kept_out = nullptr,
                              const IpFlatEmit *ipf = nullptr);
size_t ivf_bucket_units_bytes(int64_t cap_entries);
unsigned ivf_bucket_scatter_blocks(int64_t cap_entries);
// final bound + the survivors into their queries' row buckets (csrc/ivf_collect.hip); launch_ivf_bucket_finish(..., d_brow) re-scores them
void launch_ivf_bucket_scatter(const unsigned long long *d_strm, const float *d_su, int64_t cap, const unsigned long long *d_cnt,
                               const unsigned *d_gslot, int nclass, int kf, int64_t nq, float *d_bf, unsigned *d_brow, unsigned *d_bcount,
                               int bpitch, int *d_kept_blk, unsigned long long *d_units, unsigned *d_unit_cnt, hipStream_t st);
// probed lists that provably hold none of a query's k nearest rows -> -1 in d_out (csrc/ivf_collect.hip ivf_probe_prune_kernel); np <= 256
void launch_ivf_probe_prune(const float *d_x, int64_t nq, int d, const float *d_cD, const int64_t *d_cI, int np, int k, const float *d_cn,
                            const unsigned *d_list_max, const int64_t *d_list_off, int64_t *d_out, int *d_kept, hipStream_t st);
void launch_ivf_shadow_verify(const float *d_cmat, const float *d_cD, const int64_t *d_cI, int64_t nq, int nlist, int np, int d, int k,
                              const float *d_qn, const float *d_cn, const unsigned *d_list_max, const int64_t *d_lb, const int64_t *d_le,
                              const float *d_D, const int64_t *d_I, const unsigned *d_ymax_bits, int *d_fail_cnt, int *d_fail_q,
                              hipStream_t st);
size_t ivf_rowmask_bytes(int64_t nrows_mf);
void launch_ivf_rowmask(SelectorDev sel, const int64_t *d_rowids_mf, const int *d_perm, const int64_t *d_idmap, int64_t nrows_mf,
                        void *d_mask, hipStream_t st);
void launch_collect_flat_items(void *d_items, int *d_nitems, int *d_qidx, int64_t nq, int64_t n, hipStream_t st);
DirectPlan plan_flat_direct_extra(const FlatGeom &g, int64_t nq, int64_t n, int64_t k);
void launch_flat_direct_extra(const FlatGeom &g, const DirectPlan &p, int metric, float metric_arg, int d,
                              const float *d_xq, int64_t nq, FlatDB db, int64_t k, SelectorDev sel,
                              const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, hipStream_t st);
void launch_flat_direct_ex(const FlatGeom &g, const DirectPlan &p, int metric, bool formula, const float *d_xq,
                           const float *d_xn, int64_t nq, FlatDB db, int64_t k, SelectorDev sel,
                           const int64_t *d_idmap, float *d_pd, int32_t *d_pi, unsigned *d_gslot, hipStream_t st);

} // namespace mvs
