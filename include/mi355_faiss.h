/*
 * include/mi355_faiss.h -- C ABI of libmi355faiss.so, the MI355X-native replacement for the FAISS
 * calls the reference DuckDB extension makes on its vector-search hot path.
 *
 * Every entry point replaces ONE FAISS C++ symbol that /root/reference/src/faiss_extension.cpp (or
 * src/gpu/gpu.cpp) reaches; the citation after each declaration is that call site.  Plain pointers
 * and sizes only: the faiss::-namespaced C++ adaptor (duckdb-faiss-ext_amd/compat/faiss/...) and the
 * Python ctypes host (duckdb-faiss-ext_amd/pyhost/mi355_faiss.py) are both thin layers over this file,
 * and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - return 0 on success; nonzero = failure, text in mvs_last_error() (thread-local).  The text
 *     carries FAISS's exception message because the reference pattern-matches substrings of it:
 *       "should be at least as large as number of clusters"      src/faiss_extension.cpp:400,592
 *       "add_with_ids not implemented for this type of index"    src/faiss_extension.cpp:523
 *       "This index type is not implemented"                     src/gpu/gpu.cpp:52
 *       "Invalid GPU device"                                     src/gpu/gpu.cpp:56
 *   - x / ids / D / I are HOST pointers owned by the caller and valid only for the duration of the
 *     call (DuckDB vector buffers, new[] arrays: src/faiss_extension.cpp:626-627); the library
 *     copies through pinned staging.  The *_device variants take device pointers instead.
 *   - an index may be called from a different OS thread each time (DuckDB workers); calls on one
 *     index are serialised internally as the reference's faiss_lock does (:394,:506,:581,:629).
 *   - the product path has NO CPU fallback: every call fails loudly if no gfx950 device is usable.
 */
#ifndef MI355_FAISS_H
#define MI355_FAISS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* faiss::MetricType enumerators the glue exposes (src/faiss_extension.cpp:58-68) */
#define MVS_METRIC_INNER_PRODUCT 0
#define MVS_METRIC_L2 1
#define MVS_METRIC_L1 2
#define MVS_METRIC_Linf 3
#define MVS_METRIC_Lp 4
#define MVS_METRIC_Canberra 20
#define MVS_METRIC_BrayCurtis 21
#define MVS_METRIC_JensenShannon 22
#define MVS_METRIC_Jaccard 23

/* dynamic_cast targets of the glue (src/faiss_extension.cpp:127,133,671,675,691,704; gpu.cpp:69) */
#define MVS_KIND_FLAT 1    /* faiss::IndexFlat / IndexFlatL2 / IndexFlatIP */
#define MVS_KIND_IDMAP 2   /* faiss::IndexIDMap                            */
#define MVS_KIND_IVFFLAT 3 /* faiss::IndexIVFFlat (an IndexIVF)            */
#define MVS_KIND_HNSW 4    /* faiss::IndexHNSWFlat (an IndexHNSW)          */

#define MVS_SEL_NONE 0
#define MVS_SEL_BITMAP 1 /* faiss::IDSelectorBitmap(n_bytes, bitmap)  src/faiss_extension.cpp:959  */
#define MVS_SEL_BATCH 2  /* faiss::IDSelectorBatch(n, ids)            src/faiss_extension.cpp:1008 */

typedef struct mvs_index mvs_index;

/* faiss::SearchParameters / SearchParametersIVF / SearchParametersHNSW as built by
 * innerCreateSearchParameters (src/faiss_extension.cpp:668-721).  Zero = FAISS default. */
typedef struct mvs_search_params {
	int64_t nprobe;       /* SearchParametersIVF::nprobe   (:683-686), default 1  */
	int64_t efSearch;     /* SearchParametersHNSW::efSearch (:696-699), default 16; for IVF<n>_HNSW<m> the efSearch of
	                         the coarse quantizer's SearchParametersHNSW (quantizer_params, :679-681) */
	int32_t sel_kind;     /* MVS_SEL_*; SearchParameters::sel (:678,:694,:719)     */
	int32_t reserved;
	const void *sel_data; /* bitmap bytes | int64 ids; HOST memory owned by the caller */
	int64_t sel_n;        /* bitmap: bytes; batch: number of ids */
} mvs_search_params;

/* faiss::FaissException::msg / what()  -- src/faiss_extension.cpp:397,514,584,632 */
const char *mvs_last_error(void);

/* faiss::index_factory(d, description, metric)  -- src/faiss_extension.cpp:154-155.
 * The index is created device-native on the device named by env MVS_DEVICE (default 0). */
int mvs_index_factory(mvs_index **out, int d, const char *description, int metric);
/* ~Index (unique_ptr<faiss::Index> dropped by ObjectCache)  -- src/faiss_extension.cpp:264 */
void mvs_index_free(mvs_index *ix);

/* Index::d / ntotal / is_trained / metric_type  -- src/faiss_extension.cpp:159,355,490,518 */
int mvs_index_d(const mvs_index *ix);
int64_t mvs_index_ntotal(const mvs_index *ix);
int mvs_index_is_trained(const mvs_index *ix);
int mvs_index_metric_type(const mvs_index *ix);
/* the glue's dynamic_cast to IndexIDMap / IndexIVF / IndexHNSW -- returns MVS_KIND_* */
int mvs_index_kind(const mvs_index *ix);
/* IndexIDMap::index (:129,:673) ; IndexIVF::quantizer (:680).  Borrowed pointers, NULL if n/a. */
mvs_index *mvs_index_idmap_sub(mvs_index *ix);
mvs_index *mvs_index_ivf_quantizer(mvs_index *ix);
/* IVF introspection (IndexIVF::nlist, quantizer centroids): lets parity tests share centroids with the oracle */
int64_t mvs_index_ivf_nlist(const mvs_index *ix);
int mvs_index_ivf_get_centroids(mvs_index *ix, float *out /* nlist*d */);
int mvs_index_ivf_set_centroids(mvs_index *ix, const float *centroids /* nlist*d; marks trained */);
/* IndexHNSW::hnsw.efConstruction = v  -- src/faiss_extension.cpp:136-139 */
int mvs_index_hnsw_set_ef_construction(mvs_index *ix, int v);
/* the value the next add will build with (IDMap wrappers are looked through); -1 if the index is not HNSW */
int mvs_index_hnsw_get_ef_construction(mvs_index *ix);
/* HNSW introspection (HNSW::levels / offsets / neighbors, FAISS's flat layout: 2M slots at level 0, M above, -1 =
 * empty): lets parity tests compare the device-built graph with the oracle's.  graph_info returns the number of
 * neighbour slots (offsets[ntotal]) or -1 if the index is not an HNSW index. */
int64_t mvs_index_hnsw_graph_info(mvs_index *ix, int *max_level, int *entry_point);
/* measurement only (bench.py): counters of the last search run with kernel timing on -- distance evaluations (what FAISS's walk
   evaluates: the algorithmic unit of SURVEY 8d), f32 rows actually fetched, bf16 rows looked at first (csrc/hnsw.hip, "bf16 first look") */
int mvs_index_hnsw_walk_stats(mvs_index *ix, double *evaluations, double *f32_rows, double *bf16_rows);
int mvs_index_hnsw_get_graph(mvs_index *ix, int32_t *levels /* ntotal */, int64_t *offsets /* ntotal+1 */,
                             int32_t *neighbors /* offsets[ntotal] */);

/* Index::train(n, x)  -- src/faiss_extension.cpp:396,583 */
int mvs_index_train(mvs_index *ix, int64_t n, const float *x);
/* Index::add(n, x)  -- src/faiss_extension.cpp:512,609 */
int mvs_index_add(mvs_index *ix, int64_t n, const float *x);
/* Index::add_with_ids(n, x, ids)  -- src/faiss_extension.cpp:510,607 */
int mvs_index_add_with_ids(mvs_index *ix, int64_t n, const float *x, const int64_t *ids);
/* Index::search(n, x, k, distances, labels, params)  -- src/faiss_extension.cpp:631 */
int mvs_index_search(mvs_index *ix, int64_t n, const float *x, int64_t k, float *distances, int64_t *labels,
                     const mvs_search_params *params);

/* faiss::gpu::index_cpu_to_gpu(resources, device, index)  -- src/gpu/gpu.cpp:48.
 * Indexes are already device-native; this migrates the index to `device` (no-op if it is there). */
int mvs_index_to_gpu(mvs_index *ix, int device);
int mvs_index_device(const mvs_index *ix);
/* the same call with FAISS's ownership: returns a NEW index on `device` holding a copy of `src` (the glue replaces
 * entry.index with the result and drops the old object, src/gpu/gpu.cpp:48) */
int mvs_index_clone_to_gpu(mvs_index **out, const mvs_index *src, int device);

/* ---- several devices behind the same surface -------------------------------------------------
 * The reference's hook names ONE device (MoveToGPUFunction, src/gpu/gpu.cpp:34-63 -> index_cpu_to_gpu(res, device,
 * index) :48).  Three ways reach all GPUs of the node WITHOUT touching src/faiss_extension.cpp:
 *   - faiss_to_gpu(name, -1): mvs_index_clone_to_gpu(out, src, -1) returns the index spread over every device of env
 *     MVS_DEVICES ("0,1,...,7"; default: all visible devices);
 *   - env MVS_DEVICES set when faiss_create / faiss_load run: mvs_index_factory / mvs_read_index build it sharded;
 *   - mvs_index_shard_to_gpus: the same conversion in place, for hosts that hold the handle.
 * Flat / IDMap,Flat / IVF<n>,Flat are ROW-SHARDED (IVF: one set of centroids, trained once, on every device; every
 * inverted list split); HNSW is REPLICATED and the queries are split.  Results are identical to the single-device
 * index bit for bit (labels and distances, including inner-product boundary ties): one exchange of the per-shard
 * (value, global row) blocks -- option "shard_exchange" 0 = per-device D2H, 1 = one ncclAllGather over xGMI -- then the
 * host k-way merge.  mvs_index_shard_info returns the shard count (0 = not sharded). */
int mvs_index_shard_to_gpus(mvs_index *ix, const int *devices, int ndev);
int mvs_index_shard_info(const mvs_index *ix, int *devices, int max_devices, int64_t *rows_per_shard,
                         int64_t *last_tie_queries);

/* faiss::write_index / read_index  -- src/faiss_extension.cpp:199,234 */
int mvs_write_index(const mvs_index *ix, const char *filename);
int mvs_read_index(mvs_index **out, const char *filename);

/* ---- device-resident variants (same semantics, inputs/outputs already in HBM) --------------------
 * Used by bench.py (the metric is quoted with inputs resident in HBM) and by the multi-GPU host,
 * which hands the per-shard (distance,label) blocks to RCCL without a host round trip.
 * `stream` is a hipStream_t used exactly as given (NULL = HIP's null stream, which is PyTorch's default
 * stream); the call only enqueues work and is ordered after/before the index's own host-API stream with
 * events, so host-API and device-API calls may be mixed freely. */
int mvs_index_add_device(mvs_index *ix, int64_t n, const float *d_x, const int64_t *d_ids, void *stream);
int mvs_index_search_device(mvs_index *ix, int64_t n, const float *d_x, int64_t k, float *d_distances,
                            int64_t *d_labels, const mvs_search_params *params, void *stream);
/* label offset added to implicit (non-IDMap) labels: row-sharded multi-GPU search returns GLOBAL ids */
int mvs_index_set_label_offset(mvs_index *ix, int64_t offset);

/* k-way merge of per-shard results [nshard][n][k] (global labels) with the FAISS ordering rule.
 * Host arrays.  This is the "host k-way merge" that follows the RCCL all-gather. */
int mvs_merge_shards(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out,
                     int64_t *I_out);

/* The same merge ON THE DEVICE, straight from the gathered records: d_records = [nshard][n][kk][2] int64 {value bits in the
 * low word, global label} as the all-gather delivers them (pyhost/sharded.py pack_records); keeps the kout <= kk best per
 * query.  raw = 1: the pure order (what step 1 of the tie protocol below needs); raw = 0: FAISS's print order.  With 8
 * GPUs the per-rank search of the headline is ~3 ms and a host merge of 8 x 10k x 10 candidates costs more than that. */
int mvs_merge_records_device(int metric, int64_t n, int kk, int kout, int nshard, const int64_t *d_records, int raw,
                             float *d_D_out, int64_t *d_I_out, void *stream);

/* ---- inner-product boundary ties across PROCESSES (one rank per GPU, pyhost/sharded.py under torchrun) ----------
 * FAISS's CMin heap keeps an arrival-order dependent subset of the rows tied at the k-th score.  A row
 * shard therefore hands over its k+1 best in the PURE order (option "ip_exact_ties" = 0, search with k+1):
 *   1. gather, mvs_merge_shards_raw -> merged top-(k+1), pure order;
 *   2. queries whose k-th and (k+1)-th scores are bit-equal: every rank reports, per such query, its k smallest GLOBAL
 *      rows with score >= T (mvs_index_tie_candidates_device; T = the k-th score), gather, keep the k smallest;
 *   3. mvs_finish_ip_ties writes FAISS's print order for all queries and the heap's outcome for the flagged ones.
 * (A ShardedIndex does the same inside the library.) */
int mvs_merge_shards_raw(int metric, int64_t n, int64_t kk, int nshard, const float *D, const int64_t *I, float *D_out,
                         int64_t *I_out);
int mvs_index_tie_candidates_device(mvs_index *ix, int64_t nf, const float *d_xf, const float *d_T, int64_t k,
                                    int64_t *d_rows_out, const mvs_search_params *params, void *stream);
int mvs_finish_ip_ties(int64_t n, int64_t k, int64_t kk, const float *raw_D, const int64_t *raw_I, int64_t nf,
                       const int64_t *flagged, const int64_t *first_rows, float *D_out, int64_t *I_out);

/* ---- IVF exact distance ties across PROCESSES (round 5; DESIGN.md 3.5 "row-sharded IVF") ----------------------------------
 * FAISS's IVFFlatScanner feeds a heap in ARRIVAL order -- probe rank of the list, then position in the list
 * (IndexIVF::search_preassigned behind /root/reference/src/faiss_extension.cpp:631) -- so which rows tied at the k-th value
 * survive depends on arrival order.  A row shard hands over its k + 1 best in the PURE order (option "ivf_exact_ties" = 0,
 * search with k + 1, stored ids = global rows); for a query whose merged k-th and (k + 1)-th values are bit-equal every
 * rank reports its first k rows NOT WORSE than T in arrival order -- value, stored id, probe rank of the list (-1 padded) --
 * and the merge rank takes the first k of the union by (probe rank, id) = A_k and applies the closed form of csrc/ivf_ties.hip
 * (pyhost/sharded.py merge_ivf_exact; the in-library ShardedIndex does the same in resolve_ties_ivf).
 * d_flag = {nf, query numbers ...} on the device; d_x = the WHOLE batch of the search that has just run on this index (its coarse
 * assignment is reused: the call must follow that search directly -- checked: another d_x, more flagged queries than the batch held or
 * a query number outside it is an error); d_T [nf]; outputs [nf][k].  mvs_index_get_stat "ivf_ids_ascending" = 1 while every id added
 * so far exceeded all before it -- the cross-process merge's arrival order (probe rank, id) is FAISS's only then. */
int mvs_index_ivf_tie_emit_device(mvs_index *ix, int64_t nf, const int *d_flag, const float *d_x, const float *d_T, int64_t k,
                                  float *d_v_out, int64_t *d_id_out, int *d_rank_out, const mvs_search_params *params,
                                  void *stream);

/* ---- synthetic data (counter-based, identical on host oracle and device) and diagnostics -------- */
int mvs_synth_uniform_device(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, void *stream);
int mvs_synth_clustered_device(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers,
                               float sigma, void *stream);
/* name + launch geometry + algorithmic flops/bytes of the dominant kernel of the last search on this
 * index (bench.py's roofline object); returns 0 and fills the fields */
typedef struct mvs_kernel_info {
	char name[64];
	double flops;       /* algorithmic flops of the launch            */
	double bytes;       /* algorithmic HBM bytes of the launch        */
	double last_ms;     /* HIP-event duration of that launch, if timed */
	int32_t grid, block, lds_bytes, nsplit;
} mvs_kernel_info;
/* Diagnostics (no counterpart in the reference): ntiles independent v_mfma_f32_16x16x32_bf16 instructions, D = A B + C, on host
 * buffers -- A [ntiles][16 rows][32 k] and Bt [ntiles][16 columns][32 k] as bf16 bit patterns, C / D [ntiles][16][16] f32.  The
 * coarse filters' error bound models this instruction's internal accumulation; tests/test_mfma_model_gpu.py measures it. */
int mvs_debug_mfma_bf16_16x16x32(const uint16_t *A, const uint16_t *Bt, const float *C, float *D, int64_t ntiles);
int mvs_index_last_kernel_info(const mvs_index *ix, mvs_kernel_info *out);
/* when enabled, the dominant kernel of every search is bracketed by HIP events on its stream */
int mvs_index_set_kernel_timing(mvs_index *ix, int enabled);
/* number of timed launches so far and the sum of their HIP-event durations */
int mvs_index_kernel_time_stats(mvs_index *ix, int *count, double *total_ms);
/* implementation knobs (never needed by the reference glue): "force_direct" = 0/1 */
int mvs_index_set_option(mvs_index *ix, const char *key, int64_t value);
/* bf16x3 prefilter of the Flat BLAS-branch search (csrc/flat_bf16.hip; results are those of the exact f32 kernel): queries
 * it served so far, how many of them could not be proven and were re-run on the exact kernel, the largest observed
 * |approx - exact| / (||x|| ||y||) among re-scored candidates and the bound c(d) the proof uses */
int mvs_index_prefilter_stats(mvs_index *ix, int64_t *queries, int64_t *fallback_queries, float *max_rel_err,
                              float *err_bound);
/* bf16 coarse filter of the same search (csrc/flat_collect.hip; option prefilter = 2): queries it served, candidates it
 * re-scored exactly for them, batches whose candidate stream overflowed (served by the bf16x3 path instead) */
int mvs_index_collect_stats(mvs_index *ix, int64_t *queries, int64_t *candidates, int64_t *overflows);
/* IVF, diagnostics: the (query, probed list) pairs of the index's last coarse-filter search (nq x nprobe) and how many of them were
 * scanned.  IndexIVF::search (faiss/IndexIVF.cpp search_preassigned, reached from src/faiss_extension.cpp:631) scans every probed
 * list; this path leaves out the lists that PROVABLY hold none of a query's k nearest rows (triangle inequality on the coarse
 * distance and the list's radius, option ivf_probe_prune, L2 without an IDSelector) -- labels and distances are unchanged.
 * forced_drains: how often a scan wavefront of the last search had to empty its LDS hit queue in the middle of a tile.
 * admitted: candidates the scan of the last search admitted under its running bounds; mvs_index_collect_stats counts those that
 * also passed the bound the scan ended with and were re-scored exactly (option ivf_cl_refilter). */
int mvs_index_ivf_probe_stats(mvs_index *ix, int64_t *pairs, int64_t *pairs_scanned, int64_t *forced_drains, int64_t *admitted);
/* Flat L2 index, diagnostics of its shadow clustering (rows that cluster are answered through an internal IVF index of the same rows
 * with a per-query exactness proof, csrc/index.hip FlatIndex::shadow_search -- results are those of faiss::IndexFlat::search,
 * src/faiss_extension.cpp:631).  stats[8] = {state (0 not wanted, 1 in use, -1 given up on this data), rows the shadow holds (-1: none),
 * queries answered through it, of those re-run on the Flat kernels (unproven), builds (k-means), extensions (rows appended after
 * add()), device bytes the shadow holds, its nlist}; build_seconds = time spent building / extending it inside search calls. */
int mvs_index_shadow_stats(mvs_index *ix, int64_t *stats, double *build_seconds);
/* diagnostics by name (tests, bench): "coarse_bf16_queries" = queries of an IVF index whose coarse quantisation (IndexIVF::search ->
 * quantizer->search, src/faiss_extension.cpp:631) ran as a bf16 filter + exact re-scoring (csrc/coarse_bf16.hip);
 * "coarse_bf16_exhaustive" = of those, queries that were computed against every centroid (list overflow / no finite bound);
 * "coarse_bf16_candidates" = centroids re-scored exactly in the last such call, summed over its queries; "flat_outlier_rows" = rows of a
 * Flat index kept out of its bf16 coarse-filter store because of their norm (they join every query's candidates: csrc/flat_collect.hip);
 * "hnsw_build_distances" / "hnsw_build_shortcuts" = distance evaluations of an HNSW index's builds so far / add_link calls that took the
 * full-list short cut instead of HNSW::shrink_neighbor_list's pairwise pass (csrc/hnsw.hip; IndexHNSW::add, src/faiss_extension.cpp:510) */
int mvs_index_get_stat(mvs_index *ix, const char *name, int64_t *value);
/* named ranges for rocprofv3 --marker-trace (roctx; bound at run time, only under a profiler or with MVS_ROCTX=1): the library
 * marks its own stages (row staging, Flat / IVF search, shard search, exchange, merge); a host that merges shard results itself
 * (pyhost/sharded.py: the exchange of src/gpu/gpu.cpp:48's multi-GPU layout) brackets its stages with these */
int mvs_trace_push(const char *name);
int mvs_trace_pop(void);
int mvs_device_count(void);
const char *mvs_version(void);

#ifdef __cplusplus
}
#endif
#endif
