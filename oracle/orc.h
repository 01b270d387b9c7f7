/*
 * oracle/orc.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the FAISS behaviour that the reference DuckDB
 * extension depends on for its hot path (IndexFlat / IndexIDMap /
 * IndexIVFFlat add + train + search, IDSelectorBitmap / IDSelectorBatch).
 *
 * FAISS itself is an UN-VENDORED git submodule of the reference
 * (/root/reference/.gitmodules:5-7, /root/reference/faiss/ is empty; version
 * only bounded to >= 1.11 by /root/reference/faiss.patch), so this file
 * restates FAISS's published algorithm and anchors parity on the reference's
 * own call sites and golden vectors:
 *     src/faiss_extension.cpp:154  index_factory
 *     src/faiss_extension.cpp:396,583  Index::train
 *     src/faiss_extension.cpp:510,607  Index::add_with_ids
 *     src/faiss_extension.cpp:512,609  Index::add
 *     src/faiss_extension.cpp:631  Index::search
 *     test/sql/faiss.test:16-38, faiss2.test:17-42, faiss3.test:22-68 (goldens)
 *
 * PARITY PIN: the oracle is checked against every golden vector the
 * reference's tests hold for this path (d=8, N=1000, nq=10, k=2, inner
 * product -> per-pair path only).  The BLAS-formula path (nq >= 20), the L2
 * metric and IVF have NO golden values anywhere in the reference: for those
 * the oracle is "parity unpinned by the reference" and is pinned only by its
 * own naive-vs-packed cross checks (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libmi355faiss.so) never links or calls it.
 *
 * Canonical arithmetic (shared, bit for bit, with the HIP kernels):
 *   inner product   ip  = fmaf(x[d-1],y[d-1], ... fmaf(x[1],y[1], fmaf(x[0],y[0], 0)))
 *   squared norm    nrm = same chain with y = x
 *   L2, BLAS path   dis = (nrm_x + nrm_y) - 2*ip ; if (dis < 0) dis = 0
 *                   [FAISS utils/distances.cpp exhaustive_L2sqr_blas; sgemm's
 *                    summation order is implementation-defined, the k-ordered
 *                    fma chain is what v_mfma_f32_32x32x2_f32 computes]
 *   L2, pair path   dis = chain of fmaf(t,t,acc), t = x[k]-y[k]   (fvec_L2sqr)
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_METRIC_INNER_PRODUCT 0 /* faiss::METRIC_INNER_PRODUCT */
#define ORC_METRIC_L2 1            /* faiss::METRIC_L2 */

#define ORC_SEL_NONE 0
#define ORC_SEL_BITMAP 1 /* faiss::IDSelectorBitmap(n_bytes, bitmap)   src/faiss_extension.cpp:959 */
#define ORC_SEL_BATCH 2  /* faiss::IDSelectorBatch(n, ids)             src/faiss_extension.cpp:1008 */

#define ORC_PATH_AUTO 0 /* FAISS dispatch: sel || nq < 20 -> pair, else blas */
#define ORC_PATH_PAIR 1
#define ORC_PATH_BLAS 2
#define ORC_PATH_OPENBLAS 3 /* the BLAS branch on the real OpenBLAS sgemm (orc_openblas_load first); FAISS's 4096 x 1024 blocking */

typedef struct orc_index orc_index;

typedef struct {
	int64_t nprobe;   /* SearchParametersIVF::nprobe, default 1 (src/faiss_extension.cpp:683-686) */
	int64_t efSearch; /* SearchParametersHNSW::efSearch, default 16 (:696-699) */
	int sel_kind;     /* ORC_SEL_* */
	const void *sel_data;
	int64_t sel_n; /* bitmap: number of BYTES; batch: number of ids */
	int force_path; /* ORC_PATH_*; test hook */
} orc_params;

const char *orc_last_error(void);

/* faiss::index_factory(d, desc, metric)  -- src/faiss_extension.cpp:154-155 */
orc_index *orc_index_factory(int d, const char *desc, int metric);
void orc_index_free(orc_index *ix);
int orc_d(const orc_index *ix);
int64_t orc_ntotal(const orc_index *ix);
int orc_is_trained(const orc_index *ix);
int orc_metric(const orc_index *ix);
/* 0 on success, nonzero + orc_last_error() otherwise (the text carries FAISS's message) */
int orc_train(orc_index *ix, int64_t n, const float *x);
int orc_add(orc_index *ix, int64_t n, const float *x);
int orc_add_with_ids(orc_index *ix, int64_t n, const float *x, const int64_t *ids);
int orc_search(const orc_index *ix, int64_t nq, const float *x, int64_t k, float *D, int64_t *I,
               const orc_params *params);
/* IVF introspection for parity tests (share centroids / lists with the GPU index) */
int64_t orc_ivf_nlist(const orc_index *ix);
int orc_ivf_get_centroids(const orc_index *ix, float *out); /* nlist*d */
int orc_ivf_set_centroids(orc_index *ix, const float *c);   /* marks trained */
int64_t orc_ivf_list_size(const orc_index *ix, int64_t list_no);
int orc_ivf_get_list(const orc_index *ix, int64_t list_no, int64_t *ids, float *codes);

/* HNSW (orc_hnsw.c): efConstruction setter (src/faiss_extension.cpp:136-139) and graph export for parity tests */
int orc_hnsw_set_ef_construction_ix(orc_index *ix, int v);
/* tie rule of MinimaxHeap::pop_min in the search walk: 0 = FAISS's array order (default), 1 = smallest id (the device walk's) */
void orc_hnsw_set_pop_min_rule(int rule);
int orc_hnsw_get_pop_min_rule(void);
int64_t orc_hnsw_graph_size(orc_index *ix, int *max_level, int32_t *entry_point); /* total neighbor slots, -1 if not HNSW */
int orc_hnsw_get_graph(orc_index *ix, int *levels, int64_t *offsets, int32_t *neighbors);
int orc_hnsw_set_graph(orc_index *ix, int64_t n, const float *x, const int *levels, const int64_t *offsets,
                       const int32_t *neighbors, int32_t entry_point, int max_level); /* search-only afterwards */

/* stand-alone kernels (used by tests and the cpu_baseline leg) */
/* Index::metric_arg of the Lp metric (FAISS default 0; the glue never sets it) -- process-wide in the oracle */
void orc_set_metric_arg(float v);
void orc_norms(const float *x, int64_t n, int d, float *out);
int orc_flat_search(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k,
                    float *D, int64_t *I, const orc_params *params, const int64_t *id_map);
/* naive triple-loop versions of the two paths (cross-check of the packed AVX2 path) */
int orc_flat_search_naive(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k,
                          float *D, int64_t *I, int path);
/* k-way merge of per-shard results with the FAISS ordering rule (multi-GPU merge oracle) */
void orc_merge_shards(int metric, int64_t nq, int64_t k, int nshard, const float *D, const int64_t *I, float *Dout,
                      int64_t *Iout);
/* counter-based synthetic generator shared with the device (splitmix64 -> 24-bit uniform [0,1)) */
void orc_synth_uniform(float *out, int64_t n_rows, int d, uint64_t seed, int64_t row0);
void orc_synth_clustered(float *out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers, float sigma);
/* dlopen an OpenBLAS with the 64-bit-integer scipy symbol prefix (numpy.libs/libscipy_openblas64_*.so, version 0.3.29 = the
 * reference's vcpkg pin) for ORC_PATH_OPENBLAS; 0 on success */
int orc_openblas_load(const char *path);
const char *orc_openblas_config(void);
void orc_openblas_set_num_threads(int n);
/* test hook: 0 = heaps at every k (FAISS switches to ReservoirTopN at k >= 100; for L2 the two must give the same result) */
void orc_set_reservoir(int on);
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
