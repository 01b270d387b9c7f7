/*
 * oracle/orc_core.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See orc.h.
 *
 * Restates, in plain C, the FAISS algorithms reached from the reference's call
 * sites (src/faiss_extension.cpp:154,396,510,512,583,607,609,631).  FAISS is an
 * un-vendored submodule of the reference (empty /root/reference/faiss), so the
 * "follows" citations below name the UPSTREAM FAISS file whose published
 * behaviour each function restates; nothing here is copied from a source file.
 *
 *   heaps / ordering            faiss/utils/Heap.h, utils/ordered_key_value.h
 *   result handler insert rule  faiss/impl/ResultHandler.h (strict compare)
 *   flat search dispatch        faiss/IndexFlat.cpp, faiss/utils/distances.cpp
 *   id map                      faiss/IndexIDMap.cpp
 *   selectors                   faiss/impl/IDSelector.cpp
 *   k-means                     faiss/Clustering.cpp, faiss/utils/random.cpp
 *   IVF add / search            faiss/IndexIVF.cpp, faiss/IndexIVFFlat.cpp, invlists/InvertedLists.cpp
 *   factory                     faiss/index_factory.cpp
 */
#define _GNU_SOURCE
#include "orc.h"
#include "orc_internal.h"

#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ errors */

static __thread char g_err[512];

const char *orc_last_error(void) {
	return g_err;
}

/* FAISS formats exceptions as "Error in <func> at <file>:<line>: <msg>" (impl/FaissException.cpp);
 * the reference pattern-matches substrings of <msg> (src/faiss_extension.cpp:400,523,592). */
static int fail(const char *func, const char *file, const char *fmt, ...) {
	char msg[384];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(msg, sizeof msg, fmt, ap);
	va_end(ap);
	snprintf(g_err, sizeof g_err, "Error in %s at %s: %s", func, file, msg);
	return 1;
}

int orc_num_threads(void) {
	return omp_get_max_threads();
}
void orc_set_num_threads(int n) {
	if (n > 0)
		omp_set_num_threads(n);
}

/* --------------------------------------------------- mt19937 (std::mt19937) */

typedef orc_mt19937 mt19937_t;

static void mt_seed(mt19937_t *r, uint32_t s) {
	r->mt[0] = s;
	for (int i = 1; i < 624; i++)
		r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
	r->idx = 624;
}
static uint32_t mt_next(mt19937_t *r) {
	if (r->idx >= 624) {
		for (int i = 0; i < 624; i++) {
			uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu);
			uint32_t v = r->mt[(i + 397) % 624] ^ (y >> 1);
			if (y & 1u)
				v ^= 0x9908b0dfu;
			r->mt[i] = v;
		}
		r->idx = 0;
	}
	uint32_t y = r->mt[r->idx++];
	y ^= y >> 11;
	y ^= (y << 7) & 0x9d2c5680u;
	y ^= (y << 15) & 0xefc60000u;
	y ^= y >> 18;
	return y;
}
/* faiss::RandomGenerator::rand_int(max) = mt() % max ; rand_float() = mt() / float(mt.max()) */
static int mt_rand_int(mt19937_t *r, int max) {
	return (int)(mt_next(r) % (uint32_t)max);
}
static float mt_rand_float(mt19937_t *r) {
	return (float)mt_next(r) / (float)4294967295u;
}
/* faiss::rand_perm (utils/random.cpp): Fisher-Yates, i2 = i + rand_int(n - i) */
static void rand_perm(int *perm, size_t n, int64_t seed) {
	for (size_t i = 0; i < n; i++)
		perm[i] = (int)i;
	mt19937_t rng;
	mt_seed(&rng, (uint32_t)seed);
	for (size_t i = 0; i + 1 < n; i++) {
		int i2 = (int)i + mt_rand_int(&rng, (int)(n - i));
		int t = perm[i];
		perm[i] = perm[i2];
		perm[i2] = t;
	}
}

/* ------------------------------------------------------------ k-best heaps */
/* CMax<float,int64> keeps the k SMALLEST (L2); CMin keeps the k LARGEST (IP).
 * "worse(a,ia,b,ib)" is FAISS's C::cmp2(a,b,ia,ib): a sits nearer the root than b. */

static inline int worse(int is_max, float a, int64_t ia, float b, int64_t ib) {
	if (is_max)
		return (a > b) || (a == b && ia > ib);
	return (a < b) || (a == b && ia < ib);
}
static inline float neutral(int is_max) {
	return is_max ? FLT_MAX : -FLT_MAX;
}
/* C::cmp(top, dis): strict; an element EQUAL to the current worst is rejected (ResultHandler.h) */
static inline int accepts(int is_max, float top, float dis) {
	return is_max ? (top > dis) : (top < dis);
}
static void heap_init(int64_t k, float *hv, int64_t *hi, int is_max) {
	for (int64_t j = 0; j < k; j++) {
		hv[j] = neutral(is_max);
		hi[j] = -1;
	}
}
static void heap_sift_from_root(int64_t k, float *hv0, int64_t *hi0, int is_max, float v, int64_t id) {
	float *hv = hv0 - 1;
	int64_t *hi = hi0 - 1;
	int64_t i = 1;
	for (;;) {
		int64_t l = 2 * i, r = l + 1, c;
		if (l > k)
			break;
		c = (r <= k && worse(is_max, hv[r], hi[r], hv[l], hi[l])) ? r : l;
		if (!worse(is_max, hv[c], hi[c], v, id))
			break;
		hv[i] = hv[c];
		hi[i] = hi[c];
		i = c;
	}
	hv[i] = v;
	hi[i] = id;
}
static inline void heap_replace_top(int64_t k, float *hv, int64_t *hi, int is_max, float v, int64_t id) {
	heap_sift_from_root(k, hv, hi, is_max, v, id);
}
/* heap_reorder: pop the root to the back repeatedly; entries whose id is -1 are
 * dropped and re-filled with (neutral, -1) at the end (Heap.h heap_reorder). */
static void heap_reorder(int64_t k, float *hv, int64_t *hi, int is_max) {
	int64_t ii = 0;
	for (int64_t i = 0; i < k; i++) {
		float v = hv[0];
		int64_t id = hi[0];
		int64_t n = k - i; /* live heap size */
		/* pop: move last into the root and sift */
		if (n > 1)
			heap_sift_from_root(n - 1, hv, hi, is_max, hv[n - 1], hi[n - 1]);
		hv[k - ii - 1] = v;
		hi[k - ii - 1] = id;
		if (id != -1)
			ii++;
	}
	memmove(hv, hv + k - ii, (size_t)ii * sizeof *hv);
	memmove(hi, hi + k - ii, (size_t)ii * sizeof *hi);
	for (; ii < k; ii++) {
		hv[ii] = neutral(is_max);
		hi[ii] = -1;
	}
}


/* ------------------------------------------------- ReservoirTopN (k >= 100) */
/* faiss/impl/ResultHandler.h ReservoirTopN + faiss/utils/partitioning.cpp partition_fuzzy_median3, restated: from
 * k = distance_compute_min_k_reservoir = 100 on, knn_L2sqr / knn_inner_product (utils/distances.cpp, both the BLAS and the
 * per-pair branch) collect results in a reservoir of capacity (2k + 15) & ~15 instead of a heap:
 *   add(val, id):   if (C::cmp(threshold, val)) { if (i == capacity) shrink_fuzzy(); vals[i] = val; ids[i] = id; i++; }
 *   shrink_fuzzy(): threshold = partition_fuzzy(vals, ids, capacity, n, (capacity + n) / 2, &i) -- keeps between n and
 *                   (capacity + n) / 2 of the best entries IN ARRAY ORDER (equal values at the threshold: the first ones)
 *   to_result():    the first n entries are pushed on a heap, the others pass the strict heap rule, heap_reorder prints.
 * L2 (rows arrive in ascending id): the retained set is the k smallest (value, id) -- the heap's result, bit for bit
 * (tests/test_oracle_semantics.py).  Inner product: which rows TIED at the k-th score survive depends on where the sampled
 * thresholds fell, so the whole history is replayed here; the device does the same for the queries it flags. */
#define ORC_MIN_K_RESERVOIR 100
typedef struct {
	int is_max;
	int64_t n, cap, i;
	float thr;
	float *vals;
	int64_t *ids;
} reservoir_t;

static inline int rcmp(int is_max, float a, float b) { /* C::cmp(a, b) */
	return is_max ? a > b : a < b;
}
static float median3f(float a, float b, float c) {
	if (a > b) {
		float t = a;
		a = b;
		b = t;
	}
	if (c > b)
		return b;
	if (c > a)
		return c;
	return a;
}
static float sample_threshold_median3(int is_max, const float *vals, int64_t n, float thresh_inf, float thresh_sup) {
	const uint64_t big_prime = 6700417;
	float val3[3];
	int vi = 0;
	for (uint64_t i = 0; i < (uint64_t)n; i++) {
		const float v = vals[(i * big_prime) % (uint64_t)n];
		if (rcmp(is_max, v, thresh_inf) && rcmp(is_max, thresh_sup, v)) { /* thresh_inf < v < thresh_sup for CMax */
			val3[vi++] = v;
			if (vi == 3)
				break;
		}
	}
	if (vi == 3)
		return median3f(val3[0], val3[1], val3[2]);
	if (vi != 0)
		return val3[0];
	return thresh_inf;
}
/* partition_fuzzy_median3: returns the threshold, *q_out = entries kept (q_min <= q <= q_max) */
static float partition_fuzzy(int is_max, float *vals, int64_t *ids, int64_t n, int64_t q_min, int64_t q_max, int64_t *q_out) {
	if (q_min == 0) {
		if (q_out)
			*q_out = 0;
		return neutral(!is_max);
	}
	if (q_max >= n) {
		if (q_out)
			*q_out = q_max;
		return neutral(is_max);
	}
	float thresh_inf = neutral(!is_max); /* C::Crev::neutral() */
	float thresh_sup = neutral(is_max);
	float thresh = median3f(vals[0], vals[n / 2], vals[n - 1]);
	int64_t n_eq = 0, n_lt = 0, q = 0;
	for (int it = 0; it < 200; it++) {
		n_lt = n_eq = 0;
		for (int64_t j = 0; j < n; j++) { /* count_lt_and_eq */
			if (rcmp(is_max, thresh, vals[j]))
				n_lt++;
			else if (vals[j] == thresh)
				n_eq++;
		}
		if (n_lt <= q_min) {
			if (n_lt + n_eq >= q_min) {
				q = q_min;
				break;
			}
			thresh_inf = thresh;
		} else if (n_lt <= q_max) {
			q = n_lt;
			break;
		} else {
			thresh_sup = thresh;
		}
		const float new_thresh = sample_threshold_median3(is_max, vals, n, thresh_inf, thresh_sup);
		if (new_thresh == thresh_inf) /* nothing between thresh_inf and thresh_sup */
			break;
		thresh = new_thresh;
	}
	int64_t n_eq_1 = q - n_lt;
	if (n_eq_1 < 0) { /* more than q entries at the lower bound */
		q = q_min;
		thresh = nextafterf(thresh, is_max ? -INFINITY : INFINITY); /* C::Crev::nextafter */
		n_eq_1 = q;
	}
	int64_t wp = 0; /* compress_array: order preserving; of the entries equal to thresh the first n_eq_1 stay */
	for (int64_t j = 0; j < n; j++) {
		if (rcmp(is_max, thresh, vals[j])) {
			vals[wp] = vals[j];
			ids[wp] = ids[j];
			wp++;
		} else if (n_eq_1 > 0 && vals[j] == thresh) {
			vals[wp] = vals[j];
			ids[wp] = ids[j];
			wp++;
			n_eq_1--;
		}
	}
	if (q_out)
		*q_out = wp;
	return thresh;
}
static void reservoir_begin(reservoir_t *r, int is_max, int64_t k, float *vals, int64_t *ids) {
	r->is_max = is_max;
	r->n = k;
	r->cap = (2 * k + 15) & ~(int64_t)15;
	r->i = 0;
	r->thr = neutral(is_max);
	r->vals = vals;
	r->ids = ids;
}
static inline void reservoir_add(reservoir_t *r, float val, int64_t id) { /* the caller has checked C::cmp(threshold, val) */
	if (r->i == r->cap)
		r->thr = partition_fuzzy(r->is_max, r->vals, r->ids, r->cap, r->n, (r->cap + r->n) / 2, &r->i);
	r->vals[r->i] = val;
	r->ids[r->i] = id;
	r->i++;
}
static void reservoir_to_result(const reservoir_t *r, float *hv, int64_t *hi) {
	const int64_t n = r->n, m = r->i < n ? r->i : n;
	heap_init(n, hv, hi, r->is_max);
	/* heap_push of the first min(i, n) entries == strict inserts into the neutral-filled heap (values equal to the neutral
	 * element aside, which FAISS's own push would keep: they print as (neutral, id) there and as (neutral, -1) here) */
	for (int64_t j = 0; j < m; j++)
		if (accepts(r->is_max, hv[0], r->vals[j]))
			heap_replace_top(n, hv, hi, r->is_max, r->vals[j], r->ids[j]);
	for (int64_t j = n; j < r->i; j++) /* heap_addn */
		if (accepts(r->is_max, hv[0], r->vals[j]))
			heap_replace_top(n, hv, hi, r->is_max, r->vals[j], r->ids[j]);
	heap_reorder(n, hv, hi, r->is_max);
}
static int g_reservoir = 1; /* orc_set_reservoir(0): heaps at every k (test hook: the two must agree for L2) */
void orc_set_reservoir(int on) {
	g_reservoir = on;
}
static inline int use_reservoir(int64_t k) {
	return g_reservoir && k >= ORC_MIN_K_RESERVOIR;
}

/* ------------------------------------------------------- scalar primitives */

static inline float ip_chain(const float *x, const float *y, int d) {
	float acc = 0.f;
	for (int k = 0; k < d; k++)
		acc = fmaf(x[k], y[k], acc);
	return acc;
}
static inline float l2_chain(const float *x, const float *y, int d) {
	float acc = 0.f;
	for (int k = 0; k < d; k++) {
		float t = x[k] - y[k];
		acc = fmaf(t, t, acc);
	}
	return acc;
}
void orc_norms(const float *x, int64_t n, int d, float *out) {
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; i++)
		out[i] = ip_chain(x + i * d, x + i * d, d);
}

/* ---------------------------------------------------------------- selectors */

typedef orc_sel sel_t;

static int cmp_i64(const void *a, const void *b) {
	int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
	return (x > y) - (x < y);
}
static int sel_build(sel_t *s, const orc_params *p) {
	memset(s, 0, sizeof *s);
	if (!p || p->sel_kind == ORC_SEL_NONE)
		return 0;
	s->kind = p->sel_kind;
	s->n = p->sel_n;
	if (p->sel_kind == ORC_SEL_BITMAP) {
		s->bitmap = (const uint8_t *)p->sel_data;
	} else if (p->sel_kind == ORC_SEL_BATCH) {
		/* IDSelectorBatch = bloom filter + unordered_set: pure set membership */
		s->sorted = (int64_t *)malloc((size_t)(p->sel_n > 0 ? p->sel_n : 1) * sizeof(int64_t));
		memcpy(s->sorted, p->sel_data, (size_t)p->sel_n * sizeof(int64_t));
		qsort(s->sorted, (size_t)p->sel_n, sizeof(int64_t), cmp_i64);
	} else {
		return fail("orc_search", "IDSelector", "unknown selector kind %d", p->sel_kind);
	}
	return 0;
}
static void sel_free(sel_t *s) {
	free(s->sorted);
}
static inline int sel_member(const sel_t *s, int64_t id) {
	if (s->kind == ORC_SEL_BITMAP) {
		/* IDSelectorBitmap::is_member: (id>>3) < n && (bitmap[id>>3] >> (id&7)) & 1 */
		uint64_t i = (uint64_t)id;
		if ((i >> 3) >= (uint64_t)s->n)
			return 0;
		return (s->bitmap[i >> 3] >> (i & 7)) & 1;
	}
	if (s->kind == ORC_SEL_BATCH) {
		int64_t lo = 0, hi = s->n;
		while (lo < hi) {
			int64_t mid = (lo + hi) >> 1;
			if (s->sorted[mid] < id)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo < s->n && s->sorted[lo] == id;
	}
	return 1;
}

/* exported to orc_hnsw.c */
void orc_mt_seed(orc_mt19937 *r, uint32_t s) {
	mt_seed(r, s);
}
int orc_mt_rand_int(orc_mt19937 *r, int max) {
	return mt_rand_int(r, max);
}
float orc_mt_rand_float(orc_mt19937 *r) {
	return mt_rand_float(r);
}
void orc_heap_init(int64_t k, float *hv, int64_t *hi, int is_max) {
	heap_init(k, hv, hi, is_max);
}
void orc_heap_replace_top(int64_t k, float *hv, int64_t *hi, int is_max, float v, int64_t id) {
	heap_replace_top(k, hv, hi, is_max, v, id);
}
void orc_heap_reorder(int64_t k, float *hv, int64_t *hi, int is_max) {
	heap_reorder(k, hv, hi, is_max);
}
int orc_sel_member(const orc_sel *s, int64_t id) {
	return sel_member(s, id);
}

/* ---- the other metrics the glue registers (src/faiss_extension.cpp:58-68): faiss/utils/extra_distances-inl.h
 * VectorDistance<mt>::operator() -- plain sequential float loops (reference implementations of fvec_L1 / fvec_Linf;
 * FAISS's SIMD builds may reassociate L1) -- and knn_extra_metrics: one heap per query, strict insert, rows ascending;
 * CMax (smallest kept) except for the similarity metric Jaccard (is_similarity_metric: CMin, like inner product). */
static inline int metric_is_similarity(int metric) {
	return metric == ORC_METRIC_INNER_PRODUCT || metric == 23 /* Jaccard */;
}
static float extra_distance(int metric, float metric_arg, const float *x, const float *y, int d) {
	switch (metric) {
	case 2: { /* L1 */
		float res = 0;
		for (int i = 0; i < d; i++) {
			const float tmp = x[i] - y[i];
			res += fabsf(tmp);
		}
		return res;
	}
	case 3: { /* Linf */
		float res = 0;
		for (int i = 0; i < d; i++)
			res = fmaxf(res, fabsf(x[i] - y[i]));
		return res;
	}
	case 4: { /* Lp */
		float accu = 0;
		for (int i = 0; i < d; i++) {
			const float diff = fabsf(x[i] - y[i]);
			accu += powf(diff, metric_arg);
		}
		return accu;
	}
	case 20: { /* Canberra */
		float accu = 0;
		for (int i = 0; i < d; i++) {
			const float xi = x[i], yi = y[i];
			accu += fabsf(xi - yi) / (fabsf(xi) + fabsf(yi));
		}
		return accu;
	}
	case 21: { /* BrayCurtis */
		float num = 0, den = 0;
		for (int i = 0; i < d; i++) {
			const float xi = x[i], yi = y[i];
			num += fabsf(xi - yi);
			den += fabsf(xi + yi);
		}
		return num / den;
	}
	case 22: { /* JensenShannon */
		float accu = 0;
		for (int i = 0; i < d; i++) {
			const float xi = x[i], yi = y[i];
			const float mi = 0.5f * (xi + yi);
			const float kl1 = -xi * logf(mi / xi);
			const float kl2 = -yi * logf(mi / yi);
			accu += kl1 + kl2;
		}
		return 0.5f * accu;
	}
	case 23: { /* Jaccard (defined for non-negative vectors) */
		float num = 0, den = 0;
		for (int i = 0; i < d; i++) {
			num += fminf(x[i], y[i]);
			den += fmaxf(x[i], y[i]);
		}
		return num / den;
	}
	}
	return NAN;
}
static float g_metric_arg = 0.f; /* Index::metric_arg: the glue never sets it (FAISS default 0) */
void orc_set_metric_arg(float v) {
	g_metric_arg = v;
}

/* ------------------------------------------------------------- flat search */
/* per-pair path: utils/distances.cpp exhaustive_{L2sqr,inner_product}_seq */
static void search_pair(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k, float *D,
                        int64_t *I, const sel_t *sel, const int64_t *id_map) {
	const int is_max = !metric_is_similarity(metric);
	const float marg = g_metric_arg;
#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t i = 0; i < nq; i++) {
		const float *x = xq + i * d;
		float *hv = D + i * k;
		int64_t *hi = I + i * k;
		/* knn_L2sqr / knn_inner_product / knn_extra_metrics: heap below k = 100, reservoir from there on */
		const int resv = use_reservoir(k);
		reservoir_t rs;
		float *rv = NULL;
		int64_t *ri = NULL;
		if (resv) {
			const int64_t cap = (2 * k + 15) & ~(int64_t)15;
			rv = (float *)malloc((size_t)cap * sizeof(float));
			ri = (int64_t *)malloc((size_t)cap * sizeof(int64_t));
			reservoir_begin(&rs, is_max, k, rv, ri);
		} else {
			heap_init(k, hv, hi, is_max);
		}
		for (int64_t j = 0; j < nb; j++) {
			if (sel && sel->kind && !sel_member(sel, id_map ? id_map[j] : j))
				continue;
			float dis = metric == ORC_METRIC_L2              ? l2_chain(x, xb + j * d, d)
			            : metric == ORC_METRIC_INNER_PRODUCT ? ip_chain(x, xb + j * d, d)
			                                                 : extra_distance(metric, marg, x, xb + j * d, d);
			if (resv) {
				if (accepts(is_max, rs.thr, dis))
					reservoir_add(&rs, dis, j);
			} else if (accepts(is_max, hv[0], dis))
				heap_replace_top(k, hv, hi, is_max, dis, j);
		}
		if (resv) {
			reservoir_to_result(&rs, hv, hi);
			free(rv);
			free(ri);
		} else {
			heap_reorder(k, hv, hi, is_max);
		}
	}
}

/* BLAS path: utils/distances.cpp exhaustive_{L2sqr,inner_product}_blas.
 * FAISS: ip = sgemm over (4096-query x 1024-row) blocks; dis = xn + yn - 2 ip, clamped at 0; every query's
 * heap sees the rows in ascending order (HeapBlockResultHandler::add_results).
 * Here "sgemm" is an AVX2 micro-kernel whose per-element arithmetic is the k-ordered fma chain (bitwise equal
 * to ip_chain): the vector lanes run across QUERIES (queries packed k-major once), database rows are streamed
 * row-major and broadcast, so no per-block packing or barrier is needed; one thread owns a group of queries
 * and walks the database front to back, which keeps FAISS's arrival order per query exactly. */
#define BLAS_DBS 1024 /* rows per ip buffer refill (distance_compute_blas_database_bs) */
#define MRW 6         /* database rows per micro-kernel call */

/* ip[r][q] for 16 queries (xt: [d][16], k-major) x MRW rows (row-major, stride ldy) */
static inline void mk_q16(int d, const float *xt, const float *y0, int64_t ldy, float *out /* [MRW][16] */) {
	__m256 a00 = _mm256_setzero_ps(), a01 = a00, a10 = a00, a11 = a00, a20 = a00, a21 = a00;
	__m256 a30 = a00, a31 = a00, a40 = a00, a41 = a00, a50 = a00, a51 = a00;
	const float *y1 = y0 + ldy, *y2 = y1 + ldy, *y3 = y2 + ldy, *y4 = y3 + ldy, *y5 = y4 + ldy;
	for (int k = 0; k < d; k++) {
		const __m256 x0 = _mm256_loadu_ps(xt + (size_t)k * 16), x1 = _mm256_loadu_ps(xt + (size_t)k * 16 + 8);
		__m256 b;
		b = _mm256_broadcast_ss(y0 + k);
		a00 = _mm256_fmadd_ps(x0, b, a00);
		a01 = _mm256_fmadd_ps(x1, b, a01);
		b = _mm256_broadcast_ss(y1 + k);
		a10 = _mm256_fmadd_ps(x0, b, a10);
		a11 = _mm256_fmadd_ps(x1, b, a11);
		b = _mm256_broadcast_ss(y2 + k);
		a20 = _mm256_fmadd_ps(x0, b, a20);
		a21 = _mm256_fmadd_ps(x1, b, a21);
		b = _mm256_broadcast_ss(y3 + k);
		a30 = _mm256_fmadd_ps(x0, b, a30);
		a31 = _mm256_fmadd_ps(x1, b, a31);
		b = _mm256_broadcast_ss(y4 + k);
		a40 = _mm256_fmadd_ps(x0, b, a40);
		a41 = _mm256_fmadd_ps(x1, b, a41);
		b = _mm256_broadcast_ss(y5 + k);
		a50 = _mm256_fmadd_ps(x0, b, a50);
		a51 = _mm256_fmadd_ps(x1, b, a51);
	}
	_mm256_storeu_ps(out, a00);
	_mm256_storeu_ps(out + 8, a01);
	_mm256_storeu_ps(out + 16, a10);
	_mm256_storeu_ps(out + 24, a11);
	_mm256_storeu_ps(out + 32, a20);
	_mm256_storeu_ps(out + 40, a21);
	_mm256_storeu_ps(out + 48, a30);
	_mm256_storeu_ps(out + 56, a31);
	_mm256_storeu_ps(out + 64, a40);
	_mm256_storeu_ps(out + 72, a41);
	_mm256_storeu_ps(out + 80, a50);
	_mm256_storeu_ps(out + 88, a51);
}
static inline void mk_q16_1(int d, const float *xt, const float *y0, float *out /* [16] */) {
	__m256 a0 = _mm256_setzero_ps(), a1 = a0;
	for (int k = 0; k < d; k++) {
		const __m256 b = _mm256_broadcast_ss(y0 + k);
		a0 = _mm256_fmadd_ps(_mm256_loadu_ps(xt + (size_t)k * 16), b, a0);
		a1 = _mm256_fmadd_ps(_mm256_loadu_ps(xt + (size_t)k * 16 + 8), b, a1);
	}
	_mm256_storeu_ps(out, a0);
	_mm256_storeu_ps(out + 8, a1);
}

static void search_blas(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k, float *D,
                        int64_t *I) {
	const int is_max = metric == ORC_METRIC_L2;
	float *xn = NULL, *yn = NULL;
	if (is_max) {
		xn = (float *)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(float));
		yn = (float *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(float));
		orc_norms(xq, nq, d, xn);
		orc_norms(xb, nb, d, yn);
	}
	const int64_t ngroups = (nq + 15) / 16;
#pragma omp parallel
	{
		float *xt = (float *)aligned_alloc(64, (size_t)d * 16 * sizeof(float));
		float *ipbuf = (float *)aligned_alloc(64, (size_t)BLAS_DBS * 16 * sizeof(float));
		/* k >= 100: ReservoirBlockResultHandler instead of HeapBlockResultHandler (one reservoir per query of the group) */
		const int resv = use_reservoir(k);
		const int64_t rcap = (2 * k + 15) & ~(int64_t)15;
		reservoir_t rs[16];
		float *rv = resv ? (float *)malloc((size_t)16 * rcap * sizeof(float)) : NULL;
		int64_t *ri = resv ? (int64_t *)malloc((size_t)16 * rcap * sizeof(int64_t)) : NULL;
#pragma omp for schedule(dynamic, 1)
		for (int64_t g = 0; g < ngroups; g++) {
			const int64_t q0 = g * 16;
			const int nqg = (int)(nq - q0 < 16 ? nq - q0 : 16);
			for (int kk = 0; kk < d; kk++)
				for (int q = 0; q < 16; q++)
					xt[(size_t)kk * 16 + q] = q < nqg ? xq[(q0 + q) * d + kk] : 0.f;
			for (int q = 0; q < nqg; q++) {
				if (resv)
					reservoir_begin(&rs[q], is_max, k, rv + (size_t)q * rcap, ri + (size_t)q * rcap);
				else
					heap_init(k, D + (q0 + q) * k, I + (q0 + q) * k, is_max);
			}
			for (int64_t j0 = 0; j0 < nb; j0 += BLAS_DBS) {
				const int64_t jb = nb - j0 < BLAS_DBS ? nb - j0 : BLAS_DBS;
				int64_t j = 0;
				for (; j + MRW <= jb; j += MRW)
					mk_q16(d, xt, xb + (j0 + j) * d, d, ipbuf + j * 16);
				for (; j < jb; j++)
					mk_q16_1(d, xt, xb + (j0 + j) * d, ipbuf + j * 16);
				/* HeapBlockResultHandler::add_results: per query, rows ascending, strict compare */
				for (int q = 0; q < nqg; q++) {
					float *hv = D + (q0 + q) * k;
					int64_t *hi = I + (q0 + q) * k;
					float thr = resv ? rs[q].thr : hv[0];
					if (is_max) {
						const float xni = xn[q0 + q];
						for (int64_t jj = 0; jj < jb; jj++) {
							float dis = (xni + yn[j0 + jj]) - 2.0f * ipbuf[jj * 16 + q];
							if (dis < 0)
								dis = 0;
							if (thr > dis) {
								if (resv) {
									reservoir_add(&rs[q], dis, j0 + jj);
									thr = rs[q].thr;
								} else {
									heap_replace_top(k, hv, hi, 1, dis, j0 + jj);
									thr = hv[0];
								}
							}
						}
					} else {
						for (int64_t jj = 0; jj < jb; jj++) {
							const float dis = ipbuf[jj * 16 + q];
							if (thr < dis) {
								if (resv) {
									reservoir_add(&rs[q], dis, j0 + jj);
									thr = rs[q].thr;
								} else {
									heap_replace_top(k, hv, hi, 0, dis, j0 + jj);
									thr = hv[0];
								}
							}
						}
					}
				}
			}
			for (int q = 0; q < nqg; q++) {
				if (resv)
					reservoir_to_result(&rs[q], D + (q0 + q) * k, I + (q0 + q) * k);
				else
					heap_reorder(k, D + (q0 + q) * k, I + (q0 + q) * k, is_max);
			}
		}
		free(xt);
		free(ipbuf);
		free(rv);
		free(ri);
	}
	free(xn);
	free(yn);
}

/* BLAS path on the REAL OpenBLAS (ORC_PATH_OPENBLAS): the same branch of utils/distances.cpp
 * (exhaustive_L2sqr_blas / exhaustive_inner_product_blas) with FAISS's own blocking -- 4096 queries
 * (distance_compute_blas_query_bs) x 1024 rows (distance_compute_blas_database_bs) per sgemm -- and the library the
 * reference links (/root/reference/CMakeLists.txt:78-90 find_package(BLAS), vcpkg_ports/openblas/vcpkg.json:3 = 0.3.29;
 * numpy's wheel bundles that very version as libscipy_openblas64_).  sgemm's summation order is OpenBLAS's own, so this is
 * the INDEPENDENT reference for label stability: near-ties at rounding level may rank differently than under the
 * k-ordered chain of search_blas above (tests/ and bench.py count such slots and check each against the rounding band).
 * FAISS calls the Fortran symbol sgemm_("Transpose", "Not transpose", nyi, nxi, d, 1, y, d, x, d, 0, ip_block, nyi);
 * cblas_sgemm(ColMajor, Trans, NoTrans, ...) is the same routine behind OpenBLAS's C front end.
 * Norms: FAISS's fvec_norms_L2sqr is a SIMD loop whose order depends on the build; the k-ordered chain is used here. */
#include <dlfcn.h>
typedef void (*cblas_sgemm64_fn)(int order, int ta, int tb, int64_t m, int64_t n, int64_t k, float alpha, const float *a,
                                 int64_t lda, const float *b, int64_t ldb, float beta, float *c, int64_t ldc);
static cblas_sgemm64_fn g_sgemm = NULL;
static void (*g_blas_set_threads)(int) = NULL;
static char g_blas_config[256] = "";

int orc_openblas_load(const char *path) {
	if (g_sgemm)
		return 0;
	void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
	if (!h)
		return fail("orc_openblas_load", "oracle/orc_core.c", "dlopen(%s): %s", path, dlerror());
	g_sgemm = (cblas_sgemm64_fn)dlsym(h, "scipy_cblas_sgemm64_");
	if (!g_sgemm)
		return fail("orc_openblas_load", "oracle/orc_core.c", "%s has no scipy_cblas_sgemm64_", path);
	g_blas_set_threads = (void (*)(int))dlsym(h, "scipy_openblas_set_num_threads64_");
	const char *(*cfg)(void) = (const char *(*)(void))dlsym(h, "scipy_openblas_get_config64_");
	if (cfg)
		snprintf(g_blas_config, sizeof g_blas_config, "%s", cfg());
	return 0;
}
const char *orc_openblas_config(void) {
	return g_blas_config;
}
void orc_openblas_set_num_threads(int n) {
	if (g_blas_set_threads && n > 0)
		g_blas_set_threads(n);
}

#define OB_QBS 4096 /* distance_compute_blas_query_bs */
static int search_openblas(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k, float *D,
                           int64_t *I) {
	if (!g_sgemm)
		return fail("orc_flat_search", "oracle/orc_core.c", "ORC_PATH_OPENBLAS: call orc_openblas_load() first");
	const int is_max = metric == ORC_METRIC_L2;
	float *xn = NULL, *yn = NULL;
	if (is_max) {
		xn = (float *)malloc((size_t)(nq > 0 ? nq : 1) * sizeof(float));
		yn = (float *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(float));
		orc_norms(xq, nq, d, xn);
		orc_norms(xb, nb, d, yn);
	}
	float *ip_block = (float *)malloc((size_t)OB_QBS * BLAS_DBS * sizeof(float));
	const int resv = use_reservoir(k);
	const int64_t rcap = (2 * k + 15) & ~(int64_t)15;
	const int64_t nres = nq < OB_QBS ? nq : OB_QBS;
	reservoir_t *rs = resv ? (reservoir_t *)malloc((size_t)nres * sizeof(reservoir_t)) : NULL;
	float *rv = resv ? (float *)malloc((size_t)nres * rcap * sizeof(float)) : NULL;
	int64_t *ri = resv ? (int64_t *)malloc((size_t)nres * rcap * sizeof(int64_t)) : NULL;
	for (int64_t i0 = 0; i0 < nq; i0 += OB_QBS) {
		const int64_t i1 = i0 + OB_QBS < nq ? i0 + OB_QBS : nq;
#pragma omp parallel for
		for (int64_t i = i0; i < i1; i++) { /* res.begin_multiple */
			if (resv)
				reservoir_begin(&rs[i - i0], is_max, k, rv + (size_t)(i - i0) * rcap, ri + (size_t)(i - i0) * rcap);
			else
				heap_init(k, D + i * k, I + i * k, is_max);
		}
		for (int64_t j0 = 0; j0 < nb; j0 += BLAS_DBS) {
			const int64_t j1 = j0 + BLAS_DBS < nb ? j0 + BLAS_DBS : nb;
			const int64_t nyi = j1 - j0, nxi = i1 - i0;
			g_sgemm(102 /* CblasColMajor */, 112 /* CblasTrans */, 111 /* CblasNoTrans */, nyi, nxi, d, 1.0f, xb + j0 * d, d,
			        xq + i0 * d, d, 0.0f, ip_block, nyi);
#pragma omp parallel for
			for (int64_t i = i0; i < i1; i++) { /* distance formula + HeapBlockResultHandler::add_results */
				const float *ip_line = ip_block + (i - i0) * nyi;
				float *hv = D + i * k;
				int64_t *hi = I + i * k;
				reservoir_t *r = resv ? &rs[i - i0] : NULL;
				float thr = resv ? r->thr : hv[0];
				if (is_max) {
					const float xni = xn[i];
					for (int64_t j = 0; j < nyi; j++) {
						float dis = (xni + yn[j0 + j]) - 2.0f * ip_line[j];
						if (dis < 0)
							dis = 0;
						if (thr > dis) {
							if (resv) {
								reservoir_add(r, dis, j0 + j);
								thr = r->thr;
							} else {
								heap_replace_top(k, hv, hi, 1, dis, j0 + j);
								thr = hv[0];
							}
						}
					}
				} else {
					for (int64_t j = 0; j < nyi; j++) {
						const float dis = ip_line[j];
						if (thr < dis) {
							if (resv) {
								reservoir_add(r, dis, j0 + j);
								thr = r->thr;
							} else {
								heap_replace_top(k, hv, hi, 0, dis, j0 + j);
								thr = hv[0];
							}
						}
					}
				}
			}
		}
#pragma omp parallel for
		for (int64_t i = i0; i < i1; i++) { /* res.end_multiple */
			if (resv)
				reservoir_to_result(&rs[i - i0], D + i * k, I + i * k);
			else
				heap_reorder(k, D + i * k, I + i * k, is_max);
		}
	}
	free(rs);
	free(rv);
	free(ri);
	free(ip_block);
	free(xn);
	free(yn);
	return 0;
}

static void translate_ids(int64_t n, int64_t *I, const int64_t *id_map) {
	if (!id_map)
		return;
	for (int64_t i = 0; i < n; i++)
		I[i] = I[i] < 0 ? I[i] : id_map[I[i]];
}

/* IndexFlat::search -> knn_L2sqr / knn_inner_product dispatch (distance_compute_blas_threshold = 20) */
static int flat_search_impl(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k,
                            float *D, int64_t *I, const orc_params *params, const int64_t *id_map) {
	if (k <= 0)
		return fail("virtual void faiss::IndexFlat::search(...) const", "faiss/IndexFlat.cpp", "Error: 'k > 0' failed");
	const int extra = metric != ORC_METRIC_L2 && metric != ORC_METRIC_INNER_PRODUCT;
	if (extra && !(metric == 2 || metric == 3 || metric == 4 || (metric >= 20 && metric <= 23)))
		return fail("orc_flat_search", "faiss/IndexFlat.cpp", "metric type %d not supported by the oracle", metric);
	sel_t sel;
	if (sel_build(&sel, params))
		return 1;
	int path = params ? params->force_path : ORC_PATH_AUTO;
	if (extra) /* IndexFlat::search -> knn_extra_metrics: always per pair */
		path = ORC_PATH_PAIR;
	if (path == ORC_PATH_AUTO)
		path = (sel.kind || nq < 20) ? ORC_PATH_PAIR : ORC_PATH_BLAS;
	if ((path == ORC_PATH_BLAS || path == ORC_PATH_OPENBLAS) && sel.kind) {
		sel_free(&sel);
		return fail("orc_flat_search", "faiss/utils/distances.cpp", "selector requires the per-pair path");
	}
	int rc = 0;
	if (path == ORC_PATH_PAIR)
		search_pair(metric, d, nb, xb, nq, xq, k, D, I, sel.kind ? &sel : NULL, id_map);
	else if (path == ORC_PATH_OPENBLAS)
		rc = search_openblas(metric, d, nb, xb, nq, xq, k, D, I);
	else
		search_blas(metric, d, nb, xb, nq, xq, k, D, I);
	sel_free(&sel);
	return rc;
}

int orc_flat_search(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k, float *D,
                    int64_t *I, const orc_params *params, const int64_t *id_map) {
	int rc = flat_search_impl(metric, d, nb, xb, nq, xq, k, D, I, params, id_map);
	if (!rc)
		translate_ids(nq * k, I, id_map);
	return rc;
}

/* naive triple loops: the definition the packed path must equal bit for bit */
int orc_flat_search_naive(int metric, int d, int64_t nb, const float *xb, int64_t nq, const float *xq, int64_t k,
                          float *D, int64_t *I, int path) {
	const int is_max = metric == ORC_METRIC_L2;
	for (int64_t i = 0; i < nq; i++) {
		const float *x = xq + i * d;
		float *hv = D + i * k;
		int64_t *hi = I + i * k;
		heap_init(k, hv, hi, is_max);
		float xn = ip_chain(x, x, d);
		for (int64_t j = 0; j < nb; j++) {
			const float *y = xb + j * d;
			float dis;
			if (!is_max) {
				dis = ip_chain(x, y, d);
			} else if (path == ORC_PATH_PAIR) {
				dis = l2_chain(x, y, d);
			} else {
				float yn = ip_chain(y, y, d);
				float ip = ip_chain(x, y, d);
				dis = (xn + yn) - 2.0f * ip;
				if (dis < 0)
					dis = 0;
			}
			if (accepts(is_max, hv[0], dis))
				heap_replace_top(k, hv, hi, is_max, dis, j);
		}
		heap_reorder(k, hv, hi, is_max);
	}
	return 0;
}

/* --------------------------------------------------------------- index types */

enum { IX_FLAT = 1, IX_IDMAP = 2, IX_IVFFLAT = 3, IX_HNSW = 4 };

typedef struct {
	int64_t n, cap;
	int64_t *ids;
	float *codes;
} invlist_t;

struct orc_index {
	int type, d, metric;
	int64_t ntotal;
	int is_trained;
	/* flat (IndexFlatCodes: append-only row-major codes) */
	float *xb;
	int64_t cap;
	/* idmap */
	orc_index *sub;
	int64_t *id_map;
	int64_t idcap;
	/* ivf */
	orc_index *quantizer;
	int64_t nlist, nprobe;
	invlist_t *lists;
	int spherical;
	/* hnsw (IndexHNSWFlat: storage = IndexFlat in ->sub is NOT used; rows live in xb like a flat index) */
	orc_hnsw *hnsw;
};

int orc_d(const orc_index *ix) {
	return ix->d;
}
int64_t orc_ntotal(const orc_index *ix) {
	return ix->ntotal;
}
int orc_is_trained(const orc_index *ix) {
	return ix->is_trained;
}
int orc_metric(const orc_index *ix) {
	return ix->metric;
}

static orc_index *new_flat(int d, int metric) {
	orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
	ix->type = IX_FLAT;
	ix->d = d;
	ix->metric = metric;
	ix->is_trained = 1;
	return ix;
}
static void flat_reset(orc_index *ix) {
	ix->ntotal = 0;
}
static int flat_add(orc_index *ix, int64_t n, const float *x) {
	if (ix->ntotal + n > ix->cap) {
		int64_t nc = ix->cap ? ix->cap : 1024;
		while (nc < ix->ntotal + n)
			nc *= 2;
		ix->xb = (float *)realloc(ix->xb, (size_t)nc * ix->d * sizeof(float));
		ix->cap = nc;
	}
	memcpy(ix->xb + ix->ntotal * ix->d, x, (size_t)n * ix->d * sizeof(float));
	ix->ntotal += n;
	return 0;
}

void orc_index_free(orc_index *ix) {
	if (!ix)
		return;
	free(ix->xb);
	free(ix->id_map);
	orc_index_free(ix->sub);
	orc_index_free(ix->quantizer);
	if (ix->lists) {
		for (int64_t l = 0; l < ix->nlist; l++) {
			free(ix->lists[l].ids);
			free(ix->lists[l].codes);
		}
		free(ix->lists);
	}
	orc_hnsw_free(ix->hnsw);
	free(ix);
}

/* faiss::index_factory subset (index_factory.cpp): the strings the reference and its
 * tests use -- "Flat", "IDMap,<sub>", "IDMap2,<sub>", "IVF<n>,Flat". */
static orc_index *factory_rec(int d, const char *desc, int metric, const char *full) {
	if (!strncmp(desc, "IDMap2,", 7) || !strncmp(desc, "IDMap,", 6)) {
		const char *rest = strchr(desc, ',') + 1;
		orc_index *sub = factory_rec(d, rest, metric, full);
		if (!sub)
			return NULL;
		orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
		ix->type = IX_IDMAP;
		ix->d = d;
		ix->metric = metric;
		ix->sub = sub;
		ix->is_trained = sub->is_trained;
		return ix;
	}
	if (!strcmp(desc, "Flat"))
		return new_flat(d, metric);
	if (!strncmp(desc, "IVF", 3)) {
		char *end;
		long nlist = strtol(desc + 3, &end, 10);
		int hnsw_M = 0;
		if (end != desc + 3 && !strncmp(end, "_HNSW", 5)) { /* "IVF<n>_HNSW<m>,Flat": HNSW coarse quantizer, M default 32 */
			char *end2;
			long m = strtol(end + 5, &end2, 10);
			hnsw_M = end2 == end + 5 ? 32 : (int)m;
			end = end2;
		}
		if (end != desc + 3 && nlist > 0 && !strcmp(end, ",Flat")) {
			orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
			ix->type = IX_IVFFLAT;
			ix->d = d;
			ix->metric = metric;
			ix->nlist = nlist;
			ix->nprobe = 1;
			ix->quantizer = new_flat(d, metric); /* IndexFlat(d, metric) coarse quantizer */
			if (hnsw_M > 1) {                    /* IndexHNSWFlat(d, M, metric) */
				ix->quantizer->type = IX_HNSW;
				ix->quantizer->hnsw = orc_hnsw_new(hnsw_M);
			}
			ix->lists = (invlist_t *)calloc((size_t)nlist, sizeof(invlist_t));
			ix->is_trained = 0;
			/* IndexIVF ctor: "Spherical by default if the metric is inner_product" */
			ix->spherical = metric == ORC_METRIC_INNER_PRODUCT;
			return ix;
		}
	}
	if (!strncmp(desc, "HNSW", 4)) {
		/* "HNSW<M>" and "HNSW<M>,Flat" -> IndexHNSWFlat(d, M, metric); bare "HNSW" means M = 32 */
		char *end;
		long M = strtol(desc + 4, &end, 10);
		if (end == desc + 4)
			M = 32;
		if (M > 1 && (!*end || !strcmp(end, ",Flat"))) {
			orc_index *ix = new_flat(d, metric);
			ix->type = IX_HNSW;
			ix->hnsw = orc_hnsw_new((int)M);
			return ix;
		}
	}
	fail("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
	     "could not parse index string %s", full);
	return NULL;
}
orc_index *orc_index_factory(int d, const char *desc, int metric) {
	if (d <= 0) {
		fail("orc_index_factory", "faiss/index_factory.cpp", "invalid dimension %d", d);
		return NULL;
	}
	return factory_rec(d, desc, metric, desc);
}

/* --------------------------------------------------------------- clustering */
/* faiss/Clustering.cpp: niter 25, nredo 1, seed 1234, min/max points per centroid 39/256 */

static void renorm_l2(int d, int64_t n, float *x) {
	for (int64_t i = 0; i < n; i++) {
		float *xi = x + i * d;
		float nr = ip_chain(xi, xi, d);
		if (nr > 0) {
			const float inv = 1.0f / sqrtf(nr);
			for (int j = 0; j < d; j++)
				xi[j] *= inv;
		}
	}
}

static int kmeans_train(int d, int64_t k, int64_t nx, const float *x_in, orc_index *qz, int spherical, float *centroids) {
	/* ClusteringParameters defaults except niter: faiss/IndexIVF.cpp Level1Quantizer::Level1Quantizer sets
	 * cp.niter = 10 and this k-means is only reached through IndexIVF::train (train_q1) */
	const int niter = 10, max_pts = 256, min_pts = 39;
	const int64_t seed = 1234;
	if (nx < k)
		return fail("virtual void faiss::Clustering::train_encoded(...)", "faiss/Clustering.cpp",
		            "Error: 'nx >= k' failed: Number of training points (%ld) should be at least as large as number "
		            "of clusters (%ld)",
		            (long)nx, (long)k);
	for (int64_t i = 0; i < nx * d; i++)
		if (!isfinite(x_in[i]))
			return fail("virtual void faiss::Clustering::train_encoded(...)", "faiss/Clustering.cpp",
			            "input contains NaN's or Inf's");
	const float *x = x_in;
	float *xsub = NULL;
	if (nx > k * max_pts) {
		/* subsample_training_set: first k*256 entries of rand_perm(nx, seed) */
		int *perm = (int *)malloc((size_t)nx * sizeof(int));
		rand_perm(perm, (size_t)nx, seed);
		int64_t n2 = k * max_pts;
		xsub = (float *)malloc((size_t)n2 * d * sizeof(float));
		for (int64_t i = 0; i < n2; i++)
			memcpy(xsub + i * d, x_in + (int64_t)perm[i] * d, (size_t)d * sizeof(float));
		free(perm);
		x = xsub;
		nx = n2;
	} else if (nx < k * min_pts) {
		fprintf(stderr, "WARNING clustering %ld points to %ld centroids: please provide at least %ld training points\n",
		        (long)nx, (long)k, (long)(k * min_pts));
	}
	if (nx == k) {
		memcpy(centroids, x, (size_t)k * d * sizeof(float));
		flat_reset(qz);
		flat_add(qz, k, centroids);
		free(xsub);
		return 0;
	}
	int *perm = (int *)malloc((size_t)nx * sizeof(int));
	rand_perm(perm, (size_t)nx, seed + 1);
	for (int64_t i = 0; i < k; i++)
		memcpy(centroids + i * d, x + (int64_t)perm[i] * d, (size_t)d * sizeof(float));
	free(perm);
	if (spherical)
		renorm_l2(d, k, centroids);
	flat_reset(qz);
	flat_add(qz, k, centroids);

	int64_t *assign = (int64_t *)malloc((size_t)nx * sizeof(int64_t));
	float *dis = (float *)malloc((size_t)nx * sizeof(float));
	float *hassign = (float *)malloc((size_t)k * sizeof(float));
	int rc = 0;
	for (int it = 0; it < niter && !rc; it++) {
		rc = flat_search_impl(qz->metric, d, qz->ntotal, qz->xb, nx, x, 1, dis, assign, NULL, NULL);
		if (rc)
			break;
		/* compute_centroids: sums in input order, then scale by 1/count */
		memset(centroids, 0, (size_t)k * d * sizeof(float));
		memset(hassign, 0, (size_t)k * sizeof(float));
		for (int64_t i = 0; i < nx; i++) {
			int64_t ci = assign[i];
			float *c = centroids + ci * d;
			const float *xi = x + i * d;
			hassign[ci] += 1.0f;
			for (int j = 0; j < d; j++)
				c[j] += xi[j];
		}
		for (int64_t ci = 0; ci < k; ci++) {
			if (hassign[ci] == 0)
				continue;
			float norm = 1 / hassign[ci];
			float *c = centroids + ci * d;
			for (int j = 0; j < d; j++)
				c[j] *= norm;
		}
		/* split_clusters: refill void clusters from a cluster chosen proportionally to its size */
		{
			const float EPS = (float)(1 / 1024.);
			mt19937_t rng;
			mt_seed(&rng, 1234);
			for (int64_t ci = 0; ci < k; ci++) {
				if (hassign[ci] != 0)
					continue;
				int64_t cj;
				for (cj = 0;; cj = (cj + 1) % k) {
					float p = (hassign[cj] - 1.0f) / (float)(nx - k);
					float r = mt_rand_float(&rng);
					if (r < p)
						break;
				}
				memcpy(centroids + ci * d, centroids + cj * d, (size_t)d * sizeof(float));
				for (int j = 0; j < d; j++) {
					if (j % 2 == 0) {
						centroids[ci * d + j] *= 1 + EPS;
						centroids[cj * d + j] *= 1 - EPS;
					} else {
						centroids[ci * d + j] *= 1 - EPS;
						centroids[cj * d + j] *= 1 + EPS;
					}
				}
				hassign[ci] = hassign[cj] / 2;
				hassign[cj] -= hassign[ci];
			}
		}
		if (spherical)
			renorm_l2(d, k, centroids);
		flat_reset(qz);
		flat_add(qz, k, centroids);
	}
	free(assign);
	free(dis);
	free(hassign);
	free(xsub);
	return rc;
}

/* ---------------------------------------------------------------------- IVF */

static void invlist_append(invlist_t *l, int d, int64_t id, const float *x) {
	if (l->n == l->cap) {
		int64_t nc = l->cap ? l->cap * 2 : 16;
		l->ids = (int64_t *)realloc(l->ids, (size_t)nc * sizeof(int64_t));
		l->codes = (float *)realloc(l->codes, (size_t)nc * d * sizeof(float));
		l->cap = nc;
	}
	l->ids[l->n] = id;
	memcpy(l->codes + l->n * d, x, (size_t)d * sizeof(float));
	l->n++;
}

static int hnsw_search(const orc_index *ix, int64_t nq, const float *xq, int64_t k, float *D, int64_t *I,
                       const orc_params *params, const int64_t *id_map);
/* coarse quantizer search: IndexFlat, or IndexHNSWFlat for "IVF<n>_HNSW<m>,Flat" (efSearch = the quantizer_params the
 * glue builds at src/faiss_extension.cpp:679-681, default 16) */
static int quantizer_search(const orc_index *qz, int64_t nq, const float *xq, int64_t k, float *D, int64_t *I,
                            int64_t efSearch) {
	if (qz->type == IX_HNSW) {
		orc_params p;
		memset(&p, 0, sizeof p);
		p.efSearch = efSearch;
		return hnsw_search(qz, nq, xq, k, D, I, &p, NULL);
	}
	return flat_search_impl(qz->metric, qz->d, qz->ntotal, qz->xb, nq, xq, k, D, I, NULL, NULL);
}

/* IndexIVF::train -> Level1Quantizer::train_q1.  Flat quantizer: quantizer_trains_alone == 0, k-means assigns with
 * the quantizer itself.  HNSW quantizer (index_factory sets quantizer_trains_alone = 2): "kmeans training on a flat
 * index + add the centroids to the quantizer" -- the assigner is an IndexFlatL2 whatever the metric. */
static int ivf_train(orc_index *ix, int64_t n, const float *x) {
	if (ix->quantizer->is_trained && ix->quantizer->ntotal == ix->nlist) {
		ix->is_trained = 1; /* "IVF quantizer does not need training." */
		return 0;
	}
	float *cent = (float *)malloc((size_t)ix->nlist * ix->d * sizeof(float));
	int rc;
	if (ix->quantizer->type == IX_HNSW) {
		orc_index *assigner = new_flat(ix->d, ORC_METRIC_L2);
		rc = kmeans_train(ix->d, ix->nlist, n, x, assigner, ix->spherical, cent);
		if (!rc) {
			memcpy(cent, assigner->xb, (size_t)ix->nlist * ix->d * sizeof(float)); /* final centroids */
			rc = orc_add(ix->quantizer, ix->nlist, cent);
		}
		orc_index_free(assigner);
	} else {
		flat_reset(ix->quantizer);
		rc = kmeans_train(ix->d, ix->nlist, n, x, ix->quantizer, ix->spherical, cent);
	}
	free(cent);
	if (rc)
		return rc;
	ix->is_trained = 1;
	return 0;
}

/* IndexIVF::add_with_ids: blocks of 65536; quantizer->assign = search(k=1) labels;
 * IndexIVFFlat::add_core appends (id, raw vector) in input order to list assign[i] */
static int ivf_add(orc_index *ix, int64_t n, const float *x, const int64_t *xids) {
	if (!ix->is_trained)
		return fail("virtual void faiss::IndexIVFFlat::add_core(...)", "faiss/IndexIVFFlat.cpp",
		            "Error: 'is_trained' failed");
	const int64_t bs = 65536;
	for (int64_t i0 = 0; i0 < n; i0 += bs) {
		int64_t nb = n - i0 < bs ? n - i0 : bs;
		int64_t *assign = (int64_t *)malloc((size_t)nb * sizeof(int64_t));
		float *dis = (float *)malloc((size_t)nb * sizeof(float));
		int rc = quantizer_search(ix->quantizer, nb, x + i0 * ix->d, 1, dis, assign, 16);
		if (rc) {
			free(assign);
			free(dis);
			return rc;
		}
		for (int64_t i = 0; i < nb; i++) {
			int64_t list_no = assign[i];
			if (list_no < 0)
				continue;
			int64_t id = xids ? xids[i0 + i] : ix->ntotal + i;
			invlist_append(&ix->lists[list_no], ix->d, id, x + (i0 + i) * ix->d);
		}
		ix->ntotal += nb;
		free(assign);
		free(dis);
	}
	return 0;
}

/* IndexIVF::search -> search_preassigned + IVFFlatScanner::scan_codes */
static int ivf_search(const orc_index *ix, int64_t nq, const float *xq, int64_t k, float *D, int64_t *I,
                      const orc_params *params, const int64_t *id_map) {
	if (k <= 0)
		return fail("virtual void faiss::IndexIVF::search(...) const", "faiss/IndexIVF.cpp", "Error: 'k > 0' failed");
	int64_t nprobe = params && params->nprobe > 0 ? params->nprobe : ix->nprobe;
	if (nprobe > ix->nlist)
		nprobe = ix->nlist;
	if (nprobe <= 0)
		return fail("virtual void faiss::IndexIVF::search(...) const", "faiss/IndexIVF.cpp",
		            "Error: 'nprobe > 0' failed");
	sel_t sel;
	if (sel_build(&sel, params))
		return 1;
	const int is_max = ix->metric == ORC_METRIC_L2;
	const int d = ix->d;
	int64_t *keys = (int64_t *)malloc((size_t)nq * nprobe * sizeof(int64_t));
	float *cdis = (float *)malloc((size_t)nq * nprobe * sizeof(float));
	/* coarse quantisation: quantizer->search(n, x, nprobe) on the whole batch (FAISS slices the
	 * batch by OpenMP thread count, which makes its pair/BLAS choice machine dependent) */
	int rc = quantizer_search(ix->quantizer, nq, xq, nprobe, cdis, keys,
	                          params && params->efSearch > 0 ? params->efSearch : 16);
	if (!rc) {
#pragma omp parallel for schedule(dynamic, 1)
		for (int64_t i = 0; i < nq; i++) {
			const float *x = xq + i * d;
			float *hv = D + i * k;
			int64_t *hi = I + i * k;
			heap_init(k, hv, hi, is_max);
			for (int64_t p = 0; p < nprobe; p++) {
				int64_t key = keys[i * nprobe + p];
				if (key < 0)
					continue; /* not enough centroids for multiprobe */
				const invlist_t *l = &ix->lists[key];
				for (int64_t j = 0; j < l->n; j++) {
					if (sel.kind && !sel_member(&sel, id_map ? id_map[l->ids[j]] : l->ids[j]))
						continue;
					const float *y = l->codes + j * d;
					float dis = is_max ? l2_chain(x, y, d) : ip_chain(x, y, d);
					if (accepts(is_max, hv[0], dis))
						heap_replace_top(k, hv, hi, is_max, dis, l->ids[j]);
				}
			}
			heap_reorder(k, hv, hi, is_max);
		}
	}
	free(keys);
	free(cdis);
	sel_free(&sel);
	return rc;
}

int64_t orc_ivf_nlist(const orc_index *ix) {
	while (ix->type == IX_IDMAP)
		ix = ix->sub;
	return ix->type == IX_IVFFLAT ? ix->nlist : 0;
}
int orc_ivf_get_centroids(const orc_index *ix, float *out) {
	while (ix->type == IX_IDMAP)
		ix = ix->sub;
	if (ix->type != IX_IVFFLAT || ix->quantizer->ntotal != ix->nlist)
		return fail("orc_ivf_get_centroids", "oracle", "not a trained IVF index");
	memcpy(out, ix->quantizer->xb, (size_t)ix->nlist * ix->d * sizeof(float));
	return 0;
}
int orc_ivf_set_centroids(orc_index *ix, const float *c) {
	orc_index *top = ix;
	while (ix->type == IX_IDMAP)
		ix = ix->sub;
	if (ix->type != IX_IVFFLAT)
		return fail("orc_ivf_set_centroids", "oracle", "not an IVF index");
	if (ix->quantizer->type == IX_HNSW) { /* centroids are inserted into the (still empty) graph */
		if (ix->quantizer->ntotal != 0)
			return fail("orc_ivf_set_centroids", "oracle", "the HNSW coarse quantizer already holds centroids");
		int rc = orc_add(ix->quantizer, ix->nlist, c);
		if (rc)
			return rc;
	} else {
		flat_reset(ix->quantizer);
		flat_add(ix->quantizer, ix->nlist, c);
	}
	ix->is_trained = 1;
	for (orc_index *p = top; p->type == IX_IDMAP; p = p->sub)
		p->is_trained = 1;
	return 0;
}
int64_t orc_ivf_list_size(const orc_index *ix, int64_t list_no) {
	while (ix->type == IX_IDMAP)
		ix = ix->sub;
	if (ix->type != IX_IVFFLAT || list_no < 0 || list_no >= ix->nlist)
		return -1;
	return ix->lists[list_no].n;
}
int orc_ivf_get_list(const orc_index *ix, int64_t list_no, int64_t *ids, float *codes) {
	while (ix->type == IX_IDMAP)
		ix = ix->sub;
	if (ix->type != IX_IVFFLAT || list_no < 0 || list_no >= ix->nlist)
		return fail("orc_ivf_get_list", "oracle", "bad list");
	const invlist_t *l = &ix->lists[list_no];
	if (ids)
		memcpy(ids, l->ids, (size_t)l->n * sizeof(int64_t));
	if (codes)
		memcpy(codes, l->codes, (size_t)l->n * ix->d * sizeof(float));
	return 0;
}

/* ------------------------------------------------------------ public Index API */

int orc_train(orc_index *ix, int64_t n, const float *x) {
	switch (ix->type) {
	case IX_FLAT:
	case IX_HNSW: /* IndexHNSW::train trains the storage; flat storage: nothing to do */
		return 0; /* Index::train: does nothing by default */
	case IX_IDMAP: {
		int rc = orc_train(ix->sub, n, x);
		ix->is_trained = ix->sub->is_trained;
		return rc;
	}
	case IX_IVFFLAT:
		return ivf_train(ix, n, x);
	}
	return fail("orc_train", "oracle", "bad index");
}

int orc_add(orc_index *ix, int64_t n, const float *x) {
	switch (ix->type) {
	case IX_FLAT:
		return flat_add(ix, n, x);
	case IX_IDMAP:
		return fail("virtual void faiss::IndexIDMapTemplate<IndexT>::add(...)", "faiss/IndexIDMap.cpp",
		            "add does not make sense with IndexIDMap, use add_with_ids");
	case IX_IVFFLAT:
		return ivf_add(ix, n, x, NULL);
	case IX_HNSW: {
		/* IndexHNSW::add: storage->add(n, x) then hnsw_add_vertices(*this, n0, n, x, ...) */
		int64_t n0 = ix->ntotal;
		flat_add(ix, n, x);
		orc_hnsw_add(ix->hnsw, n0, n, ix->xb, ix->d, ix->metric == ORC_METRIC_L2);
		return 0;
	}
	}
	return fail("orc_add", "oracle", "bad index");
}

int orc_add_with_ids(orc_index *ix, int64_t n, const float *x, const int64_t *ids) {
	switch (ix->type) {
	case IX_FLAT:
	case IX_HNSW:
		/* Index::add_with_ids default (faiss/Index.cpp) -- substring matched at src/faiss_extension.cpp:523,
		 * user-visible text pinned by test/sql/faiss4.test:19-22 */
		return fail("virtual void faiss::Index::add_with_ids(faiss::idx_t, const float*, const faiss::idx_t*)",
		            "faiss/Index.cpp", "add_with_ids not implemented for this type of index");
	case IX_IDMAP: {
		int rc = orc_add(ix->sub, n, x);
		if (rc)
			return rc;
		if (ix->ntotal + n > ix->idcap) {
			int64_t nc = ix->idcap ? ix->idcap : 1024;
			while (nc < ix->ntotal + n)
				nc *= 2;
			ix->id_map = (int64_t *)realloc(ix->id_map, (size_t)nc * sizeof(int64_t));
			ix->idcap = nc;
		}
		memcpy(ix->id_map + ix->ntotal, ids, (size_t)n * sizeof(int64_t));
		ix->ntotal = ix->sub->ntotal;
		return 0;
	}
	case IX_IVFFLAT:
		return ivf_add(ix, n, x, ids);
	}
	return fail("orc_add_with_ids", "oracle", "bad index");
}

/* IndexHNSW::search (IndexHNSW.cpp hnsw_search): per query HNSW::search, efSearch from SearchParametersHNSW */
static int hnsw_search(const orc_index *ix, int64_t nq, const float *xq, int64_t k, float *D, int64_t *I,
                       const orc_params *params, const int64_t *id_map) {
	sel_t sel;
	if (sel_build(&sel, params))
		return 1;
	const int ef = params && params->efSearch > 0 ? (int)params->efSearch : 16;
	const int is_l2 = ix->metric == ORC_METRIC_L2;
#pragma omp parallel
	{
		uint8_t *visited = (uint8_t *)calloc((size_t)(ix->ntotal > 0 ? ix->ntotal : 1), 1);
#pragma omp for schedule(dynamic, 4)
		for (int64_t q = 0; q < nq; q++)
			orc_hnsw_search_one(ix->hnsw, ix->xb, ix->d, is_l2, xq + q * ix->d, k, ef, D + q * k, I + q * k, visited,
			                    sel.kind ? &sel : NULL, id_map);
		free(visited);
	}
	sel_free(&sel);
	return 0;
}

static int search_rec(const orc_index *ix, int64_t nq, const float *x, int64_t k, float *D, int64_t *I,
                      const orc_params *params, const int64_t *id_map) {
	switch (ix->type) {
	case IX_FLAT:
		return flat_search_impl(ix->metric, ix->d, ix->ntotal, ix->xb, nq, x, k, D, I, params, id_map);
	case IX_IVFFLAT:
		return ivf_search(ix, nq, x, k, D, I, params, id_map);
	case IX_HNSW:
		return hnsw_search(ix, nq, x, k, D, I, params, id_map);
	case IX_IDMAP: {
		/* IndexIDMap::search: a user selector is wrapped in IDSelectorTranslated(id_map, sel);
		 * inner search; labels[i] = labels[i] < 0 ? labels[i] : id_map[labels[i]] */
		if (id_map)
			return fail("orc_search", "faiss/IndexIDMap.cpp", "nested IDMap not supported by the oracle");
		int rc = search_rec(ix->sub, nq, x, k, D, I, params, ix->id_map);
		if (!rc)
			translate_ids(nq * k, I, ix->id_map);
		return rc;
	}
	}
	return fail("orc_search", "oracle", "bad index");
}

int orc_search(const orc_index *ix, int64_t nq, const float *x, int64_t k, float *D, int64_t *I,
               const orc_params *params) {
	return search_rec(ix, nq, x, k, D, I, params, NULL);
}

/* ------------------------------------------------------------- shard merging */
/* Row-sharded search: each shard returns k results per query, already in FAISS order, with
 * GLOBAL labels.  The merged list is the k best under the total order
 *   L2: (distance asc, label asc)      IP: membership (score desc, label asc), printed
 *   with equal scores in descending label order (heap_reorder's order) -- DESIGN.md "ties". */
typedef struct {
	float v;
	int64_t id;
} pair_t;
static int g_merge_is_max;
static int cmp_pair(const void *a, const void *b) {
	const pair_t *x = (const pair_t *)a, *y = (const pair_t *)b;
	if (x->id < 0 || y->id < 0) {
		if (x->id < 0 && y->id < 0)
			return 0;
		return x->id < 0 ? 1 : -1;
	}
	if (x->v != y->v) {
		if (g_merge_is_max)
			return x->v < y->v ? -1 : 1;
		return x->v > y->v ? -1 : 1;
	}
	return (x->id > y->id) - (x->id < y->id);
}
void orc_merge_shards(int metric, int64_t nq, int64_t k, int nshard, const float *D, const int64_t *I, float *Dout,
                      int64_t *Iout) {
	const int is_max = metric == ORC_METRIC_L2;
	g_merge_is_max = is_max;
	pair_t *buf = (pair_t *)malloc((size_t)nshard * k * sizeof(pair_t));
	for (int64_t q = 0; q < nq; q++) {
		int64_t n = 0;
		for (int s = 0; s < nshard; s++)
			for (int64_t j = 0; j < k; j++) {
				buf[n].v = D[((int64_t)s * nq + q) * k + j];
				buf[n].id = I[((int64_t)s * nq + q) * k + j];
				n++;
			}
		qsort(buf, (size_t)n, sizeof(pair_t), cmp_pair);
		int64_t m = 0;
		for (; m < k && m < n && buf[m].id >= 0; m++) {
			Dout[q * k + m] = buf[m].v;
			Iout[q * k + m] = buf[m].id;
		}
		if (!is_max) {
			/* equal scores are printed in descending label order */
			int64_t a = 0;
			while (a < m) {
				int64_t b = a + 1;
				while (b < m && Dout[q * k + b] == Dout[q * k + a])
					b++;
				for (int64_t lo = a, hi = b - 1; lo < hi; lo++, hi--) {
					int64_t t = Iout[q * k + lo];
					Iout[q * k + lo] = Iout[q * k + hi];
					Iout[q * k + hi] = t;
				}
				a = b;
			}
		}
		for (; m < k; m++) {
			Dout[q * k + m] = neutral(is_max);
			Iout[q * k + m] = -1;
		}
	}
	free(buf);
}

/* ------------------------------------------------------- synthetic generator */
/* splitmix64 of (seed + (row*d+col+1)*gamma); top 24 bits -> uniform [0,1) f32.  The device
 * generator (duckdb-faiss-ext_amd/csrc/synth.hip) is the same integer arithmetic. */
static inline uint64_t splitmix(uint64_t z) {
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static inline float u01(uint64_t seed, uint64_t ctr) {
	uint64_t z = splitmix(seed + (ctr + 1) * 0x9E3779B97F4A7C15ull);
	return (float)(z >> 40) * (1.0f / 16777216.0f);
}
void orc_synth_uniform(float *out, int64_t n_rows, int d, uint64_t seed, int64_t row0) {
#pragma omp parallel for schedule(static)
	for (int64_t r = 0; r < n_rows; r++)
		for (int c = 0; c < d; c++)
			out[r * d + c] = u01(seed, (uint64_t)(row0 + r) * (uint64_t)d + (uint64_t)c);
}
/* Gaussian-ish mixture without transcendental functions (Irwin-Hall of 4 uniforms, exactly
 * reproducible on host and device): value = centre[c][col] + sigma * g */
static inline float ih4(uint64_t seed, uint64_t ctr) {
	float a = u01(seed, 4 * ctr), b = u01(seed, 4 * ctr + 1), c = u01(seed, 4 * ctr + 2), e = u01(seed, 4 * ctr + 3);
	return (((a + b) + (c + e)) - 2.0f) * 1.7320508f;
}
void orc_synth_clustered(float *out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers, float sigma) {
	const uint64_t cseed = 0xC0FFEEull; /* centres are shared by database and queries */
#pragma omp parallel for schedule(static)
	for (int64_t r = 0; r < n_rows; r++) {
		uint64_t row = (uint64_t)(row0 + r);
		uint64_t c = splitmix(seed ^ (row * 0xD1B54A32D192ED03ull + 0x5851F42D4C957F2Dull)) % (uint64_t)n_centers;
		for (int col = 0; col < d; col++) {
			float centre = ih4(cseed, c * (uint64_t)d + (uint64_t)col);
			float g = ih4(seed, row * (uint64_t)d + (uint64_t)col);
			out[r * d + col] = fmaf(sigma, g, centre);
		}
	}
}

/* ------------------------------------------------------------------ HNSW API */
static orc_index *hnsw_of(orc_index *ix) {
	if (ix && ix->type == IX_IDMAP)
		ix = ix->sub;
	if (ix && ix->type == IX_IVFFLAT) /* "IVF<n>_HNSW<m>,Flat": the coarse quantizer */
		ix = ix->quantizer;
	return ix && ix->type == IX_HNSW ? ix : NULL;
}
int orc_hnsw_set_ef_construction_ix(orc_index *ix, int v) {
	orc_index *h = hnsw_of(ix);
	if (!h)
		return fail("orc_hnsw_set_ef_construction", "oracle", "not an HNSW index");
	orc_hnsw_set_ef_construction(h->hnsw, v);
	return 0;
}
int64_t orc_hnsw_graph_size(orc_index *ix, int *max_level, int32_t *entry_point) {
	orc_index *h = hnsw_of(ix);
	if (!h)
		return -1;
	if (max_level)
		*max_level = orc_hnsw_max_level(h->hnsw);
	if (entry_point)
		*entry_point = orc_hnsw_entry_point(h->hnsw);
	return orc_hnsw_nb_total(h->hnsw);
}
int orc_hnsw_get_graph(orc_index *ix, int *levels, int64_t *offsets, int32_t *neighbors) {
	orc_index *h = hnsw_of(ix);
	if (!h)
		return fail("orc_hnsw_get_graph", "oracle", "not an HNSW index");
	orc_hnsw_export(h->hnsw, levels, offsets, neighbors);
	return 0;
}
/* replace rows + graph of an (empty or not) HNSW index by externally built ones; the level RNG is NOT advanced, so
 * the index is meant for search only afterwards */
int orc_hnsw_set_graph(orc_index *ix, int64_t n, const float *x, const int *levels, const int64_t *offsets,
                       const int32_t *neighbors, int32_t entry_point, int max_level) {
	orc_index *h = ix && ix->type == IX_HNSW ? ix : NULL;
	if (!h)
		return fail("orc_hnsw_set_graph", "oracle", "not a plain HNSW index");
	h->ntotal = 0;
	flat_add(h, n, x);
	orc_hnsw_import(h->hnsw, n, levels, offsets, neighbors, entry_point, max_level);
	return 0;
}
