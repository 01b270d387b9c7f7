/*
 * oracle/orc_hnsw.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): restatement of FAISS's IndexHNSWFlat.
 *
 * Reached from the reference through index_factory("HNSW<M>" / "IDMap,HNSW<M>", src/faiss_extension.cpp:154,
 * README.md:61), hnsw.efConstruction (:136-139), Index::add (:512) and Index::search with
 * SearchParametersHNSW{efSearch, sel} (:631, :691-702).  FAISS is an un-vendored submodule: this restates the
 * published algorithm of [UPSTREAM] faiss/impl/HNSW.cpp (set_default_probas, random_level, prepare_level_tab,
 * add_with_locks, greedy_update_nearest, search_neighbors_to_add, shrink_neighbor_list, add_link, MinimaxHeap,
 * search_from_candidates, HNSW::search) and faiss/IndexHNSW.cpp (hnsw_add_vertices, IndexHNSW::search) --
 * SURVEY.md Appendix A.8.  The reference holds NO golden values for HNSW results: parity unpinned by the reference.
 *
 * Determinism: FAISS builds with OpenMP and per-vertex locks, so its graph depends on thread interleaving; this
 * restatement is the single-thread order (order[] walked front to back).  Distances are the canonical k-ordered fma
 * chains of orc.h; inner product is searched as the negated value (NegativeDistanceComputer).
 */
#include "orc_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- small containers */

typedef struct {
	float d;
	int32_t id;
} nd_t;

/* binary heap of nd_t; farther_top = 1: top is the LARGEST d (std::priority_queue<NodeDistCloser>),
 * farther_top = 0: top is the SMALLEST d (std::priority_queue<NodeDistFarther>) */
typedef struct {
	nd_t *a;
	int n, cap, farther_top;
} pq_t;
static void pq_init(pq_t *q, int farther_top) {
	q->a = NULL;
	q->n = q->cap = 0;
	q->farther_top = farther_top;
}
static void pq_free(pq_t *q) {
	free(q->a);
	q->a = NULL;
	q->n = q->cap = 0;
}
/* std::priority_queue leaves the order of equal distances to the library's heap routines; the oracle fixes it with
 * the id as a secondary key so that the result is implementation independent (DESIGN.md "HNSW ties") */
static inline int pq_before(const pq_t *q, nd_t x, nd_t y) { /* x nearer the top than y */
	if (x.d != y.d)
		return q->farther_top ? x.d > y.d : x.d < y.d;
	return q->farther_top ? x.id > y.id : x.id < y.id;
}
static void pq_push(pq_t *q, float d, int32_t id) {
	if (q->n == q->cap) {
		q->cap = q->cap ? 2 * q->cap : 64;
		q->a = (nd_t *)realloc(q->a, (size_t)q->cap * sizeof(nd_t));
	}
	int i = q->n++;
	nd_t v = {d, id};
	while (i > 0) {
		int p = (i - 1) / 2;
		if (!pq_before(q, v, q->a[p]))
			break;
		q->a[i] = q->a[p];
		i = p;
	}
	q->a[i] = v;
}
static nd_t pq_top(const pq_t *q) {
	return q->a[0];
}
static void pq_pop(pq_t *q) {
	nd_t v = q->a[--q->n];
	int i = 0;
	for (;;) {
		int l = 2 * i + 1, r = l + 1, c;
		if (l >= q->n)
			break;
		c = (r < q->n && pq_before(q, q->a[r], q->a[l])) ? r : l;
		if (!pq_before(q, q->a[c], v))
			break;
		q->a[i] = q->a[c];
		i = c;
	}
	if (q->n > 0)
		q->a[i] = v;
}

/* ---------------------------------------------------------------- HNSW structure */

struct orc_hnsw {
	int M;
	int nprob;
	double *assign_probas;
	int *cum_nn; /* cum_nneighbor_per_level, nprob + 1 entries */
	int *levels; /* per vertex: level + 1 */
	int64_t *offsets; /* n + 1 */
	int32_t *neighbors;
	int64_t n, cap_n, cap_nb;
	int32_t entry_point;
	int max_level;
	int efConstruction, efSearch;
	orc_mt19937 rng; /* RandomGenerator(12345) */
};

orc_hnsw *orc_hnsw_new(int M) {
	orc_hnsw *h = (orc_hnsw *)calloc(1, sizeof *h);
	h->M = M;
	h->entry_point = -1;
	h->max_level = -1;
	h->efConstruction = 40;
	h->efSearch = 16;
	orc_mt_seed(&h->rng, 12345);
	/* set_default_probas(M, 1 / log(M)) */
	const double mult = 1.0 / log((double)M);
	int nn = 0;
	h->assign_probas = (double *)malloc(64 * sizeof(double));
	h->cum_nn = (int *)malloc(65 * sizeof(int));
	h->cum_nn[0] = 0;
	for (int level = 0; level < 64; level++) {
		double proba = exp(-level / mult) * (1 - exp(-1 / mult));
		if (proba < 1e-9)
			break;
		h->assign_probas[h->nprob] = proba;
		nn += level == 0 ? M * 2 : M;
		h->cum_nn[++h->nprob] = nn;
	}
	h->offsets = (int64_t *)malloc(sizeof(int64_t));
	h->offsets[0] = 0;
	return h;
}
void orc_hnsw_free(orc_hnsw *h) {
	if (!h)
		return;
	free(h->assign_probas);
	free(h->cum_nn);
	free(h->levels);
	free(h->offsets);
	free(h->neighbors);
	free(h);
}
void orc_hnsw_set_ef_construction(orc_hnsw *h, int v) {
	h->efConstruction = v;
}
static inline int nb_neighbors(const orc_hnsw *h, int layer) {
	return h->cum_nn[layer + 1] - h->cum_nn[layer];
}
static inline void neighbor_range(const orc_hnsw *h, int64_t no, int layer, int64_t *b, int64_t *e) {
	const int64_t o = h->offsets[no];
	*b = o + h->cum_nn[layer];
	*e = o + h->cum_nn[layer + 1];
}
static int random_level(orc_hnsw *h) {
	double f = orc_mt_rand_float(&h->rng);
	for (int level = 0; level < h->nprob; level++) {
		if (f < h->assign_probas[level])
			return level;
		f -= h->assign_probas[level];
	}
	return h->nprob - 1;
}

/* distance computer over the flat storage; IP -> negated (NegativeDistanceComputer).
 *
 * Canonical arithmetic of the HNSW path (shared bit for bit with csrc/hnsw.hip, where one 64-lane wavefront reads a
 * row as coalesced float4 and reduces across lanes):
 *   acc[k mod 256] = fmaf(t, t, acc[k mod 256])  walking k upwards          (t = q[k]-y[k]; IP: fmaf(q[k], y[k], .))
 *   lane[l]        = (acc[4l] + acc[4l+1]) + (acc[4l+2] + acc[4l+3])        l = 0..63
 *   total          = adjacent-pair binary tree over lane[0..63]
 * FAISS's own fvec_L2sqr / fvec_inner_product use SIMD partial sums whose order is ISA dependent; any fixed order
 * is a valid restatement, distances agree with the k-ordered chain to a few ulp. */
typedef struct {
	const float *xb;
	int d, is_l2;
	const float *q;
} dc_t;
static inline float dc_q(const dc_t *c, int32_t i) {
	const float *y = c->xb + (int64_t)i * c->d;
	const float *q = c->q;
	const int d = c->d;
	/* lanes that hold data: nl = next power of two >= ceil(min(d,256)/4); the remaining lanes of the 64-lane tree
	 * are +0.0 and x + 0.0 == x, so the tree over nl lanes IS the tree over 64 */
	int nl = 1;
	while (nl * 4 < d && nl < 64)
		nl *= 2;
	const int na = nl * 4;
	float acc[256];
	for (int j = 0; j < na; j++)
		acc[j] = 0.f;
	int k0 = 0;
	if (c->is_l2) {
		for (; k0 + 256 <= d; k0 += 256)
			for (int j = 0; j < 256; j++) {
				float t = q[k0 + j] - y[k0 + j];
				acc[j] = fmaf(t, t, acc[j]);
			}
		for (int j = 0; k0 + j < d; j++) {
			float t = q[k0 + j] - y[k0 + j];
			acc[j] = fmaf(t, t, acc[j]);
		}
	} else {
		for (; k0 + 256 <= d; k0 += 256)
			for (int j = 0; j < 256; j++)
				acc[j] = fmaf(q[k0 + j], y[k0 + j], acc[j]);
		for (int j = 0; k0 + j < d; j++)
			acc[j] = fmaf(q[k0 + j], y[k0 + j], acc[j]);
	}
	float lane[64];
	for (int l = 0; l < nl; l++)
		lane[l] = (acc[4 * l] + acc[4 * l + 1]) + (acc[4 * l + 2] + acc[4 * l + 3]);
	for (int m = nl / 2; m >= 1; m >>= 1)
		for (int g = 0; g < m; g++)
			lane[g] = lane[2 * g] + lane[2 * g + 1];
	return c->is_l2 ? lane[0] : -lane[0];
}
static inline float dc_sym(const dc_t *c, int32_t i, int32_t j) {
	dc_t t = *c;
	t.q = c->xb + (int64_t)j * c->d; /* symmetric_dis(i, j) = dis(b_j as query, b_i) */
	return dc_q(&t, i);
}

/* ---------------------------------------------------------------- construction */

static void greedy_update_nearest(const orc_hnsw *h, const dc_t *dc, int level, int32_t *nearest, float *d_nearest) {
	for (;;) {
		int32_t prev = *nearest;
		int64_t b, e;
		neighbor_range(h, *nearest, level, &b, &e);
		for (int64_t i = b; i < e; i++) {
			int32_t v = h->neighbors[i];
			if (v < 0)
				break;
			float dis = dc_q(dc, v);
			if (dis < *d_nearest) {
				*nearest = v;
				*d_nearest = dis;
			}
		}
		if (*nearest == prev)
			return;
	}
}

/* shrink_neighbor_list on a farther-top result set (only if it holds >= max_size entries) */
static void shrink_neighbor_list(const dc_t *dc, pq_t *results, int max_size) {
	if (results->n < max_size)
		return;
	pq_t closest;
	pq_init(&closest, 0);
	while (results->n > 0) {
		nd_t t = pq_top(results);
		pq_push(&closest, t.d, t.id);
		pq_pop(results);
	}
	nd_t *out = (nd_t *)malloc((size_t)max_size * sizeof(nd_t));
	int nout = 0;
	while (closest.n > 0) {
		nd_t v1 = pq_top(&closest);
		pq_pop(&closest);
		int good = 1;
		for (int j = 0; j < nout; j++) {
			float d12 = dc_sym(dc, out[j].id, v1.id);
			if (d12 < v1.d) {
				good = 0;
				break;
			}
		}
		if (good) {
			out[nout++] = v1;
			if (nout >= max_size)
				break;
		}
	}
	for (int j = 0; j < nout; j++)
		pq_push(results, out[j].d, out[j].id);
	free(out);
	pq_free(&closest);
}

static void add_link(orc_hnsw *h, const dc_t *dc, int32_t src, int32_t dest, int level) {
	int64_t b, e;
	neighbor_range(h, src, level, &b, &e);
	if (h->neighbors[e - 1] == -1) { /* room left: first free slot */
		int64_t i = e;
		while (i > b) {
			if (h->neighbors[i - 1] != -1)
				break;
			i--;
		}
		h->neighbors[i] = dest;
		return;
	}
	pq_t rs;
	pq_init(&rs, 1);
	pq_push(&rs, dc_sym(dc, src, dest), dest);
	for (int64_t i = b; i < e; i++)
		pq_push(&rs, dc_sym(dc, src, h->neighbors[i]), h->neighbors[i]);
	shrink_neighbor_list(dc, &rs, (int)(e - b));
	int64_t i = b;
	while (rs.n) {
		h->neighbors[i++] = pq_top(&rs).id;
		pq_pop(&rs);
	}
	while (i < e)
		h->neighbors[i++] = -1;
	pq_free(&rs);
}

static void search_neighbors_to_add(const orc_hnsw *h, const dc_t *dc, pq_t *results, int32_t entry, float d_entry,
                                    int level, uint8_t *visited) {
	pq_t cand;
	pq_init(&cand, 0);
	pq_push(&cand, d_entry, entry);
	pq_push(results, d_entry, entry);
	visited[entry] = 1;
	/* visited marks are undone at the end (VisitedTable::advance) */
	int32_t *touched = (int32_t *)malloc(64 * sizeof(int32_t));
	int ntouched = 0, captouched = 64;
	touched[ntouched++] = entry;
	while (cand.n > 0) {
		nd_t cur = pq_top(&cand);
		if (cur.d > pq_top(results).d)
			break;
		pq_pop(&cand);
		int64_t b, e;
		neighbor_range(h, cur.id, level, &b, &e);
		for (int64_t i = b; i < e; i++) {
			int32_t v = h->neighbors[i];
			if (v < 0)
				break;
			if (visited[v])
				continue;
			visited[v] = 1;
			if (ntouched == captouched) {
				captouched *= 2;
				touched = (int32_t *)realloc(touched, (size_t)captouched * sizeof(int32_t));
			}
			touched[ntouched++] = v;
			float dis = dc_q(dc, v);
			if (results->n < h->efConstruction || pq_top(results).d > dis) {
				pq_push(results, dis, v);
				pq_push(&cand, dis, v);
				if (results->n > h->efConstruction)
					pq_pop(results);
			}
		}
	}
	for (int i = 0; i < ntouched; i++)
		visited[touched[i]] = 0;
	free(touched);
	pq_free(&cand);
}

static void add_links_starting_from(orc_hnsw *h, const dc_t *dc, int32_t pt_id, int32_t nearest, float d_nearest,
                                    int level, uint8_t *visited) {
	pq_t link_targets;
	pq_init(&link_targets, 1);
	search_neighbors_to_add(h, dc, &link_targets, nearest, d_nearest, level, visited);
	const int M = nb_neighbors(h, level);
	shrink_neighbor_list(dc, &link_targets, M);
	int n_add = link_targets.n;
	int32_t *to_add = (int32_t *)malloc((size_t)(n_add > 0 ? n_add : 1) * sizeof(int32_t));
	int na = 0;
	while (link_targets.n) {
		int32_t other = pq_top(&link_targets).id;
		add_link(h, dc, pt_id, other, level);
		to_add[na++] = other;
		pq_pop(&link_targets);
	}
	for (int i = 0; i < na; i++)
		add_link(h, dc, to_add[i], pt_id, level);
	free(to_add);
	pq_free(&link_targets);
}

static void add_point(orc_hnsw *h, dc_t *dc, int pt_level, int32_t pt_id, uint8_t *visited) {
	int32_t nearest = h->entry_point;
	if (nearest == -1) {
		h->max_level = pt_level;
		h->entry_point = pt_id;
		return;
	}
	int level = h->max_level;
	float d_nearest = dc_q(dc, nearest);
	for (; level > pt_level; level--)
		greedy_update_nearest(h, dc, level, &nearest, &d_nearest);
	for (; level >= 0; level--)
		add_links_starting_from(h, dc, pt_id, nearest, d_nearest, level, visited);
	if (pt_level > h->max_level) {
		h->max_level = pt_level;
		h->entry_point = pt_id;
	}
}

/* hnsw_add_vertices (IndexHNSW.cpp), single thread.  xb = the flat storage AFTER the new rows were appended. */
void orc_hnsw_add(orc_hnsw *h, int64_t n0, int64_t n, const float *xb, int d, int is_l2) {
	if (n == 0)
		return;
	const int64_t ntotal = n0 + n;
	/* prepare_level_tab */
	if (ntotal > h->cap_n) {
		h->cap_n = ntotal * 2;
		h->levels = (int *)realloc(h->levels, (size_t)h->cap_n * sizeof(int));
		h->offsets = (int64_t *)realloc(h->offsets, (size_t)(h->cap_n + 1) * sizeof(int64_t));
	}
	for (int64_t i = 0; i < n; i++)
		h->levels[n0 + i] = random_level(h) + 1;
	int max_level = 0;
	for (int64_t i = 0; i < n; i++) {
		int pt_level = h->levels[n0 + i] - 1;
		if (pt_level > max_level)
			max_level = pt_level;
		h->offsets[n0 + i + 1] = h->offsets[n0 + i] + h->cum_nn[pt_level + 1];
	}
	if (h->offsets[ntotal] > h->cap_nb) {
		h->cap_nb = h->offsets[ntotal] * 2;
		h->neighbors = (int32_t *)realloc(h->neighbors, (size_t)h->cap_nb * sizeof(int32_t));
	}
	for (int64_t i = h->offsets[n0]; i < h->offsets[ntotal]; i++)
		h->neighbors[i] = -1;
	h->n = ntotal;
	/* bucket sort by level, then per level (highest first) a shuffle with rng2(789) */
	int nlev = max_level + 1;
	int *hist = (int *)calloc((size_t)nlev, sizeof(int));
	for (int64_t i = 0; i < n; i++)
		hist[h->levels[n0 + i] - 1]++;
	int *off = (int *)calloc((size_t)nlev + 1, sizeof(int));
	for (int l = 0; l < nlev; l++)
		off[l + 1] = off[l] + hist[l];
	int32_t *order = (int32_t *)malloc((size_t)n * sizeof(int32_t));
	for (int64_t i = 0; i < n; i++) {
		int pt_level = h->levels[n0 + i] - 1;
		order[off[pt_level]++] = (int32_t)(n0 + i);
	}
	uint8_t *visited = (uint8_t *)calloc((size_t)ntotal, 1);
	orc_mt19937 rng2;
	orc_mt_seed(&rng2, 789);
	dc_t dc = {xb, d, is_l2, NULL};
	int i1 = (int)n;
	for (int pt_level = nlev - 1; pt_level >= 0; pt_level--) {
		int i0 = i1 - hist[pt_level];
		for (int j = i0; j < i1; j++) {
			int j2 = j + orc_mt_rand_int(&rng2, i1 - j);
			int32_t t = order[j];
			order[j] = order[j2];
			order[j2] = t;
		}
		for (int i = i0; i < i1; i++) {
			int32_t pt_id = order[i];
			dc.q = xb + (int64_t)pt_id * d;
			add_point(h, &dc, pt_level, pt_id, visited);
		}
		i1 = i0;
	}
	free(visited);
	free(order);
	free(off);
	free(hist);
}

/* ---------------------------------------------------------------- search */

/* MinimaxHeap: max-heap on (dis, id) with capacity n; pop_min leaves a tombstone (id = -1) that keeps its distance
 * and its slot; push on a full heap evicts the root (which may be a tombstone).  Heap primitives = Heap.h
 * heap_push / heap_pop with CMax::cmp2. */
typedef struct {
	int n, k, nvalid;
	float *dis;
	int32_t *ids;
} mmh_t;
static inline int cmax2(float a1, float b1, int32_t a2, int32_t b2) {
	return (a1 > b1) || (a1 == b1 && a2 > b2);
}
static void mm_heap_push(int k, float *v0, int32_t *i0, float val, int32_t id) {
	float *v = v0 - 1;
	int32_t *ix = i0 - 1;
	int i = k;
	while (i > 1) {
		int f = i >> 1;
		if (!cmax2(val, v[f], id, ix[f]))
			break;
		v[i] = v[f];
		ix[i] = ix[f];
		i = f;
	}
	v[i] = val;
	ix[i] = id;
}
static void mm_heap_pop(int k, float *v0, int32_t *i0) {
	float *v = v0 - 1;
	int32_t *ix = i0 - 1;
	float val = v[k];
	int32_t id = ix[k];
	int i = 1;
	for (;;) {
		int i1 = i << 1, i2 = i1 + 1;
		if (i1 > k)
			break;
		if (i2 == k + 1 || cmax2(v[i1], v[i2], ix[i1], ix[i2])) {
			if (cmax2(val, v[i1], id, ix[i1]))
				break;
			v[i] = v[i1];
			ix[i] = ix[i1];
			i = i1;
		} else {
			if (cmax2(val, v[i2], id, ix[i2]))
				break;
			v[i] = v[i2];
			ix[i] = ix[i2];
			i = i2;
		}
	}
	v[i] = v[k];
	ix[i] = ix[k];
}
static void mm_push(mmh_t *m, int32_t i, float v) {
	if (m->k == m->n) {
		if (v >= m->dis[0])
			return;
		if (m->ids[0] != -1)
			--m->nvalid;
		mm_heap_pop(m->k--, m->dis, m->ids);
	}
	mm_heap_push(++m->k, m->dis, m->ids, v, i);
	++m->nvalid;
}
/* Which of several EQUAL minima pop_min takes.  0 (default) = FAISS: the heap ARRAY is scanned from the back and only a strictly
 * smaller distance replaces the current pick, so the winner is whichever equal entry sits last in the array -- a function of the
 * heap's memory layout (faiss/impl/HNSW.cpp MinimaxHeap::pop_min).  1 = the smallest id among the equal minima: the rule of the
 * device walk (csrc/hnsw.hip keeps its candidates as (distance, id)-sorted arrays in registers).  The two differ only when two
 * candidates are at bit-equal distance; bench.py's C5 line counts on how many queries that changes anything. */
static int g_pop_min_rule = 0;
void orc_hnsw_set_pop_min_rule(int rule) {
	g_pop_min_rule = rule;
}
int orc_hnsw_get_pop_min_rule(void) {
	return g_pop_min_rule;
}
static int32_t mm_pop_min(mmh_t *m, float *vmin_out) {
	int i = m->k - 1;
	while (i >= 0) {
		if (m->ids[i] != -1)
			break;
		i--;
	}
	if (i == -1)
		return -1;
	int imin = i;
	float vmin = m->dis[i];
	i--;
	while (i >= 0) {
		if (m->ids[i] != -1 && (m->dis[i] < vmin || (g_pop_min_rule == 1 && m->dis[i] == vmin && m->ids[i] < m->ids[imin]))) {
			vmin = m->dis[i];
			imin = i;
		}
		i--;
	}
	*vmin_out = vmin;
	int32_t ret = m->ids[imin];
	m->ids[imin] = -1;
	--m->nvalid;
	return ret;
}
static int mm_count_below(const mmh_t *m, float thresh) {
	int nb = 0;
	for (int i = 0; i < m->k; i++)
		if (m->dis[i] < thresh)
			nb++;
	return nb;
}

/* result heap = CMax k-heap of orc_core (strict insert); selector applies to results only */
void orc_hnsw_search_one(const orc_hnsw *h, const float *xb, int d, int is_l2, const float *q, int64_t k, int efSearch,
                         float *hv, int64_t *hi, uint8_t *visited, const orc_sel *sel, const int64_t *id_map) {
	orc_heap_init(k, hv, hi, 1);
	if (h->entry_point == -1) { /* HNSW::search returns at once; IndexHNSW::search still flips the sign for IP */
		orc_heap_reorder(k, hv, hi, 1);
		if (!is_l2)
			for (int64_t j = 0; j < k; j++)
				hv[j] = -hv[j];
		return;
	}
	dc_t dc = {xb, d, is_l2, q};
	int32_t nearest = h->entry_point;
	float d_nearest = dc_q(&dc, nearest);
	for (int level = h->max_level; level >= 1; level--)
		greedy_update_nearest(h, &dc, level, &nearest, &d_nearest);
	const int ef = efSearch > k ? efSearch : (int)k;
	mmh_t cand;
	cand.n = ef;
	cand.k = cand.nvalid = 0;
	cand.dis = (float *)malloc((size_t)ef * sizeof(float));
	cand.ids = (int32_t *)malloc((size_t)ef * sizeof(int32_t));
	mm_push(&cand, nearest, d_nearest);
	int32_t *touched = (int32_t *)malloc(1024 * sizeof(int32_t));
	int ntouched = 0, captouched = 1024;
	/* search_from_candidates(level 0) */
	float threshold = hv[0];
	for (int i = 0; i < cand.k; i++) {
		int32_t v1 = cand.ids[i];
		float dd = cand.dis[i];
		if (!sel || orc_sel_member(sel, id_map ? id_map[v1] : v1)) {
			if (dd < threshold) {
				orc_heap_replace_top(k, hv, hi, 1, dd, v1);
				threshold = hv[0];
			}
		}
		visited[v1] = 1;
		touched[ntouched++] = v1;
	}
	while (cand.nvalid > 0) {
		float d0 = 0;
		int32_t v0 = mm_pop_min(&cand, &d0);
		/* check_relative_distance: stop when efSearch stored distances (popped ones included) are below d0 */
		if (mm_count_below(&cand, d0) >= efSearch)
			break;
		int64_t b, e;
		neighbor_range(h, v0, 0, &b, &e);
		for (int64_t j = b; j < e; j++) {
			int32_t v1 = h->neighbors[j];
			if (v1 < 0)
				break;
			if (visited[v1])
				continue;
			visited[v1] = 1;
			if (ntouched == captouched) {
				captouched *= 2;
				touched = (int32_t *)realloc(touched, (size_t)captouched * sizeof(int32_t));
			}
			touched[ntouched++] = v1;
			float dd = dc_q(&dc, v1);
			if (!sel || orc_sel_member(sel, id_map ? id_map[v1] : v1)) {
				if (dd < threshold) {
					orc_heap_replace_top(k, hv, hi, 1, dd, v1);
					threshold = hv[0];
				}
			}
			mm_push(&cand, v1, dd);
		}
	}
	for (int i = 0; i < ntouched; i++)
		visited[touched[i]] = 0;
	free(touched);
	free(cand.dis);
	free(cand.ids);
	orc_heap_reorder(k, hv, hi, 1);
	if (!is_l2) /* IndexHNSW::search: "we need to revert the negated distances" */
		for (int64_t j = 0; j < k; j++)
			hv[j] = -hv[j];
}

int64_t orc_hnsw_n(const orc_hnsw *h) {
	return h->n;
}
int orc_hnsw_max_level(const orc_hnsw *h) {
	return h->max_level;
}
int32_t orc_hnsw_entry_point(const orc_hnsw *h) {
	return h->entry_point;
}
/* flat export for parity tests: levels[n], offsets[n+1], neighbors[offsets[n]] */
void orc_hnsw_export(const orc_hnsw *h, int *levels, int64_t *offsets, int32_t *neighbors) {
	memcpy(levels, h->levels, (size_t)h->n * sizeof(int));
	memcpy(offsets, h->offsets, (size_t)(h->n + 1) * sizeof(int64_t));
	memcpy(neighbors, h->neighbors, (size_t)h->offsets[h->n] * sizeof(int32_t));
}
int64_t orc_hnsw_nb_total(const orc_hnsw *h) {
	return h->offsets[h->n];
}

/* adopt a graph built elsewhere (the device index): lets the oracle walk the SAME graph for search parity and for the
 * cpu_baseline leg at sizes where a single-thread oracle build would take hours */
void orc_hnsw_import(orc_hnsw *h, int64_t n, const int *levels, const int64_t *offsets, const int32_t *neighbors,
                     int32_t entry_point, int max_level) {
	free(h->levels);
	free(h->offsets);
	free(h->neighbors);
	h->levels = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
	h->offsets = (int64_t *)malloc((size_t)(n + 1) * sizeof(int64_t));
	h->neighbors = (int32_t *)malloc((size_t)(offsets[n] > 0 ? offsets[n] : 1) * sizeof(int32_t));
	memcpy(h->levels, levels, (size_t)n * sizeof(int));
	memcpy(h->offsets, offsets, (size_t)(n + 1) * sizeof(int64_t));
	memcpy(h->neighbors, neighbors, (size_t)offsets[n] * sizeof(int32_t));
	h->n = h->cap_n = n;
	h->cap_nb = offsets[n];
	h->entry_point = entry_point;
	h->max_level = max_level;
}
