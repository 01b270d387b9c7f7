/*
 * oracle/orc_internal.h -- CPU ORACLE (TEST INFRASTRUCTURE): primitives shared between orc_core.c and orc_hnsw.c.
 */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "orc.h"
#include <stdint.h>

typedef struct {
	uint32_t mt[624];
	int idx;
} orc_mt19937;
void orc_mt_seed(orc_mt19937 *r, uint32_t s);
int orc_mt_rand_int(orc_mt19937 *r, int max);
float orc_mt_rand_float(orc_mt19937 *r);

/* CMax (is_max = 1) / CMin k-best heaps, Heap.h semantics */
void orc_heap_init(int64_t k, float *hv, int64_t *hi, int is_max);
void orc_heap_replace_top(int64_t k, float *hv, int64_t *hi, int is_max, float v, int64_t id);
void orc_heap_reorder(int64_t k, float *hv, int64_t *hi, int is_max);

typedef struct {
	int kind;
	const uint8_t *bitmap;
	int64_t n; /* bytes (bitmap) or ids (batch) */
	int64_t *sorted; /* batch: sorted copy */
} orc_sel;
int orc_sel_member(const orc_sel *s, int64_t id);

typedef struct orc_hnsw orc_hnsw;
orc_hnsw *orc_hnsw_new(int M);
void orc_hnsw_free(orc_hnsw *h);
void orc_hnsw_set_ef_construction(orc_hnsw *h, int v);
void orc_hnsw_add(orc_hnsw *h, int64_t n0, int64_t n, const float *xb, int d, int is_l2);
void orc_hnsw_search_one(const orc_hnsw *h, const float *xb, int d, int is_l2, const float *q, int64_t k, int efSearch,
                         float *hv, int64_t *hi, uint8_t *visited, const orc_sel *sel, const int64_t *id_map);
int64_t orc_hnsw_n(const orc_hnsw *h);
int orc_hnsw_max_level(const orc_hnsw *h);
int32_t orc_hnsw_entry_point(const orc_hnsw *h);
int64_t orc_hnsw_nb_total(const orc_hnsw *h);
void orc_hnsw_import(orc_hnsw *h, int64_t n, const int *levels, const int64_t *offsets, const int32_t *neighbors,
                     int32_t entry_point, int max_level);
void orc_hnsw_export(const orc_hnsw *h, int *levels, int64_t *offsets, int32_t *neighbors);
#endif
