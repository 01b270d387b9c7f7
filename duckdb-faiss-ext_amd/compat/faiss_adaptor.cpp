// compat/faiss_adaptor.cpp -- the faiss:: classes of compat/faiss/*.h implemented over the C ABI
// (include/mi355_faiss.h).  Built as libfaiss_mi355.so: the library the reference's CMake would link in place of
// `add_subdirectory(faiss)` / target `faiss` (/root/reference/CMakeLists.txt:58,67-71).
#include "faiss/Index.h"
#include "faiss/IndexFlat.h"
#include "faiss/IndexHNSW.h"
#include "faiss/IndexIDMap.h"
#include "faiss/IndexIVF.h"
#include "faiss/gpu/GpuCloner.h"
#include "faiss/gpu/GpuIndexIVF.h"
#include "faiss/index_factory.h"
#include "faiss/index_io.h"

#include <cstring>

namespace faiss {

void throw_last_error() {
	throw FaissException(mvs_last_error());
}

void Index::refresh() {
	d = mvs_index_d(handle);
	ntotal = mvs_index_ntotal(handle);
	is_trained = mvs_index_is_trained(handle) != 0;
	metric_type = (MetricType)mvs_index_metric_type(handle);
}
Index::~Index() {
	if (handle && owns_handle)
		mvs_index_free(handle);
}
void Index::train(idx_t n, const float *x) {
	if (mvs_index_train(handle, n, x))
		throw_last_error();
	refresh();
}
void Index::add(idx_t n, const float *x) {
	before_add();
	if (mvs_index_add(handle, n, x)) {
		refresh();
		throw_last_error();
	}
	refresh();
}
void Index::add_with_ids(idx_t n, const float *x, const idx_t *xids) {
	before_add();
	if (mvs_index_add_with_ids(handle, n, x, xids)) {
		refresh();
		throw_last_error();
	}
	refresh();
}

// innerCreateSearchParameters (src/faiss_extension.cpp:668-721) hands us SearchParameters / SearchParametersIVF /
// SearchParametersHNSW with an optional IDSelectorBitmap / IDSelectorBatch
void fill_params(const Index *, const SearchParameters *params, mvs_search_params *out) {
	memset(out, 0, sizeof *out);
	if (!params)
		return;
	if (auto ivf = dynamic_cast<const SearchParametersIVF *>(params)) {
		out->nprobe = (int64_t)ivf->nprobe;
		// IVF<n>_HNSW<m>: the glue hangs the coarse quantizer's SearchParametersHNSW here (:679-681)
		if (auto qh = dynamic_cast<const SearchParametersHNSW *>(ivf->quantizer_params))
			out->efSearch = qh->efSearch;
	}
	if (auto hnsw = dynamic_cast<const SearchParametersHNSW *>(params))
		out->efSearch = hnsw->efSearch;
	if (params->sel) {
		if (auto bm = dynamic_cast<const IDSelectorBitmap *>(params->sel)) {
			out->sel_kind = MVS_SEL_BITMAP;
			out->sel_data = bm->bitmap;
			out->sel_n = (int64_t)bm->n;
		} else if (auto bt = dynamic_cast<const IDSelectorBatch *>(params->sel)) {
			out->sel_kind = MVS_SEL_BATCH;
			out->sel_data = bt->ids.data();
			out->sel_n = (int64_t)bt->ids.size();
		} else {
			throw FaissException("Error in faiss::Index::search: this IDSelector type is not implemented on the MI355X path");
		}
	}
}
void Index::search(idx_t n, const float *x, idx_t k, float *distances, idx_t *labels,
                   const SearchParameters *params) const {
	mvs_search_params p;
	fill_params(this, params, &p);
	if (mvs_index_search(handle, n, x, k, distances, labels, &p))
		throw_last_error();
}

IndexIDMap::~IndexIDMap() {
	delete index;
}
void IndexIDMap::before_add() {
	if (index)
		index->before_add();
}
IndexIVF::~IndexIVF() {
	delete quantizer;
}
void IndexHNSW::before_add() {
	(void)mvs_index_hnsw_set_ef_construction(handle, hnsw.efConstruction);
}

Index *Index::wrap(mvs_index *h, bool owned) {
	Index *ix = nullptr;
	switch (mvs_index_kind(h)) {
	case MVS_KIND_IDMAP: {
		auto *m = new IndexIDMap;
		m->handle = h;
		m->index = wrap(mvs_index_idmap_sub(h), false);
		ix = m;
		break;
	}
	case MVS_KIND_IVFFLAT: {
		auto *v = new IndexIVFFlat;
		v->handle = h;
		if (mvs_index *q = mvs_index_ivf_quantizer(h)) {
			v->quantizer = wrap(q, false);
			v->nlist = (size_t)mvs_index_ntotal(q);
		}
		ix = v;
		break;
	}
	case MVS_KIND_HNSW:
		ix = new IndexHNSWFlat;
		ix->handle = h;
		break;
	default:
		if (mvs_index_metric_type(h) == METRIC_L2)
			ix = new IndexFlatL2;
		else if (mvs_index_metric_type(h) == METRIC_INNER_PRODUCT)
			ix = new IndexFlatIP;
		else
			ix = new IndexFlat;
		ix->handle = h;
	}
	ix->owns_handle = owned;
	ix->refresh();
	return ix;
}

Index *index_factory(int d, const char *description, MetricType metric) {
	mvs_index *h = nullptr;
	if (mvs_index_factory(&h, d, description, (int)metric))
		throw_last_error();
	return Index::wrap(h, true);
}
void write_index(const Index *idx, const char *fname) {
	if (mvs_write_index(idx->handle, fname))
		throw_last_error();
}
Index *read_index(const char *fname, int) {
	mvs_index *h = nullptr;
	if (mvs_read_index(&h, fname))
		throw_last_error();
	return Index::wrap(h, true);
}

namespace gpu {
faiss::Index *index_cpu_to_gpu(GpuResourcesProvider *, int device, const faiss::Index *index) {
	mvs_index *h = nullptr;
	if (mvs_index_clone_to_gpu(&h, index->handle, device))
		throw_last_error();
	return Index::wrap(h, true);
}
} // namespace gpu

} // namespace faiss
