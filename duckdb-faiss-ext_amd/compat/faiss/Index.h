// compat/faiss/Index.h -- faiss::Index as the reference's glue sees it, implemented over the C ABI of
// libmi355faiss.so (include/mi355_faiss.h).  Data members d / ntotal / is_trained are read directly by the glue
// (src/faiss_extension.cpp:159,355,490,518) and are refreshed after every mutating call.
#pragma once
#include "MetricType.h"
#include "impl/FaissException.h"
#include "impl/IDSelector.h"

#include "../../../include/mi355_faiss.h"

#include <cstddef>
namespace faiss {

struct SearchParameters {
	IDSelector *sel = nullptr; // src/faiss_extension.cpp:678,694,719
	virtual ~SearchParameters() {
	}
};

struct Index {
	int d = 0;
	idx_t ntotal = 0;
	bool verbose = false;
	bool is_trained = true;
	MetricType metric_type = METRIC_L2;
	float metric_arg = 0;

	virtual ~Index();
	virtual void train(idx_t n, const float *x);                              // :396,:583
	virtual void add(idx_t n, const float *x);                                // :512,:609
	virtual void add_with_ids(idx_t n, const float *x, const idx_t *xids);    // :510,:607
	virtual void search(idx_t n, const float *x, idx_t k, float *distances, idx_t *labels,
	                    const SearchParameters *params = nullptr) const;      // :631

	// --- adaptor plumbing (not part of FAISS) ---
	mvs_index *handle = nullptr;
	bool owns_handle = true;
	void refresh();
	static Index *wrap(mvs_index *h, bool owned); // builds the IndexIDMap / IndexIVF / IndexHNSW / IndexFlat graph
	// pushes members the glue assigns directly on the wrapper (IndexHNSW::hnsw.efConstruction, :136-139) to the
	// device index before rows arrive; wrappers forward it down the chain
	virtual void before_add() {
	}
};

[[noreturn]] void throw_last_error();
void fill_params(const Index *index, const SearchParameters *params, mvs_search_params *out);

} // namespace faiss
