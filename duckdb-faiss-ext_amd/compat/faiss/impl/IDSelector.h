// compat/faiss/impl/IDSelector.h -- IDSelectorBitmap (src/faiss_extension.cpp:959), IDSelectorBatch (:1008).
// The selector memory stays owned by the glue (mask_tmp / a local vector) and only has to outlive the search call;
// the device path copies it to HBM per call (csrc/index.hip SelectorHolder).
#pragma once
#include "../MetricType.h"

#include <cstddef>
#include <cstdint>
#include <vector>
namespace faiss {
struct IDSelector {
	virtual bool is_member(idx_t id) const = 0;
	virtual ~IDSelector() {
	}
};
struct IDSelectorBitmap : IDSelector {
	size_t n;
	const uint8_t *bitmap;
	IDSelectorBitmap(size_t n_, const uint8_t *bitmap_) : n(n_), bitmap(bitmap_) {
	}
	bool is_member(idx_t ii) const final {
		uint64_t i = (uint64_t)ii;
		if ((i >> 3) >= n)
			return false;
		return (bitmap[i >> 3] >> (i & 7)) & 1;
	}
};
struct IDSelectorBatch : IDSelector {
	std::vector<idx_t> ids; // FAISS keeps a bloom filter + hash set; membership is what matters
	IDSelectorBatch(size_t n, const idx_t *indices) : ids(indices, indices + n) {
	}
	bool is_member(idx_t id) const final {
		for (idx_t v : ids)
			if (v == id)
				return true;
		return false;
	}
};
} // namespace faiss
