// compat/faiss/IndexIVF.h -- IndexIVF::quantizer (src/faiss_extension.cpp:680), SearchParametersIVF (:677-686)
#pragma once
#include "Index.h"
namespace faiss {
struct SearchParametersIVF : SearchParameters {
	size_t nprobe = 1;
	size_t max_codes = 0;
	SearchParameters *quantizer_params = nullptr;
};
struct IndexIVF : Index {
	Index *quantizer = nullptr; // borrowed view
	size_t nlist = 0;
	size_t nprobe = 1;
	~IndexIVF() override;
};
struct IndexIVFFlat : IndexIVF {};
} // namespace faiss
