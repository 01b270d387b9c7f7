// compat/faiss/IndexHNSW.h -- IndexHNSW::hnsw.efConstruction (src/faiss_extension.cpp:138), SearchParametersHNSW
// (:693-699); IndexPQ / SearchParametersPQ are dynamic_cast targets only (:704-706).
#pragma once
#include "Index.h"
#include "impl/HNSW.h"
namespace faiss {
struct SearchParametersHNSW : SearchParameters {
	int efSearch = 16;
	bool check_relative_distance = true;
	bool bounded_queue = true;
};
struct IndexHNSW : Index {
	HNSW hnsw;
	Index *storage = nullptr;
	void before_add() override; // pushes hnsw.efConstruction to the device index
};
struct IndexHNSWFlat : IndexHNSW {};
struct SearchParametersPQ : SearchParameters {};
struct IndexPQ : Index {};
} // namespace faiss
