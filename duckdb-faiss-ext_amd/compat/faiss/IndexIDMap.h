// compat/faiss/IndexIDMap.h -- the glue reads and re-assigns IndexIDMap::index (src/faiss_extension.cpp:127-130,671-674)
#pragma once
#include "Index.h"
namespace faiss {
struct IndexIDMap : Index {
	Index *index = nullptr; // borrowed view of the sub-index (owned by the device object)
	bool own_fields = false;
	~IndexIDMap() override;
	void before_add() override; // "IDMap,HNSW32" + map{'efConstruction':..}: the glue sets it on `index`, adds on the wrapper
};
} // namespace faiss
