#pragma once
namespace faiss {
struct HNSW {
	int efConstruction = 40; // set by the glue: src/faiss_extension.cpp:136-139
	int efSearch = 16;
};
} // namespace faiss
