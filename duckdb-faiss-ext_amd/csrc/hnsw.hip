// csrc/hnsw.hip -- IndexHNSWFlat on device (placeholder until the graph-walk kernel lands).
#include "index.h"
namespace mvs {
IndexBase *make_hnsw_index(int, const std::string &desc, int) {
	if (desc.rfind("HNSW", 0) == 0)
		throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
		            "This index type is not implemented on the MI355X path yet: %s", desc.c_str());
	return nullptr;
}
bool hnsw_set_ef_construction(IndexBase *, int) {
	return false;
}
} // namespace mvs
