// csrc/ivf.hip -- IndexIVFFlat on device (placeholder until the segmented list-scan kernel lands).
#include "index.h"
namespace mvs {
IndexBase *make_ivf_index(int, const std::string &desc, int) {
	if (desc.rfind("IVF", 0) == 0)
		throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
		            "This index type is not implemented on the MI355X path yet: %s", desc.c_str());
	return nullptr;
}
IndexBase *ivf_quantizer_of(IndexBase *) {
	return nullptr;
}
} // namespace mvs
