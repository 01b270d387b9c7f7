// csrc/index_io.hip -- faiss::write_index / read_index (src/faiss_extension.cpp:199,234); "next" row 8f-4.
#include "index.h"
namespace mvs {
void write_index_file(const IndexBase *, const char *) {
	throw_faiss("void faiss::write_index(const faiss::Index*, const char*)", "faiss/impl/index_write.cpp",
	            "write_index is not implemented on the MI355X path yet");
}
IndexBase *read_index_file(const char *) {
	throw_faiss("faiss::Index* faiss::read_index(const char*, int)", "faiss/impl/index_read.cpp",
	            "read_index is not implemented on the MI355X path yet");
}
} // namespace mvs
