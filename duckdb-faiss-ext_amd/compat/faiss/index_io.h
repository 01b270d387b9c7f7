#pragma once
#include "Index.h"
namespace faiss {
void write_index(const Index *idx, const char *fname); // src/faiss_extension.cpp:199
Index *read_index(const char *fname, int io_flags = 0); // :234
} // namespace faiss
