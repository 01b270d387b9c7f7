#pragma once
#include "Index.h"
namespace faiss {
struct IndexFlat : Index {};
struct IndexFlatL2 : IndexFlat {};
struct IndexFlatIP : IndexFlat {};
} // namespace faiss
