#pragma once
#include "../Index.h"
#include "StandardGpuResources.h"
namespace faiss {
namespace gpu {
// src/gpu/gpu.cpp:48.  Indexes are device-native already; this returns a new index object that lives on `device`
// (the glue replaces entry.index with it and drops the old one).
faiss::Index *index_cpu_to_gpu(GpuResourcesProvider *provider, int device, const faiss::Index *index);
} // namespace gpu
} // namespace faiss
