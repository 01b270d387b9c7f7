#pragma once
#include "../IndexIVF.h"
namespace faiss {
namespace gpu {
struct GpuIndexIVF : faiss::Index {}; // dynamic_cast target only (src/gpu/gpu.cpp:69)
} // namespace gpu
} // namespace faiss
