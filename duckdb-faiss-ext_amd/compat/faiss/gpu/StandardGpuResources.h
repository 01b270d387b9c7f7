#pragma once
namespace faiss {
namespace gpu {
struct GpuResourcesProvider {
	virtual ~GpuResourcesProvider() {
	}
};
// src/gpu/gpu.cpp:45 -- streams / pinned staging live inside each device index on the MI355X path
struct StandardGpuResources : GpuResourcesProvider {};
} // namespace gpu
} // namespace faiss
