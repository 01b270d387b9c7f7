// tools/micro/gather_bw.hip -- what does HBM deliver for the HNSW access pattern?  Every wavefront reads random
// whole rows (row_bytes each, coalesced float4 across the 64 lanes), G independent rows in flight, no dependence
// between iterations except the register accumulate.  Build: hipcc -O3 --offload-arch=gfx950 gather_bw.hip -o gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
	z += 0x9e3779b97f4a7c15ull;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return z ^ (z >> 31);
}

template <int NI, int G>
__global__ __launch_bounds__(64) void gather_kernel(const float4 *vecs, long long n, int dp4, int iters, float *out, int sequential) {
	const int lane = threadIdx.x;
	float acc = 0.f;
	for (int t = 0; t < iters; t++) {
		float4 y[G][NI];
#pragma unroll
		for (int g = 0; g < G; g++) {
			long long id;
			if (sequential)
				id = (((long long)blockIdx.x * iters + t) * G + g) % n;
			else
				id = (long long)(mix(((unsigned long long)blockIdx.x << 32) ^ ((unsigned long long)t * G + g)) % (unsigned long long)n);
			id = __builtin_amdgcn_readfirstlane((int)id);
#pragma unroll
			for (int i = 0; i < NI; i++) {
				const int idx = lane + 64 * i;
				y[g][i] = idx < dp4 ? vecs[id * dp4 + idx] : make_float4(0, 0, 0, 0);
			}
		}
#pragma unroll
		for (int g = 0; g < G; g++)
#pragma unroll
			for (int i = 0; i < NI; i++)
				acc += y[g][i].x + y[g][i].y + y[g][i].z + y[g][i].w;
	}
	if (acc == 12345.678f)
		out[blockIdx.x] = acc;
}

template <int NI, int G>
void run(const float4 *vecs, long long n, int dp4, int grid, float *out, int sequential) {
	const int iters = 4096 / G;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((gather_kernel<NI, G>), dim3(grid), dim3(64), 0, 0, vecs, n, dp4, 64, out, sequential);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL((gather_kernel<NI, G>), dim3(grid), dim3(64), 0, 0, vecs, n, dp4, iters, out, sequential);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double bytes = (double)grid * iters * G * dp4 * 16.0;
	printf("%s rows of %5d B, grid %5d waves, G=%d rows in flight: %8.1f GB/s (%.2f ms)\n", sequential ? "sequential" : "random    ",
	       dp4 * 16, grid, G, bytes / (ms * 1e-3) / 1e9, ms);
}

int main(int argc, char **argv) {
	const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
	const int d = argc > 2 ? atoi(argv[2]) : 768;
	const int dp4 = d / 4;
	float4 *vecs;
	float *out;
	CK(hipMalloc(&vecs, (size_t)n * dp4 * 16));
	CK(hipMemset(vecs, 0, (size_t)n * dp4 * 16));
	CK(hipMalloc(&out, 1 << 20));
	printf("table: %lld rows x %d floats = %.2f GB\n", n, d, (double)n * dp4 * 16 / 1e9);
	for (int grid : {1024, 2048, 4096, 8192}) {
		if (d <= 256) {
			run<1, 1>(vecs, n, dp4, grid, out, 0);
			run<1, 4>(vecs, n, dp4, grid, out, 0);
			run<1, 8>(vecs, n, dp4, grid, out, 0);
			run<1, 8>(vecs, n, dp4, grid, out, 1);
		} else if (d <= 768) {
			run<3, 1>(vecs, n, dp4, grid, out, 0);
			run<3, 2>(vecs, n, dp4, grid, out, 0);
			run<3, 4>(vecs, n, dp4, grid, out, 0);
			run<3, 8>(vecs, n, dp4, grid, out, 0);
			run<3, 4>(vecs, n, dp4, grid, out, 1);
		} else {
			run<6, 1>(vecs, n, dp4, grid, out, 0);
			run<6, 2>(vecs, n, dp4, grid, out, 0);
			run<6, 4>(vecs, n, dp4, grid, out, 0);
			run<6, 2>(vecs, n, dp4, grid, out, 1);
		}
	}
	return 0;
}
