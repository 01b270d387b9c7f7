// How fast can 1 MB chunks of PAGEABLE host memory (valid only during the call) reach the device?  Variants of FlatIndex::add's
// staging (csrc/index.hip): memcpy into a pinned ring + hipMemcpyAsync; hipMemcpyAsync straight from the pageable chunk; the
// staging memcpy split over helper threads; non-temporal stores.  hipcc -O3 -o h2d_chunks h2d_chunks.cpp -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x)                                                                                                                     \
	do {                                                                                                                          \
		hipError_t e_ = (x);                                                                                                      \
		if (e_ != hipSuccess) {                                                                                                   \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                               \
			return 1;                                                                                                             \
		}                                                                                                                         \
	} while (0)
static void nt_copy(void *dst, const void *src, size_t bytes) { // 32-byte non-temporal stores (dst 32-byte aligned)
	const __m256i *s = (const __m256i *)src;
	__m256i *d = (__m256i *)dst;
	for (size_t i = 0; i < bytes / 32; ++i)
		_mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
	_mm_sfence();
}
struct Pool { // helpers spin on a generation counter (the caller is one of the copiers)
	std::vector<std::thread> th;
	std::atomic<int> gen {0}, done {0};
	std::atomic<bool> stop {false};
	char *dst = nullptr;
	const char *src = nullptr;
	size_t bytes = 0;
	int parts = 1;
	bool nt = false;
	void work(int p) {
		size_t per = (bytes / parts + 63) & ~(size_t)63, a = per * p, b = a + per > bytes ? bytes : a + per;
		if (a < b) {
			if (nt)
				nt_copy(dst + a, src + a, b - a);
			else
				memcpy(dst + a, src + a, b - a);
		}
	}
	void start(int helpers) {
		for (int h = 0; h < helpers; ++h)
			th.emplace_back([this, h] {
				int seen = 0;
				for (;;) {
					while (gen.load(std::memory_order_acquire) == seen && !stop.load())
						_mm_pause();
					if (stop.load())
						return;
					seen = gen.load();
					work(h + 1);
					done.fetch_add(1, std::memory_order_release);
				}
			});
	}
	void copy(char *d, const char *s, size_t n, bool use_nt) {
		dst = d, src = s, bytes = n, nt = use_nt, parts = (int)th.size() + 1;
		done.store(0);
		gen.fetch_add(1, std::memory_order_release);
		work(0);
		while (done.load(std::memory_order_acquire) < (int)th.size())
			_mm_pause();
	}
	~Pool() {
		stop.store(true);
		for (auto &t : th)
			t.join();
	}
};
int main(int argc, char **argv) {
	const size_t chunk = argc > 1 ? (size_t)atol(argv[1]) : (1u << 20), total = (size_t)1 << 30, nch = total / chunk;
	std::vector<char> src(total);
	for (size_t i = 0; i < total; i += 4096)
		src[i] = (char)i;
	char *dev = nullptr;
	CK(hipMalloc((void **)&dev, total));
	hipStream_t st;
	CK(hipStreamCreate(&st));
	const int NB = 8;
	char *pin[NB];
	hipEvent_t ev[NB];
	for (int i = 0; i < NB; ++i) {
		CK(hipHostMalloc((void **)&pin[i], chunk, hipHostMallocDefault));
		CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
	}
	auto run = [&](const char *name, auto &&body) {
		for (int rep = 0; rep < 2; ++rep) {
			const auto t0 = std::chrono::steady_clock::now();
			for (size_t c = 0; c < nch; ++c)
				body(c);
			(void)hipStreamSynchronize(st);
			const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (rep)
				printf("%-58s %7.2f GB/s  %7.1f us per %zu KB chunk\n", name, total / s / 1e9, s / nch * 1e6, chunk >> 10);
		}
	};
	run("memcpy -> pinned ring -> hipMemcpyAsync (FlatIndex::add)", [&](size_t c) {
		const int i = (int)(c % NB);
		(void)hipEventSynchronize(ev[i]);
		memcpy(pin[i], src.data() + c * chunk, chunk);
		(void)hipMemcpyAsync(dev + c * chunk, pin[i], chunk, hipMemcpyHostToDevice, st);
		(void)hipEventRecord(ev[i], st);
	});
	run("non-temporal copy -> pinned ring -> hipMemcpyAsync", [&](size_t c) {
		const int i = (int)(c % NB);
		(void)hipEventSynchronize(ev[i]);
		nt_copy(pin[i], src.data() + c * chunk, chunk);
		(void)hipMemcpyAsync(dev + c * chunk, pin[i], chunk, hipMemcpyHostToDevice, st);
		(void)hipEventRecord(ev[i], st);
	});
	run("hipMemcpyAsync straight from the pageable chunk", [&](size_t c) { (void)hipMemcpyAsync(dev + c * chunk, src.data() + c * chunk, chunk, hipMemcpyHostToDevice, st); });
	run("hipMemcpy (synchronous) from the pageable chunk", [&](size_t c) { (void)hipMemcpy(dev + c * chunk, src.data() + c * chunk, chunk, hipMemcpyHostToDevice); });
	run("memcpy only (no device copy)", [&](size_t c) { memcpy(pin[c % NB], src.data() + c * chunk, chunk); });
	run("non-temporal copy only", [&](size_t c) { nt_copy(pin[c % NB], src.data() + c * chunk, chunk); });
	run("hipMemcpyAsync from pinned only (no staging copy)", [&](size_t c) { (void)hipMemcpyAsync(dev + c * chunk, pin[c % NB], chunk, hipMemcpyHostToDevice, st); });
	for (int helpers : {1, 3}) {
		Pool pool;
		pool.start(helpers);
		char nm[128];
		for (int nt = 0; nt < 2; ++nt) {
			snprintf(nm, sizeof nm, "%s split over %d threads -> pinned -> hipMemcpyAsync", nt ? "non-temporal copy" : "memcpy", helpers + 1);
			run(nm, [&](size_t c) {
				const int i = (int)(c % NB);
				(void)hipEventSynchronize(ev[i]);
				pool.copy(pin[i], src.data() + c * chunk, chunk, nt);
				(void)hipMemcpyAsync(dev + c * chunk, pin[i], chunk, hipMemcpyHostToDevice, st);
				(void)hipEventRecord(ev[i], st);
			});
		}
	}
	return 0;
}
