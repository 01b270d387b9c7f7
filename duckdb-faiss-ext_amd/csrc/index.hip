// csrc/index.hip -- device-native index objects + the C ABI of include/mi355_faiss.h.
//
// Host side of the hot path.  Mirrors the FAISS object graph the reference's glue walks
// (/root/reference/src/faiss_extension.cpp:123-144 setIndexParameters, :668-721 innerCreateSearchParameters):
//   FlatIndex     <- faiss::IndexFlat{L2,IP}      rows in HBM, padded row-major + per-row squared norms
//   IDMapIndex    <- faiss::IndexIDMap            int64 id_map in HBM, translated in the merge kernel
//   IVFFlatIndex  <- faiss::IndexIVFFlat          (csrc/ivf.hip)
// Every index owns a HIP stream and a ring of pinned staging buffers: add() copies the caller's rows into
// pinned memory (the caller's DataChunk buffer is only valid during the call, :493-512) and returns while
// hipMemcpyAsync + the norm kernel run on the stream.
#include "index.h"

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>

#include <algorithm>
#include <dlfcn.h>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

namespace mvs {

static thread_local std::string g_last_error;

void throw_faiss(const char *func, const char *file, const char *fmt, ...) {
	char msg[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(msg, sizeof msg, fmt, ap);
	va_end(ap);
	// FAISS: "Error in <func> at <file>:<line>: <msg>" (impl/FaissException.cpp)
	const char *base = strrchr(file, '/');
	throw Error(std::string("Error in ") + func + " at " + (base ? base + 1 : file) + ": " + msg);
}

// ------------------------------------------------------------------------------------------ buffers

void DevBuf::reserve(size_t bytes) {
	if (bytes <= cap)
		return;
	if (p)
		MVS_HIP(hipFree(p));
	p = nullptr;
	cap = 0;
	size_t want = bytes + bytes / 4 + 256;
	MVS_HIP(hipMalloc(&p, want));
	cap = want;
}
void DevBuf::release() {
	if (p)
		(void)hipFree(p);
	p = nullptr;
	cap = 0;
}

PinnedRing::~PinnedRing() {
	for (int i = 0; i < NB; ++i) {
		if (buf[i])
			(void)hipHostFree(buf[i]);
		if (ev[i])
			(void)hipEventDestroy(ev[i]);
	}
}
void PinnedRing::drop_events() { // (the owner moves to another device: events are created again, there, by the next acquire)
	for (int i = 0; i < NB; ++i)
		if (ev[i]) {
			(void)hipEventSynchronize(ev[i]);
			(void)hipEventDestroy(ev[i]);
			ev[i] = nullptr;
		}
}
// The staging copy of add(): pageable rows -> a pinned slot that only the copy engine reads afterwards.  Non-temporal stores skip
// the read-for-ownership of the destination lines (tools/micro/h2d_chunks.cpp on the GPU box, 1 MB chunks: 23.9 us against memcpy's
// 35.5; with the hipMemcpyAsync behind it 30.0 against 26.4 GB/s).  dst is 16-byte aligned (slots are page aligned), src need not be.
static void stage_copy(void *dst, const void *src, size_t bytes) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
	if ((uintptr_t)dst & 15) { // (chunks of rows whose size is no multiple of 16 bytes follow each other unaligned: plain copy)
		memcpy(dst, src, bytes);
		return;
	}
	typedef float v4f __attribute__((ext_vector_type(4)));
	const size_t n16 = bytes / 16;
	const char *s = (const char *)src;
	v4f *d = (v4f *)dst;
	for (size_t i = 0; i < n16; ++i) {
		v4f v;
		memcpy(&v, s + 16 * i, 16);
		__builtin_nontemporal_store(v, d + i);
	}
	if (bytes & 15)
		memcpy((char *)dst + 16 * n16, s + 16 * n16, bytes & 15);
	__builtin_ia32_sfence(); // (the stores are weakly ordered: visible before the copy engine is told to read)
#else
	memcpy(dst, src, bytes);
#endif
}
// Round 6: a chunk of the glue's size (1 MB) is copied by THREE threads (MVS_STAGE_THREADS).  The call runs under the glue's faiss_lock and seven of eight
// calls are nothing but this copy (profiles/r6_ingest.txt: 36 us per MB with the other workers filling their next chunks, the
// ceiling of the call pattern); helper threads that take the other parts shorten exactly that.  A helper spins between the
// calls of an ingest burst (they follow each other within ~50 us) and goes to sleep 300 us after the last one; callers of different
// indexes do not queue for the helpers (try_lock: the loser copies alone).  MVS_STAGE_THREADS=1 keeps the copy on the calling thread.
namespace {
struct StageHelper {
	std::atomic<uint64_t> seq {0}, done {0};
	const void *src = nullptr;
	void *dst = nullptr;
	size_t bytes = 0;
	std::atomic<bool> sleeping {false};
	std::mutex mu;
	std::condition_variable cv;
	void run() {
		uint64_t seen = 0;
		for (;;) {
			const auto t0 = std::chrono::steady_clock::now();
			unsigned spins = 0;
			while (seq.load(std::memory_order_acquire) == seen) {
				__builtin_ia32_pause();
				if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
					std::unique_lock<std::mutex> lk(mu);
					sleeping.store(true, std::memory_order_release);
					cv.wait(lk, [&] { return seq.load(std::memory_order_acquire) != seen; });
					sleeping.store(false, std::memory_order_release);
				}
			}
			seen = seq.load(std::memory_order_acquire);
			stage_copy(dst, src, bytes);
			done.store(seen, std::memory_order_release);
		}
	}
	void wake() {
		if (sleeping.load(std::memory_order_acquire)) {
			{ std::lock_guard<std::mutex> lk(mu); }
			cv.notify_one();
		}
	}
};
} // namespace
// dst <- src with the helpers' hands where it pays (>= 256 KB, 16-byte aligned parts); MVS_STAGE_THREADS = copying threads in all (1 .. 4)
static void stage_copy_mt(void *dst, const void *src, size_t bytes) {
	static const int nthreads = []() {
		const char *e = getenv("MVS_STAGE_THREADS");
		const int v = e ? atoi(e) : 3;
		return v < 1 ? 1 : (v > 4 ? 4 : v);
	}();
	static std::mutex owner; // (callers of different indexes do not queue for the helpers: the loser copies alone)
	static StageHelper *helpers[3] = {nullptr, nullptr, nullptr};
	if (nthreads <= 1 || bytes < ((size_t)256 << 10) || ((uintptr_t)dst & 15) || !owner.try_lock()) {
		stage_copy(dst, src, bytes);
		return;
	}
	const int nh = nthreads - 1;
	const size_t part = (bytes / (size_t)nthreads) & ~(size_t)63;
	uint64_t job[3];
	for (int i = 0; i < nh; ++i) {
		if (!helpers[i]) {
			helpers[i] = new StageHelper; // (never destroyed: its thread may outlive static destruction)
			StageHelper *hp = helpers[i];
			std::thread([hp]() { hp->run(); }).detach();
		}
		StageHelper &h = *helpers[i];
		const size_t off = part * (size_t)(i + 1), len = i + 1 == nh ? bytes - off : part;
		h.src = (const char *)src + off, h.dst = (char *)dst + off, h.bytes = len;
		job[i] = h.seq.fetch_add(1, std::memory_order_acq_rel) + 1;
		h.wake();
	}
	stage_copy(dst, src, part);
	for (int i = 0; i < nh; ++i) {
		StageHelper &h = *helpers[i];
		unsigned spins = 0;
		while (h.done.load(std::memory_order_acquire) != job[i]) {
			__builtin_ia32_pause();
			if ((++spins & 4095u) == 0)
				h.wake(); // (the helper was on its way to sleep when the job was posted)
		}
	}
	owner.unlock();
}
int PinnedRing::acquire(size_t bytes) {
	const int i = next;
	next = (next + 1) % NB;
	if (!ev[i])
		MVS_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
	else
		MVS_HIP(hipEventSynchronize(ev[i])); // previous copy out of this slot has finished
	if (bytes > cap[i]) {
		if (buf[i])
			MVS_HIP(hipHostFree(buf[i]));
		buf[i] = nullptr;
		size_t want = std::max(bytes, (size_t)SLOT_BYTES);
		MVS_HIP(hipHostMalloc(&buf[i], want, hipHostMallocDefault));
		cap[i] = want;
	}
	return i;
}
void PinnedRing::release(int i, hipStream_t st) {
	MVS_HIP(hipEventRecord(ev[i], st));
}

void ensure_dynamic_lds(const void *kernel, size_t bytes) {
	static std::mutex mu;
	static std::map<std::pair<int, const void *>, size_t> raised;
	int dev = 0;
	MVS_HIP(hipGetDevice(&dev));
	std::lock_guard<std::mutex> g(mu);
	size_t &cur = raised[{dev, kernel}];
	if (bytes > cur) {
		MVS_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
		cur = bytes;
	}
}

// ------------------------------------------------------------------------------------------ base

static thread_local int tl_ctor_device = -1;
CtorDevice::CtorDevice(int dev) : prev(tl_ctor_device) {
	if (dev >= 0)
		tl_ctor_device = dev;
}
CtorDevice::~CtorDevice() {
	tl_ctor_device = prev;
}
static int env_device() {
	if (tl_ctor_device >= 0)
		return tl_ctor_device;
	const char *e = getenv("MVS_DEVICE");
	return e ? atoi(e) : 0;
}

IndexBase::IndexBase(int kind_, int d_, int metric_) : kind(kind_), d(d_), metric(metric_) {
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		throw_faiss("mvs::IndexBase::IndexBase", __FILE__,
		            "no MI355X (gfx950) device is usable (%s): the MI355X vector-search path has no CPU fallback",
		            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	device = env_device();
	if (device < 0 || device >= ndev)
		throw_faiss("mvs::IndexBase::IndexBase", __FILE__, "Invalid GPU device %d", device);
	MVS_HIP(hipSetDevice(device));
	MVS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}
IndexBase::~IndexBase() {
	forget_current_tuning(&tune_); // (the destroying thread's next launch read freed memory; other threads enter through use_device())
	if (stream) {
		(void)hipSetDevice(device);
		(void)hipStreamSynchronize(stream);
		(void)hipStreamDestroy(stream);
	}
	for (auto &p : timing_events) {
		(void)hipEventDestroy(p.first);
		(void)hipEventDestroy(p.second);
	}
}
// make stream `waiter` wait for everything enqueued so far on `signal`
void stream_wait(hipStream_t waiter, hipStream_t signal) {
	if (waiter == signal)
		return;
	hipEvent_t e;
	MVS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	MVS_HIP(hipEventRecord(e, signal));
	MVS_HIP(hipStreamWaitEvent(waiter, e, 0));
	MVS_HIP(hipEventDestroy(e));
}

void IndexBase::use_device() const {
	MVS_HIP(hipSetDevice(device)); // DuckDB calls from arbitrary worker threads
	set_current_tuning(&tune_);    // ... and this index's tuning is what the launches of this call read (csrc/common.h Tuning)
}
void IndexBase::train(int64_t, const float *) {
	// faiss::Index::train: "does nothing by default"
}
void IndexBase::add_with_ids(int64_t, const float *, const int64_t *) {
	// faiss/Index.cpp -- substring matched at src/faiss_extension.cpp:523; user text pinned by faiss4.test:22
	throw_faiss("virtual void faiss::Index::add_with_ids(faiss::idx_t, const float*, const faiss::idx_t*)",
	            "faiss/Index.cpp", "add_with_ids not implemented for this type of index");
}
void IndexBase::add_with_ids_device(int64_t, const float *, const int64_t *, hipStream_t) {
	throw_faiss("virtual void faiss::Index::add_with_ids(faiss::idx_t, const float*, const faiss::idx_t*)",
	            "faiss/Index.cpp", "add_with_ids not implemented for this type of index");
}

// host-pointer search = stage queries, device search, copy back (src/faiss_extension.cpp:623-638)
void IndexBase::search(int64_t nq, const float *x, int64_t k, float *D, int64_t *I, const mvs_search_params *params) {
	use_device();
	if (k <= 0)
		throw_faiss("virtual void faiss::Index::search(...) const", "faiss/Index.cpp", "Error: 'k > 0' failed");
	if (nq <= 0)
		return;
	ws_hx.reserve((size_t)nq * d * sizeof(float));
	ws_hD.reserve((size_t)nq * k * sizeof(float));
	ws_hI.reserve((size_t)nq * k * sizeof(int64_t));
	const size_t xbytes = (size_t)nq * d * sizeof(float);
	int slot = pinned.acquire(xbytes);
	memcpy(pinned.buf[slot], x, xbytes);
	MVS_HIP(hipMemcpyAsync(ws_hx.p, pinned.buf[slot], xbytes, hipMemcpyHostToDevice, stream));
	pinned.release(slot, stream);
	search_device(nq, (const float *)ws_hx.p, k, (float *)ws_hD.p, (int64_t *)ws_hI.p, params, stream);
	const size_t dbytes = (size_t)nq * k * sizeof(float), ibytes = (size_t)nq * k * sizeof(int64_t);
	slot = pinned.acquire(dbytes + ibytes);
	char *hb = (char *)pinned.buf[slot];
	MVS_HIP(hipMemcpyAsync(hb, ws_hD.p, dbytes, hipMemcpyDeviceToHost, stream));
	MVS_HIP(hipMemcpyAsync(hb + dbytes, ws_hI.p, ibytes, hipMemcpyDeviceToHost, stream));
	MVS_HIP(hipStreamSynchronize(stream));
	memcpy(D, hb, dbytes);
	memcpy(I, hb + dbytes, ibytes);
	pinned.release(slot, stream);
}

void IndexBase::tie_emit(const int *, int, const float *, const float *, int64_t, const mvs_search_params *, const int64_t *, float *,
                         int64_t *, int *, hipStream_t) {
	throw_faiss("mvs::IndexBase::tie_emit", __FILE__, "not an IVF row shard");
}
void IndexBase::begin_kernel_timing(hipStream_t st) {
	if (!timing_enabled)
		return;
	hipEvent_t a, b;
	MVS_HIP(hipEventCreate(&a));
	MVS_HIP(hipEventCreate(&b));
	timing_events.emplace_back(a, b);
	MVS_HIP(hipEventRecord(a, st));
}
void IndexBase::end_kernel_timing(hipStream_t st) {
	if (!timing_enabled)
		return;
	MVS_HIP(hipEventRecord(timing_events.back().second, st));
}
void IndexBase::resolve_kernel_timing(int *count, double *total_ms) {
	for (auto &p : timing_events) {
		MVS_HIP(hipEventSynchronize(p.second));
		float ms = 0.f;
		MVS_HIP(hipEventElapsedTime(&ms, p.first, p.second));
		timing_total_ms += ms;
		timing_count++;
		kinfo.last_ms = ms;
		(void)hipEventDestroy(p.first);
		(void)hipEventDestroy(p.second);
	}
	timing_events.clear();
	*count = timing_count;
	*total_ms = timing_total_ms;
}

// ------------------------------------------------------------------------------------------ selector

SelectorDev SelectorHolder::upload(const mvs_search_params *p, hipStream_t st) {
	SelectorDev s;
	memset(&s, 0, sizeof s);
	if (!p || p->sel_kind == MVS_SEL_NONE)
		return s;
	s.kind = p->sel_kind;
	if (p->sel_kind == MVS_SEL_BITMAP) {
		// faiss::IDSelectorBitmap(n_bytes, bitmap): src/faiss_extension.cpp:959
		buf.reserve((size_t)std::max<int64_t>(p->sel_n, 1));
		if (p->sel_n > 0)
			MVS_HIP(hipMemcpyAsync(buf.p, p->sel_data, (size_t)p->sel_n, hipMemcpyHostToDevice, st));
		s.bitmap = (const uint8_t *)buf.p;
		s.nbytes = p->sel_n;
	} else if (p->sel_kind == MVS_SEL_BATCH) {
		// faiss::IDSelectorBatch(n, ids): bloom filter + hash set == set membership; sorted array + bisect here
		sorted.assign((const int64_t *)p->sel_data, (const int64_t *)p->sel_data + p->sel_n);
		std::sort(sorted.begin(), sorted.end());
		buf.reserve(std::max<size_t>(sorted.size() * sizeof(int64_t), 8));
		if (!sorted.empty())
			MVS_HIP(hipMemcpyAsync(buf.p, sorted.data(), sorted.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
		s.sorted_ids = (const int64_t *)buf.p;
		s.nids = (int64_t)sorted.size();
	} else {
		throw_faiss("mvs::SelectorHolder::upload", __FILE__, "unknown selector kind %d", p->sel_kind);
	}
	return s;
}

// ------------------------------------------------------------------------------------------ Flat

FlatIndex::FlatIndex(int d_, int metric_) : IndexBase(MVS_KIND_FLAT, d_, metric_) {
	if (metric != METRIC_L2 && metric != METRIC_IP && !metric_is_extra(metric))
		throw_faiss("mvs::FlatIndex::FlatIndex", __FILE__, "metric type %d is not a faiss::MetricType the glue registers",
		            metric);
	geom = flat_geom_for(d);
	is_trained = true;
	if (const char *e = getenv("MVS_LAZY_ADDS")) // (same-box A/B of the ingest staging through host/boundary_driver, which sets no options)
		lazy_adds = atoi(e) != 0;
}
FlatIndex::~FlatIndex() {
	delete shadow;
	shadow = nullptr;
	(void)hipSetDevice(device);
	if (stream)
		(void)hipStreamSynchronize(stream);
	reap_retired(true);
	if (vecs)
		(void)hipFree(vecs);
	if (norms)
		(void)hipFree(norms);
	if (h_flag_count)
		(void)hipHostFree(h_flag_count);
	if (vecs_bf)
		(void)hipFree(vecs_bf);
	if (d_max_norm_bits)
		(void)hipFree(d_max_norm_bits);
}

void FlatIndex::drop_bf16_rows() {
	if (vecs_bf)
		(void)hipFree(vecs_bf);
	vecs_bf = nullptr;
	bf_cap = bf_rows = 0;
	if (vecs_h1)
		(void)hipFree(vecs_h1);
	if (beta_h1)
		(void)hipFree(beta_h1);
	if (mu_h1)
		(void)hipFree(mu_h1);
	vecs_h1 = nullptr;
	beta_h1 = mu_h1 = nullptr;
	h1_cap = h1_rows = 0;
	if (d_outl)
		(void)hipFree(d_outl);
	d_outl = nullptr;
	h1_outliers = 0;
	if (d_max_norm_bits)
		(void)hipFree(d_max_norm_bits);
	d_max_norm_bits = nullptr;
}

// rows [bf_rows, ntotal) of the f32 store -> bf16 hi/lo (+ running maximum of the squared norms); derived data, so add(),
// clone() and to_device() never have to know about it
void FlatIndex::ensure_bf16_rows(hipStream_t st) {
	if (bf_rows == ntotal && vecs_bf)
		return;
	if (ntotal > bf_cap || !vecs_bf) {
		unsigned short *nb = nullptr;
		const int64_t nc = std::max<int64_t>(cap, ntotal);
		// + 64 rows: the kernel prefetches up to two tiles past the last row without clamping
		const size_t nbytes = ((size_t)nc + 64) * 2 * geom.dp * sizeof(unsigned short);
		MVS_HIP(hipMalloc((void **)&nb, nbytes));
		MVS_HIP(hipMemsetAsync(nb, 0, nbytes, st));
		if (bf_rows > 0)
			MVS_HIP(hipMemcpyAsync(nb, vecs_bf, (size_t)bf_rows * 2 * geom.dp * sizeof(unsigned short),
			                       hipMemcpyDeviceToDevice, st));
		MVS_HIP(hipStreamSynchronize(st));
		if (vecs_bf)
			MVS_HIP(hipFree(vecs_bf));
		vecs_bf = nb;
		bf_cap = nc;
	}
	if (!d_max_norm_bits) {
		MVS_HIP(hipMalloc((void **)&d_max_norm_bits, 64));
		MVS_HIP(hipMemsetAsync(d_max_norm_bits, 0, 64, st));
	}
	launch_rows_to_bf16(geom, vecs, bf_rows, ntotal - bf_rows, vecs_bf, norms, d_max_norm_bits, st);
	bf_rows = ntotal;
}

// the same for the coarse filter's store: rows as bf16 only (csrc/flat_collect.hip)
void FlatIndex::ensure_h1_rows(hipStream_t st) {
	if (h1_rows == ntotal && vecs_h1)
		return;
	const int dp1 = collect_store_dims(d); // row pitch of the bf16 store: 128 up to d = 128, else 256 / 384 / 512
	if (!d_max_norm_bits) {
		MVS_HIP(hipMalloc((void **)&d_max_norm_bits, 64));
		MVS_HIP(hipMemsetAsync(d_max_norm_bits, 0, 64, st));
	}
	if (!mu_h1) { // the centre is fixed at the first build (any vector is valid; rows added later only fit it less well)
		MVS_HIP(hipMalloc((void **)&mu_h1, (size_t)std::max(geom.dp, 1024) * sizeof(float)));
		const int64_t nm = std::min<int64_t>(ntotal, (int64_t)1 << 20);
		launch_collect_mean(geom, vecs, nm, mu_h1, st);
		// ... and so is the outlier threshold tau = 64 x the mean ||y - mu||^2 of those rows (csrc/flat_collect.hip "outlier rows").  Only
		// stores large enough for the coarse filter to matter: an IVF quantiser's centroids, which csrc/coarse_bf16.hip searches through
		// this store too, are never taken out of it.
		const unsigned inf_bits = 0x7f800000u;
		MVS_HIP(hipMemcpyAsync(d_max_norm_bits + 3, &inf_bits, sizeof inf_bits, hipMemcpyHostToDevice, st));
		if (ntotal >= 262144 && outlier_rows) {
			MVS_HIP(hipMalloc((void **)&d_outl, (size_t)(1 + CL_OUTL_CAP) * sizeof(int)));
			MVS_HIP(hipMemsetAsync(d_outl, 0, (size_t)(1 + CL_OUTL_CAP) * sizeof(int), st));
			launch_collect_outlier_threshold(norms, nm, mu_h1, geom.d, d_max_norm_bits, st);
		}
		MVS_HIP(hipStreamSynchronize(st)); // (inf_bits is a stack variable)
	}
	if (ntotal > h1_cap || !vecs_h1) {
		unsigned short *nb = nullptr;
		float *nbeta = nullptr;
		const int64_t nc = std::max<int64_t>(cap, ntotal);
		const size_t nbytes = ((size_t)nc + 192) * dp1 * sizeof(unsigned short); // + 192 rows: unclamped prefetch of a 64-row block
		MVS_HIP(hipMalloc((void **)&nb, nbytes));
		MVS_HIP(hipMalloc((void **)&nbeta, ((size_t)nc + 192) * sizeof(float)));
		MVS_HIP(hipMemsetAsync(nb, 0, nbytes, st));
		MVS_HIP(hipMemsetAsync(nbeta, 0, ((size_t)nc + 192) * sizeof(float), st));
		if (h1_rows > 0) {
			MVS_HIP(hipMemcpyAsync(nb, vecs_h1, (size_t)h1_rows * dp1 * sizeof(unsigned short), hipMemcpyDeviceToDevice, st));
			MVS_HIP(hipMemcpyAsync(nbeta, beta_h1, (size_t)h1_rows * sizeof(float), hipMemcpyDeviceToDevice, st));
		}
		MVS_HIP(hipStreamSynchronize(st));
		if (vecs_h1)
			MVS_HIP(hipFree(vecs_h1));
		if (beta_h1)
			MVS_HIP(hipFree(beta_h1));
		vecs_h1 = nb;
		beta_h1 = nbeta;
		h1_cap = nc;
	}
	if (dp1 > 128) // csrc/flat_collect_wide.hip
		launch_rows_to_bf16_wide(metric, vecs, geom.dp, geom.pair_interleaved ? 1 : 0, d, dp1, h1_rows, ntotal - h1_rows, mu_h1, vecs_h1,
		                         beta_h1, norms, d_max_norm_bits, st, d_outl);
	else
		launch_rows_to_bf16_hi(geom, metric, vecs, h1_rows, ntotal - h1_rows, mu_h1, vecs_h1, beta_h1, norms, d_max_norm_bits, st, d_outl);
	h1_rows = ntotal;
	if (d_outl) { // (one 4-byte read per conversion: the scans' launch code needs to know whether there is anything to append)
		int cnt = 0;
		MVS_HIP(hipMemcpyAsync(&cnt, d_outl, sizeof cnt, hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		h1_outliers = std::min(cnt, CL_OUTL_CAP);
		outl_total = cnt;
	}
}

// Coarse filter front half (csrc/flat_collect.hip): bound estimation pre-pass, the scan, candidates grouped by query and
// re-scored exactly.  Leaves the kk best exact candidates per query in *pd1 / *pi1 ([nq][kk]) and the queries whose bound
// is not finite in fail_q.  false: the candidate stream overflowed (the caller uses the bf16x3 path instead).
// kk: entries selected per query; kf <= kk: the k the FILTER works with (class slots, bound rank) -- see search_prefilter_pass
bool FlatIndex::collect_candidates(int64_t nq, const float *d_x, int kk, float **pd1_out, int32_t **pi1_out, int *fail_cnt,
                                   int *fail_q, const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st,
                                   bool defer_count, int kf) {
	if (kf <= 0 || kf > kk)
		kf = kk;
	// lists beyond 128 entries (round 6, d <= 128 store): the bounds come from a pass of their own over nranges row ranges, the scan runs
	// against them frozen, the selection is a segmented sort -- csrc/flat_collect.hip "lists beyond 128 entries"
	const bool bigk = kf > 128;
	if (bigk)
		defer_count = false; // (the candidate count sizes the sort: read back behind the scan)
	ensure_h1_rows(st);
	// IDSelector: one bit per row, built per search (the selector sees idmap[row] behind an IndexIDMap, the row number else)
	const bool has_sel = params && params->sel_kind != MVS_SEL_NONE;
	const unsigned long long *rowmask = nullptr;
	if (has_sel) {
		SelectorDev sel = selector.upload(params, st);
		ws_rowmask.reserve(collect_rowmask_bytes(ntotal));
		launch_collect_rowmask(sel, d_idmap, ntotal, (unsigned long long *)ws_rowmask.p, st);
		rowmask = (const unsigned long long *)ws_rowmask.p;
	}
	const int dp1 = collect_store_dims(d);
	const bool wide = dp1 > 128; // csrc/flat_collect_wide.hip: no one-wavefront-per-segment path
	ws_qn.reserve((size_t)nq * sizeof(float));
	const int64_t nq128 = (nq + 255) / 256 * 256;
	ws_e2.reserve((size_t)nq128 * sizeof(float));
	// (big lists: pass A looks at a quarter of the rows -- any rows give a valid bound, fewer rows a lower one and ~ntotal / rows times k
	// candidates -- unless the candidates of a large batch would not fit: then at all of them)
	// (d <= 128: the ranges are row splits of the register pre-pass kernel, 16 class maxima each, ceil(k / ranges) <= 8 of them decide; the
	// wide stores: 128 class slots per range through the scan kernel's bound-estimation instances, <= 64 decide)
	const bool bk_seed = bigk && !wide;
	const int bk_per = cl_bigk_per > 0 ? cl_bigk_per : 8; // (rows of a split that decide its bound: the bk_per-th best of its 16 class maxima)
	int bk_ranges = bigk ? (bk_seed ? (kf + bk_per - 1) / bk_per : (kf + 63) / 64) : 0;
	if (bk_seed) { // (a small batch has few query blocks: more, shorter splits -- up to four times as many -- fill the device; the bound loosens a little)
		const int nqb = (int)((nq + 511) / 512);
		if (nqb * bk_ranges < 256 && bk_ranges < 64) // (k = 200 at 32 queries: 3.5 -> 2.3 ms; k = 1000 has splits enough and only loses tightness)
			bk_ranges = std::max(bk_ranges, std::min(256 / nqb, 4 * bk_ranges));
	}
	int64_t bk_rows = 0; // rows per range
	if (bigk) {
		// (a large batch pays the matrix pipe for pass A and the rare path, the gather and the sort for every candidate: all rows = a
		// tighter bound = a third of the candidates; a small batch pays bandwidth: a quarter of the rows.  option cl_bigk_whole: -1 auto)
		// (measured, N = 10 M, profiles/r6_big_k.txt: k = 1000 at 2 048 / 10 000 queries 15.2 / 70 ms on a quarter, 11.2 / 48 on all rows; k = 200: 8.3 / 31 against 13 / 36)
		const bool whole = cl_bigk_whole >= 0 ? cl_bigk_whole != 0 : ((nq >= 512 && kf >= 512) || (double)nq * kf * 16.0 > (double)((int64_t)1 << 28));
		const int64_t want = whole ? ntotal : std::max<int64_t>(ntotal / 4, (int64_t)bk_ranges * (bk_seed ? 2048 : 16384));
		bk_rows = std::min<int64_t>(std::max<int64_t>(want / bk_ranges, bk_seed ? 1024 : 4096), ntotal / bk_ranges) / 64 * 64;
	}
	ws_gthr.reserve((size_t)nq * collect_slot_stride(kf, collect_store_dims(d)) * sizeof(unsigned) * (bigk && !bk_seed ? bk_ranges : 1) + 64);
	if (bk_seed)
		ws_seed.reserve((size_t)bk_ranges * (size_t)nq * 16 * sizeof(float));
	ws_seg.reserve(256 + (size_t)2 * nq * sizeof(int));
	// Round 5 (d <= 128 store): fragments, ||x||^2, bounds, neutral class slots and the zeroed control block in ONE launch
	// (csrc/flat_collect.hip collect_query_prep_kernel) instead of four kernels and a memset
	const bool prep1 = !wide && cl_prep1;
	if (prep1) {
		ws_pfq.reserve(collect_qfrag_bytes(geom, nq));
		launch_collect_query_prep(metric, d_x, nq, d, mu_h1, d_max_norm_bits, ws_pfq.p, (float *)ws_qn.p, (float *)ws_e2.p, fail_cnt, fail_q,
		                          (unsigned *)ws_gthr.p, collect_slot_stride(kf, collect_store_dims(d)), (int *)ws_seg.p,
		                          (int *)((char *)ws_seg.p + 256), st);
	} else {
	if (wide) {
		ws_pfq.reserve(collect_qfrag_bytes_ex(dp1, collect_wide_qblock(dp1), nq));
		launch_collect_pack_queries_ex(d, dp1, collect_wide_qblock(dp1), metric, d_x, nq, mu_h1, ws_pfq.p, st);
	} else {
		ws_pfq.reserve(collect_qfrag_bytes(geom, nq));
		launch_collect_pack_queries(geom, metric, d_x, nq, mu_h1, ws_pfq.p, st);
	}
	if (metric == METRIC_L2) // (inner product re-scores without them)
		launch_query_norms(d_x, nq, d, (float *)ws_qn.p, st);
	// (the bounds kernel writes NaN into the slots behind the last query itself)
	launch_collect_bounds(metric, d_x, nq, d, mu_h1, d_max_norm_bits, (float *)ws_e2.p, fail_cnt, fail_q, st);
	}
	// candidate stream: 4096 entries per query to start with (option cl_stream_cap; at least 2^20), or what the last overflow
	// showed this index's data to need (cl_cap_hint, up to 16384 per query: clustered rows with large norms admit thousands)
	int64_t cap_entries = cl_stream_cap_per_query > 0 ? std::max<int64_t>(nq * cl_stream_cap_per_query, 1024)
	                                                  : std::max<int64_t>(nq * std::max<int64_t>(4096, cl_cap_hint), (int64_t)1 << 20);
	if (bigk && (ntotal / std::max(bk_ranges, 1) < 192 || bk_rows < 64))
		return false; // (ranges too short to mean anything: the exact kernels)
	if (bigk && cl_stream_cap_per_query <= 0) // (~ k ntotal / (rows of pass A) candidates per query, times the slack of a class bound)
		cap_entries = std::min<int64_t>(((int64_t)1 << 31) - 4096,
		                                std::max<int64_t>(cap_entries, nq * std::min<int64_t>(ntotal, 4 * (int64_t)kf * ntotal / std::max<int64_t>(bk_rows * bk_ranges, 1))));
	size_t half = ((size_t)cap_entries * 8 + 255) & ~(size_t)255;
	ws_stream.reserve(256 + 2 * half);
	// one zeroed control block {stream count | per-query segments} (round 4: one memset instead of three per search)
	if (!prep1)
		MVS_HIP(hipMemsetAsync(ws_seg.p, 0, 256 + (size_t)2 * nq * sizeof(int), st));
	unsigned long long *cnt = (unsigned long long *)ws_seg.p;
	int *const seg = (int *)((char *)ws_seg.p + 256);
	unsigned long long *stream = (unsigned long long *)((char *)ws_stream.p + 256);
	unsigned long long *sorted = (unsigned long long *)((char *)ws_stream.p + 256 + half);
	ws_pbnd.reserve(collect_bound_table_bytes(nq));
	float *pbnd = wide ? nullptr : (float *)ws_pbnd.p; // (the d <= 128 scan only)
	if (!wide && cl_seed_stage)
		ws_seed.reserve(collect_seed_stage_bytes(nq));
	if (bk_seed)
		launch_collect_big_bounds_seed(geom, metric, ws_pfq.p, vecs_h1, beta_h1, ntotal, nq, kf, bk_ranges, bk_rows, (const float *)ws_e2.p,
		                               (float *)ws_seed.p, rowmask, (float *)ws_pbnd.p, st);
	else if (bigk)
		launch_collect_big_bounds(geom, metric, ws_pfq.p, vecs_h1, beta_h1, ntotal, nq, kf, bk_ranges, bk_rows, (const float *)ws_e2.p,
		                          (unsigned *)ws_gthr.p, rowmask, (float *)ws_pbnd.p, st);
	else
	launch_collect_prepare(geom, metric, ws_pfq.p, vecs_h1, beta_h1, ntotal, nq, kf, (const float *)ws_e2.p,
	                       (unsigned *)ws_gthr.p, cnt, rowmask, pbnd, st, true, prep1, (!wide && cl_seed_stage) ? (float *)ws_seed.p : nullptr);
	int grid = 0, nsplit = 0, lds = 0;
	const bool few = !wide && nq <= 128 && collect_slot_stride(kf, collect_store_dims(d)) == 16 && ntotal < ((int64_t)1 << 31) && cl_small_path; // (one work item; at 256 queries: 1.92 vs 1.55 ms)
	int64_t ncand = 0;
	// the bucketed finish (see cl_fbucket in csrc/index.h): the common d = 128 shape; round 6: inner product too -- faiss_create's default
	// metric (src/faiss_extension.cpp:105) gets the headline's pipeline, the select kernel prints FAISS's CMin-heap order and the tie flags
	const bool fb = !bigk && cl_fbucket && !cl_fbucket_off && cl_out_D && (metric == METRIC_L2 || metric == METRIC_IP) && !wide && !few && d == 128 &&
	                geom.dp == 128 && kk <= 64 && nq * (int64_t)cl_fpitch < ((int64_t)1 << 31);
	cl_emitted = false;
	cl_wrf_used = false;
	float *stream_s = nullptr, *fb_thr = nullptr;
	unsigned *fb_rows = nullptr;
	unsigned long long *fb_keys = nullptr, *fb_units = nullptr;
	int *fb_kept = nullptr; // per-workgroup survivor counts of the scatter kernel (their sum -> the control block's header @8)
	auto fb_layout = [&]() { // (after every change of cap_entries / of the pitch: a DevBuf keeps nothing when it grows, so the scan's
		// per-entry values and the buckets live in two buffers -- a larger pitch must not lose the values)
		const size_t sb = ((size_t)cap_entries * 4 + 255) & ~(size_t)255, tb = ((size_t)nq * 4 + 255) & ~(size_t)255;
		const size_t rb = ((size_t)nq * cl_fpitch * 4 + 255) & ~(size_t)255, kb = (size_t)nq * cl_fpitch * 8;
		ws_fbk.reserve(sb + tb + 256);
		const size_t unb = (ivf_bucket_units_bytes(cap_entries) + 255) & ~(size_t)255;
		ws_fbr.reserve(rb + kb + unb + (size_t)ivf_bucket_scatter_blocks(cap_entries) * 4 + 256);
		stream_s = (float *)ws_fbk.p;
		fb_thr = (float *)((char *)ws_fbk.p + sb);
		fb_rows = (unsigned *)ws_fbr.p;
		fb_keys = (unsigned long long *)((char *)ws_fbr.p + rb);
		fb_units = (unsigned long long *)((char *)ws_fbr.p + rb + kb);
		fb_kept = (int *)((char *)ws_fbr.p + rb + kb + unb);
	};
	if (fb)
		fb_layout();
	// (round 5) the 512 < d <= 1536 stores: the big kernel keeps every candidate's coarse value too; the final-bound filter then compacts
	// the stream in front of the sort (a candidate costs 3-6 KB of f32 row there)
	const bool wrf = !fb && !bigk && wide && cl_wide_refilter && !strcmp(collect_wide_kernel_name(dp1), "flat_bf16_big_kernel");
	auto wrf_layout = [&]() {
		const size_t sb = ((size_t)cap_entries * 4 + 255) & ~(size_t)255;
		ws_fbk.reserve(sb + (((size_t)nq * 4 + 255) & ~(size_t)255) + 256);
		stream_s = (float *)ws_fbk.p;
		fb_thr = (float *)((char *)ws_fbk.p + sb);
	};
	if (wrf)
		wrf_layout();
	for (int attempt = 0;; ++attempt) {
	begin_kernel_timing(st);
	if (few) {
		// Small batches are bound by streaming the bf16 store, not by the matrix pipe: the one-wavefront-per-segment kernel of
		// the IVF scan (csrc/ivf_collect.hip: the database as one list, identity query map, gamma = 0) keeps more bytes in
		// flight per CU than two 256-thread workgroups do
		const int nit = (int)((nq + 127) / 128), seg_rows = 2048;
		const int nseg = (int)((ntotal + seg_rows - 1) / seg_rows);
		const size_t it_bytes = 256, q_bytes = ((size_t)nq * 4 + 255) & ~(size_t)255, g_bytes = (size_t)nit * 128 * sizeof(float);
		ws_items1.reserve(it_bytes + 256 + q_bytes + g_bytes);
		char *b = (char *)ws_items1.p;
		int *nitems_dev = (int *)(b + it_bytes);
		int *qidx = (int *)(b + it_bytes + 256);
		float *gam = (float *)(b + it_bytes + 256 + q_bytes);
		MVS_HIP(hipMemsetAsync(gam, 0, g_bytes, st));
		launch_collect_flat_items(b, nitems_dev, qidx, nq, ntotal, st);
		launch_ivf_collect_scan(b, nitems_dev, nit, qidx, ws_pfq.p, gam, (const float *)ws_e2.p, vecs_h1, beta_h1,
		                        (unsigned *)ws_gthr.p, stream, cnt, cap_entries, kf, seg_rows, nseg, 1, (const unsigned *)rowmask, st);
		grid = nit * nseg;
		nsplit = nseg;
		lds = 20544;
	} else {
		launch_collect_scan(geom, metric, ws_pfq.p, vecs_h1, beta_h1, ntotal, nq, kf, (const float *)ws_e2.p,
		                    (unsigned *)ws_gthr.p, stream, cnt, cap_entries, rowmask, bigk ? (float *)ws_pbnd.p : pbnd, st, &grid, &nsplit, &lds, stream_s,
		                    bigk);
	}
	end_kernel_timing(st);
	if (h1_outliers > 0) // the rows kept out of the store join every query's candidates (csrc/flat_collect.hip "outlier rows")
		launch_collect_append_outliers(d_outl, h1_outliers, nq, stream, stream_s, cnt, cap_entries, rowmask, st);
	if (!h_flag_count)
		MVS_HIP(hipHostMalloc((void **)&h_flag_count, 64, hipHostMallocDefault));
	cl_report_cnt = defer_count && cl_est_per_query > 0; // (deferred: the caller's report kernel carries the count, no copy of its own)
	if (!cl_report_cnt)
		MVS_HIP(hipMemcpyAsync(h_flag_count + 10, cnt, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
	if (defer_count && cl_est_per_query > 0) { // the caller looks at the count after its own synchronisation
		// (size of the sort: collect_sort_estimate of what the previous search of this index produced per query, in units of 64 K entries)
		cl_deferred_cap = fb ? cap_entries : std::min<int64_t>(cap_entries, collect_sort_estimate(cl_est_per_query, nq)); // (no sort to size)
		break;
	}
	defer_count = false; // (the first search of an index has no estimate yet: the synchronous way, which leaves one)
	MVS_HIP(hipStreamSynchronize(st));
	unsigned long long ncand_u;
	memcpy(&ncand_u, h_flag_count + 10, sizeof ncand_u);
	ncand = (int64_t)ncand_u;
	cl_last_candidates = ncand;
	if (ncand <= cap_entries)
		break;
	// The stream overflowed.  Usually a FEW queries are responsible (a query that sits on a vector stored thousands of times:
	// every copy is a candidate, rightly).  Round 2 handed the whole batch to the bf16x3 / f32 kernels (a 4-11x cliff, VERDICT r2
	// weak #5); now the queries that hold more than their share of the stream are taken out of the coarse filter (2E = NaN:
	// nothing of theirs passes; they join the fail list and are re-run on the exact kernel) and the scan runs once more for
	// the others, with the class slots already warm.  If nobody stands out, or the second scan overflows too: as before.
	++cl_overflows;
	if (attempt > 0 || few)
		return false;
	const int64_t grow_max = cl_stream_cap_per_query > 0 ? 4 * (int64_t)cl_stream_cap_per_query : std::max<int64_t>(16384, bigk ? 64 * (int64_t)kf : 0);
	if (bigk && (ncand + ncand / 8 > nq * grow_max || ncand + ncand / 8 >= ((int64_t)1 << 31)))
		return false; // (frozen bounds: no query can be taken out of this scan -- the exact kernels take the batch)
	if (ncand + ncand / 8 <= nq * grow_max) {
		// (a) the data simply admits more rows per query than the stream was sized for: a larger stream, one more scan (the
		// class slots are warm: it admits no more than the first), and the next search of this index starts with that size
		cap_entries = ncand + ncand / 8;
		if (cl_stream_cap_per_query <= 0)
			cl_cap_hint = std::max<int64_t>(cl_cap_hint, (cap_entries + nq - 1) / nq);
		half = ((size_t)cap_entries * 8 + 255) & ~(size_t)255;
		ws_stream.reserve(256 + 2 * half); // (a DevBuf keeps nothing when it grows: the counter -- in ws_seg -- is reset below anyway)
		stream = (unsigned long long *)((char *)ws_stream.p + 256);
		sorted = (unsigned long long *)((char *)ws_stream.p + 256 + half);
		if (fb)
			fb_layout();
		if (wrf)
			wrf_layout();
	} else {
		// (b) a few queries hold far more than their share: out of the coarse filter with them, one more scan for the others
		ws_qcount.reserve((size_t)(nq + 16) * sizeof(int));
		MVS_HIP(hipMemsetAsync(ws_qcount.p, 0, (size_t)(nq + 16) * sizeof(int), st));
		const int64_t share = std::max<int64_t>(1, cap_entries / nq);
		const int nheavy = launch_collect_drop_heavy(stream, cap_entries, nq, (int)std::min<int64_t>(4 * share, 1 << 30), (int *)ws_qcount.p,
		                                             (float *)ws_e2.p, fail_cnt, fail_q, st); // (syncs the stream)
		if (nheavy <= 0 || nheavy > nq / 2)
			return false;
		cl_heavy_total += nheavy;
	}
	MVS_HIP(hipMemsetAsync(cnt, 0, 16, st));
	} // attempt
	if (!defer_count) {
		cl_queries_total += nq;
		cl_candidates_total += ncand;
		cl_est_per_query = (double)ncand / (double)std::max<int64_t>(nq, 1) + 1e-6;
		cl_deferred_cap = 0; // (tells the caller that this pass was synchronous)
	}
	bool fb_done = false;
	if (fb) {
		// final bound -> survivors into the queries' row buckets -> exact values (FAISS's BLAS-branch formula, or the per-pair sum with a
		// selector / fewer than 20 queries) -> selection, FAISS's order, labels: straight into the caller's arrays
		unsigned *bcount = (unsigned *)seg; // (zero: the preparation kernel / the memset above; the select kernel leaves it zero)
		launch_collect_final_thr((const unsigned *)ws_gthr.p, d, kf, (const float *)ws_e2.p, nq, fb_thr, st);
		if (!h_cl_hdr)
			MVS_HIP(hipHostMalloc((void **)&h_cl_hdr, 256, hipHostMallocDefault));
		for (;;) {
			launch_ivf_bucket_scatter(stream, stream_s, cap_entries, cnt, nullptr, 0, 0, nq, fb_thr, fb_rows, bcount, cl_fpitch, fb_kept, fb_units,
			                          (unsigned *)cnt + 4, st);
			IvfFlatArith fa;
			memset(&fa, 0, sizeof fa);
			if (!(has_sel || nq < 20))
				fa.qn = (const float *)ws_qn.p, fa.yn = norms;
			if (metric == METRIC_IP) { // the k-ordered chain either way (fvec_inner_product = the BLAS-branch oracle arithmetic)
				IpFlatEmit ipf;
				memset(&ipf, 0, sizeof ipf);
				if (cl_out_flags)
					ipf.flags = *cl_out_flags;
				ipf.kout = cl_out_kout, ipf.D = cl_out_D, ipf.I = (long long *)cl_out_I;
				ipf.idmap = (const long long *)cl_out_map, ipf.label_offset = cl_out_off;
				// (reset = false: the per-query counts stay for the tie pass -- FlatIndex::resolve_ip_ties reads A_k off the buckets; the
				// next search's preparation zeroes them)
				cl_fb_keys = fb_keys, tie_bcount = bcount, tie_bpitch = cl_fpitch;
				launch_ivf_bucket_finish(METRIC_IP, nullptr, cap_entries, nullptr, fb_keys, bcount, cl_fpitch, nq, d_x, d, vecs, geom.dp, nullptr, kk,
				                         nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
				                         (unsigned long long *)((char *)ws_seg.p + 192), nullptr, nullptr, nullptr, false, st, nullptr, 0, fb_rows,
				                         geom.pair_interleaved ? 1 : 0, fb_units, (const unsigned *)cnt + 4, fb_kept,
				                         (int)ivf_bucket_scatter_blocks(cap_entries), cnt + 1, &ipf);
			} else
			launch_ivf_bucket_finish(METRIC_L2, nullptr, cap_entries, nullptr, fb_keys, bcount, cl_fpitch, nq, d_x, d, vecs, geom.dp, nullptr, kk,
			                         cl_out_D, cl_out_I, nullptr, cl_out_map, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
			                         (unsigned long long *)((char *)ws_seg.p + 192), nullptr, nullptr, nullptr, true, st, &fa, cl_out_off, fb_rows,
			                         geom.pair_interleaved ? 1 : 0, fb_units, (const unsigned *)cnt + 4, fb_kept,
			                         (int)ivf_bucket_scatter_blocks(cap_entries), cnt + 1);
			if (defer_count) { // (the caller looks at the header behind its own synchronisation: its report kernel copies it)
				fb_done = true;
				break;
			}
			MVS_HIP(hipMemcpyAsync(h_cl_hdr, ws_seg.p, 256, hipMemcpyDeviceToHost, st));
			MVS_HIP(hipStreamSynchronize(st));
			unsigned long long bmax = 0;
			memcpy(&bmax, (const char *)h_cl_hdr + 200, sizeof bmax);
			if ((int64_t)bmax <= cl_fpitch) {
				fb_done = true;
				break;
			}
			// some query has more survivors than its bucket holds: a larger pitch (the index remembers it) and the finish once more -- the
			// scan's stream and values are still there; beyond 16 384 per query the sorted pipeline takes this index for good
			const int64_t want = ((int64_t)bmax + (int64_t)bmax / 4 + 63) / 64 * 64;
			if (want > 16384 || nq * want >= ((int64_t)1 << 31)) {
				cl_fbucket_off = true;
				break;
			}
			cl_fpitch = (int)want;
			fb_layout();
			MVS_HIP(hipMemsetAsync((char *)ws_seg.p + 8, 0, 248 + (size_t)nq * sizeof(int), st)); // kept / unit counts, statistics, the bucket counters
			if (cl_out_flags) // (inner product: the tie flags of the abandoned finish were taken from truncated buckets)
				MVS_HIP(hipMemsetAsync(cl_out_flags->count, 0, sizeof(int), st));
		}
	}
	if (fb && !fb_done && cl_out_flags) // (the sorted pipeline below flags the boundary ties itself)
		MVS_HIP(hipMemsetAsync(cl_out_flags->count, 0, sizeof(int), st));
	if (fb_done) {
		cl_emitted = true;
		*pd1_out = nullptr;
		*pi1_out = nullptr;
		cl_sorted = nullptr;
	} else {
	// (the filtered stream is sorted in device-count mode whatever the count mode of the scan was)
	size_t temp = (defer_count || wrf) ? collect_sort_temp_bytes_est(defer_count ? cl_deferred_cap : std::max<int64_t>(ncand, 1), nq)
	                                   : (ncand > 0 ? collect_sort_temp_bytes(ncand, nq) : 0);
	if (kk > 128 && ncand > 0) // (the selection of a big list is a segmented sort of the exact keys)
		temp = std::max(temp, collect_select_big_temp_bytes(ncand, nq));
	ws_sorttmp.reserve(std::max<size_t>(temp, 16));
	const size_t ex_bytes = ((size_t)nq * kk * sizeof(float) + 255) & ~(size_t)255;
	ws_ex.reserve(ex_bytes + (size_t)nq * kk * sizeof(int32_t));
	float *pd1 = (float *)ws_ex.p;
	int32_t *pi1 = (int32_t *)((char *)ws_ex.p + ex_bytes);
	// (with a selector or fewer than 20 queries FAISS takes its per-pair branch: L2 = sum (x_k - y_k)^2; inner product is the
	// same chain either way)
	if (wrf) {
		// stream -> (filter) -> `sorted` -> (sort by query) -> stream: the survivors' count lives in the control block's header @8
		launch_collect_final_thr((const unsigned *)ws_gthr.p, d, kf, (const float *)ws_e2.p, nq, fb_thr, st);
		launch_stream_refilter(stream, stream_s, cap_entries, cnt, fb_thr, sorted, cnt + 1, st);
		launch_collect_rescore(metric, sorted, stream, defer_count ? cl_deferred_cap : std::max<int64_t>(ncand, 1), ws_sorttmp.p, temp, nq, kk, d_x, geom,
		                       vecs, norms, (const float *)ws_qn.p, seg, pd1, pi1, has_sel || nq < 20, st, cnt + 1, true);
		cl_sorted = stream;
		cl_wrf_used = true;
		if (!h_cl_hdr)
			MVS_HIP(hipHostMalloc((void **)&h_cl_hdr, 256, hipHostMallocDefault));
		if (!cl_report_cnt) // (synchronous count mode: the caller's report kernel does not carry the header)
			MVS_HIP(hipMemcpyAsync(h_cl_hdr, ws_seg.p, 256, hipMemcpyDeviceToHost, st));
	} else {
	launch_collect_rescore(metric, stream, sorted, defer_count ? cl_deferred_cap : ncand, ws_sorttmp.p, temp, nq, kk, d_x, geom, vecs, norms,
	                       (const float *)ws_qn.p, seg, pd1, pi1, has_sel || nq < 20, st, defer_count ? cnt : nullptr, true);
	cl_sorted = sorted;
	}
	*pd1_out = pd1;
	*pi1_out = pi1;
	}
	snprintf(kinfo.name, sizeof kinfo.name, "%s", wide ? collect_wide_kernel_name(collect_store_dims(d)) : "flat_bf16_collect_kernel");
	kinfo.flops = 2.0 * (double)nq * (double)ntotal * d;
	// one pass over the bf16 store (row pitch of the store + the row's f32 term) + the queries + the results
	kinfo.bytes = (double)ntotal * (collect_store_dims(d) * 2.0 + 4.0) + (double)nq * d * 4.0 + (double)nq * kk * 12.0;
	kinfo.grid = grid;
	kinfo.block = few ? 64 : 256;
	kinfo.lds_bytes = lds;
	kinfo.nsplit = nsplit;
	return true;
}

void FlatIndex::reset() {
	if (pend_slot >= 0) { // (staged rows die with the index's contents: the slot goes back to the ring unused)
		pend_rows = 0;
		flush_adds();
	}
	ntotal = 0;
	++mut_gen; // (rows change in place from here on: a shadow clustering of the old rows must not answer -- ADVICE r5)
	drop_bf16_rows();
	drop_shadow();
	if (shadow_state == 1)
		shadow_state = 0;
}

void FlatIndex::copy_rows_to_host(float *out) {
	use_device();
	flush_adds();
	MVS_HIP(hipStreamSynchronize(stream));
	if (ntotal <= 0)
		return;
	std::vector<float> tmp((size_t)ntotal * geom.dp);
	MVS_HIP(hipMemcpy(tmp.data(), vecs, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
	for (int64_t r = 0; r < ntotal; ++r) {
		const float *s = &tmp[(size_t)r * geom.dp];
		for (int kk = 0; kk < d; ++kk) {
			int src = kk;
			if (geom.pair_interleaved) { // stored [k0,k2,k1,k3] (bit 4 of r clear) or [k1,k3,k0,k2]
				static const int pos0[4] = {0, 2, 1, 3}, pos1[4] = {2, 0, 3, 1};
				src = (kk & ~3) + (((r >> 4) & 1) ? pos1[kk & 3] : pos0[kk & 3]);
			}
			out[(size_t)r * d + kk] = s[src];
		}
	}
}

void FlatIndex::grow(int64_t need, hipStream_t st) {
	if (need <= cap)
		return;
	// doubling for chunked ingest (every growth re-allocates and copies all rows: 19 growths by 1.5 were 35-49 ms of a 10 M row
	// ingest's 400), exactly `need` for an add larger than that.  Round 6: by FOUR while the store is below 2 GB (a growth costs ~ 3 ms of
	// hipMalloc under faiss_lock whatever its size: 14 of them were 39 of a 10 M-row ingest's 275 ms; 288 GB of HBM can afford the slack)
	const int64_t row_bytes = (int64_t)geom.dp * 4;
	const int64_t nc = std::max<int64_t>(need, (cap * row_bytes < ((int64_t)2 << 30) ? 4 : 2) * cap + 4096);
	float *nv = nullptr, *nn = nullptr;
	// +64 floats: the LDS-DMA staging reads whole 64-float pieces and may run past the last row
	MVS_HIP(hipMalloc((void **)&nv, ((size_t)nc * geom.dp + 64) * sizeof(float)));
	MVS_HIP(hipMalloc((void **)&nn, ((size_t)nc + 64) * sizeof(float))); // + 64: the prefilter stages norms past the end
	// (rows staged but not flushed yet -- pend_rows -- are not on the device: only what is there is copied)
	const int64_t have = ntotal - pend_rows;
	if (have > 0) {
		MVS_HIP(hipMemcpyAsync(nv, vecs, (size_t)have * geom.dp * sizeof(float), hipMemcpyDeviceToDevice, st));
		MVS_HIP(hipMemcpyAsync(nn, norms, (size_t)have * sizeof(float), hipMemcpyDeviceToDevice, st));
	}
	// round 6: the old buffers are freed when the copy out of them has finished -- checked at the next growth, flush or reader, never
	// waited for here (round 5: a stream synchronisation + two hipFree per growth, 4 ms each, 10 % of a 10 M-row ingest under faiss_lock)
	retire_buffers(st, vecs, norms);
	vecs = nv;
	norms = nn;
	cap = nc;
}
void FlatIndex::retire_buffers(hipStream_t st, void *a, void *b) {
	// (not reaped here: hipFree waits for the whole device -- during an ingest that is a stall under faiss_lock; readers reap)
	if (!a && !b)
		return;
	Retired r;
	r.a = a, r.b = b;
	MVS_HIP(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
	MVS_HIP(hipEventRecord(r.done, st));
	retired.push_back(r);
}
void FlatIndex::reap_retired(bool wait) {
	for (size_t i = 0; i < retired.size();) {
		Retired &r = retired[i];
		if (wait)
			(void)hipEventSynchronize(r.done);
		else if (hipEventQuery(r.done) != hipSuccess) {
			++i;
			continue;
		}
		(void)hipEventDestroy(r.done);
		if (r.a)
			(void)hipFree(r.a);
		if (r.b)
			(void)hipFree(r.b);
		retired.erase(retired.begin() + (long)i);
	}
}
// Ingest staging (round 6).  The glue adds <= 2048 rows per call under faiss_lock
// (src/faiss_extension.cpp:475-547); round 5 paid a staging copy, an H2D copy, two kernel launches and an event record per call
// (39-60 us per 1 MB: 13 GB/s of a 57 GB/s link).  Now a call only copies its rows behind the previous call's in the current pinned slot
// (8 MB); the H2D copy, pack_rows and the norms run ONCE per slot -- when it is full, or when anything wants to read the rows.
bool FlatIndex::flush_adds() {
	if (pend_slot < 0)
		return false;
	TraceRange tr("mvs:stage_rows (H2D + pack_rows + norms of one pinned slot)");
	const bool sent = pend_rows > 0;
	if (pend_rows > 0) {
		ws_add.reserve((size_t)2 * PinnedRing::SLOT_BYTES);
		float *raw = (float *)((char *)ws_add.p + (size_t)add_flip * PinnedRing::SLOT_BYTES);
		add_flip ^= 1;
		MVS_HIP(hipMemcpyAsync(raw, add_ring.buf[pend_slot], pend_bytes, hipMemcpyHostToDevice, stream));
		add_ring.release(pend_slot, stream);
		launch_pack_rows(geom, raw, pend_rows, vecs + (size_t)pend_row0 * geom.dp, pend_row0, stream);
		launch_query_norms(raw, pend_rows, d, norms + pend_row0, stream);
		++add_flushes;
	}
	pend_slot = -1;
	pend_bytes = 0;
	pend_rows = 0;
	return sent;
}

// faiss::IndexFlatCodes::add: append n*d floats  (src/faiss_extension.cpp:512,609)
void FlatIndex::add(int64_t n, const float *x) {
	use_device();
	if (n <= 0)
		return;
	if (ntotal + n > (int64_t)0x7fffffff - 1024)
		throw_faiss("mvs::FlatIndex::add", __FILE__, "a single-device shard holds at most 2^31 rows");
	grow(ntotal + n, stream);
	const size_t nbytes = (size_t)n * d * sizeof(float);
	if (lazy_adds && nbytes <= PinnedRing::SLOT_BYTES / 2) {
		// a DataChunk-sized add: behind the rows already staged in the current slot; the device side runs once per slot (flush_adds)
		if (pend_slot >= 0 && pend_bytes + nbytes > PinnedRing::SLOT_BYTES)
			flush_adds();
		if (pend_slot < 0) {
			pend_slot = add_ring.acquire(PinnedRing::SLOT_BYTES);
			pend_bytes = 0, pend_rows = 0, pend_row0 = ntotal;
		}
		stage_copy_mt((char *)add_ring.buf[pend_slot] + pend_bytes, x, nbytes);
		pend_bytes += nbytes;
		pend_rows += n;
		ntotal += n;
		if (pend_bytes + nbytes > PinnedRing::SLOT_BYTES) // (no room for another chunk of this size: the copy engine takes the slot
			flush_adds();                                 // now, while the callers stage the next one)
		return;
	}
	flush_adds();
	// pinned staging in slots of <= SLOT_BYTES (the caller's buffer is free again when we return); each slot is
	// copied H2D into a raw device buffer and re-laid out into the storage format by pack_rows
	const int64_t rows_per_slot = std::max<int64_t>(1, (int64_t)PinnedRing::SLOT_BYTES / ((int64_t)d * 4));
	ws_add.reserve((size_t)2 * PinnedRing::SLOT_BYTES);
	for (int64_t r0 = 0; r0 < n; r0 += rows_per_slot) {
		const int64_t nr = std::min(rows_per_slot, n - r0);
		const size_t bytes = (size_t)nr * d * sizeof(float);
		const int slot = add_ring.acquire(bytes);
		stage_copy_mt(add_ring.buf[slot], x + r0 * d, bytes);
		float *raw = (float *)((char *)ws_add.p + (size_t)add_flip * PinnedRing::SLOT_BYTES);
		add_flip ^= 1;
		MVS_HIP(hipMemcpyAsync(raw, add_ring.buf[slot], bytes, hipMemcpyHostToDevice, stream));
		add_ring.release(slot, stream);
		launch_pack_rows(geom, raw, nr, vecs + (size_t)(ntotal + r0) * geom.dp, ntotal + r0, stream);
		launch_query_norms(raw, nr, d, norms + ntotal + r0, stream);
	}
	ntotal += n;
}

void FlatIndex::add_device(int64_t n, const float *d_x, hipStream_t st) {
	use_device();
	if (n <= 0)
		return;
	flush_adds();
	stream_wait(st, stream); // earlier host-API adds live on our own stream
	grow(ntotal + n, st);
	launch_pack_rows(geom, d_x, n, vecs + (size_t)ntotal * geom.dp, ntotal, st);
	launch_query_norms(d_x, n, d, norms + ntotal, st);
	stream_wait(stream, st); // later host-API calls see these rows
	ntotal += n;
}

__global__ void fill_results_kernel(float *D, long long *I, long long total, float v) {
	long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total) {
		D[i] = v;
		I[i] = -1;
	}
}

// L1, Linf, Lp, Canberra, BrayCurtis, JensenShannon, Jaccard [IndexFlat::search -> knn_extra_metrics,
// faiss/utils/extra_distances.cpp]: per-pair value on the LDS-staged flat_direct kernel (one instance per metric), the
// partial lists merged under the metric's order.  A cold path of the reference (the glue only forwards the metric
// name); no MFMA shape exists for these, the kernel is VALU-bound (|x-y| chains) or transcendental-bound (Lp, JS).
// Jaccard is a similarity: CMin lists like inner product, exact ties at the k-th score resolved in the pure order.
void FlatIndex::search_extra_metric(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
                                    const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st) {
	if (k > flat_direct_max_k())
		throw_faiss("mvs::FlatIndex::search", __FILE__, "k = %lld exceeds the supported maximum %lld", (long long)k,
		            (long long)flat_direct_max_k());
	const int om = metric_order(metric);
	const int64_t *out_map = raw_rows ? nullptr : d_idmap;
	const int64_t out_off = raw_rows ? 0 : label_offset;
	FlatDB db {vecs, norms, ntotal};
	memset(&kinfo, 0, sizeof kinfo);
	DirectPlan p = plan_flat_direct_extra(geom, nq, ntotal, k);
	const int64_t nq_pad = (nq + p.qgroup - 1) / p.qgroup * p.qgroup;
	ws_q.reserve((size_t)nq_pad * geom.dp * sizeof(float));
	MVS_HIP(hipMemsetAsync(ws_q.p, 0, (size_t)nq_pad * geom.dp * sizeof(float), st));
	launch_pad_rows(d_x, nq, d, (float *)ws_q.p, geom.dp, st);
	ws_pd.reserve((size_t)p.nsplit * nq * k * sizeof(float));
	ws_pi.reserve((size_t)p.nsplit * nq * k * sizeof(int32_t));
	SelectorDev sel = selector.upload(params, st);
	begin_kernel_timing(st);
	ws_gthr.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
	launch_init_slots((unsigned *)ws_gthr.p, nq, k, om, st);
	launch_flat_direct_extra(geom, p, metric, metric_arg, d, (const float *)ws_q.p, nq, db, k, sel, d_idmap,
	                         (float *)ws_pd.p, (int32_t *)ws_pi.p, (unsigned *)ws_gthr.p, st);
	end_kernel_timing(st);
	launch_merge_partials(om, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, p.nsplit, nq, k, out_map, out_off, d_D,
	                      d_I, st, k, nullptr);
	snprintf(kinfo.name, sizeof kinfo.name, "flat_direct_kernel (metric %d)", metric);
	const int64_t ngroups = (nq + p.qgroup - 1) / p.qgroup;
	kinfo.flops = 3.0 * (double)nq * (double)ntotal * d;
	kinfo.bytes = (double)ngroups * (double)ntotal * geom.dp * 4.0;
	kinfo.grid = p.grid;
	kinfo.block = 256;
	kinfo.lds_bytes = (int)p.lds_bytes;
	kinfo.nsplit = p.nsplit;
}

// IndexFlat::search dispatch (faiss/IndexFlat.cpp, utils/distances.cpp):
//   sel || nq < 20 -> per-pair arithmetic (flat_direct.hip); else BLAS-branch arithmetic (flat_mfma.hip)
void FlatIndex::search_flat(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
                            const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st) {
	use_device();
	TraceRange tr("mvs:flat_search");
	if (!retired.empty())
		reap_retired(false);
	if (flush_adds()) // (rows staged by add() reach the device before anything reads them)
		stream_wait(st, stream);
	if (k <= 0)
		throw_faiss("virtual void faiss::IndexFlat::search(...) const", "faiss/IndexFlat.cpp", "Error: 'k > 0' failed");
	if (nq <= 0)
		return;
	if (ntotal == 0) {
		const long long total = (long long)nq * k;
		hipLaunchKernelGGL(fill_results_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_D,
		                   (long long *)d_I, total, metric_order(metric) == METRIC_L2 ? FLT_MAX : -FLT_MAX);
		return;
	}
	// the scratch buffers below are shared by all searches of this index: order this call after the last one
	if (have_last_search) // (a null last_search_stream is HIP's NULL stream, a legitimate caller stream)
		stream_wait(st, last_search_stream);
	last_search_stream = st;
	have_last_search = true;
	if (metric_is_extra(metric)) {
		search_extra_metric(nq, d_x, k, d_D, d_I, params, d_idmap, st);
		return;
	}
	const bool has_sel = params && params->sel_kind != MVS_SEL_NONE;
	const int64_t mfma_kmax = flat_mfma_max_k(geom);
	// Inner product, exact ties at the k-th score: FAISS's CMin heap keeps/evicts equal scores by
	// arrival order.  The kernels keep the pure order (score desc, row id asc) -- a function of the data alone, so
	// it merges across splits -- with ONE extra entry per list; the merge flags the queries whose k-th and (k+1)-th
	// scores are bit-equal and resolve_ip_ties() replays the heap's outcome for those (usually none).
	const int64_t k_user = k;
	// (k >= 100: FAISS's reservoir instead of its heap -- resolve_ip_ties replays that; any kernel may serve the k + 1 search)
	const bool tie_detect = metric == METRIC_IP && ip_exact_ties && !raw_rows && ntotal > k &&
	                        (k + 1 <= mfma_kmax || (k >= 100 && k + 1 <= std::min(flat_direct_max_k(), reservoir_replay_max_k())));
	const int64_t *out_map = raw_rows ? nullptr : d_idmap; // label translation of the merge
	const int64_t out_off = raw_rows ? 0 : label_offset;
	if (tie_detect)
		k = k + 1;
	TieFlags fl = {nullptr, nullptr, nullptr, nullptr};
	if (tie_detect) {
		// raw candidate lists of the flagged queries: k entries each, or the prefilter's k + margin (search_prefilter)
		const size_t kflag = (size_t)(k + std::max<int64_t>(16, k / 2));
		ws_flag.reserve(16 + (size_t)nq * 4 + (size_t)nq * kflag * 8);
		fl.count = (int *)ws_flag.p;
		fl.query = fl.count + 4;
		fl.val = (float *)(fl.query + nq);
		fl.row = (int *)(fl.val + (size_t)nq * kflag);
		MVS_HIP(hipMemsetAsync(fl.count, 0, sizeof(int), st));
	}
	const TieFlags *flp = tie_detect ? &fl : nullptr;
	// Inner product + selector: FAISS's per-pair fvec_inner_product is the k-ordered chain the MFMA computes, so the
	// filtered search -- the reference's signature feature, on its default metric -- stays on the fused kernel, which
	// masks the rejected rows in its epilogue.  (L2 + selector is Sum (x-y)^2 per pair: packed scan kernel.)
	// The same identity covers small inner-product batches (nq < 20, FAISS's per-pair branch): from 8 queries on the
	// MFMA kernel beats the per-pair kernels even with a mostly empty 128-query block.
	// (queries re-run for the prefilter belong to a batch FAISS sends down its BLAS branch, however few they are)
	const bool small_batch = pf_suppressed ? pf_pair_branch : nq < 20; // (a re-run inherits the branch of its batch)
	const bool ip_on_mfma = metric == METRIC_IP && (has_sel || small_batch) && nq >= 8 &&
	                        k <= (has_sel ? flat_mfma_max_k_lds(geom) : mfma_kmax) && !force_direct && !force_staged;
	const bool direct = ((has_sel || small_batch) && !ip_on_mfma) || k > mfma_kmax || force_direct;
	FlatDB db {vecs, norms, ntotal};
	memset(&kinfo, 0, sizeof kinfo);
	// L2 + selector, or a small batch (FAISS's per-pair branch, nq < 20) on a large database: the bf16 coarse filter masks by
	// the selector and re-scores its candidates with the per-pair arithmetic FAISS uses there (csrc/flat_collect.hip); same
	// results as the packed scan / staged kernels
	// (inner product + selector beyond the fused kernel's LDS lists -- k in the dozens and above -- likewise, instead of the per-pair scan)
	if (((has_sel && (metric == METRIC_L2 || !ip_on_mfma)) || small_batch) && !force_direct && !force_staged &&
	    search_prefilter(nq, d_x, k_user, k, d_D, d_I, params, d_idmap, out_map, out_off, flp, st)) {
		// handled
	} else if (k > mfma_kmax && !has_sel && !small_batch && !force_direct && !force_staged &&
	           search_prefilter(nq, d_x, k_user, k, d_D, d_I, params, d_idmap, out_map, out_off, flp, st)) {
		// (round 6) lists beyond the fused kernel's: the coarse filter's big-list path took the batch (FlatIndex::collect_candidates "bigk")
	} else if (direct) {
		if (k > flat_direct_max_k())
			throw_faiss("mvs::FlatIndex::search", __FILE__, "k = %lld exceeds the supported maximum %lld",
			            (long long)k, (long long)flat_direct_max_k());
		// BLAS-branch value when FAISS would have used sgemm (nq >= 20, no selector) but k is too large for
		// the fused kernel's LDS lists
		const bool formula = metric == METRIC_L2 && !has_sel && !small_batch;
		// 1-4 queries are pure streaming: the LDS-staged kernel (coalesced tiles, 1 or 4 chains per thread) reaches
		// 3.2 TB/s there, the thread-per-row scan 1.6; from ~8 queries on the packed scan wins (3x at nq = 2000)
		if (!formula && !force_staged && nq > 4 && ivf_scan_supported(geom.dp, k)) {
			// per-pair arithmetic (exhaustive_L2sqr_seq / exhaustive_inner_product_seq): packed-fp32 scan kernel of
			// csrc/ivf_scan.hip in grid mode
			const int ngroups = (int)((nq + 19) / 20);
			const int64_t ntiles = (ntotal + 255) / 256;
			int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(4096 / ngroups, ntiles / 4));
			while (nsplit > 1 && (size_t)(nsplit + 1) * k * 8 > 150 * 1024) // the merge kernel keeps nsplit*k candidates in LDS
				nsplit /= 2;
			const int64_t split_rows = (ntiles + nsplit - 1) / nsplit * 256;
			nsplit = (ntotal + split_rows - 1) / split_rows;
			ws_q.reserve((size_t)nq * geom.dp * sizeof(float));
			launch_pad_rows(d_x, nq, d, (float *)ws_q.p, geom.dp, st);
			ws_pd.reserve((size_t)nsplit * nq * k * sizeof(float));
			ws_pi.reserve((size_t)nsplit * nq * k * sizeof(int32_t));
			ws_xi.reserve(ivf_scan_query_pack_bytes(geom.dp, ngroups));
			SelectorDev sel = selector.upload(params, st);
			begin_kernel_timing(st);
			ws_gthr.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
			launch_init_slots((unsigned *)ws_gthr.p, nq, k, metric, st);
			launch_pair_scan(geom.dp, geom.pair_interleaved, metric, (const float *)ws_q.p, nq, vecs, ntotal, k, (int)nsplit,
			                 split_rows, sel, d_idmap, (float *)ws_pd.p, (int32_t *)ws_pi.p, (unsigned *)ws_gthr.p,
			                 (float *)ws_xi.p, st);
			end_kernel_timing(st);
			launch_merge_partials(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, (int)nsplit, nq, k, out_map,
			                      out_off, d_D, d_I, st, k_user, flp);
			snprintf(kinfo.name, sizeof kinfo.name, "flat_pair_scan (ivf_scan_kernel)");
			kinfo.flops = 2.0 * (double)nq * (double)ntotal * d * (metric == METRIC_L2 ? 1.5 : 1.0);
			kinfo.bytes = (double)ngroups * (double)ntotal * geom.dp * 4.0;
			kinfo.grid = (int)(nsplit * ngroups);
			kinfo.block = 256;
			kinfo.lds_bytes = (int)ivf_scan_lds_bytes(k);
			kinfo.nsplit = (int)nsplit;
			if (tie_detect)
				resolve_ip_ties(nq, d_x, k_user, fl, sel, d_idmap, d_D, d_I, st);
			return;
		}
		DirectPlan p = plan_flat_direct(geom, nq, ntotal, k);
		const int64_t nq_pad = (nq + p.qgroup - 1) / p.qgroup * p.qgroup;
		ws_q.reserve((size_t)nq_pad * geom.dp * sizeof(float));
		MVS_HIP(hipMemsetAsync(ws_q.p, 0, (size_t)nq_pad * geom.dp * sizeof(float), st));
		launch_pad_rows(d_x, nq, d, (float *)ws_q.p, geom.dp, st);
		float *qn = nullptr;
		if (formula) {
			ws_qn.reserve((size_t)nq * sizeof(float));
			qn = (float *)ws_qn.p;
			launch_row_norms((const float *)ws_q.p, nq, geom.dp, qn, st);
		}
		const int nparts = p.nsplit; // the 4 per-wave lists are merged inside the workgroup
		ws_pd.reserve((size_t)nparts * nq * k * sizeof(float));
		ws_pi.reserve((size_t)nparts * nq * k * sizeof(int32_t));
		SelectorDev sel = selector.upload(params, st);
		begin_kernel_timing(st);
		ws_gthr.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
		launch_init_slots((unsigned *)ws_gthr.p, nq, k, metric, st);
		launch_flat_direct_ex(geom, p, metric, formula, (const float *)ws_q.p, qn, nq, db, k, sel, d_idmap,
		                      (float *)ws_pd.p, (int32_t *)ws_pi.p, (unsigned *)ws_gthr.p, st);
		end_kernel_timing(st);
		launch_merge_partials(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, nparts, nq, k, out_map,
		                      out_off, d_D, d_I, st, k_user, flp);
		if (tie_detect)
			resolve_ip_ties(nq, d_x, k_user, fl, sel, d_idmap, d_D, d_I, st);
		snprintf(kinfo.name, sizeof kinfo.name, "flat_direct_kernel");
		const int64_t ngroups = (nq + p.qgroup - 1) / p.qgroup;
		kinfo.flops = 2.0 * (double)nq * (double)ntotal * d * (metric == METRIC_L2 && !formula ? 1.5 : 1.0);
		kinfo.bytes = (double)ngroups * (double)ntotal * geom.dp * 4.0;
		kinfo.grid = p.grid;
		kinfo.block = 256;
		kinfo.lds_bytes = (int)p.lds_bytes;
		kinfo.nsplit = p.nsplit;
	} else if (search_prefilter(nq, d_x, k_user, k, d_D, d_I, params, d_idmap, out_map, out_off, flp, st)) {
		// bf16x3 prefilter + exact re-scoring took the batch (csrc/flat_bf16.hip); same results
	} else {
		FlatSearchPlan p = plan_flat_mfma(geom, nq, ntotal, k);
		ws_q.reserve(qfrag_floats(geom, nq) * sizeof(float));
		ws_qn.reserve((size_t)nq * sizeof(float));
		launch_pack_queries(geom, d_x, nq, (float *)ws_q.p, (float *)ws_qn.p, st);
		ws_pd.reserve((size_t)p.nsplit * nq * k * sizeof(float));
		ws_pi.reserve((size_t)p.nsplit * nq * k * sizeof(int32_t));
		ws_gthr.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64); // shared threshold slots
		begin_kernel_timing(st);
		SelectorDev sel = selector.upload(params, st);
		launch_flat_mfma(geom, p, metric, (const float *)ws_q.p, (const float *)ws_qn.p, nq, db, k, (float *)ws_pd.p,
		                 (int32_t *)ws_pi.p, (unsigned *)ws_gthr.p, st, &sel, d_idmap);
		end_kernel_timing(st);
		launch_merge_partials(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, p.nsplit, nq, k, out_map,
		                      out_off, d_D, d_I, st, k_user, flp);
		const int main_nsplit = p.nsplit, main_grid = p.grid;
		const size_t main_lds = p.lds_bytes;
		if (tie_detect)
			resolve_ip_ties(nq, d_x, k_user, fl, sel, d_idmap, d_D, d_I, st);
		snprintf(kinfo.name, sizeof kinfo.name, "flat_mfma_kernel");
		kinfo.flops = 2.0 * (double)nq * (double)ntotal * d;
		kinfo.bytes = (double)ntotal * d * 4.0 + (double)nq * d * 4.0 + (double)nq * k_user * 12.0;
		kinfo.grid = main_grid;
		kinfo.block = 256;
		kinfo.lds_bytes = (int)main_lds;
		kinfo.nsplit = main_nsplit;
	}
}

// BLAS-branch search through the bf16x3 prefilter (csrc/flat_bf16.hip).  Returns false when the shape is not served
// (the caller then runs the exact f32 kernel).  kk = k_user (+1 with inner-product tie detection).
bool FlatIndex::search_prefilter(int64_t nq, const float *d_x, int64_t k_user, int64_t kk, float *d_D, int64_t *d_I,
                                 const mvs_search_params *params, const int64_t *d_idmap, const int64_t *out_map,
                                 int64_t out_off, const TieFlags *flp, hipStream_t st) {
	// First pass: nothing between the scan and the re-scoring waits for the host (the candidate count stays on the device).  If the
	// count, read after the search's own synchronisation, shows that the stream overflowed, the search runs again the round-3 way:
	// count read back behind the scan, stream grown or the heavy queries taken out, scan repeated.
	bool overflow = false;
	const bool ok = search_prefilter_pass(nq, d_x, k_user, kk, d_D, d_I, params, d_idmap, out_map, out_off, flp, st, cl_defer, &overflow);
	if (!overflow)
		return ok;
	return search_prefilter_pass(nq, d_x, k_user, kk, d_D, d_I, params, d_idmap, out_map, out_off, flp, st, false, &overflow);
}
bool FlatIndex::search_prefilter_pass(int64_t nq, const float *d_x, int64_t k_user, int64_t kk, float *d_D, int64_t *d_I,
                                      const mvs_search_params *params, const int64_t *d_idmap, const int64_t *out_map,
                                      int64_t out_off, const TieFlags *flp, hipStream_t st, bool defer, bool *overflow) {
	*overflow = false;
	if (flp) // (a second pass must not see the tie flags the first one recorded from a truncated candidate set)
		MVS_HIP(hipMemsetAsync(flp->count, 0, sizeof(int), st));
	// 128 < d <= 1024: only the coarse filter exists (csrc/flat_collect_wide.hip; no bf16x3 behind it)
	const bool wide = collect_store_dims(d) > 128;
	// (16 < d <= 32: the coarse filter only, as for the wide stores)
	const bool cl_only = wide || !prefilter_supported(geom);
	// The coarse FILTER works with the user's k even when the search carries one entry more for the inner-product tie detection
	// (kk = k_user + 1): a row tied with the k-th score passes any bound derived from k rows -- the bound is on values -- so the
	// (k + 1)-th entry of the SELECTION is a tied row if there is one.  The bound is then the k-th, not the (k + 1)-th, best class,
	// and k = 32 / k = 16 stay on the 32- / 16-class instances (wide stores: k = 32 stays on the filter at all) -- ADVICE r3, low.
	const int64_t kf = kk > k_user ? k_user : kk;
	const int cl_kmax0 = cl_k32 ? collect_max_k(d) : std::min(16, collect_max_k(d)); // 32 row classes at d <= 128 (option cl_k32), else 16
	// (round 6: beyond 128 entries the d <= 128 store's filter takes its bounds from a pass of its own and selects by a segmented sort --
	// FlatIndex::collect_candidates "bigk"; option cl_bigk = 0: the exact kernels as before)
	const bool bigk_ok = cl_bigk && (!wide || collect_wide_max_classes(collect_store_dims(d)) >= 128) && collect_supported(geom) && kf > 128 - (kk - kf) && kk <= std::min<int64_t>(wide ? 2049 : 4097, flat_direct_max_k()) /* (what the fall-back can serve) */ && ntotal >= 65536 && ntotal >= 64 * kf &&
	                     ntotal < ((int64_t)1 << 31) && (prefilter_mode == 2 || prefilter_mode < 0) && nq * kk < ((int64_t)1 << 31);
	const int cl_kmax = bigk_ok ? 4096 : (int)std::min<int64_t>(cl_kmax0, 128 - (kk - kf)); // (collect_select_kernel: kk <= 128 entries)
	// (lists beyond 40: only the coarse filter of the d <= 128 store, up to 128 -- four subsets of 32 row classes, round 4)
	if (prefilter_mode == 0 || pf_suppressed || (!prefilter_supported(geom) && !collect_supported(geom)) || (kk > 40 && kf > cl_kmax))
		return false;
	// an IDSelector: only the coarse filter handles it (SEL instances); otherwise the exact kernels' SEL instances do
	const bool has_sel = params && params->sel_kind != MVS_SEL_NONE;
	if (cl_only && (kf > (bigk_ok ? 4096 : std::min<int64_t>(collect_max_k(d), 128 - (kk - kf))) || prefilter_mode == 1))
		return false;
	if (has_sel && !((prefilter_mode == 2 || prefilter_mode < 0) && collect_supported(geom) && kf <= cl_kmax))
		return false;
	// auto: the contraction must dominate.  The coarse filter wins from FAISS's first BLAS-branch batch on (N = 10M: 1.3 ms vs
	// 3.5 ms at 64 queries, 1.7 vs 11.0 at 500); the bf16x3 kernel needs whole 256-query blocks to pay
	const bool collect_ok = (prefilter_mode == 2 || prefilter_mode < 0) && collect_supported(geom) && kf <= cl_kmax;
	// (N = 131 072: 0.35 vs 0.46 ms at 64 queries, 1.36 vs 3.88 at 10k; N = 65 536: 1.40 vs 2.25 at 10k but 1.07 vs 0.61 at 2048;
	// below that the f32 kernel's ~0.3 ms wins everywhere)
	const bool big_enough = ntotal >= 262144 || (collect_ok && ntotal >= 65536 && (double)nq * (double)ntotal >= 5e8) ||
	                        (collect_ok && ntotal >= 131072 && nq <= 128);
	if (prefilter_mode < 0 && (!big_enough || (nq < 512 && !collect_ok)))
		return false;
	if (nq < 20 && !collect_ok) // (FAISS's per-pair branch: only the coarse filter re-scores in that arithmetic)
		return false;
	if (ntotal < 4096 || ntotal <= kk)
		return false;
	// Rows that cluster (round 5): through the shadow IVF index, when one is wanted (see FlatIndex::shadow_search)
	if (metric == METRIC_L2 && !has_sel && !flp && kk == k_user && nq >= 256 && shadow_mode != 0 && (shadow_state == 1 || shadow_mode == 1) &&
	    shadow_search(nq, d_x, k_user, d_D, d_I, params, d_idmap, out_map, out_off, st))
		return true;
	ws_fail.reserve(64 + (size_t)nq * sizeof(int));
	int *fail_cnt = (int *)ws_fail.p, *fail_q = fail_cnt + 16;
	MVS_HIP(hipMemsetAsync(fail_cnt, 0, sizeof(int), st));
	float *pd1 = nullptr;
	int32_t *pi1 = nullptr;
	int kp = 0;
	bool collected = false;
	// Coarse filter (one bf16 product per pair, candidates by a proven bound): mode 2 forces it, auto prefers it where
	// its kernel exists (d = 128 geometry, lists of <= 16); on a stream overflow the bf16x3 path below takes the batch
	if ((prefilter_mode == 2 || prefilter_mode < 0) && collect_supported(geom) && kf <= cl_kmax) {
		// Data on which the filter gave up (every query sits on thousands of copies of its nearest row: two scans that only count,
		// then the exact kernels anyway) is not asked again at once: the next 4, 8, ... 64 large searches of this index go straight to
		// the fall-back (all-duplicates, N = 2 M: 1.3 s per batch with the two futile scans, 0.07 s without)
		if (cl_skip > 0 && nq >= 20 && !has_sel && !cl_only && kk <= 40) {
			--cl_skip;
		} else {
			kp = (int)kk;
			// (collect_candidates may write the final lists itself: L2 as they are, inner product with the tie flags of this search)
			const bool direct = (metric == METRIC_L2 && !flp && kk == k_user) || (metric == METRIC_IP && kk <= k_user + 1 && (flp || kk == k_user));
			cl_out_D = direct ? d_D : nullptr, cl_out_I = direct ? d_I : nullptr, cl_out_map = out_map, cl_out_off = out_off;
			cl_out_flags = flp, cl_out_kout = (int)k_user;
			collected = collect_candidates(nq, d_x, kp, &pd1, &pi1, fail_cnt, fail_q, params, d_idmap, st, defer, (int)(bigk_ok ? std::max<int64_t>(kf, 129) : kf));
			cl_out_D = nullptr, cl_out_I = nullptr, cl_out_flags = nullptr;
			if (!collected) {
				MVS_HIP(hipMemsetAsync(fail_cnt, 0, sizeof(int), st));
				cl_skip_len = std::min(64, std::max(4, 2 * cl_skip_len));
				cl_skip = cl_skip_len;
			} else if (!defer) {
				cl_skip_len = 0;
			}
		}
	}
	if (!collected && (has_sel || nq < 20 || cl_only || kk > 40))
		return false; // (stream overflow under a selector / in the per-pair branch / beyond the bf16x3 lists: the exact kernels take the batch)
	if (!collected) {
	// candidates per query (<= 64: one lane each in the proof).  The margin sets how often a query cannot be proven: at the
	// headline (N = 10M, d = 128) 5 spare ranks leave ~3 of 10 000 queries to the exact kernel, 8 spare ranks ~none
	kp = (int)(kk + std::max<int64_t>(pf_margin, kk / 2));
	if (kp > 16 && kk + 5 <= 16)
		kp = 16; // one 16-slot window of shared thresholds: a second window costs more than the lost margin (measured: 61 vs 70 ms)
	ensure_bf16_rows(st);
	FlatSearchPlan p = plan_prefilter(geom, nq, ntotal, kp);
	ws_pfq.reserve(prefilter_qfrag_bytes(geom, nq));
	ws_qn.reserve((size_t)nq * sizeof(float));
	launch_pack_queries_bf16(geom, d_x, nq, ws_pfq.p, st);
	launch_query_norms(d_x, nq, d, (float *)ws_qn.p, st);
	ws_pd.reserve((size_t)p.nsplit * nq * kp * sizeof(float));
	ws_pi.reserve((size_t)p.nsplit * nq * kp * sizeof(int32_t));
	ws_gthr.reserve((size_t)nq * std::max(32, (kp + 15) / 16 * 16) * sizeof(unsigned) + 64);
	begin_kernel_timing(st);
	launch_prefilter(geom, p, metric, ws_pfq.p, (const float *)ws_qn.p, nq, vecs_bf, norms, ntotal, kp, (float *)ws_pd.p,
	                 (int32_t *)ws_pi.p, (unsigned *)ws_gthr.p, st);
	end_kernel_timing(st);
	// approximate top-kp per query (plain row numbers)
	const size_t ca_bytes = ((size_t)nq * kp * sizeof(float) + 255) & ~(size_t)255;
	ws_cand.reserve(ca_bytes + (size_t)nq * kp * sizeof(int64_t));
	float *ca = (float *)ws_cand.p;
	int64_t *ci = (int64_t *)((char *)ws_cand.p + ca_bytes);
	launch_merge_partials(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, p.nsplit, nq, kp, nullptr, 0, ca, ci, st);
	// exact values of the candidates + the per-query proof
	const size_t ex_bytes = ((size_t)nq * kp * sizeof(float) + 255) & ~(size_t)255;
	ws_ex.reserve(ex_bytes + (size_t)nq * kp * sizeof(int32_t));
	pd1 = (float *)ws_ex.p;
	pi1 = (int32_t *)((char *)ws_ex.p + ex_bytes);
	launch_rescore_verify(metric, ca, ci, nq, kp, (int)kk, d_x, geom, vecs, norms, (const float *)ws_qn.p, d_max_norm_bits,
	                      pd1, pi1, fail_cnt, fail_q, d_max_norm_bits + 4, st);
	snprintf(kinfo.name, sizeof kinfo.name, "flat_bf16x3_kernel");
	kinfo.flops = 2.0 * (double)nq * (double)ntotal * d;
	kinfo.bytes = (double)ntotal * d * 4.0 + (double)nq * d * 4.0 + (double)nq * k_user * 12.0;
	kinfo.grid = p.grid;
	kinfo.block = 256;
	kinfo.lds_bytes = (int)p.lds_bytes;
	kinfo.nsplit = p.nsplit;
	}
	// the exact candidates through the normal merge: FAISS order, labels, inner-product tie flags
	const bool emitted = collected && cl_emitted; // (the bucketed finish printed the lists already)
	if (emitted)
		;
	else if (collected && metric == METRIC_L2 && !flp) // (the coarse filter's selection is in FAISS's L2 order already: labels only)
		launch_emit_sorted(pd1, pi1, nq, kp, k_user, out_map, out_off, d_D, d_I, st);
	else
		launch_merge_partials(metric, pd1, pi1, 1, nq, kp, out_map, out_off, d_D, d_I, st, k_user, flp, collected /* the coarse filter's selection is sorted */);
	if (!h_flag_count)
		MVS_HIP(hipHostMalloc((void **)&h_flag_count, 64, hipHostMallocDefault));
	// (one kernel writes fail count, rounding residual and -- deferred count mode -- the scan's entry count / the bucket header to pinned memory)
	launch_collect_report(collected ? ws_seg.p : nullptr, fail_cnt, d_max_norm_bits + 4, h_flag_count,
	                      ((emitted || (collected && cl_wrf_used)) && cl_report_cnt) ? h_cl_hdr : nullptr, collected && cl_report_cnt, st);
	if (flp) {
		if (collected && defer && cl_deferred_cap > 0) {
			// (the sort covered cl_deferred_cap entries -- if the scan produced more, the tie pass below would work on a
			// truncated candidate set, for k >= 100 with a full score pass per flagged query, only for the whole search to be repeated)
			MVS_HIP(hipStreamSynchronize(st));
			unsigned long long ncand_u;
			memcpy(&ncand_u, h_flag_count + 10, sizeof ncand_u);
			if ((int64_t)ncand_u > cl_deferred_cap) {
				cl_last_candidates = (int64_t)ncand_u;
				cl_est_per_query = (double)ncand_u / (double)std::max<int64_t>(nq, 1) + 1e-6;
				*overflow = true;
				return false;
			}
		}
		SelectorDev tsel;
		memset(&tsel, 0, sizeof tsel);
		if (has_sel)
			tsel = selector.upload(params, st);
		// (after the coarse filter the rows at or above the boundary score all are candidates: no second pass over the database)
		tie_sorted = collected && tie_from_candidates ? cl_sorted : nullptr;
		tie_bucket = (emitted && tie_from_candidates && k_user < 100) ? cl_fb_keys : nullptr; // (k >= 100: the reservoir replay reads scores, not A_k)
		try {
			resolve_ip_ties(nq, d_x, k_user, *flp, tsel, d_idmap, d_D, d_I, st, kp); // (syncs the stream)
		} catch (...) {
			tie_sorted = nullptr, tie_bucket = nullptr;
			throw;
		}
		tie_sorted = nullptr, tie_bucket = nullptr;
	} else {
		MVS_HIP(hipStreamSynchronize(st));
	}
	if (emitted) { // (synchronised above) a bucket too small for some query: its list is incomplete -- the sorted pipeline takes over for good
		unsigned long long bmax = 0, kept = 0;
		memcpy(&bmax, (const char *)h_cl_hdr + 200, sizeof bmax);
		memcpy(&kept, (const char *)h_cl_hdr + 8, sizeof kept);
		cl_last_rescored = (int64_t)kept; // (survivors of the final-bound filter: what the exact stage re-scored)
		if ((int64_t)bmax > cl_fpitch) { // (the synchronous pass grows the pitch itself from there on)
			const int64_t want = ((int64_t)bmax + (int64_t)bmax / 4 + 63) / 64 * 64;
			if (want > 16384 || nq * want >= ((int64_t)1 << 31))
				cl_fbucket_off = true;
			else
				cl_fpitch = (int)want;
			*overflow = true;
			return false;
		}
	}
	if (collected && defer && cl_deferred_cap > 0) { // the stream synchronised above: the candidate count of the scan is on the host now
		unsigned long long ncand_u;
		memcpy(&ncand_u, h_flag_count + 10, sizeof ncand_u);
		cl_last_candidates = (int64_t)ncand_u;
		cl_est_per_query = (double)ncand_u / (double)std::max<int64_t>(nq, 1) + 1e-6;
		if ((int64_t)ncand_u > cl_deferred_cap) {
			*overflow = true; // (more entries than the sort covered: the synchronous pass overwrites the results written so far)
			return false;
		}
		cl_queries_total += nq;
		cl_candidates_total += (int64_t)ncand_u;
	}
	if (collected && cl_wrf_used && !emitted) { // (the wide stores' filter: survivors in the control block's header @8)
		unsigned long long kept = 0;
		memcpy(&kept, (const char *)h_cl_hdr + 8, sizeof kept);
		cl_last_rescored = (int64_t)kept;
	}
	if ((emitted || (collected && cl_wrf_used)) && cl_last_rescored >= 0) // (mvs_index_collect_stats then reports what the exact stage re-scored, as for an IVF index)
		cl_rescored_total += cl_last_rescored, cl_rescored_queries += nq, cl_admitted_in_fb += cl_last_candidates;
	const int nf = h_flag_count[8];
	pf_last_fallback = nf;
	pf_queries_total += nq;
	pf_fallback_total += nf;
	// the global-centring filter admits thousands of rows per query: this data clusters (or is badly conditioned) -- the next large
	// search goes through a shadow IVF index of the rows (built then; FlatIndex::shadow_search)
	if (collected && shadow_state == 0 && shadow_mode < 0 && metric == METRIC_L2 && !has_sel && nq >= 256 && ntotal >= 262144 && d % 32 == 0 &&
	    d <= 128 && kk <= 32 && cl_last_candidates > 600 * nq)
		shadow_state = 1;
	memcpy(&pf_max_rel_err, h_flag_count + 9, sizeof(float));
	if (nf > 0) {
		// queries whose candidate set could not be proven complete: the exact kernel decides (results overwrite theirs)
		const mvs_kernel_info keep = kinfo;
		const size_t xf_bytes = ((size_t)nf * d * sizeof(float) + 255) & ~(size_t)255;
		const size_t df_bytes = ((size_t)nf * k_user * sizeof(float) + 255) & ~(size_t)255;
		ws_fb.reserve(xf_bytes + df_bytes + (size_t)nf * k_user * sizeof(int64_t));
		float *xf = (float *)ws_fb.p;
		float *Df = (float *)((char *)ws_fb.p + xf_bytes);
		int64_t *If = (int64_t *)((char *)Df + df_bytes);
		launch_gather_query_rows(d_x, d, fail_q, nf, xf, st);
		pf_suppressed = true;
		pf_pair_branch = nq < 20;
		const bool timing = timing_enabled;
		timing_enabled = false; // the bench's dominant kernel stays the prefilter launch
		try {
			search_flat(nf, xf, k_user, Df, If, params, d_idmap, st);
		} catch (...) {
			pf_suppressed = false;
			timing_enabled = timing;
			throw;
		}
		pf_suppressed = false;
		timing_enabled = timing;
		launch_scatter_rows(fail_q, nf, k_user, Df, If, d_D, d_I, st);
		kinfo = keep;
	}
	return true;
}

// ---- shadow clustering of a Flat L2 index (round 5) ------------------------------------------------------------------
// The coarse filter's error bound scales with ||x - mu|| ||y - mu|| for ONE centre mu.  Rows that form clusters far from each other
// (embeddings; the C3 mixture: 1024 centres of norm^2 ~ 128, neighbours at distance^2 ~ 2.6) leave 2E larger than the spread of
// distances inside the query's cluster: the whole cluster is admitted (9 764 candidates per query, 45.6 ms per 10k batch against
// 17.3 on uniform rows).  The cure is the IVF index's per-list centring -- so the Flat index keeps one: k-means over its rows
// (sqrt(N) lists), the rows list by list as bf16 residuals, and a large batch runs
//   1. the IVF coarse filter over the nprobe nearest lists (C3's kernels), candidates re-scored with THIS index's arithmetic
//      ((xn + yn) - 2 ip on the original rows, ties by row number) -- the result is exact IF no other list holds a better row;
//   2. the proof, per query, that none does: every row of list j is at least ||x - c_j|| - r_j away (ivf_shadow_verify_kernel);
//   3. the queries that cannot be proven (none on clustered rows; all on uniform rows) again on the Flat kernels.
// More than a tenth of a batch unproven: the data does not cluster, the shadow is dropped for good.  false: the batch takes the normal path.
void FlatIndex::drop_shadow() {
	delete shadow;
	shadow = nullptr;
	shadow_rows = -1;
	shadow_trained_rows = 0;
	shadow_gen = 0;
}
// Build (or extend) the shadow from the rows WHERE THEY ARE (round 6; round 5 copied every row to the host, re-laid it out there, trained
// and re-added from host memory inside the first large search, and dropped the whole clustering on every add()):
//   * training sample: min(N, 256 nlist) rows at an even stride, unpacked on the device, ONE copy to the host for the k-means driver
//     (csrc/ivf.hip kmeans keeps FAISS's host-side control flow); the sample has exactly the size FAISS would subsample to;
//   * rows: unpacked in 1 M-row blocks on the device and appended through the IVF index's device add -- no host round trip;
//   * rows added to the Flat index later are APPENDED (assign + append is the IVF add): the next large search extends the shadow by
//     rows [shadow_rows, ntotal); the clustering is re-trained only after the index has doubled since it was trained;
//   * reset() (and anything else that changes rows in place) bumps mut_gen: a shadow of another generation is dropped.
bool FlatIndex::shadow_sync(hipStream_t st) {
	if (shadow && (shadow_gen != mut_gen || shadow_rows > ntotal || ntotal > 2 * std::max<int64_t>(shadow_trained_rows, 1)))
		drop_shadow();
	if (shadow && shadow_rows == ntotal)
		return true;
	const auto t0 = std::chrono::steady_clock::now();
	MVS_HIP(hipStreamSynchronize(st));
	MVS_HIP(hipStreamSynchronize(stream)); // (adds of the host API)
	const int64_t blk = (int64_t)1 << 20;
	DevBuf rows;
	bool built = false;
	if (!shadow) {
		int lg = 0;
		while (((int64_t)1 << (2 * lg + 1)) < ntotal) // nlist = the power of two nearest to sqrt(N) (N = 10 M: 4 096)
			++lg;
		const int64_t nl = std::min<int64_t>(8192, std::max<int64_t>(256, (int64_t)1 << lg));
		std::unique_ptr<IndexBase> iv(make_ivf_index(d, "IVF" + std::to_string(nl) + ",Flat", METRIC_L2));
		if (!iv)
			return false;
		iv->adopt_tuning(tune_); // (the Flat index's knobs reach its shadow and the shadow's quantizer)
		const int64_t ns = std::min<int64_t>(ntotal, nl * 256), stride = std::max<int64_t>(1, ntotal / ns);
		rows.reserve((size_t)std::max(ns, std::min(blk, ntotal)) * d * sizeof(float));
		launch_unpack_rows(geom, vecs, 0, stride, ns, (float *)rows.p, st);
		std::vector<float> sample((size_t)ns * d);
		MVS_HIP(hipMemcpyAsync(sample.data(), rows.p, sample.size() * sizeof(float), hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		iv->train(ns, sample.data());
		shadow = iv.release();
		shadow_rows = 0;
		shadow_trained_rows = ntotal;
		shadow_gen = mut_gen;
		shadow->set_timing(timing_enabled);
		use_device();
		built = true;
	} else {
		rows.reserve((size_t)std::min(blk, ntotal - shadow_rows) * d * sizeof(float));
	}
	for (int64_t r0 = shadow_rows; r0 < ntotal; r0 += blk) {
		const int64_t n = std::min(blk, ntotal - r0);
		launch_unpack_rows(geom, vecs, r0, 1, n, (float *)rows.p, st);
		shadow->add_device(n, (const float *)rows.p, st); // (labels = row numbers: the shadow's ids continue where it stopped)
		use_device();
	}
	MVS_HIP(hipStreamSynchronize(st)); // (`rows` is freed at scope exit)
	shadow_rows = ntotal;
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	shadow_build_seconds += sec;
	if (built)
		++shadow_builds;
	else
		++shadow_extends;
	return true;
}
bool FlatIndex::shadow_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
                              const int64_t *d_idmap, const int64_t *out_map, int64_t out_off, hipStream_t st) {
	if (d % 32 != 0 || d > 128 || k > 32 || ntotal < 65536 || ntotal >= ((int64_t)1 << 31) || ntotal <= k)
		return false;
	if (!shadow_sync(st))
		return false;
	ensure_h1_rows(st); // (the proof needs the largest ||y||^2 of the rows: d_max_norm_bits[0], kept with the coarse filter's store)
	ws_qn.reserve((size_t)nq * sizeof(float));
	ws_fail.reserve(64 + (size_t)nq * sizeof(int));
	int *fail_cnt = (int *)ws_fail.p, *fail_q = fail_cnt + 16;
	launch_query_norms(d_x, nq, d, (float *)ws_qn.p, st);
	if (!h_flag_count)
		MVS_HIP(hipHostMalloc((void **)&h_flag_count, 64, hipHostMallocDefault));
	// (a batch the coarse quantiser's distance matrix or the pair count does not cover is served in pieces -- round 5 gave
	// the shadow up for good on the first such batch)
	const int64_t cover = shadow->flat_shadow_max_queries(shadow_nprobe);
	if (cover < 256)
		return false;
	mvs_kernel_info kacc;
	memset(&kacc, 0, sizeof kacc);
	int nf_total = 0;
	for (int64_t q0 = 0; q0 < nq; q0 += cover) {
		const int64_t m = std::min(cover, nq - q0);
		MVS_HIP(hipMemsetAsync(fail_cnt, 0, sizeof(int), st));
		const int rc = shadow->flat_shadow_search(m, d_x + q0 * d, k, (const float *)ws_qn.p + q0, d_D + q0 * k, d_I + q0 * k, out_map, out_off,
		                                          d_max_norm_bits, fail_cnt, fail_q, shadow_nprobe, st);
		use_device();
		if (rc != 0) {
			// 1: this CALL's shape is not served (too few queries in the piece, ...): the batch takes the normal path, the shadow stays;
			// 2: the candidate stream / a bucket beyond their limits -- rows stored thousands of times: the shadow cannot serve this data
			if (rc == 2) {
				shadow_state = -1;
				drop_shadow();
			}
			return false;
		}
		MVS_HIP(hipMemcpyAsync(h_flag_count + 8, fail_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		const int nf = h_flag_count[8];
		shadow_queries += m;
		shadow_unproven += nf;
		nf_total += nf;
		if (q0 == 0)
			kacc = shadow->kinfo;
		else
			kacc.flops += shadow->kinfo.flops, kacc.bytes += shadow->kinfo.bytes;
		if ((int64_t)nf * 10 > m) { // the lists of this data overlap: nothing is gained (uniform rows: every query) -- never again
			shadow_state = -1;
			drop_shadow();
			return false;
		}
		if (nf > 0) { // the unproven queries: the Flat kernels decide (results overwrite theirs)
			// Their batch went down FAISS's BLAS branch (nq >= 20: (xn + yn) - 2 ip); fewer than 20 of them on their own would take the
			// per-pair branch (sum (x - y)^2: other last bits -- round 5 returned those, found by round 6's extension test).  The re-run
			// is padded to 20 queries with copies of the first; only the first nf results are used.
			const int nr = nf < 20 ? 20 : nf;
			for (int i = nf; i < nr; ++i)
				MVS_HIP(hipMemcpyAsync(fail_q + i, fail_q, sizeof(int), hipMemcpyDeviceToDevice, st));
			const size_t xf_bytes = ((size_t)nr * d * sizeof(float) + 255) & ~(size_t)255;
			const size_t df_bytes = ((size_t)nr * k * sizeof(float) + 255) & ~(size_t)255;
			ws_fb.reserve(xf_bytes + df_bytes + (size_t)nr * k * sizeof(int64_t));
			float *xf = (float *)ws_fb.p;
			float *Df = (float *)((char *)ws_fb.p + xf_bytes);
			int64_t *If = (int64_t *)((char *)Df + df_bytes);
			launch_gather_query_rows(d_x + q0 * d, d, fail_q, nr, xf, st);
			const int keep_state = shadow_state, keep_mode = shadow_mode;
			shadow_state = 0, shadow_mode = 0; // (the re-run must not come back here)
			const bool timing = timing_enabled;
			timing_enabled = false;
			try {
				search_flat(nr, xf, k, Df, If, params, d_idmap, st);
			} catch (...) {
				shadow_state = keep_state, shadow_mode = keep_mode, timing_enabled = timing;
				throw;
			}
			shadow_state = keep_state, shadow_mode = keep_mode, timing_enabled = timing;
			launch_scatter_rows(fail_q, nf, k, Df, If, d_D + q0 * k, d_I + q0 * k, st);
			if (q0 + cover < nq)
				MVS_HIP(hipStreamSynchronize(st)); // (ws_fb / the fail list are reused by the next piece)
		}
	}
	kinfo = kacc;
	snprintf(kinfo.name, sizeof kinfo.name, "ivf_bf16_collect_kernel (flat shadow)");
	pf_last_fallback = nf_total;
	pf_queries_total += nq;
	pf_fallback_total += nf_total;
	return true;
}

// Tie pass of an inner-product search.  The flag count is the only host-visible decision of a search: one 4-byte
// D2H + stream sync per IP search (the search is >= 100 us of kernels).  Flagged queries are re-run through the SAME
// contraction (bit-identical scores) with the TIE epilogue, which collects per query the k smallest row ids whose
// score is >= the boundary score T; tie_resolve_kernel then applies FAISS's heap outcome (csrc/util_kernels.hip).
void FlatIndex::resolve_ip_ties(int64_t nq, const float *d_x, int64_t k, const TieFlags &fl, SelectorDev sel,
                                const int64_t *d_idmap, float *d_D, int64_t *d_I, hipStream_t st, int64_t kraw_in) {
	if (!h_flag_count)
		MVS_HIP(hipHostMalloc((void **)&h_flag_count, 64, hipHostMallocDefault));
	MVS_HIP(hipMemcpyAsync(h_flag_count, fl.count, sizeof(int), hipMemcpyDeviceToHost, st));
	MVS_HIP(hipStreamSynchronize(st));
	const int nf = *h_flag_count;
	if (nf <= 0)
		return;
	const int64_t kraw = kraw_in > 0 ? kraw_in : k + 1; // length of the raw candidate lists the merge recorded
	if (k >= 100) {
		// FAISS collects k >= 100 results in a ReservoirTopN, not a heap (utils/distances.cpp): which of the rows tied at the k-th
		// score survive depends on the whole stream -- replayed for the flagged queries (csrc/flat_reservoir.hip)
		const size_t xf_b = ((size_t)nf * d * sizeof(float) + 255) & ~(size_t)255, t_b = ((size_t)nf * sizeof(float) + 255) & ~(size_t)255;
		const size_t ov_b = ((size_t)nf * k * sizeof(float) + 255) & ~(size_t)255, oi_b = ((size_t)nf * k * sizeof(int64_t) + 255) & ~(size_t)255;
		const int64_t F = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nf, 32768), // (flagged queries ride in grid.y)
		                                                     ((int64_t)1 << 30) / std::max<int64_t>(ntotal * 4, 1)));
		ws_tie.reserve(xf_b + t_b + 2 * ov_b + 2 * oi_b + (size_t)F * ntotal * sizeof(float) + 256);
		char *b = (char *)ws_tie.p;
		float *xf = (float *)b, *T = (float *)(b + xf_b), *ov = (float *)(b + xf_b + t_b), *Df = (float *)(b + xf_b + t_b + ov_b);
		int32_t *oi = (int32_t *)(b + xf_b + t_b + 2 * ov_b);
		int64_t *If = (int64_t *)(b + xf_b + t_b + 2 * ov_b + oi_b);
		float *scores = (float *)(b + xf_b + t_b + 2 * ov_b + 2 * oi_b);
		launch_gather_flagged(d_x, d, fl, nf, kraw, k, xf, T, st);
		for (int64_t f0 = 0; f0 < nf; f0 += F) {
			const int m = (int)std::min<int64_t>(F, nf - f0);
			launch_reservoir_replay(xf + f0 * d, m, d, vecs, geom.dp, geom.pair_interleaved ? 1 : 0, ntotal, k, sel, d_idmap, T + f0, scores,
			                        ov + f0 * k, oi + f0 * k, st);
		}
		launch_merge_partials(METRIC_IP, ov, oi, 1, nf, k, d_idmap, label_offset, Df, If, st); // print order, labels
		launch_scatter_rows(fl.query, nf, k, Df, If, d_D, d_I, st);
		return;
	}
	const size_t xf_bytes = ((size_t)nf * d * sizeof(float) + 255) & ~(size_t)255;
	const size_t t_bytes = ((size_t)nf * sizeof(float) + 255) & ~(size_t)255;
	const size_t td_bytes = ((size_t)nf * k * sizeof(float) + 255) & ~(size_t)255;
	ws_tie.reserve(xf_bytes + t_bytes + td_bytes + (size_t)nf * k * sizeof(int64_t));
	float *xf = (float *)ws_tie.p;
	float *T = (float *)((char *)ws_tie.p + xf_bytes);
	float *tD = (float *)((char *)T + t_bytes);
	int64_t *tI = (int64_t *)((char *)tD + td_bytes);
	launch_gather_flagged(d_x, d, fl, nf, kraw, k, xf, T, st);
	(void)tD;
	if (tie_sorted)
		launch_collect_tie_rows(tie_sorted, (const int *)((const char *)ws_seg.p + 256), nq, fl.query, T, nf, (int)k, tI, st);
	else if (tie_bucket) // (round 6: the bucketed finish keeps every survivor's exact key in its query's bucket)
		launch_collect_tie_rows_bucket(tie_bucket, tie_bcount, tie_bpitch, fl.query, T, nf, (int)k, tI, st);
	else
		tie_candidates(nf, xf, T, k, tI, sel, d_idmap, st);
	launch_tie_resolve(fl, nf, kraw, k, tI, d_idmap, label_offset, d_D, d_I, st);
}

__global__ void offset_rows_kernel(long long *I, long long total, long long off) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total && I[i] >= 0)
		I[i] += off;
}
void FlatIndex::offset_rows(int64_t *d_rows, int64_t total, hipStream_t st) {
	if (total <= 0 || label_offset == 0)
		return;
	hipLaunchKernelGGL(offset_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (long long *)d_rows,
	                   (long long)total, (long long)label_offset);
}

void FlatIndex::tie_candidates(int64_t nf, const float *d_xf, const float *d_T, int64_t k, int64_t *d_rows_out,
                               SelectorDev sel, const int64_t *d_selmap, hipStream_t st) {
	use_device();
	flush_adds();
	if (nf <= 0)
		return;
	if (ntotal == 0) {
		MVS_HIP(hipMemsetAsync(d_rows_out, 0xff, (size_t)nf * k * sizeof(int64_t), st));
		return;
	}
	FlatSearchPlan p = plan_flat_mfma(geom, nf, ntotal, k);
	ws_q.reserve(qfrag_floats(geom, nf) * sizeof(float));
	launch_pack_queries(geom, d_xf, nf, (float *)ws_q.p, nullptr, st);
	ws_pd.reserve((size_t)p.nsplit * nf * k * sizeof(float));
	ws_pi.reserve((size_t)p.nsplit * nf * k * sizeof(int32_t));
	ws_gthr.reserve((size_t)nf * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
	ws_qn.reserve((size_t)nf * k * sizeof(float)); // the merge's distance output (all 0 / FLT_MAX), unused
	FlatDB db {vecs, norms, ntotal};
	launch_flat_mfma_tie(geom, p, (const float *)ws_q.p, d_T, nf, db, k, (float *)ws_pd.p, (int32_t *)ws_pi.p,
	                     (unsigned *)ws_gthr.p, st, &sel, d_selmap);
	// k smallest (0, row id): the L2-ordered merge; ids come out as plain row numbers
	launch_merge_partials(METRIC_L2, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, p.nsplit, nf, k, nullptr, 0,
	                      (float *)ws_qn.p, d_rows_out, st);
}

bool FlatIndex::coarse_topk(int64_t nq, const float *d_x, int64_t np, float *d_D, int64_t *d_I, hipStream_t st, bool need_matrix) {
	flush_adds();
	if (!tune().coarse_select || (metric != METRIC_L2 && metric != METRIC_IP) || nq < 20)
		return false; // (fewer than 20 queries: FAISS's per-pair branch, other arithmetic for L2)
	if (!need_matrix && metric == METRIC_L2 && tune().coarse_bf16 && coarse_bf16_supported(d, ntotal, np) && (geom.dp & 3) == 0) {
		// round 6 (csrc/coarse_bf16.hip): this index's own coarse-filter operands -- the centred bf16 store, the query fragments, ||x||^2
		// and 2E(q) of collect_query_prep_kernel -- one filter workgroup per 32 queries, one exact wavefront per query
		use_device();
		stream_wait(st, stream); // adds were enqueued on our own stream
		if (have_last_search)
			stream_wait(st, last_search_stream);
		last_search_stream = st;
		have_last_search = true;
		ensure_h1_rows(st);
		if (nq > 65536) { // (k-means assigns 10^6 training rows at once: in equal pieces of <= 64 K queries, the scratch stays that size)
			const int64_t pieces = (nq + 65535) / 65536, per = ((nq + pieces - 1) / pieces + 63) / 64 * 64; // (>= 32 K each: never a < 20-query tail)
			for (int64_t q0 = 0; q0 < nq;) {
				int64_t m = std::min<int64_t>(per, nq - q0);
				if (nq - q0 - m < 20)
					m = nq - q0;
				(void)coarse_topk(m, d_x + q0 * d, np, d_D + q0 * np, d_I + q0 * np, st, need_matrix);
				q0 += m;
			}
			return true;
		}
		ws_qn.reserve((size_t)nq * sizeof(float));
		ws_e2.reserve((size_t)((nq + 255) / 256 * 256) * sizeof(float));
		ws_pfq.reserve(collect_qfrag_bytes(geom, nq));
		ws_seg.reserve(256 + (size_t)2 * nq * sizeof(int));
		ws_fail.reserve(64 + (size_t)nq * sizeof(int));
		int *fail_cnt = (int *)ws_fail.p, *fail_q = fail_cnt + 16;
		MVS_HIP(hipMemsetAsync(fail_cnt, 0, sizeof(int), st));
		// (no class slots here: stride 0; queries without a finite bound come back with e2 = NaN and are computed exhaustively)
		launch_collect_query_prep(metric, d_x, nq, d, mu_h1, d_max_norm_bits, ws_pfq.p, (float *)ws_qn.p, (float *)ws_e2.p, fail_cnt, fail_q,
		                          nullptr, 0, (int *)ws_seg.p, (int *)((char *)ws_seg.p + 256), st);
		const size_t cb = (coarse_bf16_cand_bytes(nq) + 255) & ~(size_t)255, nb = ((size_t)nq * sizeof(int) + 255) & ~(size_t)255;
		const size_t lb = (coarse_bf16_cls_bytes(nq, ntotal, np) + 255) & ~(size_t)255;
		// (the exhaustive-query counter sits at the FRONT: the buffer's layout behind it changes with nq)
		const bool fresh = ws_cb16.cap < 256 + cb + nb + lb;
		ws_cb16.reserve(256 + cb + nb + lb);
		if (fresh)
			MVS_HIP(hipMemsetAsync(ws_cb16.p, 0, 256, st));
		cb16_stats_off = 0;
		char *const cbase = (char *)ws_cb16.p + 256;
		begin_kernel_timing(st);
		launch_coarse_bf16(d_x, nq, d, ws_pfq.p, (const float *)ws_qn.p, (const float *)ws_e2.p, vecs_h1, beta_h1, vecs, geom.dp,
		                   geom.pair_interleaved ? 1 : 0, norms, ntotal, np, (unsigned short *)cbase, (int *)(cbase + cb), (float *)(cbase + cb + nb), d_D, d_I,
		                   label_offset, (unsigned long long *)ws_cb16.p, st);
		end_kernel_timing(st);
		cb16_queries += nq;
		cb16_last_nq = nq, cb16_ccount_off = 256 + cb;
		return true;
	}
	// inner product: one entry more than asked for, the merge flags boundary ties and resolve_ip_ties replays FAISS's heap
	const bool ip = metric == METRIC_IP;
	if (ip && !(ip_exact_ties && np + 1 <= flat_mfma_max_k(geom)))
		return false;
	const int64_t kk = ip ? np + 1 : np;
	if (!coarse_select_supported(ntotal, kk))
		return false;
	use_device();
	stream_wait(st, stream); // adds were enqueued on our own stream
	if (have_last_search) // the scratch buffers are shared by all searches of this index (FlatIndex::search_flat)
		stream_wait(st, last_search_stream);
	last_search_stream = st;
	have_last_search = true;
	const int64_t qchunk = std::max<int64_t>(64, std::min<int64_t>(nq, ((int64_t)512 << 20) / (ntotal * 4) / 64 * 64));
	ws_q.reserve((size_t)qchunk * ntotal * sizeof(float));
	ws_qn.reserve((size_t)nq * sizeof(float));
	ws_pd.reserve((size_t)nq * kk * sizeof(float));
	ws_pi.reserve((size_t)nq * kk * sizeof(int32_t));
	TieFlags fl = {nullptr, nullptr, nullptr, nullptr};
	if (ip) {
		const size_t kflag = (size_t)(kk + std::max<int64_t>(16, kk / 2));
		ws_flag.reserve(16 + (size_t)nq * 4 + (size_t)nq * kflag * 8);
		fl.count = (int *)ws_flag.p;
		fl.query = fl.count + 4;
		fl.val = (float *)(fl.query + nq);
		fl.row = (int *)(fl.val + (size_t)nq * kflag);
		MVS_HIP(hipMemsetAsync(fl.count, 0, sizeof(int), st));
	} else {
		launch_query_norms(d_x, nq, d, (float *)ws_qn.p, st);
	}
	begin_kernel_timing(st);
	for (int64_t q0 = 0; q0 < nq; q0 += qchunk) {
		const int64_t m = std::min(qchunk, nq - q0);
		// (L2: the selection kernel writes the ordered lists itself; inner product keeps the merge -- it flags the boundary ties)
		launch_coarse_select(d_x + q0 * d, m, d, vecs, geom.dp, geom.pair_interleaved ? 1 : 0, ntotal, (const float *)ws_qn.p + q0, norms,
		                     kk, ip ? 0 : 1, (float *)ws_q.p, (float *)ws_pd.p + q0 * kk, (int32_t *)ws_pi.p + q0 * kk, st,
		                     ip ? nullptr : d_D + q0 * np, ip ? nullptr : d_I + q0 * np, label_offset);
	}
	end_kernel_timing(st);
	if (ip)
		launch_merge_partials(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, 1, nq, kk, nullptr, label_offset, d_D, d_I, st, np, &fl);
	if (ip) {
		SelectorDev nosel;
		memset(&nosel, 0, sizeof nosel);
		resolve_ip_ties(nq, d_x, np, fl, nosel, nullptr, d_D, d_I, st);
	}
	return true;
}

void FlatIndex::search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
                              const mvs_search_params *params, hipStream_t st) {
	use_device();
	flush_adds();
	stream_wait(st, stream); // adds were enqueued on our own stream
	search_flat(nq, d_x, k, d_D, d_I, params, nullptr, st);
}

void FlatIndex::to_device(int new_device) {
	if (new_device == device)
		return;
	int ndev = 0;
	MVS_HIP(hipGetDeviceCount(&ndev));
	if (new_device < 0 || new_device >= ndev)
		throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "Invalid GPU device %d", new_device);
	use_device();
	flush_adds();
	MVS_HIP(hipStreamSynchronize(stream));
	reap_retired(true);
	add_ring.drop_events();
	pinned.drop_events();
	float *nv = nullptr, *nn = nullptr;
	MVS_HIP(hipSetDevice(new_device));
	if (cap > 0) {
		MVS_HIP(hipMalloc((void **)&nv, ((size_t)cap * geom.dp + 64) * sizeof(float)));
		MVS_HIP(hipMalloc((void **)&nn, ((size_t)cap + 64) * sizeof(float)));
		MVS_HIP(hipMemset(nv, 0, ((size_t)cap * geom.dp + 64) * sizeof(float)));
		if (ntotal > 0) {
			MVS_HIP(hipMemcpyPeer(nv, new_device, vecs, device, (size_t)ntotal * geom.dp * sizeof(float)));
			MVS_HIP(hipMemcpyPeer(nn, new_device, norms, device, (size_t)ntotal * sizeof(float)));
		}
	}
	hipStream_t ns;
	MVS_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
	MVS_HIP(hipSetDevice(device));
	if (vecs)
		MVS_HIP(hipFree(vecs));
	if (norms)
		MVS_HIP(hipFree(norms));
	drop_bf16_rows();
	for (DevBuf *b : {&ws_flag, &ws_tie, &ws_pfq, &ws_cand, &ws_ex, &ws_fail, &ws_fb, &ws_e2, &ws_stream, &ws_sorttmp, &ws_seg, &ws_pbnd,
	                  &ws_rowmask, &ws_items1, &ws_qcount, &ws_fbk, &ws_fbr, &ws_seed})
		b->release();
	drop_shadow(); // (the shadow clustering lives on the old device: rebuilt on demand)
	ws_q.release();
	ws_qn.release();
	ws_pd.release();
	ws_pi.release();
	ws_xi.release();
	ws_gthr.release();
	ws_add.release();
	ws_hx.release();
	ws_hD.release();
	ws_hI.release();
	selector.buf.release();
	MVS_HIP(hipStreamDestroy(stream));
	vecs = nv;
	norms = nn;
	stream = ns;
	device = new_device;
	MVS_HIP(hipSetDevice(device));
}

static void check_device(int dev) {
	int ndev = 0;
	MVS_HIP(hipGetDeviceCount(&ndev));
	if (dev < 0 || dev >= ndev)
		throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "Invalid GPU device %d", dev);
}

IndexBase *FlatIndex::clone(int on_device) {
	check_device(on_device);
	use_device();
	flush_adds();
	MVS_HIP(hipStreamSynchronize(stream));
	auto *c = new FlatIndex(d, metric);
	try {
		c->to_device(on_device);
		c->use_device();
		c->grow(ntotal, c->stream);
		if (ntotal > 0) {
			MVS_HIP(hipMemcpyPeer(c->vecs, on_device, vecs, device, (size_t)ntotal * geom.dp * sizeof(float)));
			MVS_HIP(hipMemcpyPeer(c->norms, on_device, norms, device, (size_t)ntotal * sizeof(float)));
		}
		c->ntotal = ntotal;
		c->label_offset = label_offset;
	} catch (...) {
		delete c;
		throw;
	}
	return c;
}

// ------------------------------------------------------------------------------------------ IDMap

IDMapIndex::IDMapIndex(IndexBase *sub_) : IndexBase(MVS_KIND_IDMAP, sub_->d, sub_->metric), sub(sub_) {
	is_trained = sub->is_trained;
}
IDMapIndex::~IDMapIndex() {
	(void)hipSetDevice(device);
	if (ids)
		(void)hipFree(ids);
	delete sub;
}
void IDMapIndex::train(int64_t n, const float *x) {
	sub->train(n, x);
	is_trained = sub->is_trained;
}
void IDMapIndex::add(int64_t, const float *) {
	throw_faiss("virtual void faiss::IndexIDMapTemplate<IndexT>::add(faiss::idx_t, const float*)",
	            "faiss/IndexIDMap.cpp", "add does not make sense with IndexIDMap, use add_with_ids");
}
void IDMapIndex::add_device(int64_t, const float *, hipStream_t) {
	throw_faiss("virtual void faiss::IndexIDMapTemplate<IndexT>::add(faiss::idx_t, const float*)",
	            "faiss/IndexIDMap.cpp", "add does not make sense with IndexIDMap, use add_with_ids");
}
void IDMapIndex::grow_ids(int64_t need, hipStream_t st) {
	if (need <= idcap)
		return;
	// (as FlatIndex::grow; by eight while the array is below 64 MB: every growth ends in a hipFree that waits for the device)
	const int64_t nc = std::max<int64_t>(need, (idcap < ((int64_t)8 << 20) ? 8 : 2) * idcap + 4096);
	int64_t *ni = nullptr;
	MVS_HIP(hipMalloc((void **)&ni, (size_t)nc * sizeof(int64_t)));
	flush_ids(); // (staged ids go to the old array first: the copy below carries them over)
	if (ntotal > 0)
		MVS_HIP(hipMemcpyAsync(ni, ids, (size_t)ntotal * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
	MVS_HIP(hipStreamSynchronize(st));
	if (ids)
		MVS_HIP(hipFree(ids));
	ids = ni;
	idcap = nc;
}
// IndexIDMap::add_with_ids: index->add(n, x); id_map.push_back(ids...)  (src/faiss_extension.cpp:510,607)
void IDMapIndex::add_with_ids(int64_t n, const float *x, const int64_t *xids) {
	use_device();
	if (n <= 0)
		return;
	sub->add(n, x);
	use_device();
	grow_ids(ntotal + n, stream);
	const size_t bytes = (size_t)n * sizeof(int64_t);
	// round 6: the ids of DataChunk-sized adds are staged like their rows (FlatIndex::add) -- one H2D copy per full slot, not per call
	if (bytes > PinnedRing::SLOT_BYTES / 2 || (idp_slot >= 0 && idp_bytes + bytes > PinnedRing::SLOT_BYTES))
		flush_ids();
	if (bytes > PinnedRing::SLOT_BYTES / 2) {
		const int slot = id_ring.acquire(bytes);
		memcpy(id_ring.buf[slot], xids, bytes);
		MVS_HIP(hipMemcpyAsync(ids + ntotal, id_ring.buf[slot], bytes, hipMemcpyHostToDevice, stream));
		id_ring.release(slot, stream);
	} else {
		if (idp_slot < 0) {
			idp_slot = id_ring.acquire(PinnedRing::SLOT_BYTES);
			idp_bytes = 0, idp_row0 = ntotal;
		}
		memcpy((char *)id_ring.buf[idp_slot] + idp_bytes, xids, bytes);
		idp_bytes += bytes;
	}
	ntotal = sub->ntotal;
}
void IDMapIndex::flush_ids() {
	if (idp_slot < 0)
		return;
	if (idp_bytes > 0)
		MVS_HIP(hipMemcpyAsync(ids + idp_row0, id_ring.buf[idp_slot], idp_bytes, hipMemcpyHostToDevice, stream));
	id_ring.release(idp_slot, stream);
	idp_slot = -1;
	idp_bytes = 0;
}
void IDMapIndex::add_with_ids_device(int64_t n, const float *d_x, const int64_t *d_ids, hipStream_t st) {
	use_device();
	if (n <= 0)
		return;
	flush_ids();
	stream_wait(st, stream);
	sub->add_device(n, d_x, st);
	grow_ids(ntotal + n, st);
	MVS_HIP(hipMemcpyAsync(ids + ntotal, d_ids, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
	stream_wait(stream, st);
	ntotal = sub->ntotal;
}
// IndexIDMap::search: selector tests EXTERNAL ids (IDSelectorTranslated), labels = id_map[internal]
void IDMapIndex::search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
                               const mvs_search_params *params, hipStream_t st) {
	use_device();
	flush_ids();
	stream_wait(st, stream); // id_map writes happened on our own stream; the sub-index searches on `st`
	sub->search_mapped(nq, d_x, k, d_D, d_I, params, ids, st);
	kinfo = sub->kinfo;
}
void IDMapIndex::search(int64_t nq, const float *x, int64_t k, float *D, int64_t *I, const mvs_search_params *params) {
	IndexBase::search(nq, x, k, D, I, params);
}
void IDMapIndex::to_device(int new_device) {
	if (new_device == device)
		return;
	sub->to_device(new_device);
	use_device();
	flush_ids();
	MVS_HIP(hipStreamSynchronize(stream));
	id_ring.drop_events();
	pinned.drop_events();
	MVS_HIP(hipStreamSynchronize(stream));
	int64_t *ni = nullptr;
	if (idcap > 0) {
		MVS_HIP(hipSetDevice(new_device));
		MVS_HIP(hipMalloc((void **)&ni, (size_t)idcap * sizeof(int64_t)));
		if (ntotal > 0)
			MVS_HIP(hipMemcpyPeer(ni, new_device, ids, device, (size_t)ntotal * sizeof(int64_t)));
		MVS_HIP(hipSetDevice(device));
		MVS_HIP(hipFree(ids));
	}
	ws_hx.release();
	ws_hD.release();
	ws_hI.release();
	MVS_HIP(hipStreamDestroy(stream));
	MVS_HIP(hipSetDevice(new_device));
	MVS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
	ids = ni;
	device = new_device;
}

IndexBase *IDMapIndex::clone(int on_device) {
	check_device(on_device);
	use_device();
	flush_ids();
	MVS_HIP(hipStreamSynchronize(stream));
	IndexBase *subc = sub->clone(on_device);
	IDMapIndex *c = nullptr;
	try {
		c = new IDMapIndex(subc);
	} catch (...) {
		delete subc;
		throw;
	}
	try {
		c->to_device(on_device); // sub is already there: only moves this wrapper's stream / ids
		c->use_device();
		c->grow_ids(ntotal, c->stream);
		if (ntotal > 0)
			MVS_HIP(hipMemcpyPeer(c->ids, on_device, ids, device, (size_t)ntotal * sizeof(int64_t)));
		c->ntotal = ntotal;
	} catch (...) {
		delete c;
		throw;
	}
	return c;
}

void FlatIndex::search_mapped(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
                              const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st) {
	use_device();
	flush_adds();
	stream_wait(st, stream); // adds were enqueued on our own stream
	search_flat(nq, d_x, k, d_D, d_I, params, d_idmap, st);
}

// ------------------------------------------------------------------------------------------ host images

void FlatIndex::to_host(HostIndex &out) {
	out.kind = MVS_KIND_FLAT;
	out.d = d;
	out.metric = metric;
	out.metric_arg = metric_arg;
	out.ntotal = ntotal;
	out.is_trained = true;
	out.rows.resize((size_t)ntotal * d);
	copy_rows_to_host(out.rows.data());
}
void IDMapIndex::to_host(HostIndex &out) {
	use_device();
	flush_ids();
	MVS_HIP(hipStreamSynchronize(stream));
	out.kind = MVS_KIND_IDMAP;
	out.d = d;
	out.metric = metric;
	out.ntotal = ntotal;
	out.is_trained = is_trained;
	out.sub.reset(new HostIndex);
	sub->to_host(*out.sub);
	out.ids.resize((size_t)ntotal);
	if (ntotal > 0)
		MVS_HIP(hipMemcpy(out.ids.data(), ids, (size_t)ntotal * sizeof(int64_t), hipMemcpyDeviceToHost));
}

IndexBase *index_from_host(const HostIndex &h, int device) {
	CtorDevice scope(device);
	switch (h.kind) {
	case MVS_KIND_FLAT: {
		auto *f = new FlatIndex(h.d, h.metric);
		f->metric_arg = h.metric_arg;
		try {
			if (h.ntotal > 0)
				f->add(h.ntotal, h.rows.data());
		} catch (...) {
			delete f;
			throw;
		}
		return f;
	}
	case MVS_KIND_IDMAP: {
		if (!h.sub)
			throw_faiss("mvs::index_from_host", __FILE__, "IDMap image without a sub-index");
		if (h.sub->ntotal != (int64_t)h.ids.size())
			throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
			            "IDMap id_map size %zu does not match the sub-index ntotal %lld", h.ids.size(),
			            (long long)h.sub->ntotal);
		IndexBase *subi = index_from_host(*h.sub, device);
		IDMapIndex *m = nullptr;
		try {
			m = new IDMapIndex(subi);
		} catch (...) {
			delete subi;
			throw;
		}
		try {
			m->adopt_ids(h.ids.data(), (int64_t)h.ids.size());
		} catch (...) {
			delete m;
			throw;
		}
		return m;
	}
	case MVS_KIND_IVFFLAT:
		return ivf_from_host(h, device);
	case MVS_KIND_HNSW:
		return hnsw_from_host(h, device);
	}
	throw_faiss("mvs::index_from_host", __FILE__, "unknown index kind %d", h.kind);
}

// IndexIDMap image load: the sub-index already holds the rows, only the id_map is missing
void IDMapIndex::adopt_ids(const int64_t *xids, int64_t n) {
	use_device();
	grow_ids(n, stream);
	if (n > 0)
		MVS_HIP(hipMemcpyAsync(ids, xids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, stream));
	MVS_HIP(hipStreamSynchronize(stream));
	ntotal = sub->ntotal;
	is_trained = sub->is_trained;
}

// ------------------------------------------------------------------------------------------ factory

// faiss::index_factory subset (faiss/index_factory.cpp) -- the strings the reference and its tests use:
// "Flat" (faiss.test:8), "IDMap,Flat" (faiss2.test:8), "IDMap,IVF1,Flat", "IVF<n>,Flat", "HNSW<M>"
static IndexBase *factory_rec(int d, const std::string &desc, int metric, const std::string &full) {
	if (desc.rfind("IDMap2,", 0) == 0 || desc.rfind("IDMap,", 0) == 0) {
		IndexBase *sub = factory_rec(d, desc.substr(desc.find(',') + 1), metric, full);
		try {
			return new IDMapIndex(sub);
		} catch (...) {
			delete sub;
			throw;
		}
	}
	if (desc == "Flat")
		return new FlatIndex(d, metric);
	if (IndexBase *ix = make_ivf_index(d, desc, metric))
		return ix;
	if (IndexBase *ix = make_hnsw_index(d, desc, metric))
		return ix;
	throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
	            "could not parse index string %s", full.c_str());
}

IndexBase *index_factory(int d, const char *description, int metric) {
	if (d <= 0)
		throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)",
		            "faiss/index_factory.cpp", "invalid dimension %d", d);
	return factory_rec(d, description, metric, description);
}

} // namespace mvs

// =================================================================================================
// C ABI
// =================================================================================================
using namespace mvs;

struct mvs_index {
	IndexBase *impl;
	bool owned;
	mvs_index *sub_handle = nullptr;
	mvs_index *quantizer_handle = nullptr;
	std::mutex mu; // the reference's faiss_lock serialises calls per index; be safe for other hosts
};

#define MVS_API_BEGIN                                                                                                  \
	try {
#define MVS_API_END                                                                                                    \
	}                                                                                                                  \
	catch (const std::exception &e) {                                                                                  \
		g_last_error = e.what();                                                                                       \
		return 1;                                                                                                      \
	}                                                                                                                  \
	catch (...) {                                                                                                      \
		g_last_error = "unknown error";                                                                                \
		return 1;                                                                                                      \
	}                                                                                                                  \
	return 0;

extern "C" {

const char *mvs_last_error(void) {
	return g_last_error.c_str();
}
const char *mvs_version(void) {
	return "mi355-faiss 0.1 (drop-in for duckdb-faiss-ext 0.12.1 hot path)";
}
int mvs_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int mvs_index_factory(mvs_index **out, int d, const char *description, int metric) {
	MVS_API_BEGIN
	*out = nullptr;
	// env MVS_DEVICES=0,1,...,7: the index is created row-sharded / replicated over those devices (csrc/sharded.hip)
	const std::vector<int> devs = shard_devices_from_env();
	IndexBase *impl = devs.size() > 1 ? make_sharded_index(d, description, metric, devs) : index_factory(d, description, metric);
	auto *h = new mvs_index;
	h->impl = impl;
	h->owned = true;
	*out = h;
	MVS_API_END
}
void mvs_index_free(mvs_index *ix) {
	if (!ix)
		return;
	delete ix->sub_handle;
	delete ix->quantizer_handle;
	if (ix->owned) {
		try {
			delete ix->impl;
		} catch (...) {
		}
	}
	delete ix;
}
int mvs_index_d(const mvs_index *ix) {
	return ix->impl->d;
}
int64_t mvs_index_ntotal(const mvs_index *ix) {
	return ix->impl->ntotal;
}
int mvs_index_is_trained(const mvs_index *ix) {
	return ix->impl->is_trained ? 1 : 0;
}
int mvs_index_metric_type(const mvs_index *ix) {
	return ix->impl->metric;
}
int mvs_index_kind(const mvs_index *ix) {
	return ix->impl->kind;
}
int mvs_index_device(const mvs_index *ix) {
	return ix->impl->device;
}
mvs_index *mvs_index_idmap_sub(mvs_index *ix) {
	if (ix->impl->kind != MVS_KIND_IDMAP)
		return nullptr;
	if (!ix->sub_handle) {
		ix->sub_handle = new mvs_index;
		ix->sub_handle->impl =
		    is_sharded(ix->impl) ? sharded_inner_view(ix->impl) : static_cast<IDMapIndex *>(ix->impl)->sub;
		ix->sub_handle->owned = false;
	}
	return ix->sub_handle;
}
mvs_index *mvs_index_ivf_quantizer(mvs_index *ix) {
	IndexBase *q = ivf_quantizer_of(sharded_inner_view(ix->impl));
	if (!q)
		return nullptr;
	if (!ix->quantizer_handle) {
		ix->quantizer_handle = new mvs_index;
		ix->quantizer_handle->impl = q;
		ix->quantizer_handle->owned = false;
	}
	return ix->quantizer_handle;
}
static IndexBase *unwrap_idmap(IndexBase *p) {
	p = sharded_inner_view(p); // a sharded index answers for its first shard / replica
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	return p;
}
int64_t mvs_index_ivf_nlist(const mvs_index *ix) {
	return ivf_nlist_of(unwrap_idmap(ix->impl));
}
int mvs_index_ivf_get_centroids(mvs_index *ix, float *out) {
	MVS_API_BEGIN
	if (!ivf_get_centroids(unwrap_idmap(ix->impl), out))
		throw_faiss("mvs_index_ivf_get_centroids", __FILE__, "not an IVF index");
	MVS_API_END
}
int mvs_index_ivf_set_centroids(mvs_index *ix, const float *centroids) {
	MVS_API_BEGIN
	bool ok = true;
	sharded_for_each(ix->impl, [&](IndexBase *top) { // every row shard probes the same lists
		IndexBase *p = top;
		while (p->kind == MVS_KIND_IDMAP)
			p = static_cast<IDMapIndex *>(p)->sub;
		ok = ok && ivf_set_centroids(p, centroids);
		for (IndexBase *w = top; w->kind == MVS_KIND_IDMAP; w = static_cast<IDMapIndex *>(w)->sub)
			w->is_trained = true;
	});
	if (!ok)
		throw_faiss("mvs_index_ivf_set_centroids", __FILE__, "not an IVF index");
	ix->impl->is_trained = true;
	MVS_API_END
}
int mvs_index_hnsw_set_ef_construction(mvs_index *ix, int v) {
	MVS_API_BEGIN
	bool ok = true;
	sharded_for_each(ix->impl, [&](IndexBase *top) { // every replica builds with the same parameter
		IndexBase *p = top;
		while (p->kind == MVS_KIND_IDMAP)
			p = static_cast<IDMapIndex *>(p)->sub;
		ok = ok && hnsw_set_ef_construction(p, v);
	});
	if (!ok)
		throw_faiss("mvs_index_hnsw_set_ef_construction", __FILE__, "not an HNSW index");
	MVS_API_END
}
int mvs_index_hnsw_get_ef_construction(mvs_index *ix) {
	return hnsw_get_ef_construction(unwrap_idmap(ix->impl));
}
int64_t mvs_index_hnsw_graph_info(mvs_index *ix, int *max_level, int *entry_point) {
	try {
		return hnsw_graph_info(unwrap_idmap(ix->impl), max_level, entry_point);
	} catch (...) {
		return -1;
	}
}
int mvs_index_hnsw_walk_stats(mvs_index *ix, double *evaluations, double *f32_rows, double *bf16_rows) {
	MVS_API_BEGIN
	if (!hnsw_walk_stats(unwrap_idmap(ix->impl), evaluations, f32_rows, bf16_rows))
		throw_faiss("mvs_index_hnsw_walk_stats", __FILE__, "not an HNSW index");
	MVS_API_END
}
int mvs_index_hnsw_get_graph(mvs_index *ix, int32_t *levels, int64_t *offsets, int32_t *neighbors) {
	MVS_API_BEGIN
	if (!hnsw_get_graph(unwrap_idmap(ix->impl), levels, offsets, neighbors))
		throw_faiss("mvs_index_hnsw_get_graph", __FILE__, "not an HNSW index");
	MVS_API_END
}

int mvs_index_train(mvs_index *ix, int64_t n, const float *x) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	ix->impl->train(n, x);
	MVS_API_END
}
int mvs_index_add(mvs_index *ix, int64_t n, const float *x) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	ix->impl->add(n, x);
	MVS_API_END
}
int mvs_index_add_with_ids(mvs_index *ix, int64_t n, const float *x, const int64_t *ids) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	ix->impl->add_with_ids(n, x, ids);
	MVS_API_END
}
int mvs_index_search(mvs_index *ix, int64_t n, const float *x, int64_t k, float *distances, int64_t *labels,
                     const mvs_search_params *params) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	ix->impl->search(n, x, k, distances, labels, params);
	MVS_API_END
}
int mvs_index_to_gpu(mvs_index *ix, int device) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	ix->impl->to_device(device);
	MVS_API_END
}
int mvs_index_clone_to_gpu(mvs_index **out, const mvs_index *src, int device) {
	MVS_API_BEGIN
	*out = nullptr;
	std::lock_guard<std::mutex> g(const_cast<mvs_index *>(src)->mu);
	IndexBase *impl = nullptr;
	if (device < 0) { // faiss_to_gpu(name, -1): every device of MVS_DEVICES, or all visible ones
		std::vector<int> devs = shard_devices_from_env();
		if (devs.empty())
			for (int i = 0; i < mvs_device_count(); ++i)
				devs.push_back(i);
		HostIndex img;
		src->impl->to_host(img);
		impl = shard_from_host(img, devs);
	} else {
		impl = src->impl->clone(device);
	}
	impl->adopt_tuning(src->impl->tune_);
	auto *h = new mvs_index;
	h->impl = impl;
	h->owned = true;
	*out = h;
	MVS_API_END
}
int mvs_index_shard_to_gpus(mvs_index *ix, const int *devices, int ndev) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	if (ndev <= 0 || !devices)
		throw_faiss("mvs_index_shard_to_gpus", __FILE__, "Invalid GPU device list");
	if (ix->sub_handle || ix->quantizer_handle) {
		// borrowed views of the old object graph (IndexIDMap::index, IndexIVF::quantizer) die with it
		delete ix->sub_handle;
		delete ix->quantizer_handle;
		ix->sub_handle = ix->quantizer_handle = nullptr;
	}
	HostIndex img;
	ix->impl->to_host(img);
	IndexBase *sharded = shard_from_host(img, std::vector<int>(devices, devices + ndev));
	sharded->adopt_tuning(ix->impl->tune_);
	if (ix->owned)
		delete ix->impl;
	ix->impl = sharded;
	ix->owned = true;
	MVS_API_END
}
int mvs_index_prefilter_stats(mvs_index *ix, int64_t *queries, int64_t *fallback_queries, float *max_rel_err,
                              float *err_bound) {
	MVS_API_BEGIN
	IndexBase *p = sharded_inner_view(ix->impl);
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	if (p->kind != MVS_KIND_FLAT)
		throw_faiss("mvs_index_prefilter_stats", __FILE__, "not a Flat index");
	auto *f = static_cast<FlatIndex *>(p);
	if (queries)
		*queries = f->pf_queries_total;
	if (fallback_queries)
		*fallback_queries = f->pf_fallback_total;
	if (max_rel_err)
		*max_rel_err = f->pf_max_rel_err;
	if (err_bound)
		*err_bound = prefilter_cerr(f->d);
	MVS_API_END
}
int mvs_index_collect_stats(mvs_index *ix, int64_t *queries, int64_t *candidates, int64_t *overflows) {
	MVS_API_BEGIN
	IndexBase *p = sharded_inner_view(ix->impl);
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	if (p->kind != MVS_KIND_FLAT) {
		if (p->collect_stats(queries, candidates, overflows)) // IVF: its own coarse filter (csrc/ivf_collect.hip)
			return 0;
		throw_faiss("mvs_index_collect_stats", __FILE__, "not a Flat or IVF index");
	}
	auto *f = static_cast<FlatIndex *>(p);
	if (queries)
		*queries = f->cl_queries_total;
	if (candidates) // (the bucketed finish re-scores the survivors of the final-bound filter only: admitted - (admitted - re-scored) of those searches)
		*candidates = f->cl_candidates_total - (f->cl_admitted_in_fb - f->cl_rescored_total);
	if (overflows)
		*overflows = f->cl_overflows;
	MVS_API_END
}
int mvs_trace_push(const char *name) {
	trace_push(name ? name : "mvs");
	return 0;
}
int mvs_trace_pop(void) {
	trace_pop();
	return 0;
}
int mvs_index_get_stat(mvs_index *ix, const char *name, int64_t *value) {
	MVS_API_BEGIN
	IndexBase *p = sharded_inner_view(ix->impl);
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	if (!name || !value)
		throw_faiss("mvs_index_get_stat", __FILE__, "null argument");
	if (!strcmp(name, "flat_outlier_rows")) { // rows of a Flat index kept out of its coarse-filter store (found so far; csrc/flat_collect.hip)
		if (p->kind != MVS_KIND_FLAT)
			throw_faiss("mvs_index_get_stat", __FILE__, "%s: not a Flat index", name);
		*value = static_cast<FlatIndex *>(p)->outl_total;
	} else if (!strcmp(name, "coarse_bf16_queries") || !strcmp(name, "coarse_bf16_exhaustive") || !strcmp(name, "coarse_bf16_candidates")) {
		// IVF: queries whose coarse quantisation ran on csrc/coarse_bf16.hip / of those, computed against every centroid
		IndexBase *qz = ivf_quantizer_of(p);
		if (!qz || qz->kind != MVS_KIND_FLAT)
			throw_faiss("mvs_index_get_stat", __FILE__, "%s: not an IVF index with a Flat quantizer", name);
		auto *f = static_cast<FlatIndex *>(qz);
		if (!strcmp(name, "coarse_bf16_queries")) {
			*value = f->cb16_queries;
		} else if (!strcmp(name, "coarse_bf16_candidates")) { // candidates re-scored exactly in the LAST call (its last piece), summed
			int64_t sum = 0;
			if (f->ws_cb16.p && f->cb16_last_nq > 0) {
				f->use_device();
				MVS_HIP(hipDeviceSynchronize());
				std::vector<int> cc((size_t)f->cb16_last_nq);
				MVS_HIP(hipMemcpy(cc.data(), (const char *)f->ws_cb16.p + f->cb16_ccount_off, cc.size() * sizeof(int), hipMemcpyDeviceToHost));
				for (int v : cc)
					sum += v > 0 ? v : 0;
			}
			*value = sum;
		} else {
			unsigned long long v = 0;
			if (f->ws_cb16.p) {
				f->use_device();
				MVS_HIP(hipDeviceSynchronize());
				MVS_HIP(hipMemcpy(&v, (const char *)f->ws_cb16.p + f->cb16_stats_off, sizeof v, hipMemcpyDeviceToHost));
			}
			*value = (int64_t)v;
		}
	} else if (!p->named_stat(name, value)) {
		throw_faiss("mvs_index_get_stat", __FILE__, "unknown statistic %s", name);
	}
	MVS_API_END
}
int mvs_index_shadow_stats(mvs_index *ix, int64_t *stats /* [8] */, double *build_seconds) {
	MVS_API_BEGIN
	IndexBase *p = sharded_inner_view(ix->impl);
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	if (p->kind != MVS_KIND_FLAT)
		throw_faiss("mvs_index_shadow_stats", __FILE__, "not a Flat index");
	auto *f = static_cast<FlatIndex *>(p);
	if (stats) {
		stats[0] = f->shadow_state, stats[1] = f->shadow ? f->shadow_rows : -1, stats[2] = f->shadow_queries, stats[3] = f->shadow_unproven;
		stats[4] = f->shadow_builds, stats[5] = f->shadow_extends, stats[6] = f->shadow ? (int64_t)f->shadow->device_bytes() : 0;
		stats[7] = f->shadow ? ivf_nlist_of(f->shadow) : 0;
	}
	if (build_seconds)
		*build_seconds = f->shadow_build_seconds;
	MVS_API_END
}
int mvs_index_ivf_probe_stats(mvs_index *ix, int64_t *pairs, int64_t *pairs_scanned, int64_t *forced_drains, int64_t *admitted) {
	MVS_API_BEGIN
	IndexBase *p = sharded_inner_view(ix->impl);
	while (p->kind == MVS_KIND_IDMAP)
		p = static_cast<IDMapIndex *>(p)->sub;
	if (p->kind == MVS_KIND_FLAT && static_cast<FlatIndex *>(p)->shadow) // a Flat index answering through its shadow clustering
		p = static_cast<FlatIndex *>(p)->shadow;
	if (!p->probe_stats(pairs, pairs_scanned, forced_drains, admitted)) {
		if (p->kind != MVS_KIND_FLAT)
			throw_faiss("mvs_index_ivf_probe_stats", __FILE__, "not an IVF index");
		// a Flat index: no probes; `admitted` = candidates its coarse filter admitted in the last search (mvs_index_collect_stats counts
		// the re-scored ones)
		if (pairs)
			*pairs = 0;
		if (pairs_scanned)
			*pairs_scanned = 0;
		if (forced_drains)
			*forced_drains = 0;
		if (admitted)
			*admitted = static_cast<FlatIndex *>(p)->cl_last_candidates;
	}
	MVS_API_END
}
int mvs_index_shard_info(const mvs_index *ix, int *devices, int max_devices, int64_t *rows_per_shard,
                         int64_t *last_tie_queries) {
	return sharded_info(ix->impl, devices, max_devices, rows_per_shard, last_tie_queries);
}
int mvs_index_add_device(mvs_index *ix, int64_t n, const float *d_x, const int64_t *d_ids, void *stream) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	if (d_ids)
		ix->impl->add_with_ids_device(n, d_x, d_ids, (hipStream_t)stream);
	else
		ix->impl->add_device(n, d_x, (hipStream_t)stream);
	MVS_API_END
}
int mvs_index_search_device(mvs_index *ix, int64_t n, const float *d_x, int64_t k, float *d_distances,
                            int64_t *d_labels, const mvs_search_params *params, void *stream) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	if (k <= 0)
		throw_faiss("virtual void faiss::Index::search(...) const", "faiss/Index.cpp", "Error: 'k > 0' failed");
	ix->impl->search_device(n, d_x, k, d_distances, d_labels, params, (hipStream_t)stream);
	MVS_API_END
}
int mvs_index_set_label_offset(mvs_index *ix, int64_t offset) {
	MVS_API_BEGIN
	ix->impl->set_label_offset(offset);
	MVS_API_END
}

int mvs_write_index(const mvs_index *ix, const char *filename) {
	MVS_API_BEGIN
	write_index_file(ix->impl, filename);
	MVS_API_END
}
int mvs_read_index(mvs_index **out, const char *filename) {
	MVS_API_BEGIN
	*out = nullptr;
	IndexBase *impl = read_index_file(filename);
	auto *h = new mvs_index;
	h->impl = impl;
	h->owned = true;
	*out = h;
	MVS_API_END
}

int mvs_merge_shards(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out,
                     int64_t *I_out) {
	MVS_API_BEGIN
	merge_shards_host(metric, n, k, nshard, D, I, D_out, I_out);
	MVS_API_END
}

int mvs_merge_records_device(int metric, int64_t n, int kk, int kout, int nshard, const int64_t *d_records, int raw,
                             float *d_D_out, int64_t *d_I_out, void *stream) {
	MVS_API_BEGIN
	launch_merge_records(metric, d_records, nshard, n, kk, kout, raw != 0, d_D_out, d_I_out, (hipStream_t)stream);
	MVS_API_END
}

int mvs_merge_shards_raw(int metric, int64_t n, int64_t kk, int nshard, const float *D, const int64_t *I, float *D_out,
                         int64_t *I_out) {
	MVS_API_BEGIN
	merge_shards_raw_host(metric, n, kk, nshard, D, I, D_out, I_out);
	MVS_API_END
}
int mvs_finish_ip_ties(int64_t n, int64_t k, int64_t kk, const float *raw_D, const int64_t *raw_I, int64_t nf,
                       const int64_t *flagged, const int64_t *first_rows, float *D_out, int64_t *I_out) {
	MVS_API_BEGIN
	finish_ip_ties_host(n, k, kk, raw_D, raw_I, nf, flagged, first_rows, D_out, I_out);
	MVS_API_END
}
int mvs_index_tie_candidates_device(mvs_index *ix, int64_t nf, const float *d_xf, const float *d_T, int64_t k,
                                    int64_t *d_rows_out, const mvs_search_params *params, void *stream) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	if (ix->impl->kind != MVS_KIND_FLAT || is_sharded(ix->impl))
		throw_faiss("mvs_index_tie_candidates_device", __FILE__, "a plain single-device Flat index is required");
	auto *f = static_cast<FlatIndex *>(ix->impl);
	hipStream_t st = (hipStream_t)stream;
	f->use_device();
	stream_wait(st, f->stream);
	SelectorDev sel = f->upload_selector(params, st);
	f->tie_candidates(nf, d_xf, d_T, k, d_rows_out, sel, nullptr, st);
	f->offset_rows(d_rows_out, nf * k, st); // shard rows -> global rows (label offset)
	MVS_API_END
}
int mvs_index_ivf_tie_emit_device(mvs_index *ix, int64_t nf, const int *d_flag, const float *d_x, const float *d_T, int64_t k,
                                  float *d_v_out, int64_t *d_id_out, int *d_rank_out, const mvs_search_params *params,
                                  void *stream) {
	MVS_API_BEGIN
	std::lock_guard<std::mutex> g(ix->mu);
	if (ix->impl->kind != MVS_KIND_IVFFLAT || is_sharded(ix->impl))
		throw_faiss("mvs_index_ivf_tie_emit_device", __FILE__, "a plain single-device IVF index is required");
	// (the coarse assignment the call reuses belongs to ONE batch -- another pointer is another batch as far as anyone can tell)
	if (nf > 0 && ix->impl->last_batch_ptr() != d_x)
		throw_faiss("mvs_index_ivf_tie_emit_device", __FILE__, "the batch at %p is not the one of the search that has just run on this index (%p)",
		            (const void *)d_x, (const void *)ix->impl->last_batch_ptr());
	if (nf > 0)
		ix->impl->tie_emit(d_flag, (int)nf, d_x, d_T, k, params, nullptr, d_v_out, d_id_out, d_rank_out, (hipStream_t)stream);
	MVS_API_END
}
int mvs_synth_uniform_device(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, void *stream) {
	MVS_API_BEGIN
	launch_synth_uniform(d_out, n_rows, d, seed, row0, (hipStream_t)stream);
	MVS_API_END
}
int mvs_synth_clustered_device(float *d_out, int64_t n_rows, int d, uint64_t seed, int64_t row0, int n_centers,
                               float sigma, void *stream) {
	MVS_API_BEGIN
	launch_synth_clustered(d_out, n_rows, d, seed, row0, n_centers, sigma, (hipStream_t)stream);
	MVS_API_END
}

int mvs_debug_mfma_bf16_16x16x32(const uint16_t *A, const uint16_t *Bt, const float *C, float *D, int64_t ntiles) {
	MVS_API_BEGIN
	if (ntiles <= 0)
		return 0;
	DevBuf a, b, c, d;
	a.reserve((size_t)ntiles * 512 * 2), b.reserve((size_t)ntiles * 512 * 2), c.reserve((size_t)ntiles * 256 * 4), d.reserve((size_t)ntiles * 256 * 4);
	MVS_HIP(hipMemcpy(a.p, A, (size_t)ntiles * 512 * 2, hipMemcpyHostToDevice));
	MVS_HIP(hipMemcpy(b.p, Bt, (size_t)ntiles * 512 * 2, hipMemcpyHostToDevice));
	MVS_HIP(hipMemcpy(c.p, C, (size_t)ntiles * 256 * 4, hipMemcpyHostToDevice));
	launch_mfma_bf16_probe((const unsigned short *)a.p, (const unsigned short *)b.p, (const float *)c.p, (float *)d.p, ntiles, nullptr);
	MVS_HIP(hipDeviceSynchronize());
	MVS_HIP(hipMemcpy(D, d.p, (size_t)ntiles * 256 * 4, hipMemcpyDeviceToHost));
	MVS_API_END
}
int mvs_index_last_kernel_info(const mvs_index *ix, mvs_kernel_info *out) {
	MVS_API_BEGIN
	*out = ix->impl->kinfo;
	MVS_API_END
}
int mvs_index_set_kernel_timing(mvs_index *ix, int enabled) {
	MVS_API_BEGIN
	ix->impl->set_timing(enabled != 0);
	MVS_API_END
}
int mvs_index_kernel_time_stats(mvs_index *ix, int *count, double *total_ms) {
	MVS_API_BEGIN
	ix->impl->resolve_timing(count, total_ms);
	MVS_API_END
}
int mvs_index_set_option(mvs_index *ix, const char *key, int64_t value) {
	MVS_API_BEGIN
	if (!ix->impl->set_option(key, value))
		throw_faiss("mvs_index_set_option", __FILE__, "unknown option %s", key);
	MVS_API_END
}

} // extern "C"

namespace mvs {
namespace {
struct Roctx {
	int (*push)(const char *) = nullptr;
	int (*pop)() = nullptr;
	Roctx() {
		bool want = false;
		if (const char *e = getenv("MVS_ROCTX"))
			want = e[0] && e[0] != '0';
		for (const char *v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "HSA_TOOLS_LIB", "LD_PRELOAD"})
			if (const char *e = getenv(v))
				want = want || strstr(e, "rocprof") != nullptr;
		if (!want)
			return;
		for (const char *n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
			void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
			if (!h)
				continue;
			push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
			pop = (int (*)())dlsym(h, "roctxRangePop");
			if (push && pop)
				return;
			push = nullptr, pop = nullptr;
		}
	}
};
const Roctx &roctx() {
	static const Roctx r;
	return r;
}
} // namespace
void trace_push(const char *name) {
	const Roctx &r = roctx();
	if (r.push)
		(void)r.push(name);
}
void trace_pop() {
	const Roctx &r = roctx();
	if (r.pop)
		(void)r.pop();
}
// Per-index tuning (round 5).  Every knob below used to be a process-wide int set through whichever index
// happened to receive set_option -- DuckDB searches different indexes from different threads, so one index's A/B switch changed
// the path (and raced with the launches) of another.  Now an index owns its Tuning; use_device(), the first statement of every
// entry point, makes it the calling thread's current one; the launch functions read tune().x.
static thread_local const Tuning *t_tune = nullptr;
const Tuning &tune() {
	static const Tuning defaults;
	return t_tune ? *t_tune : defaults;
}
void set_current_tuning(const Tuning *t) {
	t_tune = t;
}
void forget_current_tuning(const Tuning *t) {
	if (t_tune == t)
		t_tune = nullptr;
}
bool IndexBase::set_tuning(const char *key, int64_t v) {
	struct Key {
		const char *name;
		int Tuning::*field;
		int mode; // 0: the value; 1: v != 0; 2: 2 or 3; 3: 4 or 8; 4: low two bits; 5: profiling library only; 6: the value, >= 100 profiling only
	};
	static const Key keys[] = {
	    {"cl_big_mode", &Tuning::big_mode, 4},         // flat_bf16_big_kernel pipeline A/B: bit 0 cross-tile fragment prefetch, bit 1 spread LDS-DMA
	    {"cl_wide_big", &Tuning::wide_big, 1},         // 512 < d <= 1024 coarse filter: one wave per SIMD, all of k resident (1) or the k-split kernel (0)
	    {"cl_wide512_ksplit", &Tuning::wide512_ksplit, 1}, // 384 < d <= 512 coarse filter on the k-split kernel (1) or on wide<16,1,2> (0)
	    {"cl_wide384_ncb", &Tuning::wide384_ncb, 2},   // 256 < d <= 384 coarse filter: column blocks per wave (2 | 3)
	    {"cl_ksplit_opt", &Tuning::ksplit_opt, 0},
	    {"cl_ksplit_ncb", &Tuning::ksplit_ncb, 2},     // column blocks per wave pair of the k-split coarse filter (2 | 3)
	    {"cl_ksplit_waves", &Tuning::ksplit_waves, 3}, // 512 < d <= 768: waves per workgroup of the k-split coarse filter (4 or 8)
	    {"ivf_cl_refresh", &Tuning::ivf_cl_refresh, 0}, // IVF coarse filter: tiles (32 rows) between two refreshes of a wave's bounds (0 = 1,1,1,1,4.. 16)
	    {"ivf_coarse_mfma", &Tuning::coarse_mfma, 0},  // IVF coarse distance matrix on the f32 matrix pipe (1) or the vector ALU (0); same bits
	    {"ivf_coarse_persistent", &Tuning::coarse_persistent, 0}, // coarse distance matrix: persistent workgroups (1) or one per tile (0, default: faster)
	    {"ivf_cl_lds_pad", &Tuning::ivf_cl_lds_pad, 0},
	    {"ivf_cl_xcd", &Tuning::ivf_cl_xcd, 0},        // IVF coarse filter: items of one list on one XCD (1) or dealt round-robin over the XCDs (0)
	    {"ivf_coarse_select", &Tuning::coarse_select, 1}, // IVF coarse quantiser: distance matrix + selection (1) or the k-list kernels (0)
	    {"ivf_coarse_bf16", &Tuning::coarse_bf16, 1},     // L2 coarse quantiser: bf16 filter + exact re-scoring (1, csrc/coarse_bf16.hip) or distance matrix + selection (0)
	    {"cl_abl", &Tuning::cl_abl, 5},                // wrong-result ablation knobs: profiling library only
	    {"ivf_cl_abl", &Tuning::ivf_cl_abl, 5},
	    {"coarse_abl", &Tuning::coarse_abl, 5},
	    {"pf_abl", &Tuning::pf_abl, 5},
	    {"cl_bound_mode", &Tuning::cl_bound_mode, 0},  // bf16 rounding term of the coarse filter's bound: actual residual norms (1) | worst case (0)
	    {"cl_tab", &Tuning::cl_tab, 1},                // d <= 128 scan: pass bounds through the global table (1, default) or derived per wave (0: round 3)
	    {"cl_nsplit", &Tuning::cl_nsplit, 0},          // coarse filter: row splits of the main scan (0 = planned)
	    {"cl_nc32_from", &Tuning::cl_nc32_from, 0},    // 32 row classes from this kk on (default 17: only where 16 classes cannot serve)
	    {"cl_seed_regs", &Tuning::cl_seed_regs, 0},    // d <= 128 pre-pass: class maxima in registers (1) or the scan kernel's rare path (0)
	    {"cl_seed_split", &Tuning::cl_seed_split, 0},
	    {"cl_seed_rows", &Tuning::cl_seed_rows, 0},    // coarse filter: rows of the bound-estimation pre-pass
	    {"cl_seed_reg_rows", &Tuning::cl_seed_reg_rows, 0},
	    {"pf_sched", &Tuning::pf_sched, 0},
	    {"pf_classes32", &Tuning::pf_classes32, 0},
	    {"pf_seed", &Tuning::pf_seed, 0},              // rows of the prefilter's seeding pre-pass (0 = off)
	    {"pf_nsplit", &Tuning::pf_nsplit, 0},
	    {"mfma_global_lists", &Tuning::mfma_global_lists, 0},
	    {"mfma_warm", &Tuning::mfma_warm, 0},
	    {"mfma_nsplit", &Tuning::mfma_nsplit, 0},
	    {"mfma_variant", &Tuning::mfma_variant, 6},    // A/B switch between kernel generations; >= 100: ablations, profiling library only
	};
	for (const Key &e : keys) {
		if (strcmp(key, e.name))
			continue;
		int val = (int)v;
		switch (e.mode) {
		case 1: val = v != 0; break;
		case 2: val = v == 2 ? 2 : 3; break;
		case 3: val = v == 4 ? 4 : 8; break;
		case 4: val = (int)(v & 3); break;
		case 5:
#ifndef MVS_PROFILING
			return false;
#endif
			break;
		case 6:
#ifndef MVS_PROFILING
			if (v >= 100)
				return false;
#endif
			break;
		default: break;
		}
		tune_.*(e.field) = val;
		return true;
	}
	return false;
}
bool FlatIndex::set_option(const char *key, int64_t v) {
	if (set_tuning(key, v))
		return true;
	if (!strcmp(key, "outlier_rows")) { // 0: no row is kept out of the coarse-filter store (round 5; takes effect at the store's next first build)
		outlier_rows = v != 0;
		return true;
	}
	if (!strcmp(key, "lazy_adds")) { // 0: every add() reaches the device at once (round 5's ingest, for A/B)
		flush_adds();
		lazy_adds = v != 0;
		return true;
	}
	if (!strcmp(key, "metric_arg_bits")) { // faiss::Index::metric_arg as IEEE-754 bits (the option channel carries integers)
		const uint32_t b = (uint32_t)v;
		memcpy(&metric_arg, &b, 4);
		return true;
	}
	if (!strcmp(key, "force_staged")) { // per-pair path on the LDS-staged flat_direct kernel instead of the scan kernel
		force_staged = v != 0;
		return true;
	}
	if (!strcmp(key, "prefilter")) { // -1 auto, 0 off (exact f32 kernel only), 1 wherever the bf16x3 kernel supports the shape
		prefilter_mode = (int)v;
		return true;
	}
	if (!strcmp(key, "tie_from_candidates")) {
		tie_from_candidates = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_k32")) {
		cl_k32 = v != 0;
		return true;
	}
	if (!strcmp(key, "flat_shadow")) { // -1 auto (built when the coarse filter admits thousands of rows per query), 0 never, 1 at once
		shadow_mode = (int)v;
		if (v != 0 && shadow_state < 0)
			shadow_state = 0;
		return true;
	}
	if (!strcmp(key, "flat_shadow_nprobe")) {
		shadow_nprobe = (int)std::max<int64_t>(1, v);
		return true;
	}
	if (!strcmp(key, "cl_seed_stage")) { // 0: the register pre-pass publishes its class maxima with atomics (rounds 3-4)
		cl_seed_stage = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_wide_refilter")) {
		cl_wide_refilter = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_fbucket")) { // 0: the sorted pipeline behind the d = 128 L2 coarse filter (round 4); 1: the bucketed finish
		cl_fbucket = v != 0;
		cl_fbucket_off = false;
		return true;
	}
	if (!strcmp(key, "cl_fpitch")) { // bucket entries per query of the bucketed finish (tests: provokes the overflow)
		cl_fpitch = (int)std::max<int64_t>(64, std::min<int64_t>(16384, (v + 63) / 64 * 64));
		cl_fbucket_off = false;
		return true;
	}
	if (!strcmp(key, "cl_prep1")) { // 0: round 4's separate query-preparation kernels (A/B)
		cl_prep1 = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_bigk_whole")) { // big lists, pass A: -1 auto, 0 a quarter of the rows, 1 all of them (A/B)
		cl_bigk_whole = (int)v;
		return true;
	}
	if (!strcmp(key, "cl_bigk_per")) { // big lists, d <= 128: rows of a pass-A split that decide its bound (1 .. 12; default 8)
		cl_bigk_per = (int)std::min<int64_t>(12, std::max<int64_t>(0, v));
		return true;
	}
	if (!strcmp(key, "cl_bigk")) { // 0: lists beyond 128 entries on the exact kernels (round 5; A/B)
		cl_bigk = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_small_path")) { // 0: small batches on flat_bf16_collect_kernel as well (A/B)
		cl_small_path = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_est")) { // tests: pretend the previous search had v candidates per query (a sort sized too small is re-run)
		cl_est_per_query = (double)v + 1e-6;
		return true;
	}
	if (!strcmp(key, "cl_defer_count")) { // 1 (default): no host round trip for the candidate count between scan and re-scoring
		cl_defer = v != 0;
		return true;
	}
	if (!strcmp(key, "cl_stream_cap")) { // coarse filter: candidate-stream entries per query (diagnostics: provoke the overflow paths)
		cl_stream_cap_per_query = (int)v;
		return true;
	}
	if (!strcmp(key, "pf_margin")) {
		pf_margin = (int)std::min<int64_t>(16, std::max<int64_t>(1, v));
		return true;
	}
	if (!strcmp(key, "raw_rows")) {
		raw_rows = v != 0;
		return true;
	}
	if (!strcmp(key, "ip_exact_ties")) { // 0: pure (score desc, id asc) order, no tie pass (diagnostics)
		ip_exact_ties = v != 0;
		return true;
	}
	if (!strcmp(key, "force_direct")) {
		force_direct = v != 0;
		return true;
	}
	return false;
}
} // namespace mvs
