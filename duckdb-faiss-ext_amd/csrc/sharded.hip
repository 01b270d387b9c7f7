// csrc/sharded.hip -- one index over several MI355X devices, INSIDE the library.
//
// The reference's GPU hook addresses ONE device (faiss::gpu::index_cpu_to_gpu(res, device, index),
// /root/reference/src/gpu/gpu.cpp:48, registered src/faiss_extension.cpp:1042-1048).  A DuckDB process that loads the
// extension therefore reaches more than one GPU only if the sharding lives below that surface: faiss_to_gpu(name, -1)
// (or env MVS_DEVICES=0,1,...,7 at index_factory / read_index time) yields a ShardedIndex, and every later
// Index::add / add_with_ids / train / search of the unmodified glue (:396,:510,:512,:583,:607,:609,:631) fans out.
//
//   Flat / IDMap,Flat     ROW SHARDS.  A DataChunk-sized add (<= 2048 rows, :475-547) goes whole to one device
//                         (round robin; large adds are cut into one contiguous piece per device).  Every shard
//                         keeps, per row, the GLOBAL row number (FAISS's internal id: the order that breaks ties) and
//                         the user label.
//   IVF<n>,Flat           centroids trained once on the global training set and replicated; every inverted list is
//                         row-sharded; the lists store global row numbers.
//   HNSW<M>               REPLICAS ONLY (the graph walk does not shard): queries are split across devices.
//
// search = one host thread + stream per device (raw shard search: pure order, k+1 candidates for inner product), ONE
// exchange, host k-way merge, and for inner product the cross-shard tie pass (FlatIndex::tie_candidates).  Exchange
// backends (option "shard_exchange" / env MVS_SHARD_EXCHANGE):
//   host : per-device D2H of the (value, global row) blocks into pinned memory, merge on the host
//   rccl : ONE ncclAllGather of packed 16-byte {value, global row} records over xGMI (in-process communicators,
//          ncclCommInitAll), D2H from the first device, the same host merge     [north_star's exchange]
// RCCL is bound at run time (dlopen): the library keeps no link-time dependency on it, and virtual shards that share
// one device (tests on a 1-GPU box) can only use the host backend -- RCCL refuses two ranks on one device.
#include "index.h"

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <thread>

namespace mvs {

void merge_raw_lists_host(int metric, int64_t nq, int64_t kk, int nshard, const float *const *D, const int64_t *const *G,
                          float *val, int64_t *gnum);
void resolve_ip_tie_host(int64_t k, const float *raw_v, const int64_t *raw_g, const int64_t *first, float *out_v,
                         int64_t *out_g);

namespace {

// ---- one persistent host thread per device --------------------------------------------------------------------
class Worker {
public:
	explicit Worker(int device) : dev(device), th([this] { loop(); }) {
	}
	~Worker() {
		{
			std::lock_guard<std::mutex> g(mu);
			stop = true;
		}
		cv.notify_all();
		th.join();
	}
	void run(std::function<void()> f) {
		std::unique_lock<std::mutex> g(mu);
		cv.wait(g, [this] { return !busy; });
		job = std::move(f);
		busy = true;
		err = nullptr;
		g.unlock();
		cv.notify_all();
	}
	void wait() {
		std::unique_lock<std::mutex> g(mu);
		cv.wait(g, [this] { return !busy; });
		if (err) {
			auto e = err;
			err = nullptr;
			std::rethrow_exception(e);
		}
	}

private:
	void loop() {
		(void)hipSetDevice(dev);
		for (;;) {
			std::function<void()> f;
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [this] { return stop || (busy && job); });
				if (stop)
					return;
				f = std::move(job);
				job = nullptr;
			}
			std::exception_ptr e;
			try {
				f();
			} catch (...) {
				e = std::current_exception();
			}
			{
				std::lock_guard<std::mutex> g(mu);
				err = e;
				busy = false;
			}
			cv.notify_all();
		}
	}
	int dev;
	std::mutex mu;
	std::condition_variable cv;
	std::function<void()> job;
	bool busy = false, stop = false;
	std::exception_ptr err;
	std::thread th; // last member: starts after the state above is constructed
};

// ---- RCCL, bound at run time -------------------------------------------------------------------------------------
struct Rccl {
	typedef int (*CommInitAll)(void **, int, const int *);
	typedef int (*CommDestroy)(void *);
	typedef int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t);
	typedef int (*Group)(void);
	typedef const char *(*ErrStr)(int);
	CommInitAll comm_init_all = nullptr;
	CommDestroy comm_destroy = nullptr;
	AllGather all_gather = nullptr;
	Group group_start = nullptr, group_end = nullptr;
	ErrStr err_str = nullptr;
	std::string why;
	bool ok = false;
	Rccl() {
		// RCCL must sit on the HIP runtime this process already uses (PyTorch wheels bundle libamdhip64 AND librccl; two
		// HIP runtimes in one process corrupt each other): look next to the loaded libamdhip64 first
		void *h = nullptr;
		std::vector<std::string> names;
		Dl_info info;
		if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
			std::string dir(info.dli_fname);
			const size_t sl = dir.rfind('/');
			if (sl != std::string::npos) {
				dir.resize(sl + 1);
				names.push_back(dir + "librccl.so");
				names.push_back(dir + "librccl.so.1");
			}
		}
		for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
			names.push_back(n);
		for (const std::string &n : names) {
			// (RTLD_LOCAL: only dlsym on the handle is used.  With RTLD_GLOBAL a PyTorch imported LATER in the same process bound some of
			// its symbols into this copy and the process aborted at exit with "double free or corruption")
			h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
			if (h)
				break;
		}
		if (!h) {
			why = std::string("librccl.so could not be loaded: ") + dlerror();
			return;
		}
		comm_init_all = (CommInitAll)dlsym(h, "ncclCommInitAll");
		comm_destroy = (CommDestroy)dlsym(h, "ncclCommDestroy");
		all_gather = (AllGather)dlsym(h, "ncclAllGather");
		group_start = (Group)dlsym(h, "ncclGroupStart");
		group_end = (Group)dlsym(h, "ncclGroupEnd");
		err_str = (ErrStr)dlsym(h, "ncclGetErrorString");
		ok = comm_init_all && comm_destroy && all_gather && group_start && group_end;
		if (!ok)
			why = "librccl.so lacks ncclCommInitAll / ncclAllGather / ncclGroupStart";
	}
	static Rccl &get() {
		static Rccl r;
		return r;
	}
};
constexpr int NCCL_INT8 = 0; // ncclInt8 / ncclChar (rccl.h ncclDataType_t)

struct Rec { // packed exchange record
	float v;
	int32_t pad;
	int64_t g;
};

__global__ void map_rows_kernel(long long *I, long long total, const long long *gnum) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total) {
		const long long r = I[i];
		I[i] = r < 0 ? -1ll : gnum[r];
	}
}
__global__ void pack_records_kernel(const float *D, const long long *G, long long total, Rec *out) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total) {
		Rec r;
		r.v = D[i];
		r.pad = 0;
		r.g = G[i];
		out[i] = r;
	}
}
// merged pure lists (value, global row) [nq][kk] -> D / I [nq][k]: inner product prints runs of equal scores in descending
// row order (heap_reorder over a CMin heap), labels = id map or row + offset; flag: {count, queries...} of the queries whose
// k-th and (k+1)-th scores are bit-equal (the cross-shard tie pass decides those)
__global__ void sharded_finish_kernel(const float *__restrict__ mv, const long long *__restrict__ mg, int kk, int k, long long nq,
                                      int is_l2, const long long *__restrict__ idmap, long long label_offset,
                                      float *__restrict__ D, long long *__restrict__ I, int *__restrict__ flag) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nq * k)
		return;
	const long long q = i / k;
	const int j = (int)(i - q * k);
	const float *v = mv + q * kk;
	const long long *gq = mg + q * kk;
	int src = j;
	if (!is_l2 && gq[j] >= 0) {
		int a = j, b = j;
		while (a > 0 && gq[a - 1] >= 0 && v[a - 1] == v[j])
			--a;
		while (b + 1 < k && gq[b + 1] >= 0 && v[b + 1] == v[j])
			++b;
		src = a + (b - j);
	}
	const long long g = gq[src];
	D[i] = v[src];
	I[i] = g < 0 ? -1ll : (idmap ? idmap[g] : g + label_offset);
	if (flag && j == k - 1 && kk > k && gq[k] >= 0 && v[k] == v[k - 1])
		flag[1 + atomicAdd(flag, 1)] = (int)q;
}
__global__ void iota_kernel(long long *out, long long n, long long start) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = start + i;
}

struct GrowI64 { // device int64 array that keeps its contents when it grows
	int64_t *p = nullptr;
	int64_t cap = 0;
	void ensure(int64_t need, int64_t used, hipStream_t st) {
		if (need <= cap)
			return;
		int64_t nc = cap ? cap : 4096;
		while (nc < need)
			nc = nc + nc / 2 + 4096;
		int64_t *np = nullptr;
		MVS_HIP(hipMalloc((void **)&np, (size_t)nc * sizeof(int64_t)));
		if (used > 0)
			MVS_HIP(hipMemcpyAsync(np, p, (size_t)used * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
		MVS_HIP(hipStreamSynchronize(st));
		if (p)
			MVS_HIP(hipFree(p));
		p = np;
		cap = nc;
	}
	void release() {
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
};

struct HostPinned {
	void *p = nullptr;
	size_t cap = 0;
	void *get(size_t bytes, bool portable = false) { // portable: every device of the node may DMA from / into it
		if (bytes > cap) {
			if (p)
				(void)hipHostFree(p);
			p = nullptr;
			MVS_HIP(hipHostMalloc(&p, bytes + bytes / 4 + 256, portable ? hipHostMallocPortable : hipHostMallocDefault));
			cap = bytes + bytes / 4 + 256;
		}
		return p;
	}
	~HostPinned() {
		if (p)
			(void)hipHostFree(p);
	}
};

} // namespace

class ShardedIndex : public IndexBase {
public:
	enum Mode { ROWS_FLAT, ROWS_IVF, REPLICAS };
	Mode mode;
	bool has_idmap = false;
	int G;
	std::vector<int> devs;
	std::vector<IndexBase *> shards;
	std::vector<std::unique_ptr<Worker>> workers;
	std::vector<hipStream_t> streams; // the device work of a call runs on the shard's own stream
	// ROWS_FLAT: per shard, per row: global row number and (IDMap) user label
	std::vector<GrowI64> gnum_dev, label_dev;
	// ROWS_IVF + IDMap: the whole id_map on every device (the selector tests id_map[stored global row])
	std::vector<GrowI64> idmap_dev;
	std::vector<int64_t> idmap_host; // global row -> user label (IDMap)
	int next_shard = 0;
	int exchange = 0; // 0 host gather, 1 rccl
	std::vector<void *> comms;
	// per-shard scratch
	struct Scratch {
		DevBuf x, D, I, rec, recv, T, rows;
		DevBuf mv, mg, oD, oI, flag; // first device only: merged pure lists, finished results, tie flags {count, queries...}
		HostPinned hx, hD, hI, hT, hrows;
		void release() {
			for (DevBuf *b : {&x, &D, &I, &rec, &recv, &T, &rows, &mv, &mg, &oD, &oI, &flag})
				b->release();
		}
	};
	std::vector<Scratch> sc;
	HostPinned hq, hout;          // the batch's queries (one pinned copy for all devices) and its finished results
	std::vector<hipEvent_t> ev;   // per shard: its records have arrived on the first device
	hipEvent_t ev_x = nullptr;    // device-pointer entry: the caller's stream has produced the queries
	int64_t last_flagged = 0; // diagnostics: queries the last search sent through the cross-shard tie pass

	ShardedIndex(int d_, const std::string &desc, int metric_, const std::vector<int> &devices)
	    : IndexBase(MVS_KIND_FLAT, d_, metric_), G((int)devices.size()), devs(devices) {
		if (G < 1)
			throw_faiss("mvs::ShardedIndex", __FILE__, "no devices");
		std::string inner = desc;
		if (inner.rfind("IDMap2,", 0) == 0 || inner.rfind("IDMap,", 0) == 0) {
			has_idmap = true;
			inner = inner.substr(inner.find(',') + 1);
		}
		if (inner == "Flat")
			mode = ROWS_FLAT;
		else if (inner.rfind("IVF", 0) == 0)
			mode = ROWS_IVF;
		else
			mode = REPLICAS; // HNSW...: index_factory below rejects what it cannot parse
		device = devs[0];
		try {
			for (int g = 0; g < G; ++g) {
				CtorDevice scope(devs[g]);
				// replicas keep their own IDMap wrapper (the whole index lives on every device)
				shards.push_back(index_factory(d, mode == REPLICAS ? desc.c_str() : inner.c_str(), metric));
				if (mode == ROWS_FLAT)
					shards.back()->set_option("raw_rows", 1);
				if (mode == ROWS_IVF) {
					shards.back()->set_option("ivf_raw_ids", 1);
					// exact distance ties are resolved ACROSS the shards (resolve_ties_ivf): a shard hands over its pure (value, id) order
					shards.back()->set_option("ivf_exact_ties", 0);
				}
			}
		} catch (...) {
			for (auto *s : shards)
				delete s;
			throw;
		}
		init_common();
	}
	// replicas built elsewhere (hnsw_from_host on every device)
	ShardedIndex(std::vector<IndexBase *> reps, const std::vector<int> &devices, bool idmap)
	    : IndexBase(MVS_KIND_FLAT, reps[0]->d, reps[0]->metric), mode(REPLICAS), has_idmap(idmap), G((int)devices.size()),
	      devs(devices), shards(std::move(reps)) {
		device = devs[0];
		init_common();
		ntotal = shards[0]->ntotal;
	}
	void init_common() {
		kind = has_idmap ? MVS_KIND_IDMAP : shards[0]->kind;
		is_trained = shards[0]->is_trained;
		gnum_dev.resize(G);
		label_dev.resize(G);
		idmap_dev.resize(G);
		sc.resize(G);
		for (int g = 0; g < G; ++g) {
			workers.emplace_back(new Worker(devs[g]));
			streams.push_back(shards[g]->stream);
		}
		if (const char *e = getenv("MVS_SHARD_EXCHANGE"))
			exchange = !strcmp(e, "rccl") ? 1 : 0;
	}
	~ShardedIndex() override {
		for (auto &w : workers) {
			try {
				w->wait();
			} catch (...) {
			}
		}
		workers.clear();
		for (int g = 0; g < G && g < (int)ev.size(); ++g)
			if (ev[g]) {
				(void)hipSetDevice(devs[g]);
				(void)hipEventDestroy(ev[g]);
			}
		ev.clear();
		if (ev_x)
			(void)hipEventDestroy(ev_x);
		ev_x = nullptr;
		for (void *c : comms)
			if (c && !getenv("MVS_RCCL_KEEP_COMMS")) // (diagnostic: leave the communicators to the process's exit)
				Rccl::get().comm_destroy(c);
		comms.clear();
		for (int g = 0; g < G; ++g) {
			(void)hipSetDevice(devs[g]);
			gnum_dev[g].release();
			label_dev[g].release();
			idmap_dev[g].release();
			sc[g].release();
			delete shards[g];
		}
		(void)hipSetDevice(device);
	}

	// what the glue's dynamic_casts see below an IDMap / for IVF and HNSW parameters (borrowed)
	IndexBase *inner_view() {
		if (mode == REPLICAS && has_idmap)
			return static_cast<IDMapIndex *>(shards[0])->sub;
		return shards[0];
	}

	template <typename F>
	void on_all(F &&f) {
		for (int g = 0; g < G; ++g)
			workers[g]->run([&f, g] { f(g); });
		std::exception_ptr first;
		for (int g = 0; g < G; ++g) {
			try {
				workers[g]->wait();
			} catch (...) {
				if (!first)
					first = std::current_exception();
			}
		}
		if (first)
			std::rethrow_exception(first);
	}

	// ------------------------------------------------------------------------------------------------ train
	void train(int64_t n, const float *x) override {
		if (mode == ROWS_IVF) {
			// the reference trains on ALL rows it was given (:583): one k-means on the first device, the nlist x d
			// centroids replicated (2 MB at IVF4096), so every device probes the same lists
			shards[0]->train(n, x);
			const int64_t nlist = ivf_nlist_of(shards[0]);
			if (ivf_quantizer_of(shards[0])->kind == MVS_KIND_FLAT) {
				std::vector<float> cent((size_t)nlist * d);
				ivf_get_centroids(shards[0], cent.data());
				for (int g = 1; g < G; ++g)
					ivf_set_centroids(shards[g], cent.data());
			} else { // IVF<n>_HNSW<m>: every device trains its own (deterministic k-means, own coarse graph)
				for (int g = 1; g < G; ++g)
					shards[g]->train(n, x);
			}
		} else {
			on_all([&](int g) { shards[g]->train(n, x); });
		}
		is_trained = shards[0]->is_trained;
	}

	// ------------------------------------------------------------------------------------------------ add
	struct Piece {
		int g;
		int64_t off, cnt;
	};
	std::vector<Piece> plan_add(int64_t n) {
		std::vector<Piece> p;
		if (n >= (int64_t)G * 4096) {
			for (int g = 0; g < G; ++g) {
				const int64_t a = n * g / G, b = n * (g + 1) / G;
				if (b > a)
					p.push_back({g, a, b - a});
			}
		} else {
			p.push_back({next_shard, 0, n});
			next_shard = (next_shard + 1) % G;
		}
		return p;
	}
	void add_rows(int64_t n, const float *x, const int64_t *ids) {
		if (n <= 0)
			return;
		if (mode == REPLICAS) {
			// the glue assigns hnsw.efConstruction on the wrapper of the first replica (:136-139): spread it
			const int efc = hnsw_get_ef_construction(unwrapped(shards[0]));
			for (int g = 1; g < G && efc > 0; ++g)
				hnsw_set_ef_construction(unwrapped(shards[g]), efc);
			on_all([&](int g) {
				if (ids)
					shards[g]->add_with_ids(n, x, ids);
				else
					shards[g]->add(n, x);
			});
			ntotal = shards[0]->ntotal;
			return;
		}
		if (!is_trained)
			throw_faiss("virtual void faiss::IndexIVF::add_core(...)", "faiss/IndexIVF.cpp", "Error: 'is_trained' failed");
		const int64_t base = ntotal;
		std::vector<Piece> pieces = plan_add(n);
		std::vector<std::vector<Piece>> per(G);
		for (auto &p : pieces)
			per[p.g].push_back(p);
		if (has_idmap)
			idmap_host.insert(idmap_host.end(), ids, ids + n);
		on_all([&](int g) {
			for (const Piece &p : per[g]) {
				IndexBase *s = shards[g];
				const int64_t have = s->ntotal;
				if (mode == ROWS_FLAT) {
					s->add(p.cnt, x + p.off * d);
					hipStream_t st = s->stream;
					gnum_dev[g].ensure(have + p.cnt, have, st);
					hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((p.cnt + 255) / 256)), dim3(256), 0, st,
					                   (long long *)gnum_dev[g].p + have, (long long)p.cnt, (long long)(base + p.off));
					if (has_idmap) {
						label_dev[g].ensure(have + p.cnt, have, st);
						// ids are the caller's pageable memory, valid during this call only
						MVS_HIP(hipMemcpyAsync(label_dev[g].p + have, ids + p.off, (size_t)p.cnt * sizeof(int64_t),
						                       hipMemcpyHostToDevice, st));
						MVS_HIP(hipStreamSynchronize(st));
					}
				} else { // ROWS_IVF: the lists store global row numbers
					std::vector<int64_t> gn((size_t)p.cnt);
					for (int64_t i = 0; i < p.cnt; ++i)
						gn[(size_t)i] = base + p.off + i;
					s->add_with_ids(p.cnt, x + p.off * d, gn.data());
				}
			}
			if (has_idmap && (mode == ROWS_IVF || g == 0)) { // the whole id_map on every device (Flat: the merging device only)
				hipStream_t st = shards[g]->stream;
				idmap_dev[g].ensure(base + n, base, st);
				MVS_HIP(hipMemcpyAsync(idmap_dev[g].p + base, ids, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
				MVS_HIP(hipStreamSynchronize(st));
			}
		});
		ntotal = base + n;
	}
	void add(int64_t n, const float *x) override {
		if (has_idmap)
			throw_faiss("virtual void faiss::IndexIDMapTemplate<IndexT>::add(faiss::idx_t, const float*)",
			            "faiss/IndexIDMap.cpp", "add does not make sense with IndexIDMap, use add_with_ids");
		add_rows(n, x, nullptr);
	}
	void add_with_ids(int64_t n, const float *x, const int64_t *ids) override {
		if (!has_idmap && mode != ROWS_IVF) { // IndexFlat / IndexHNSW: faiss4.test:19-22
			IndexBase::add_with_ids(n, x, ids);
			return;
		}
		if (!has_idmap) { // plain IndexIVF::add_with_ids: the given ids ARE the stored ids
			// (row shards order ties by the stored id; an IDMap on top keeps FAISS's internal numbering instead)
			if (n <= 0)
				return;
			std::vector<Piece> pieces = plan_add(n);
			std::vector<std::vector<Piece>> per(G);
			for (auto &p : pieces)
				per[p.g].push_back(p);
			on_all([&](int g) {
				for (const Piece &p : per[g])
					shards[g]->add_with_ids(p.cnt, x + p.off * d, ids + p.off);
			});
			ntotal += n;
			return;
		}
		add_rows(n, x, ids);
	}
	static IndexBase *unwrapped(IndexBase *p) {
		while (p->kind == MVS_KIND_IDMAP)
			p = static_cast<IDMapIndex *>(p)->sub;
		return p;
	}
	void add_device(int64_t n, const float *d_x, hipStream_t st) override {
		std::vector<float> hx((size_t)n * d);
		MVS_HIP(hipMemcpyAsync(hx.data(), d_x, hx.size() * sizeof(float), hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		add(n, hx.data());
	}
	void add_with_ids_device(int64_t n, const float *d_x, const int64_t *d_ids, hipStream_t st) override {
		std::vector<float> hx((size_t)n * d);
		std::vector<int64_t> hi((size_t)n);
		MVS_HIP(hipMemcpyAsync(hx.data(), d_x, hx.size() * sizeof(float), hipMemcpyDeviceToHost, st));
		MVS_HIP(hipMemcpyAsync(hi.data(), d_ids, hi.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
		MVS_HIP(hipStreamSynchronize(st));
		add_with_ids(n, hx.data(), hi.data());
	}

	// ------------------------------------------------------------------------------------------------ search
	// Row shards: every device searches its rows (pure order, GLOBAL row numbers, k + 1 entries for inner product), packs
	// 16-byte {value, global row} records, the records of all shards meet on the FIRST device -- peer copies over xGMI
	// ("host" exchange: nothing passes through the host any more) or ONE ncclAllGather -- and are merged THERE
	// (merge_records_kernel, one wave per query); a finish kernel prints FAISS's order with the user labels and flags
	// exact inner-product ties at rank k.  The host sees nq x k results and one flag count.  (Round 2 merged G x nq x k
	// candidates on 16 host threads and staged queries and results of the device-pointer entry through the host.)
	void search(int64_t nq, const float *x, int64_t k, float *D, int64_t *I, const mvs_search_params *params) override {
		if (k <= 0)
			throw_faiss("virtual void faiss::Index::search(...) const", "faiss/Index.cpp", "Error: 'k > 0' failed");
		if (nq <= 0)
			return;
		if (mode == REPLICAS) {
			const int64_t per = (nq + G - 1) / G;
			on_all([&](int g) {
				const int64_t q0 = std::min(nq, per * g), q1 = std::min(nq, per * (g + 1));
				if (q1 > q0)
					shards[g]->search(q1 - q0, x + q0 * d, k, D + q0 * k, I + q0 * k, params);
			});
			kinfo = shards[0]->kinfo;
			return;
		}
		// the caller's pageable queries: ONE copy into pinned memory every device reads its H2D from
		const size_t xbytes = (size_t)nq * d * sizeof(float);
		float *hx = (float *)hq.get(xbytes, true);
		memcpy(hx, x, xbytes);
		search_rows(nq, hx, nullptr, -1, nullptr, k, params, D, I, nullptr, nullptr, x);
	}

	void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   hipStream_t st) override {
		if (k <= 0)
			throw_faiss("virtual void faiss::Index::search(...) const", "faiss/Index.cpp", "Error: 'k > 0' failed");
		if (nq <= 0)
			return;
		if (mode == REPLICAS) { // (replicas split the batch: staged through the host as the host-pointer entry does)
			std::vector<float> hxv((size_t)nq * d), hD((size_t)nq * k);
			std::vector<int64_t> hI((size_t)nq * k);
			MVS_HIP(hipMemcpyAsync(hxv.data(), d_x, hxv.size() * sizeof(float), hipMemcpyDeviceToHost, st));
			MVS_HIP(hipStreamSynchronize(st));
			search(nq, hxv.data(), k, hD.data(), hI.data(), params);
			MVS_HIP(hipMemcpyAsync(d_D, hD.data(), hD.size() * sizeof(float), hipMemcpyHostToDevice, st));
			MVS_HIP(hipMemcpyAsync(d_I, hI.data(), hI.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
			MVS_HIP(hipStreamSynchronize(st));
			return;
		}
		// queries stay on the devices: the shard on the caller's device reads them in place, the others fetch them with a
		// peer copy once the caller's stream has produced them
		hipPointerAttribute_t at;
		MVS_HIP(hipPointerGetAttributes(&at, d_x));
		const int xdev = at.device;
		MVS_HIP(hipSetDevice(xdev));
		if (!ev_x)
			MVS_HIP(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming));
		MVS_HIP(hipEventRecord(ev_x, st));
		search_rows(nq, nullptr, d_x, xdev, ev_x, k, params, nullptr, nullptr, d_D, d_I, nullptr);
	}

	// hx: pinned host queries (host-pointer entry) or d_x on device xdev, valid once x_ready has passed (device-pointer entry);
	// results to D / I (host) or d_D / d_I (device memory of any device); x_pageable: the caller's queries for the (rare)
	// cross-shard tie pass, fetched from d_x when null
	void search_rows(int64_t nq, const float *hx, const float *d_x, int xdev, hipEvent_t x_ready, int64_t k,
	                 const mvs_search_params *params, float *D, int64_t *I, float *d_D, int64_t *d_I, const float *x_pageable) {
		const bool is_l2 = metric_order(metric) == METRIC_L2;
		// inner product: one extra candidate per list detects an exact tie at the k-th score (FlatIndex::search_flat)
		// (k >= 100: FAISS's reservoir, whose outcome depends on the global interleaving of the shards' rows -- pure order there)
		// IVF row shards (round 4): FAISS's heap sees the probed lists' rows in ARRIVAL order (probe rank, position in the list); the
		// shards hand over k + 1 entries in the pure order and the queries tied at the k-th value take resolve_ties_ivf
		const bool ivf_ties = mode == ROWS_IVF && ivf_exact_ties && ntotal > k && k + 1 <= 1024;
		const bool tie_detect = (metric == METRIC_IP && mode == ROWS_FLAT && ntotal > k && k + 1 <= 256 && k < 100) || ivf_ties;
		const int64_t kk = tie_detect ? k + 1 : k;
		const size_t cells = (size_t)nq * kk, ocells = (size_t)nq * k;
		const bool use_rccl = exchange == 1 && ensure_comms();
		Scratch &w0 = sc[0];
		MVS_HIP(hipSetDevice(devs[0]));
		w0.recv.reserve(cells * sizeof(Rec) * G);
		w0.mv.reserve(cells * sizeof(float));
		w0.mg.reserve(cells * sizeof(int64_t));
		w0.oD.reserve(ocells * sizeof(float));
		w0.oI.reserve(ocells * sizeof(int64_t));
		w0.flag.reserve((size_t)(nq + 16) * sizeof(int));
		if (ev.empty()) {
			ev.assign((size_t)G, nullptr);
			for (int g = 0; g < G; ++g) {
				MVS_HIP(hipSetDevice(devs[g]));
				MVS_HIP(hipEventCreateWithFlags(&ev[g], hipEventDisableTiming));
			}
			MVS_HIP(hipSetDevice(devs[0]));
		}
		trace_push("mvs:shard_search (every shard, its own stream)");
		on_all([&](int g) { shard_search(g, nq, hx, d_x, xdev, x_ready, kk, params, use_rccl); });
		trace_pop();
		hipStream_t s0 = streams[0];
		TraceRange tr_x(use_rccl ? "mvs:exchange (ncclAllGather of 16-byte records) + merge" : "mvs:exchange (peer copies) + merge");
		if (use_rccl) {
			// ONE all-gather of the packed {value, global row} records over xGMI; the first device merges its copy
			Rccl &r = Rccl::get();
			check_nccl(r.group_start(), "ncclGroupStart");
			for (int g = 0; g < G; ++g) {
				MVS_HIP(hipSetDevice(devs[g]));
				sc[g].recv.reserve(cells * sizeof(Rec) * G);
				check_nccl(r.all_gather(sc[g].rec.p, sc[g].recv.p, cells * sizeof(Rec), NCCL_INT8, comms[g], streams[g]),
				           "ncclAllGather");
			}
			check_nccl(r.group_end(), "ncclGroupEnd");
			MVS_HIP(hipSetDevice(devs[0]));
		} else {
			MVS_HIP(hipSetDevice(devs[0]));
			for (int g = 1; g < G; ++g)
				MVS_HIP(hipStreamWaitEvent(s0, ev[g], 0)); // shard g's records have landed in w0.recv
		}
		launch_merge_records(metric, (const int64_t *)w0.recv.p, G, nq, (int)kk, (int)kk, true, (float *)w0.mv.p, (int64_t *)w0.mg.p, s0);
		// FAISS's print order, user labels, tie flags -- on the device
		const int64_t *idmap0 = has_idmap ? idmap_dev[0].p : nullptr;
		MVS_HIP(hipMemsetAsync(w0.flag.p, 0, sizeof(int), s0));
		hipLaunchKernelGGL(sharded_finish_kernel, dim3((unsigned)((ocells + 255) / 256)), dim3(256), 0, s0, (const float *)w0.mv.p,
		                   (const long long *)w0.mg.p, (int)kk, (int)k, (long long)nq, is_l2 ? 1 : 0, (const long long *)idmap0,
		                   (long long)label_offset, (float *)w0.oD.p, (long long *)w0.oI.p, tie_detect ? (int *)w0.flag.p : nullptr);
		MVS_HIP(hipGetLastError());
		const size_t d_off = 64, i_off = d_off + ((ocells * sizeof(float) + 7) & ~(size_t)7);
		char *ho = (char *)hout.get(i_off + ocells * sizeof(int64_t));
		int *hflag = (int *)ho;
		float *hD = (float *)(ho + d_off);
		int64_t *hI = (int64_t *)(ho + i_off);
		MVS_HIP(hipMemcpyAsync(hflag, w0.flag.p, sizeof(int), hipMemcpyDeviceToHost, s0));
		if (D) {
			MVS_HIP(hipMemcpyAsync(hD, w0.oD.p, ocells * sizeof(float), hipMemcpyDeviceToHost, s0));
			MVS_HIP(hipMemcpyAsync(hI, w0.oI.p, ocells * sizeof(int64_t), hipMemcpyDeviceToHost, s0));
		} else {
			copy_between(d_D, w0.oD.p, devs[0], ocells * sizeof(float), s0);
			copy_between(d_I, w0.oI.p, devs[0], ocells * sizeof(int64_t), s0);
		}
		for (int g = use_rccl ? 0 : G; g < G; ++g) { // (rccl: every rank's gather must finish before its buffers are reused)
			MVS_HIP(hipSetDevice(devs[g]));
			MVS_HIP(hipStreamSynchronize(streams[g]));
		}
		MVS_HIP(hipSetDevice(devs[0]));
		MVS_HIP(hipStreamSynchronize(s0));
		if (D) {
			memcpy(D, hD, ocells * sizeof(float));
			memcpy(I, hI, ocells * sizeof(int64_t));
		}
		const int nflag = tie_detect ? *hflag : 0;
		last_flagged = nflag;
		kinfo = shards[0]->kinfo;
		if (nflag <= 0)
			return;
		// ---- exact inner-product ties at rank k across shards (rare): the host finishes those queries -------------------
		std::vector<int> fq((size_t)nflag);
		std::vector<float> mv(cells);
		std::vector<int64_t> mg(cells);
		MVS_HIP(hipMemcpy(fq.data(), (const int *)w0.flag.p + 1, (size_t)nflag * sizeof(int), hipMemcpyDeviceToHost));
		MVS_HIP(hipMemcpy(mv.data(), w0.mv.p, cells * sizeof(float), hipMemcpyDeviceToHost));
		MVS_HIP(hipMemcpy(mg.data(), w0.mg.p, cells * sizeof(int64_t), hipMemcpyDeviceToHost));
		std::vector<int64_t> flagged(fq.begin(), fq.end());
		std::sort(flagged.begin(), flagged.end());
		std::vector<float> xv;
		if (!x_pageable) {
			xv.resize((size_t)nq * d);
			MVS_HIP(hipSetDevice(xdev));
			MVS_HIP(hipMemcpy(xv.data(), d_x, xv.size() * sizeof(float), hipMemcpyDeviceToHost));
			MVS_HIP(hipSetDevice(devs[0]));
			x_pageable = xv.data();
		}
		std::vector<float> Dt;
		std::vector<int64_t> It;
		float *Dp = D;
		int64_t *Ip = I;
		if (!D) { // device-pointer entry: patch the finished block on the host and hand it back
			Dt.resize(ocells);
			It.resize(ocells);
			MVS_HIP(hipMemcpy(Dt.data(), w0.oD.p, ocells * sizeof(float), hipMemcpyDeviceToHost));
			MVS_HIP(hipMemcpy(It.data(), w0.oI.p, ocells * sizeof(int64_t), hipMemcpyDeviceToHost));
			Dp = Dt.data();
			Ip = It.data();
		}
		if (ivf_ties)
			resolve_ties_ivf(flagged, x_pageable, k, kk, mv, mg, params, Dp, Ip);
		else
			resolve_ties(flagged, x_pageable, k, kk, mv, mg, params, Dp, Ip);
		if (!D) {
			MVS_HIP(hipMemcpy(w0.oD.p, Dt.data(), ocells * sizeof(float), hipMemcpyHostToDevice));
			MVS_HIP(hipMemcpy(w0.oI.p, It.data(), ocells * sizeof(int64_t), hipMemcpyHostToDevice));
			copy_between(d_D, w0.oD.p, devs[0], ocells * sizeof(float), s0);
			copy_between(d_I, w0.oI.p, devs[0], ocells * sizeof(int64_t), s0);
			MVS_HIP(hipStreamSynchronize(s0));
		}
		kinfo = shards[0]->kinfo;
	}
	// dst (device memory of whichever device owns it) <- src on device src_dev, on stream st of src_dev
	static void copy_between(void *dst, const void *src, int src_dev, size_t bytes, hipStream_t st) {
		hipPointerAttribute_t at;
		MVS_HIP(hipPointerGetAttributes(&at, dst));
		if (at.device == src_dev)
			MVS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
		else
			MVS_HIP(hipMemcpyPeerAsync(dst, at.device, src, src_dev, bytes, st));
	}
	int64_t to_label(int64_t g) const {
		if (g < 0)
			return -1;
		return has_idmap ? idmap_host[(size_t)g] : g + label_offset;
	}

	// raw search of shard g: queries to the device (H2D from the shared pinned copy, a peer copy, or in place), pure-order
	// top-kk with GLOBAL row numbers, packed into records; host exchange: the records go straight into the first device's
	// receive buffer (slot g) and ev[g] marks their arrival
	void shard_search(int g, int64_t nq, const float *hx, const float *d_x, int xdev, hipEvent_t x_ready, int64_t kk,
	                  const mvs_search_params *params, bool for_rccl) {
		IndexBase *s = shards[g];
		s->use_device();
		hipStream_t st = streams[g];
		Scratch &w = sc[g];
		const size_t xbytes = (size_t)nq * d * sizeof(float), cells = (size_t)nq * kk;
		w.D.reserve(cells * sizeof(float));
		w.I.reserve(cells * sizeof(int64_t));
		const float *xq = nullptr;
		if (hx) {
			w.x.reserve(xbytes);
			MVS_HIP(hipMemcpyAsync(w.x.p, hx, xbytes, hipMemcpyHostToDevice, st));
			xq = (const float *)w.x.p;
		} else {
			MVS_HIP(hipStreamWaitEvent(st, x_ready, 0));
			if (xdev == devs[g]) {
				xq = d_x;
			} else {
				w.x.reserve(xbytes);
				MVS_HIP(hipMemcpyPeerAsync(w.x.p, devs[g], d_x, xdev, xbytes, st));
				xq = (const float *)w.x.p;
			}
		}
		if (mode == ROWS_FLAT) {
			auto *f = static_cast<FlatIndex *>(s);
			const int64_t *selmap = has_idmap ? label_dev[g].p : gnum_dev[g].p; // the selector tests the label
			f->search_flat(nq, xq, kk, (float *)w.D.p, (int64_t *)w.I.p, params, selmap, st);
			if (f->ntotal > 0)
				hipLaunchKernelGGL(map_rows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st,
				                   (long long *)w.I.p, (long long)cells, (const long long *)gnum_dev[g].p);
		} else {
			s->search_mapped(nq, xq, kk, (float *)w.D.p, (int64_t *)w.I.p, params, has_idmap ? idmap_dev[g].p : nullptr, st);
		}
		// rccl: packed into the send buffer; otherwise slot g of the first device's receive buffer, written in place when the
		// shard lives there, else packed locally and pushed with one peer copy
		const bool in_place = !for_rccl && devs[g] == devs[0];
		Rec *out = in_place ? (Rec *)sc[0].recv.p + cells * g : nullptr;
		if (!out) {
			w.rec.reserve(cells * sizeof(Rec));
			out = (Rec *)w.rec.p;
		}
		hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const float *)w.D.p,
		                   (const long long *)w.I.p, (long long)cells, out);
		MVS_HIP(hipGetLastError());
		if (!for_rccl) {
			if (!in_place)
				MVS_HIP(hipMemcpyPeerAsync((Rec *)sc[0].recv.p + cells * g, devs[0], w.rec.p, devs[g], cells * sizeof(Rec), st));
			MVS_HIP(hipEventRecord(ev[g], st));
		}
	}

	// cross-shard tie pass (inner product): every shard reports, per flagged query, its k smallest global rows with
	// score >= T; the k smallest of the union are the first k arrivals FAISS's single heap would have seen
	void resolve_ties(const std::vector<int64_t> &flagged, const float *x, int64_t k, int64_t kk, const std::vector<float> &mv,
	                  const std::vector<int64_t> &mg, const mvs_search_params *params, float *D, int64_t *I) {
		const int64_t nf = (int64_t)flagged.size();
		std::vector<float> xf((size_t)nf * d), T((size_t)nf);
		for (int64_t f = 0; f < nf; ++f) {
			memcpy(&xf[(size_t)f * d], x + flagged[(size_t)f] * d, (size_t)d * sizeof(float));
			T[(size_t)f] = mv[(size_t)flagged[(size_t)f] * kk + k - 1];
		}
		on_all([&](int g) {
			auto *fl = static_cast<FlatIndex *>(shards[g]);
			fl->use_device();
			hipStream_t st = streams[g];
			Scratch &w = sc[g];
			w.x.reserve(xf.size() * sizeof(float));
			w.T.reserve((size_t)nf * sizeof(float));
			w.rows.reserve((size_t)nf * k * sizeof(int64_t));
			MVS_HIP(hipMemcpyAsync(w.x.p, xf.data(), xf.size() * sizeof(float), hipMemcpyHostToDevice, st));
			MVS_HIP(hipMemcpyAsync(w.T.p, T.data(), (size_t)nf * sizeof(float), hipMemcpyHostToDevice, st));
			SelectorDev sel = fl->upload_selector(params, st);
			fl->tie_candidates(nf, (const float *)w.x.p, (const float *)w.T.p, k, (int64_t *)w.rows.p, sel,
			                   has_idmap ? label_dev[g].p : gnum_dev[g].p, st);
			if (fl->ntotal > 0)
				hipLaunchKernelGGL(map_rows_kernel, dim3((unsigned)((nf * k + 255) / 256)), dim3(256), 0, st,
				                   (long long *)w.rows.p, (long long)(nf * k), (const long long *)gnum_dev[g].p);
			int64_t *hr = (int64_t *)w.hrows.get((size_t)nf * k * sizeof(int64_t));
			MVS_HIP(hipMemcpyAsync(hr, w.rows.p, (size_t)nf * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
			MVS_HIP(hipStreamSynchronize(st));
		});
		std::vector<int64_t> first((size_t)k), og((size_t)k);
		std::vector<float> ov((size_t)k);
		for (int64_t f = 0; f < nf; ++f) {
			std::vector<int64_t> all;
			for (int g = 0; g < G; ++g) {
				const int64_t *hr = (const int64_t *)sc[g].hrows.p + f * k;
				for (int64_t j = 0; j < k; ++j)
					if (hr[j] >= 0)
						all.push_back(hr[j]);
			}
			std::sort(all.begin(), all.end());
			for (int64_t j = 0; j < k; ++j)
				first[(size_t)j] = j < (int64_t)all.size() ? all[(size_t)j] : -1;
			const int64_t q = flagged[(size_t)f];
			resolve_ip_tie_host(k, &mv[(size_t)q * kk], &mg[(size_t)q * kk], first.data(), ov.data(), og.data());
			for (int64_t j = 0; j < k; ++j) {
				D[q * k + j] = ov[(size_t)j];
				I[q * k + j] = to_label(og[(size_t)j]);
			}
		}
	}

	// Cross-shard tie pass of a row-sharded IVF index.  FAISS's single heap would have been fed the probed lists in probe order,
	// each list front to back; a list's rows are spread over the shards, every shard holding a subsequence in insertion order with
	// GLOBAL row numbers as stored ids -- so the arrival key of a row is (probe rank of its list, global row).  Every shard reports,
	// per flagged query, its first k rows not worse than T in that order (IndexBase::tie_emit: the shard's coarse assignment of this
	// batch is still in place and is the same on every shard); the first k of the union are A_k, and the closed form of
	// csrc/ivf_ties.hip decides: result = {rows better than T} + {tied rows of A_k minus the G extreme ids}, G = #(rows better
	// than T outside A_k) -- the LARGEST ids go for L2 (CMax root), the smallest for inner product.
	bool ivf_exact_ties = true; // option ivf_exact_ties on the sharded handle: 0 = pure (value, id) order across shards
	void resolve_ties_ivf(const std::vector<int64_t> &flagged, const float *x, int64_t k, int64_t kk, const std::vector<float> &mv,
	                      const std::vector<int64_t> &mg, const mvs_search_params *params, float *D, int64_t *I) {
		const int64_t nf = (int64_t)flagged.size();
		const bool is_l2 = metric_order(metric) == METRIC_L2;
		std::vector<int> hflagq((size_t)nf + 1);
		std::vector<float> T((size_t)nf);
		hflagq[0] = (int)nf;
		for (int64_t f = 0; f < nf; ++f) {
			hflagq[(size_t)f + 1] = (int)flagged[(size_t)f];
			T[(size_t)f] = mv[(size_t)flagged[(size_t)f] * kk + k - 1];
		}
		struct Emit {
			std::vector<float> v;
			std::vector<int64_t> id;
			std::vector<int> p;
		};
		std::vector<Emit> em((size_t)G);
		on_all([&](int g) {
			IndexBase *sh = shards[g];
			sh->use_device();
			hipStream_t st = streams[g];
			Scratch &w = sc[g];
			const size_t cells = (size_t)nf * k;
			const size_t t_b = ((size_t)nf * sizeof(float) + 255) & ~(size_t)255, id_b = (cells * sizeof(int64_t) + 255) & ~(size_t)255,
			             v_b = (cells * sizeof(float) + 255) & ~(size_t)255;
			w.T.reserve(t_b + ((size_t)nf + 1) * sizeof(int) + 256);
			w.rows.reserve(id_b + v_b + cells * sizeof(int) + 256);
			float *dT = (float *)w.T.p;
			int *dflag = (int *)((char *)w.T.p + t_b);
			int64_t *did = (int64_t *)w.rows.p;
			float *dv = (float *)((char *)w.rows.p + id_b);
			int *dp = (int *)((char *)w.rows.p + id_b + v_b);
			MVS_HIP(hipMemcpyAsync(dT, T.data(), (size_t)nf * sizeof(float), hipMemcpyHostToDevice, st));
			MVS_HIP(hipMemcpyAsync(dflag, hflagq.data(), ((size_t)nf + 1) * sizeof(int), hipMemcpyHostToDevice, st));
			// the batch's queries up to the last flagged one (the kernel indexes them by query number)
			const int64_t nqx = flagged.back() + 1;
			DevBuf xtmp;
			xtmp.reserve((size_t)nqx * d * sizeof(float));
			MVS_HIP(hipMemcpyAsync(xtmp.p, x, (size_t)nqx * d * sizeof(float), hipMemcpyHostToDevice, st));
			const float *xq = (const float *)xtmp.p;
			sh->tie_emit(dflag, (int)nf, xq, dT, k, params, has_idmap ? idmap_dev[g].p : nullptr, dv, did, dp, st);
			em[g].v.resize(cells);
			em[g].id.resize(cells);
			em[g].p.resize(cells);
			MVS_HIP(hipMemcpyAsync(em[g].v.data(), dv, cells * sizeof(float), hipMemcpyDeviceToHost, st));
			MVS_HIP(hipMemcpyAsync(em[g].id.data(), did, cells * sizeof(int64_t), hipMemcpyDeviceToHost, st));
			MVS_HIP(hipMemcpyAsync(em[g].p.data(), dp, cells * sizeof(int), hipMemcpyDeviceToHost, st));
			MVS_HIP(hipStreamSynchronize(st));
		});
		struct Ent {
			int p;
			int64_t id;
			float v;
		};
		for (int64_t f = 0; f < nf; ++f) {
			const int64_t q = flagged[(size_t)f];
			const float Tq = T[(size_t)f];
			std::vector<Ent> all;
			for (int g = 0; g < G; ++g)
				for (int64_t j = 0; j < k; ++j) {
					const size_t c = (size_t)f * k + j;
					if (em[g].id[c] >= 0)
						all.push_back({em[g].p[c], em[g].id[c], em[g].v[c]});
				}
			std::sort(all.begin(), all.end(), [](const Ent &a, const Ent &b) { return a.p != b.p ? a.p < b.p : a.id < b.id; });
			if ((int64_t)all.size() > k)
				all.resize((size_t)k); // A_k
			// rows strictly better than T: all of them are in the merged pure list (fewer than k exist)
			std::vector<std::pair<float, int64_t>> res;
			for (int64_t j = 0; j < k; ++j) {
				const float v = mv[(size_t)q * kk + j];
				const int64_t gid = mg[(size_t)q * kk + j];
				if (gid >= 0 && (is_l2 ? v < Tq : v > Tq))
					res.push_back({v, gid});
			}
			const int64_t nbetter = (int64_t)res.size();
			int64_t in_ak = 0;
			std::vector<int64_t> tied;
			for (const Ent &e : all) {
				if (is_l2 ? e.v < Tq : e.v > Tq)
					++in_ak;
				else if (e.v == Tq)
					tied.push_back(e.id);
			}
			const int64_t Gev = nbetter - in_ak;
			std::sort(tied.begin(), tied.end());
			const int64_t nt = (int64_t)tied.size();
			// L2: the Gev largest ids are evicted, ascending id; inner product: the Gev smallest, printed in descending id
			if (is_l2) {
				for (int64_t t = 0; t < nt - Gev; ++t)
					res.push_back({Tq, tied[(size_t)t]});
			} else {
				for (int64_t t = nt - 1; t >= Gev; --t)
					res.push_back({Tq, tied[(size_t)t]});
				// (the better part came out of the pure order -- score desc, id asc: equal scores print in descending id)
				int64_t a = 0;
				while (a < nbetter) {
					int64_t b = a + 1;
					while (b < nbetter && res[(size_t)b].first == res[(size_t)a].first)
						++b;
					std::reverse(res.begin() + a, res.begin() + b);
					a = b;
				}
			}
			for (int64_t j = 0; j < k; ++j) {
				const bool have = j < (int64_t)res.size();
				D[q * k + j] = have ? res[(size_t)j].first : (is_l2 ? FLT_MAX : -FLT_MAX);
				I[q * k + j] = have ? to_label(res[(size_t)j].second) : -1;
			}
		}
	}

	bool ensure_comms() {
		if (!comms.empty())
			return true;
		Rccl &r = Rccl::get();
		if (!r.ok)
			throw_faiss("mvs::ShardedIndex::search", __FILE__, "shard_exchange = rccl: %s", r.why.c_str());
		for (int a = 0; a < G; ++a)
			for (int b = a + 1; b < G; ++b)
				if (devs[a] == devs[b])
					throw_faiss("mvs::ShardedIndex::search", __FILE__,
					            "shard_exchange = rccl needs one device per shard (device %d holds two); use the host exchange",
					            devs[a]);
		comms.assign((size_t)G, nullptr);
		check_nccl(r.comm_init_all(comms.data(), G, devs.data()), "ncclCommInitAll");
		return true;
	}
	static void check_nccl(int rc, const char *what) {
		if (rc != 0) {
			Rccl &r = Rccl::get();
			throw_faiss("mvs::ShardedIndex", __FILE__, "%s failed: %s", what, r.err_str ? r.err_str(rc) : "rccl error");
		}
	}

	void search_mapped(int64_t, const float *, int64_t, float *, int64_t *, const mvs_search_params *, const int64_t *,
	                   hipStream_t) override {
		throw_faiss("mvs::ShardedIndex::search_mapped", __FILE__, "a sharded index cannot sit under an IDMap wrapper");
	}
	void to_device(int) override {
		throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp",
		            "a sharded index is moved by cloning (index_cpu_to_gpu returns a new index)");
	}
	IndexBase *clone(int on_device) override {
		HostIndex h;
		to_host(h);
		return index_from_host(h, on_device);
	}

	// host image of the EQUIVALENT unsharded index (write_index, clone): rows back in global order
	void to_host(HostIndex &out) override {
		if (mode == REPLICAS) {
			shards[0]->to_host(out);
			return;
		}
		HostIndex body;
		if (mode == ROWS_FLAT) {
			body.kind = MVS_KIND_FLAT;
			body.d = d;
			body.metric = metric;
			body.ntotal = ntotal;
			body.rows.resize((size_t)ntotal * d);
			for (int g = 0; g < G; ++g) {
				auto *f = static_cast<FlatIndex *>(shards[g]);
				const int64_t n = f->ntotal;
				if (n == 0)
					continue;
				std::vector<float> rows((size_t)n * d);
				f->copy_rows_to_host(rows.data());
				std::vector<int64_t> gn((size_t)n);
				MVS_HIP(hipSetDevice(devs[g]));
				MVS_HIP(hipMemcpy(gn.data(), gnum_dev[g].p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
				for (int64_t r = 0; r < n; ++r)
					memcpy(&body.rows[(size_t)gn[(size_t)r] * d], &rows[(size_t)r * d], (size_t)d * sizeof(float));
			}
		} else {
			shards[0]->to_host(body); // header, quantizer, shard 0's part of every list
			for (int g = 1; g < G; ++g) {
				HostIndex part;
				shards[g]->to_host(part);
				for (size_t l = 0; l < body.list_ids.size(); ++l) {
					body.list_ids[l].insert(body.list_ids[l].end(), part.list_ids[l].begin(), part.list_ids[l].end());
					body.list_codes[l].insert(body.list_codes[l].end(), part.list_codes[l].begin(), part.list_codes[l].end());
				}
			}
			// inside a list: arrival order = ascending stored id when the ids are global row numbers
			for (size_t l = 0; l < body.list_ids.size(); ++l) {
				auto &ids = body.list_ids[l];
				auto &codes = body.list_codes[l];
				std::vector<size_t> ord(ids.size());
				for (size_t i = 0; i < ord.size(); ++i)
					ord[i] = i;
				std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return ids[a] < ids[b]; });
				std::vector<int64_t> ni(ids.size());
				std::vector<float> nc(codes.size());
				for (size_t i = 0; i < ord.size(); ++i) {
					ni[i] = ids[ord[i]];
					memcpy(&nc[i * d], &codes[ord[i] * d], (size_t)d * sizeof(float));
				}
				ids.swap(ni);
				codes.swap(nc);
			}
			body.ntotal = ntotal;
		}
		if (!has_idmap) {
			out = std::move(body);
			return;
		}
		out.kind = MVS_KIND_IDMAP;
		out.d = d;
		out.metric = metric;
		out.ntotal = ntotal;
		out.is_trained = is_trained;
		out.ids = idmap_host;
		out.sub.reset(new HostIndex(std::move(body)));
	}

	void set_label_offset(int64_t off) override {
		label_offset = off;
	}
	void adopt_tuning(const Tuning &t) override {
		tune_ = t;
		for (IndexBase *sh : shards)
			sh->adopt_tuning(t);
	}
	bool set_option(const char *key, int64_t v) override {
		if (!strcmp(key, "shard_exchange")) { // 0 = host gather, 1 = rccl all-gather
			exchange = (int)v;
			return true;
		}
		if (!strcmp(key, "ivf_exact_ties") && mode == ROWS_IVF) { // across shards (the shards themselves keep the pure order)
			ivf_exact_ties = v != 0;
			return true;
		}
		bool any = false;
		for (auto *s : shards)
			any |= s->set_option(key, v);
		return any;
	}
	void set_timing(bool on) override {
		for (auto *s : shards)
			s->set_timing(on);
	}
	void resolve_timing(int *count, double *total_ms) override {
		shards[0]->resolve_timing(count, total_ms);
		kinfo = shards[0]->kinfo;
	}

};

// ---- entry points used by the C ABI ---------------------------------------------------------------------------
std::vector<int> shard_devices_from_env() {
	std::vector<int> out;
	const char *e = getenv("MVS_DEVICES");
	if (!e || !*e)
		return out;
	const char *p = e;
	while (*p) {
		char *end = nullptr;
		const long v = strtol(p, &end, 10);
		if (end == p)
			break;
		out.push_back((int)v);
		p = *end == ',' ? end + 1 : end;
	}
	return out;
}
static void check_devices(const std::vector<int> &devs) {
	int ndev = 0;
	MVS_HIP(hipGetDeviceCount(&ndev));
	for (int v : devs)
		if (v < 0 || v >= ndev)
			throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "Invalid GPU device %d", v);
}
IndexBase *make_sharded_index(int d, const char *desc, int metric, const std::vector<int> &devices) {
	check_devices(devices);
	CtorDevice scope(devices[0]); // the wrapper's own (unused) stream lives with the first shard
	return new ShardedIndex(d, desc, metric, devices);
}
bool is_sharded(const IndexBase *ix) {
	return dynamic_cast<const ShardedIndex *>(ix) != nullptr;
}
IndexBase *sharded_inner_view(IndexBase *ix) {
	auto *s = dynamic_cast<ShardedIndex *>(ix);
	return s ? s->inner_view() : ix;
}
void sharded_for_each(IndexBase *ix, const std::function<void(IndexBase *)> &f) {
	auto *s = dynamic_cast<ShardedIndex *>(ix);
	if (!s) {
		f(ix);
		return;
	}
	for (auto *sh : s->shards)
		f(sh);
}
int sharded_info(const IndexBase *ix, int *devices, int max_devices, int64_t *rows_per_shard, int64_t *last_flagged) {
	auto *s = dynamic_cast<const ShardedIndex *>(ix);
	if (!s)
		return 0;
	for (int g = 0; g < s->G && g < max_devices; ++g) {
		if (devices)
			devices[g] = s->devs[g];
		if (rows_per_shard)
			rows_per_shard[g] = s->shards[g]->ntotal;
	}
	if (last_flagged)
		*last_flagged = s->last_flagged;
	return s->G;
}

static std::string factory_string_of(const HostIndex &h) {
	switch (h.kind) {
	case MVS_KIND_FLAT:
		return "Flat";
	case MVS_KIND_IDMAP:
		return "IDMap," + factory_string_of(*h.sub);
	case MVS_KIND_IVFFLAT:
		if (h.sub && h.sub->kind == MVS_KIND_HNSW && h.sub->cum_nneighbor_per_level.size() >= 2)
			return "IVF" + std::to_string(h.nlist) + "_HNSW" + std::to_string(h.sub->cum_nneighbor_per_level[1] / 2) + ",Flat";
		return "IVF" + std::to_string(h.nlist) + ",Flat";
	case MVS_KIND_HNSW:
		return "HNSW" + std::to_string(h.cum_nneighbor_per_level.size() >= 2 ? h.cum_nneighbor_per_level[1] / 2 : 32);
	}
	throw_faiss("mvs::shard_index", __FILE__, "unknown index kind %d", h.kind);
}

// An existing (single-device or host-image) index spread over `devices`: same rows, same labels, same answers.
IndexBase *shard_from_host(const HostIndex &h, const std::vector<int> &devices) {
	check_devices(devices);
	const HostIndex *body = h.kind == MVS_KIND_IDMAP ? h.sub.get() : &h;
	if (!body)
		throw_faiss("mvs::shard_index", __FILE__, "IDMap image without a sub-index");
	const bool idmap = h.kind == MVS_KIND_IDMAP;
	CtorDevice scope(devices[0]);
	if (body->kind == MVS_KIND_HNSW) { // replicas: the stored graph, copied to every device
		std::vector<IndexBase *> reps;
		try {
			for (int dv : devices)
				reps.push_back(index_from_host(h, dv));
			return new ShardedIndex(reps, devices, idmap);
		} catch (...) {
			for (auto *r : reps)
				delete r;
			throw;
		}
	}
	auto *s = new ShardedIndex(h.d, factory_string_of(h), h.metric, devices);
	try {
		if (body->kind == MVS_KIND_FLAT) {
			if (h.ntotal > 0) {
				if (idmap)
					s->add_with_ids(h.ntotal, body->rows.data(), h.ids.data());
				else
					s->add(h.ntotal, body->rows.data());
			}
		} else if (body->kind == MVS_KIND_IVFFLAT) {
			if (body->is_trained && body->sub && body->sub->kind == MVS_KIND_FLAT && body->sub->ntotal == body->nlist) {
				for (auto *sh : s->shards)
					ivf_set_centroids(sh, body->sub->rows.data());
				s->is_trained = true;
			} else if (body->is_trained) {
				throw_faiss("mvs::shard_index", __FILE__, "sharding a trained IVF index needs a Flat coarse quantizer image");
			}
			// rows back in arrival order (stored ids of an image are sequential under an IDMap, arbitrary otherwise)
			std::vector<std::pair<int64_t, const float *>> rows;
			for (size_t l = 0; l < body->list_ids.size(); ++l)
				for (size_t j = 0; j < body->list_ids[l].size(); ++j)
					rows.emplace_back(body->list_ids[l][j], &body->list_codes[l][j * (size_t)h.d]);
			std::stable_sort(rows.begin(), rows.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
			std::vector<float> x(rows.size() * (size_t)h.d);
			std::vector<int64_t> ids(rows.size());
			for (size_t i = 0; i < rows.size(); ++i) {
				memcpy(&x[i * (size_t)h.d], rows[i].second, (size_t)h.d * sizeof(float));
				ids[i] = idmap ? h.ids[(size_t)rows[i].first] : rows[i].first;
			}
			if (!rows.empty())
				s->add_with_ids((int64_t)rows.size(), x.data(), ids.data());
		} else {
			throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "This index type is not implemented");
		}
	} catch (...) {
		delete s;
		throw;
	}
	return s;
}

} // namespace mvs
