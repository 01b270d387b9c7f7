// host/boundary_driver.cpp -- stand-alone driver that talks to the MI355X path EXACTLY the way the reference's glue
// does, through the faiss:: adaptor (compat/faiss/*.h): DuckDB itself is not available in this image, so this file
// reproduces the call patterns of /root/reference/src/faiss_extension.cpp around the FAISS boundary --
//   CreateFunction        :146-164   index_factory, needs_training = !is_trained
//   AddFunction           :475-547   <= 2048-row DataChunks from several worker threads, faiss_lock, error text
//   AddFinaliseFunction   :549-615   last thread out trains on ALL rows, then adds the rows not yet added
//   searchIntoVector      :621-666   new[] result arrays, faiss_lock, search, scatter into (rank,label,distance)
//   createSearchParameters:668-727   IDMap recursion, SearchParametersIVF/HNSW, selector
//   MoveToGPUFunction     gpu.cpp:34-63
// and prints the result rows so that tests/test_boundary_driver_gpu.py can compare them with the reference's
// golden vectors (test/sql/faiss.test, faiss3.test, faiss4.test, faiss7.test).
//
//   boundary_driver golden <training.csv> <queries.csv>
//   boundary_driver ingest <n> <d> <threads> [index]  (concurrent DataChunk ingest + self-query check; index = factory
//                                                      string, default "IDMap,Flat"; IVF trains in AddFinalise on all rows)
//   boundary_driver linkrate                          (host -> device copy rate of this box: pinned and pageable)
#include "faiss/Index.h"
#include "faiss/IndexHNSW.h"
#include "faiss/IndexIDMap.h"
#include "faiss/IndexIVF.h"
#include "faiss/gpu/GpuCloner.h"
#include "faiss/gpu/StandardGpuResources.h"
#include "faiss/index_factory.h"
#include "faiss/index_io.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t STANDARD_VECTOR_SIZE = 2048; // DuckDB DataChunk capacity

// FaissIndexEntry (src/include/index.hpp:12-56), reduced to what the path touches
struct IndexEntry {
	std::unique_ptr<std::mutex> faiss_lock {new std::mutex()};
	std::unique_ptr<faiss::Index> index;
	bool needs_training = true;
	bool custom_labels = false;
	std::atomic<uint64_t> currently_adding {0};
	std::unique_ptr<std::mutex> add_lock {new std::mutex()};
	std::vector<float> add_data;
	std::vector<faiss::idx_t> add_labels;
	size_t size = 0, added = 0;
};

struct InvalidInput : std::runtime_error {
	using std::runtime_error::runtime_error;
};

// CreateFunction :146-164 (default metric INNER_PRODUCT :105)
std::unique_ptr<IndexEntry> create(int d, const std::string &desc, faiss::MetricType metric = faiss::METRIC_INNER_PRODUCT) {
	auto e = std::make_unique<IndexEntry>();
	e->index.reset(faiss::index_factory(d, desc.c_str(), metric));
	e->needs_training = !e->index->is_trained;
	return e;
}

// AddFunction :475-547 for one DataChunk
// MVS_INGEST_PROFILE=1: the duration of every add call under the lock (where does an ingest's time go: the steady calls or the few
// that grow the device buffers)
static std::vector<double> g_add_us;
static const bool g_add_profile = getenv("MVS_INGEST_PROFILE") != nullptr;
void add_chunk(IndexEntry &entry, size_t n, const float *x, const faiss::idx_t *ids) {
	if (!entry.needs_training) {
		entry.faiss_lock->lock();
		try {
			const auto t0 = std::chrono::steady_clock::now();
			if (entry.custom_labels)
				entry.index->add_with_ids((faiss::idx_t)n, x, ids);
			else
				entry.index->add((faiss::idx_t)n, x);
			if (g_add_profile)
				g_add_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
		} catch (faiss::FaissException exception) {
			entry.faiss_lock->unlock();
			std::string msg = exception.msg;
			if (msg.find("add_with_ids not implemented for this type of index") != std::string::npos)
				throw InvalidInput("Unable to add data: This type of index does not support adding with IDs. "
				                   "Consider prefixing the index string with IDMap when creating the index.");
			throw InvalidInput(std::string("Unable to add data: ") + exception.what());
		}
		entry.faiss_lock->unlock();
		return;
	}
	std::lock_guard<std::mutex> g(*entry.add_lock);
	entry.add_data.insert(entry.add_data.end(), x, x + n * entry.index->d);
	if (entry.custom_labels)
		entry.add_labels.insert(entry.add_labels.end(), ids, ids + n);
	entry.size += n;
}

// AddFinaliseFunction :549-615
void add_finalise(IndexEntry &entry) {
	size_t total, added;
	{
		std::lock_guard<std::mutex> g(*entry.add_lock);
		entry.currently_adding--;
		if (entry.currently_adding != 0)
			return;
		total = entry.size;
		added = entry.added;
		if (added == total)
			return;
		entry.added = total;
	}
	if (entry.add_data.empty())
		return;
	std::lock_guard<std::mutex> g(*entry.faiss_lock);
	try {
		entry.index->train((faiss::idx_t)total, entry.add_data.data());
	} catch (faiss::FaissException exception) {
		std::string msg = exception.msg;
		if (msg.find("should be at least as large as number of clusters") != std::string::npos)
			throw InvalidInput("Index needs to be trained, but amount of datapoints is too small. Considere adding more "
			                   "data. (" + msg + ")");
		throw InvalidInput("Error occured while training index: " + msg);
	}
	const faiss::idx_t nnew = (faiss::idx_t)(total - added);
	const float *xnew = entry.add_data.data() + added * entry.index->d;
	if (entry.custom_labels)
		entry.index->add_with_ids(nnew, xnew, entry.add_labels.data() + added);
	else
		entry.index->add(nnew, xnew);
	entry.needs_training = !entry.index->is_trained;
}

// faiss_add((SELECT [id,] vec ...), name): chunks of <= 2048 rows pulled by `nthreads` workers
void faiss_add(IndexEntry &entry, size_t n, const float *x, const faiss::idx_t *ids, int nthreads) {
	entry.custom_labels = ids != nullptr;
	const size_t d = (size_t)entry.index->d;
	const size_t nchunks = (n + STANDARD_VECTOR_SIZE - 1) / STANDARD_VECTOR_SIZE;
	std::atomic<size_t> next {0};
	std::vector<std::thread> workers;
	std::mutex err_lock;
	std::string err;
	entry.currently_adding += nthreads; // AddLocalInit per worker :462-473
	for (int t = 0; t < nthreads; ++t)
		workers.emplace_back([&] {
			try {
				std::vector<float> chunk;
				std::vector<faiss::idx_t> idc;
				for (;;) {
					size_t c = next++;
					if (c >= nchunks)
						break;
					size_t r0 = c * STANDARD_VECTOR_SIZE, nr = std::min(STANDARD_VECTOR_SIZE, n - r0);
					// the chunk lives in a buffer that is only valid during the call -- the worker's DataChunk, whose vector buffers
					// DuckDB REUSES from chunk to chunk (a fresh 1 MB allocation per chunk costs a page-fault storm that an
					// operator pipeline does not have: 8 threads of them slowed every add call from 39 to 60 us)
					chunk.assign(x + r0 * d, x + (r0 + nr) * d);
					if (ids)
						idc.assign(ids + r0, ids + r0 + nr);
					add_chunk(entry, nr, chunk.data(), ids ? idc.data() : nullptr);
				}
				add_finalise(entry);
			} catch (const std::exception &e) {
				std::lock_guard<std::mutex> g(err_lock);
				err = e.what();
			}
		});
	for (auto &w : workers)
		w.join();
	if (!err.empty())
		throw InvalidInput(err);
}

// innerCreateSearchParameters :668-721
std::vector<std::shared_ptr<faiss::SearchParameters>> create_search_parameters(faiss::Index *index, faiss::IDSelector *sel,
                                                                               int nprobe, int efSearch) {
	if (auto idmap = dynamic_cast<faiss::IndexIDMap *>(index))
		return create_search_parameters(idmap->index, sel, nprobe, efSearch);
	if (auto ivf = dynamic_cast<faiss::IndexIVF *>(index)) {
		auto p = std::make_shared<faiss::SearchParametersIVF>();
		p->sel = sel;
		auto ret = create_search_parameters(ivf->quantizer, nullptr, 0, 0);
		p->quantizer_params = ret[0].get();
		if (nprobe > 0)
			p->nprobe = (size_t)nprobe;
		ret.insert(ret.begin(), p);
		return ret;
	}
	if (dynamic_cast<faiss::IndexHNSW *>(index)) {
		auto p = std::make_shared<faiss::SearchParametersHNSW>();
		p->sel = sel;
		if (efSearch > 0)
			p->efSearch = efSearch;
		return {p};
	}
	if (dynamic_cast<faiss::IndexPQ *>(index))
		return {std::make_shared<faiss::SearchParametersPQ>()};
	auto p = std::make_shared<faiss::SearchParameters>();
	p->sel = sel;
	return {p};
}

struct ResultRow {
	int rank;
	int64_t label;
	float distance;
};
// searchIntoVector :621-666
std::vector<ResultRow> search_into_vector(IndexEntry &entry, size_t nq, const float *x, size_t k,
                                          faiss::SearchParameters *params) {
	std::unique_ptr<faiss::idx_t[]> labels(new faiss::idx_t[nq * k]);
	std::unique_ptr<float[]> distances(new float[nq * k]);
	entry.faiss_lock->lock();
	try {
		entry.index->search((faiss::idx_t)nq, x, (faiss::idx_t)k, distances.get(), labels.get(), params);
	} catch (faiss::FaissException exception) {
		entry.faiss_lock->unlock();
		throw InvalidInput("Error occured while searching: " + exception.msg);
	}
	entry.faiss_lock->unlock();
	std::vector<ResultRow> out(nq * k);
	for (size_t r = 0; r < nq; ++r)
		for (size_t j = 0; j < k; ++j)
			out[r * k + j] = {(int)j, labels[r * k + j], distances[r * k + j]};
	return out;
}
// faiss_search over a column of queries: one call per DataChunk (<= 2048 rows)
std::vector<ResultRow> faiss_search(IndexEntry &entry, size_t nq, const float *x, size_t k, faiss::IDSelector *sel = nullptr,
                                    int nprobe = 0, int efSearch = 0) {
	std::vector<ResultRow> all;
	const size_t d = (size_t)entry.index->d;
	for (size_t q0 = 0; q0 < nq; q0 += STANDARD_VECTOR_SIZE) {
		size_t nn = std::min(STANDARD_VECTOR_SIZE, nq - q0);
		auto params = create_search_parameters(entry.index.get(), sel, nprobe, efSearch);
		auto rows = search_into_vector(entry, nn, x + q0 * d, k, params[0].get());
		all.insert(all.end(), rows.begin(), rows.end());
	}
	return all;
}

bool read_csv(const char *path, std::vector<faiss::idx_t> &ids, std::vector<float> &vecs, int &d) {
	std::ifstream f(path);
	if (!f)
		return false;
	std::string line;
	d = 0;
	while (std::getline(f, line)) {
		if (line.empty())
			continue;
		std::stringstream ss(line);
		std::string cell;
		int col = 0;
		while (std::getline(ss, cell, ',')) {
			if (col == 0)
				ids.push_back((faiss::idx_t)std::stoll(cell));
			else
				vecs.push_back((float)std::stod(cell)); // LIST<DOUBLE> -> FLOAT cast, :292-293
			++col;
		}
		d = col - 1;
	}
	return true;
}

void print_rows(const char *tag, const std::vector<ResultRow> &rows) {
	for (auto &r : rows)
		printf("%s\t%d\t%lld\t%.9g\n", tag, r.rank, (long long)r.label, r.distance);
}

int run_golden(const char *train_csv, const char *query_csv) {
	std::vector<faiss::idx_t> ids, qids;
	std::vector<float> xb, xq;
	int d = 0, dq = 0;
	if (!read_csv(train_csv, ids, xb, d) || !read_csv(query_csv, qids, xq, dq) || d != dq) {
		fprintf(stderr, "cannot read fixtures\n");
		return 2;
	}
	const size_t n = ids.size(), nq = qids.size();
	{ // test/sql/faiss.test: Flat, no ids
		auto e = create(d, "Flat");
		faiss_add(*e, n, xb.data(), nullptr, 1);
		print_rows("flat", faiss_search(*e, nq, xq.data(), 2));
	}
	{ // test/sql/faiss3.test: IDMap,Flat + bitmap filter 'column0>100'
		auto e = create(d, "IDMap,Flat");
		faiss_add(*e, n, xb.data(), ids.data(), 3);
		print_rows("idmap", faiss_search(*e, nq, xq.data(), 2));
		faiss::idx_t maxid = 0;
		for (auto v : ids)
			maxid = std::max(maxid, v);
		std::vector<uint8_t> mask((size_t)maxid / 8 + 1, 0); // :765-767
		for (auto v : ids)
			if (v > 100)
				mask[(size_t)v >> 3] |= (uint8_t)(1u << (v & 7));
		faiss::IDSelectorBitmap sel(mask.size(), mask.data()); // :959
		print_rows("filter", faiss_search(*e, nq, xq.data(), 2, &sel));
		std::vector<faiss::idx_t> keep;
		for (auto v : ids)
			if (v > 100)
				keep.push_back(v);
		faiss::IDSelectorBatch selb(keep.size(), keep.data()); // :1008
		print_rows("filterset", faiss_search(*e, nq, xq.data(), 2, &selb));
	}
	{ // test/sql/faiss4.test / faiss6.test: ids on a plain Flat index
		auto e = create(d, "Flat", faiss::METRIC_L2);
		try {
			faiss_add(*e, n, xb.data(), ids.data(), 1);
			printf("error\tnone\n");
		} catch (const InvalidInput &ex) {
			printf("error\t%s\n", ex.what());
		}
		faiss_add(*e, n, xb.data(), nullptr, 2);
		printf("ntotal\t%lld\n", (long long)e->index->ntotal);
	}
	{ // test/sql/faiss7.test: N = 1 < k = 2 with a filter that excludes the only row
		auto e = create(2, "IDMap,Flat");
		float v[2] = {0.0040321066f, 0.023423655f};
		faiss::idx_t id = 231;
		faiss_add(*e, 1, v, &id, 1);
		float q[2] = {-0.04529257f, 0.024853613f};
		uint8_t mask[29] = {0};
		faiss::IDSelectorBitmap sel(sizeof mask, mask);
		print_rows("small", faiss_search(*e, 1, q, 2));
		print_rows("smallfilter", faiss_search(*e, 1, q, 2, &sel));
	}
	{ // faiss_to_gpu(name, device): src/gpu/gpu.cpp:34-63
		auto e = create(d, "IDMap,Flat");
		faiss_add(*e, n, xb.data(), ids.data(), 2);
		faiss::gpu::StandardGpuResources res;
		e->index.reset(faiss::gpu::index_cpu_to_gpu(&res, 0, e->index.get()));
		print_rows("togpu", faiss_search(*e, nq, xq.data(), 2));
		try {
			e->index.reset(faiss::gpu::index_cpu_to_gpu(&res, 4096, e->index.get()));
			printf("gpuerror\tnone\n");
		} catch (faiss::FaissException exception) {
			printf("gpuerror\t%s\n", exception.msg.find("Invalid GPU device") != std::string::npos ? "Invalid GPU index"
			                                                                                         : exception.msg.c_str());
		}
	}
	return 0;
}

// concurrent ingest the way DuckDB drives faiss_add for a large table, then self-queries
int run_ingest(size_t n, int d, int threads, const char *desc = "IDMap,Flat") {
	std::vector<float> xb(n * (size_t)d);
	uint64_t s = 88172645463325252ull;
	for (auto &v : xb) {
		s ^= s << 13;
		s ^= s >> 7;
		s ^= s << 17;
		v = (float)(s >> 40) * (1.0f / 16777216.0f);
	}
	std::vector<faiss::idx_t> ids(n);
	for (size_t i = 0; i < n; ++i)
		ids[i] = (faiss::idx_t)(1000000 + 7 * i);
	const bool with_ids = !strncmp(desc, "IDMap", 5);
	auto e = create(d, desc, faiss::METRIC_L2);
	const auto t0 = std::chrono::steady_clock::now();
	faiss_add(*e, n, xb.data(), with_ids ? ids.data() : nullptr, threads);
	// add() returns while the last H2D copies are still in flight: a 1-query search drains the index's stream
	(void)faiss_search(*e, 1, xb.data(), 1);
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (g_add_profile && !g_add_us.empty()) {
		std::vector<double> v = g_add_us;
		std::sort(v.begin(), v.end());
		double tot = 0, slow = 0;
		size_t nslow = 0;
		for (double u : v) {
			tot += u;
			if (u > 500.0)
				slow += u, ++nslow;
		}
		printf("ingestprofile\t%zu add calls: %.1f ms under the lock of %.1f ms; median %.1f us, p90 %.1f us; %zu calls over 500 us = %.1f ms\n",
		       v.size(), tot / 1e3, sec * 1e3, v[v.size() / 2], v[v.size() * 9 / 10], nslow, slow / 1e3);
	}
	printf("ingestrate\t%.0f rows/s (%s: %zu rows x %d dims in %zu add calls of <= 2048 rows from %d threads: %.3f s, %.2f GB/s "
	       "of row data)\n",
	       (double)n / sec, desc, n, d, (n + 2047) / 2048, threads, sec, (double)n * d * 4 / sec / 1e9);
	printf("ingestjson\t{\"index\": \"%s\", \"rows\": %zu, \"d\": %d, \"threads\": %d, \"seconds\": %.4f, \"rows_per_s\": %.0f, "
	       "\"GBps\": %.3f}\n",
	       desc, n, d, threads, sec, (double)n / sec, (double)n * d * 4 / sec / 1e9);
	if (!with_ids || strstr(desc, "HNSW") || strstr(desc, "IVF")) { // (approximate indexes: no self-query guarantee; the count is the check)
		const bool okc = (size_t)e->index->ntotal == n;
		printf("ingest\t%s ntotal=%lld, threads=%d\n", okc ? "OK" : "FAIL", (long long)e->index->ntotal, threads);
		return okc ? 0 : 1;
	}
	if ((size_t)e->index->ntotal != n) {
		printf("ingest\tFAIL ntotal %lld\n", (long long)e->index->ntotal);
		return 1;
	}
	// chunks arrive in nondeterministic order; every vector must still find itself under its own label
	const size_t nq = std::min<size_t>(n, 4096);
	auto rows = faiss_search(*e, nq, xb.data(), 1);
	size_t ok = 0;
	for (size_t i = 0; i < nq; ++i)
		ok += rows[i].label == ids[i] && rows[i].distance <= 1e-5f;
	printf("ingest\t%s %zu/%zu self-queries, ntotal=%lld, threads=%d\n", ok == nq ? "OK" : "FAIL", ok, nq,
	       (long long)e->index->ntotal, threads);
	return ok == nq ? 0 : 1;
}

// host -> device copy rate of this box (what the ingest rates are a fraction of): 256 MiB from pinned and from pageable memory
extern "C" {
int hipHostMalloc(void **, size_t, unsigned);
int hipHostFree(void *);
int hipMalloc(void **, size_t);
int hipFree(void *);
int hipMemcpy(void *, const void *, size_t, int);
int hipDeviceSynchronize();
}
int run_linkrate() {
	const size_t bytes = (size_t)256 << 20;
	void *dev = nullptr, *pin = nullptr;
	if (hipMalloc(&dev, bytes) != 0 || hipHostMalloc(&pin, bytes, 0) != 0) {
		printf("linkrate\tFAIL allocation\n");
		return 1;
	}
	std::vector<char> page(bytes, 1);
	memset(pin, 1, bytes);
	double best[2] = {0, 0};
	for (int rep = 0; rep < 4; ++rep)
		for (int kind = 0; kind < 2; ++kind) {
			hipDeviceSynchronize();
			const auto t0 = std::chrono::steady_clock::now();
			hipMemcpy(dev, kind == 0 ? pin : (void *)page.data(), bytes, 1 /* hipMemcpyHostToDevice */);
			hipDeviceSynchronize();
			const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			best[kind] = std::max(best[kind], (double)bytes / sec / 1e9);
		}
	printf("linkjson\t{\"h2d_pinned_GBps\": %.2f, \"h2d_pageable_GBps\": %.2f, \"bytes\": %zu}\n", best[0], best[1], bytes);
	hipFree(dev);
	hipHostFree(pin);
	return 0;
}

// README.md:61 index "IDMap,HNSW32": CreateFunction with the efConstruction parameter (:127-139), chunked faiss_add with
// ids, faiss_search with SearchParametersHNSW (:691-702), then faiss_save / faiss_load (:188-240) and the same search
int run_hnsw(size_t n, int d, int threads, const char *path) {
	std::vector<float> xb(n * (size_t)d), xq(256 * (size_t)d);
	uint64_t s = 0x2545F4914F6CDD1Dull;
	auto next = [&]() {
		s ^= s << 13;
		s ^= s >> 7;
		s ^= s << 17;
		return (float)(s >> 40) * (1.0f / 16777216.0f);
	};
	for (auto &v : xb)
		v = next();
	for (auto &v : xq)
		v = next();
	std::vector<faiss::idx_t> ids(n);
	for (size_t i = 0; i < n; ++i)
		ids[i] = (faiss::idx_t)(500 + 3 * i);
	auto e = create(d, "IDMap,HNSW32", faiss::METRIC_L2);
	{ // the glue's parameter application: unwrap IDMap, dynamic_cast to IndexHNSW, set hnsw.efConstruction
		faiss::Index *ix = e->index.get();
		if (auto idmap = dynamic_cast<faiss::IndexIDMap *>(ix))
			ix = idmap->index;
		auto hnsw = dynamic_cast<faiss::IndexHNSW *>(ix);
		if (!hnsw) {
			printf("hnsw\tFAIL dynamic_cast<IndexHNSW*>\n");
			return 1;
		}
		hnsw->hnsw.efConstruction = 64;
	}
	faiss_add(*e, n, xb.data(), ids.data(), threads);
	// the graph must have been built with the value the glue assigned on the IDMap's sub-index wrapper
	const int efc = mvs_index_hnsw_get_ef_construction(e->index->handle);
	printf("hnswefc\t%s device efConstruction=%d (set 64 through IndexIDMap::index)\n", efc == 64 ? "OK" : "FAIL", efc);
	if (efc != 64)
		return 1;
	auto exact = create(d, "IDMap,Flat", faiss::METRIC_L2);
	faiss_add(*exact, n, xb.data(), ids.data(), 1);
	const size_t nq = 256, k = 10;
	auto got = faiss_search(*e, nq, xq.data(), k, nullptr, 0, 128);
	auto want = faiss_search(*exact, nq, xq.data(), k);
	size_t hits = 0;
	for (size_t q = 0; q < nq; ++q)
		for (size_t a = 0; a < k; ++a)
			for (size_t b = 0; b < k; ++b)
				hits += got[q * k + a].label == want[q * k + b].label;
	const double recall = (double)hits / (double)(nq * k);
	printf("hnsw\t%s recall@10 %.4f ntotal=%lld\n", recall >= 0.9 ? "OK" : "FAIL", recall, (long long)e->index->ntotal);
	// SaveFunction :199 / LoadFunction :234
	faiss::write_index(e->index.get(), path);
	IndexEntry loaded;
	loaded.index.reset(faiss::read_index(path));
	auto again = faiss_search(loaded, nq, xq.data(), k, nullptr, 0, 128);
	size_t same = 0;
	for (size_t i = 0; i < got.size(); ++i)
		same += got[i].label == again[i].label && got[i].distance == again[i].distance;
	printf("hnswio\t%s %zu/%zu identical rows after write_index/read_index, d=%d ntotal=%lld trained=%d\n",
	       same == got.size() ? "OK" : "FAIL", same, got.size(), loaded.index->d, (long long)loaded.index->ntotal,
	       (int)loaded.index->is_trained);
	return recall >= 0.9 && same == got.size() ? 0 : 1;
}

} // namespace

int main(int argc, char **argv) {
	try {
		if (argc >= 4 && !strcmp(argv[1], "golden"))
			return run_golden(argv[2], argv[3]);
		if (argc >= 5 && !strcmp(argv[1], "ingest"))
			return run_ingest((size_t)atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), argc >= 6 ? argv[5] : "IDMap,Flat");
		if (argc >= 2 && !strcmp(argv[1], "linkrate"))
			return run_linkrate();
		if (argc >= 6 && !strcmp(argv[1], "hnsw"))
			return run_hnsw((size_t)atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), argv[5]);
	} catch (const std::exception &e) {
		fprintf(stderr, "fatal: %s\n", e.what());
		return 3;
	}
	fprintf(stderr, "usage: boundary_driver golden <training.csv> <queries.csv> | ingest <n> <d> <threads> [index] | linkrate | "
	                "hnsw <n> <d> <threads> <index file>\n");
	return 2;
}
