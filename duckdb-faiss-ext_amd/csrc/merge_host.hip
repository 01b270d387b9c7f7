// csrc/merge_host.hip -- host k-way merge of per-shard (distance,label) blocks.
// Follows the RCCL all-gather / per-device gather in the row-sharded multi-GPU search: every shard returns its best
// candidates per query with GLOBAL ids; the merged list is the best of the union under the pure order
//   L2: (distance asc, id asc)                    == what one CMax heap over all rows keeps
//   IP: (score desc, id asc); equal scores are PRINTED in descending id order (heap_reorder over a CMin heap)
// [faiss/utils/Heap.h heap_reorder; faiss::IndexShards merges with the same heap rule].
// The inputs are never modified (they may be read-only / mmapped RCCL receive buffers).
#include "index.h"

#include <algorithm>
#include <thread>

namespace mvs {

namespace {

struct Cand {
	float v;
	int64_t id;    // ordering id (global row number)
	int64_t label; // what the caller sees
};

inline bool cand_before(bool is_l2, const Cand &a, const Cand &b) {
	if (a.v != b.v)
		return is_l2 ? a.v < b.v : a.v > b.v;
	return a.id < b.id;
}

// print order of an inner-product result: runs of equal scores in descending id
inline void print_order_ip(Cand *c, int64_t m) {
	int64_t a = 0;
	while (a < m) {
		int64_t b = a + 1;
		while (b < m && c[b].v == c[a].v)
			++b;
		std::reverse(c + a, c + b);
		a = b;
	}
}

template <typename F>
void parallel_queries(int64_t n, F &&body) {
	int nt = (int)std::min<int64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / 512 + 1);
	if (nt <= 1) {
		body((int64_t)0, n);
		return;
	}
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t)
		th.emplace_back([&, t] { body(n * t / nt, n * (t + 1) / nt); });
	for (auto &x : th)
		x.join();
}

} // namespace

void merge_raw_lists_host(int metric, int64_t nq, int64_t kk, int nshard, const float *const *D, const int64_t *const *G,
                          float *val, int64_t *gnum);
void resolve_ip_tie_host(int64_t k, const float *raw_v, const int64_t *raw_g, const int64_t *first, float *out_v,
                         int64_t *out_g);

// queries are independent: the merge runs on up to 16 host threads (it sits on the critical path of every multi-GPU
// search step, after the exchange).  Per query: collect the nshard*k candidates, order them, keep k.
void merge_shards_host(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out,
                       int64_t *I_out) {
	const bool is_l2 = metric_order(metric) == METRIC_L2;
	const float neutral = is_l2 ? FLT_MAX : -FLT_MAX;
	parallel_queries(n, [&](int64_t q0, int64_t q1) {
		std::vector<Cand> c((size_t)nshard * k);
		for (int64_t q = q0; q < q1; ++q) {
			int64_t m = 0;
			for (int s = 0; s < nshard; ++s) {
				const float *ds = D + ((size_t)s * n + q) * k;
				const int64_t *is = I + ((size_t)s * n + q) * k;
				for (int64_t j = 0; j < k; ++j)
					if (is[j] >= 0)
						c[(size_t)m++] = {ds[j], is[j], is[j]};
			}
			const int64_t keep = std::min(m, k);
			std::partial_sort(c.begin(), c.begin() + keep, c.begin() + m,
			                  [&](const Cand &a, const Cand &b) { return cand_before(is_l2, a, b); });
			if (!is_l2)
				print_order_ip(c.data(), keep);
			for (int64_t j = 0; j < k; ++j) {
				D_out[q * k + j] = j < keep ? c[(size_t)j].v : neutral;
				I_out[q * k + j] = j < keep ? c[(size_t)j].label : -1;
			}
		}
	});
}

// Host twin of merge_records_kernel (csrc/util_kernels.hip) for the shapes whose candidates do not fit a workgroup's LDS
// (nshard * kk beyond ~6 000 entries: k in the thousands on 8 shards -- the Go harness asks for up to ~2 000 rows per query,
// /root/reference/go/main_test.go:26-32).  rec: [nshard][nq][kk][2] int64 records {value bits (low 32), global label} on the HOST.
void merge_records_host(int metric, const int64_t *rec, int nshard, int64_t nq, int kk, int kout, bool raw, float *D_out,
                        int64_t *I_out) {
	const bool is_l2 = metric_order(metric) == METRIC_L2;
	const float neutral = is_l2 ? FLT_MAX : -FLT_MAX;
	parallel_queries(nq, [&](int64_t q0, int64_t q1) {
		std::vector<Cand> c((size_t)nshard * kk);
		for (int64_t q = q0; q < q1; ++q) {
			int64_t m = 0;
			for (int s = 0; s < nshard; ++s) {
				const int64_t *r = rec + (((size_t)s * nq + q) * kk) * 2;
				for (int j = 0; j < kk; ++j)
					if (r[2 * j + 1] >= 0) {
						const int32_t bits = (int32_t)r[2 * j];
						float v;
						__builtin_memcpy(&v, &bits, 4);
						c[(size_t)m++] = {v, r[2 * j + 1], r[2 * j + 1]};
					}
			}
			const int64_t keep = std::min<int64_t>(m, kout);
			std::partial_sort(c.begin(), c.begin() + keep, c.begin() + m,
			                  [&](const Cand &a, const Cand &b) { return cand_before(is_l2, a, b); });
			if (!is_l2 && !raw)
				print_order_ip(c.data(), keep);
			for (int64_t j = 0; j < kout; ++j) {
				D_out[q * kout + j] = j < keep ? c[(size_t)j].v : neutral;
				I_out[q * kout + j] = j < keep ? c[(size_t)j].label : -1;
			}
		}
	});
}

// The same merge for a multi-PROCESS host (pyhost/sharded.py under torchrun): shard blocks [nshard][n][kk] as gathered,
// output the first kk of the union in the PURE order (no print reversal), -1 / neutral padded.
void merge_shards_raw_host(int metric, int64_t n, int64_t kk, int nshard, const float *D, const int64_t *I, float *D_out,
                           int64_t *I_out) {
	std::vector<const float *> dp((size_t)nshard);
	std::vector<const int64_t *> ip((size_t)nshard);
	for (int s = 0; s < nshard; ++s) {
		dp[(size_t)s] = D + (size_t)s * n * kk;
		ip[(size_t)s] = I + (size_t)s * n * kk;
	}
	merge_raw_lists_host(metric, n, kk, nshard, dp.data(), ip.data(), D_out, I_out);
}

// FAISS print order of the first k of every raw list + the closed-form outcome for the flagged queries.
// first: [nf][k] the k smallest global rows with score >= T of flagged query f (ascending, -1 padded).
void finish_ip_ties_host(int64_t n, int64_t k, int64_t kk, const float *raw_v, const int64_t *raw_g, int64_t nf,
                         const int64_t *fq, const int64_t *first, float *D_out, int64_t *I_out) {
	for (int64_t q = 0; q < n; ++q) {
		std::vector<Cand> c;
		for (int64_t j = 0; j < k; ++j)
			if (raw_g[q * kk + j] >= 0)
				c.push_back({raw_v[q * kk + j], raw_g[q * kk + j], 0});
		print_order_ip(c.data(), (int64_t)c.size());
		for (int64_t j = 0; j < k; ++j) {
			D_out[q * k + j] = j < (int64_t)c.size() ? c[(size_t)j].v : -FLT_MAX;
			I_out[q * k + j] = j < (int64_t)c.size() ? c[(size_t)j].id : -1;
		}
	}
	for (int64_t f = 0; f < nf; ++f) {
		const int64_t q = fq[f];
		resolve_ip_tie_host(k, raw_v + q * kk, raw_g + q * kk, first + f * k, D_out + q * k, I_out + q * k);
	}
}

// ---- ShardedIndex (csrc/sharded.hip): merge of RAW shard lists ---------------------------------------------------
// Shard s hands over kk candidates per query: value, global row number (the ordering id; < 0 = empty slot).  Output: the
// first kk of the union in the pure order (val[nq][kk], gnum[nq][kk]; unfilled slots gnum = -1).
void merge_raw_lists_host(int metric, int64_t nq, int64_t kk, int nshard, const float *const *D, const int64_t *const *G,
                          float *val, int64_t *gnum) {
	const bool is_l2 = metric_order(metric) == METRIC_L2;
	const float neutral = is_l2 ? FLT_MAX : -FLT_MAX;
	parallel_queries(nq, [&](int64_t q0, int64_t q1) {
		std::vector<Cand> c((size_t)nshard * kk);
		for (int64_t q = q0; q < q1; ++q) {
			int64_t m = 0;
			for (int s = 0; s < nshard; ++s)
				for (int64_t j = 0; j < kk; ++j)
					if (G[s][q * kk + j] >= 0)
						c[(size_t)m++] = {D[s][q * kk + j], G[s][q * kk + j], 0};
			const int64_t keep = std::min(m, kk);
			std::partial_sort(c.begin(), c.begin() + keep, c.begin() + m,
			                  [&](const Cand &a, const Cand &b) { return cand_before(is_l2, a, b); });
			for (int64_t j = 0; j < kk; ++j) {
				val[q * kk + j] = j < keep ? c[(size_t)j].v : neutral;
				gnum[q * kk + j] = j < keep ? c[(size_t)j].id : -1;
			}
		}
	});
}

// FAISS's CMin-heap outcome for one query with an exact tie at its k-th score (same closed form as
// tie_resolve_kernel, csrc/util_kernels.hip): raw = merged top-(k+1) in the pure order, first = the k smallest global
// row numbers with score >= T (ascending, -1 padded).  Writes k (value, gnum) pairs in heap_reorder's print order.
void resolve_ip_tie_host(int64_t k, const float *raw_v, const int64_t *raw_g, const int64_t *first, float *out_v,
                         int64_t *out_g) {
	const float T = raw_v[k - 1];
	int64_t ngt = 0;
	while (ngt < k && raw_v[ngt] > T)
		++ngt;
	auto above = [&](int64_t g) {
		for (int64_t j = 0; j < ngt; ++j)
			if (raw_g[j] == g)
				return true;
		return false;
	};
	int64_t in_a = 0;
	for (int64_t j = 0; j < k; ++j)
		if (first[j] >= 0 && above(first[j]))
			++in_a;
	const int64_t G = ngt - in_a;
	std::vector<Cand> c;
	for (int64_t j = 0; j < ngt; ++j)
		c.push_back({raw_v[j], raw_g[j], 0});
	int64_t seen = 0;
	for (int64_t j = 0; j < k && (int64_t)c.size() < k; ++j) {
		if (first[j] < 0 || above(first[j]))
			continue;
		if (seen++ < G)
			continue;
		c.push_back({T, first[j], 0});
	}
	print_order_ip(c.data(), (int64_t)c.size());
	for (int64_t j = 0; j < k; ++j) {
		out_v[j] = j < (int64_t)c.size() ? c[(size_t)j].v : -FLT_MAX;
		out_g[j] = j < (int64_t)c.size() ? c[(size_t)j].id : -1;
	}
}

} // namespace mvs
