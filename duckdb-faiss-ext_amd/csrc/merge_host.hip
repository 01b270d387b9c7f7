// csrc/merge_host.hip -- host k-way merge of per-shard (distance,label) blocks.
// Follows the RCCL all-gather in the row-sharded multi-GPU search: every shard returns its k best per query
// in FAISS order with GLOBAL labels; the merged list is the k best of the union under
//   L2: (distance asc, label asc)                    == what one CMax heap over all rows keeps
//   IP: (score desc, label asc) membership, equal scores printed in descending label order
// [faiss/utils/Heap.h heap_reorder; faiss::IndexShards merges with the same heap rule]
#include "index.h"

#include <algorithm>
#include <thread>

namespace mvs {

static void merge_range(int metric, int64_t q0, int64_t q1, int64_t n, int64_t k, int nshard, const float *D,
                        const int64_t *I, float *D_out, int64_t *I_out) {
	const bool is_l2 = metric == METRIC_L2;
	const float neutral = is_l2 ? FLT_MAX : -FLT_MAX;
	std::vector<int> pos((size_t)nshard);
	for (int64_t q = q0; q < q1; ++q) {
		// shard lists are sorted: classic k-way merge by repeatedly taking the best head
		std::fill(pos.begin(), pos.end(), 0);
		int64_t m = 0;
		for (; m < k; ++m) {
			int best = -1;
			float bv = 0;
			int64_t bi = 0;
			for (int s = 0; s < nshard; ++s) {
				// within a shard equal IP scores are printed in descending label order; membership prefers the
				// smaller label, so scan the run of equal scores for its smallest unconsumed label
				if (pos[s] >= k)
					continue;
				const float *ds = D + ((size_t)s * n + q) * k;
				const int64_t *is = I + ((size_t)s * n + q) * k;
				int p = pos[s];
				if (is[p] < 0)
					continue;
				float v = ds[p];
				int64_t id = is[p];
				if (!is_l2) {
					int e = p;
					while (e + 1 < k && is[e + 1] >= 0 && ds[e + 1] == v)
						++e;
					id = is[e]; // smallest label of the run sits at its end
				}
				bool better = best < 0 || (is_l2 ? (v < bv || (v == bv && id < bi)) : (v > bv || (v == bv && id < bi)));
				if (better) {
					best = s;
					bv = v;
					bi = id;
				}
			}
			if (best < 0)
				break;
			D_out[q * k + m] = bv;
			I_out[q * k + m] = bi;
			if (is_l2) {
				pos[best]++;
			} else {
				// consume the run's last element: shrink the run from its end by marking it used
				const float *ds = D + ((size_t)best * n + q) * k;
				int64_t *is = const_cast<int64_t *>(I + ((size_t)best * n + q) * k);
				int p = pos[best], e = p;
				while (e + 1 < k && is[e + 1] >= 0 && ds[e + 1] == ds[p])
					++e;
				// rotate: move is[e] out by shifting [p, e) right by one; cheaper: swap with head and advance
				std::swap(is[p], is[e]);
				pos[best]++;
				// restore descending order of the remaining run [p+1, e]
				std::sort(is + p + 1, is + e + 1, [](int64_t a, int64_t b) { return a > b; });
			}
		}
		if (!is_l2) {
			// print equal scores in descending label order
			int64_t a = 0;
			while (a < m) {
				int64_t b = a + 1;
				while (b < m && D_out[q * k + b] == D_out[q * k + a])
					++b;
				std::reverse(I_out + q * k + a, I_out + q * k + b);
				a = b;
			}
		}
		for (; m < k; ++m) {
			D_out[q * k + m] = neutral;
			I_out[q * k + m] = -1;
		}
	}
}

// queries are independent: the merge runs on up to 16 host threads (it sits on the critical path of every multi-GPU
// search step, after the all-gather)
void merge_shards_host(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out,
                       int64_t *I_out) {
	int nt = (int)std::min<int64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())), n / 512 + 1);
	if (nt <= 1) {
		merge_range(metric, 0, n, n, k, nshard, D, I, D_out, I_out);
		return;
	}
	std::vector<std::thread> th;
	for (int t = 0; t < nt; ++t)
		th.emplace_back(merge_range, metric, n * t / nt, n * (t + 1) / nt, n, k, nshard, D, I, D_out, I_out);
	for (auto &x : th)
		x.join();
}

} // namespace mvs
