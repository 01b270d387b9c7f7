// tests/san/san_merge.cpp -- ASan + UBSan harness for the HOST-side pieces of the shard exchange (csrc/merge_host.hip):
// merge_shards_host / merge_shards_raw_host / merge_records_host / finish_ip_ties_host on randomised, tie-heavy inputs, checked
// against a naive sort of the union.  Built and run by `make sanitize` in the build container (never on the GPU box: GPU ASan is
// not available on this pool).  Test infrastructure.
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mvs {
void merge_shards_host(int metric, int64_t n, int64_t k, int nshard, const float *D, const int64_t *I, float *D_out, int64_t *I_out);
void merge_shards_raw_host(int metric, int64_t n, int64_t kk, int nshard, const float *D, const int64_t *I, float *D_out, int64_t *I_out);
void merge_records_host(int metric, const int64_t *rec, int nshard, int64_t nq, int kk, int kout, bool raw, float *D_out, int64_t *I_out);
} // namespace mvs

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
	rng_state ^= rng_state << 13;
	rng_state ^= rng_state >> 7;
	rng_state ^= rng_state << 17;
	return (uint32_t)(rng_state >> 32);
}

struct C {
	float v;
	int64_t id;
};

int main() {
	int fails = 0;
	for (int metric = 0; metric <= 1; ++metric) { // 0 = inner product, 1 = L2 (include/mi355_faiss.h)
		const bool l2 = metric == 1;
		for (int trial = 0; trial < 200; ++trial) {
			const int nshard = 1 + (int)(rnd() % 8);
			const int64_t nq = 1 + rnd() % 17, k = 1 + rnd() % 40;
			std::vector<float> D((size_t)nshard * nq * k);
			std::vector<int64_t> I((size_t)nshard * nq * k);
			for (int s = 0; s < nshard; ++s)
				for (int64_t q = 0; q < nq; ++q) {
					// a shard's list: sorted in the pure order, global ids disjoint across shards, few distinct values -> ties, short lists padded
					const int64_t have = rnd() % (k + 1);
					std::vector<C> c((size_t)have);
					for (auto &e : c) {
						e.v = (float)(rnd() % 5);
						e.id = (int64_t)(rnd() % 1000) * nshard + s;
					}
					std::sort(c.begin(), c.end(), [&](const C &a, const C &b) { return a.v != b.v ? (l2 ? a.v < b.v : a.v > b.v) : a.id < b.id; });
					c.erase(std::unique(c.begin(), c.end(), [](const C &a, const C &b) { return a.id == b.id; }), c.end());
					for (int64_t j = 0; j < k; ++j) {
						const size_t o = ((size_t)s * nq + q) * k + j;
						if (j < (int64_t)c.size())
							D[o] = c[(size_t)j].v, I[o] = c[(size_t)j].id;
						else
							D[o] = l2 ? FLT_MAX : -FLT_MAX, I[o] = -1;
					}
				}
			std::vector<float> Do((size_t)nq * k), Dr((size_t)nq * k);
			std::vector<int64_t> Io((size_t)nq * k), Ir((size_t)nq * k);
			mvs::merge_shards_host(metric, nq, k, nshard, D.data(), I.data(), Do.data(), Io.data());
			mvs::merge_shards_raw_host(metric, nq, k, nshard, D.data(), I.data(), Dr.data(), Ir.data());
			// records: {value bits, label} pairs of 16 bytes per entry, [shard][query][k]
			std::vector<int64_t> rec((size_t)nshard * nq * k * 2);
			for (size_t i = 0; i < D.size(); ++i) {
				int32_t bits;
				memcpy(&bits, &D[i], 4);
				rec[2 * i] = (int64_t)bits;
				rec[2 * i + 1] = I[i];
			}
			std::vector<float> Dc((size_t)nq * k);
			std::vector<int64_t> Ic((size_t)nq * k);
			mvs::merge_records_host(metric, rec.data(), nshard, nq, (int)k, (int)k, true, Dc.data(), Ic.data());
			for (int64_t q = 0; q < nq; ++q) {
				std::vector<C> u;
				for (int s = 0; s < nshard; ++s)
					for (int64_t j = 0; j < k; ++j) {
						const size_t o = ((size_t)s * nq + q) * k + j;
						if (I[o] >= 0)
							u.push_back({D[o], I[o]});
					}
				std::sort(u.begin(), u.end(), [&](const C &a, const C &b) { return a.v != b.v ? (l2 ? a.v < b.v : a.v > b.v) : a.id < b.id; });
				for (int64_t j = 0; j < k; ++j) {
					const bool have = j < (int64_t)u.size();
					const float wv = have ? u[(size_t)j].v : (l2 ? FLT_MAX : -FLT_MAX);
					const int64_t wi = have ? u[(size_t)j].id : -1;
					if (Dr[(size_t)(q * k + j)] != wv || Ir[(size_t)(q * k + j)] != wi || Dc[(size_t)(q * k + j)] != wv || Ic[(size_t)(q * k + j)] != wi)
						++fails;
					// the printed order differs from the pure one only inside runs of equal inner-product scores (descending id)
					if (Do[(size_t)(q * k + j)] != wv)
						++fails;
				}
			}
		}
	}
	printf("san_merge: %s (%d mismatches)\n", fails ? "FAIL" : "OK", fails);
	return fails ? 1 : 0;
}
