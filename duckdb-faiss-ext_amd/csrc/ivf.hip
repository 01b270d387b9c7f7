// csrc/ivf.hip -- faiss::IndexIVFFlat on device ("IVF<nlist>,Flat").
//
// Replaces what the reference reaches through index_factory (:154), Index::train (:396,:583), add/add_with_ids
// (:607,:609 -- one call with all rows after training) and search with SearchParametersIVF{nprobe,sel} (:631,
// :675-689) of /root/reference/src/faiss_extension.cpp.  Restated FAISS behaviour [UPSTREAM: faiss/IndexIVF.cpp,
// IndexIVFFlat.cpp, Clustering.cpp, utils/random.cpp; see oracle/orc_core.c for the line-by-line restatement]:
//   train  : k-means (niter 10 = Level1Quantizer's cp.niter, seed 1234, <= 256 points per centroid subsample via rand_perm, centroids initialised
//            from rand_perm(seed+1), empty-cluster splitting, spherical for inner product).  The ASSIGNMENT step --
//            11 TFLOP at IVF4096 -- runs on the fused MFMA Flat kernel (k = 1); the centroid update keeps FAISS's
//            sequential summation order on device too (stable sort by assignment + one thread per (centroid, dim),
//            csrc/kmeans_update.hip); the host only replays the RNG-driven empty-cluster splits on k x d values.
//   add    : assign = quantizer search k=1 in blocks of 65536 rows; (id, raw vector) appended to list assign[i] in
//            input order.  Device store: append-only rows + a CSR view (rows grouped by list, input order inside a
//            list) rebuilt lazily before the first search after an add.
//   search : coarse = quantizer search k=nprobe; list scan in per-pair arithmetic (IVFFlatScanner::scan_codes):
//            LIST-MAJOR -- queries are grouped per probed list so every list is streamed from HBM once per <= 20
//            queries (flat_direct.hip item mode, HBM-bound), then one merge per query over its nprobe partial lists.
#include "index.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <atomic>
#include <chrono>
#include <thread>

namespace mvs {

namespace {

// faiss::rand_perm (utils/random.cpp): Fisher-Yates with mt19937, i2 = i + rng() % (n - i)
std::vector<int> rand_perm(size_t n, int64_t seed) {
	std::vector<int> perm(n);
	for (size_t i = 0; i < n; i++)
		perm[i] = (int)i;
	std::mt19937 mt((unsigned)seed);
	for (size_t i = 0; i + 1 < n; i++) {
		int i2 = (int)i + (int)(mt() % (uint32_t)(n - i));
		std::swap(perm[i], perm[i2]);
	}
	return perm;
}

float chain_norm(const float *x, int d) {
	float acc = 0.f;
	for (int k = 0; k < d; k++)
		acc = fmaf(x[k], x[k], acc);
	return acc;
}
void renorm_l2(int d, int64_t n, float *x) {
	for (int64_t i = 0; i < n; i++) {
		float *xi = x + i * d;
		float nr = chain_norm(xi, d);
		if (nr > 0) {
			const float inv = 1.0f / sqrtf(nr);
			for (int j = 0; j < d; j++)
				xi[j] *= inv;
		}
	}
}

} // namespace

// rows of the padded MFMA store: dst[r][0..d) = src[perm[r]][0..d) (src rows have stride sdp), zeros where perm < 0
__global__ void ivf_gather_plain_kernel(const float *src, int sdp, const int *perm, long long n, int d, float *dst) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * d)
		return;
	const long long r = i / d;
	const int j = (int)(i - r * d);
	const int p = perm[r];
	dst[i] = p >= 0 ? src[(size_t)p * sdp + j] : 0.f;
}
__global__ void ivf_fill_empty_kernel(float *D, long long *I, long long n, float neutral) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		D[i] = neutral;
		I[i] = -1;
	}
}
__global__ void ivf_gather_ids_kernel(const long long *src, const int *perm, long long n, long long *dst) {
	const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n)
		dst[r] = perm[r] >= 0 ? src[perm[r]] : -1;
}

// rows of the MFMA list store (mf order) -> residuals against their list's centroid (padding rows stay zero)
__global__ void ivf_residual_kernel(float *rows, const int *perm, long long n, int d, const int *list_of_blk64, const float *cent) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * d)
		return;
	const long long r = i / d;
	const int j = (int)(i - r * d);
	if (perm[r] >= 0)
		rows[i] = __fsub_rn(rows[i], cent[(size_t)list_of_blk64[r >> 6] * d + j]);
}
// mean row of every list (rows in the padded MFMA order, [lb, le) per list): out[l][0..d); an empty list keeps what out holds
__global__ void ivf_list_mean_kernel(const float *__restrict__ rows, const long long *__restrict__ lb, const long long *__restrict__ le,
                                     int d, float *__restrict__ out) {
	const long long l = blockIdx.x, b = lb[l], e = le[l];
	if (e <= b)
		return;
	for (int j = threadIdx.x; j < d; j += blockDim.x) {
		float acc = 0.f;
		for (long long r = b; r < e; ++r)
			acc += rows[(size_t)r * d + j];
		out[(size_t)l * d + j] = acc / (float)(e - b);
	}
}
// selected (value, position) lists [nq][kk], best first -> D / I [nq][k] with labels = stored ids
// flagged queries -> list
// ---- lists beyond 32 entries through the bf16 filter (round 6; IVFFlatIndex::collect_search_big)
// B(q) in the scan's value space (larger = better: -distance | score) from the kk-th best exact value of the query's nearest lists
__global__ void ivf_big_bound_kernel(const float *__restrict__ D, int kk, long long nq, int is_l2, float *__restrict__ bfix) {
	const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= nq)
		return;
	const float v = D[q * kk + kk - 1];
	// (fewer than kk rows in those lists: FLT_MAX | -FLT_MAX -- no bound, everything passes)
	bfix[q] = is_l2 ? (v < FLT_MAX ? -v : -FLT_MAX) : (v > -FLT_MAX ? v : -FLT_MAX);
}
// The bound without the host (one workgroup per query): exact values of the first <= R rows of the query's nearest lists, in probe
// order (the scanner's chains on the list-sorted f32 rows), then the kk-th best of them by a 32-step search over the keys.  Rows in
// insertion order are a sample of their list: kk of them at least B good -- a valid bound, a little lower than the whole lists' would be.
__device__ __forceinline__ bool ivf_big_sel_member(const SelectorDev &s, long long id) {
	if (s.kind == MVS_SEL_BITMAP) {
		const unsigned long long u = (unsigned long long)id;
		if ((u >> 3) >= (unsigned long long)s.nbytes)
			return false;
		return (s.bitmap[u >> 3] >> (u & 7)) & 1;
	}
	if (s.kind == MVS_SEL_BATCH) {
		long long lo = 0, hi = s.nids;
		while (lo < hi) {
			const long long mid = (lo + hi) >> 1;
			if (s.sorted_ids[mid] < id)
				lo = mid + 1;
			else
				hi = mid;
		}
		return lo < s.nids && s.sorted_ids[lo] == id;
	}
	return true;
}
__global__ __launch_bounds__(256) void ivf_big_bound_direct_kernel(const float *__restrict__ xq, int d, int dp, const float *__restrict__ rows,
                                                                  const long long *__restrict__ probes, int np,
                                                                  const long long *__restrict__ list_off, int R, int kk, int is_l2,
                                                                  SelectorDev sel, const long long *__restrict__ rowids,
                                                                  const long long *__restrict__ idmap, float *__restrict__ bfix) {
	extern __shared__ unsigned big_keys[]; // [R] order-preserving keys (smaller = better) | 8 words of reduction scratch | [R] rows | the tile
	unsigned *red = big_keys + R;
	int *rowidx = (int *)(red + 8);
	float(*tile)[33] = (float(*)[33])(rowidx + R); // [256][33]: a slab of 32 dimensions of 256 rows (coalesced reads, conflict-free walks)
	const long long q = blockIdx.x;
	const int t = threadIdx.x;
	const float *x = xq + (size_t)q * d;
	int have = 0; // (uniform) rows taken so far
	for (int p = 0; p < np && have < R; ++p) {
		const long long l = probes[q * np + p];
		if (l < 0)
			continue;
		const long long b = list_off[l];
		const int len = (int)(list_off[l + 1] - b), take = len < R - have ? len : R - have;
		for (int i = t; i < take; i += 256)
			rowidx[have + i] = (int)(b + i);
		have += take;
	}
	__syncthreads();
	for (int base = 0; base < have; base += 256) {
		const int mine = base + t;
		float acc = 0.f;
		for (int s0 = 0; s0 < d; s0 += 32) {
			__syncthreads();
#pragma unroll
			for (int i = 0; i < 8; ++i) { // 2 048 chunks of 16 bytes: chunk c = row (c >> 3), floats 4 (c & 7) ..
				const int c = i * 256 + t, r = c >> 3, j4 = (c & 7) * 4;
				float4 v = {0.f, 0.f, 0.f, 0.f};
				if (base + r < have && s0 + j4 < dp)
					v = *(const float4 *)(rows + (size_t)rowidx[base + r] * dp + s0 + j4);
				tile[r][j4] = v.x, tile[r][j4 + 1] = v.y, tile[r][j4 + 2] = v.z, tile[r][j4 + 3] = v.w;
			}
			__syncthreads();
			const int kend = d - s0 < 32 ? d - s0 : 32;
			for (int j = 0; j < kend; ++j) { // the scanner's chains: k ascending
				if (is_l2) {
					const float tt = x[s0 + j] - tile[t][j];
					acc = fmaf(tt, tt, acc);
				} else {
					acc = fmaf(x[s0 + j], tile[t][j], acc);
				}
			}
		}
		if (mine < have) {
			bool ok = true;
			if (sel.kind != MVS_SEL_NONE) {
				const long long lab = rowids[rowidx[mine]];
				ok = ivf_big_sel_member(sel, idmap ? idmap[lab] : lab);
			}
			const unsigned fb = __float_as_uint(acc), fk = fb ^ ((fb >> 31) ? 0xFFFFFFFFu : 0x80000000u); // unsigned order = float order
			big_keys[mine] = ok ? (is_l2 ? fk : ~fk) : 0xFFFFFFFFu; // (rejected rows: behind everything; NaN sorts last either way)
		}
	}
	__syncthreads();
	// the kk-th smallest key: the largest U with fewer than kk keys below it, bit by bit
	unsigned U = 0u;
	for (int bit = 31; bit >= 0; --bit) {
		const unsigned tt = U | (1u << bit);
		int c = 0;
		for (int i = t; i < have; i += 256)
			c += big_keys[i] < tt ? 1 : 0;
		for (int o = 32; o >= 1; o >>= 1)
			c += __shfl_xor(c, o);
		if ((t & 63) == 0)
			red[t >> 6] = (unsigned)c;
		__syncthreads();
		const int total = (int)(red[0] + red[1] + red[2] + red[3]);
		__syncthreads();
		if (total < kk)
			U = tt;
	}
	if (t == 0) {
		float B = -FLT_MAX; // (fewer than kk admissible rows in reach: no bound, everything passes)
		if (have >= kk && U != 0xFFFFFFFFu) {
			const unsigned fk = is_l2 ? U : ~U;
			const float v = __uint_as_float((fk & 0x80000000u) ? (fk ^ 0x80000000u) : ~fk);
			B = is_l2 ? -v : v;
			if (!(B > -FLT_MAX))
				B = -FLT_MAX;
		}
		bfix[q] = B;
	}
}
// stream entries (q << 32 | padded row) -> (q << 32 | position in the list-sorted store); padding rows never pass the scan
__global__ void ivf_big_translate_kernel(unsigned long long *__restrict__ strm, long long n, const int *__restrict__ perm) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const unsigned long long e = strm[i];
	const int pos = perm[(unsigned)e];
	strm[i] = (e & 0xffffffff00000000ull) | (unsigned)(pos < 0 ? 0 : pos);
}
// (value, position) lists [nq][k], best first -> D / I: positions (inside the exact-tie wrapper) or stored ids through the id map
__global__ void ivf_big_emit_kernel(const float *__restrict__ pd, const int *__restrict__ pi, long long total, int is_l2,
                                    const long long *__restrict__ rowids, const long long *__restrict__ idmap, float *__restrict__ D,
                                    long long *__restrict__ I) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total)
		return;
	const int pos = pi[i];
	long long lab = -1;
	if (pos >= 0) {
		lab = rowids ? rowids[pos] : (long long)pos;
		if (idmap)
			lab = idmap[lab];
	}
	D[i] = pos >= 0 ? pd[i] : (is_l2 ? FLT_MAX : -FLT_MAX);
	I[i] = lab;
}
// tie_emit's precondition: flag = {nf, query numbers ...}; a number outside [0, nq) sets *bad (pinned host memory)
__global__ void ivf_flag_check_kernel(const int *__restrict__ flag, int nf, int nq, int *bad) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nf && (flag[1 + i] < 0 || flag[1 + i] >= nq))
		*bad = 1;
}
__global__ void ivf_max_norm_kernel(const float *norms, long long n, unsigned *out_bits) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	float v = i < n ? norms[i] : 0.f;
	for (int o = 32; o >= 1; o >>= 1)
		v = fmaxf(v, __shfl_xor(v, o));
	if ((threadIdx.x & 63) == 0 && __float_as_uint(v) > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) // (agent scope: a plain load stays as first cached)
		atomicMax(out_bits, __float_as_uint(v)); // squared norms are >= 0: the bit pattern orders like the value
}

// L2 list scan through the MFMA ITEMS kernel as a PREFILTER (same scheme as csrc/flat_bf16.hip).  The kernel's value F is the
// BLAS-branch formula on RESIDUAL rows and queries, (rxn + ryn) - 2<r_x, r_y>, r = fl(v - centroid); IVFFlatScanner computes
// S = sum (x - y)^2 on the original vectors.  With u = 2^-24, D* the real distance, N_l = rxn_l + ryn for a row of list l:
//   rounding of the residuals:   | ||r_x - r_y||^2 - D* | <= 4 u N_l
//   formula (three chains + two roundings):  | F - ||r_x - r_y||^2 | <= (2d + 2) u N_l
//   scanner chain:               | S - D* | <= (d + 3) u D* <= 2 (d + 3) u N_l
// so |F - S| <= E_l = (4d + 12) u N_l (the kernel uses 1.25 x that, with ryn = the largest residual-row norm of the index).
// rxn_l = ||x - c_l||^2 is small for the lists the query is close to and in the hundreds for far probes, so the bound is
// taken over the RELEVANT lists only: those that hold one of the kp candidates, plus every probed list that the triangle
// inequality cannot exclude -- a row of list l has D* >= (||x - c_l|| - ||y - c_l||)^2 >= lb_l = (sqrt(rxn_l) - sqrt(ryn_max))^2.
// One wave per query; lane p <-> probe p (distances to the probed centroids), lane j <-> candidate j of the merged top-kp (F
// values, best first, row POSITIONS in the MFMA list store): exact_j = the scanner's value of the ORIGINAL row (CSR store,
// through perm; t = x_k - y_k, acc = fmaf(t, t, acc), k ascending).  Proof, with E = max E_l over relevant lists:
//   the k best candidates have S <= a_(k) + E, so the exact k-th value T <= a_(k) + E;
//   a row of an irrelevant list has S >= lb_l (1 - eps) > a_(k) + 2E >= T: not in the exact top-k;
//   a row of a relevant list in the exact top-k has F <= S + E <= a_(k) + 2E < a_(kp): it is one of the kp candidates.
// Queries that cannot be proven are appended to fail_q and re-run on the scanner kernel with the same coarse assignment.
__global__ __launch_bounds__(64) void ivf_rescore_verify_kernel(const float *__restrict__ ca, const long long *__restrict__ ci,
                                                               int kp, int k, const float *__restrict__ x, int d,
                                                               const float *__restrict__ rows_csr, int dp_csr,
                                                               const int *__restrict__ perm,
                                                               const long long *__restrict__ coarse, int np,
                                                               const float *__restrict__ cent,
                                                               const int *__restrict__ list_of_blk64,
                                                               const unsigned *__restrict__ max_norm_bits,
                                                               float *__restrict__ pd1, int *__restrict__ pi1,
                                                               int *__restrict__ fail_cnt, int *__restrict__ fail_q) {
	const long long q = blockIdx.x;
	const int j = threadIdx.x;
	const float *xq = x + q * d;
	long long pos = -1;
	float av = FLT_MAX;
	if (j < kp) {
		pos = ci[q * kp + j];
		av = ca[q * kp + j];
	}
	float ex = FLT_MAX;
	int clist = -1;
	if (pos >= 0) {
		const float *y = rows_csr + (size_t)perm[pos] * dp_csr;
		float acc = 0.f;
		for (int kk = 0; kk < d; ++kk) {
			const float t = __fsub_rn(xq[kk], y[kk]);
			acc = fmaf(t, t, acc);
		}
		ex = acc;
		clist = list_of_blk64[pos >> 6];
	}
	if (j < kp) {
		pd1[q * kp + j] = ex;
		pi1[q * kp + j] = (int)pos;
	}
	const int navail = __popcll(__builtin_amdgcn_ballot_w64(pos >= 0));
	if (navail < kp)
		return; // every comparable row of the probed lists is a candidate: nothing to prove
	const float a_k = __shfl(av, k - 1), a_kp = __shfl(av, kp - 1);
	const float rymax = __uint_as_float(*max_norm_bits);
	const float cu = 1.25f * (float)(4 * d + 12) * 5.9604645e-8f;
	bool ok = kp > k;
	// pass 1: E0 over the candidates' own lists; pass 2: add the lists the triangle inequality cannot exclude; pass 3: check
	float e_rel = 0.f;
	for (int pass = 0; pass < 3 && ok; ++pass) {
		float emax = 0.f;
		bool bad = false;
		for (int p0 = 0; p0 < np; p0 += 64) {
			const int p = p0 + j;
			const long long l = p < np ? coarse[q * np + p] : -1;
			float rx = 0.f, lb = FLT_MAX;
			if (l >= 0) {
				const float *c = cent + (size_t)l * d;
				for (int kk = 0; kk < d; ++kk) {
					const float r = __fsub_rn(xq[kk], c[kk]);
					rx = fmaf(r, r, rx);
				}
				const float gap = sqrtf(rx) - sqrtf(rymax);
				lb = gap > 0.f ? gap * gap * 0.9999f : 0.f;
			}
			bool rel = false; // holds a candidate?
			for (int jj = 0; jj < kp; ++jj)
				rel |= l >= 0 && __shfl(clist, jj) == (int)l;
			if (pass >= 1)
				rel |= l >= 0 && lb <= a_k + 8.f * e_rel;
			if (pass < 2) {
				if (rel)
					emax = fmaxf(emax, cu * (rx + rymax));
			} else if (l >= 0 && !rel && !(lb > a_k + 2.f * e_rel)) {
				bad = true;
			}
		}
		if (pass < 2) {
			for (int o = 32; o >= 1; o >>= 1)
				emax = fmaxf(emax, __shfl_xor(emax, o));
			e_rel = emax;
		} else if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
			ok = false;
		}
	}
	ok = ok && a_kp > a_k + 2.f * e_rel; // (false for NaN / inf as well)
	if (!ok && j == 0)
		fail_q[atomicAdd(fail_cnt, 1)] = (int)q;
}
__global__ void ivf_map_labels_kernel(long long *I, long long total, const long long *idmap) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < total && I[i] >= 0)
		I[i] = idmap[I[i]];
}

class IVFFlatIndex : public IndexBase {
public:
	// owned; nlist centroids.  IndexFlat(d, metric) for "IVF<n>,Flat"; IndexHNSWFlat(d, M, metric) for
	// "IVF<n>_HNSW<M>,Flat" (reference Makefile:93), in which case k-means assigns with a temporary IndexFlatL2 and
	// the final centroids are inserted into the graph (index_factory: quantizer_trains_alone = 2)
	IndexBase *quantizer;
	int64_t nlist;
	int hnsw_M; // 0 = flat quantizer
	int64_t nprobe = 1;
	bool spherical;
	int dp; // padded row length of the list store (plain row-major)

	IVFFlatIndex(int d_, int64_t nlist_, int metric_, int hnsw_M_ = 0)
	    : IndexBase(MVS_KIND_IVFFLAT, d_, metric_), nlist(nlist_), hnsw_M(hnsw_M_) {
		if (metric != METRIC_L2 && metric != METRIC_IP)
			throw_faiss("mvs::IVFFlatIndex", __FILE__, "metric type %d is not implemented on the MI355X path", metric);
		if (hnsw_M > 0)
			quantizer = make_hnsw_index(d, "HNSW" + std::to_string(hnsw_M), metric);
		else
			quantizer = new FlatIndex(d, metric);
		is_trained = false;
		spherical = metric == METRIC_IP; // IndexIVF ctor: "Spherical by default if the metric is inner_product"
		dp = d <= 8 ? 8 : (d <= 16 ? 16 : (d + 31) / 32 * 32);
	}
	~IVFFlatIndex() override {
		(void)hipSetDevice(device);
		if (stream)
			(void)hipStreamSynchronize(stream);
		if (raw)
			(void)hipFree(raw);
		if (h_fail)
			(void)hipHostFree(h_fail);
		delete quantizer;
	}

	// ---------------------------------------------------------------------------------------------- train
	void train(int64_t n, const float *x) override {
		use_device();
		if (quantizer->ntotal == nlist) { // "IVF quantizer does not need training."
			is_trained = true;
			return;
		}
		const auto t0 = std::chrono::steady_clock::now();
		struct Report { // (MVS_INGEST_PROFILE=1: where an IVF ingest's time goes -- host/boundary_driver ingest)
			std::chrono::steady_clock::time_point t0;
			int64_t n;
			~Report() {
				if (getenv("MVS_INGEST_PROFILE"))
					fprintf(stderr, "ivfprofile\ttrain(%lld rows): %.3f s\n", (long long)n,
					        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
			}
		} report {t0, n};
		if (hnsw_M > 0) {
			// Level1Quantizer::train_q1, quantizer_trains_alone == 2: k-means on an IndexFlatL2, centroids -> quantizer
			FlatIndex assigner(d, METRIC_L2);
			kmeans(n, x, &assigner);
			std::vector<float> cent((size_t)nlist * d);
			assigner.copy_rows_to_host(cent.data());
			quantizer->add(nlist, cent.data());
		} else {
			kmeans(n, x, static_cast<FlatIndex *>(quantizer));
		}
		is_trained = true;
	}

	void kmeans(int64_t nx, const float *x_in, FlatIndex *qz) {
		// Clustering defaults (faiss/Clustering.h) except niter: Level1Quantizer's constructor sets cp.niter = 10 for
		// every IndexIVF ("typically used for large clusterings", faiss/IndexIVF.cpp)
		const int niter = 10, max_pts = 256, min_pts = 39;
		const int64_t seed = 1234, k = nlist;
		if (nx < k)
			throw_faiss("virtual void faiss::Clustering::train_encoded(...)", "faiss/Clustering.cpp",
			            "Error: 'nx >= k' failed: Number of training points (%ld) should be at least as large as number "
			            "of clusters (%ld)",
			            (long)nx, (long)k);
		{
			// FAISS checks EVERY training value on the host (faiss/Clustering.cpp); 1.28 G values at C3's ingest: split over threads
			const int64_t tot = nx * d;
			const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(16, std::min<int64_t>(std::thread::hardware_concurrency(), tot >> 22)));
			std::atomic<bool> bad(false);
			auto scan = [&](int64_t b, int64_t e) {
				bool ok = true;
				for (int64_t i = b; i < e && ok; i += 4096) {
					const int64_t m = std::min<int64_t>(e, i + 4096);
					float acc = 0.f; // (x * 0 is 0 for every finite x, NaN otherwise: a branch-free pass the compiler vectorises)
					for (int64_t j = i; j < m; ++j)
						acc += x_in[j] * 0.0f;
					ok = acc == 0.f;
				}
				if (!ok)
					bad.store(true);
			};
			std::vector<std::thread> th;
			for (int t = 1; t < nt; ++t)
				th.emplace_back(scan, tot * t / nt, tot * (t + 1) / nt);
			scan(0, tot / nt);
			for (auto &t : th)
				t.join();
			if (bad.load())
				throw_faiss("virtual void faiss::Clustering::train_encoded(...)", "faiss/Clustering.cpp",
				            "input contains NaN's or Inf's");
		}
		std::vector<float> sub;
		const float *x = x_in;
		if (nx > k * max_pts) { // subsample_training_set
			std::vector<int> perm = rand_perm((size_t)nx, seed);
			const int64_t n2 = k * max_pts;
			sub.resize((size_t)n2 * d);
			for (int64_t i = 0; i < n2; i++)
				memcpy(&sub[(size_t)i * d], x_in + (int64_t)perm[i] * d, (size_t)d * sizeof(float));
			x = sub.data();
			nx = n2;
		} else if (nx < k * min_pts) {
			fprintf(stderr, "WARNING clustering %ld points to %ld centroids: please provide at least %ld training points\n",
			        (long)nx, (long)k, (long)(k * min_pts));
		}
		std::vector<float> cent((size_t)k * d);
		if (nx == k) {
			memcpy(cent.data(), x, cent.size() * sizeof(float));
			qz->reset();
			qz->add(k, cent.data());
			return;
		}
		{
			std::vector<int> perm = rand_perm((size_t)nx, seed + 1);
			for (int64_t i = 0; i < k; i++)
				memcpy(&cent[(size_t)i * d], x + (int64_t)perm[i] * d, (size_t)d * sizeof(float));
		}
		if (spherical)
			renorm_l2(d, k, cent.data());
		qz->reset();
		qz->add(k, cent.data());

		// the training sample lives on device for the 25 assignment passes
		DevBuf dx, dD, dI;
		dx.reserve((size_t)nx * d * sizeof(float));
		dD.reserve((size_t)nx * sizeof(float));
		dI.reserve((size_t)nx * sizeof(int64_t));
		MVS_HIP(hipMemcpyAsync(dx.p, x, (size_t)nx * d * sizeof(float), hipMemcpyHostToDevice, stream));
		std::vector<float> hassign((size_t)k);
		DevBuf dcent, dhass, dws;
		dcent.reserve((size_t)k * d * sizeof(float));
		dhass.reserve((size_t)k * sizeof(float));
		const size_t wsb = kmeans_update_ws_bytes(nx, k);
		dws.reserve(wsb);
		for (int it = 0; it < niter; it++) {
			// (round 6: the assignment is a Flat search with k = 1 over a few thousand rows -- csrc/coarse_bf16.hip serves it, same labels)
			if (!qz->coarse_topk(nx, (const float *)dx.p, 1, (float *)dD.p, (int64_t *)dI.p, stream, false))
				qz->search_device(nx, (const float *)dx.p, 1, (float *)dD.p, (int64_t *)dI.p, nullptr, stream);
			use_device();
			// compute_centroids on device in FAISS's summation order (csrc/kmeans_update.hip); only the k x d
			// centroids and the k counts come back for the (rare, RNG-driven) empty-cluster splits
			launch_kmeans_update((const float *)dx.p, nx, d, (const int64_t *)dI.p, k, (float *)dcent.p, (float *)dhass.p,
			                     dws.p, wsb, stream);
			MVS_HIP(hipMemcpyAsync(cent.data(), dcent.p, cent.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
			MVS_HIP(hipMemcpyAsync(hassign.data(), dhass.p, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, stream));
			MVS_HIP(hipStreamSynchronize(stream));
			{
				double counted = 0;
				for (float h : hassign)
					counted += h;
				if ((int64_t)counted != nx) // compute_centroids: FAISS_ASSERT(ci >= 0 && ci < k)
					throw_faiss("void faiss::compute_centroids(...)", "faiss/Clustering.cpp",
					            "Error: 'ci >= 0 && ci < k' failed: %lld of %lld training points have no finite nearest "
					            "centroid (distance overflow)",
					            (long long)(nx - (int64_t)counted), (long long)nx);
			}
			// split_clusters
			{
				const float EPS = (float)(1 / 1024.);
				std::mt19937 mt(1234);
				for (int64_t ci = 0; ci < k; ci++) {
					if (hassign[(size_t)ci] != 0)
						continue;
					int64_t cj;
					for (cj = 0;; cj = (cj + 1) % k) {
						const float p = (hassign[(size_t)cj] - 1.0f) / (float)(nx - k);
						const float r = (float)mt() / (float)mt.max();
						if (r < p)
							break;
					}
					memcpy(&cent[(size_t)ci * d], &cent[(size_t)cj * d], (size_t)d * sizeof(float));
					for (int j = 0; j < d; j++) {
						if (j % 2 == 0) {
							cent[(size_t)ci * d + j] *= 1 + EPS;
							cent[(size_t)cj * d + j] *= 1 - EPS;
						} else {
							cent[(size_t)ci * d + j] *= 1 - EPS;
							cent[(size_t)cj * d + j] *= 1 + EPS;
						}
					}
					hassign[(size_t)ci] = hassign[(size_t)cj] / 2;
					hassign[(size_t)cj] -= hassign[(size_t)ci];
				}
			}
			if (spherical)
				renorm_l2(d, k, cent.data());
			qz->reset();
			qz->add(k, cent.data());
		}
	}

	// ---------------------------------------------------------------------------------------------- add
	void grow(int64_t need) {
		if (need <= cap)
			return;
		int64_t nc = cap ? cap : 4096;
		while (nc < need)
			nc = nc + nc / 2 + 4096;
		float *nr = nullptr;
		MVS_HIP(hipMalloc((void **)&nr, (size_t)nc * dp * sizeof(float)));
		if (ntotal > 0)
			MVS_HIP(hipMemcpyAsync(nr, raw, (size_t)ntotal * dp * sizeof(float), hipMemcpyDeviceToDevice, stream));
		MVS_HIP(hipStreamSynchronize(stream));
		if (raw)
			MVS_HIP(hipFree(raw));
		raw = nr;
		cap = nc;
	}

	// d_x: [n][d] rows already on device (on `stream` order); ids (host) may be null
	void add_core_device(int64_t n, const float *d_x, const int64_t *ids_host) {
		if (!is_trained)
			throw_faiss("virtual void faiss::IndexIVFFlat::add_core(...)", "faiss/IndexIVFFlat.cpp",
			            "Error: 'is_trained' failed");
		if (ntotal + n > (int64_t)0x7fffffff - 1024)
			throw_faiss("mvs::IVFFlatIndex::add", __FILE__, "a single-device shard holds at most 2^31 rows");
		grow(ntotal + n);
		// (the cross-process tie merge rebuilds FAISS's arrival order inside a list as (probe rank, stored id) -- true only
		// while ids grow with insertion order; mvs_index_get_stat "ivf_ids_ascending" says whether they still do)
		if (ids_host)
			for (int64_t i = 0; i < n && ids_ascending; ++i) {
				ids_ascending = ids_host[i] > last_id_seen;
				last_id_seen = ids_host[i];
			}
		else {
			ids_ascending = ids_ascending && ntotal > last_id_seen;
			last_id_seen = std::max<int64_t>(last_id_seen, ntotal + n - 1);
		}
		const int64_t bs = 65536; // IndexIVF::add_with_ids block size
		DevBuf dD, dI;
		dD.reserve((size_t)std::min(bs, n) * sizeof(float));
		dI.reserve((size_t)std::min(bs, n) * sizeof(int64_t));
		std::vector<int64_t> lab((size_t)std::min(bs, n));
		assign_h.reserve((size_t)(ntotal + n));
		ids_h.reserve((size_t)(ntotal + n));
		for (int64_t i0 = 0; i0 < n; i0 += bs) {
			const int64_t nb = std::min(bs, n - i0);
			if (!(hnsw_M == 0 && static_cast<FlatIndex *>(quantizer)->coarse_topk(nb, d_x + i0 * d, 1, (float *)dD.p, (int64_t *)dI.p, stream, false)))
				quantizer->search_device(nb, d_x + i0 * d, 1, (float *)dD.p, (int64_t *)dI.p, nullptr, stream);
			use_device();
			MVS_HIP(hipMemcpyAsync(lab.data(), dI.p, (size_t)nb * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
			launch_pad_rows(d_x + i0 * d, nb, d, raw + (size_t)(ntotal + i0) * dp, dp, stream);
			MVS_HIP(hipStreamSynchronize(stream));
			for (int64_t i = 0; i < nb; i++) {
				assign_h.push_back((int32_t)lab[(size_t)i]);
				ids_h.push_back(ids_host ? ids_host[i0 + i] : label_offset + ntotal + i0 + i);
			}
		}
		ntotal += n;
		dirty = true;
	}
	// row shards: the implicit ids of a shard start at its first global row; they are stored in the lists,
	// so the offset has to be known before the first add
	void set_label_offset(int64_t off) override {
		if (ntotal > 0 && off != label_offset)
			throw_faiss("mvs::IVFFlatIndex::set_label_offset", __FILE__,
			            "the label offset of an IVF index must be set before rows are added");
		label_offset = off;
	}
	void add_host(int64_t n, const float *x, const int64_t *ids) {
		use_device();
		if (n <= 0)
			return;
		const auto t0 = std::chrono::steady_clock::now();
		DevBuf dx;
		dx.reserve((size_t)n * d * sizeof(float));
		MVS_HIP(hipMemcpyAsync(dx.p, x, (size_t)n * d * sizeof(float), hipMemcpyHostToDevice, stream));
		add_core_device(n, (const float *)dx.p, ids);
		if (getenv("MVS_INGEST_PROFILE"))
			fprintf(stderr, "ivfprofile\tadd(%lld rows: H2D + assign + append): %.3f s\n", (long long)n,
			        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	}
	void add(int64_t n, const float *x) override {
		add_host(n, x, nullptr);
	}
	void add_with_ids(int64_t n, const float *x, const int64_t *ids) override {
		add_host(n, x, ids);
	}
	void add_device(int64_t n, const float *d_x, hipStream_t st) override {
		use_device();
		if (n <= 0)
			return;
		stream_wait(stream, st);
		add_core_device(n, d_x, nullptr);
	}
	void add_with_ids_device(int64_t n, const float *d_x, const int64_t *d_ids, hipStream_t st) override {
		use_device();
		if (n <= 0)
			return;
		stream_wait(stream, st);
		std::vector<int64_t> ids((size_t)n);
		MVS_HIP(hipMemcpy(ids.data(), d_ids, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
		add_core_device(n, d_x, ids.data());
	}

	// CSR view: rows grouped by list, input order inside a list (ArrayInvertedLists semantics)
	void build_lists() {
		if (!dirty)
			return;
		const auto t0_bl = std::chrono::steady_clock::now();
		struct ReportBl {
			std::chrono::steady_clock::time_point t0;
			~ReportBl() {
				if (getenv("MVS_INGEST_PROFILE"))
					fprintf(stderr, "ivfprofile\tbuild_lists (CSR view): %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
			}
		} report_bl {t0_bl};
		list_off.assign((size_t)nlist + 1, 0);
		for (int64_t i = 0; i < ntotal; i++) {
			const int32_t l = assign_h[(size_t)i];
			if (l >= 0)
				list_off[(size_t)l + 1]++;
		}
		for (int64_t l = 0; l < nlist; l++)
			list_off[(size_t)l + 1] += list_off[(size_t)l];
		nsorted = list_off[(size_t)nlist];
		std::vector<int64_t> cursor(list_off.begin(), list_off.end() - 1);
		std::vector<int32_t> perm((size_t)nsorted);
		std::vector<int64_t> sid((size_t)nsorted);
		for (int64_t i = 0; i < ntotal; i++) {
			const int32_t l = assign_h[(size_t)i];
			if (l < 0)
				continue;
			const int64_t p = cursor[(size_t)l]++;
			perm[(size_t)p] = (int32_t)i;
			sid[(size_t)p] = ids_h[(size_t)i];
		}
		codes.reserve(((size_t)nsorted * dp + 64) * sizeof(float));
		rowids.reserve((size_t)std::max<int64_t>(nsorted, 1) * sizeof(int64_t));
		DevBuf dperm;
		dperm.reserve((size_t)std::max<int64_t>(nsorted, 1) * sizeof(int32_t));
		if (nsorted > 0) {
			MVS_HIP(hipMemcpyAsync(dperm.p, perm.data(), (size_t)nsorted * sizeof(int32_t), hipMemcpyHostToDevice, stream));
			MVS_HIP(hipMemcpyAsync(rowids.p, sid.data(), (size_t)nsorted * sizeof(int64_t), hipMemcpyHostToDevice, stream));
			launch_gather_rows(raw, (const int *)dperm.p, nsorted, dp, (float *)codes.p, stream);
		}
		list_off_dev.reserve(list_off.size() * sizeof(int64_t));
		MVS_HIP(hipMemcpyAsync(list_off_dev.p, list_off.data(), list_off.size() * sizeof(int64_t), hipMemcpyHostToDevice,
		                       stream));
		MVS_HIP(hipStreamSynchronize(stream));
		dirty = false;
		mf_dirty = true;
	}

	// Second view of the lists for the MFMA variant of the scan (csrc/flat_mfma.hip ITEMS): the Flat storage format
	// (pair-interleaved rows of geom.dp floats) with every list padded to a multiple of 64 rows, plus row norms.
	// Built lazily the first time that variant runs.
	// need_f32: also the f32 (MFMA-packed) copy of the rows + their norms, which only the f32 ITEMS scans read (inner product,
	// options ivf_mfma = 1 / 2); the L2 default (csrc/ivf_collect.hip) reads the bf16 residual copy and the list-sorted store
	void build_lists_mf(bool need_f32 = true) {
		build_lists();
		if (!mf_dirty && (!need_f32 || mf_have_f32))
			return;
		mf_have_f32 = need_f32;
		geom = flat_geom_for(d);
		std::vector<int64_t> pb((size_t)nlist), pe((size_t)nlist);
		int64_t pos = 0;
		for (int64_t l = 0; l < nlist; l++) {
			const int64_t len = list_off[(size_t)l + 1] - list_off[(size_t)l];
			pb[(size_t)l] = pos;
			pe[(size_t)l] = pos + len;
			pos += (len + 63) / 64 * 64;
		}
		nrows_mf = pos;
		std::vector<int32_t> perm((size_t)std::max<int64_t>(nrows_mf, 1), -1);
		for (int64_t l = 0; l < nlist; l++)
			for (int64_t j = 0; j < pe[(size_t)l] - pb[(size_t)l]; j++)
				perm[(size_t)(pb[(size_t)l] + j)] = (int32_t)(list_off[(size_t)l] + j);
		DevBuf dperm, tmp;
		dperm.reserve(perm.size() * sizeof(int32_t));
		tmp.reserve(std::max<size_t>((size_t)nrows_mf * d * sizeof(float), 16));
		if (need_f32) {
			codes_mf.reserve(((size_t)nrows_mf * geom.dp + 64) * sizeof(float));
			norms_mf.reserve(std::max<size_t>((size_t)nrows_mf * sizeof(float), 16));
		}
		rowids_mf.reserve(std::max<size_t>((size_t)nrows_mf * sizeof(int64_t), 16));
		lb_dev.reserve((size_t)nlist * sizeof(int64_t));
		le_dev.reserve((size_t)nlist * sizeof(int64_t));
		MVS_HIP(hipMemcpyAsync(dperm.p, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
		MVS_HIP(hipMemcpyAsync(lb_dev.p, pb.data(), (size_t)nlist * sizeof(int64_t), hipMemcpyHostToDevice, stream));
		MVS_HIP(hipMemcpyAsync(le_dev.p, pe.data(), (size_t)nlist * sizeof(int64_t), hipMemcpyHostToDevice, stream));
		if (nrows_mf > 0) {
			const long long tot = (long long)nrows_mf * d;
			hipLaunchKernelGGL(ivf_gather_plain_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
			                   (const float *)codes.p, dp, (const int *)dperm.p, (long long)nrows_mf, d, (float *)tmp.p);
			hipLaunchKernelGGL(ivf_gather_ids_kernel, dim3((unsigned)((nrows_mf + 255) / 256)), dim3(256), 0, stream,
			                   (const long long *)rowids.p, (const int *)dperm.p, (long long)nrows_mf,
			                   (long long *)rowids_mf.p);
			MVS_HIP(hipGetLastError());
			// L2: the MFMA list store holds RESIDUALS against the list centroid.  The kernel's (xn + yn) - 2<x,y> loses its
			// precision to cancellation when the norms dwarf the distance (the C3 mixture: norms ~130, distances ~2.5);
			// on residuals both are of the size of the distance, and ||(x-c) - (y-c)||^2 is the same distance.
			mf_residual = metric == METRIC_L2;
			// inner product: the f32 ITEMS scan reads the rows themselves -- packed before the residuals overwrite `tmp`
			if (need_f32 && !mf_residual) {
				launch_pack_rows(geom, (const float *)tmp.p, nrows_mf, (float *)codes_mf.p, 0, stream);
				launch_query_norms((const float *)tmp.p, nrows_mf, d, (float *)norms_mf.p, stream);
			}
			{
				std::vector<float> cent((size_t)nlist * d);
				get_centroids(cent.data());
				std::vector<int32_t> lob((size_t)(nrows_mf / 64 + 1), 0);
				for (int64_t l = 0; l < nlist; l++)
					for (int64_t b = pb[(size_t)l] / 64; b < (pb[(size_t)l] + (pe[(size_t)l] - pb[(size_t)l] + 63) / 64 * 64) / 64; b++)
						lob[(size_t)b] = (int32_t)l;
				cent_dev.reserve(cent.size() * sizeof(float));
				list_of_blk.reserve(lob.size() * sizeof(int32_t));
				MVS_HIP(hipMemcpyAsync(cent_dev.p, cent.data(), cent.size() * sizeof(float), hipMemcpyHostToDevice, stream));
				MVS_HIP(hipMemcpyAsync(list_of_blk.p, lob.data(), lob.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
				// Inner product: IndexIVF's spherical k-means keeps UNIT-norm centroids, so y - c_list is as long as y itself and the
				// coarse filter's bound (it scales with ||x|| max||y'||) admitted whole lists once the bf16 unit roundoff was
				// corrected (C3 shape: 12.6 ms per batch instead of 3.8).  Any vector may centre a list (<x, y> = <x, y'> + <x, m>):
				// the list's MEAN row is the tight one.  (L2: the centroid IS about the mean.)  Only the coarse filter reads cent_dev
				// for inner product.
				if (metric == METRIC_IP)
					hipLaunchKernelGGL(ivf_list_mean_kernel, dim3((unsigned)nlist), dim3(128), 0, stream, (const float *)tmp.p,
					                   (const long long *)lb_dev.p, (const long long *)le_dev.p, d, (float *)cent_dev.p);
				hipLaunchKernelGGL(ivf_residual_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, (float *)tmp.p,
				                   (const int *)dperm.p, (long long)nrows_mf, d, (const int *)list_of_blk.p, (const float *)cent_dev.p);
				have_bfr = d <= 128;
				if (have_bfr) { // the coarse filter's view of the residuals: bf16 rows, beta (L2: -||y'||^2, IP: 0), per-list max
					codes_bfr.reserve(((size_t)nrows_mf + 192) * 128 * sizeof(unsigned short));
					beta_mf.reserve(((size_t)nrows_mf + 192) * sizeof(float));
					list_max.reserve((size_t)2 * nlist * sizeof(unsigned)); // [nlist] largest ||y'||^2 | [nlist] largest ||y' - bf16(y')||^2
					MVS_HIP(hipMemsetAsync(codes_bfr.p, 0, ((size_t)nrows_mf + 192) * 128 * sizeof(unsigned short), stream));
					MVS_HIP(hipMemsetAsync(beta_mf.p, 0, ((size_t)nrows_mf + 192) * sizeof(float), stream));
					MVS_HIP(hipMemsetAsync(list_max.p, 0, (size_t)2 * nlist * sizeof(unsigned), stream));
					launch_ivf_rows_to_bf16((const float *)tmp.p, nrows_mf, d, (const int *)list_of_blk.p,
					                        (unsigned short *)codes_bfr.p, (float *)beta_mf.p, (unsigned *)list_max.p, nlist, stream);
					if (!mf_residual) // inner product: s = <x, y'> + <x, c>, no row term
						MVS_HIP(hipMemsetAsync(beta_mf.p, 0, ((size_t)nrows_mf + 192) * sizeof(float), stream));
				}
				MVS_HIP(hipStreamSynchronize(stream)); // cent / lob are host temporaries
			}
			if (need_f32 && mf_residual) {
				launch_pack_rows(geom, (const float *)tmp.p, nrows_mf, (float *)codes_mf.p, 0, stream);
				launch_query_norms((const float *)tmp.p, nrows_mf, d, (float *)norms_mf.p, stream);
			}
		}
		perm_mf.reserve(perm.size() * sizeof(int32_t));
		MVS_HIP(hipMemcpyAsync(perm_mf.p, dperm.p, perm.size() * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
		max_norm_mf.reserve(64);
		MVS_HIP(hipMemsetAsync(max_norm_mf.p, 0, 64, stream));
		if (nrows_mf > 0 && need_f32)
			hipLaunchKernelGGL(ivf_max_norm_kernel, dim3((unsigned)((nrows_mf + 255) / 256)), dim3(256), 0, stream,
			                   (const float *)norms_mf.p, (long long)nrows_mf, (unsigned *)max_norm_mf.p);
		MVS_HIP(hipStreamSynchronize(stream));
		mf_dirty = false;
	}

	// ---------------------------------------------------------------------------------------------- search
	// row shard of a ShardedIndex: A_k of the flagged queries of the last search (csrc/ivf_ties.hip EMIT mode)
	int64_t last_np = 0;
	int64_t last_coarse_nq = 0;           // the batch whose coarse assignment ws_cI holds (tie_emit reuses it)
	const float *last_coarse_x = nullptr;
	bool ids_ascending = true;            // every id added so far was larger than all before it (plain add(): always)
	int64_t last_id_seen = -1;
	const float *last_batch_ptr() const override {
		return last_coarse_x;
	}
	bool named_stat(const char *name, int64_t *v) override {
		if (!strcmp(name, "ivf_ids_ascending")) {
			*v = ids_ascending ? 1 : 0;
			return true;
		}
		return false;
	}
	void tie_emit(const int *d_flag, int nf, const float *d_x, const float *d_T, int64_t k, const mvs_search_params *params,
	              const int64_t *d_idmap_sel, float *d_v, int64_t *d_id, int *d_p, hipStream_t st) override {
		use_device();
		if (nf <= 0)
			return;
		// (the preconditions, checked -- the coarse assignment in ws_cI is that of the LAST search's batch)
		// (the C ABI entry also insists on the very pointer of that search -- mvs_index_ivf_tie_emit_device; the in-library sharded index
		// hands over its own copy of the same batch)
		if (nf > last_coarse_nq)
			throw_faiss("mvs::IVFFlatIndex::tie_emit", __FILE__, "%lld flagged queries, but the search that has just run on this index had %lld",
			            (long long)nf, (long long)last_coarse_nq);
		stream_wait(stream, st);
		{
			if (!h_fail)
				MVS_HIP(hipHostMalloc((void **)&h_fail, 512, hipHostMallocDefault));
			h_fail[120] = 0;
			hipLaunchKernelGGL(ivf_flag_check_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, stream, d_flag, nf, (int)last_coarse_nq, h_fail + 120);
			MVS_HIP(hipStreamSynchronize(stream));
			if (h_fail[120] != 0)
				throw_faiss("mvs::IVFFlatIndex::tie_emit", __FILE__, "a flagged query number lies outside the last search's batch of %lld queries",
				            (long long)last_coarse_nq);
		}
		if (ntotal == 0 || last_np <= 0) {
			MVS_HIP(hipMemsetAsync(d_id, 0xff, (size_t)nf * k * sizeof(int64_t), stream));
			MVS_HIP(hipMemsetAsync(d_p, 0xff, (size_t)nf * k * sizeof(int), stream));
			MVS_HIP(hipMemsetAsync(d_v, 0, (size_t)nf * k * sizeof(float), stream));
		} else {
			build_lists();
			SelectorDev tsel = selector.upload(params, stream);
			launch_ivf_tie_emit(metric, d_flag, nf, d_x, d, d_T, (int)k, (const int64_t *)ws_cI.p, (int)last_np, (const int64_t *)list_off_dev.p,
			                    (const float *)codes.p, dp, (const int64_t *)rowids.p, tsel, d_idmap_sel, d_v, d_id, d_p, stream);
		}
		stream_wait(st, stream);
	}
	void search_mapped(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   const int64_t *d_idmap, hipStream_t st) override {
		use_device();
		if (k <= 0)
			throw_faiss("virtual void faiss::IndexIVF::search(...) const", "faiss/IndexIVF.cpp", "Error: 'k > 0' failed");
		if (nq <= 0)
			return;
		int64_t np = params && params->nprobe > 0 ? params->nprobe : nprobe;
		np = std::min(np, nlist); // IndexIVF::search: nprobe = min(nlist, params->nprobe)
		if (np <= 0)
			throw_faiss("virtual void faiss::IndexIVF::search(...) const", "faiss/IndexIVF.cpp",
			            "Error: 'nprobe > 0' failed");
		TraceRange tr(raw_pos ? "mvs:ivf_search (inside the exact-tie wrapper)" : "mvs:ivf_search");
		last_np = np;
		// our stream carries the adds / list build; the caller's stream carries the queries
		stream_wait(stream, st);
		if (ntotal == 0) { // trained but empty: every heap stays at its neutral value
			const long long tot = nq * k;
			hipLaunchKernelGGL(ivf_fill_empty_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, d_D,
			                   (long long *)d_I, tot, metric == METRIC_L2 ? FLT_MAX : -FLT_MAX);
			MVS_HIP(hipGetLastError());
			memset(&kinfo, 0, sizeof kinfo);
			stream_wait(st, stream);
			return;
		}
		build_lists();
		// 1. coarse quantisation on the whole batch (FAISS slices the batch by OpenMP thread; see oracle ivf_search)
		ws_cD.reserve((size_t)nq * np * sizeof(float));
		ws_cI.reserve((size_t)nq * np * sizeof(int64_t));
		mvs_search_params qp;
		memset(&qp, 0, sizeof qp);
		qp.efSearch = params ? params->efSearch : 0; // quantizer_params of an HNSW coarse quantizer (:679-681)
		if (!reuse_coarse) { // (a prefilter re-run keeps the coarse assignment of the batch its queries came from)
			TraceRange trc("mvs:ivf_coarse_quantiser");
			last_coarse_nq = nq, last_coarse_x = d_x;
			// a Flat L2 quantizer of a few thousand centroids: distance matrix + per-query selection (csrc/coarse_select.hip)
			const bool done = hnsw_M == 0 && static_cast<FlatIndex *>(quantizer)->coarse_topk(nq, d_x, np, (float *)ws_cD.p,
			                                                                                  (int64_t *)ws_cI.p, stream, shadow != nullptr);
			if (!done)
				quantizer->search_device(nq, d_x, np, (float *)ws_cD.p, (int64_t *)ws_cI.p, hnsw_M > 0 ? &qp : nullptr, stream);
		}
		// Inner product: the MFMA variant is the same k-ordered chain as IVFFlatScanner's fvec_inner_product -> default.
		// L2: it evaluates ||x||^2 + ||y||^2 - 2<x,y> (the Flat BLAS-branch arithmetic) instead of the scanner's
		// sum (x-y)^2, i.e. the same neighbours up to rounding-level near-ties -> opt-in (option ivf_mfma = 1); the
		// default keeps the scanner's arithmetic bit for bit.
		// Exact distance ties (csrc/ivf_ties.hip): every path runs with ONE extra entry and emits (value, position in the
		// list-sorted store) in its pure order; the finish kernel prints equal values by stored id and flags the queries tied
		// at the k-th value, the tie pass replays FAISS's heap (arrival order = probe rank, then list position) for those.
		// (k so large that A_k does not fit the tie pass's LDS -- ~12 700 at d = 128 -- keeps the pure order, as k >= 16 384 always did)
		if (exact_ties && !raw_pos && !(metric == METRIC_L2 && (mfma_mode == 1 || mfma_mode == 2)) && k < ((int64_t)1 << 14) &&
		    ivf_tie_pass_fits(d, k)) {
			const int64_t kx = k + 1;
			ws_tD.reserve((size_t)nq * kx * sizeof(float));
			ws_tI.reserve((size_t)nq * kx * sizeof(int64_t));
			ws_tflag.reserve((size_t)(nq + 16) * sizeof(int));
			raw_pos = true;
			reuse_coarse = true; // (the coarse assignment above)
			const int64_t *idmap_out = (d_idmap && !raw_ids) ? d_idmap : nullptr;
			// (round 5: the bucket path of collect_search prints the final lists and runs the tie pass itself -- fin_done)
			fin_D = d_D, fin_I = d_I, fin_idmap = idmap_out, fin_k = k, fin_done = false;
			try {
				search_mapped(nq, d_x, kx, (float *)ws_tD.p, (int64_t *)ws_tI.p, params, d_idmap, stream);
			} catch (...) {
				raw_pos = reuse_coarse = false;
				fin_D = nullptr, fin_I = nullptr;
				throw;
			}
			raw_pos = reuse_coarse = false;
			fin_D = nullptr, fin_I = nullptr;
			if (fin_done) {
				fin_done = false;
				stream_wait(st, stream);
				return;
			}
			launch_ivf_finish(metric, (const float *)ws_tD.p, (const int64_t *)ws_tI.p, nq, (int)kx, (int)k, (const int64_t *)rowids.p,
			                  idmap_out, d_D, d_I, (int *)ws_tflag.p, stream);
			SelectorDev tsel = selector.upload(params, stream);
			launch_ivf_tie_pass(metric, (const int *)ws_tflag.p, nq, d_x, d, (const float *)ws_tD.p, (int)kx, (int)k,
			                    (const int64_t *)ws_cI.p, (int)np, (const int64_t *)list_off_dev.p, (const float *)codes.p, dp,
			                    (const int64_t *)rowids.p, tsel, d_idmap, idmap_out, d_D, d_I, stream);
			stream_wait(st, stream);
			return;
		}
		// (round 6) lists of 33 .. 2048 entries: the bf16 filter against a frozen bound (collect_search_big)
		if (cl_big && !force_select && (raw_pos && k > 1 ? k - 1 : k) > 32 && k <= 2049 && (metric == METRIC_L2 || metric == METRIC_IP) && collect_mode != 0 &&
		    mfma_mode < 0 && !pf_suppressed && !shadow && hnsw_M == 0 && d <= 128 && (dp == 32 || dp == 64 || dp == 128) && nq * np < ((int64_t)1 << 26) &&
		    nq * k < ((int64_t)1 << 31) && nsorted >= 16 * k && collect_search_big(nq, d_x, k, d_D, d_I, params, d_idmap, st, np))
			return;
		if (k > 256 || force_select) { // beyond the k-list kernels: all distances + segmented sort (csrc/ivf_select.hip)
			select_search(nq, d_x, k, d_D, d_I, params, d_idmap, st, np);
			return;
		}
		// default: bf16 coarse filter on residual rows + exact scanner-arithmetic re-scoring (csrc/ivf_collect.hip)
		if ((metric == METRIC_L2 || metric == METRIC_IP) && collect_mode != 0 && mfma_mode < 0 && !pf_suppressed && (raw_pos && k > 1 ? k - 1 : k) <= 32 && d <= 128 &&
		    dp % 4 == 0 && dp <= 128 && nq * np < ((int64_t)1 << 26)) { // (faster than the scanner kernel from one query on)
			if (collect_search(nq, d_x, k, d_D, d_I, params, d_idmap, st, np))
				return;
		}
		const bool want_mfma = mfma_mode == 1 || (mfma_mode < 0 && metric == METRIC_IP);
		const bool small = nq * np < (int64_t)1 << 26;
		// L2, default: the MFMA scan as a prefilter + exact scanner-arithmetic re-scoring with a per-query proof
		// (The scan runs on RESIDUALS against the list centroid: on the original vectors the formula's error scales with
		// ||x||^2 + ||y||^2 while the distance does not -- at the C3 mixture, norms ~130 and distances ~2.5, the proof
		// failed for most queries.  k + 4 candidates: with more the k-lists push the workgroup past half a CU's LDS.)
		const int64_t kp = k + 4;
		// Opt-in (ivf_mfma = 2): exact, but at C3 it lands at 780 k QPS against the scanner kernel's 805-817 k -- the ITEMS
		// launch takes 8.4 ms for k + 4 = 14 candidates (5.7 ms for inner product with k = 10), the re-runs 1.5 ms.
		if (metric == METRIC_L2 && mfma_mode == 2 && !pf_suppressed && small && kp <= 64 && nq >= 64 &&
		    flat_mfma_items_supported(flat_geom_for(d), kp)) {
			mfma_prefilter_search(nq, d_x, k, (int)kp, d_D, d_I, params, d_idmap, st, np);
			return;
		}
		if (want_mfma && small && flat_mfma_items_supported(flat_geom_for(d), k)) {
			mfma_grouped_search(nq, d_x, k, d_D, d_I, params, d_idmap, st, np);
			return;
		}
		const bool fast_scan = use_fast_scan && ivf_scan_supported(dp, k) && small;
		if (fast_scan)
			device_grouped_search(nq, d_x, k, d_D, d_I, params, d_idmap, st, np);
		else
			host_grouped_search(nq, d_x, k, d_D, d_I, params, d_idmap, st, np);
	}

	// list scan with the work items built ON DEVICE from the coarse labels: no host round trip inside a search
	void device_grouped_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                           const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st, int64_t np) {
		const int64_t npairs = nq * np;
		const int max_items = ivf_group_max_items(npairs, nlist, 20);
		ws_items.reserve((size_t)max_items * 16);
		ws_qidx.reserve((size_t)npairs * sizeof(int32_t));
		ws_slots.reserve((size_t)npairs * sizeof(int32_t));
		ws_group.reserve(ivf_group_ws_ints(nlist) * sizeof(int));
		group_clean_p = nullptr; // (the counters are left as this grouping made them)
		int *d_nitems = nullptr, *d_cnt = nullptr;
		launch_ivf_group((const int64_t *)ws_cI.p, nq, (int)np, nlist, 20, 5, (const int64_t *)list_off_dev.p,
		                 (const int64_t *)list_off_dev.p + 1, (int *)ws_group.p, ws_items.p, (int *)ws_qidx.p,
		                 (int *)ws_slots.p, &d_nitems, &d_cnt, stream);
		ws_q.reserve((size_t)nq * dp * sizeof(float));
		ws_pd.reserve((size_t)max_items * 20 * k * sizeof(float));
		ws_pi.reserve((size_t)max_items * 20 * k * sizeof(int32_t));
		ws_xi.reserve(ivf_scan_query_pack_bytes(dp, max_items));
		launch_pad_rows(d_x, nq, d, (float *)ws_q.p, dp, stream);
		SelectorDev sel = selector.upload(params, stream);
		memset(&kinfo, 0, sizeof kinfo);
		ws_gslot.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
		launch_init_slots((unsigned *)ws_gslot.p, nq, k, metric, stream);
		begin_kernel_timing(stream);
		launch_ivf_scan(dp, metric, (const float *)ws_q.p, (const float *)codes.p, nsorted, (const int64_t *)rowids.p, k,
		                ws_items.p, max_items, (const int *)ws_qidx.p, sel, d_idmap, (float *)ws_pd.p, (int32_t *)ws_pi.p,
		                (unsigned *)ws_gslot.p, (float *)ws_xi.p, d_nitems, stream);
		end_kernel_timing(stream);
		launch_merge_items(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, (const int *)ws_slots.p, (int)np, nq,
		                   k, raw_pos ? nullptr : (const int64_t *)rowids.p, (raw_ids || raw_pos) ? nullptr : d_idmap, d_D, d_I, stream);
		snprintf(kinfo.name, sizeof kinfo.name, "ivf_scan_kernel");
		kinfo.grid = max_items;
		kinfo.block = 256;
		kinfo.lds_bytes = (int)ivf_scan_lds_bytes(k);
		kinfo.nsplit = (int)np;
		if (timing_enabled) { // algorithmic bytes need the per-list pair counts: fetched only for a bench
			std::vector<int> cnt((size_t)nlist);
			MVS_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nlist * sizeof(int), hipMemcpyDeviceToHost, stream));
			MVS_HIP(hipStreamSynchronize(stream));
			double bytes = 0, pairs = 0;
			for (int64_t l = 0; l < nlist; l++) {
				const double len = (double)(list_off[(size_t)l + 1] - list_off[(size_t)l]);
				bytes += (double)((cnt[(size_t)l] + 19) / 20) * len * dp * 4.0;
				pairs += (double)cnt[(size_t)l] * len;
			}
			kinfo.bytes = bytes; // list-major algorithmic bytes: every item streams its list once
			kinfo.flops = pairs * d * (metric == METRIC_L2 ? 3.0 : 2.0);
		}
		stream_wait(st, stream);
	}

	// query fragments of the work items; L2: residuals against the item's list centroid + their norms per item slot
	const float *pack_item_queries(const float *d_x, int64_t nq, int max_items, const int *d_nitems) {
		ws_xi.reserve(flat_mfma_item_query_floats(geom, max_items) * sizeof(float));
		if (!mf_residual) {
			launch_ivf_pack_item_fragments(d_x, d, geom.kc, geom.nch, ws_items.p, d_nitems, max_items, (const int *)ws_qidx.p,
			                               (float *)ws_xi.p, stream);
			return nullptr;
		}
		ws_iqn.reserve((size_t)max_items * 128 * sizeof(float));
		ws_qmaxn.reserve((size_t)nq * sizeof(unsigned) + 64);
		MVS_HIP(hipMemsetAsync(ws_qmaxn.p, 0, (size_t)nq * sizeof(unsigned), stream));
		launch_ivf_pack_item_fragments_residual(d_x, d, geom.kc, geom.nch, ws_items.p, d_nitems, max_items,
		                                        (const int *)ws_qidx.p, (float *)ws_xi.p, (const float *)cent_dev.p,
		                                        (const int *)list_of_blk.p, (float *)ws_iqn.p, (unsigned *)ws_qmaxn.p, stream);
		return (const float *)ws_iqn.p;
	}

	// IVF list scan as a segmented variant of the fused Flat kernel: items of <= 128 queries per list
	void mfma_grouped_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                         const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st, int64_t np) {
		build_lists_mf();
		const int G = flat_mfma_item_slots(), shift = 7;
		const int64_t npairs = nq * np;
		const int max_items = ivf_group_max_items(npairs, nlist, G);
		ws_items.reserve((size_t)max_items * 16);
		ws_qidx.reserve((size_t)npairs * sizeof(int32_t));
		ws_slots.reserve((size_t)npairs * sizeof(int32_t));
		ws_group.reserve(ivf_group_ws_ints(nlist) * sizeof(int));
		group_clean_p = nullptr;
		int *d_nitems = nullptr, *d_cnt = nullptr;
		launch_ivf_group((const int64_t *)ws_cI.p, nq, (int)np, nlist, G, shift, (const int64_t *)lb_dev.p,
		                 (const int64_t *)le_dev.p, (int *)ws_group.p, ws_items.p, (int *)ws_qidx.p, (int *)ws_slots.p,
		                 &d_nitems, &d_cnt, stream);
		const float *item_qn = pack_item_queries(d_x, nq, max_items, d_nitems);
		ws_q.reserve((size_t)nq * sizeof(float)); // query norms (L2)
		launch_query_norms(d_x, nq, d, (float *)ws_q.p, stream);
		ws_pd.reserve((size_t)max_items * G * k * sizeof(float));
		ws_pi.reserve((size_t)max_items * G * k * sizeof(int32_t));
		SelectorDev sel = selector.upload(params, stream);
		memset(&kinfo, 0, sizeof kinfo);
		ws_gslot.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
		begin_kernel_timing(stream);
		launch_flat_mfma_items(geom, metric, (const float *)ws_xi.p, (const float *)ws_q.p, nq, (const float *)codes_mf.p,
		                       (const float *)norms_mf.p, nrows_mf, k, ws_items.p, d_nitems, max_items,
		                       (const int *)ws_qidx.p, (const int64_t *)rowids_mf.p, &sel, d_idmap, (float *)ws_pd.p,
		                       (int32_t *)ws_pi.p, (unsigned *)ws_gslot.p, stream, item_qn);
		end_kernel_timing(stream);
		launch_merge_items(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, (const int *)ws_slots.p, (int)np, nq,
		                   k, raw_pos ? nullptr : (const int64_t *)rowids_mf.p, (raw_ids || raw_pos) ? nullptr : d_idmap, d_D, d_I, stream,
		                   G, shift);
		if (raw_pos) // positions in the padded MFMA store -> the list-sorted store
			launch_ivf_mf_to_csr(d_I, nq * k, (const int *)perm_mf.p, stream);
		snprintf(kinfo.name, sizeof kinfo.name, "ivf_mfma_scan (flat_mfma_resident_kernel items)");
		kinfo.grid = max_items;
		kinfo.block = 256;
		kinfo.nsplit = (int)np;
		if (timing_enabled) {
			std::vector<int> cnt((size_t)nlist);
			MVS_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nlist * sizeof(int), hipMemcpyDeviceToHost, stream));
			MVS_HIP(hipStreamSynchronize(stream));
			double bytes = 0, pairs = 0;
			for (int64_t l = 0; l < nlist; l++) {
				const double len = (double)(list_off[(size_t)l + 1] - list_off[(size_t)l]);
				bytes += (double)((cnt[(size_t)l] + G - 1) / G) * len * geom.dp * 4.0;
				pairs += (double)cnt[(size_t)l] * len;
			}
			kinfo.bytes = bytes; // list-major algorithmic bytes: every item streams its list once
			kinfo.flops = pairs * d * 2.0;
		}
		stream_wait(st, stream);
	}

	// L2: ITEMS scan with kp = k + 6 candidates per query (every list streamed ONCE for <= 128 of its queries), exact
	// re-scoring in IVFFlatScanner's arithmetic, proof, re-run of the unproven queries on the scanner kernel
	// bf16 coarse filter (csrc/ivf_collect.hip).  false: the candidate stream overflowed (the caller uses the scanner kernel).
	bool collect_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                    const int64_t *d_idmap, hipStream_t st, int64_t np) {
		// Candidates go to per-query buckets and ONE finish chain behind the scan (csrc/collect_bucket.h): nothing to estimate; a bucket or
		// stream that proved too small (the counts are read at the search's one synchronisation) is grown and the pass repeated
		cl_bpitch_try = 0;
		cl_cap_try = 0;
		for (int attempt = 0; attempt < 6; ++attempt) {
			bool overflow = false;
			if (shadow && attempt > 0) // (the caller's fail list holds the abandoned attempt's entries)
				MVS_HIP(hipMemsetAsync(shadow->fail_cnt, 0, sizeof(int), stream));
			const bool ok = collect_search_pass(nq, d_x, k, d_D, d_I, params, d_idmap, st, np, &overflow);
			if (!overflow)
				return ok;
		}
		return false;
	}
	bool collect_search_pass(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                         const int64_t *d_idmap, hipStream_t st, int64_t np, bool *overflow) {
		build_lists_mf(false);
		if (!have_bfr)
			return false;
		// Inside the exact-tie wrapper k is the user's k + 1 (the extra entry only tells whether the k-th value is tied).  The FILTER
		// works with the user's k: a row tied with the k-th value passes any bound derived from k rows (the bound is on values), so the
		// (k + 1)-th entry of the selection is a tied row if there is one -- and k = 32 stays on the 32-class instance, k = 16 on the
		// 16-class one, instead of falling to the next instance or (k = 32) to the scanner kernel; the bound is the
		// k-th, not the (k + 1)-th, best class.
		const int G = 128, shift = 7, kk = (int)k, kf = raw_pos && k > 1 ? (int)k - 1 : (int)k;
		const int64_t npairs = nq * np;
		const int max_items = ivf_group_max_items(npairs, nlist, G);
		ws_items.reserve((size_t)max_items * 16);
		ws_qidx.reserve((size_t)npairs * sizeof(int32_t));
		ws_slots.reserve((size_t)npairs * sizeof(int32_t));
		// Both groupings (the nearest-list pre-pass and the main pass) in three launches (csrc/ivf_scan.hip launch_ivf_group2), both
		// packings in one (launch_ivf_collect_pack2, which also sets the class slots neutral and clears the control block's header), and
		// NO memset: the grouping counters are zeroed by the scatter kernel that last needed them, the per-query control words by the
		// select kernel that last read them -- a buffer is cleared by the host only when it was (re)allocated.
		const size_t group_ints = (ivf_group_ws_ints(nlist) + 63) & ~(size_t)63;
		ws_group.reserve(2 * group_ints * sizeof(int));
		if (ws_group.p != group_clean_p || ws_group.cap != group_clean_cap)
			MVS_HIP(hipMemsetAsync(ws_group.p, 0, 2 * group_ints * sizeof(int), stream));
		group_clean_p = nullptr;
		ws_xi.reserve(ivf_collect_xi_bytes(max_items));
		ws_ig.reserve((size_t)max_items * 128 * sizeof(float));
		ws_ie2.reserve((size_t)max_items * 128 * sizeof(float));
		// control block: header {stream count @0 | survivors @8 | unit count @16 | fail count @64 | forced drains @128 | bucket stats @192} |
		// per-query fail flags | hits / finished units per query | tie flags {count, queries}
		const size_t ctl_bytes = 256 + (size_t)3 * nq * sizeof(int) + ((size_t)nq + 64) * sizeof(int);
		ws_qfail.reserve(ctl_bytes);
		int *const ctl_qfail = (int *)((char *)ws_qfail.p + 256), *const ctl_seg = ctl_qfail + nq;
		int *const ctl_flag = ctl_seg + 2 * nq;
		unsigned long long *const ctl_stats = (unsigned long long *)((char *)ws_qfail.p + 192);
		const int nclass = kf > 16 ? 32 : 16; // row classes per query (ivf_bf16_collect_kernel<NC>)
		ws_gslot.reserve((size_t)nq * nclass * sizeof(unsigned) + 64);
		if (ws_qfail.p != ctl_clean_p || ws_qfail.cap != ctl_clean_cap || ctl_clean_nq != nq)
			MVS_HIP(hipMemsetAsync(ws_qfail.p, 0, ctl_bytes, stream));
		ctl_clean_p = nullptr;
		// candidate stream: 4096 entries per query to start with, or what the last overflow showed this index's data to need
		int64_t cap_entries = cl_stream_cap_per_query > 0 ? std::max<int64_t>(nq * cl_stream_cap_per_query, 1024) // (option ivf_cl_stream_cap: tests)
		                                                  : std::max<int64_t>(nq * std::max<int64_t>(4096, cl_cap_hint), (int64_t)1 << 20);
		if (cl_cap_try > 0)
			cap_entries = cl_cap_try; // (the repeated pass of a search whose stream overflowed)
		// buckets: cl_bpitch entries of 8 bytes per query (a multiple of 64; grown when a query's count exceeded it) + the unit list
		const int bpitch = cl_bpitch_try > 0 ? cl_bpitch_try
		                   : (int)(cl_stream_cap_per_query > 0 ? std::max<int64_t>(64, (cl_stream_cap_per_query + 63) / 64 * 64) : cl_bpitch);
		const size_t half = ((size_t)cap_entries * 8 + 255) & ~(size_t)255;
		ws_stream.reserve(256 + half + (size_t)nq * bpitch * 8); // the stream, then the buckets: [nq][bpitch] keys of 8 bytes
		unsigned long long *cnt = (unsigned long long *)ws_qfail.p; // (zeroed with the control block above)
		unsigned long long *strm = (unsigned long long *)((char *)ws_stream.p + 256);
		unsigned long long *bkeys = (unsigned long long *)((char *)ws_stream.p + 256 + half); // the buckets
		// final-bound filter: u per entry, Bf per query, the filtered stream; its count lives in the control block's header @8
		const bool refilter = cl_refilter;
		float *strm_u = nullptr, *bf_q = nullptr;
		unsigned long long *strm2 = nullptr, *cnt2 = (unsigned long long *)ws_qfail.p + 1;
		// (second cut: the survivors go to per-query ROW buckets and one wavefront per query re-scores them -- d = 128 only)
		const bool bexact = refilter && d == 128 && dp == 128;
		unsigned *brow = nullptr, *const bunit_cnt = (unsigned *)ws_qfail.p + 4; // (the unit count: control block header, byte 16)
		unsigned long long *bunits = nullptr;
		int *bkept = nullptr; // per-workgroup survivor counts of the scatter kernel (summed into cnt2 by the selection kernel)
		if (refilter) {
			const size_t ub = ((size_t)cap_entries * 4 + 255) & ~(size_t)255, bb = ((size_t)nq * 4 + 255) & ~(size_t)255;
			const size_t rb = bexact ? (((size_t)nq * bpitch * 4 + 255) & ~(size_t)255) : (size_t)cap_entries * 8;
			const size_t unb = bexact ? ((ivf_bucket_units_bytes(cap_entries) + 255) & ~(size_t)255) : 0;
			ws_stream2.reserve(ub + bb + rb + unb + (bexact ? (size_t)ivf_bucket_scatter_blocks(cap_entries) * 4 : 0) + 256);
			strm_u = (float *)ws_stream2.p;
			bf_q = (float *)((char *)ws_stream2.p + ub);
			strm2 = (unsigned long long *)((char *)ws_stream2.p + ub + bb);
			brow = bexact ? (unsigned *)strm2 : nullptr;
			bunits = bexact ? (unsigned long long *)((char *)ws_stream2.p + ub + bb + rb) : nullptr;
			bkept = bexact ? (int *)((char *)ws_stream2.p + ub + bb + rb + unb) : nullptr;
		}
		memset(&kinfo, 0, sizeof kinfo);
		// IDSelector: one bit per padded row, built per search (the selector sees the stored id, through the id map if any)
		const unsigned *rowmask = nullptr;
		if (params && params->sel_kind != MVS_SEL_NONE) {
			SelectorDev sel = selector.upload(params, stream);
			ws_rowmask.reserve(ivf_rowmask_bytes(nrows_mf));
			launch_ivf_rowmask(sel, (const int64_t *)rowids_mf.p, (const int *)perm_mf.p, d_idmap, nrows_mf, ws_rowmask.p, stream);
			rowmask = (const unsigned *)ws_rowmask.p;
		}
		// pre-pass: the first 256 rows of every query's nearest list, publish only (warms the query's bound); then every probed
		// list in segments of 512 rows, one wavefront per (work item, segment)
		int64_t max_list = 0;
		for (int64_t l = 0; l < nlist; l++)
			max_list = std::max(max_list, list_off[(size_t)l + 1] - list_off[(size_t)l]);
		const int seg_rows = cl_seg_rows, nseg = (int)((max_list + seg_rows - 1) / seg_rows);
		int *d_nitems = nullptr;
		// Round 5, probe pruning (csrc/ivf_collect.hip ivf_probe_prune_kernel): the grouping sees -1 for the probes that cannot matter
		const int64_t *probe_keys = (const int64_t *)ws_cI.p;
		const bool prune = cl_prune && metric == METRIC_L2 && hnsw_M == 0 && np <= 256 && np > 1 && !(params && params->sel_kind != MVS_SEL_NONE);
		if (prune) {
			ws_cIp.reserve((size_t)nq * np * sizeof(int64_t));
			ws_kept.reserve((size_t)nq * sizeof(int));
			launch_ivf_probe_prune(d_x, nq, d, (const float *)ws_cD.p, (const int64_t *)ws_cI.p, (int)np, kk,
			                       static_cast<FlatIndex *>(quantizer)->row_norms(), (const unsigned *)list_max.p,
			                       (const int64_t *)list_off_dev.p, (int64_t *)ws_cIp.p, (int *)ws_kept.p, stream);
			probe_keys = (const int64_t *)ws_cIp.p;
		}
		cl_last_pairs = npairs, cl_last_nq = nq;
		cl_pairs_pruned_pending = prune;
		if (!prune)
			cl_last_pairs_kept = npairs;
		{
			const int max_items0 = ivf_group_max_items(nq, nlist, G);
			ws_items0.reserve((size_t)max_items0 * 16);
			ws_qidx0.reserve((size_t)nq * sizeof(int32_t));
			int *d_nitems0 = nullptr;
			ws_xi0.reserve(ivf_collect_xi_bytes(max_items0));
			ws_ig0.reserve((size_t)max_items0 * 128 * sizeof(float));
			ws_ie20.reserve((size_t)max_items0 * 128 * sizeof(float));
			// (tried and dropped, profiles/r5_c3_ab.txt: the scan waves building their fragments / gamma / 2E themselves from the f32
			// queries instead of reading packed ones -- no packing kernel, 0.5 GB less traffic, but every segment wave of an item
			// repeats the item's conversion behind eight dependent load round trips: scan 0.77 -> 0.98 ms, step 1.58 -> 1.70)
			launch_ivf_group2(probe_keys, nq, (int)np, nlist, G, shift, (const int64_t *)lb_dev.p, (const int64_t *)le_dev.p,
			                  (int *)ws_group.p, (int *)ws_group.p + group_ints, ws_items0.p, (int *)ws_qidx0.p, (int *)ws_slots.p, ws_items.p,
			                  (int *)ws_qidx.p, nullptr, &d_nitems0, &d_nitems, stream);
			group_clean_p = ws_group.p, group_clean_cap = ws_group.cap; // (zero again behind the scatter kernel)
			launch_ivf_collect_pack2(metric, d_x, d, nq, (const int *)ws_slots.p, ws_items0.p, ws_xi0.p, (float *)ws_ig0.p, (float *)ws_ie20.p,
			                         ws_items.p, d_nitems, max_items, (const int *)ws_qidx.p, ws_xi.p, (float *)ws_ig.p, (float *)ws_ie2.p,
			                         (const float *)cent_dev.p, (const int *)list_of_blk.p, (const unsigned *)list_max.p, ctl_qfail, nlist,
			                         (unsigned *)ws_gslot.p, nclass, (int *)ws_qfail.p, ctl_flag, stream);
			launch_ivf_collect_scan(ws_items0.p, d_nitems0, max_items0, (const int *)ws_qidx0.p, ws_xi0.p, (const float *)ws_ig0.p,
			                        (const float *)ws_ie20.p, (const unsigned short *)codes_bfr.p, (const float *)beta_mf.p,
			                        (unsigned *)ws_gslot.p, strm, cnt, cap_entries, kf, cl_near_rows, 1, 0, rowmask, stream);
			begin_kernel_timing(stream);
			launch_ivf_collect_scan(ws_items.p, d_nitems, max_items, (const int *)ws_qidx.p, ws_xi.p, (const float *)ws_ig.p,
			                        (const float *)ws_ie2.p, (const unsigned short *)codes_bfr.p, (const float *)beta_mf.p,
			                        (unsigned *)ws_gslot.p, strm, cnt, cap_entries, kf, seg_rows, nseg, 1, rowmask, stream, strm_u);
			end_kernel_timing(stream);
		}
		// queries without a finite bound -> fail list (compacted by the select kernel)
		ws_fail.reserve(64 + (size_t)nq * sizeof(int));
		int *fail_cnt = (int *)((char *)ws_qfail.p + 64), *fail_q = (int *)ws_fail.p + 16; // (the count: in the zeroed control block)
		if (!h_fail)
			MVS_HIP(hipHostMalloc((void **)&h_fail, 512, hipHostMallocDefault)); // [0, 64): round 4's words; [256, 512): the control block's header
		fin_done = false;
		{
			// ONE finish chain: exact values, selection, and (inside the exact-tie wrapper) FAISS's print order + boundary flags; the tie pass
			// for the flagged queries is enqueued behind it, BEFORE the search's one synchronisation (rounds 3-4 launched the finish
			// kernel and the tie pass after it: two launch latencies with the GPU idle)
			const bool fin = raw_pos && fin_D != nullptr && kk == fin_k + 1 && !shadow;
			const unsigned long long *ex_strm = strm, *ex_cnt = cnt;
			if (bexact) {
				launch_ivf_bucket_scatter(strm, strm_u, cap_entries, cnt, (const unsigned *)ws_gslot.p, nclass, kf, nq, bf_q, brow,
				                          (unsigned *)ctl_seg, bpitch, bkept, bunits, bunit_cnt, stream);
			} else if (refilter) {
				launch_ivf_refilter(strm, strm_u, cap_entries, cnt, (const unsigned *)ws_gslot.p, nclass, kf, nq, bf_q, strm2, cnt2, stream);
				ex_strm = strm2, ex_cnt = cnt2;
			}
			if (shadow) {
				// Flat shadow: the candidates re-scored in the FLAT index's arithmetic, labels = Flat row numbers, the fail list is the
				// caller's; then the proof that no unprobed list matters (csrc/ivf_collect.hip ivf_shadow_verify_kernel)
				IvfFlatArith fa;
				fa.qn = shadow->qn, fa.yn = (const float *)norms_csr.p, fa.rowids = (const long long *)rowids.p;
				launch_ivf_bucket_finish(METRIC_L2, ex_strm, cap_entries, ex_cnt, bkeys, (unsigned *)ctl_seg, bpitch, nq, d_x, d, (const float *)codes.p,
				                         dp, (const int *)perm_mf.p, kk, d_D, d_I, nullptr, shadow->out_map, 0, nullptr, nullptr, nullptr, nullptr,
				                         nullptr, ctl_stats, ctl_qfail, shadow->fail_cnt, shadow->fail_q, true, stream, &fa, shadow->out_off, brow, 0, bunits, bunit_cnt, bkept,
				                         bkept ? (int)ivf_bucket_scatter_blocks(cap_entries) : 0, cnt2);
				FlatIndex *qz = static_cast<FlatIndex *>(quantizer);
				launch_ivf_shadow_verify(qz->coarse_matrix(), (const float *)ws_cD.p, probe_keys, nq, (int)nlist, (int)np, d, kk,
				                         shadow->qn, qz->row_norms(), (const unsigned *)list_max.p, (const int64_t *)lb_dev.p,
				                         (const int64_t *)le_dev.p, d_D, d_I, shadow->ymax_bits, shadow->fail_cnt, shadow->fail_q, stream);
			} else
			launch_ivf_bucket_finish(metric, ex_strm, cap_entries, ex_cnt, bkeys, (unsigned *)ctl_seg, bpitch, nq, d_x, d,
			                         (const float *)codes.p, dp, (const int *)perm_mf.p, kk, d_D, d_I, raw_pos ? nullptr : (const int64_t *)rowids.p,
			                         (d_idmap && !raw_ids && !raw_pos) ? d_idmap : nullptr, fin ? (int)fin_k : 0, fin ? fin_D : nullptr,
			                         fin ? fin_I : nullptr, (const int64_t *)rowids.p, fin ? fin_idmap : nullptr, fin ? ctl_flag : nullptr, ctl_stats,
			                         ctl_qfail, fail_cnt, fail_q, true, stream, nullptr, 0, brow, 0, bunits, bunit_cnt, bkept,
			                         bkept ? (int)ivf_bucket_scatter_blocks(cap_entries) : 0, cnt2);
			ctl_clean_p = ws_qfail.p, ctl_clean_cap = ws_qfail.cap, ctl_clean_nq = nq;
			if (fin) {
				SelectorDev tsel = selector.upload(params, stream);
				launch_ivf_tie_pass(metric, ctl_flag, nq, d_x, d, d_D, (int)kk, (int)fin_k, (const int64_t *)ws_cI.p, (int)np,
				                    (const int64_t *)list_off_dev.p, (const float *)codes.p, dp, (const int64_t *)rowids.p, tsel, d_idmap,
				                    fin_idmap, fin_D, fin_I, stream);
			}
			// (ONE copy of the control block's 256-byte header: stream count @0, fail count @64, bucket statistics @192)
			MVS_HIP(hipMemcpyAsync(h_fail + 64, ws_qfail.p, 256, hipMemcpyDeviceToHost, stream));
			snprintf(kinfo.name, sizeof kinfo.name, "ivf_bf16_collect_kernel");
			kinfo.grid = max_items * nseg;
			kinfo.block = 64;
			kinfo.nsplit = (int)np;
			kinfo.bytes = (double)nrows_mf * 256.0;               // every list's bf16 rows once (each list is probed by >= 1 item)
			kinfo.flops = (double)nq * np * ((double)nsorted / nlist) * d * 2.0; // (average list length)
			MVS_HIP(hipStreamSynchronize(stream)); // the one host round trip of the search
			unsigned long long st2[2], nstream = 0;
			memcpy(st2, (const char *)(h_fail + 64) + 192, sizeof st2);
			memcpy(&nstream, h_fail + 64, sizeof nstream);
			h_fail[0] = h_fail[64 + 16]; // (the fail count, where the code below reads it)
			cl_last_bursts = (int64_t)(unsigned)h_fail[64 + 32]; // mid-tile drains of a scan wave's hit queue (header byte 128)
			if ((int64_t)nstream > cap_entries) { // the STREAM overflowed (duplicate-heavy lists): grown once per size, as in round 3
				++cl_overflows;
				if ((int64_t)(nstream + nstream / 8) > nq * (int64_t)16384)
					return false;
				if (cl_stream_cap_per_query <= 0)
					cl_cap_hint = std::max<int64_t>(cl_cap_hint, ((int64_t)(nstream + nstream / 8) + nq - 1) / nq);
				cl_cap_try = (int64_t)(nstream + nstream / 8);
				*overflow = true;
				return false;
			}
			if ((int64_t)st2[1] > bpitch) { // some query's bucket was too small: its result is incomplete
				++cl_overflows;
				const int64_t want = ((int64_t)st2[1] + (int64_t)st2[1] / 4 + 63) / 64 * 64;
				if (want > 16384)
					return false; // (duplicate-heavy lists: the scanner kernel takes the batch, as before)
				cl_bpitch_try = (int)want;
				if (cl_stream_cap_per_query <= 0) // (option ivf_cl_stream_cap: tests -- the index does not remember the size)
					cl_bpitch = (int)want;
				*overflow = true;
				return false;
			}
			unsigned long long nkept = nstream;
			if (refilter)
				memcpy(&nkept, (const char *)(h_fail + 64) + 8, sizeof nkept);
			cl_last_admitted = (int64_t)nstream;
			cl_queries_total += nq;
			cl_candidates_total += (int64_t)nkept; // (what the exact stage re-scored)
			fin_done = fin;
			if (shadow) { // (the fail list is the caller's: it re-runs those queries on the Flat kernels)
				stream_wait(st, stream);
				return true;
			}
		}
		const int nf = *h_fail;
		pf_queries_total += nq;
		pf_fallback_total += nf;
		if (nf > 0) { // re-run on the scanner kernel with the same coarse assignment
			fin_done = false; // (their rows of the pure lists change below: the wrapper prints the batch again)
			const mvs_kernel_info keep = kinfo;
			const size_t xf_bytes = ((size_t)nf * d * sizeof(float) + 255) & ~(size_t)255;
			const size_t df_bytes = ((size_t)nf * k * sizeof(float) + 255) & ~(size_t)255;
			ws_fb.reserve(xf_bytes + df_bytes + (size_t)nf * k * sizeof(int64_t));
			float *xf = (float *)ws_fb.p;
			float *Df = (float *)((char *)ws_fb.p + xf_bytes);
			int64_t *If = (int64_t *)((char *)Df + df_bytes);
			launch_gather_query_rows(d_x, d, fail_q, nf, xf, stream);
			DevBuf csub, csave;
			csub.reserve((size_t)nf * np * sizeof(int64_t));
			csave.reserve((size_t)nf * np * sizeof(int64_t)); // the rows of ws_cI overwritten below (the tie pass reads them later)
			launch_gather_query_rows((const float *)ws_cI.p, (int)(2 * np), fail_q, nf, (float *)csub.p, stream);
			MVS_HIP(hipMemcpyAsync(csave.p, ws_cI.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			MVS_HIP(hipMemcpyAsync(ws_cI.p, csub.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			pf_suppressed = true;
			reuse_coarse = true;
			const bool timing = timing_enabled;
			timing_enabled = false;
			try {
				search_mapped(nf, xf, k, Df, If, params, d_idmap, stream);
			} catch (...) {
				pf_suppressed = reuse_coarse = false;
				timing_enabled = timing;
				throw;
			}
			pf_suppressed = false;
			reuse_coarse = raw_pos; // (inside the exact-tie wrapper the batch's coarse assignment stays in force)
			timing_enabled = timing;
			MVS_HIP(hipMemcpyAsync(ws_cI.p, csave.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			MVS_HIP(hipStreamSynchronize(stream)); // csub / csave are freed at scope exit
			launch_scatter_rows(fail_q, nf, k, Df, If, d_D, d_I, stream);
			kinfo = keep;
		}
		stream_wait(st, stream);
		return true;
	}

	void mfma_prefilter_search(int64_t nq, const float *d_x, int64_t k, int kp, float *d_D, int64_t *d_I,
	                           const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st, int64_t np) {
		build_lists_mf();
		const int G = flat_mfma_item_slots(), shift = 7;
		const int64_t npairs = nq * np;
		const int max_items = ivf_group_max_items(npairs, nlist, G);
		ws_items.reserve((size_t)max_items * 16);
		ws_qidx.reserve((size_t)npairs * sizeof(int32_t));
		ws_slots.reserve((size_t)npairs * sizeof(int32_t));
		ws_group.reserve(ivf_group_ws_ints(nlist) * sizeof(int));
		group_clean_p = nullptr;
		int *d_nitems = nullptr, *d_cnt = nullptr;
		launch_ivf_group((const int64_t *)ws_cI.p, nq, (int)np, nlist, G, shift, (const int64_t *)lb_dev.p,
		                 (const int64_t *)le_dev.p, (int *)ws_group.p, ws_items.p, (int *)ws_qidx.p, (int *)ws_slots.p,
		                 &d_nitems, &d_cnt, stream);
		const float *item_qn = pack_item_queries(d_x, nq, max_items, d_nitems);
		ws_q.reserve((size_t)nq * sizeof(float));
		launch_query_norms(d_x, nq, d, (float *)ws_q.p, stream);
		ws_pd.reserve((size_t)max_items * G * kp * sizeof(float));
		ws_pi.reserve((size_t)max_items * G * kp * sizeof(int32_t));
		SelectorDev sel = selector.upload(params, stream);
		memset(&kinfo, 0, sizeof kinfo);
		ws_gslot.reserve((size_t)nq * ((kp + 15) / 16 * 16) * sizeof(unsigned) + 64);
		begin_kernel_timing(stream);
		launch_flat_mfma_items(geom, metric, (const float *)ws_xi.p, (const float *)ws_q.p, nq, (const float *)codes_mf.p,
		                       (const float *)norms_mf.p, nrows_mf, kp, ws_items.p, d_nitems, max_items,
		                       (const int *)ws_qidx.p, (const int64_t *)rowids_mf.p, &sel, d_idmap, (float *)ws_pd.p,
		                       (int32_t *)ws_pi.p, (unsigned *)ws_gslot.p, stream, item_qn);
		end_kernel_timing(stream);
		// merged top-kp per query: formula values + row POSITIONS in the MFMA list store
		const size_t ca_bytes = ((size_t)nq * kp * sizeof(float) + 255) & ~(size_t)255;
		ws_cand.reserve(ca_bytes + (size_t)nq * kp * sizeof(int64_t));
		float *ca = (float *)ws_cand.p;
		int64_t *ci = (int64_t *)((char *)ws_cand.p + ca_bytes);
		launch_merge_items(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, (const int *)ws_slots.p, (int)np, nq, kp,
		                   nullptr, nullptr, ca, ci, stream, G, shift);
		const size_t ex_bytes = ((size_t)nq * kp * sizeof(float) + 255) & ~(size_t)255;
		ws_ex.reserve(ex_bytes + (size_t)nq * kp * sizeof(int32_t));
		float *pd1 = (float *)ws_ex.p;
		int32_t *pi1 = (int32_t *)((char *)ws_ex.p + ex_bytes);
		ws_fail.reserve(64 + (size_t)nq * sizeof(int));
		int *fail_cnt = (int *)ws_fail.p, *fail_q = fail_cnt + 16;
		MVS_HIP(hipMemsetAsync(fail_cnt, 0, sizeof(int), stream));
		hipLaunchKernelGGL(ivf_rescore_verify_kernel, dim3((unsigned)nq), dim3(64), 0, stream, ca, (const long long *)ci, kp,
		                   (int)k, d_x, d, (const float *)codes.p, dp, (const int *)perm_mf.p, (const long long *)ws_cI.p, (int)np,
		                   (const float *)cent_dev.p, (const int *)list_of_blk.p, (const unsigned *)max_norm_mf.p, pd1, pi1,
		                   fail_cnt, fail_q);
		MVS_HIP(hipGetLastError());
		// exact values -> the k best by (value, position), labels = stored ids (then the id map of an IDMap wrapper)
		launch_merge_partials(metric, pd1, pi1, 1, nq, kp, (const int64_t *)rowids_mf.p, 0, d_D, d_I, stream, k, nullptr);
		if (d_idmap && !raw_ids) {
			const long long tot = (long long)nq * k;
			hipLaunchKernelGGL(ivf_map_labels_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
			                   (long long *)d_I, tot, (const long long *)d_idmap);
		}
		if (!h_fail)
			MVS_HIP(hipHostMalloc((void **)&h_fail, 64, hipHostMallocDefault));
		MVS_HIP(hipMemcpyAsync(h_fail, fail_cnt, sizeof(int), hipMemcpyDeviceToHost, stream));
		snprintf(kinfo.name, sizeof kinfo.name, "ivf_mfma_prefilter (flat_mfma_resident_kernel items)");
		kinfo.grid = max_items;
		kinfo.block = 256;
		kinfo.nsplit = (int)np;
		std::vector<int> cnt;
		if (timing_enabled) {
			cnt.resize((size_t)nlist);
			MVS_HIP(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nlist * sizeof(int), hipMemcpyDeviceToHost, stream));
		}
		MVS_HIP(hipStreamSynchronize(stream));
		if (timing_enabled) {
			double bytes = 0, pairs = 0;
			for (int64_t l = 0; l < nlist; l++) {
				const double len = (double)(list_off[(size_t)l + 1] - list_off[(size_t)l]);
				bytes += (double)((cnt[(size_t)l] + G - 1) / G) * len * geom.dp * 4.0;
				pairs += (double)cnt[(size_t)l] * len;
			}
			kinfo.bytes = bytes;
			kinfo.flops = pairs * d * 2.0;
		}
		const int nf = *h_fail;
		pf_queries_total += nq;
		pf_fallback_total += nf;
		if (nf > 0) {
			const mvs_kernel_info keep = kinfo;
			const size_t xf_bytes = ((size_t)nf * d * sizeof(float) + 255) & ~(size_t)255;
			const size_t df_bytes = ((size_t)nf * k * sizeof(float) + 255) & ~(size_t)255;
			ws_fb.reserve(xf_bytes + df_bytes + (size_t)nf * k * sizeof(int64_t));
			float *xf = (float *)ws_fb.p;
			float *Df = (float *)((char *)ws_fb.p + xf_bytes);
			int64_t *If = (int64_t *)((char *)Df + df_bytes);
			launch_gather_query_rows(d_x, d, fail_q, nf, xf, stream);
			// their probe lists, compacted to the front of the coarse-label buffer (np int64 = 2 np floats per query)
			DevBuf csub, csave;
			csub.reserve((size_t)nf * np * sizeof(int64_t));
			csave.reserve((size_t)nf * np * sizeof(int64_t)); // the rows of ws_cI overwritten below: a row shard's tie pass (tie_emit) reads the
			                                                  // batch's whole coarse assignment after the search
			launch_gather_query_rows((const float *)ws_cI.p, (int)(2 * np), fail_q, nf, (float *)csub.p, stream);
			MVS_HIP(hipMemcpyAsync(csave.p, ws_cI.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			MVS_HIP(hipMemcpyAsync(ws_cI.p, csub.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			pf_suppressed = true;
			reuse_coarse = true;
			const bool timing = timing_enabled;
			timing_enabled = false;
			try {
				search_mapped(nf, xf, k, Df, If, params, d_idmap, stream);
			} catch (...) {
				pf_suppressed = reuse_coarse = false;
				timing_enabled = timing;
				throw;
			}
			pf_suppressed = reuse_coarse = false;
			timing_enabled = timing;
			MVS_HIP(hipMemcpyAsync(ws_cI.p, csave.p, (size_t)nf * np * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
			MVS_HIP(hipStreamSynchronize(stream)); // csub / csave are freed at scope exit
			launch_scatter_rows(fail_q, nf, k, Df, If, d_D, d_I, stream);
			kinfo = keep;
		}
		stream_wait(st, stream);
	}

	void host_grouped_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I,
	                         const mvs_search_params *params, const int64_t *d_idmap, hipStream_t st, int64_t np) {
		std::vector<int64_t> keys((size_t)nq * np);
		MVS_HIP(hipMemcpyAsync(keys.data(), ws_cI.p, keys.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
		MVS_HIP(hipStreamSynchronize(stream));
		// 2. list-major work items: every probed list x groups of <= 20 of the queries that probe it
		std::vector<int32_t> cnt((size_t)nlist + 1, 0);
		for (int64_t key : keys)
			if (key >= 0)
				cnt[(size_t)key + 1]++;
		for (int64_t l = 0; l < nlist; l++)
			cnt[(size_t)l + 1] += cnt[(size_t)l];
		const int32_t npairs = cnt[(size_t)nlist];
		std::vector<int32_t> qidx((size_t)std::max(npairs, 1)), cur(cnt.begin(), cnt.end() - 1);
		std::vector<int32_t> pair_slot((size_t)nq * np, -1); // (q, probe) -> position in qidx
		for (int64_t q = 0; q < nq; q++)
			for (int64_t p = 0; p < np; p++) {
				const int64_t key = keys[(size_t)(q * np + p)];
				if (key < 0)
					continue; // fewer than nprobe centroids
				const int32_t pos = cur[(size_t)key]++;
				qidx[(size_t)pos] = (int32_t)q;
				pair_slot[(size_t)(q * np + p)] = pos;
			}
		struct Item {
			int32_t row_begin, row_end, qoff, nq;
		};
		std::vector<Item> items;
		std::vector<int32_t> item_of_pos((size_t)std::max(npairs, 1));
		for (int64_t l = 0; l < nlist; l++) {
			const int32_t a = cnt[(size_t)l], b = cnt[(size_t)l + 1];
			for (int32_t g = a; g < b; g += 20) {
				const int32_t ng = std::min(20, b - g);
				for (int32_t t = 0; t < ng; t++)
					item_of_pos[(size_t)(g + t)] = (int32_t)items.size();
				items.push_back({(int32_t)list_off[(size_t)l], (int32_t)list_off[(size_t)l + 1], g, ng});
			}
		}
		std::vector<int32_t> slots((size_t)nq * np, -1);
		for (size_t i = 0; i < slots.size(); i++) {
			const int32_t pos = pair_slot[i];
			if (pos >= 0) {
				const int32_t it = item_of_pos[(size_t)pos];
				slots[i] = (it << 5) | (pos - items[(size_t)it].qoff);
			}
		}
		const int nitems = (int)items.size();
		// 3. upload, scan, merge
		ws_items.reserve(std::max<size_t>(items.size() * sizeof(Item), 16));
		ws_qidx.reserve(qidx.size() * sizeof(int32_t));
		ws_slots.reserve(slots.size() * sizeof(int32_t));
		ws_q.reserve((size_t)nq * dp * sizeof(float));
		ws_pd.reserve(std::max<size_t>((size_t)nitems * 20 * k * sizeof(float), 16));
		ws_pi.reserve(std::max<size_t>((size_t)nitems * 20 * k * sizeof(int32_t), 16));
		if (nitems > 0) {
			MVS_HIP(hipMemcpyAsync(ws_items.p, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice, stream));
			MVS_HIP(hipMemcpyAsync(ws_qidx.p, qidx.data(), qidx.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
		}
		MVS_HIP(hipMemcpyAsync(ws_slots.p, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
		launch_pad_rows(d_x, nq, d, (float *)ws_q.p, dp, stream);
		SelectorDev sel = selector.upload(params, stream);
		memset(&kinfo, 0, sizeof kinfo);
		ws_gslot.reserve((size_t)nq * ((k + 15) / 16 * 16) * sizeof(unsigned) + 64);
		launch_init_slots((unsigned *)ws_gslot.p, nq, k, metric, stream);
		begin_kernel_timing(stream);
		const bool fast_scan = false;
			launch_direct_items(dp, metric, (const float *)ws_q.p, nq, (const float *)codes.p, nsorted,
			                    (const int64_t *)rowids.p, k, ws_items.p, nitems, (const int *)ws_qidx.p, sel, d_idmap,
			                    (float *)ws_pd.p, (int32_t *)ws_pi.p, (unsigned *)ws_gslot.p, stream);
		end_kernel_timing(stream);
		launch_merge_items(metric, (const float *)ws_pd.p, (const int32_t *)ws_pi.p, (const int *)ws_slots.p, (int)np, nq,
		                   k, raw_pos ? nullptr : (const int64_t *)rowids.p, (raw_ids || raw_pos) ? nullptr : d_idmap, d_D, d_I, stream);
		// the caller's stream continues after ours; pageable staging vectors die with this frame
		MVS_HIP(hipStreamSynchronize(stream));
		stream_wait(st, stream);
		snprintf(kinfo.name, sizeof kinfo.name, fast_scan ? "ivf_scan_kernel" : "ivf_list_scan (flat_direct_kernel items)");
		double bytes = 0, pairs = 0;
		for (const Item &it : items) {
			bytes += (double)(it.row_end - it.row_begin) * dp * 4.0;
			pairs += (double)(it.row_end - it.row_begin) * it.nq;
		}
		kinfo.bytes = bytes;                 // list-major algorithmic bytes: every item streams its list once
		kinfo.flops = pairs * d * (metric == METRIC_L2 ? 3.0 : 2.0);
		kinfo.grid = nitems;
		kinfo.block = 256;
		kinfo.lds_bytes = (int)(fast_scan ? ivf_scan_lds_bytes(k) : direct_items_lds_bytes(dp, k));
		kinfo.nsplit = (int)np;
	}
	// k > 256 (the harness's post-filter runs ask for ~2 000 rows, go/main_test.go:17-45): every probed list's distances as
	// sortable keys, one segmented sort per chunk of queries, first k decoded.  The batch is cut so that one chunk holds at
	// most 2^26 candidates (1 GiB of keys, double-buffered).
	// (np_stride > np: only the np nearest of the np_stride lists the coarse quantiser assigned -- collect_search_big's bound)
	void select_search(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   const int64_t *d_idmap, hipStream_t st, int64_t np, int64_t np_stride = 0) {
		if (np_stride <= 0)
			np_stride = np;
		std::vector<int64_t> cl_all((size_t)nq * np_stride);
		MVS_HIP(hipMemcpyAsync(cl_all.data(), ws_cI.p, cl_all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
		MVS_HIP(hipStreamSynchronize(stream));
		std::vector<int64_t> cl((size_t)nq * np);
		for (int64_t q = 0; q < nq; ++q)
			for (int64_t p = 0; p < np; ++p)
				cl[(size_t)(q * np + p)] = cl_all[(size_t)(q * np_stride + p)];
		ws_q.reserve((size_t)nq * dp * sizeof(float));
		launch_pad_rows(d_x, nq, d, (float *)ws_q.p, dp, stream);
		SelectorDev sel = selector.upload(params, stream);
		memset(&kinfo, 0, sizeof kinfo);
		const int64_t budget = (int64_t)1 << 26;
		std::vector<IvfSelectPair> pairs;
		std::vector<int> seg;
		double bytes = 0, cands = 0;
		int64_t q0 = 0;
		while (q0 < nq) {
			pairs.clear();
			seg.assign(1, 0);
			int64_t total = 0, q1 = q0;
			for (; q1 < nq; ++q1) {
				int64_t tq = 0;
				for (int64_t p = 0; p < np; ++p) {
					const int64_t l = cl[(size_t)(q1 * np + p)];
					if (l >= 0)
						tq += list_off[(size_t)l + 1] - list_off[(size_t)l];
				}
				if (q1 > q0 && total + tq > budget)
					break;
				if (tq >= ((int64_t)1 << 31) - total)
					throw_faiss("mvs::IVFFlatIndex::search", __FILE__, "one query probes %lld rows: too many for k = %lld",
					            (long long)tq, (long long)k);
				for (int64_t p = 0; p < np; ++p) {
					const int64_t l = cl[(size_t)(q1 * np + p)];
					if (l < 0)
						continue; // fewer than nprobe centroids
					const int64_t len = list_off[(size_t)l + 1] - list_off[(size_t)l];
					if (len > 0)
						pairs.push_back({total, (int)list_off[(size_t)l], (int)len, (int)q1, 0});
					total += len;
				}
				seg.push_back((int)total);
			}
			const int64_t nseg = q1 - q0;
			ws_items.reserve(std::max<size_t>(pairs.size() * sizeof(IvfSelectPair), 16));
			ws_slots.reserve(seg.size() * sizeof(int));
			ws_pd.reserve(std::max<size_t>((size_t)total * 8, 16));
			ws_pi.reserve(std::max<size_t>((size_t)total * 8, 16));
			const size_t temp = total > 0 ? ivf_select_temp_bytes(total, nseg) : 0;
			ws_xi.reserve(std::max<size_t>(temp, 16));
			if (!pairs.empty())
				MVS_HIP(hipMemcpyAsync(ws_items.p, pairs.data(), pairs.size() * sizeof(IvfSelectPair), hipMemcpyHostToDevice,
				                       stream));
			MVS_HIP(hipMemcpyAsync(ws_slots.p, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, stream));
			begin_kernel_timing(stream);
			launch_ivf_select(metric, (const float *)ws_q.p, dp, (const float *)codes.p, (const int64_t *)rowids.p,
			                  (const IvfSelectPair *)ws_items.p, (int)pairs.size(), (const int *)ws_slots.p, nseg, total, k, sel,
			                  d_idmap, (raw_ids || raw_pos) ? nullptr : d_idmap, (unsigned long long *)ws_pd.p,
			                  (unsigned long long *)ws_pi.p, ws_xi.p, temp, d_D + q0 * k, d_I + q0 * k, stream, raw_pos);
			end_kernel_timing(stream);
			MVS_HIP(hipStreamSynchronize(stream)); // the pageable staging vectors are reused by the next chunk
			bytes += (double)total * dp * 4.0;
			cands += (double)total;
			q0 = q1;
		}
		stream_wait(st, stream);
		snprintf(kinfo.name, sizeof kinfo.name, "ivf_select (all distances + segmented sort)");
		kinfo.bytes = bytes + cands * 8.0 * 2.0 * 9.0; // list rows once per probing query + 8 radix passes + the key write
		kinfo.flops = cands * d * (metric == METRIC_L2 ? 3.0 : 2.0);
		kinfo.block = 256;
		kinfo.nsplit = (int)np;
	}
	// Lists beyond 32 entries on the bf16 filter (round 6).  The class slots of the scan reach k = 32; past that the scanner kernel
	// (k <= 256) and the all-distances + sort path served: 6-41 ms at C3's shape against 1 ms at k = 32 (tools/ivf_k_bench.py).  As for the
	// Flat index (csrc/flat_collect.hip "lists beyond 128 entries") the bound comes from elsewhere and the scan runs against it FROZEN:
	//   A. B(q) = the k-th best EXACT value among the rows of the query's nearest lists (select_search over the first np_a probes: a few
	//      thousand rows per query) -- k real rows are at least that good, so every row of the result has s >= B - E(list);
	//   B. the grouped bf16 scan of all probed lists against B (ivf_bf16_collect_kernel, a.bfix), candidates into the stream;
	//   C. positions in the list-sorted store, grouped by query, re-scored with the scanner's arithmetic (fvec_L2sqr / fvec_inner_product
	//      = the Flat per-pair chains: collect_exact_kernel), the k best by (value, position) -- the order of the k-list kernels, of
	//      select_search and of the exact-tie wrapper's contract.
	// false: not served / the stream overflowed / a query without a finite bound -- the caller's older paths take the batch.
	bool collect_search_big(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                        const int64_t *d_idmap, hipStream_t st, int64_t np) {
		build_lists_mf(false);
		if (!have_bfr || nsorted <= 0)
			return false;
		const int kk = (int)k, kf = raw_pos && k > 1 ? (int)k - 1 : (int)k;
		const int G = 128, shift = 7;
		// A. the bound: the nearest lists that hold ~3 k rows on average
		const double avg_len = (double)nsorted / (double)nlist;
		const int64_t np_a = std::min<int64_t>(np, std::max<int64_t>(1, (int64_t)(3.0 * kf / std::max(avg_len, 1.0)) + 1));
		ws_bfix.reserve((size_t)nq * sizeof(float));
		if (cl_big >= 1 && cl_big != 2) {
			// (default) on the device: the first R = 4 k (256 .. 8192) rows of the nearest lists, no host round trip
			const int R = (int)std::min<int64_t>(8192, std::max<int64_t>(256, 4 * (int64_t)kf));
			SelectorDev bsel = selector.upload(params, stream);
			const size_t blds = (size_t)(2 * R + 8) * sizeof(unsigned) + (size_t)256 * 33 * sizeof(float);
			ensure_dynamic_lds((const void *)ivf_big_bound_direct_kernel, blds);
			hipLaunchKernelGGL(ivf_big_bound_direct_kernel, dim3((unsigned)nq), dim3(256), blds, stream, d_x, d, dp,
			                   (const float *)codes.p, (const long long *)ws_cI.p, (int)np, (const long long *)list_off_dev.p, R, kf,
			                   metric == METRIC_L2 ? 1 : 0, bsel, (const long long *)rowids.p, (const long long *)d_idmap, (float *)ws_bfix.p);
			MVS_HIP(hipGetLastError());
		} else {
		// (option ivf_cl_big = 2: the bound from ALL rows of the nearest lists through the select path -- tighter, two host round trips)
		ws_bD.reserve((size_t)nq * kf * sizeof(float));
		ws_bI.reserve((size_t)nq * kf * sizeof(int64_t));
		{
			const bool rp = raw_pos, ri = raw_ids;
			raw_pos = true; // (positions: no label translation in the bound pass)
			try {
				select_search(nq, d_x, kf, (float *)ws_bD.p, (int64_t *)ws_bI.p, params, d_idmap, stream, np_a, np);
			} catch (...) {
				raw_pos = rp, raw_ids = ri;
				throw;
			}
			raw_pos = rp, raw_ids = ri;
		}
		hipLaunchKernelGGL(ivf_big_bound_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, (const float *)ws_bD.p, kf, (long long)nq,
		                   metric == METRIC_L2 ? 1 : 0, (float *)ws_bfix.p);
		}
		// B. grouping, packing (collect_search_pass's launches; the nearest-list item set is built and not scanned), the frozen scan
		const int64_t npairs = nq * np;
		const int max_items = ivf_group_max_items(npairs, nlist, G);
		ws_items.reserve((size_t)max_items * 16);
		ws_qidx.reserve((size_t)npairs * sizeof(int32_t));
		ws_slots.reserve((size_t)npairs * sizeof(int32_t));
		const size_t group_ints = (ivf_group_ws_ints(nlist) + 63) & ~(size_t)63;
		ws_group.reserve(2 * group_ints * sizeof(int));
		if (ws_group.p != group_clean_p || ws_group.cap != group_clean_cap)
			MVS_HIP(hipMemsetAsync(ws_group.p, 0, 2 * group_ints * sizeof(int), stream));
		group_clean_p = nullptr;
		ws_xi.reserve(ivf_collect_xi_bytes(max_items));
		ws_ig.reserve((size_t)max_items * 128 * sizeof(float));
		ws_ie2.reserve((size_t)max_items * 128 * sizeof(float));
		const size_t ctl_bytes = 256 + (size_t)3 * nq * sizeof(int) + ((size_t)nq + 64) * sizeof(int);
		ws_qfail.reserve(ctl_bytes);
		MVS_HIP(hipMemsetAsync(ws_qfail.p, 0, ctl_bytes, stream));
		ctl_clean_p = nullptr;
		int *const ctl_qfail = (int *)((char *)ws_qfail.p + 256), *const ctl_seg = ctl_qfail + nq;
		int *const ctl_flag = ctl_seg + 2 * nq;
		ws_gslot.reserve((size_t)nq * 32 * sizeof(unsigned) + 64);
		// (~ k (rows probed) / (rows of the bound's lists) candidates per query, with slack; grown once when it proves too small)
		if ((double)nq * (double)std::max<int64_t>(4096, 16 * (int64_t)kf) >= 1.5e9)
			return false; // (the grouping and the sort index their entries with 32 bits)
		int64_t cap_entries = std::max<int64_t>(nq * std::max<int64_t>(4096, 16 * (int64_t)kf), (int64_t)1 << 20);
		const unsigned *rowmask = nullptr;
		if (params && params->sel_kind != MVS_SEL_NONE) {
			SelectorDev sel = selector.upload(params, stream);
			ws_rowmask.reserve(ivf_rowmask_bytes(nrows_mf));
			launch_ivf_rowmask(sel, (const int64_t *)rowids_mf.p, (const int *)perm_mf.p, d_idmap, nrows_mf, ws_rowmask.p, stream);
			rowmask = (const unsigned *)ws_rowmask.p;
		}
		int64_t max_list = 0;
		for (int64_t l = 0; l < nlist; l++)
			max_list = std::max(max_list, list_off[(size_t)l + 1] - list_off[(size_t)l]);
		const int seg_rows = cl_seg_rows, nseg = (int)((max_list + seg_rows - 1) / seg_rows);
		const int max_items0 = ivf_group_max_items(nq, nlist, G);
		ws_items0.reserve((size_t)max_items0 * 16);
		ws_qidx0.reserve((size_t)nq * sizeof(int32_t));
		ws_xi0.reserve(ivf_collect_xi_bytes(max_items0));
		ws_ig0.reserve((size_t)max_items0 * 128 * sizeof(float));
		ws_ie20.reserve((size_t)max_items0 * 128 * sizeof(float));
		int *d_nitems = nullptr, *d_nitems0 = nullptr;
		launch_ivf_group2((const int64_t *)ws_cI.p, nq, (int)np, nlist, G, shift, (const int64_t *)lb_dev.p, (const int64_t *)le_dev.p,
		                  (int *)ws_group.p, (int *)ws_group.p + group_ints, ws_items0.p, (int *)ws_qidx0.p, (int *)ws_slots.p, ws_items.p,
		                  (int *)ws_qidx.p, nullptr, &d_nitems0, &d_nitems, stream);
		group_clean_p = ws_group.p, group_clean_cap = ws_group.cap;
		launch_ivf_collect_pack2(metric, d_x, d, nq, (const int *)ws_slots.p, ws_items0.p, ws_xi0.p, (float *)ws_ig0.p, (float *)ws_ie20.p,
		                         ws_items.p, d_nitems, max_items, (const int *)ws_qidx.p, ws_xi.p, (float *)ws_ig.p, (float *)ws_ie2.p,
		                         (const float *)cent_dev.p, (const int *)list_of_blk.p, (const unsigned *)list_max.p, ctl_qfail, nlist,
		                         (unsigned *)ws_gslot.p, 32, (int *)ws_qfail.p, ctl_flag, stream);
		unsigned long long *cnt = (unsigned long long *)ws_qfail.p;
		if (!h_fail)
			MVS_HIP(hipHostMalloc((void **)&h_fail, 512, hipHostMallocDefault));
		memset(&kinfo, 0, sizeof kinfo);
		unsigned long long nstream = 0;
		size_t half = 0;
		for (int attempt = 0;; ++attempt) {
			half = ((size_t)cap_entries * 8 + 255) & ~(size_t)255;
			ws_stream.reserve(256 + 2 * half);
			unsigned long long *strm = (unsigned long long *)((char *)ws_stream.p + 256);
			begin_kernel_timing(stream);
			launch_ivf_collect_scan(ws_items.p, d_nitems, max_items, (const int *)ws_qidx.p, ws_xi.p, (const float *)ws_ig.p, (const float *)ws_ie2.p,
			                        (const unsigned short *)codes_bfr.p, (const float *)beta_mf.p, (unsigned *)ws_gslot.p, strm, cnt, cap_entries, kf,
			                        seg_rows, nseg, 1, rowmask, stream, nullptr, (const float *)ws_bfix.p);
			end_kernel_timing(stream);
			MVS_HIP(hipMemcpyAsync(h_fail + 64, ws_qfail.p, 256, hipMemcpyDeviceToHost, stream));
			std::vector<int> qf;
			if (attempt == 0) { // (the packing kernel flags the queries whose bound is not finite; nobody compacts them here)
				qf.resize((size_t)nq);
				MVS_HIP(hipMemcpyAsync(qf.data(), ctl_qfail, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, stream));
			}
			MVS_HIP(hipStreamSynchronize(stream));
			memcpy(&nstream, h_fail + 64, sizeof nstream);
			bool any_fail = false;
			for (int v : qf)
				any_fail |= v != 0;
			if (any_fail) { // a query without a finite bound: the older paths take the batch
				ctl_clean_p = nullptr;
				return false;
			}
			if ((int64_t)nstream <= cap_entries)
				break;
			++cl_overflows;
			if (attempt > 0 || (int64_t)(nstream + nstream / 8) > nq * std::max<int64_t>(16384, 64 * (int64_t)kf) || (int64_t)(nstream + nstream / 8) >= ((int64_t)1 << 31)) {
				ctl_clean_p = nullptr;
				return false;
			}
			cap_entries = (int64_t)(nstream + nstream / 8);
			MVS_HIP(hipMemsetAsync(cnt, 0, 16, stream));
		}
		ctl_clean_p = nullptr; // (the control block is not in the state collect_search_pass leaves it in)
		unsigned long long *strm = (unsigned long long *)((char *)ws_stream.p + 256);
		unsigned long long *sorted = (unsigned long long *)((char *)ws_stream.p + 256 + half);
		const int64_t ncand = (int64_t)nstream;
		cl_last_admitted = ncand;
		cl_queries_total += nq;
		cl_candidates_total += ncand;
		cl_last_pairs = cl_last_pairs_kept = npairs, cl_last_nq = nq;
		cl_pairs_pruned_pending = false;
		// C. positions, grouping, the scanner's arithmetic, selection
		if (ncand > 0)
			hipLaunchKernelGGL(ivf_big_translate_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, stream, strm, (long long)ncand,
			                   (const int *)perm_mf.p);
		size_t temp = ncand > 0 ? collect_sort_temp_bytes(ncand, nq) : 0;
		if (kk > 128 && ncand > 0)
			temp = std::max(temp, collect_select_big_temp_bytes(ncand, nq));
		ws_pd.reserve(std::max<size_t>(temp, 16));
		ws_big_seg.reserve(256 + (size_t)2 * nq * sizeof(int));
		MVS_HIP(hipMemsetAsync(ws_big_seg.p, 0, 256 + (size_t)2 * nq * sizeof(int), stream));
		const size_t ex_bytes = ((size_t)nq * kk * sizeof(float) + 255) & ~(size_t)255;
		ws_ex.reserve(ex_bytes + (size_t)nq * kk * sizeof(int32_t));
		float *pd1 = (float *)ws_ex.p;
		int32_t *pi1 = (int32_t *)((char *)ws_ex.p + ex_bytes);
		FlatGeom g2 = flat_geom_for(d);
		g2.pair_interleaved = false; // (the list-sorted store holds plain rows of dp floats)
		g2.dp = dp;
		launch_collect_rescore(metric, strm, sorted, ncand, ws_pd.p, temp, nq, kk, d_x, g2, (const float *)codes.p, nullptr, nullptr,
		                       (int *)((char *)ws_big_seg.p + 256), pd1, pi1, true /* the scanner's per-pair arithmetic */, stream, nullptr, true);
		const long long total = (long long)nq * kk;
		hipLaunchKernelGGL(ivf_big_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float *)pd1, (const int *)pi1, total,
		                   metric == METRIC_L2 ? 1 : 0, raw_pos ? nullptr : (const long long *)rowids.p,
		                   (d_idmap && !raw_ids && !raw_pos) ? (const long long *)d_idmap : nullptr, d_D, (long long *)d_I);
		MVS_HIP(hipGetLastError());
		snprintf(kinfo.name, sizeof kinfo.name, "ivf_bf16_collect_kernel");
		kinfo.grid = max_items * nseg;
		kinfo.block = 64;
		kinfo.nsplit = (int)np;
		kinfo.bytes = (double)nrows_mf * 256.0;
		kinfo.flops = (double)nq * np * ((double)nsorted / nlist) * d * 2.0;
		pf_queries_total += nq;
		stream_wait(st, stream);
		return true;
	}
	void search_device(int64_t nq, const float *d_x, int64_t k, float *d_D, int64_t *d_I, const mvs_search_params *params,
	                   hipStream_t st) override {
		search_mapped(nq, d_x, k, d_D, d_I, params, nullptr, st);
	}

	void to_device(int new_device) override {
		if (new_device == device)
			return;
		throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp",
		            "moving an IVF index between devices in place is not implemented on the MI355X path; clone it");
	}
	// deep copy through the host image (faiss::gpu::index_cpu_to_gpu also starts from host memory)
	IndexBase *clone(int on_device) override {
		int ndev = 0;
		MVS_HIP(hipGetDeviceCount(&ndev));
		if (on_device < 0 || on_device >= ndev)
			throw_faiss("faiss::gpu::index_cpu_to_gpu", "faiss/gpu/GpuCloner.cpp", "Invalid GPU device %d", on_device);
		HostIndex h;
		to_host(h);
		return index_from_host(h, on_device);
	}
	// ArrayInvertedLists image: per list, rows and ids in insertion order
	void to_host(HostIndex &out) override {
		use_device();
		MVS_HIP(hipStreamSynchronize(stream));
		out.kind = MVS_KIND_IVFFLAT;
		out.d = d;
		out.metric = metric;
		out.ntotal = ntotal;
		out.is_trained = is_trained;
		out.nlist = nlist;
		out.nprobe = nprobe;
		out.sub.reset(new HostIndex);
		quantizer->to_host(*out.sub);
		out.list_ids.assign((size_t)nlist, {});
		out.list_codes.assign((size_t)nlist, {});
		std::vector<float> rows((size_t)ntotal * dp);
		if (ntotal > 0)
			MVS_HIP(hipMemcpy(rows.data(), raw, rows.size() * sizeof(float), hipMemcpyDeviceToHost));
		for (int64_t i = 0; i < ntotal; i++) {
			const int32_t l = assign_h[(size_t)i];
			if (l < 0)
				continue;
			out.list_ids[(size_t)l].push_back(ids_h[(size_t)i]);
			auto &c = out.list_codes[(size_t)l];
			c.insert(c.end(), &rows[(size_t)i * dp], &rows[(size_t)i * dp] + d);
		}
	}
	// image load: rows enter in list order, which keeps the insertion order inside every list
	void adopt_lists(const HostIndex &h) {
		use_device();
		int64_t n = 0;
		for (const auto &l : h.list_ids)
			n += (int64_t)l.size();
		grow(n);
		std::vector<float> rows((size_t)n * dp, 0.f);
		assign_h.clear();
		ids_h.clear();
		int64_t r = 0;
		for (int64_t l = 0; l < nlist; l++) {
			const auto &li = h.list_ids[(size_t)l];
			const auto &lc = h.list_codes[(size_t)l];
			for (size_t j = 0; j < li.size(); j++, r++) {
				memcpy(&rows[(size_t)r * dp], &lc[j * (size_t)d], (size_t)d * sizeof(float));
				assign_h.push_back((int32_t)l);
				ids_h.push_back(li[j]);
			}
		}
		if (n > 0)
			MVS_HIP(hipMemcpy(raw, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
		ntotal = n;
		dirty = true;
	}
	void set_timing(bool on) override {
		timing_enabled = on;
	}
	void adopt_tuning(const Tuning &t) override {
		tune_ = t;
		quantizer->adopt_tuning(t);
	}
	bool set_option(const char *key, int64_t v) override {
		if (!strcmp(key, "ivf_cl_big")) { // 0: lists beyond 32 entries on the scanner / select kernels (round 5; A/B)
			cl_big = (int)v;
			return true;
		}
		if (!strcmp(key, "ivf_collect")) {
			collect_mode = (int)v;
			return true;
		}
		if (!strcmp(key, "ivf_mfma")) {
			mfma_mode = (int)v;
			return true;
		}
		if (!strcmp(key, "ivf_raw_ids")) { // row shards: labels = the stored ids even when an id map feeds the selector
			raw_ids = v != 0;
			return true;
		}
		if (!strcmp(key, "ivf_select")) { // 1 = the all-distances + segmented-sort path for any k (default: k > 256 only)
			force_select = v != 0;
			return true;
		}
		if (!strcmp(key, "ivf_cl_stream_cap")) { // candidate-stream entries per query (0: 4096, or what the last overflow needed)
			cl_stream_cap_per_query = v;
			return true;
		}
		if (!strcmp(key, "ivf_cl_seg_rows")) { // rows per (work item, segment) wavefront of the main pass: a multiple of 32
			if (v < 32 || v > 65536 || v % 32)
				return false;
			cl_seg_rows = (int)v;
			return true;
		}
		if (!strcmp(key, "ivf_cl_near_rows")) { // rows of the nearest list the pre-pass looks at (a multiple of 32; A/B)
			cl_near_rows = (int)std::max<int64_t>(32, std::min<int64_t>(4096, (v + 31) / 32 * 32));
			return true;
		}
		if (!strcmp(key, "ivf_cl_refilter")) {
			cl_refilter = v != 0;
			return true;
		}
		if (!strcmp(key, "ivf_probe_prune")) {
			cl_prune = v != 0;
			return true;
		}
		if (!strcmp(key, "ivf_exact_ties")) {
			exact_ties = v != 0;
			return true;
		}
		if (!strcmp(key, "ivf_fast_scan")) { // 0 = the LDS-staged flat_direct item kernel
			use_fast_scan = v != 0;
			return true;
		}
		// tuning knobs (csrc/common.h Tuning): this index's own copy AND the quantizer's -- their launches read whichever index
		// entered last on the calling thread, and both must say the same
		const bool mine = set_tuning(key, v);
		return quantizer->set_option(key, v) || mine;
	}
	bool use_fast_scan = true;
	bool exact_ties = true; // option ivf_exact_ties: 0 = the scan kernels' pure (value, position) order, no tie pass (diagnostics)
	bool raw_pos = false;   // inside the exact-tie wrapper: the paths emit positions in the list-sorted store, no id map
	bool force_select = false;
	bool raw_ids = false;
	int collect_mode = -1; // option ivf_collect: -1 auto, 0 never, 1 wherever the kernel exists (d <= 128, k <= 32)
	int mfma_mode = -1; // option ivf_mfma: -1 auto (inner product only), 0 never, 1 always, 2 = L2 prefilter + exact re-scoring

	// ---- Flat shadow (round 5; csrc/index.hip FlatIndex::shadow_search) ---------------------------------------------------------
	// This index as the internal clustering of a Flat L2 index: nprobe nearest lists through the coarse filter, candidates re-scored
	// in the Flat arithmetic, then a proof per query that the unprobed lists cannot matter (ivf_shadow_verify_kernel).  Queries that
	// cannot be proven (and those whose bound was not finite) are appended to the caller's fail list.  false: the path did not run
	// (stream / bucket beyond their limits, coarse quantiser not applicable) -- the caller takes its normal path for the batch.
	struct ShadowCtx {
		const float *qn;           // [nq] ||x||^2, k-ordered chains (the Flat index's re-scoring uses the same)
		const int64_t *out_map;    // IDMap labels of the Flat rows, or nullptr
		int64_t out_off;           // ... else label = row + out_off
		const unsigned *ymax_bits; // the Flat index's largest ||y||^2
		int *fail_cnt, *fail_q;
	};
	const ShadowCtx *shadow = nullptr;
	DevBuf norms_csr; // ||y||^2 of every row by position in the list-sorted store (k-ordered chains: the Flat index's norms)
	int64_t norms_csr_rows = -1;
	int64_t flat_shadow_max_queries(int shadow_nprobe) override {
		const int64_t np = std::min<int64_t>(std::max(shadow_nprobe, 1), nlist);
		const int64_t by_pairs = (((int64_t)1 << 26) - 1) / np;
		const int64_t by_matrix = std::max<int64_t>(64, ((int64_t)512 << 20) / (std::max<int64_t>(nlist, 1) * 4) / 64 * 64); // FlatIndex::coarse_matrix_covers
		return std::min(by_pairs, by_matrix) / 64 * 64;
	}
	size_t device_bytes() const override {
		return (size_t)cap * dp * sizeof(float) + codes.cap + codes_bfr.cap + beta_mf.cap + rowids.cap + norms_csr.cap;
	}
	int flat_shadow_search(int64_t nq, const float *d_x, int64_t k, const float *d_qn, float *d_D, int64_t *d_I, const int64_t *d_out_map,
	                       int64_t out_off, const unsigned *d_ymax_bits, int *d_fail_cnt, int *d_fail_q, int shadow_nprobe,
	                       hipStream_t st) override {
		use_device();
		if (metric != METRIC_L2 || hnsw_M != 0 || d != dp || k > 32 || nq < 20 || ntotal <= k)
			return 1;
		const int64_t np = std::min<int64_t>(shadow_nprobe, nlist);
		if (nq * np >= ((int64_t)1 << 26))
			return 1;
		stream_wait(stream, st);
		build_lists_mf(false);
		if (!have_bfr)
			return 2;
		if (norms_csr_rows != nsorted) {
			norms_csr.reserve((size_t)std::max<int64_t>(nsorted, 1) * sizeof(float));
			launch_query_norms((const float *)codes.p, nsorted, d, (float *)norms_csr.p, stream);
			norms_csr_rows = nsorted;
		}
		ws_cD.reserve((size_t)nq * np * sizeof(float));
		ws_cI.reserve((size_t)nq * np * sizeof(int64_t));
		FlatIndex *qz = static_cast<FlatIndex *>(quantizer);
		if (!qz->coarse_topk(nq, d_x, np, (float *)ws_cD.p, (int64_t *)ws_cI.p, stream) || !qz->coarse_matrix_covers(nq))
			return 1;
		use_device();
		ShadowCtx ctx;
		ctx.qn = d_qn, ctx.out_map = d_out_map, ctx.out_off = out_off, ctx.ymax_bits = d_ymax_bits, ctx.fail_cnt = d_fail_cnt, ctx.fail_q = d_fail_q;
		shadow = &ctx;
		last_np = np;
		bool ok = false;
		try {
			ok = collect_search(nq, d_x, k, d_D, d_I, nullptr, nullptr, st, np);
		} catch (...) {
			shadow = nullptr;
			throw;
		}
		shadow = nullptr;
		return ok ? 0 : 2;
	}
	// introspection for parity tests
	void get_centroids(float *out) {
		use_device();
		HostIndex h;
		quantizer->to_host(h);
		const std::vector<float> &rows = h.kind == MVS_KIND_FLAT ? h.rows : h.sub->rows;
		memcpy(out, rows.data(), rows.size() * sizeof(float));
	}

private:
	float *raw = nullptr; // append-only [cap][dp]
	int64_t cap = 0;
	std::vector<int32_t> assign_h;
	std::vector<int64_t> ids_h;
	bool dirty = false;
	std::vector<int64_t> list_off;
	int64_t nsorted = 0;
	DevBuf codes, rowids;
	DevBuf ws_cD, ws_cI, ws_items, ws_qidx, ws_slots, ws_q, ws_pd, ws_pi, ws_gslot, ws_xi, ws_group, list_off_dev;
	// MFMA view of the lists (build_lists_mf)
	FlatGeom geom {};
	bool mf_dirty = true;
	int64_t nrows_mf = 0;
	DevBuf codes_mf, norms_mf, rowids_mf, lb_dev, le_dev, max_norm_mf, cent_dev, list_of_blk, perm_mf, ws_iqn, ws_qmaxn;
	bool mf_residual = false;
	// bf16 coarse filter (csrc/ivf_collect.hip): residual rows as bf16, -||y'||^2, the largest ||y'||^2 of every list
	DevBuf codes_bfr, beta_mf, list_max, ws_ig, ws_ie2, ws_qfail, ws_stream, ws_sorttmp, ws_seg, ws_rowmask;
	bool have_bfr = false, mf_have_f32 = false;
	int64_t cl_queries_total = 0, cl_candidates_total = 0, cl_overflows = 0, cl_cap_hint = 0, cl_stream_cap_per_query = 0;
	bool collect_stats(int64_t *queries, int64_t *candidates, int64_t *overflows) override {
		if (queries)
			*queries = cl_queries_total;
		if (candidates)
			*candidates = cl_candidates_total;
		if (overflows)
			*overflows = cl_overflows;
		return true;
	}
	// (query, list) pairs of the last coarse-filter search and how many of them the scan kept (mvs_index_ivf_probe_stats; the per-query
	// counts stay on the device until somebody asks)
	bool probe_stats(int64_t *pairs, int64_t *scanned, int64_t *bursts, int64_t *admitted) override {
		use_device();
		if (cl_pairs_pruned_pending && cl_last_nq > 0) {
			MVS_HIP(hipStreamSynchronize(stream));
			std::vector<int> h((size_t)cl_last_nq);
			MVS_HIP(hipMemcpy(h.data(), ws_kept.p, h.size() * sizeof(int), hipMemcpyDeviceToHost));
			cl_last_pairs_kept = 0;
			for (int v : h)
				cl_last_pairs_kept += v;
			cl_pairs_pruned_pending = false;
		}
		if (pairs)
			*pairs = cl_last_pairs;
		if (scanned)
			*scanned = cl_last_pairs_kept;
		if (bursts)
			*bursts = cl_last_bursts;
		if (admitted)
			*admitted = cl_last_admitted;
		return true;
	}
	int64_t cl_last_nq = 0, cl_last_bursts = 0;
	int cl_seg_rows = 512;       // option ivf_cl_seg_rows
	int cl_bpitch = 1024;        // bucket entries per query (grown on demand up to 16 384)
	int cl_near_rows = 256;      // option ivf_cl_near_rows: rows of every query's nearest list the publish-only pre-pass walks
	bool cl_refilter = true;     // option ivf_cl_refilter: candidates that do not pass the bound the scan ENDED with are dropped before the exact stage
	DevBuf ws_stream2;           // {u per stream entry | Bf per query | the filtered stream}
	int64_t cl_last_admitted = 0; // stream entries of the last search (before the final-bound filter)
	bool cl_prune = true;        // option ivf_probe_prune: probed lists that provably hold none of a query's k nearest rows are not scanned (L2)
	int64_t cl_last_pairs = 0, cl_last_pairs_kept = 0; // (query, list) pairs of the last coarse-filter search / of those, scanned
	DevBuf ws_cIp, ws_kept;
	bool cl_pairs_pruned_pending = false; // ws_kept of the last search has not been summed yet (collect statistics do it on demand)
	void *group_clean_p = nullptr, *ctl_clean_p = nullptr; // the buffers known to be left zeroed by the previous search's kernels
	size_t group_clean_cap = 0, ctl_clean_cap = 0;
	int64_t ctl_clean_nq = 0;
	DevBuf ws_items0, ws_qidx0, ws_xi0, ws_ig0, ws_ie20; // the nearest-list pre-pass's own work items (prep2)
	int cl_bpitch_try = 0;       // ... of the repeated pass of the search in progress
	int64_t cl_cap_try = 0;      // stream entries of the repeated pass of a search whose stream overflowed
	// the exact-tie wrapper's final outputs, handed to the bucket path (which sets fin_done when it printed them itself)
	float *fin_D = nullptr;
	int64_t *fin_I = nullptr;
	const int64_t *fin_idmap = nullptr;
	int64_t fin_k = 0;
	bool fin_done = false;
	DevBuf ws_cand, ws_ex, ws_fail, ws_fb, ws_tD, ws_tI, ws_tflag;
	DevBuf ws_bD, ws_bI, ws_bfix, ws_big_seg; // collect_search_big: the bound pass's lists, B(q), the grouping's segments
	int cl_big = 1;                           // option ivf_cl_big: lists beyond 32 entries on the bf16 filter (0: the scanner / select kernels, round 5)
	int *h_fail = nullptr; // pinned
	bool pf_suppressed = false; // while the queries the proof rejected are re-run on the scanner kernel
	bool reuse_coarse = false;
	int64_t pf_fallback_total = 0, pf_queries_total = 0;
	SelectorHolder selector;
};

IndexBase *make_ivf_index(int d, const std::string &desc, int metric) {
	if (desc.rfind("IVF", 0) != 0)
		return nullptr;
	char *end = nullptr;
	const long nlist = strtol(desc.c_str() + 3, &end, 10);
	if (end == desc.c_str() + 3 || nlist <= 0)
		return nullptr;
	if (!strcmp(end, ",Flat"))
		return new IVFFlatIndex(d, nlist, metric);
	if (!strncmp(end, "_HNSW", 5)) { // "IVF<n>_HNSW<m>,Flat" (reference Makefile:93): HNSW coarse quantizer, M default 32
		char *end2 = nullptr;
		long M = strtol(end + 5, &end2, 10);
		if (end2 == end + 5)
			M = 32;
		if (!strcmp(end2, ",Flat") && M > 1)
			return new IVFFlatIndex(d, nlist, metric, (int)M);
	}
	throw_faiss("faiss::Index* faiss::index_factory(int, const char*, faiss::MetricType)", "faiss/index_factory.cpp",
	            "This index type is not implemented on the MI355X path yet: %s", desc.c_str());
}
IndexBase *ivf_from_host(const HostIndex &h, int device) {
	CtorDevice scope(device);
	if (!h.sub || (h.sub->kind != MVS_KIND_FLAT && h.sub->kind != MVS_KIND_HNSW))
		throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp",
		            "only Flat and HNSWFlat coarse quantizers are implemented on the MI355X path");
	const bool hq = h.sub->kind == MVS_KIND_HNSW;
	if (hq && (h.sub->cum_nneighbor_per_level.size() < 2))
		throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp", "bad HNSW level table");
	if ((int64_t)h.list_ids.size() != h.nlist || (int64_t)h.list_codes.size() != h.nlist)
		throw_faiss("faiss::Index* faiss::read_index(...)", "faiss/impl/index_read.cpp", "inverted lists do not match nlist");
	auto *v = new IVFFlatIndex(h.d, h.nlist, h.metric, hq ? h.sub->cum_nneighbor_per_level[1] / 2 : 0);
	try {
		v->nprobe = h.nprobe;
		if (hq) { // adopt the stored graph instead of rebuilding it
			IndexBase *q = hnsw_from_host(*h.sub, device);
			delete v->quantizer;
			v->quantizer = q;
		} else if (h.sub->ntotal > 0)
			v->quantizer->add(h.sub->ntotal, h.sub->rows.data());
		v->is_trained = h.is_trained;
		v->adopt_lists(h);
	} catch (...) {
		delete v;
		throw;
	}
	return v;
}
IndexBase *ivf_quantizer_of(IndexBase *ix) {
	if (ix->kind != MVS_KIND_IVFFLAT)
		return nullptr;
	return static_cast<IVFFlatIndex *>(ix)->quantizer;
}
int64_t ivf_nlist_of(IndexBase *ix) {
	return ix->kind == MVS_KIND_IVFFLAT ? static_cast<IVFFlatIndex *>(ix)->nlist : 0;
}
bool ivf_get_centroids(IndexBase *ix, float *out) {
	if (ix->kind != MVS_KIND_IVFFLAT)
		return false;
	static_cast<IVFFlatIndex *>(ix)->get_centroids(out);
	return true;
}
bool ivf_set_centroids(IndexBase *ix, const float *c) {
	if (ix->kind != MVS_KIND_IVFFLAT)
		return false;
	auto *v = static_cast<IVFFlatIndex *>(ix);
	if (v->hnsw_M > 0) {
		if (v->quantizer->ntotal != 0)
			throw_faiss("mvs_index_ivf_set_centroids", __FILE__, "the HNSW coarse quantizer already holds centroids");
	} else {
		static_cast<FlatIndex *>(v->quantizer)->reset();
	}
	v->quantizer->add(v->nlist, c);
	v->is_trained = true;
	return true;
}

} // namespace mvs
