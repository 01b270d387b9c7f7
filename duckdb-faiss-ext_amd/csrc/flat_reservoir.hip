// csrc/flat_reservoir.hip -- IndexFlat inner-product search with k >= 100 and an EXACT tie at the k-th score: FAISS's
// ReservoirTopN outcome, replayed on the device for the queries the merge flags (usually none).
//
// From k = distance_compute_min_k_reservoir = 100 on, knn_inner_product / knn_L2sqr (faiss/utils/distances.cpp, reached from
// /root/reference/src/faiss_extension.cpp:631 -- the Go harness asks for up to ~2 000 rows per query, go/main_test.go:26-32) keep
// their candidates in a reservoir of capacity (2k + 15) & ~15 (faiss/impl/ResultHandler.h ReservoirTopN) that is cut back to
// between k and (capacity + k) / 2 entries by partition_fuzzy_median3 (faiss/utils/partitioning.cpp) whenever it is full, and
// only at the end pass through the heap rule.  For L2 -- rows arrive in ascending id, CMax heap -- the result is the k smallest
// (distance, id), the same pure function the heap gives: nothing to do.  For inner product the rows TIED at the k-th score that
// survive depend on where the sampled thresholds fell, i.e. on the whole history of the stream, so a flagged query gets exactly
// that: every row's score (the k-ordered fma chain of every other kernel), then one wavefront walks the scores front to back
// through the reservoir (oracle/orc_core.c reservoir_t is the CPU twin, function for function).
// The final stage -- "the first k stored entries are pushed on a heap, the rest pass the strict heap rule" -- is the closed
// form of DESIGN.md 3.5 applied to the STORED entries (they sit in arrival = row order): T = the k-th best score, A_k = the first
// k stored entries with score >= T, result = {entries above T} + {tied entries of A_k minus the G with the smallest rows},
// G = #(entries above T outside A_k).
#include "index.h"

namespace mvs {

namespace {

constexpr int RS_CHUNK = 4096; // scores staged per round of the replay (16 KB of LDS behind the reservoir)

__device__ __forceinline__ bool rs_member(const SelectorDev &s, long long id) {
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

// scores[f][row] = fmaf chain over k = 0 .. d-1 of xf[f][k] * y[row][k] (the arithmetic of every inner-product kernel and of
// the oracle's ip_chain); rows an IDSelector rejects: NaN (no compare admits them, as FAISS never sees them)
__global__ __launch_bounds__(256) void ip_scores_kernel(const float *__restrict__ xf, int d, const float *__restrict__ vecs, int dp,
                                                       int interleaved, long long n, SelectorDev sel,
                                                       const long long *__restrict__ idmap, float *__restrict__ scores) {
	const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int f = blockIdx.y;
	if (row >= n)
		return;
	float acc = __uint_as_float(0x7fc00000u);
	if (sel.kind == MVS_SEL_NONE || rs_member(sel, idmap ? idmap[row] : row)) {
		const float *x = xf + (size_t)f * d;
		const float *y = vecs + (size_t)row * dp;
		const bool odd = interleaved && ((row >> 4) & 1);
		acc = 0.f;
		for (int g4 = 0; g4 < d; g4 += 4) {
			float v0, v1, v2, v3;
			if (dp % 4 == 0) {
				const float4 s = *(const float4 *)(y + g4);
				if (!interleaved)
					v0 = s.x, v1 = s.y, v2 = s.z, v3 = s.w;
				else if (odd) // stored [k1,k3,k0,k2] (FlatGeom::pair_interleaved)
					v0 = s.z, v1 = s.x, v2 = s.w, v3 = s.y;
				else // stored [k0,k2,k1,k3]
					v0 = s.x, v1 = s.z, v2 = s.y, v3 = s.w;
			} else {
				v0 = y[g4], v1 = g4 + 1 < dp ? y[g4 + 1] : 0.f, v2 = g4 + 2 < dp ? y[g4 + 2] : 0.f, v3 = g4 + 3 < dp ? y[g4 + 3] : 0.f;
			}
			acc = fmaf(x[g4], v0, acc);
			if (g4 + 1 < d)
				acc = fmaf(x[g4 + 1], v1, acc);
			if (g4 + 2 < d)
				acc = fmaf(x[g4 + 2], v2, acc);
			if (g4 + 3 < d)
				acc = fmaf(x[g4 + 3], v3, acc);
		}
	}
	scores[(size_t)f * n + row] = acc;
}

// (round 6) the same scores with COALESCED row reads: a workgroup's 256 rows pass through LDS in slabs of 32 dimensions (128 contiguous
// bytes per row and slab: eight lanes x 16 bytes), thread t then walks ITS row's slab -- the k-ordered chain is per row, so the rows
// cannot be split across lanes, but their bytes need not be fetched 512 bytes apart by neighbouring threads (17 ms per flagged batch at
// N = 10 M, d = 128 with the thread-per-row loads above; this one runs at the store's streaming rate).  dp % 32 == 0.
__global__ __launch_bounds__(256) void ip_scores_tiled_kernel(const float *__restrict__ xf, int d, const float *__restrict__ vecs, int dp,
                                                             int interleaved, long long n, SelectorDev sel,
                                                             const long long *__restrict__ idmap, float *__restrict__ scores) {
	__shared__ float tile[256][33];
	const int t = threadIdx.x, f = blockIdx.y;
	const long long row0 = (long long)blockIdx.x * 256, row = row0 + t;
	const float *x = xf + (size_t)f * d;
	const bool odd = interleaved && ((row >> 4) & 1);
	float acc = 0.f;
	for (int s0 = 0; s0 < d; s0 += 32) {
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 8; ++i) { // 2 048 chunks of 16 bytes: chunk c = row (c >> 3), floats 4 (c & 7) ..
			const int c = i * 256 + t, r = c >> 3, j4 = (c & 7) * 4;
			float4 v = {0.f, 0.f, 0.f, 0.f};
			if (row0 + r < n)
				v = *(const float4 *)(vecs + (size_t)(row0 + r) * dp + s0 + j4);
			tile[r][j4] = v.x, tile[r][j4 + 1] = v.y, tile[r][j4 + 2] = v.z, tile[r][j4 + 3] = v.w;
		}
		__syncthreads();
#pragma unroll
		for (int g4 = 0; g4 < 32; g4 += 4) {
			const float a0 = tile[t][g4], a1 = tile[t][g4 + 1], a2 = tile[t][g4 + 2], a3 = tile[t][g4 + 3];
			float v0, v1, v2, v3;
			if (!interleaved)
				v0 = a0, v1 = a1, v2 = a2, v3 = a3;
			else if (odd) // stored [k1,k3,k0,k2] (FlatGeom::pair_interleaved)
				v0 = a2, v1 = a0, v2 = a3, v3 = a1;
			else // stored [k0,k2,k1,k3]
				v0 = a0, v1 = a2, v2 = a1, v3 = a3;
			const int k0 = s0 + g4;
			if (k0 < d)
				acc = fmaf(x[k0], v0, acc);
			if (k0 + 1 < d)
				acc = fmaf(x[k0 + 1], v1, acc);
			if (k0 + 2 < d)
				acc = fmaf(x[k0 + 2], v2, acc);
			if (k0 + 3 < d)
				acc = fmaf(x[k0 + 3], v3, acc);
		}
	}
	if (row >= n)
		return;
	if (sel.kind != MVS_SEL_NONE && !rs_member(sel, idmap ? idmap[row] : row))
		acc = __uint_as_float(0x7fc00000u);
	scores[(size_t)f * n + row] = acc;
}

__device__ __forceinline__ float rs_median3(float a, float b, float c) {
	if (a > b) {
		const float t = a;
		a = b;
		b = t;
	}
	if (c > b)
		return b;
	if (c > a)
		return c;
	return a;
}
// C = CMin<float, int64> (keeps the LARGEST): C::cmp(a, b) = a < b
__device__ __forceinline__ bool rs_cmp(float a, float b) {
	return a < b;
}

// One wavefront per flagged query.  LDS: vals[cap] f32 | rows[cap] i32.  Everything that FAISS decides sequentially is decided
// here in the same order; the wave only parallelises counting, probing and compaction, whose results do not depend on order.
// (round 6: FOUR wavefronts per query.  All four stage the scores -- a quarter of a 16 384-score round each, sixteen 16-byte loads
// per lane in flight -- and note which groups of 64 hold a score above the threshold as it stood; wavefront 0 alone then walks those
// groups through the reservoir, in order.  One wavefront could not pull 40 MB of scores per query faster than 6 ms.)
__device__ __forceinline__ void rs_wave_fence() { // a wave's LDS operations execute in order: only the compiler must not reorder them
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(256) void reservoir_replay_kernel(const float *__restrict__ scores, long long n, int k, int cap,
                                                              const float *__restrict__ Tq, float *__restrict__ out_v,
                                                              int *__restrict__ out_r) {
	extern __shared__ __attribute__((aligned(16))) float rs_lds[];
	float *vals = rs_lds;
	int *rows = (int *)(rs_lds + cap);
	const int f = blockIdx.x, lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const float *sc = scores + (size_t)f * n;
	const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
	float thr = -FLT_MAX; // C::neutral()
	int fill = 0;

	auto shrink = [&]() { // threshold = partition_fuzzy<C>(vals, ids, capacity, n = k, (capacity + k) / 2, &i)
		rs_wave_fence(); // (wavefront 0 only: orders the appends of other lanes before the reads below)
		const int nn = cap, q_min = k, q_max = (cap + k) / 2;
		float thresh_inf = FLT_MAX;   // C::Crev::neutral()
		float thresh_sup = -FLT_MAX;  // C::neutral()
		float thresh = rs_median3(vals[0], vals[nn / 2], vals[nn - 1]);
		int n_lt = 0, n_eq = 0, q = 0;
		for (int it = 0; it < 200; ++it) {
			n_lt = n_eq = 0;
			for (int j0 = 0; j0 < nn; j0 += 64) { // count_lt_and_eq
				const int j = j0 + lane;
				const bool in = j < nn;
				const float v = in ? vals[j] : 0.f;
				n_lt += __popcll(__builtin_amdgcn_ballot_w64(in && rs_cmp(thresh, v)));
				n_eq += __popcll(__builtin_amdgcn_ballot_w64(in && v == thresh));
			}
			if (n_lt <= q_min) {
				if (n_lt + n_eq >= q_min) {
					q = q_min;
					break;
				}
				thresh_inf = thresh;
			} else if (n_lt <= q_max) {
				q = n_lt;
				break;
			} else {
				thresh_sup = thresh;
			}
			// sample_threshold_median3: the first three probes vals[(j * 6700417) % n], j = 0, 1, ..., strictly between the bounds
			float val3[3];
			int vi = 0;
			for (int j0 = 0; j0 < nn && vi < 3; j0 += 64) {
				const int j = j0 + lane;
				const bool in = j < nn;
				const float v = in ? vals[(int)(((unsigned long long)j * 6700417ull) % (unsigned long long)nn)] : 0.f;
				unsigned long long m = __builtin_amdgcn_ballot_w64(in && rs_cmp(v, thresh_inf) && rs_cmp(thresh_sup, v));
				while (m != 0ull && vi < 3) {
					const int L = __builtin_ctzll(m);
					m &= m - 1ull;
					val3[vi++] = __shfl(v, L);
				}
			}
			const float new_thresh = vi == 3 ? rs_median3(val3[0], val3[1], val3[2]) : (vi != 0 ? val3[0] : thresh_inf);
			if (new_thresh == thresh_inf)
				break;
			thresh = new_thresh;
		}
		int n_eq_1 = q - n_lt;
		if (n_eq_1 < 0) {
			q = q_min;
			thresh = nextafterf(thresh, INFINITY); // C::Crev::nextafter
			n_eq_1 = q;
		}
		// compress_array: order preserving; of the entries equal to thresh the first n_eq_1 stay
		int wp = 0, eq_seen = 0;
		for (int j0 = 0; j0 < nn; j0 += 64) {
			const int j = j0 + lane;
			const bool in = j < nn;
			const float v = in ? vals[j] : 0.f;
			const int r = in ? rows[j] : 0;
			const bool better = in && rs_cmp(thresh, v);
			const unsigned long long eqm = __builtin_amdgcn_ballot_w64(in && !better && v == thresh);
			const bool eq_keep = ((eqm >> lane) & 1ull) && eq_seen + __popcll(eqm & lt_mask) < n_eq_1;
			const bool keep = better || eq_keep;
			const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
			if (keep) {
				const int pos = wp + __popcll(km & lt_mask);
				vals[pos] = v;
				rows[pos] = r;
			}
			wp += __popcll(km);
			eq_seen += __popcll(eqm);
		}
		fill = wp;
		thr = thresh;
		rs_wave_fence();
	};

	float *chunk = (float *)(rows + cap);                                  // [4][RS_CHUNK] this round's scores
	unsigned long long *gm_lds = (unsigned long long *)(chunk + 4 * RS_CHUNK); // [4] groups of each quarter worth a visit
	float *thr_lds = (float *)(gm_lds + 4);                                // [1] the threshold as wavefront 0 left it
	if (threadIdx.x == 0)
		thr_lds[0] = thr;
	for (long long rbase = 0; rbase < n; rbase += 4 * RS_CHUNK) {
		__syncthreads(); // (the previous round's walk is over; thr_lds is current)
		{
			const float thr_s = thr_lds[0]; // (it only rises: groups selected under an older value are a superset)
			const long long cbase = rbase + (long long)wave * RS_CHUNK;
			float *dstc = chunk + wave * RS_CHUNK;
			unsigned long long gmask = 0ull;
			if (cbase + RS_CHUNK <= n && ((size_t)(sc + cbase) & 15) == 0) { // (uniform) a whole aligned quarter
				float4 x[RS_CHUNK / 256];
#pragma unroll
				for (int i = 0; i < RS_CHUNK / 256; ++i)
					x[i] = *(const float4 *)(sc + cbase + (long long)(i * 64 + lane) * 4);
#pragma unroll
				for (int i = 0; i < RS_CHUNK / 256; ++i) { // load i covers groups 4 i .. 4 i + 3, sixteen lanes each
					*(float4 *)(dstc + (i * 64 + lane) * 4) = x[i];
					const unsigned long long m =
					    __builtin_amdgcn_ballot_w64(rs_cmp(thr_s, x[i].x) || rs_cmp(thr_s, x[i].y) || rs_cmp(thr_s, x[i].z) || rs_cmp(thr_s, x[i].w));
					const unsigned long long q4 = (m & 0xffffull ? 1ull : 0ull) | (m & 0xffff0000ull ? 2ull : 0ull) |
					                              (m & 0xffff00000000ull ? 4ull : 0ull) | (m & 0xffff000000000000ull ? 8ull : 0ull);
					gmask |= q4 << (4 * i);
				}
			} else if (cbase < n) {
				for (int i = 0; i < RS_CHUNK / 64; ++i) {
					const long long e = cbase + (long long)i * 64 + lane;
					dstc[i * 64 + lane] = e < n ? sc[e] : __uint_as_float(0x7fc00000u);
				}
				const long long left = n - cbase;
				const int ngroups = (int)((left < RS_CHUNK ? left : RS_CHUNK) + 63) / 64;
				gmask = ngroups >= 64 ? ~0ull : ((1ull << ngroups) - 1ull);
			}
			if (lane == 0)
				gm_lds[wave] = gmask;
		}
		__syncthreads();
		if (wave != 0)
			continue;
	for (int w = 0; w < 4; ++w) {
		unsigned long long gmask = gm_lds[w];
		const long long cbase = rbase + (long long)w * RS_CHUNK;
		const float *cw = chunk + w * RS_CHUNK;
	while (gmask != 0ull) {
		const int g = __builtin_ctzll(gmask);
		gmask &= gmask - 1ull;
		const long long row = cbase + (long long)g * 64 + lane;
		const float v = cw[g * 64 + lane];
		unsigned long long mask = __builtin_amdgcn_ballot_w64(rs_cmp(thr, v)); // C::cmp(threshold, val); NaN: never (also behind the last row)
		while (mask != 0ull) {
			const int cnt = __popcll(mask);
			const int room = cap - fill;
			const int rank = __popcll(mask & lt_mask);
			const bool mine = (mask >> lane) & 1ull;
			if (cnt <= room) { // nobody meets a full reservoir: everyone appends, in lane = row order
				if (mine) {
					vals[fill + rank] = v;
					rows[fill + rank] = (int)row;
				}
				fill += cnt;
				mask = 0ull;
			} else {
				if (mine && rank < room) {
					vals[fill + rank] = v;
					rows[fill + rank] = (int)row;
				}
				fill = cap;
				// the next admitted entry finds i == capacity: shrink_fuzzy(), then IT is stored whatever the new threshold is
				const int L = __builtin_ctzll(__builtin_amdgcn_ballot_w64(mine && rank == room));
				const unsigned long long rest = mask & ~(~0ull >> (63 - L)); // the admitted lanes behind L
				shrink();
				if (lane == L) {
					vals[fill] = v;
					rows[fill] = (int)row;
				}
				fill += 1;
				// the entries behind it face the new threshold
				mask = rest & __builtin_amdgcn_ballot_w64(rs_cmp(thr, v));
			}
		}
	}
	}
		if (lane == 0)
			thr_lds[0] = thr;
	}
	if (wave != 0)
		return;

	// to_result as the closed form over the stored entries (array order = row order)
	rs_wave_fence();
	const float T = Tq[f];
	int n_ge = 0, G = 0;
	for (int j0 = 0; j0 < fill; j0 += 64) {
		const int j = j0 + lane;
		const bool in = j < fill;
		const float v = in ? vals[j] : 0.f;
		const bool ge = in && v >= T;
		const unsigned long long gm = __builtin_amdgcn_ballot_w64(ge);
		const int rk = n_ge + __popcll(gm & lt_mask);
		G += __popcll(__builtin_amdgcn_ballot_w64(ge && v > T && rk >= k));
		n_ge += __popcll(gm);
	}
	int wp = 0, ge_seen = 0, tied_seen = 0;
	for (int j0 = 0; j0 < fill; j0 += 64) {
		const int j = j0 + lane;
		const bool in = j < fill;
		const float v = in ? vals[j] : 0.f;
		const int r = in ? rows[j] : -1;
		const bool ge = in && v >= T;
		const unsigned long long gm = __builtin_amdgcn_ballot_w64(ge);
		const int rk = ge_seen + __popcll(gm & lt_mask);
		const bool tied_in = ge && v == T && rk < k;
		const unsigned long long tm = __builtin_amdgcn_ballot_w64(tied_in);
		const int trk = tied_seen + __popcll(tm & lt_mask);
		const bool keep = (ge && v > T) || (tied_in && trk >= G);
		const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
		const int pos = wp + __popcll(km & lt_mask);
		if (keep && pos < k) {
			out_v[(size_t)f * k + pos] = v;
			out_r[(size_t)f * k + pos] = r;
		}
		wp += __popcll(km);
		ge_seen += __popcll(gm);
		tied_seen += __popcll(tm);
	}
	for (int j = (wp < k ? wp : k) + lane; j < k; j += 64) { // (fewer than k admissible rows)
		out_v[(size_t)f * k + j] = -FLT_MAX;
		out_r[(size_t)f * k + j] = -1;
	}
}

} // namespace

int64_t reservoir_replay_max_k() { // vals + rows of the reservoir in one wave's LDS
	return ((150 * 1024 - 4 * RS_CHUNK * 4 - 64) / 8 - 16) / 2;
}

// d_xf: [nf][d] the flagged queries, d_T: [nf] their k-th best scores; out: [nf][k] (score, row) of FAISS's result, any order
void launch_reservoir_replay(const float *d_xf, int nf, int d, const float *d_vecs, int dp, int interleaved, int64_t n, int64_t k,
                             SelectorDev sel, const int64_t *d_idmap, const float *d_T, float *d_scores, float *d_out_v,
                             int32_t *d_out_r, hipStream_t st) {
	if (nf <= 0)
		return;
	const int cap = (int)((2 * k + 15) & ~(int64_t)15);
	if (dp % 32 == 0)
		hipLaunchKernelGGL(ip_scores_tiled_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nf), dim3(256), 0, st, d_xf, d, d_vecs, dp,
		                   interleaved, (long long)n, sel, (const long long *)d_idmap, d_scores);
	else
	hipLaunchKernelGGL(ip_scores_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nf), dim3(256), 0, st, d_xf, d, d_vecs, dp,
	                   interleaved, (long long)n, sel, (const long long *)d_idmap, d_scores);
	const size_t lds = (size_t)cap * 8 + (size_t)4 * RS_CHUNK * 4 + 64;
	auto kern = reservoir_replay_kernel;
	ensure_dynamic_lds((const void *)kern, lds);
	hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(256), lds, st, (const float *)d_scores, (long long)n, (int)k, cap, d_T, d_out_v,
	                   d_out_r);
	MVS_HIP(hipGetLastError());
}

} // namespace mvs
