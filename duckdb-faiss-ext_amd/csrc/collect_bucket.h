// csrc/collect_bucket.h -- round 5: candidates in PER-QUERY BUCKETS instead of one stream sorted by query.
//
// Rounds 2-4 appended every candidate of a coarse-filter scan to one global stream (q << 32 | row), radix-sorted it by
// query (rocPRIM: histogram + two scatter passes + a tail fill + a segment kernel), re-scored it in place and selected per
// segment: eight launches between the scan and the result, and a sort whose size had to be guessed on the host from the
// previous search.  Now the re-scoring kernel reads the stream AS IT IS and puts every exact key into ITS QUERY's bucket
// (bucket[q][pitch], position from the query's counter; the atomic's round trip hides behind the row gather and the chain),
// and one wavefront per query selects that bucket's k best and writes the search's output.  Nothing is sorted, nothing is
// guessed; two launches.
//
// This header holds the part both index kinds share: selection of the kk smallest 64-bit keys of a query (value key << 32 |
// row) by one wavefront.  The exact arithmetic differs (Flat: the BLAS-branch formula, csrc/flat_collect.hip; IVF: the
// scanner's, csrc/ivf_collect.hip) and lives with its kernel.
#pragma once
#include "flat_fused.h"

namespace mvs {

constexpr unsigned long long CB_EMPTY = ~0ull;

__device__ __forceinline__ unsigned long long cb_lane64(unsigned long long v, int l) { // l uniform
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
	return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long cb_shr1(unsigned long long v) { // lane i <- lane i - 1 (lane 0: 0)
	return ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false) << 32) |
	       (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
}
// COHERENT: keys written by other wavefronts of THIS launch (behind a counter): read past this CU's vector cache and this XCD's L2
template <bool COHERENT>
__device__ __forceinline__ unsigned long long cb_load_key(const unsigned long long *p) {
	return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// The kk <= 64 smallest of keys[0 .. n) (EMPTY entries never count), ascending, entry j in lane j of the result (EMPTY where
// there are fewer).  One wavefront; surv: 256 and top: 64 entries of LDS owned by this wave.  The algorithm is round 4's
// collect_select_kernel: up to 1 024 keys in registers, U = the kk-th smallest of the 64 lane minima by value bounds the kk-th
// value from above, the few keys <= U are ranked against each other in LDS; longer inputs seed a sorted list in registers with
// that result and insert only what beats its running worst, eight loads in flight.
template <bool COHERENT>
__device__ __forceinline__ unsigned long long cb_select_wave(const unsigned long long *__restrict__ keys, int n, int kk, int lane,
                                                              unsigned long long *surv, unsigned long long *top) {
	constexpr int R = 16;
	int start = 0;
	unsigned long long seed_mine = CB_EMPTY, seed_worst = CB_EMPTY;
	{
		const int ce = n <= 1024 ? n : 1024;
		const int nr = (ce + 63) >> 6;
		unsigned long long kreg[R];
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const int i = 64 * r + lane;
			kreg[r] = (r < nr && i < ce) ? cb_load_key<COHERENT>(keys + i) : CB_EMPTY;
		}
		unsigned U = 0xffffffffu;
		if (ce > 64) {
			unsigned long long lmin = kreg[0];
#pragma unroll
			for (int r = 1; r < R; ++r)
				lmin = kreg[r] < lmin ? kreg[r] : lmin;
			const unsigned hi = (unsigned)(lmin >> 32); // (EMPTY: 0xffffffff)
			U = 0u;
#pragma unroll 1
			for (int bit = 31; bit >= 0; --bit) {
				const unsigned t = U | (1u << bit);
				if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(hi < t)) < kk)
					U = t;
			}
		}
		int total = 0;
#pragma unroll
		for (int r = 0; r < R; ++r)
			total += __builtin_popcountll(__builtin_amdgcn_ballot_w64(kreg[r] != CB_EMPTY && (unsigned)(kreg[r] >> 32) <= U));
		if (total <= 256) {
			int S = 0;
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const bool take = kreg[r] != CB_EMPTY && (unsigned)(kreg[r] >> 32) <= U;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
				if (take)
					surv[S + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = kreg[r];
				S += __builtin_popcountll(m);
			}
			top[lane] = CB_EMPTY;
			__syncthreads();
			for (int p = lane; p < S; p += 64) {
				const unsigned long long me = surv[p];
				int rank = 0;
				for (int j = 0; j < S; ++j) { // (equal keys cannot occur: a row is a candidate once per query; ranked by position if they did)
					const unsigned long long o = surv[j];
					rank += (o < me || (o == me && j < p)) ? 1 : 0;
				}
				if (rank < kk)
					top[rank] = me;
			}
			__syncthreads();
			const unsigned long long res = lane < kk ? top[lane] : CB_EMPTY;
			const unsigned long long w = top[kk - 1];
			__syncthreads(); // (surv / top are free again)
			if (ce == n)
				return res;
			seed_mine = res;
			seed_worst = w;
			start = ce;
		}
	}
	unsigned long long mine = seed_mine, worst = seed_worst;
	for (int base8 = start; base8 < n; base8 += 512) {
		unsigned long long k8[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			const int i = base8 + 64 * r + lane;
			k8[r] = i < n ? cb_load_key<COHERENT>(keys + i) : CB_EMPTY;
		}
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			const unsigned long long key = k8[r];
			unsigned long long pend = __builtin_amdgcn_ballot_w64(key < worst);
			while (pend != 0ull) {
				const int L = __builtin_ctzll(pend);
				pend &= pend - 1ull;
				const unsigned long long ck = cb_lane64(key, L);
				if (ck >= worst)
					continue;
				const int pos = __popcll(__builtin_amdgcn_ballot_w64(lane < kk && mine <= ck));
				const unsigned long long up = cb_shr1(mine);
				if (lane == pos)
					mine = ck;
				else if (lane > pos && lane < kk)
					mine = up;
				worst = cb_lane64(mine, kk - 1);
			}
		}
	}
	return mine;
}

} // namespace mvs
