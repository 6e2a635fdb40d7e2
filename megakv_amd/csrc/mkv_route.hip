// mkv_route.hip -- request routing for the sharded (multi-GPU) index.
//
// New design, not in the reference (which is single-GPU).  The
// table splits into closed shards because an entry's alternate bucket keeps
// the top bits of its primary bucket index (libgpuhash/gpu_hash.cu:66-67), so
// a request is served entirely by the shard  (hash & hash_mask_agg) >> shift.
// Before the RCCL all-to-all each rank groups its local batch by destination
// shard; this file does that grouping on the device, STABLY (insert order
// inside a shard is part of the result definition), and the inverse scatter
// for the returned location pairs.
//
// Three short kernels: per-wave histogram -> one-block exclusive scan over
// (shard-major, wave-minor) -> stable scatter, ranks from wave ballots.
#include "mkv_common.h"

namespace mkv {

constexpr int kRouteChunk = 1024;   // requests per wave
constexpr int kMaxShards = 64;

__device__ __forceinline__ uint32_t shard_of(const uint32_t *req, uint32_t hash_mask, uint32_t shift)
{
	return (req[1] & hash_mask) >> shift;
}

template <int WORDS>
__global__ __launch_bounds__(256) void k_route_count(const uint32_t *__restrict__ in, int64_t n,
		uint32_t hash_mask, uint32_t shift, int nshards, uint32_t *__restrict__ wave_counts,
		int64_t nwaves)
{
	const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (wave >= nwaves)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	const int64_t lo = wave * kRouteChunk;
	uint32_t mine = 0;      // lane s (< nshards) accumulates shard s
	for (int it = 0; it < kRouteChunk / 64; it++) {
		const int64_t e = lo + it * 64 + lane;
		const uint32_t s = e < n ? shard_of(in + e * WORDS, hash_mask, shift) : 0xffffffffu;
		for (int k = 0; k < nshards; k++) {
			const uint32_t c = (uint32_t)__popcll(__ballot(s == (uint32_t)k));
			if (lane == (uint32_t)k)
				mine += c;
		}
	}
	if (lane < (uint32_t)nshards)
		wave_counts[(int64_t)lane * nwaves + wave] = mine;
}

// exclusive scan of wave_counts in (shard-major, wave-minor) order, in place;
// also emits per-shard totals.  One block.
__global__ __launch_bounds__(1024) void k_route_scan(uint32_t *wave_counts, int64_t total_entries,
		int64_t nwaves, int nshards, int64_t *counts)
{
	__shared__ uint32_t part[1024];
	__shared__ uint32_t carry;
	const uint32_t t = threadIdx.x;
	if (t == 0)
		carry = 0;
	__syncthreads();
	for (int64_t base = 0; base < total_entries; base += 1024) {
		const int64_t i = base + t;
		const uint32_t v = i < total_entries ? wave_counts[i] : 0u;
		part[t] = v;
		__syncthreads();
		for (uint32_t off = 1; off < 1024; off <<= 1) {
			const uint32_t add = t >= off ? part[t - off] : 0u;
			__syncthreads();
			part[t] += add;
			__syncthreads();
		}
		const uint32_t incl = part[t];
		if (i < total_entries)
			wave_counts[i] = carry + incl - v;
		__syncthreads();
		if (t == 1023)
			carry += incl;
		__syncthreads();
	}
	// totals per shard = start(shard+1) - start(shard)
	if (t < (uint32_t)nshards) {
		const uint32_t start = wave_counts[(int64_t)t * nwaves];
		const uint32_t end = (int)t + 1 < nshards ? wave_counts[(int64_t)(t + 1) * nwaves] : carry;
		counts[t] = (int64_t)(end - start);
	}
}

template <int WORDS>
__global__ __launch_bounds__(256) void k_route_scatter(const uint32_t *__restrict__ in, int64_t n,
		uint32_t hash_mask, uint32_t shift, int nshards, const uint32_t *__restrict__ wave_offsets,
		int64_t nwaves, uint32_t *__restrict__ out, int32_t *__restrict__ perm)
{
	const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (wave >= nwaves)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	const int64_t lo = wave * kRouteChunk;
	// lane s keeps the running output cursor of shard s
	uint32_t cursor = lane < (uint32_t)nshards ? wave_offsets[(int64_t)lane * nwaves + wave] : 0u;
	for (int it = 0; it < kRouteChunk / 64; it++) {
		const int64_t e = lo + it * 64 + lane;
		const bool live = e < n;
		uint32_t w0 = 0, w1 = 0, w2 = 0;
		if (live) {
			w0 = in[e * WORDS];
			w1 = in[e * WORDS + 1];
			if (WORDS == 3)
				w2 = in[e * WORDS + 2];
		}
		const uint32_t s = live ? (w1 & hash_mask) >> shift : 0xffffffffu;
		uint32_t dst = 0;
		for (int k = 0; k < nshards; k++) {
			const unsigned long long m = __ballot(s == (uint32_t)k);
			const uint32_t base = __shfl(cursor, k);
			if (s == (uint32_t)k)
				dst = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
			if (lane == (uint32_t)k)
				cursor += (uint32_t)__popcll(m);
		}
		if (live) {
			out[(int64_t)dst * WORDS] = w0;
			out[(int64_t)dst * WORDS + 1] = w1;
			if (WORDS == 3)
				out[(int64_t)dst * WORDS + 2] = w2;
			perm[dst] = (int32_t)e;
		}
	}
}

__global__ __launch_bounds__(256) void k_unroute(const uint2 *__restrict__ routed,
		const int32_t *__restrict__ perm, int64_t n, uint2 *__restrict__ out)
{
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
			j += (int64_t)gridDim.x * blockDim.x)
		out[perm[j]] = routed[j];
}

}  // namespace mkv

extern "C" int mkv_route_requests(const void *in, int elem_words, int64_t num_elem,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, void *routed_out,
		int32_t *perm, int64_t *counts, void *scratch, int64_t scratch_bytes, void *stream)
{
	if ((elem_words != 2 && elem_words != 3) || num_elem < 0 || num_elem > 0x7fffffffLL ||
			nshards < 1 || nshards > mkv::kMaxShards || shard_shift > 31)
		return MKV_EINVAL;
	if (!counts)
		return MKV_EINVAL;
	hipStream_t s = (hipStream_t)stream;
	if (num_elem == 0)
		return hipMemsetAsync(counts, 0, sizeof(int64_t) * nshards, s) == hipSuccess ? MKV_OK : MKV_EHIP;
	if (!in || !routed_out || !perm || !scratch)
		return MKV_EINVAL;
	const int64_t nwaves = (num_elem + mkv::kRouteChunk - 1) / mkv::kRouteChunk;
	const int64_t entries = nwaves * nshards;
	if (scratch_bytes < entries * 4)
		return MKV_ETOOBIG;
	uint32_t *wc = reinterpret_cast<uint32_t *>(scratch);
	const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in);
	uint32_t *out32 = reinterpret_cast<uint32_t *>(routed_out);
	dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
	if (elem_words == 2)
		hipLaunchKernelGGL((mkv::k_route_count<2>), grid, block, 0, s, in32, num_elem, hash_mask_agg,
				shard_shift, nshards, wc, nwaves);
	else
		hipLaunchKernelGGL((mkv::k_route_count<3>), grid, block, 0, s, in32, num_elem, hash_mask_agg,
				shard_shift, nshards, wc, nwaves);
	hipLaunchKernelGGL(mkv::k_route_scan, dim3(1), dim3(1024), 0, s, wc, entries, nwaves, nshards, counts);
	if (elem_words == 2)
		hipLaunchKernelGGL((mkv::k_route_scatter<2>), grid, block, 0, s, in32, num_elem, hash_mask_agg,
				shard_shift, nshards, wc, nwaves, out32, perm);
	else
		hipLaunchKernelGGL((mkv::k_route_scatter<3>), grid, block, 0, s, in32, num_elem, hash_mask_agg,
				shard_shift, nshards, wc, nwaves, out32, perm);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" int mkv_unroute_results(const loc_t *routed, const int32_t *perm, int64_t num_elem,
		loc_t *out, void *stream)
{
	if (num_elem < 0)
		return MKV_EINVAL;
	if (num_elem == 0)
		return MKV_OK;
	if (!routed || !perm || !out)
		return MKV_EINVAL;
	int64_t blocks = (num_elem + 255) / 256;
	if (blocks > 2048)
		blocks = 2048;
	hipLaunchKernelGGL(mkv::k_unroute, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
			reinterpret_cast<const uint2 *>(routed), perm, num_elem, reinterpret_cast<uint2 *>(out));
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}
