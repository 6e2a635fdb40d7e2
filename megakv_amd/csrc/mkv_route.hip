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
#include "mkv_search_body.h"

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

// ===========================================================================
// Padded exchange format (no host synchronisation anywhere in a step)
//
// Each rank packs its local batch into `nshards` equal slots, one per peer:
//     slot s = [ n_get, n_set, 0, 0 | cap_get x selem_t | cap_set x ielem_t ]
// Requests keep their batch order inside a slot (stable), so after the
// all-to-all the canonical insert order of a shard is (source rank, original
// index).  Equal slots mean the collective needs no size exchange and no
// device->host copy; the counts travel in the slot header and are consumed on
// the device (segmented search, insert block counts).  A slot that would
// overflow raises a flag the caller checks at the end of a run.
// ===========================================================================
namespace mkv {

constexpr int kHdrWords = 4;

// wave w < nw_get counts gets, else sets; counts[(kind*nshards + shard) * nw_kind + wave]
__global__ __launch_bounds__(256) void k_pack_count(const uint32_t *__restrict__ gets, int64_t n_get,
		const uint32_t *__restrict__ sets, int64_t n_set, uint32_t hash_mask, uint32_t shift, int nshards,
		uint32_t *__restrict__ wc_get, int64_t nw_get, uint32_t *__restrict__ wc_set, int64_t nw_set)
{
	int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const bool is_set = wave >= nw_get;
	if (is_set)
		wave -= nw_get;
	if (is_set && wave >= nw_set)
		return;
	const uint32_t *in = is_set ? sets : gets;
	const int words = is_set ? 3 : 2;
	const int64_t n = is_set ? n_set : n_get;
	const int64_t nw = is_set ? nw_set : nw_get;
	uint32_t *wc = is_set ? wc_set : wc_get;
	const uint32_t lane = threadIdx.x & 63u;
	const int64_t lo = wave * kRouteChunk;
	uint32_t mine = 0;
	for (int it = 0; it < kRouteChunk / 64; it++) {
		const int64_t e = lo + it * 64 + lane;
		const uint32_t s = e < n ? (in[e * words + 1] & hash_mask) >> shift : 0xffffffffu;
		for (int k = 0; k < nshards; k++) {
			const uint32_t c = (uint32_t)__popcll(__ballot(s == (uint32_t)k));
			if (lane == (uint32_t)k)
				mine += c;
		}
	}
	if (lane < (uint32_t)nshards)
		wc[(int64_t)lane * nw + wave] = mine;
}

// one block of 16 waves: wave w scans the (kind, shard) segments w, w+16, ...
// (exclusive scan over that segment's per-wave counts), writes the slot header
// and raises the overflow flag
__global__ __launch_bounds__(1024) void k_pack_scan(uint32_t *wc_get, int64_t nw_get, uint32_t *wc_set,
		int64_t nw_set, int nshards, uint32_t *sendbuf, int64_t slot_words, uint32_t cap_get,
		uint32_t cap_set, int *overflow)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	for (uint32_t seg = wave; seg < 2u * (uint32_t)nshards; seg += blockDim.x >> 6) {
		const int kind = seg >= (uint32_t)nshards;
		const int s = (int)(kind ? seg - (uint32_t)nshards : seg);
		const int64_t nw = kind ? nw_set : nw_get;
		uint32_t *p = (kind ? wc_set : wc_get) + (int64_t)s * nw;
		uint32_t carry = 0;
		for (int64_t base = 0; base < nw; base += 64) {
			const int64_t i = base + lane;
			const uint32_t v = i < nw ? p[i] : 0u;
			uint32_t incl = v;
			for (int off = 1; off < 64; off <<= 1) {
				const uint32_t o = __shfl_up(incl, off);
				if ((int)lane >= off)
					incl += o;
			}
			if (i < nw)
				p[i] = carry + incl - v;
			carry += __shfl(incl, 63);
		}
		if (lane == 0) {
			const uint32_t cap = kind ? cap_set : cap_get;
			uint32_t *hdr = sendbuf + (int64_t)s * slot_words;
			if (carry > cap) {
				*overflow = 1;
				hdr[kind] = 0;             // nothing of this kind is sent for this slot
			} else {
				hdr[kind] = carry;
			}
			if (kind == 0) {
				hdr[2] = 0;
				hdr[3] = 0;
			}
		}
	}
}

__global__ __launch_bounds__(256) void k_pack_scatter(const uint32_t *__restrict__ gets, int64_t n_get,
		const uint32_t *__restrict__ sets, int64_t n_set, uint32_t hash_mask, uint32_t shift, int nshards,
		const uint32_t *__restrict__ wc_get, int64_t nw_get, const uint32_t *__restrict__ wc_set,
		int64_t nw_set, uint32_t *__restrict__ sendbuf, int64_t slot_words, uint32_t cap_get,
		uint32_t cap_set, int32_t *__restrict__ perm_get)
{
	int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const bool is_set = wave >= nw_get;
	if (is_set)
		wave -= nw_get;
	if (is_set && wave >= nw_set)
		return;
	const uint32_t *in = is_set ? sets : gets;
	const int words = is_set ? 3 : 2;
	const int64_t n = is_set ? n_set : n_get;
	const int64_t nw = is_set ? nw_set : nw_get;
	const uint32_t *wc = is_set ? wc_set : wc_get;
	const uint32_t cap = is_set ? cap_set : cap_get;
	const int64_t region = is_set ? kHdrWords + (int64_t)cap_get * 2 : kHdrWords;
	const uint32_t lane = threadIdx.x & 63u;
	const int64_t lo = wave * kRouteChunk;
	// lane s keeps the running within-slot cursor of shard s; a slot that overflowed
	// (header count 0 while requests exist) drops its requests, the flag is already set
	uint32_t cursor = 0;
	bool slot_ok = true;
	if (lane < (uint32_t)nshards) {
		cursor = wc[(int64_t)lane * nw + wave];
	}
	for (int it = 0; it < kRouteChunk / 64; it++) {
		const int64_t e = lo + it * 64 + lane;
		const bool live = e < n;
		uint32_t w0 = 0, w1 = 0, w2 = 0;
		if (live) {
			w0 = in[e * words];
			w1 = in[e * words + 1];
			if (is_set)
				w2 = in[e * words + 2];
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
		(void)slot_ok;
		if (live && dst < cap) {
			uint32_t *p = sendbuf + (int64_t)s * slot_words + region + (int64_t)dst * words;
			p[0] = w0;
			p[1] = w1;
			if (is_set)
				p[2] = w2;
		}
		if (live && !is_set)
			perm_get[e] = dst < cap ? (int32_t)((int64_t)s * cap_get + dst) : -1;   // where request e went
	}
}

// ---- one-launch pack ----------------------------------------------------------
// Work-groups of 1024 lanes.  The first `set_tiles` take the SETs, 4096 each, and keep batch
// order: per-tile counts, one meeting of the SET tiles (they are the first work-groups of the
// grid and all resident: the host checks), prefix over the earlier tiles, ranked scatter.  The
// others take the GETs, 8192 (4096) each; the order inside a GET slot is free -- `where` brings
// every result home -- so a tile ranks its requests per shard with wave ballots (one pass),
// takes its range of every slot with ONE atomic per shard (the cursors lie in different
// 128-byte lines; 122 work-groups instead of the 512 of the first version queue on them),
// gathers the requests shard by shard in LDS and writes each shard's run as one contiguous,
// coalesced piece.  (The un-pack of the results stays a gather through `where`: reading the runs
// back tile by tile through LDS was built and measured -- 8.0 / 10.0 us against 7.8 / 7.7 us at
// 1 / 8 shards: two dependent round trips either way.)
// The last work-group to finish writes the slot headers and clears the control words.
// (SETs per work-group: 4 per lane -- three 16-byte loads)
constexpr int kPackBins = 16;        // shards the one-launch pack handles
constexpr int kCursorStride = 32;    // words between two shards' cursors (a 128-byte line each)
constexpr int kPackArrive = kPackBins * kCursorStride, kPackDone = kPackArrive + kCursorStride;
constexpr int kPackCtlWords = kPackDone + kCursorStride;    // cursors, arrive, done; then the SET tiles' counts
constexpr int64_t kTileR8From = 1 << 19;                    // batches from this many GETs on: 8 per lane, else 4
constexpr int kPackMetaWords = 2 * kPackBins + 1;           // per GET tile: s_off[0 .. kPackBins], s_base[0 .. kPackBins)

// lanes per work-group of the one-launch pack (tuning "pack_threads": 256, 512 or 1024)
static inline int pack_threads()
{
	const int t = tuning().pack_threads;
	return t == 256 || t == 512 ? t : 1024;
}

__device__ __forceinline__ uint32_t pack_ld(const uint32_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Ranks of a wave's requests inside their shards, in the order (lane, then the lane's own
// requests): every lane counts its requests per shard in 16-bit fields of 64-bit words (four
// shards a word), the words are prefix-summed over the wave (six shuffle steps per word
// instead of a ballot per shard and request: 512 vector instructions at 8 shards x 8 requests,
// 4 cycles each on a 16-lane SIMD), and a lane numbers its own requests on top of what the lanes
// before it hold.  mine (lane k): the wave's count for shard k.
template <int R>
__device__ __forceinline__ void wave_rank(const uint32_t (&sh)[R], uint32_t G, uint32_t (&rank)[R], uint32_t &mine)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t nw = (G + 3u) >> 2;                  // words in use (uniform)
	unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
	for (int j = 0; j < R; j++) {
		const unsigned long long inc = sh[j] != 0xffffffffu ? 1ull << (16u * (sh[j] & 3u)) : 0ull;
		const uint32_t w = (sh[j] >> 2) & 3u;
#pragma unroll
		for (int q = 0; q < 4; q++)
			pc[q] += w == (uint32_t)q ? inc : 0ull;
	}
	unsigned long long ex[4] = {0ull, 0ull, 0ull, 0ull}, tot[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
	for (int q = 0; q < 4; q++) {
		if ((uint32_t)q < nw) {
			unsigned long long incl = pc[q];
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const unsigned long long o = __shfl_up(incl, off);
				if ((int)lane >= off)
					incl += o;
			}
			ex[q] = incl - pc[q];
			tot[q] = __shfl(incl, 63);
		}
	}
#pragma unroll
	for (int j = 0; j < R; j++) {
		const uint32_t w = (sh[j] >> 2) & 3u, f = 16u * (sh[j] & 3u);
		const unsigned long long x = w == 0u ? ex[0] : w == 1u ? ex[1] : w == 2u ? ex[2] : ex[3];
		rank[j] = (uint32_t)(x >> f) & 0xffffu;
		const unsigned long long inc = sh[j] != 0xffffffffu ? 1ull << f : 0ull;
#pragma unroll
		for (int q = 0; q < 4; q++)
			ex[q] += w == (uint32_t)q ? inc : 0ull;
	}
	const uint32_t kw = (lane >> 2) & 3u;
	const unsigned long long t = kw == 0u ? tot[0] : kw == 1u ? tot[1] : kw == 2u ? tot[2] : tot[3];
	mine = lane < G ? (uint32_t)(t >> (16u * (lane & 3u))) & 0xffffu : 0u;
}

template <int R, int THREADS>      // GETs per lane: 8 or 4; lanes per work-group: 1024, 512 or 256
__global__ __launch_bounds__(THREADS) void k_pack_tiles(const uint32_t *__restrict__ gets, int64_t n_get,
		const uint32_t *__restrict__ sets, int64_t n_set, uint32_t hash_mask, uint32_t shift, int nshards,
		uint32_t *__restrict__ sendbuf, int64_t slot_words, uint32_t cap_get, uint32_t cap_set,
		int32_t *__restrict__ perm_get, uint32_t *ctl, int *overflow, uint32_t set_tiles, uint32_t get_tiles,
		uint32_t self, uint32_t *__restrict__ self_slot, uint32_t *__restrict__ tile_meta)
{
	// (tile_meta: where every GET tile's runs went -- per tile kPackMetaWords words: the runs' places in the tile's
	// staging order, s_off[0 .. kPackBins], then their first places in the slots, s_base[0 .. kPackBins) -- for the
	// un-pack that brings a tile's results back run by run, k_unpack_tiled; null: not kept)
	// (self_slot: the slot of shard `self` is written THERE -- the rank's own receive buffer: what a
	// rank sends to itself need not pass through the exchange; self >= nshards: none)
	constexpr int kTile = THREADS * R, kSetTile = THREADS * 4, NW = THREADS / 64;
	__shared__ uint2 s_stage[kTile];
	__shared__ uint32_t s_wcnt[NW][kPackBins];
	__shared__ uint32_t s_base[kPackBins], s_off[kPackBins + 1];
	__shared__ uint32_t s_last;
	const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
	const uint32_t G = (uint32_t)nshards;
	uint32_t *cursors = ctl, *arrive = ctl + kPackArrive, *done = ctl + kPackDone;
	uint32_t *tile_counts = ctl + kPackCtlWords;
	const bool sets16 = (reinterpret_cast<uintptr_t>(sets) & 15u) == 0u;
	const bool gets16 = (reinterpret_cast<uintptr_t>(gets) & 15u) == 0u;
	const bool perm8 = (reinterpret_cast<uintptr_t>(perm_get) & 7u) == 0u;
	if (blockIdx.x < set_tiles) {
		// ---- SETs: lane L of wave w holds the four consecutive requests 256 w + 4 L .. + 3 of the tile
		const uint32_t tile = blockIdx.x;
		const int64_t e0 = (int64_t)tile * kSetTile + (int64_t)wave * 256 + (int64_t)lane * 4;
		uint32_t w[12], sh[4];
		if (e0 + 4 <= n_set && sets16) {
			const uint4 *p = reinterpret_cast<const uint4 *>(sets + e0 * 3);
			const uint4 a = p[0], b = p[1], c = p[2];
			w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
			w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
		} else {
#pragma unroll
			for (int c = 0; c < 4; c++)
#pragma unroll
				for (int j = 0; j < 3; j++)
					w[3 * c + j] = e0 + c < n_set ? sets[(e0 + c) * 3 + j] : 0u;
		}
#pragma unroll
		for (int c = 0; c < 4; c++)
			sh[c] = e0 + c < n_set ? (w[3 * c + 1] & hash_mask) >> shift : 0xffffffffu;
		// rank inside the wave, in request order (lane-major, then c)
		uint32_t rank[4];
		uint32_t mine;                         // lane k: this wave's count for shard k
		wave_rank<4>(sh, G, rank, mine);
		if (lane < kPackBins)
			s_wcnt[wave][lane] = lane < G ? mine : 0u;
		if (t < kPackBins)
			s_base[t] = 0;
		__syncthreads();
		// counts and the meeting go through agent-scope atomics only: no cache write-back
		// (an agent-scope fence flushes the L2 the GET tiles are filling -- 100 us of them)
		if (t < G) {
			uint32_t c = 0;
			for (int wv = 0; wv < NW; wv++)
				c += s_wcnt[wv][t];
			__hip_atomic_store(&tile_counts[(size_t)tile * kPackBins + t], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// (the stores have LEFT before the barrier: a work-group barrier alone does not wait for them -- its
		// release is of work-group scope -- and lane 0 signals for all of them)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (t == 0) {
			atomicAdd(arrive, 1u);
			while (pack_ld(arrive) < set_tiles)
				__builtin_amdgcn_s_sleep(1);
		}
		__syncthreads();
		for (uint32_t j = t; j < tile * G; j += blockDim.x)
			atomicAdd(&s_base[j % G], pack_ld(&tile_counts[(size_t)(j / G) * kPackBins + j % G]));
		__syncthreads();
		uint32_t cursor = 0;                   // lane k: where this wave's requests of shard k start in the slot
		if (lane < G) {
			cursor = s_base[lane];
			for (uint32_t wv = 0; wv < wave; wv++)
				cursor += s_wcnt[wv][lane];
		}
		const int64_t region = kHdrWords + (int64_t)cap_get * 2;
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const uint32_t base = __shfl(cursor, (int)(sh[c] & 63u));
			const uint32_t dst = base + rank[c];
			if (sh[c] != 0xffffffffu && dst < cap_set) {
				uint32_t *p = (sh[c] == self ? self_slot : sendbuf + (int64_t)sh[c] * slot_words) + region + (int64_t)dst * 3;
				p[0] = w[3 * c];
				p[1] = w[3 * c + 1];
				p[2] = w[3 * c + 2];
			}
		}
		if (tile == set_tiles - 1u && t < G) {
			uint32_t total = s_base[t];
			for (int wv = 0; wv < NW; wv++)
				total += s_wcnt[wv][t];
			uint32_t *hdr = t == self ? self_slot : sendbuf + (int64_t)t * slot_words;
			if (total > cap_set) {
				*overflow = 1;
				hdr[1] = 0;                    // nothing of this kind is sent for this slot
			} else {
				hdr[1] = total;
			}
		}
	} else if (blockIdx.x < set_tiles + get_tiles) {
		// ---- GETs: lane t holds the request pairs {2 (i 1024 + t), + 1} of the tile, i < R / 2
		const uint32_t tile = blockIdx.x - set_tiles;
		const int64_t tile0 = (int64_t)tile * kTile;
		uint2 rq[R];
		uint32_t sh[R], rank[R];
#pragma unroll
		for (int i = 0; i < R / 2; i++) {
			const int64_t e = tile0 + 2 * ((int64_t)i * THREADS + t);
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if (e + 2 <= n_get && gets16) {
				v = *reinterpret_cast<const uint4 *>(gets + e * 2);
			} else if (e < n_get) {
				const uint2 h = *reinterpret_cast<const uint2 *>(gets + e * 2);
				v.x = h.x;
				v.y = h.y;
				if (e + 1 < n_get) {
					const uint2 h2 = *reinterpret_cast<const uint2 *>(gets + e * 2 + 2);
					v.z = h2.x;
					v.w = h2.y;
				}
			}
			rq[2 * i] = make_uint2(v.x, v.y);
			rq[2 * i + 1] = make_uint2(v.z, v.w);
			sh[2 * i] = e < n_get ? (v.y & hash_mask) >> shift : 0xffffffffu;
			sh[2 * i + 1] = e + 1 < n_get ? (v.w & hash_mask) >> shift : 0xffffffffu;
		}
		// ranks inside the wave per shard (any order will do), counts in lane k
		uint32_t mine;
		wave_rank<R>(sh, G, rank, mine);
		if (lane < kPackBins)
			s_wcnt[wave][lane] = lane < G ? mine : 0u;
		__syncthreads();
		// the work-group's count per shard -> its range of the slot (one atomic per shard), the
		// shard's place in the LDS staging area, and where the tile went (for the un-pack)
		if (t < kPackBins) {
			uint32_t c = 0;
			for (int wv = 0; wv < NW; wv++)
				c += s_wcnt[wv][t];
			const uint32_t base = (t < G && c) ? atomicAdd(&cursors[t * kCursorStride], c) : 0u;
			s_base[t] = base;
			// exclusive prefix of the counts over the shards (16 lanes of one wave)
			uint32_t incl = c;
			for (int off = 1; off < kPackBins; off <<= 1) {
				const uint32_t o = __shfl_up(incl, off);
				if ((int)t >= off)
					incl += o;
			}
			s_off[t] = incl - c;
			if (t == kPackBins - 1)
				s_off[kPackBins] = incl;
		}
		__syncthreads();
		if (tile_meta && t <= kPackBins) {
			uint32_t *m = tile_meta + (size_t)tile * kPackMetaWords;
			m[t] = s_off[t];
			if (t < kPackBins)
				m[kPackBins + 1 + t] = s_base[t];
		}
		uint32_t wbase = 0;                    // lane k: this wave's first place inside shard k's run of the tile
		if (lane < G)
			for (uint32_t wv = 0; wv < wave; wv++)
				wbase += s_wcnt[wv][lane];
		int32_t wh[R];
#pragma unroll
		for (int j = 0; j < R; j++) {
			const uint32_t k = sh[j] & 63u;
			const uint32_t in_run = __shfl(wbase, (int)k) + rank[j];      // place inside the tile's run for shard k
			wh[j] = -1;
			if (sh[j] != 0xffffffffu) {
				s_stage[s_off[k] + in_run] = rq[j];
				const uint32_t dst = s_base[k] + in_run;
				if (dst < cap_get)
					wh[j] = (int32_t)((int64_t)k * cap_get + dst);
			}
		}
		// where the requests went (their results come back to the same place): request order, coalesced
#pragma unroll
		for (int i = 0; i < R / 2; i++) {
			const int64_t e = tile0 + 2 * ((int64_t)i * THREADS + t);
			if (e + 2 <= n_get && perm8) {
				*reinterpret_cast<int2 *>(perm_get + e) = make_int2(wh[2 * i], wh[2 * i + 1]);
			} else if (e < n_get) {
				perm_get[e] = wh[2 * i];
				if (e + 1 < n_get)
					perm_get[e + 1] = wh[2 * i + 1];
			}
		}
		__syncthreads();
		// every shard's run leaves as one contiguous piece
		for (uint32_t k = 0; k < G; k++) {
			const uint32_t lo = s_off[k], hi = s_off[k + 1u], base = s_base[k];
			uint2 *slot = reinterpret_cast<uint2 *>((k == self ? self_slot : sendbuf + (int64_t)k * slot_words) + kHdrWords);
			for (uint32_t j = lo + t; j < hi; j += THREADS) {
				const uint32_t dst = base + (j - lo);
				if (dst < cap_get)
					slot[dst] = s_stage[j];
			}
		}
	}
	__syncthreads();
	if (t == 0)
		s_last = atomicAdd(done, 1u) == gridDim.x - 1u ? 1u : 0u;
	__syncthreads();
	if (s_last) {
		// (the cursors are atomics; the slots' contents need not be visible here)
		if (t < G) {
			const uint32_t c = pack_ld(&cursors[t * kCursorStride]);
			uint32_t *hdr = t == self ? self_slot : sendbuf + (int64_t)t * slot_words;
			if (c > cap_get) {
				*overflow = 1;
				hdr[0] = 0;
			} else {
				hdr[0] = c;
			}
			if (set_tiles == 0u)
				hdr[1] = 0;
			hdr[2] = 0;
			hdr[3] = 0;
		}
		__syncthreads();
		if (t < kPackBins)
			cursors[t * kCursorStride] = 0;
		if (t == 0) {
			*arrive = 0;
			*done = 0;
		}
	}
}

// segmented search over received slots; block (seg, chunk).  Block 0 also
// publishes the per-source insert counts for the insert that follows.
__global__ __launch_bounds__(256) void k_search_slots(const uint32_t *__restrict__ recvbuf,
		int64_t slot_words, uint32_t cap_get, uint32_t *__restrict__ out, const uint4 *__restrict__ table,
		uint32_t hash_mask, uint32_t blk_mask, uint32_t chunks_per_slot, int nslots, int *set_counts)
{
	if (blockIdx.x == 0 && threadIdx.x < (uint32_t)nslots)
		set_counts[threadIdx.x] = (int)recvbuf[(int64_t)threadIdx.x * slot_words + 1];
	search_slots_body(blockIdx.x, recvbuf, slot_words, cap_get, out, table, hash_mask, blk_mask,
			chunks_per_slot, 0xffffffffu, nullptr);
}

// results come back in the layout the requests were sent in; request e went to position
// where[e] (slot * cap_get + rank), so its result is gathered from there: coalesced writes,
// 8-byte gathers from a buffer the exchange has just written
__global__ __launch_bounds__(256) void k_unpack_results(const uint2 *__restrict__ back,
		const int32_t *__restrict__ where, int64_t n_get, uint2 *__restrict__ out)
{
	// two requests per lane: one 8-byte load of their `where` entries, two gathers, one 16-byte store
	const bool wide = ((reinterpret_cast<uintptr_t>(where) & 7u) | (reinterpret_cast<uintptr_t>(out) & 15u)) == 0u;
	for (int64_t e = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); e < n_get;
			e += 2 * (int64_t)gridDim.x * blockDim.x) {
		if (wide && e + 2 <= n_get) {
			const int2 w = *reinterpret_cast<const int2 *>(where + e);
			const uint2 a = w.x >= 0 ? back[w.x] : make_uint2(0u, 0u);
			const uint2 b = w.y >= 0 ? back[w.y] : make_uint2(0u, 0u);
			*reinterpret_cast<uint4 *>(out + e) = make_uint4(a.x, a.y, b.x, b.y);
		} else {
			const int32_t w0 = where[e];
			out[e] = w0 >= 0 ? back[w0] : make_uint2(0u, 0u);
			if (e + 1 < n_get) {
				const int32_t w1 = where[e + 1];
				out[e + 1] = w1 >= 0 ? back[w1] : make_uint2(0u, 0u);
			}
		}
	}
}

// The same by TILES of the pack (k_pack_tiles, tile_meta): a tile's requests went to the slots as one contiguous run
// per shard, so their results come back as one contiguous run per shard -- streamed into LDS, a 16-byte piece a lane,
// in the order the pack staged them -- and every request picks its result from there: no gather from global memory
// (k_unpack_results asks the memory side for 8 bytes per request from eight places at once, beside a search that
// is bound by the memory side's request rate: 7.5 us alone, 17.6 us in an eight-rank schedule).
template <int R, int THREADS>
__global__ __launch_bounds__(THREADS) void k_unpack_tiled(const uint2 *__restrict__ back, const int32_t *__restrict__ where,
		int64_t n_get, uint2 *__restrict__ out, const uint32_t *__restrict__ tile_meta, int nshards, uint32_t cap_get)
{
	constexpr int kTile = THREADS * R;
	__shared__ uint2 s_stage[kTile];
	__shared__ uint32_t s_off[kPackBins + 1], s_base[kPackBins];
	const uint32_t t = threadIdx.x, G = (uint32_t)nshards;
	const uint32_t tile = blockIdx.x;
	const int64_t tile0 = (int64_t)tile * kTile;
	const uint32_t *m = tile_meta + (size_t)tile * kPackMetaWords;
	if (t <= kPackBins)
		s_off[t] = m[t];
	if (t < kPackBins)
		s_base[t] = m[kPackBins + 1 + t];
	__syncthreads();
	for (uint32_t k = 0; k < G; k++) {
		const uint32_t lo = s_off[k], n = s_off[k + 1u] - lo, base = s_base[k];
		const uint2 *src = back + (size_t)k * cap_get + base;
		// (a run starts anywhere in its slot: the first lanes take single results up to a 16-byte boundary, then pairs)
		const uint32_t head = (uint32_t)((reinterpret_cast<uintptr_t>(src) >> 3) & 1u) & (n ? 1u : 0u);
		if (t < head)
			s_stage[lo] = base < cap_get ? src[0] : make_uint2(0u, 0u);
		const uint32_t pairs = (n - head) >> 1;
		const uint4 *src4 = reinterpret_cast<const uint4 *>(src + head);
		for (uint32_t j = t; j < pairs; j += THREADS) {
			const uint32_t at = head + 2u * j;
			uint4 v = make_uint4(0u, 0u, 0u, 0u);
			if (base + at + 1u < cap_get)
				v = src4[j];
			else if (base + at < cap_get)
				v = make_uint4(src[at].x, src[at].y, 0u, 0u);
			s_stage[lo + at] = make_uint2(v.x, v.y);
			s_stage[lo + at + 1u] = make_uint2(v.z, v.w);
		}
		if (t == 0 && ((n - head) & 1u)) {
			const uint32_t at = n - 1u;
			s_stage[lo + at] = base + at < cap_get ? src[at] : make_uint2(0u, 0u);
		}
	}
	__syncthreads();
	const bool wide = ((reinterpret_cast<uintptr_t>(where) & 7u) | (reinterpret_cast<uintptr_t>(out) & 15u)) == 0u;
#pragma unroll
	for (int i = 0; i < R / 2; i++) {
		const int64_t e = tile0 + 2 * ((int64_t)i * THREADS + t);
		if (e >= n_get)
			continue;
		const int32_t w0 = where[e], w1 = e + 1 < n_get ? where[e + 1] : -1;
		uint2 a = make_uint2(0u, 0u), b = a;
		if (w0 >= 0) {
			const uint32_t k = (uint32_t)w0 / cap_get;
			a = s_stage[s_off[k] + ((uint32_t)w0 - k * cap_get - s_base[k])];
		}
		if (w1 >= 0) {
			const uint32_t k = (uint32_t)w1 / cap_get;
			b = s_stage[s_off[k] + ((uint32_t)w1 - k * cap_get - s_base[k])];
		}
		if (wide && e + 2 <= n_get) {
			*reinterpret_cast<uint4 *>(out + e) = make_uint4(a.x, a.y, b.x, b.y);
		} else {
			out[e] = a;
			if (e + 1 < n_get)
				out[e + 1] = b;
		}
	}
}

}  // namespace mkv

extern "C" int64_t mkv_pack_slot_words(int64_t cap_get, int64_t cap_set)
{
	return mkv::kHdrWords + cap_get * 2 + cap_set * 3;
}

extern "C" int64_t mkv_pack_scratch_bytes(int64_t n_get, int64_t n_set, int nshards)
{
	const int64_t nw_get = (n_get + mkv::kRouteChunk - 1) / mkv::kRouteChunk;
	const int64_t nw_set = (n_set + mkv::kRouteChunk - 1) / mkv::kRouteChunk;
	const int64_t three_pass = (nw_get + nw_set) * nshards * 4 + 64;
	const int64_t set_tiles = (n_set + 1023) / 1024;        // (the smallest SET tile: 256 lanes x 4)
	const int64_t one_pass = (mkv::kPackCtlWords + set_tiles * mkv::kPackBins) * 4 + 64;
	return three_pass > one_pass ? three_pass : one_pass;
}

extern "C" int mkv_pack_requests_self(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, void *stream);

// Will the pack of a batch with n_set SETs (any number of GETs) over nshards shards be ONE launch
// (k_pack_tiles) -- the form that can write a rank's own slot elsewhere (mkv_pack_requests_self)?
// The SET tiles of that launch meet once (they spin on an arrival counter): every one of them must
// be resident at the same time.  They are the first work-groups of the grid, so it is enough that
// the device can hold that many -- asked once per device (a CU-masked or partitioned device holds
// fewer); beyond that, and with the tuning "pack_three_pass", the three-pass pack is used.  A batch
// with fewer SETs than one that fits also fits (the sharded engine settles its form once, for its
// largest batch).
extern "C" int mkv_pack_one_launch(int64_t n_set, int nshards, int64_t scratch_bytes)
{
	static int resident_cap[64][3] = {{0}};
	int dev = 0;
	(void)hipGetDevice(&dev);
	const int threads = mkv::pack_threads(), v = threads == 1024 ? 0 : threads == 512 ? 1 : 2;
	if (dev >= 0 && dev < 64 && resident_cap[dev][v] == 0) {
		int per_cu = 0, cus = 0;
		const hipError_t e = v == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mkv::k_pack_tiles<8, 1024>, 1024, 0) :
				v == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mkv::k_pack_tiles<8, 512>, 512, 0) :
				hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mkv::k_pack_tiles<8, 256>, 256, 0);
		if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
			per_cu = cus = 0;
		resident_cap[dev][v] = per_cu * cus > 0 ? per_cu * cus : -1;
	}
	const int64_t set_tile = threads * 4;
	const int64_t set_tiles = ((n_set > 0 ? n_set : 0) + set_tile - 1) / set_tile;
	const int64_t set_tiles_max = (dev >= 0 && dev < 64 && resident_cap[dev][v] > 0) ?
			(resident_cap[dev][v] < 1024 ? resident_cap[dev][v] : 1024) : 128;
	return nshards >= 1 && nshards <= mkv::kPackBins && set_tiles <= set_tiles_max && !MKV_DIAG(pack_three_pass) &&
			scratch_bytes >= (mkv::kPackCtlWords + set_tiles * mkv::kPackBins) * 4;
}

extern "C" int mkv_pack_requests(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		void *stream)
{
	return mkv_pack_requests_self(gets, n_get, sets, n_set, hash_mask_agg, shard_shift, nshards, sendbuf, cap_get,
			cap_set, perm_get, scratch, scratch_bytes, overflow, -1, nullptr, stream);
}

// The same with the slot of shard `self` written to self_slot instead of its place in sendbuf (the
// sharded engine: a rank's requests for its own shard go straight into its receive buffer).
// self < 0: none.  Not with the three-pass fall-back (MKV_EINVAL: the caller packs without).
extern "C" int mkv_pack_requests_meta(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, uint32_t *tile_meta, void *stream);

extern "C" int mkv_pack_requests_self(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, void *stream)
{
	return mkv_pack_requests_meta(gets, n_get, sets, n_set, hash_mask_agg, shard_shift, nshards, sendbuf, cap_get, cap_set,
			perm_get, scratch, scratch_bytes, overflow, self, self_slot, nullptr, stream);
}

// words of the tile_meta array of mkv_pack_requests_meta / mkv_unpack_results_tiled for batches of up to n_get GETs
extern "C" int64_t mkv_pack_meta_words(int64_t n_get)
{
	return ((n_get > 0 ? n_get : 0) / 1024 + 2) * mkv::kPackMetaWords;      // (the smallest GET tile: 256 lanes x 4)
}

// The same; with tile_meta (mkv_pack_meta_words(n_get) words, or NULL) the one-launch pack notes where every tile's
// runs went, for mkv_unpack_results_tiled.  tile_meta is only written when the pack is ONE launch
// (mkv_pack_one_launch): a caller that wants the tiled un-pack asks that first.
extern "C" int mkv_pack_requests_meta(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, uint32_t *tile_meta, void *stream)
{
	if (self >= nshards || (self >= 0 && !self_slot))
		return MKV_EINVAL;
	if (n_get < 0 || n_set < 0 || nshards < 1 || nshards > mkv::kMaxShards || shard_shift > 31 ||
			cap_get < 0 || cap_set < 0 || cap_get > 0x7fffffffLL || cap_set > 0x7fffffffLL || !sendbuf ||
			!scratch || !overflow || (n_get && (!gets || !perm_get)) || (n_set && !sets))
		return MKV_EINVAL;
	const int64_t nw_get = (n_get + mkv::kRouteChunk - 1) / mkv::kRouteChunk;
	const int64_t nw_set = (n_set + mkv::kRouteChunk - 1) / mkv::kRouteChunk;
	if (scratch_bytes < (nw_get + nw_set) * nshards * 4)
		return MKV_ETOOBIG;
	hipStream_t s = (hipStream_t)stream;
	uint32_t *wc_get = reinterpret_cast<uint32_t *>(scratch);
	uint32_t *wc_set = wc_get + nw_get * nshards;
	const int64_t slot_words = mkv_pack_slot_words(cap_get, cap_set);
	const int64_t waves = nw_get + nw_set;
	const uint32_t *g32 = reinterpret_cast<const uint32_t *>(gets);
	const uint32_t *s32 = reinterpret_cast<const uint32_t *>(sets);
	const bool r8 = n_get >= mkv::kTileR8From;
	const int threads = mkv::pack_threads();
	const int64_t get_tile = (int64_t)threads * (r8 ? 8 : 4), set_tile = (int64_t)threads * 4;
	const int64_t set_tiles = (n_set + set_tile - 1) / set_tile;
	const int64_t get_tiles = (n_get + get_tile - 1) / get_tile;
	if (mkv_pack_one_launch(n_set, nshards, scratch_bytes)) {
		// one launch; its control words (the head of `scratch`) are zero between calls: zeroed
		// by the caller once, cleared again by the kernel's last work-group
		const int64_t blocks = set_tiles + get_tiles ? set_tiles + get_tiles : 1;
#define MKV_PACK_LAUNCH(R, T) \
		hipLaunchKernelGGL((mkv::k_pack_tiles<R, T>), dim3((unsigned)blocks), dim3(T), 0, s, g32, n_get, s32, n_set, \
				hash_mask_agg, shard_shift, nshards, sendbuf, slot_words, (uint32_t)cap_get, (uint32_t)cap_set, perm_get, \
				reinterpret_cast<uint32_t *>(scratch), overflow, (uint32_t)set_tiles, (uint32_t)get_tiles, \
				self >= 0 ? (uint32_t)self : 0xffffffffu, self_slot, tile_meta)
		if (threads == 1024) {
			if (r8) MKV_PACK_LAUNCH(8, 1024); else MKV_PACK_LAUNCH(4, 1024);
		} else if (threads == 512) {
			if (r8) MKV_PACK_LAUNCH(8, 512); else MKV_PACK_LAUNCH(4, 512);
		} else {
			if (r8) MKV_PACK_LAUNCH(8, 256); else MKV_PACK_LAUNCH(4, 256);
		}
#undef MKV_PACK_LAUNCH
		return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
	}
	if (self >= 0)
		return MKV_EINVAL;
	if (waves > 0) {
		dim3 grid((unsigned)((waves + 3) / 4)), block(256);
		hipLaunchKernelGGL(mkv::k_pack_count, grid, block, 0, s, g32, n_get, s32, n_set, hash_mask_agg,
				shard_shift, nshards, wc_get, nw_get, wc_set, nw_set);
	}
	hipLaunchKernelGGL(mkv::k_pack_scan, dim3(1), dim3(1024), 0, s, wc_get, nw_get, wc_set, nw_set, nshards,
			sendbuf, slot_words, (uint32_t)cap_get, (uint32_t)cap_set, overflow);
	if (waves > 0) {
		dim3 grid((unsigned)((waves + 3) / 4)), block(256);
		hipLaunchKernelGGL(mkv::k_pack_scatter, grid, block, 0, s, g32, n_get, s32, n_set, hash_mask_agg,
				shard_shift, nshards, wc_get, nw_get, wc_set, nw_set, sendbuf, slot_words, (uint32_t)cap_get,
				(uint32_t)cap_set, perm_get);
	}
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" int mkv_search_slots(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots,
		int64_t cap_get, int64_t cap_set, loc_t *out, const bucket_t *table, int *set_counts, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || nslots < 1 || nslots > mkv::kMaxShards || cap_get < 0 ||
			cap_get > 0x7fffffffLL || !recvbuf || !out || !table || !set_counts)
		return MKV_EINVAL;
	const uint32_t chunks = (uint32_t)((cap_get + 255) / 256);
	const uint32_t blocks = (chunks ? chunks : 1u) * (uint32_t)nslots;
	hipLaunchKernelGGL(mkv::k_search_slots, dim3(blocks), dim3(256), 0, (hipStream_t)stream, recvbuf,
			mkv_pack_slot_words(cap_get, cap_set), (uint32_t)cap_get, out,
			reinterpret_cast<const uint4 *>(table), d.hash_mask, d.blk_mask, chunks ? chunks : 1u, nslots,
			set_counts);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" int mkv_unpack_results(const loc_t *back, const int32_t *where, int64_t n_get, loc_t *out,
		void *stream)
{
	if (n_get < 0 || (n_get && (!back || !where || !out)))
		return MKV_EINVAL;
	if (n_get == 0)
		return MKV_OK;
	int64_t blocks = (n_get + 511) / 512;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(mkv::k_unpack_results, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
			reinterpret_cast<const uint2 *>(back), where, n_get, reinterpret_cast<uint2 *>(out));
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

// the un-pack by the pack's tiles (k_unpack_tiled): tile_meta as mkv_pack_requests_meta of THIS batch left it (a pack
// in one launch, the same tuning "pack_threads" since), nshards and cap_get as packed
extern "C" int mkv_unpack_results_tiled(const loc_t *back, const int32_t *where, int64_t n_get, loc_t *out,
		const uint32_t *tile_meta, int nshards, int64_t cap_get, void *stream)
{
	if (n_get < 0 || (n_get && (!back || !where || !out || !tile_meta)) || nshards < 1 || nshards > mkv::kPackBins ||
			cap_get <= 0 || cap_get > 0x7fffffffLL)
		return MKV_EINVAL;
	if (n_get == 0)
		return MKV_OK;
	const bool r8 = n_get >= mkv::kTileR8From;
	const int threads = mkv::pack_threads();
	const int64_t get_tile = (int64_t)threads * (r8 ? 8 : 4);
	const int64_t tiles = (n_get + get_tile - 1) / get_tile;
	hipStream_t s = (hipStream_t)stream;
	const uint2 *b2 = reinterpret_cast<const uint2 *>(back);
	uint2 *o2 = reinterpret_cast<uint2 *>(out);
#define MKV_UNPACK_LAUNCH(R, T) \
	hipLaunchKernelGGL((mkv::k_unpack_tiled<R, T>), dim3((unsigned)tiles), dim3(T), 0, s, b2, where, n_get, o2, tile_meta, nshards, \
			(uint32_t)cap_get)
	if (threads == 1024) {
		if (r8) MKV_UNPACK_LAUNCH(8, 1024); else MKV_UNPACK_LAUNCH(4, 1024);
	} else if (threads == 512) {
		if (r8) MKV_UNPACK_LAUNCH(8, 512); else MKV_UNPACK_LAUNCH(4, 512);
	} else {
		if (r8) MKV_UNPACK_LAUNCH(8, 256); else MKV_UNPACK_LAUNCH(4, 256);
	}
#undef MKV_UNPACK_LAUNCH
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

// ===========================================================================
// Request preparation on the device: what Mega-KV's receiver
// does per packet on the CPU -- derive the 64-bit signature word from the key
// bytes and split it into (sig, hash) -- for a whole batch of fixed-stride keys.
//
//   sig64 = first 8 key bytes                         (src/mega_recv.c:349-350)
//   with `fold` (the reference's SIGNATURE build): XOR of all further 8-byte
//   words, the 1..7 tail bytes masked              (src/mega_recv.c:351-360)
//   hash  = sig64 >> 32,  sig = low 32 bits           (src/mega_recv.c:361-362)
//
// Binning the resulting insert requests into the 2^IBLOCK_P blocks is
// mkv_route_requests with (mask, shift) = (0xffffffff, 32-IBLOCK_P) for the
// receiver's rule (src/mega_recv.c:476-477) or (HASH_MASK, MEM_P-6-IBLOCK_P)
// for the region rule of libgpuhash/test/insert_test.c:117-124.
// ===========================================================================
namespace mkv {

__global__ __launch_bounds__(256) void k_derive(const uint8_t *__restrict__ keys, int64_t n, int key_len,
		int64_t key_stride, int fold, const uint32_t *__restrict__ locs, uint32_t *__restrict__ out,
		int out_words)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
			i += (int64_t)gridDim.x * blockDim.x) {
		const uint8_t *k = keys + i * key_stride;
		unsigned long long sig = 0;
		// keys are at least 8 bytes (src/mega_common.c:37); unaligned strides are read bytewise
		const bool aligned = (((uintptr_t)k) & 7u) == 0;
		auto word = [&](int off, int nbytes) {
			unsigned long long w = 0;
			if (aligned && nbytes == 8)
				return *reinterpret_cast<const unsigned long long *>(k + off);
			for (int b = 0; b < nbytes; b++)
				w |= (unsigned long long)k[off + b] << (8 * b);
			return w;
		};
		sig = word(0, 8);
		if (fold) {
			int off = 8;
			for (; off <= key_len - 8; off += 8)
				sig ^= word(off, 8);
			if (off < key_len)
				sig ^= word(off, key_len - off);     // tail bytes, masked (:357-360)
		}
		uint32_t *o = out + i * out_words;
		o[0] = (uint32_t)sig;                        // signature: low word
		o[1] = (uint32_t)(sig >> 32);                // hash: high word
		if (out_words == 3)
			o[2] = locs[i];
	}
}

}  // namespace mkv

extern "C" int mkv_derive_requests(const void *keys, int64_t num_keys, int key_len, int64_t key_stride,
		int fold, const loc_t *locs, void *out, int out_words, void *stream)
{
	if (num_keys < 0 || key_len < 8 || key_stride < key_len || (out_words != 2 && out_words != 3) ||
			(out_words == 3 && !locs))
		return MKV_EINVAL;
	if (num_keys == 0)
		return MKV_OK;
	if (!keys || !out)
		return MKV_EINVAL;
	int64_t blocks = (num_keys + 255) / 256;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(mkv::k_derive, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
			reinterpret_cast<const uint8_t *>(keys), num_keys, key_len, key_stride, fold,
			reinterpret_cast<const uint32_t *>(locs), reinterpret_cast<uint32_t *>(out), out_words);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

// ===========================================================================
// Result consumption on the device: what the reference's
// sender does with the pair a search returns --
//     loc = out[2i]; if (loc == 0) loc = out[2i+1];     (src/mega_send.c:408-414)
//     loc == 0  =>  miss                                 (src/mega_send.c:464-466)
// for a whole batch: one resolved location per request plus hit / miss totals
// (the counters the reference aggregates per sender thread, src/mega_scheduler.c:97-116).
// With KEY_MATCH the sender also compares the key stored in the item the location names
// with the request's (src/mega_send.c:543-553: nkey equal and the bytes equal -> hit, else
// miss): a 32-bit signature has false positives.  k_resolve_verify does that on the device
// against a flat item arena standing in for the slab store (item `loc` at loc * item_stride;
// the slab arithmetic of src/slabs.c:132-146 is the host allocator's business).
// ===========================================================================
namespace mkv {

__global__ __launch_bounds__(256) void k_resolve(const uint2 *__restrict__ pairs, int64_t n,
		uint32_t *__restrict__ loc, unsigned long long *__restrict__ counters)
{
	uint32_t hits = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
			i += (int64_t)gridDim.x * blockDim.x) {
		const uint2 p = pairs[i];
		const uint32_t l = p.x ? p.x : p.y;
		loc[i] = l;
		hits += l != 0u;
	}
	for (int off = 32; off > 0; off >>= 1)
		hits += __shfl_down(hits, off);
	if ((threadIdx.x & 63u) == 0u && hits)
		atomicAdd(&counters[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u], (unsigned long long)hits);
}

// KEY_MATCH.  Only the location the sender would pick is looked at -- the primary bucket's
// if it is non-zero, else the alternate's (:408-414); a false positive in the primary bucket
// therefore hides a true entry in the alternate one, as in the reference.  The compare is the
// reference's `it->nkey == key_len && strncmp(ITEM_key(it), key, key_len) == 0`
// (src/mega_send.c:544, also :712, 820, 847): strncmp stops at the first NUL, and the
// reference's own keys are binary with NUL bytes ({k, (k & 7) << 29 | k}, src/mega_recv.c:700-704),
// so two keys that agree up to and including their first NUL ARE equal to the reference.
// BYTES = true compares all key_len bytes instead (what a host with binary keys would want).
// Four lanes serve a request: 16 bytes of the item's key and of the request's key each per
// step (dword loads where the addresses allow), every lane notes the first differing byte and
// the first NUL of its piece, a quad reduction decides.
template <bool BYTES>
__global__ __launch_bounds__(256) void k_resolve_verify(const uint2 *__restrict__ pairs, int64_t n,
		const uint8_t *__restrict__ keys, int64_t key_stride, uint32_t key_len, const uint8_t *__restrict__ arena,
		int64_t item_stride, uint64_t arena_items, uint32_t nkey_off, uint32_t key_off,
		uint32_t *__restrict__ loc, unsigned long long *__restrict__ counters)
{
	const uint32_t q = threadIdx.x & 3u;
	uint32_t hits = 0, fps = 0;
	const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 2;
	for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; i < n; i += groups) {
		const uint2 p = pairs[i];
		const uint32_t l = p.x ? p.x : p.y;
		bool ok = l != 0u && (uint64_t)l < arena_items;
		const uint8_t *it = arena + (ok ? (uint64_t)l * (uint64_t)item_stride : 0ull);
		if (ok)
			ok = ((uint32_t)it[nkey_off] | ((uint32_t)it[nkey_off + 1u] << 8)) == key_len;   // uint16_t nkey
		const uint8_t *a = it + key_off, *k = keys + i * key_stride;
		const bool a4 = (((uintptr_t)a) & 3u) == 0u, k4 = (((uintptr_t)k) & 3u) == 0u;
		uint32_t d = 0xffffffffu, z = 0xffffffffu;     // first differing byte, first NUL of the item's key
		if (ok) {
			for (uint32_t base = q * 16u; base < key_len && d == 0xffffffffu; base += 64u) {
				uint32_t av[4], kv[4];
#pragma unroll
				for (uint32_t w = 0; w < 4u; w++) {
					const uint32_t o = base + 4u * w;
					av[w] = kv[w] = 0u;
					if (o + 4u <= key_len) {
						av[w] = a4 ? *reinterpret_cast<const uint32_t *>(a + o) :
								((uint32_t)a[o] | ((uint32_t)a[o + 1u] << 8) | ((uint32_t)a[o + 2u] << 16) | ((uint32_t)a[o + 3u] << 24));
						kv[w] = k4 ? *reinterpret_cast<const uint32_t *>(k + o) :
								((uint32_t)k[o] | ((uint32_t)k[o + 1u] << 8) | ((uint32_t)k[o + 2u] << 16) | ((uint32_t)k[o + 3u] << 24));
					} else {
						for (uint32_t c = 0; c < 4u && o + c < key_len; c++) {
							av[w] |= (uint32_t)a[o + c] << (8u * c);
							kv[w] |= (uint32_t)k[o + c] << (8u * c);
						}
					}
				}
#pragma unroll
				for (uint32_t w = 0; w < 4u; w++)
#pragma unroll
					for (uint32_t c = 0; c < 4u; c++) {
						const uint32_t o = base + 4u * w + c;
						const uint32_t ab = (av[w] >> (8u * c)) & 0xffu, kb = (kv[w] >> (8u * c)) & 0xffu;
						if (o < key_len && ab != kb && d == 0xffffffffu)
							d = o;
						if (o < key_len && ab == 0u && z == 0xffffffffu)
							z = o;
					}
			}
		}
		// the quad's first difference / first NUL
		for (int m = 1; m <= 2; m <<= 1) {
			const uint32_t od = __shfl_xor(d, m), oz = __shfl_xor(z, m);
			d = od < d ? od : d;
			z = oz < z ? oz : z;
		}
		// strncmp: equal unless a byte differs before (or at) the first NUL
		ok = ok && (d == 0xffffffffu || (!BYTES && z < d));
		if (q == 0u) {
			loc[i] = ok ? l : 0u;
			hits += ok ? 1u : 0u;
			fps += (l != 0u && !ok) ? 1u : 0u;
		}
	}
	for (int off = 32; off > 0; off >>= 1) {
		hits += __shfl_down(hits, off);
		fps += __shfl_down(fps, off);
	}
	if ((threadIdx.x & 63u) == 0u) {
		if (hits)
			atomicAdd(&counters[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u], (unsigned long long)hits);
		if (fps)
			atomicAdd(&counters[66], (unsigned long long)fps);
	}
}

// folds the 64 sharded partial sums: counters[64] = hits, counters[65] = misses
__global__ void k_resolve_fold(unsigned long long *counters, long long n)
{
	unsigned long long v = counters[threadIdx.x];
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_down(v, off);
	if (threadIdx.x == 0) {
		counters[64] = v;
		counters[65] = (unsigned long long)n - v;
	}
}

}  // namespace mkv

extern "C" int mkv_resolve_results(const loc_t *pairs, int64_t num_elem, loc_t *loc, uint64_t *counters,
		void *stream)
{
	if (num_elem < 0 || !counters || (num_elem && (!pairs || !loc)))
		return MKV_EINVAL;
	hipStream_t s = (hipStream_t)stream;
	if (hipMemsetAsync(counters, 0, 66 * sizeof(uint64_t), s) != hipSuccess)
		return MKV_EHIP;
	if (num_elem) {
		int64_t blocks = (num_elem + 255) / 256;
		if (blocks > 2048)
			blocks = 2048;
		hipLaunchKernelGGL(mkv::k_resolve, dim3((unsigned)blocks), dim3(256), 0, s,
				reinterpret_cast<const uint2 *>(pairs), num_elem, loc,
				reinterpret_cast<unsigned long long *>(counters));
	}
	hipLaunchKernelGGL(mkv::k_resolve_fold, dim3(1), dim3(64), 0, s,
			reinterpret_cast<unsigned long long *>(counters), (long long)num_elem);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" int mkv_resolve_verify(const loc_t *pairs, int64_t num_elem, const void *keys, int64_t key_stride,
		int key_len, const void *arena, int64_t item_stride, int64_t arena_items, int nkey_off, int key_off,
		int compare, loc_t *loc, uint64_t *counters, void *stream)
{
	if (num_elem < 0 || (compare != MKV_KEYCMP_STRNCMP && compare != MKV_KEYCMP_BYTES) || !counters || key_len < 0 || key_len > 65535 || item_stride <= 0 || arena_items < 0 ||
			nkey_off < 0 || key_off < 0 || (int64_t)nkey_off + 2 > item_stride || (int64_t)key_off + key_len > item_stride ||
			key_stride < key_len || (num_elem && (!pairs || !loc || !keys || !arena)))
		return MKV_EINVAL;
	hipStream_t s = (hipStream_t)stream;
	if (hipMemsetAsync(counters, 0, 68 * sizeof(uint64_t), s) != hipSuccess)
		return MKV_EHIP;
	if (num_elem) {
		int64_t blocks = (num_elem + 63) / 64;          // four lanes per request
		if (blocks > 8192)
			blocks = 8192;
		if (compare == MKV_KEYCMP_BYTES)
			hipLaunchKernelGGL((mkv::k_resolve_verify<true>), dim3((unsigned)blocks), dim3(256), 0, s,
					reinterpret_cast<const uint2 *>(pairs), num_elem, reinterpret_cast<const uint8_t *>(keys), key_stride,
					(uint32_t)key_len, reinterpret_cast<const uint8_t *>(arena), item_stride, (uint64_t)arena_items,
					(uint32_t)nkey_off, (uint32_t)key_off, loc, reinterpret_cast<unsigned long long *>(counters));
		else
			hipLaunchKernelGGL((mkv::k_resolve_verify<false>), dim3((unsigned)blocks), dim3(256), 0, s,
					reinterpret_cast<const uint2 *>(pairs), num_elem, reinterpret_cast<const uint8_t *>(keys), key_stride,
					(uint32_t)key_len, reinterpret_cast<const uint8_t *>(arena), item_stride, (uint64_t)arena_items,
					(uint32_t)nkey_off, (uint32_t)key_off, loc, reinterpret_cast<unsigned long long *>(counters));
	}
	hipLaunchKernelGGL(mkv::k_resolve_fold, dim3(1), dim3(64), 0, s,
			reinterpret_cast<unsigned long long *>(counters), (long long)num_elem);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

// ===========================================================================
// Request packets on the device: the reference's receiver walks
// every UDP payload on a CPU core (src/mega_recv.c:292-545): 16-bit magic 0x1234, then
// jobs  GET = u16 0x2, u16 key_len, key   SET = u16 0x3, u16 key_len, u32 value_len,
// key, value,  until the 16-bit end mark 0xFFFF (README.md:70-81).  Here a wave takes a
// packet: all lanes copy it into LDS, lane 0 walks the job headers (the only sequential
// part), all lanes derive (sig, hash) of the jobs side by side (:349-362, the XOR fold of
// the SIGNATURE build optional).  Two passes -- count, then emit at the scanned offsets --
// keep the reference's arrival order: packet order, job order inside a packet.
//   * a GET whose key length differs from key_len is skipped (:312-315), a SET whose
//     key or value length differs likewise (:449-453); both count as bad jobs
//   * a wrong magic drops the packet (:293-296); an unknown job type ends it, keeping the
//     jobs before it (:535-537); so does reaching the end of the payload (:540-543)
//   * SET requests leave here with loc = packet * 2048 + byte offset of the key in the
//     payload: the real location is the slab item the host allocates for it (:447,
//     src/slabs.c), which is outside the hot path.
// ===========================================================================
namespace mkv {

constexpr int kPktBytes = 2048;          // payload bytes a wave stages (frames are <= 1514)
constexpr int kPktJobs = 512;            // (2048 - 4) / 4: no payload holds more jobs

struct PktWalk {
	uint32_t njobs, ngets, nsets, nbad, bad_packet;
};

__device__ __forceinline__ uint32_t pkt_u16(const uint8_t *b, uint32_t off, uint32_t len)
{
	return (off < len ? b[off] : 0u) | ((off + 1u < len ? b[off + 1u] : 0u) << 8);
}

// lane 0 of a wave: job table (kind 1 = GET, 2 = SET, 0 = skipped; offset of the key)
__device__ PktWalk walk_packet(const uint8_t *b, uint32_t len, uint32_t key_len, uint32_t value_len,
		uint16_t *job_off, uint8_t *job_kind)
{
	PktWalk w = {0u, 0u, 0u, 0u, 0u};
	if (len < 4u || pkt_u16(b, 0u, len) != 0x1234u) {
		w.bad_packet = 1u;
		return w;
	}
	uint32_t ptr = 2u;
	while (w.njobs < (uint32_t)kPktJobs) {
		const uint32_t type = pkt_u16(b, ptr, len);
		ptr += 2u;
		if (type == 0xFFFFu)
			break;
		if (type == 0x2u) {
			const uint32_t nkey = pkt_u16(b, ptr, len);
			ptr += 2u;
			const bool ok = nkey == key_len;
			job_off[w.njobs] = (uint16_t)ptr;
			job_kind[w.njobs] = ok ? 1u : 0u;
			w.njobs++;
			w.ngets += ok ? 1u : 0u;
			w.nbad += ok ? 0u : 1u;
			ptr += nkey;
		} else if (type == 0x3u) {
			const uint32_t nkey = pkt_u16(b, ptr, len);
			const uint32_t nbyte = pkt_u16(b, ptr + 2u, len) | (pkt_u16(b, ptr + 4u, len) << 16);
			ptr += 6u;
			const bool ok = nkey == key_len && nbyte == value_len;
			job_off[w.njobs] = (uint16_t)ptr;
			job_kind[w.njobs] = ok ? 2u : 0u;
			w.njobs++;
			w.nsets += ok ? 1u : 0u;
			w.nbad += ok ? 0u : 1u;
			if (nbyte > (uint32_t)kPktBytes) {
				w.bad_packet = 1u;
				break;
			}
			ptr += nkey + nbyte;
		} else {
			w.bad_packet = 1u;               // wrong job type: what came before stays
			break;
		}
		if (ptr >= len) {                    // "Exceed payload length"
			w.bad_packet = 1u;
			break;
		}
	}
	return w;
}

// EMIT = false: counts per packet.  EMIT = true: requests at the scanned offsets.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_parse(const uint8_t *__restrict__ payloads, int64_t stride,
		const int32_t *__restrict__ payload_len, int64_t npackets, uint32_t key_len, uint32_t value_len, int fold,
		uint32_t *__restrict__ cnt_get, uint32_t *__restrict__ cnt_set, uint32_t *__restrict__ gets,
		int32_t *__restrict__ get_packet, uint32_t *__restrict__ sets, int64_t cap_get, int64_t cap_set,
		unsigned long long *counts)
{
	__shared__ uint32_t s_pay[4][kPktBytes / 4];
	__shared__ uint16_t s_off[4][kPktJobs];
	__shared__ uint8_t s_kind[4][kPktJobs];
	__shared__ PktWalk s_walk[4];
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const int64_t p = (int64_t)blockIdx.x * 4 + wave;
	if (p >= npackets)
		return;                                   // (whole waves leave: no block-wide barrier below)
	int64_t len64 = payload_len ? (int64_t)payload_len[p] : stride;
	if (len64 > stride)
		len64 = stride;
	if (len64 > kPktBytes)
		len64 = kPktBytes;
	const uint32_t len = len64 > 0 ? (uint32_t)len64 : 0u;
	const uint32_t *src = reinterpret_cast<const uint32_t *>(payloads + p * stride);
	for (uint32_t i = lane; i < (len + 3u) / 4u; i += 64u)
		s_pay[wave][i] = src[i];
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
	const uint8_t *b = reinterpret_cast<const uint8_t *>(s_pay[wave]);
	if (lane == 0)
		s_walk[wave] = walk_packet(b, len, key_len, value_len, s_off[wave], s_kind[wave]);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
	const PktWalk w = s_walk[wave];
	if (!EMIT) {
		if (lane == 0) {
			cnt_get[p] = w.ngets;
			cnt_set[p] = w.nsets;
			if (w.nbad)
				atomicAdd(&counts[3], (unsigned long long)w.nbad);
			if (w.bad_packet)
				atomicAdd(&counts[2], 1ull);
		}
		return;
	}
	uint32_t at_get = cnt_get[p], at_set = cnt_set[p];      // exclusive offsets after the scan
	for (uint32_t base = 0; base < w.njobs; base += 64u) {
		const uint32_t j = base + lane;
		const uint32_t kind = j < w.njobs ? s_kind[wave][j] : 0u;
		const unsigned long long mg = __ballot(kind == 1u), ms = __ballot(kind == 2u);
		const unsigned long long below = (1ull << lane) - 1ull;
		if (kind) {
			const uint32_t off = s_off[wave][j];
			// signature word: first 8 key bytes, the rest XOR-folded in the SIGNATURE build
			uint64_t sig = 0;
			for (uint32_t i = 0; i < 8u; i++)
				sig |= (uint64_t)(off + i < len ? b[off + i] : 0u) << (8u * i);
			if (fold) {
				for (uint32_t i = 8u; i < key_len; i++)
					sig ^= (uint64_t)(off + i < len ? b[off + i] : 0u) << (8u * (i & 7u));
			}
			if (kind == 1u) {
				const int64_t dst = (int64_t)at_get + __popcll(mg & below);
				if (dst < cap_get) {
					gets[dst * 2] = (uint32_t)sig;
					gets[dst * 2 + 1] = (uint32_t)(sig >> 32);
					if (get_packet)
						get_packet[dst] = (int32_t)p;
				}
			} else {
				const int64_t dst = (int64_t)at_set + __popcll(ms & below);
				if (dst < cap_set) {
					sets[dst * 3] = (uint32_t)sig;
					sets[dst * 3 + 1] = (uint32_t)(sig >> 32);
					sets[dst * 3 + 2] = (uint32_t)p * (uint32_t)kPktBytes + off;
				}
			}
		}
		at_get += (uint32_t)__popcll(mg);
		at_set += (uint32_t)__popcll(ms);
	}
}

// exclusive scan of the per-packet counts (one work-group), totals into counts[0..1]
__global__ __launch_bounds__(1024) void k_parse_scan(uint32_t *cnt_get, uint32_t *cnt_set, int64_t npackets,
		unsigned long long *counts)
{
	__shared__ unsigned long long s_part[2][1024];
	const uint32_t t = threadIdx.x;
	const int64_t per = (npackets + blockDim.x - 1) / blockDim.x;
	const int64_t lo = (int64_t)t * per, hi = lo + per < npackets ? lo + per : npackets;
	unsigned long long g = 0, s = 0;
	for (int64_t i = lo; i < hi; i++) {
		g += cnt_get[i];
		s += cnt_set[i];
	}
	s_part[0][t] = g;
	s_part[1][t] = s;
	__syncthreads();
	for (uint32_t off = 1; off < blockDim.x; off <<= 1) {
		const unsigned long long ag = t >= off ? s_part[0][t - off] : 0ull, as = t >= off ? s_part[1][t - off] : 0ull;
		__syncthreads();
		s_part[0][t] += ag;
		s_part[1][t] += as;
		__syncthreads();
	}
	unsigned long long rg = s_part[0][t] - g, rs = s_part[1][t] - s;
	for (int64_t i = lo; i < hi; i++) {
		const uint32_t cg = cnt_get[i], cs = cnt_set[i];
		cnt_get[i] = (uint32_t)rg;
		cnt_set[i] = (uint32_t)rs;
		rg += cg;
		rs += cs;
	}
	if (t == blockDim.x - 1u) {
		counts[0] = s_part[0][t];
		counts[1] = s_part[1][t];
	}
}

}  // namespace mkv

extern "C" int64_t mkv_parse_scratch_bytes(int64_t npackets)
{
	return (npackets > 0 ? npackets : 1) * 8 + 64;
}

// counts (device, 4 x u64): requests found n_get, n_set (compare with the capacities: what
// does not fit is dropped), bad packets, skipped jobs.
extern "C" int mkv_parse_packets(const void *payloads, int64_t stride, const int32_t *payload_len,
		int64_t npackets, int key_len, int value_len, int fold, selem_t *gets, int32_t *get_packet,
		ielem_t *sets, int64_t cap_get, int64_t cap_set, unsigned long long *counts, void *scratch,
		int64_t scratch_bytes, void *stream)
{
	if (npackets < 0 || stride < 4 || (stride & 3) || key_len < 8 || key_len > 1024 || value_len < 0 ||
			cap_get < 0 || cap_set < 0 || !counts || (npackets && (!payloads || !scratch)) ||
			(cap_get && !gets) || (cap_set && !sets) || scratch_bytes < mkv_parse_scratch_bytes(npackets))
		return MKV_EINVAL;
	hipStream_t s = (hipStream_t)stream;
	if (hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), s) != hipSuccess)
		return MKV_EHIP;
	if (npackets == 0)
		return MKV_OK;
	if ((npackets + 3) / 4 > 0x7fffffffLL)
		return MKV_ETOOBIG;
	uint32_t *cnt_get = reinterpret_cast<uint32_t *>(scratch), *cnt_set = cnt_get + npackets;
	const dim3 grid((unsigned)((npackets + 3) / 4)), block(256);
	const uint8_t *pl = reinterpret_cast<const uint8_t *>(payloads);
	hipLaunchKernelGGL((mkv::k_parse<false>), grid, block, 0, s, pl, stride, payload_len, npackets,
			(uint32_t)key_len, (uint32_t)value_len, fold, cnt_get, cnt_set, nullptr, nullptr, nullptr, cap_get,
			cap_set, counts);
	hipLaunchKernelGGL(mkv::k_parse_scan, dim3(1), dim3(1024), 0, s, cnt_get, cnt_set, npackets, counts);
	hipLaunchKernelGGL((mkv::k_parse<true>), grid, block, 0, s, pl, stride, payload_len, npackets,
			(uint32_t)key_len, (uint32_t)value_len, fold, cnt_get, cnt_set, reinterpret_cast<uint32_t *>(gets),
			get_packet, reinterpret_cast<uint32_t *>(sets), cap_get, cap_set, counts);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}
