// mkv_search.hip -- batched search of the bucketed hash index on gfx950.
//
// Replaces the reference kernel hash_search (libgpuhash/gpu_hash.cu:28-75):
// for request i, out[2i] receives the location stored under `sig` in the
// primary bucket and out[2i+1] the one in the alternate bucket; BOTH buckets
// are always probed (the early exit is commented out in the reference, :61-63)
// and a miss leaves the output word untouched (:50-53).
//
// MI355X mapping (this is HBM-bound random 64-byte gathering, no MFMA):
//   * 4 lanes serve one request: lanes 0/1 own the two 32-byte halves of the
//     primary bucket's slots (4 signatures + their 4 locations each), lanes
//     2/3 the alternate bucket's.  Every lane issues two independent 16-byte
//     loads (signature quarter-row and the matching location quarter-row), so
//     one wave instruction moves 1 KiB with all 16 bytes of every lane used and
//     a whole 64-byte bucket is fetched by one lane pair -- no dependent second
//     load for the location on a hit.
//   * U requests per lane group are in flight together (2*U 16-byte loads per
//     lane, 16*U requests = 2*16*U random 64-byte lines per wave) to cover HBM
//     latency with few resident registers.
//   * the compare is lane-local (4 slots), the pair is merged with one DPP
//     quad permute; no LDS, no ballot.  If several slots of a bucket hold the
//     signature the highest slot wins -- the serialised reference order; the
//     CUDA kernel races there (gpu_hash.cu:56-58).
#include <string.h>

#include <hip/hip_ext.h>

#include "mkv_common.h"
#include "mkv_search_body.h"

namespace mkv {

template <int U, bool ZERO_MISS, bool NT>
__global__ __launch_bounds__(256) void k_search(const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	search_body<U, ZERO_MISS, NT>(blockIdx.x, in, out, table, n, hash_mask, blk_mask);
}

// (the two other layouts of the search that were built and measured, DESIGN.md 5.1: diagnostics builds only)
#ifdef MKV_DIAGNOSTICS
template <int U, bool ZERO_MISS, bool NT>
__global__ __launch_bounds__(256) void k_search_wide(const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	search_body_wide<2 * U, ZERO_MISS, NT>(blockIdx.x, in, out, table, n, hash_mask, blk_mask);
}

template <bool ZERO_MISS>
__global__ __launch_bounds__(256) void k_search_lds(const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	search_body_lds<ZERO_MISS>(blockIdx.x, in, out, table, n, hash_mask, blk_mask);
}
#endif

template <int U>
static int launch_search(const uint2 *in, uint32_t *out, const uint4 *table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask, uint32_t flags, bool nt, hipStream_t stream, hipEvent_t e0 = nullptr,
		hipEvent_t e1 = nullptr)
{
	const int64_t per_block = 4 * 16 * U;
	const int64_t blocks = (n + per_block - 1) / per_block;
	if (blocks > 0x7fffffffLL)
		return MKV_ETOOBIG;
	dim3 grid((unsigned)blocks), block(256);
	const bool zero = (flags & MKV_SEARCH_ZERO_MISS) != 0;
#ifdef MKV_DIAGNOSTICS
	if (tuning().search_lds) {
		// 64 requests per block of 4 waves
		const int64_t lblocks = (n + 63) / 64;
		if (lblocks > 0x7fffffffLL)
			return MKV_ETOOBIG;
		if (zero)
			MKV_LAUNCH_TIMED((k_search_lds<true>), dim3((unsigned)lblocks), block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		else
			MKV_LAUNCH_TIMED((k_search_lds<false>), dim3((unsigned)lblocks), block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
	}
	if (tuning().search_wide) {
		// (same requests per block: 4 waves x 8 x 2U)
		if (zero)
			MKV_LAUNCH_TIMED((k_search_wide<U, true, false>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		else
			MKV_LAUNCH_TIMED((k_search_wide<U, false, false>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
	}
#endif
	if (zero) {
		if (nt)
			MKV_LAUNCH_TIMED((k_search<U, true, true>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		else
			MKV_LAUNCH_TIMED((k_search<U, true, false>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
	} else {
		if (nt)
			MKV_LAUNCH_TIMED((k_search<U, false, true>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
		else
			MKV_LAUNCH_TIMED((k_search<U, false, false>), grid, block, stream, e0, e1, in, out, table, n, hash_mask, blk_mask);
	}
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

Tuning &tuning()
{
	static Tuning t = [] {
		Tuning v;
		const char *s = getenv("MKV_SEARCH_U");
		v.search_u = s ? atoi(s) : 4;
		if (v.search_u != 1 && v.search_u != 2 && v.search_u != 4 && v.search_u != 8)
			v.search_u = 4;
		s = getenv("MKV_SEARCH_NT");
		v.search_nt = s ? atoi(s) != 0 : 0;
		v.search_wide = v.search_lds = 0;      // (diagnostics builds: mkv_set_tuning)
		s = getenv("MKV_TAIL_ROUNDS");
		v.tail_rounds = s ? atoi(s) : 48;
		if (v.tail_rounds < 0) v.tail_rounds = 0;
		if (v.tail_rounds > 48) v.tail_rounds = 48;
		s = getenv("MKV_REQ_CAP");
		v.req_cap = s ? atoll(s) : (1ll << 22);
		if (v.req_cap < 1024) v.req_cap = 1024;
		v.debug_skip = 0;
		v.debug_first_stamp = 0;
		s = getenv("MKV_ROUND_BLOCKS");
		v.round_blocks = s ? atoi(s) : 512;
		if (v.round_blocks < 1 || v.round_blocks > 65536) v.round_blocks = 512;
		v.grid_progress = 96;
		v.mark_alone_blocks = 1024;
		s = getenv("MKV_MARK_BLOCKS");
		v.mark_blocks = s ? atoi(s) : 208;
		if (v.mark_blocks < 1) v.mark_blocks = 1;
		if (v.mark_blocks > 1024) v.mark_blocks = 1024;
		v.pack_three_pass = 0;                 // (diagnostics builds: mkv_set_tuning forces the fall-back)
		s = getenv("MKV_PACK_THREADS");
		v.pack_threads = s ? atoi(s) : 1024;
		if (v.pack_threads != 256 && v.pack_threads != 512) v.pack_threads = 1024;
		s = getenv("MKV_BARRIER_STICKY");
		v.barrier_sticky = s ? atoi(s) : 16;
		if (v.barrier_sticky < 0) v.barrier_sticky = 0;
		s = getenv("MKV_BARRIER_SPLIT");
		v.barrier_split = s ? atoi(s) : 1;
		if (v.barrier_split < 1) v.barrier_split = 1;
		if (v.barrier_split > 64) v.barrier_split = 64;
		s = getenv("MKV_GRID_MIN");
		v.grid_min = s ? atoi(s) : 2048;
		if (v.grid_min < 0) v.grid_min = 0;
		s = getenv("MKV_CALM_FACTOR");
		v.calm_factor = s ? atoi(s) : 1;
		if (v.calm_factor < 1) v.calm_factor = 1;
		if (v.calm_factor > 64) v.calm_factor = 64;
		s = getenv("MKV_TAIL_GROUPS");
		v.tail_groups = s ? atoi(s) : 32;
		if (v.tail_groups < 1) v.tail_groups = 1;
		if (v.tail_groups > 64) v.tail_groups = 64;
		s = getenv("MKV_DUP_GROUPS");
		v.dup_groups = s ? atoi(s) != 0 : 1;
		s = getenv("MKV_LAZY_ALT");
		v.lazy_alt = s ? atoi(s) : 2;
		if (v.lazy_alt < 0 || v.lazy_alt > 8) v.lazy_alt = 2;
		s = getenv("MKV_SELF_HELP");
		v.self_help = s ? atoi(s) != 0 : 1;
		s = getenv("MKV_PLANS");
		v.plans = s ? atoi(s) != 0 : 1;
		s = getenv("MKV_MERGE_ROUND2");
		v.merge_round2 = s ? atoi(s) : 1;      // (2: even when round 1 has been leaving round 2 a lot)
		s = getenv("MKV_MERGE_MAX");
		v.merge_max = s ? atoll(s) : 131072;
		s = getenv("MKV_INSERT_CHUNK");
		v.insert_chunk = s ? atoll(s) : 524288;
		if (v.insert_chunk < 1024) v.insert_chunk = 1024;
		v.insert_chunk &= ~15ll;
		return v;
	}();
	return t;
}

}  // namespace mkv

extern "C" int mkv_search_timed(const mkv_geom_t *geom, const selem_t *in, loc_t *out, const bucket_t *table,
		int64_t num_elem, uint32_t flags, void *stream, void *ev_start, void *ev_stop);

extern "C" int mkv_search(const mkv_geom_t *geom, const selem_t *in, loc_t *out,
		const bucket_t *table, int64_t num_elem, uint32_t flags, void *stream)
{
	return mkv_search_timed(geom, in, out, table, num_elem, flags, stream, nullptr, nullptr);
}

// the same; ev_start / ev_stop (HIP events, or NULL) receive the dispatch's start and end time
extern "C" int mkv_search_timed(const mkv_geom_t *geom, const selem_t *in, loc_t *out, const bucket_t *table,
		int64_t num_elem, uint32_t flags, void *stream, void *ev_start, void *ev_stop)
{
	const mkv::Tuning &tune = mkv::tuning();
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || num_elem < 0)
		return MKV_EINVAL;
	if (num_elem == 0)
		return MKV_OK;
	if (!in || !out || !table)
		return MKV_EINVAL;
	const uint2 *in2 = reinterpret_cast<const uint2 *>(in);
	const uint4 *t4 = reinterpret_cast<const uint4 *>(table);
	hipStream_t s = (hipStream_t)stream;
	const bool nt = tune.search_nt != 0;
	hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
	switch (tune.search_u) {
	case 1: return mkv::launch_search<1>(in2, out, t4, num_elem, d.hash_mask, d.blk_mask, flags, nt, s, e0, e1);
	case 2: return mkv::launch_search<2>(in2, out, t4, num_elem, d.hash_mask, d.blk_mask, flags, nt, s, e0, e1);
	case 8: return mkv::launch_search<8>(in2, out, t4, num_elem, d.hash_mask, d.blk_mask, flags, nt, s, e0, e1);
	default: return mkv::launch_search<4>(in2, out, t4, num_elem, d.hash_mask, d.blk_mask, flags, nt, s, e0, e1);
	}
}

extern "C" int mkv_set_tuning(const char *name, long long value)
{
	if (!name)
		return MKV_EINVAL;
	mkv::Tuning &t = mkv::tuning();
	if (!strcmp(name, "search_u")) {
		if (value != 1 && value != 2 && value != 4 && value != 8)
			return MKV_EINVAL;
		t.search_u = (int)value;
	} else if (!strcmp(name, "search_nt")) {
		t.search_nt = value != 0;
#ifdef MKV_DIAGNOSTICS
	} else if (!strcmp(name, "search_wide")) {
		t.search_wide = value != 0;
	} else if (!strcmp(name, "search_lds")) {
		t.search_lds = value != 0;
#endif
	} else if (!strcmp(name, "tail_rounds")) {
		if (value < 0 || value > 48)
			return MKV_EINVAL;
		t.tail_rounds = (int)value;
	} else if (!strcmp(name, "insert_chunk")) {
		if (value < 1024)
			return MKV_EINVAL;
		t.insert_chunk = value & ~15ll;
	} else if (!strcmp(name, "mark_blocks")) {
		if (value < 1 || value > 1024)
			return MKV_EINVAL;
		t.mark_blocks = (int)value;
#ifdef MKV_DIAGNOSTICS
	} else if (!strcmp(name, "pack_three_pass")) {
		t.pack_three_pass = value != 0;
#endif
	} else if (!strcmp(name, "pack_threads")) {
		if (value != 256 && value != 512 && value != 1024)
			return MKV_EINVAL;
		t.pack_threads = (int)value;
	} else if (!strcmp(name, "barrier_sticky")) {
		if (value < 0 || value > 1000000)
			return MKV_EINVAL;
		t.barrier_sticky = (int)value;
	} else if (!strcmp(name, "barrier_split")) {
		if (value < 1 || value > 64)
			return MKV_EINVAL;
		t.barrier_split = (int)value;
	} else if (!strcmp(name, "grid_min")) {
		if (value < 0 || value > (1ll << 30))
			return MKV_EINVAL;
		t.grid_min = (int)value;
	} else if (!strcmp(name, "grid_progress")) {
		if (value < 0 || value > (1ll << 30))
			return MKV_EINVAL;
		t.grid_progress = (int)value;
	} else if (!strcmp(name, "calm_factor")) {
		if (value < 1 || value > 64)
			return MKV_EINVAL;
		t.calm_factor = (int)value;
	} else if (!strcmp(name, "tail_groups")) {
		if (value < 1 || value > 64)
			return MKV_EINVAL;
		t.tail_groups = (int)value;
	} else if (!strcmp(name, "round_blocks")) {
		if (value < 1 || value > 65536)
			return MKV_EINVAL;
		t.round_blocks = (int)value;
	} else if (!strcmp(name, "mark_alone_blocks")) {
		if (value < 1 || value > 65536)
			return MKV_EINVAL;
		t.mark_alone_blocks = (int)value;
#ifdef MKV_DIAGNOSTICS
	} else if (!strcmp(name, "debug_first_stamp")) {
		t.debug_first_stamp = value;
#endif
	} else if (!strcmp(name, "dup_groups")) {
		t.dup_groups = value != 0;
	} else if (!strcmp(name, "lazy_alt")) {
		if (value < 0 || value > 8)
			return MKV_EINVAL;
		t.lazy_alt = (int)value;
	} else if (!strcmp(name, "self_help")) {
		t.self_help = value != 0;
	} else if (!strcmp(name, "plans")) {
		t.plans = value != 0;
	} else if (!strcmp(name, "merge_round2")) {
		if (value < 0 || value > 2)
			return MKV_EINVAL;
		t.merge_round2 = (int)value;
	} else if (!strcmp(name, "merge_max")) {
		if (value < 0)
			return MKV_EINVAL;
		t.merge_max = value;
#ifdef MKV_DIAGNOSTICS
	} else if (!strcmp(name, "debug_skip")) {
		t.debug_skip = (int)value;
#endif
	} else if (!strcmp(name, "req_cap")) {
		if (value < 1024)
			return MKV_EINVAL;
		t.req_cap = value;
	} else {
		return MKV_EINVAL;
	}
	return MKV_OK;
}

// the fused search+mark kernel of mkv_cycle is built for the default search tuning
extern "C" int mkv_tuning_is_default_search(void)
{
	const mkv::Tuning &t = mkv::tuning();
	return t.search_u == 4 && t.search_nt == 0 && t.search_wide == 0 && t.search_lds == 0;
}

#ifdef MKV_DIAGNOSTICS
// ---- diagnostics: what the memory system gives for random reads ---------------------
// Every 4-lane group reads `gran`-byte aligned pieces (64, 128 or 256 bytes: 16 bytes per
// lane and step) at pseudo-random places of a buffer, `unroll` pieces in flight per group
// like the search kernel -- no hashing, no compare, no result but a checksum.  The rate
// this reaches is the ceiling for a kernel whose every request is two random buckets.
namespace mkv {

template <int GRAN, int U>
__global__ __launch_bounds__(256) void k_random_read(const uint4 *__restrict__ buf, uint64_t pieces_mask,
		int64_t pieces_per_group, uint32_t *__restrict__ sink, uint32_t seed)
{
	const uint32_t lane4 = threadIdx.x & 3u;
	const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
	uint64_t x = (group + 1u) * 0x9e3779b97f4a7c15ull + seed;
	uint32_t acc = 0;
	for (int64_t it = 0; it < pieces_per_group; it += U) {
		uint4 v[U][GRAN / 64];
#pragma unroll
		for (int u = 0; u < U; u++) {
			x ^= x >> 12;
			x ^= x << 25;
			x ^= x >> 27;
			const uint64_t piece = (x * 0x2545f4914f6cdd1dull >> 20) & pieces_mask;
			const uint4 *p = buf + piece * (GRAN / 16) + lane4;
#pragma unroll
			for (int k = 0; k < GRAN / 64; k++)
				v[u][k] = p[k * 4];
		}
#pragma unroll
		for (int u = 0; u < U; u++)
#pragma unroll
			for (int k = 0; k < GRAN / 64; k++)
				acc ^= v[u][k].x ^ v[u][k].y ^ v[u][k].z ^ v[u][k].w;
	}
	if (acc == 0x12345678u)
		sink[0] = acc;          // (keeps the loads alive)
}

}  // namespace mkv

namespace mkv {
// 32-byte pieces: two lanes per piece, two pieces per 4-lane group and step
template <int U>
__global__ __launch_bounds__(256) void k_random_read32(const uint4 *__restrict__ buf, uint64_t pieces_mask,
		int64_t pieces_per_group, uint32_t *__restrict__ sink, uint32_t seed)
{
	const uint32_t lane4 = threadIdx.x & 3u;
	const uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;     // two lanes share a piece
	uint64_t x = (pair + 1u) * 0x9e3779b97f4a7c15ull + seed;
	uint32_t acc = 0;
	for (int64_t it = 0; it < pieces_per_group; it += 2 * U) {
		uint4 v[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			x ^= x >> 12;
			x ^= x << 25;
			x ^= x >> 27;
			const uint64_t piece = (x * 0x2545f4914f6cdd1dull >> 20) & pieces_mask;
			v[u] = buf[piece * 2 + (lane4 & 1u)];
		}
#pragma unroll
		for (int u = 0; u < U; u++)
			acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}
}  // namespace mkv

// reads `groups` x pieces_per_group pieces of gran bytes from buf (buf_bytes: a power of two);
// returns the bytes read, or a negative error
extern "C" long long mkv_debug_random_read(const void *buf, long long buf_bytes, int gran, long long groups,
		long long pieces_per_group, uint32_t *sink, void *stream)
{
	if (!buf || !sink || buf_bytes < 4096 || (buf_bytes & (buf_bytes - 1)) || groups < 64 || pieces_per_group < 4 ||
			(pieces_per_group & 7) || (gran != 32 && gran != 64 && gran != 128 && gran != 256))
		return MKV_EINVAL;
	const uint64_t mask = (uint64_t)buf_bytes / (uint64_t)gran - 1u;
	const dim3 grid((unsigned)((groups * 4 + 255) / 256)), block(256);
	const uint4 *b = reinterpret_cast<const uint4 *>(buf);
	hipStream_t s = (hipStream_t)stream;
	if (gran == 32)
		hipLaunchKernelGGL((mkv::k_random_read32<4>), grid, block, 0, s, b, mask, pieces_per_group, sink, 1u);
	else if (gran == 64)
		hipLaunchKernelGGL((mkv::k_random_read<64, 4>), grid, block, 0, s, b, mask, pieces_per_group, sink, 1u);
	else if (gran == 128)
		hipLaunchKernelGGL((mkv::k_random_read<128, 4>), grid, block, 0, s, b, mask, pieces_per_group, sink, 1u);
	else
		hipLaunchKernelGGL((mkv::k_random_read<256, 4>), grid, block, 0, s, b, mask, pieces_per_group, sink, 1u);
	if (hipGetLastError() != hipSuccess)
		return MKV_EHIP;
	return (long long)grid.x * 64 * pieces_per_group * gran;
}
#endif  // MKV_DIAGNOSTICS
