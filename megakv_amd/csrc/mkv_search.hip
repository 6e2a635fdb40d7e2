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
	if (MKV_DIAG(search_lds)) {
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
	if (MKV_DIAG(search_wide)) {
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
		auto env = [](const char *name, long long dflt, long long lo, long long hi) {
			const char *s = getenv(name);
			const long long x = s ? atoll(s) : dflt;
			return x < lo || x > hi ? dflt : x;
		};
		v.search_u = (int)env("MKV_SEARCH_U", 4, 1, 8);
		if (v.search_u != 1 && v.search_u != 2 && v.search_u != 4 && v.search_u != 8)
			v.search_u = 4;
		v.search_nt = (int)env("MKV_SEARCH_NT", 0, 0, 1);
		v.tail_rounds = (int)env("MKV_TAIL_ROUNDS", 48, 0, 48);
		v.req_cap = env("MKV_REQ_CAP", 1ll << 22, 1024, 1ll << 40);
		v.insert_chunk = env("MKV_INSERT_CHUNK", 524288, 1024, 1ll << 40) & ~15ll;
		v.mark_blocks = (int)env("MKV_MARK_BLOCKS", 208, 1, 1024);
		v.pack_threads = (int)env("MKV_PACK_THREADS", 1024, 256, 1024);
		if (v.pack_threads != 256 && v.pack_threads != 512)
			v.pack_threads = 1024;
		v.barrier_sticky = (int)env("MKV_BARRIER_STICKY", 16, 0, 1000000);
		v.round_blocks = (int)env("MKV_ROUND_BLOCKS", 512, 1, 65536);
		v.tail_groups = (int)env("MKV_TAIL_GROUPS", 32, 1, 64);
		v.dup_groups = (int)env("MKV_DUP_GROUPS", 1, 0, 1);
		v.lazy_alt = (int)env("MKV_LAZY_ALT", 2, 0, 8);
		v.self_help = (int)env("MKV_SELF_HELP", 1, 0, 1);
		v.plans = (int)env("MKV_PLANS", 1, 0, 1);
		v.merge_round2 = (int)env("MKV_MERGE_ROUND2", 1, 0, 2);      // (2: even when round 1 has been leaving round 2 a lot)
		v.merge_max = env("MKV_MERGE_MAX", 131072, 0, 1ll << 40);
		v.sorted_min = env("MKV_SORTED_MIN", 0, 0, 1ll << 40);       // (off by default: DESIGN.md 5.2 item 15 -- exact, not faster)
		v.sorted_cooldown = (int)env("MKV_SORTED_COOLDOWN", 32, 0, 1000000);
		return v;
	}();
	return t;
}

#ifdef MKV_DIAGNOSTICS
DiagTuning &diag_tuning()
{
	static DiagTuning d;
	return d;
}
#endif

}  // namespace mkv

extern "C" int mkv_search_timed(const mkv_geom_t *geom, const selem_t *in, loc_t *out, const bucket_t *table,
		int64_t num_elem, uint32_t flags, void *stream, void *ev_start, void *ev_stop);
extern "C" int mkv_defer_barrier(void);

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
	const int brc = mkv_defer_barrier();        // (the drop-in side's queued request copies, deferred mode: mkv_abi.hip)
	if (brc != MKV_OK)
		return brc;
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
	struct Knob { const char *name; int *i; long long *ll; long long lo, hi; };
	const Knob knobs[] = {
		{"search_nt", &t.search_nt, nullptr, 0, 1}, {"tail_rounds", &t.tail_rounds, nullptr, 0, 48},
		{"req_cap", nullptr, &t.req_cap, 1024, 1ll << 40}, {"mark_blocks", &t.mark_blocks, nullptr, 1, 1024},
		{"barrier_sticky", &t.barrier_sticky, nullptr, 0, 1000000}, {"round_blocks", &t.round_blocks, nullptr, 1, 65536},
		{"tail_groups", &t.tail_groups, nullptr, 1, 64}, {"dup_groups", &t.dup_groups, nullptr, 0, 1},
		{"lazy_alt", &t.lazy_alt, nullptr, 0, 8}, {"self_help", &t.self_help, nullptr, 0, 1}, {"plans", &t.plans, nullptr, 0, 1},
		{"merge_round2", &t.merge_round2, nullptr, 0, 2}, {"merge_max", nullptr, &t.merge_max, 0, 1ll << 40},
		{"sorted_min", nullptr, &t.sorted_min, 0, 1ll << 40}, {"sorted_cooldown", &t.sorted_cooldown, nullptr, 0, 1000000},
#ifdef MKV_DIAGNOSTICS
		{"search_wide", &mkv::diag_tuning().search_wide, nullptr, 0, 1}, {"search_lds", &mkv::diag_tuning().search_lds, nullptr, 0, 1},
		{"pack_three_pass", &mkv::diag_tuning().pack_three_pass, nullptr, 0, 1},
		{"grid_min", &mkv::diag_tuning().grid_min, nullptr, 0, 1ll << 30},
		{"grid_progress", &mkv::diag_tuning().grid_progress, nullptr, 0, 1ll << 30},
		{"calm_factor", &mkv::diag_tuning().calm_factor, nullptr, 1, 64},
		{"mark_alone_blocks", &mkv::diag_tuning().mark_alone_blocks, nullptr, 1, 65536},
		{"debug_first_stamp", nullptr, &mkv::diag_tuning().debug_first_stamp, 0, 1ll << 40},
		{"debug_skip", &mkv::diag_tuning().debug_skip, nullptr, 0, 7},
#endif
	};
	if (!strcmp(name, "search_u")) {
		if (value != 1 && value != 2 && value != 4 && value != 8)
			return MKV_EINVAL;
		t.search_u = (int)value;
		return MKV_OK;
	}
	if (!strcmp(name, "insert_chunk")) {
		if (value < 1024)
			return MKV_EINVAL;
		t.insert_chunk = value & ~15ll;
		return MKV_OK;
	}
	if (!strcmp(name, "pack_threads")) {
		if (value != 256 && value != 512 && value != 1024)
			return MKV_EINVAL;
		t.pack_threads = (int)value;
		return MKV_OK;
	}
	for (const Knob &k : knobs) {
		if (strcmp(name, k.name))
			continue;
		if (value < k.lo || value > k.hi)
			return MKV_EINVAL;
		if (k.i)
			*k.i = (int)value;
		else
			*k.ll = value;
		return MKV_OK;
	}
	return MKV_EINVAL;      // (unknown here: a diagnostics-only knob in the product library, say)
}

// the fused search+mark kernel of mkv_cycle is built for the default search tuning
extern "C" int mkv_tuning_is_default_search(void)
{
	const mkv::Tuning &t = mkv::tuning();
	return t.search_u == 4 && t.search_nt == 0 && MKV_DIAG(search_wide) == 0 && MKV_DIAG(search_lds) == 0;
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
