// mkv_search_body.h -- the search work-group body, shared by the stand-alone
// search kernel (mkv_search.hip) and the fused search+mark kernel of mkv_cycle
// (mkv_update.hip).  See mkv_search.hip for the design notes.
#pragma once

#include "mkv_common.h"

namespace mkv {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// quad_perm [1,0,3,2]: swap lanes inside each pair.
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
// quad_perm [2,3,0,1]: swap the pairs of a quad.
__device__ __forceinline__ uint32_t dpp_swap2(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
}

template <int U, bool ZERO_MISS, bool NT>
__device__ __forceinline__ void search_body(uint32_t block, const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t q = lane & 3u;       // 0,1 primary halves; 2,3 alternate halves
	const uint32_t grp = lane >> 2;     // request slot inside the wave, 0..15
	const int64_t wave = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t base = wave * (16 * U);

	u32x4 sv[U], lv[U];
	uint32_t sig[U];
	bool live[U];

#pragma unroll
	for (int u = 0; u < U; u++) {
		const int64_t e = base + u * 16 + grp;
		live[u] = e < n;
		uint2 req = live[u] ? in[e] : make_uint2(0u, 0u);    // {sig, hash}
		sig[u] = req.x;
		const uint32_t pri = req.y & hash_mask;
		const uint32_t alt = alt_bucket(req.y, req.x, hash_mask, blk_mask);
		const uint32_t b = (q & 2u) ? alt : pri;
		const u32x4 *p = reinterpret_cast<const u32x4 *>(table) + ((uint64_t)b << 2) + (q & 1u);
		if (live[u]) {
			if (NT) {
				sv[u] = __builtin_nontemporal_load(p);
				lv[u] = __builtin_nontemporal_load(p + 2);
			} else {
				sv[u] = p[0];
				lv[u] = p[2];
			}
		} else {
			sv[u] = (u32x4)(0u);
			lv[u] = sv[u];
		}
	}

#pragma unroll
	for (int u = 0; u < U; u++) {
		const int64_t e = base + u * 16 + grp;
		// highest matching slot of this lane's four
		uint32_t hit = 0u, val = 0u;
		if (sv[u].x == sig[u]) { hit = 1u; val = lv[u].x; }
		if (sv[u].y == sig[u]) { hit = 1u; val = lv[u].y; }
		if (sv[u].z == sig[u]) { hit = 1u; val = lv[u].z; }
		if (sv[u].w == sig[u]) { hit = 1u; val = lv[u].w; }
		// merge the pair: the odd lane holds slots 4..7 and takes precedence
		const uint32_t phit = dpp_swap1(hit), pval = dpp_swap1(val);
		if (q & 1u) {
			// odd lanes do not store; keep their registers consistent anyway
		} else {
			if (phit) { hit = 1u; val = pval; }
		}
		if (ZERO_MISS) {
			// lane q==0 gathers the alternate result and stores both words
			const uint32_t ahit = dpp_swap2(hit), aval = dpp_swap2(val);
			if (q == 0u && live[u]) {
				uint2 r = make_uint2(hit ? val : 0u, ahit ? aval : 0u);
				*reinterpret_cast<uint2 *>(out + 2 * e) = r;
			}
		} else {
			if ((q & 1u) == 0u && hit && live[u])
				out[2 * e + (q >> 1)] = val;
		}
	}
}

// row_shl:4 -- lane i takes the value of lane i+4 (inside its row of 16)
__device__ __forceinline__ uint32_t dpp_from_plus4(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);
}

#ifdef MKV_DIAGNOSTICS
// The same search with 8 lanes per request: a quad owns one bucket, each lane one 16-byte
// quarter of it (sig[0..3], sig[4..7], loc[0..3], loc[4..7]), so a bucket is ONE 64-byte
// access of one load instruction instead of two 32-byte halves of two instructions -- half
// the cache-line look-ups per bucket in the L1.  Lanes 0/1 of the quad compare, the
// locations come over from lanes 2/3 with a quad permute.  8 * U requests per wave.
template <int U, bool ZERO_MISS, bool NT>
__device__ __forceinline__ void search_body_wide(uint32_t block, const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t q = lane & 3u;            // quarter of the bucket
	const uint32_t side = (lane >> 2) & 1u;  // 0: primary bucket, 1: alternate
	const uint32_t grp = lane >> 3;          // request slot inside the wave, 0..7
	const int64_t wave = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const int64_t base = wave * (8 * U);

	u32x4 v[U];
	uint32_t sig[U];
	bool live[U];

#pragma unroll
	for (int u = 0; u < U; u++) {
		const int64_t e = base + u * 8 + grp;
		live[u] = e < n;
		uint2 req = live[u] ? in[e] : make_uint2(0u, 0u);    // {sig, hash}
		sig[u] = req.x;
		const uint32_t pri = req.y & hash_mask;
		const uint32_t alt = alt_bucket(req.y, req.x, hash_mask, blk_mask);
		const uint32_t b = side ? alt : pri;
		const u32x4 *p = reinterpret_cast<const u32x4 *>(table) + ((uint64_t)b << 2) + q;
		if (live[u])
			v[u] = NT ? __builtin_nontemporal_load(p) : p[0];
		else
			v[u] = (u32x4)(0u);
	}

#pragma unroll
	for (int u = 0; u < U; u++) {
		const int64_t e = base + u * 8 + grp;
		// the locations of this lane's four signatures: same quarter index, two lanes up
		const uint32_t lx = dpp_swap2(v[u].x), ly = dpp_swap2(v[u].y), lz = dpp_swap2(v[u].z),
				lw = dpp_swap2(v[u].w);
		// highest matching slot of this lane's four (lanes 2/3 compare locations: never read)
		uint32_t hit = 0u, val = 0u;
		if (v[u].x == sig[u]) { hit = 1u; val = lx; }
		if (v[u].y == sig[u]) { hit = 1u; val = ly; }
		if (v[u].z == sig[u]) { hit = 1u; val = lz; }
		if (v[u].w == sig[u]) { hit = 1u; val = lw; }
		// merge the pair: lane 1 holds slots 4..7 and takes precedence
		const uint32_t phit = dpp_swap1(hit), pval = dpp_swap1(val);
		if (q == 0u && phit) { hit = 1u; val = pval; }
		if (ZERO_MISS) {
			// lane 0 of the group gathers the alternate quad's result and stores both words
			const uint32_t ahit = dpp_from_plus4(hit), aval = dpp_from_plus4(val);
			if ((lane & 7u) == 0u && live[u]) {
				uint2 r = make_uint2(hit ? val : 0u, ahit ? aval : 0u);
				*reinterpret_cast<uint2 *>(out + 2 * e) = r;
			}
		} else {
			if (q == 0u && hit && live[u])
				out[2 * e + side] = val;
		}
	}
}

// The search in the shape the task's north star sketches, kept for the record (tuning
// "search_lds"; DESIGN.md 5.1 has the measurement): the signature rows of a wave's 16 requests
// are loaded coalesced (4 lanes x 16 bytes = both 32-byte rows of a request) and staged in
// LDS; then a lane per signature compares, a wave ballot gives every request its 16-bit match
// mask, ffs/clz picks the slot (the highest one, as everywhere here), and the lane that owns
// the request fetches the location with a second, dependent load -- the reference's
// structure (libgpuhash/gpu_hash.cu:41-72) on 64 lanes.  Bit-identical results; one request
// per 4 lanes and step, two round trips per hit.
template <bool ZERO_MISS>
__device__ __forceinline__ void search_body_lds(uint32_t block, const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n,
		uint32_t hash_mask, uint32_t blk_mask)
{
	__shared__ uint32_t srow[4][16][16];    // [wave][request][8 primary + 8 alternate signatures]
	__shared__ uint2 sreq[4][16];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const int64_t base = ((int64_t)block * 4 + wave) * 16;
	{
		const uint32_t r = lane >> 2, part = lane & 3u;
		const int64_t e = base + r;
		const bool live = e < n;
		const uint2 req = live ? in[e] : make_uint2(0u, 0u);
		const uint32_t pri = req.y & hash_mask;
		const uint32_t alt = alt_bucket(req.y, req.x, hash_mask, blk_mask);
		const uint32_t b = (part & 2u) ? alt : pri;
		const uint4 v = live ? table[((uint64_t)b << 2) + (part & 1u)] : make_uint4(0u, 0u, 0u, 0u);
		*reinterpret_cast<uint4 *>(&srow[wave][r][part * 4u]) = v;
		if (part == 0u)
			sreq[wave][r] = req;
	}
	__syncthreads();
#pragma unroll
	for (uint32_t q = 0; q < 4; q++) {
		const uint32_t rr = 4u * q + (lane >> 4), s = lane & 15u;
		const int64_t e = base + rr;
		const bool live = e < n;
		const uint2 req = sreq[wave][rr];
		const bool match = live && srow[wave][rr][s] == req.x;
		const unsigned long long m = __ballot(match);
		const uint32_t mine = (uint32_t)(m >> ((lane >> 4) * 16u)) & 0xffffu;
		if (s == 0u && live) {
			const uint32_t mp = mine & 0xffu, ma = mine >> 8;
			const uint32_t pri = req.y & hash_mask;
			const uint32_t alt = alt_bucket(req.y, req.x, hash_mask, blk_mask);
			const uint32_t *tw = reinterpret_cast<const uint32_t *>(table);
			uint32_t lp = 0u, la = 0u;
			if (mp)
				lp = tw[(uint64_t)pri * 16u + 8u + (31u - (uint32_t)__clz((int)mp))];
			if (ma)
				la = tw[(uint64_t)alt * 16u + 8u + (31u - (uint32_t)__clz((int)ma))];
			if (ZERO_MISS) {
				*reinterpret_cast<uint2 *>(out + 2 * e) = make_uint2(mp ? lp : 0u, ma ? la : 0u);
			} else {
				if (mp)
					out[2 * e] = lp;
				if (ma)
					out[2 * e + 1] = la;
			}
		}
	}
}
#endif  // MKV_DIAGNOSTICS

// Search of the slots a shard received (padded exchange format, mkv_route.hip):
// linear block index -> (slot, 256-request chunk); the slot's request count is read
// from its header on the device.
constexpr int kSlotHdrWords = 4;

// (self_out: the results of slot `self` go THERE -- the rank's own buffer of returned results:
// what a rank answers to itself need not pass through the exchange; self out of range: none)
__device__ __forceinline__ void search_slots_body(uint32_t block, const uint32_t *__restrict__ recvbuf,
		int64_t slot_words, uint32_t cap_get, uint32_t *__restrict__ out, const uint4 *__restrict__ table,
		uint32_t hash_mask, uint32_t blk_mask, uint32_t chunks_per_slot, uint32_t self, uint32_t *__restrict__ self_out)
{
	const uint32_t seg = block / chunks_per_slot;
	const uint32_t chunk = block - seg * chunks_per_slot;
	const uint32_t *slot = recvbuf + (int64_t)seg * slot_words;
	const uint32_t n = slot[0];
	if ((uint64_t)chunk * 256u >= n)
		return;
	search_body<4, true, false>(chunk, reinterpret_cast<const uint2 *>(slot + kSlotHdrWords),
			seg == self ? self_out : out + (int64_t)seg * cap_get * 2, table, (int64_t)n, hash_mask, blk_mask);
}

}  // namespace mkv
