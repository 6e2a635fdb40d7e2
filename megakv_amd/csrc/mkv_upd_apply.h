// mkv_upd_apply.h -- insert, BIG passes: the partition (k_part, and as extra work-groups of the search launch:
// k_search_part) and the apply kernel (k_apply).  Part of the insert / delete translation unit (mkv_update.hip).
//
// WHY.  A pass of 2^19 SETs (BASELINE configs[3]) run through the reservation rounds is bound by what the memory side
// does per request -- a line read and a returning atomic in the mark pass, two 4-byte stores into two sectors of a
// bucket in round 1: 8 x write amplification, 32 G memory-side operations a second in all (profiles/r05_launch_floor.txt
// 10 - 13).  Here a request costs the memory side a share of a streamed record, one line read and one whole-line
// write:
//   1. PARTITION (table-independent; rides in the search launch like the mark pass it replaces): a multi-split of the
//      pass's requests by the top bits of their primary bucket into at most 512 bins of 16-byte records {sig, hash,
//      loc, index} -- per tile of 2 048 requests a histogram in LDS, one returning atomic per (tile, bin) on the bin's
//      cursor, the records stored where the cursors say.  It also fills the request cache and tags every request
//      ST_ROUND1 -- in index order, streamed.  A bin that overflows (kSortBinCap: a pass of duplicates of one hot key)
//      makes k_apply leave EVERY request of the pass to round 2 (Ctl::sorted_fail): exact, slow, and the host is told
//      (word 5 of its flags) so that it runs the next passes the old way.
//   2. APPLY (a launch behind the search): a work-group per bin.  The bin goes into LDS and is linked into a chained
//      hash table over its requests' buckets; the request with the LOWEST index of a bucket loads the bucket (a quad of
//      lanes a line, load_bucket_quad) and applies the bucket's requests in index order in registers -- a duplicate's
//      location, else the preferred empty slot (ref libgpuhash/gpu_hash.cu:277-323) -- as long as each is settled
//      there, and writes the bucket back WHOLE.  The first request of a bucket that is not -- the bucket is full, it
//      carries signature 0, or it would rewrite a word an earlier request of this very pass wrote -- and every LATER
//      one of that bucket is LEFT OVER: tagged ST_LOSER, registered in round 2's reservation table on both its
//      buckets, listed.  An applied request logs {slot, old location} beside its record; the bin's hash table goes to
//      global memory beside it: the per-bucket log of what was committed (mkv_upd_sorted.h).
//   3. ROUND 2 over the list (k_round<2>, list mode), then the TAIL, exact as ever.  What they have to know: a left-over
//      request never registered on its alternate bucket (kLazyBit: its late use of it is checked), nobody registered in
//      the round-1 table (eviction chains run in the tail only, which checks every bucket they step on), and the
//      commits on a bucket are the reservation words' candidates PLUS what the bin holds for it.
// WHY IT IS EXACT.  Per bucket the applied requests ran in index order, before every pending request of that bucket
// (the left-over ones are a suffix of the bucket's requests by index), and they wrote pairwise different words (so
// any subset can be taken back in any order).  That is what the rounds guarantee of their commits; whatever a
// left-over request or a chain then meets on ANOTHER bucket -- commits of later requests there -- is an order
// violation, found through the bin and resolved by the tail's take-back and in-order replay like any other.
#pragma once
#include "mkv_upd_common.h"
#include "mkv_upd_sorted.h"
#include "mkv_upd_mark.h"
#include "mkv_upd_tail.h"

namespace mkv {

constexpr uint32_t kPartThreads = 256;
constexpr uint32_t kPartTile = kPartThreads * kSortPerLane;
static_assert(kSortBinsLog2 <= 9 && (1u << kSortBinsLog2) <= 2u * kPartThreads, "two bins a lane");

// ---- partition --------------------------------------------------------------------------------------------------
template <Op OP>
__device__ __forceinline__ void partition_body(uint32_t block, uint32_t nblocks, Ctl *ctl, const ReqSrc &src, const DevGeom &g,
		const Sorted &S, uint4 *reqc, uint8_t *state, uint32_t req_cap, uint32_t hi_idx)
{
	__shared__ uint32_t s_prefix[kMaxBlocks + 1];
	__shared__ const uint32_t *s_ptr[kMaxBlocks];
	__shared__ uint32_t s_total;
	__shared__ uint32_t s_hist[1u << kSortBinsLog2], s_base[1u << kSortBinsLog2];
	const uint32_t t = threadIdx.x;
	const uint32_t nblk = src.nblk;
	// this pass's round-1 stamp, as a mark pass works it out (Ctl, head)
	const unsigned long long stamp_prev = ctl->stamp_base;
	const uint32_t ep_pending = ctl->ep_pending;
	const unsigned long long stamp = stamp_prev + (ep_pending ? 3ull : 0ull);
	for (uint32_t b = t; b < nblk; b += blockDim.x) {
		const int c = src.blk_elem_num[(size_t)b * src.cnt_stride];
		s_ptr[b] = src.blk_input[b];
		s_prefix[b + 1] = c > 0 ? (uint32_t)c : 0u;
	}
	__syncthreads();
	if (t == 0) {
		unsigned long long sum = 0;
		s_prefix[0] = 0;
		for (uint32_t b = 0; b < nblk; b++) {
			sum += s_prefix[b + 1];
			s_prefix[b + 1] = (uint32_t)(sum > 0xfffffff0ull ? 0xfffffff0ull : sum);
		}
		s_total = sum > req_cap ? kNone : (uint32_t)sum;
	}
	__syncthreads();
	const bool over = s_total == kNone;
	const uint32_t total = over ? 0u : s_total;
	// what the first work-group of a mark pass leaves in the control block (mark_body): the same here
	if (block == 0 && t == 0) {
		ctl->r1_wide = ctl->r1_wide_next;
		ctl->stamp_cur = stamp;
		ctl->total = total;
		ctl->overflow = over ? 1u : 0u;
		ctl->force_tail = (over || ctl->hf_last[0] != 0u || ctl->hf_last[1] != 0u || ctl->hf_losers != 0u || ctl->hf_walk != 0u ||
				stamp > kStampMax - kStampReserve) ? 1u : 0u;
		ctl->sorted_nlosers = 0u;
		if (ep_pending) {
			ctl->first_hard = kNone;
			ctl->chain_n = 0u;
			ctl->chain_ok = 0u;
			ctl->any_loser = 0u;
		}
	}
	if (block == 0 && t < (uint32_t)kShards) {
		ctl->committed[0][t].n = 0;
		ctl->committed[1][t].n = 0;
		ctl->committed[2][t].n = 0;
		ctl->committed[3][t].n = 0;
	}
	if (block == 0)
		for (uint32_t q = t; q < kTlGroups; q += blockDim.x) {
			ctl->tl_cnt[0][q] = 0u;
			ctl->wg_rec[q].stat = make_uint4(0u, 0u, 0u, 0u);
		}
	const uint32_t end = hi_idx < total ? hi_idx : total;
	const uint32_t nbins = S.nbins;
	bool fail = false;
	for (uint32_t tile = block; (unsigned long long)tile * kPartTile < end; tile += nblocks) {
		const uint32_t first = tile * kPartTile;
		for (uint32_t b = t; b < nbins; b += blockDim.x)
			s_hist[b] = 0u;
		__syncthreads();
		// pass 1: every request's bin and its rank among the tile's requests of that bin (two ranks a register)
		uint32_t rank[kSortPerLane / 2];
		uint32_t skip = 0u;
#pragma unroll
		for (uint32_t k = 0; k < kSortPerLane; k++) {
			const uint32_t idx = first + k * kPartThreads + t;
			uint32_t r = 0u;
			if (idx < end) {
				uint32_t lo = 0, hi = nblk;             // block b: prefix[b] <= idx < prefix[b+1]
				while (hi - lo > 1) {
					const uint32_t mid = (lo + hi) >> 1;
					if (s_prefix[mid] <= idx)
						lo = mid;
					else
						hi = mid;
				}
				const uint32_t *p = s_ptr[lo] + 3ull * (idx - s_prefix[lo]);
				const uint32_t sig = p[0], hash = p[1], loc = p[2];
				if (OP != OP_DELETE && sig == 0u && loc == 0u)
					skip |= 1u << k;                    // skipped by the reference too (:259-262)
				else
					r = atomicAdd(&s_hist[(hash & g.hash_mask) >> S.bin_shift], 1u);
			}
			if (k & 1u)
				rank[k >> 1] |= r << 16;
			else
				rank[k >> 1] = r;
		}
		__syncthreads();
		for (uint32_t b = t; b < nbins; b += blockDim.x) {
			const uint32_t c = s_hist[b];
			s_base[b] = c ? atomicAdd(&S.cursor[b * kSortCursorStride], c) : 0u;
		}
		__syncthreads();
		// pass 2: the records to their bins, the request cache and the tags in index order
#pragma unroll
		for (uint32_t k = 0; k < kSortPerLane; k++) {
			const uint32_t idx = first + k * kPartThreads + t;
			if (idx >= end)
				continue;
			uint32_t lo = 0, hi = nblk;
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_prefix[mid] <= idx)
					lo = mid;
				else
					hi = mid;
			}
			const uint32_t *p = s_ptr[lo] + 3ull * (idx - s_prefix[lo]);
			const uint32_t sig = p[0], hash = p[1], loc = p[2];
			if ((skip >> k) & 1u) {
				reqc[idx] = make_uint4(sig, hash, loc, 0u);
				state[idx] = ST_DONE;
				continue;
			}
			const uint32_t bin = (hash & g.hash_mask) >> S.bin_shift;
			const uint32_t pos = s_base[bin] + ((rank[k >> 1] >> ((k & 1u) * 16u)) & 0xffffu);
			if (pos < kSortBinCap)
				S.rec[(size_t)bin * kSortBinCap + pos] = make_uint4(sig, hash, loc, idx);
			else
				fail = true;
			// (no request of such a pass registers on its alternate bucket: kLazyBit -- whoever uses it late checks)
			reqc[idx] = make_uint4(sig, hash, loc, kLazyBit | kSortedBit);
			state[idx] = ST_ROUND1;         // (k_apply retags what it leaves over)
		}
		__syncthreads();
	}
	if (fail || (over && block == 0 && t == 0))
		ctl->sorted_fail = (uint32_t)stamp;     // (an oversize batch: total 0, nothing to do -- the tail's epilogue counts the drop)
}

template <Op OP>
__global__ __launch_bounds__(kPartThreads) void k_part(Ctl *ctl, ReqSrc src, DevGeom g, Sorted S, uint4 *reqc, uint8_t *state,
		uint32_t req_cap, uint32_t hi_idx)
{
	partition_body<OP>(blockIdx.x, gridDim.x, ctl, src, g, S, reqc, state, req_cap, hi_idx);
}

// the search of a cycle and the partition of its insert in ONE launch (k_search_mark's counterpart for big passes)
template <Op OP>
__global__ __launch_bounds__(256) void k_search_part(const uint2 *__restrict__ in, uint32_t *__restrict__ out,
		const uint4 *__restrict__ table, int64_t n, uint32_t search_blocks, Ctl *ctl, ReqSrc src, DevGeom g, Sorted S,
		uint4 *reqc, uint8_t *state, uint32_t req_cap, uint32_t hi_idx)
{
	const uint32_t part_blocks = gridDim.x - search_blocks;
	if (blockIdx.x < part_blocks)
		partition_body<OP>(blockIdx.x, part_blocks, ctl, src, g, S, reqc, state, req_cap, hi_idx);
	else
		search_body<4, true, false>(blockIdx.x - part_blocks, in, out, table, n, g.hash_mask, g.blk_mask);
}

// ---- apply ------------------------------------------------------------------------------------------------------
// a bucket of four lanes' pieces back to memory: the inverse of load_bucket_quad (b = kNone: nothing to store)
__device__ __forceinline__ void store_bucket_quad(uint32_t *table, uint32_t b, const Bkt &k)
{
	const uint32_t q = threadIdx.x & 3u;
	uint4 a0 = k.s0, a1 = k.s1, a2 = k.l0, a3 = k.l1;
	quad_transpose(a0, a1, a2, a3);
	const uint32_t b0 = quad_bcast<0>(b), b1 = quad_bcast<1>(b), b2 = quad_bcast<2>(b), b3 = quad_bcast<3>(b);
	uint4 *t4 = reinterpret_cast<uint4 *>(table) + q;
	if (b0 != kNone) t4[(uint64_t)b0 << 2] = a0;
	if (b1 != kNone) t4[(uint64_t)b1 << 2] = a1;
	if (b2 != kNone) t4[(uint64_t)b2 << 2] = a2;
	if (b3 != kNone) t4[(uint64_t)b3 << 2] = a3;
}

constexpr uint32_t kApplyThreads = 512;

template <Op OP>
__global__ __launch_bounds__(kApplyThreads) void k_apply(Ctl *ctl, DevGeom g, uint32_t *table, Sorted S, uint8_t *state,
		unsigned long long *owner_next, const uint4 *reqc)
{
	__shared__ uint4 s_rec[kSortBinCap];                // 32 KB
	__shared__ uint2 s_log[kSortBinCap];                // 16 KB
	__shared__ uint32_t s_head[kSortHeads];             // 16 KB
	__shared__ uint16_t s_next[kSortBinCap];            //  4 KB
	__shared__ uint32_t s_nl, s_lbase, s_napplied;
	const uint32_t t = threadIdx.x, bin = blockIdx.x;
	const unsigned long long stamp1 = ctl->stamp_cur;
	const bool failed = ctl->sorted_fail == (uint32_t)stamp1;
	uint32_t n = S.cursor[bin * kSortCursorStride];
	__syncthreads();
	if (t == 0)
		S.cursor[bin * kSortCursorStride] = 0u;         // (for the next pass's partition)
	if (bin == 0 && t == 0) {
		if (failed)
			ctl->sorted_fallbacks = ctl->sorted_fallbacks + 1u;
		else
			ctl->sorted_passes = ctl->sorted_passes + 1u;
	}
	n = n < kSortBinCap ? n : kSortBinCap;
	// what the first work-group of round 1 does for the kernels behind it (round_body)
	if (bin == 0) {
		reset_tail_scratch(ctl);
		if (t == 0) {
			ctl->stamp_base = stamp1;
			ctl->ep_pending = 1u;
			ctl->tl_groups[0] = 0u;
		}
		for (uint32_t q = t; q < kTlGroups; q += blockDim.x)
			ctl->tl_cnt[1][q] = 0u;
	}
	const unsigned long long stamp2 = stamp1 + 1ull;
	if (failed) {
		// The partition gave up (a bin overflowed: a pass of duplicates of a few keys): NOTHING is applied here.  Every
		// request of the pass is left to round 2 -- tagged, registered there on both its buckets (folded per wave: the
		// duplicates sit next to each other), the list the identity -- and the bins are left empty.  Exact like any
		// round 2; slow (the host, told through its flag word 5, runs the next passes the old way).
		for (uint32_t q = t; q < kSortHeads; q += blockDim.x)
			S.heads[(size_t)bin * kSortHeads + q] = (uint16_t)kSortNone16;
		if (t == 0)
			S.count[bin] = 0u;
		const uint32_t total = ctl->total;
		const uint4 *reqc4 = reqc;
		for (uint32_t base = bin * blockDim.x; base < total; base += gridDim.x * blockDim.x) {
			const uint32_t idx = base + t;
			const bool live = idx < total && state[idx] == ST_ROUND1;
			const uint4 r = live ? reqc4[idx] : make_uint4(0u, 0u, 0u, 0u);
			const uint32_t P = r.y & g.hash_mask, A = alt_bucket(r.y, r.x, g.hash_mask, g.blk_mask);
			if (idx < total)
				S.losers[idx] = idx;                    // (a request that is not pending is skipped by its tag)
			if (live)
				state[idx] = ST_LOSER;
			reserve_wave(owner_next, P, A, owner_key(stamp2, idx), live);
		}
		if (bin == 0 && t == 0) {
			ctl->sorted_nlosers = total;
			ctl->any_loser = 1u;
		}
		return;
	}
	if (t == 0) {
		S.count[bin] = n;
		s_nl = 0u;
		s_napplied = 0u;
	}
	for (uint32_t q = t; q < kSortHeads; q += blockDim.x)
		s_head[q] = kSortNone16;
	__syncthreads();
	const size_t bin0 = (size_t)bin * kSortBinCap;
	for (uint32_t s = t; s < n; s += blockDim.x) {
		const uint4 r = S.rec[bin0 + s];
		s_rec[s] = r;
		s_log[s] = make_uint2(kNone, 0u);
		s_next[s] = (uint16_t)atomicExch(&s_head[(r.y & g.hash_mask) & (kSortHeads - 1u)], s);
	}
	__syncthreads();
	uint32_t napplied = 0;
	for (uint32_t s0 = 0; s0 < n; s0 += blockDim.x) {      // (whole waves: the bucket loads and stores are a quad's)
		const uint32_t s = s0 + t;
		const bool live = s < n;
		const uint4 me = live ? s_rec[s] : make_uint4(0u, 0u, 0u, 0u);
		const uint32_t P = me.y & g.hash_mask;
		const uint32_t h = P & (kSortHeads - 1u);
		// the bucket's requests are the chain's entries with this bucket: the one with the lowest index applies them all
		bool leader = live;
		uint32_t others = 0;
		if (live)
			for (uint32_t c = s_head[h]; c != kSortNone16; c = s_next[c]) {
				const uint4 rc = s_rec[c];
				if ((rc.y & g.hash_mask) == P && c != s) {
					others++;
					leader = leader && rc.w > me.w;
				}
			}
		Bkt k = load_bucket_quad(table, leader ? P : kNone);
		bool dirty = false;
		if (leader) {
			uint32_t written = 0u;
			uint2 lg;
			bool open = apply_in_registers(k, me.x, me.z, written, &lg);
			if (open) {
				s_log[s] = lg;
				dirty = true;
				napplied++;
			}
			// the others in index order: the lowest index above the last one applied, until one is not settled
			uint32_t last = me.w;
			for (uint32_t done = 0; open && done < others; done++) {
				uint32_t best = kNone, bidx = kNone;
				for (uint32_t c = s_head[h]; c != kSortNone16; c = s_next[c]) {
					const uint4 rc = s_rec[c];
					if ((rc.y & g.hash_mask) == P && rc.w > last && rc.w < bidx) {
						bidx = rc.w;
						best = c;
					}
				}
				if (best == kNone)
					break;
				const uint4 rb = s_rec[best];
				open = apply_in_registers(k, rb.x, rb.z, written, &lg);
				if (open) {
					s_log[best] = lg;
					napplied++;
				}
				last = bidx;
			}
		}
		store_bucket_quad(table, dirty ? P : kNone, k);
	}
	__syncthreads();
	// what was not applied: to round 2 -- tagged, registered on both its buckets in that round's table, listed
	uint32_t mine = 0;
	for (uint32_t s = t; s < n; s += blockDim.x)
		mine += s_log[s].x == kNone ? 1u : 0u;
	uint32_t my_at = 0;
	if (mine)
		my_at = atomicAdd(&s_nl, mine);
	for (int off = 32; off > 0; off >>= 1)
		napplied += __shfl_down(napplied, off);
	if ((t & 63u) == 0u && napplied)
		atomicAdd(&s_napplied, napplied);
	__syncthreads();
	if (t == 0 && s_nl) {
		s_lbase = atomicAdd(&ctl->sorted_nlosers, s_nl);
		ctl->any_loser = 1u;
	}
	__syncthreads();
	const uint32_t lbase = s_lbase;
	for (uint32_t s = t; s < n; s += blockDim.x) {
		const uint2 lg = s_log[s];
		S.undo[bin0 + s] = lg;
		S.next[bin0 + s] = s_next[s];
		if (lg.x == kNone) {
			const uint4 r = s_rec[s];
			const uint32_t P = r.y & g.hash_mask, A = alt_bucket(r.y, r.x, g.hash_mask, g.blk_mask);
			state[r.w] = ST_LOSER;
			reserve(owner_next, P, A, owner_key(stamp2, r.w));
			S.losers[lbase + my_at++] = r.w;
		}
	}
	for (uint32_t q = t; q < kSortHeads; q += blockDim.x)
		S.heads[(size_t)bin * kSortHeads + q] = (uint16_t)s_head[q];
	if (t == 0 && s_napplied)
		atomicAdd(&ctl->committed[0][bin & (kShards - 1)].n, s_napplied);
}

}  // namespace mkv
