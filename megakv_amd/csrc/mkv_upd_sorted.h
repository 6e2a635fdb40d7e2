// mkv_upd_sorted.h -- insert, BIG passes: partition, then apply per bucket (the structures and what the rounds and the
// tail ask of them).  Part of the insert / delete translation unit (mkv_update.hip includes it).  The kernels are in
// mkv_upd_apply.h; the design is described there and in DESIGN.md 5.2 item 15.
//
// A big pass (Tuning::sorted_min requests or more, announced) does not register on reservation words at all: its
// requests are split by the top bits of their primary bucket into at most kSortBins bins (extra work-groups of the
// search launch, k_search_part), and one work-group per bin applies them bucket by bucket in index order with ONE line
// read and ONE whole-line write per touched bucket (k_apply) -- as far as they are settled in their primary bucket.
// What it commits is a round-1 commit to everybody who asks (tag ST_ROUND1, kSortedBit in the request cache's flag
// word): it is logged -- 8 bytes per request, beside the bin -- and the tail finds it, as it finds the rounds' commits
// through the reservation words, through the BIN: every bin is left behind as a small chained hash table over its
// requests' buckets (heads / next), so "the commits on bucket X" is a walk of two or three loads.
#pragma once
#include "mkv_upd_common.h"

namespace mkv {

constexpr uint32_t kSortBinsLog2 = 9;           // at most 512 bins: a bin of a 1 GiB table covers 32 K buckets = 2 MiB
constexpr uint32_t kSortCapLog2 = 11;
constexpr uint32_t kSortBinCap = 1u << kSortCapLog2;    // requests a bin takes (2^19 requests: 1 024 on average, the fullest
                                                        // of 512 bins 1 170 or so); a pass that overflows one runs the old way
constexpr uint32_t kSortHeads = 4096;           // chain heads of a bin's hash table (by the low bits of the bucket index)
constexpr uint32_t kSortNone16 = 0xffffu;
constexpr uint32_t kSortedBit = 32u;            // request cache, flag word: the request went through a big pass's bins
constexpr uint32_t kSortPerLane = 8;            // partition: requests per lane and tile (a tile: 2 048 requests)

struct Sorted {
	uint4 *rec;             // [nbins * kSortBinCap] {sig, hash, loc, index}: the bins, in the order the partition's cursors gave
	uint32_t *cursor;       // [kSortBins] the partition's cursors: zero between passes (k_apply resets them)
	uint32_t *count;        // [kSortBins] requests of each bin in the current pass
	uint16_t *heads;        // [nbins * kSortHeads] per bin: first entry of the chain of (bucket & (kSortHeads - 1))
	uint16_t *next;         // [nbins * kSortBinCap] next entry of the chain (kSortNone16: none)
	uint2 *undo;            // [nbins * kSortBinCap] what the commit of an entry overwrote: {code, old location}; code
	                        // kNone: not applied (left over); else slot | 8 for a new entry (signature word was 0)
	uint32_t *losers;       // [request capacity] what k_apply leaves to round 2, by index (Ctl::sorted_nlosers of them)
	uint32_t bin_shift;     // bucket >> bin_shift = bin
	uint32_t nbins;
};

__device__ __forceinline__ Undo sorted_undo_of(uint32_t P, uint2 u)
{
	Undo r;
	const uint32_t s = u.x & 7u;
	if (u.x & 8u) {
		r.word0 = (uint32_t)sig_word(P, s);
		r.old0 = 0u;
		r.word1 = (uint32_t)loc_word(P, s);
		r.old1 = u.y;
	} else {
		r.word0 = (uint32_t)loc_word(P, s);
		r.old0 = u.y;
		r.word1 = kNoWord;
		r.old1 = 0u;
	}
	return r;
}

// every entry of the pass's bins whose primary bucket is b: f(entry {sig, hash, loc, index}, position in the bins)
// until it returns true
template <class F>
__device__ __forceinline__ void sorted_for_each_on(const Sorted &S, uint32_t hash_mask, uint32_t b, F f)
{
	const uint32_t bin = b >> S.bin_shift;
	uint32_t s = S.heads[(size_t)bin * kSortHeads + (b & (kSortHeads - 1u))];
	// (a chain holds what 8 buckets of the bin -- 2^15 / kSortHeads -- got in this pass: 0.25 entries on average;
	// the bound only guards against a damaged chain)
	for (uint32_t guard = 0; s != kSortNone16 && guard < kSortBinCap; guard++) {
		const size_t at = (size_t)bin * kSortBinCap + s;
		const uint4 r = S.rec[at];
		const uint32_t nx = S.next[at];
		if ((r.y & hash_mask) == b && f(r, (uint32_t)at))
			return;
		s = nx;
	}
}

// the position of request idx (primary bucket P) in the bins, kNone if it is not there
__device__ __forceinline__ uint32_t sorted_position(const Sorted &S, uint32_t hash_mask, uint32_t P, uint32_t idx)
{
	uint32_t at = kNone;
	sorted_for_each_on(S, hash_mask, P, [&](const uint4 &r, uint32_t pos) {
		if (r.w != idx)
			return false;
		at = pos;
		return true;
	});
	return at;
}

// a commit of k_apply: tagged like round 1's (no round 1 runs in such a pass), its request flagged as binned
// (what round 2 commits of the requests k_apply left over is tagged ST_ROUND2 / ST_CHAIN2 and logged the rounds' way)
__device__ __forceinline__ bool sorted_commit(uint32_t st, uint32_t aux)
{
	return (st & 0x7fu) == (uint32_t)ST_ROUND1 && (aux & kSortedBit) != 0u;
}

// the undo record of a logged commit (tag st): the rounds' from the per-request array, a big pass's from its bin
// (by value and a flag, not by pointer: a pointer to a kernel argument would put the argument into scratch memory)
__device__ __forceinline__ Undo undo_record(const Undo *undo, const Sorted &S, bool sorted_on, uint32_t hash_mask, uint32_t idx,
		const uint4 &rc, uint32_t st)
{
	if (sorted_on && sorted_commit(st, rc.w)) {
		const uint32_t P = rc.y & hash_mask;
		const uint32_t at = sorted_position(S, hash_mask, P, idx);
		if (at != kNone)
			return sorted_undo_of(P, S.undo[at]);
		Undo none;
		none.word0 = none.word1 = kNoWord;
		none.old0 = none.old1 = 0u;
		return none;
	}
	return undo[idx];
}

}  // namespace mkv
