// mkv_upd_sorted.h -- insert, BIG passes: partition, then apply per bucket (the structures and what the rounds and the
// tail ask of them).  Part of the insert / delete translation unit (mkv_update.hip includes it).  The kernels are in
// mkv_upd_apply.h; the design is described there and in DESIGN.md 5.2 item 15.
//
// A big pass (Tuning::sorted_min requests or more, announced; 0 = never, the default) does not register on reservation words at all: its
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
constexpr uint32_t kSortCursorStride = 16;
constexpr uint32_t kSortedBit = 32u;            // request cache, flag word: the request went through a big pass's bins
constexpr uint32_t kSortPerLane = 8;            // partition: requests per lane and tile (a tile: 2 048 requests)

struct Sorted {
	uint4 *rec;             // [nbins * kSortBinCap] {sig, hash, loc, index}: the bins, in the order the partition's cursors gave
	uint32_t *cursor;       // [kSortBins * kSortCursorStride] the partition's cursors, a 64-byte line each (every tile of the
	                        // pass adds to every cursor: same-line atomics go one at a time); zero between passes (k_apply)
	uint32_t *count;        // [kSortBins] requests of each bin in the current pass
	uint16_t *heads;        // [nbins * kSortHeads] per bin: first entry of the chain of (bucket & (kSortHeads - 1))
	uint16_t *next;         // [nbins * kSortBinCap] next entry of the chain (kSortNone16: none)
	uint2 *undo;            // [nbins * kSortBinCap] what the commit of an entry overwrote: {code, old location}; code
	                        // kNone: not applied (left over); else slot | 8 for a new entry (signature word was 0)
	uint32_t *losers;       // [request capacity] what k_apply leaves to round 2, by index (Ctl::sorted_nlosers of them)
	uint32_t *host_flag;    // the workspace's host-mapped flag words (word 5: this path's verdict on itself, round_body)
	uint32_t bin_shift;     // bucket >> bin_shift = bin
	uint32_t nbins;
};

// what a kernel that may serve a big pass takes for its bins: the usual instantiation takes NOTHING (an empty
// argument, no registers), the one for big passes the structure above
struct NoSorted {};
template <bool SORTED> struct SortedArg { typedef NoSorted type; };
template <> struct SortedArg<true> { typedef Sorted type; };
__device__ __forceinline__ Sorted sorted_of_arg(const Sorted &s) { return s; }
__device__ __forceinline__ Sorted sorted_of_arg(const NoSorted &) { return Sorted{}; }

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

__device__ __forceinline__ void put4(uint4 &v, uint32_t s, uint32_t x)
{
	v.x = s == 0u ? x : v.x;
	v.y = s == 1u ? x : v.y;
	v.z = s == 2u ? x : v.z;
	v.w = s == 3u ? x : v.w;
}

// what place_loaded does, on the bucket in registers only: true = settled there.  `written`: the slots requests of this
// pass have written so far -- a request that would rewrite one of them is not settled here (its commit and the earlier
// one would share a word: they could not be taken back independently).
__device__ __forceinline__ bool apply_in_registers(Bkt &k, uint32_t sig, uint32_t loc, uint32_t &written, uint2 *log)
{
	if (sig == 0u)
		return false;                       // (signature 0 "matches" every empty slot, ref :277-287: left to the tail)
	const uint32_t m = eq_mask(k.s0, k.s1, sig);
	uint32_t s, code;
	if (m) {
		s = (uint32_t)__ffs((int)m) - 1u;
		if ((written >> s) & 1u)
			return false;
		code = s;
	} else {
		const uint32_t e = eq_mask(k.s0, k.s1, 0u);
		if (!e)
			return false;
		s = preferred_slot(e, sig);
		code = s | 8u;
		if (s & 4u)
			put4(k.s1, s & 3u, sig);
		else
			put4(k.s0, s & 3u, sig);
	}
	log->x = code;
	log->y = loc_at(k, s);
	if (s & 4u)
		put4(k.l1, s & 3u, loc);
	else
		put4(k.l0, s & 3u, loc);
	written |= 1u << s;
	return true;
}

// Round 2 of a big pass, a left-over request (index idx, not settled in its primary bucket) that is the lowest
// registrant of its alternate bucket A among the pending requests: k_apply may have committed requests of HIGHER index
// there (their primary bucket) -- an order violation if the request simply went ahead.  Most of the time it can be put
// right on the spot, in registers: those commits are taken out of the bucket (their logs say how), the request takes
// its place -- a duplicate's location, else the preferred empty slot -- and they are applied again behind it in index
// order with new logs, as long as each is still settled there; from the first one that is not, they become pending
// requests of the tail (a suffix by index, like what k_apply leaves over).  Nobody else touches A meanwhile: every
// pending request that has it as primary or alternate bucket stands behind this one in the round's table, a chain
// never steps on a bucket somebody stands on, and the commits taken out are finished requests on their own primary
// bucket.  0: no later commit there (the caller goes ahead); 1: done, the request is committed in A (its undo record
// in `mine`; d0..d3: requests that are pending again, kNone: none); 2: not here -- more than four later commits,
// or the request needs an eviction chain (or carries signature 0): nothing written, the tail's.
__device__ __forceinline__ int resolve_late_alternate(uint32_t *table, const DevGeom &g, const Sorted &S, uint8_t *state,
		uint32_t sig, uint32_t loc, uint32_t idx, uint32_t A, Undo &mine, uint32_t &d0, uint32_t &d1, uint32_t &d2, uint32_t &d3)
{
	uint32_t jp0 = kNone, jp1 = kNone, jp2 = kNone, jp3 = kNone, ji0 = kNone, ji1 = kNone, ji2 = kNone, ji3 = kNone, nj = 0;
	{
		// (the walk of sorted_for_each_on, written out: a closure over nine variables ends up in scratch memory)
		const uint32_t bin = A >> S.bin_shift;
		uint32_t s = S.heads[(size_t)bin * kSortHeads + (A & (kSortHeads - 1u))];
		for (uint32_t guard = 0; s != kSortNone16 && guard < kSortBinCap; guard++) {
			const uint32_t pos = bin * kSortBinCap + s;
			const uint4 rs = S.rec[pos];
			const uint32_t nx = S.next[pos];
			if ((rs.y & g.hash_mask) == A && rs.w > idx && state[rs.w] == ST_ROUND1) {
				jp3 = nj == 3u ? pos : jp3;
				ji3 = nj == 3u ? rs.w : ji3;
				jp2 = nj == 2u ? pos : jp2;
				ji2 = nj == 2u ? rs.w : ji2;
				jp1 = nj == 1u ? pos : jp1;
				ji1 = nj == 1u ? rs.w : ji1;
				jp0 = nj == 0u ? pos : jp0;
				ji0 = nj == 0u ? rs.w : ji0;
				nj++;
			}
			s = nx;
		}
	}
	d0 = d1 = d2 = d3 = kNone;
	if (nj == 0u)
		return 0;
	if (nj > 4u)
		return 2;
	// by index, ascending (kNone sorts last): a five-exchange network
#define MKV_CSWAP(a, b, pa, pb) do { if (a > b) { const uint32_t t_ = a; a = b; b = t_; const uint32_t p_ = pa; pa = pb; pb = p_; } } while (0)
	MKV_CSWAP(ji0, ji1, jp0, jp1);
	MKV_CSWAP(ji2, ji3, jp2, jp3);
	MKV_CSWAP(ji0, ji2, jp0, jp2);
	MKV_CSWAP(ji1, ji3, jp1, jp3);
	MKV_CSWAP(ji1, ji2, jp1, jp2);
#undef MKV_CSWAP
	Bkt k = load_bucket(table, A);
	// out of the bucket (they wrote pairwise different words: any order)
#define MKV_TAKE_OUT(jp) do { if (jp != kNone) { const uint2 u_ = S.undo[jp]; const uint32_t s_ = u_.x & 7u;                      \
		if (u_.x & 8u) { if (s_ & 4u) put4(k.s1, s_ & 3u, 0u); else put4(k.s0, s_ & 3u, 0u); }                                       \
		if (s_ & 4u) put4(k.l1, s_ & 3u, u_.y); else put4(k.l0, s_ & 3u, u_.y); } } while (0)
	MKV_TAKE_OUT(jp0);
	MKV_TAKE_OUT(jp1);
	MKV_TAKE_OUT(jp2);
	MKV_TAKE_OUT(jp3);
#undef MKV_TAKE_OUT
	uint32_t written = 0u;
	uint2 lg;
	if (!apply_in_registers(k, sig, loc, written, &lg))
		return 2;
	mine = sorted_undo_of(A, lg);
	bool open = true;
	uint2 l0 = make_uint2(kNone, 0u), l1 = l0, l2 = l0, l3 = l0;
#define MKV_AGAIN(jp, lj) do { if (jp != kNone && open) { const uint4 rs_ = S.rec[jp]; open = apply_in_registers(k, rs_.x, rs_.z, written, &lj); \
		if (!open) lj = make_uint2(kNone, 0u); } } while (0)
	MKV_AGAIN(jp0, l0);
	MKV_AGAIN(jp1, l1);
	MKV_AGAIN(jp2, l2);
	MKV_AGAIN(jp3, l3);
#undef MKV_AGAIN
	uint4 *p = reinterpret_cast<uint4 *>(table + (uint64_t)A * kBucketWords);
	p[0] = k.s0;
	p[1] = k.s1;
	p[2] = k.l0;
	p[3] = k.l1;
#define MKV_RELOG(jp, ji, lj, d) do { if (jp != kNone) { S.undo[jp] = lj; if (lj.x == kNone) { state[ji] = ST_TAIL; d = ji; } } } while (0)
	MKV_RELOG(jp0, ji0, l0, d0);
	MKV_RELOG(jp1, ji1, l1, d1);
	MKV_RELOG(jp2, ji2, l2, d2);
	MKV_RELOG(jp3, ji3, l3, d3);
#undef MKV_RELOG
	return 1;
}

}  // namespace mkv
