// mkv_upd_common.h -- insert / delete: constants, control block, request cache, coherent access, reservation tables, folds
// Part of the insert / delete translation unit (mkv_update.hip includes it; the kernels share the control
// block, the reservation helpers and the request cache, and the tail launch runs round 2's body, so they
// are compiled together).  The design is described at the top of mkv_update.hip and in DESIGN.md 5.2.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "mkv_common.h"

namespace mkv {

constexpr int kMaxBlocks = 1024;    // insert blocks per call
constexpr int kMaxRegions = 32;
constexpr int kMaxTailRounds = 48;  // in-work-group rounds of the tail
constexpr uint32_t kMinRoundProgress = 24;   // in-work-group round: ~4 us against ~75 ns per replayed request / 8 regions
constexpr uint32_t kGridHandover = 1024;  // grid-wide rounds hand a list this short over to work-group 0's in-group rounds
constexpr int kGridRoundsMax = 4096;   // grid-wide rounds of a final tail with a long list
constexpr int kStampsPerCall = 2 + kMaxTailRounds + 2;   // first pass after initialisation; later passes take what they use
constexpr uint32_t kTlGroups = 512;  // the grid-wide rounds list what they leave to the tail: per work-group of a round
constexpr uint32_t kTlPer = 8;       // ... this many entries (a work-group with more, or more work-groups: the tail scans the tags)
constexpr uint32_t kLlPer = 4;       // (static_assert below: the record is 32 bytes)       // round 1 lists what it leaves to ROUND 2 too: per work-group this many
constexpr uint32_t kFewLosers = 8;   // ... and the tail launch's first work-group alone runs round 2 over a list this short
constexpr int kTailReplayMax = 24;  // the tail replays this many unfinished requests in order without trying rounds
constexpr int kTailAloneSpan = 131072; // ... or beyond this many request tags to scan
constexpr int kTailAloneMax = 1024; // the tail keeps its helper work-groups only beyond this many unfinished requests
constexpr int kSortCap = 8192;      // tail requests held in LDS; beyond: state scan
constexpr int kViolCap = 128;       // order violations recorded per turn of the tail (more are met again later)
constexpr uint8_t ST_SELECT = 0x80; // tail, transient: flag on a commit tag -- selected for roll-back
constexpr int kTailThreads = 1024;  // lanes per work-group of the tail launch (its helpers are force-inlined: the
                                    // launch stays free of scratch memory and calls)
constexpr int kDeferCap = 768;           // k_round: requests a work-group may hold back for its second phase
constexpr uint32_t kMergeLosersMax = 16;   // round 2 runs inside the tail launch while round 1 leaves it at most this many requests
                                          // (with work to do it costs what its own launch costs, and a little more: 20.7 against 19.0 us
                                          // for round 2 + tail at load factor 0.5)

enum Op : uint32_t { OP_CUCKOO = 0, OP_2CHOICE = 1, OP_DELETE = 2 };

// request state
// PENDING: registered, not yet tried; ROUND1/ROUND2: committed optimistically by
// that round; LOSER: lost a round-1 reservation (round 2 retries it); TAIL: left
// for the tail work-group.  Rounds find their work by these tags -- no atomic
// work lists, no same-address atomics on the hot path.
constexpr uint8_t ST_PENDING = 0, ST_DONE = 1, ST_ROUND1 = 2, ST_ROUND2 = 3, ST_LOSER = 4, ST_TAIL = 5,
		ST_CHAIN1 = 6, ST_CHAIN2 = 7,   // committed with an eviction chain (log in the chain log)
		ST_CHAIN_OK = 8,                // tail rounds: chain walked and all its buckets claimed
		ST_DUPW = 9,                    // tail rounds: holds a duplicate nothing can evict -- a pure loc write
		ST_DUPW_OK = 10,                // ... and takes part in this round's last-writer election of its bucket
		ST_DEFER = 11,                  // tail: a commit taken back because an earlier chain needs its bucket sits
		                                // out one round unregistered (as that round's barrier), so that the chain
		                                // finds the bucket free
		ST_SHADOW = 12;                 // round 2: a duplicate of a key whose group's writes one earlier duplicate has
		                                // made for all of them (undo[idx].word0 = that request; see kFollowBit) --
		                                // finished as long as that commit stands (shadow_live)
constexpr int kShards = 64;         // statistics counters are spread over this many words
// A request whose primary bucket shows at least this many free slots at mark time does not register
// on its alternate bucket (bit 0 of the request cache's fourth word says so): half the atomics of
// the mark pass, which runs at the memory side's atomic rate.  Should the primary bucket fill up
// before the request's turn after all, its late use of the alternate bucket is checked for later
// commits there like a chain's step (round 2: inline; tail: later_commit_on).
// Duplicate keys inside one pass (Zipf SET keys: a hot key comes hundreds of times).  Same key -- to
// the table: same signature and same primary bucket, whatever the hash bits above the bucket index --,
// same buckets: the first of them -- if it is the lowest registrant of their primary bucket -- runs in
// round 1, and every later one that finds its own key under the bucket's reservation there is a
// FOLLOWER (kFollowBit): all it will ever do is overwrite that entry's location, so only the last
// follower's location matters.  Round 1: the followers elect the last one (atomicMax per bucket).
// Round 2: the lowest follower -- the bucket's lowest registrant of that round, if no other key's
// request comes before it -- runs with the LAST follower's location and is the group's one logged
// commit (kLeadBit, the last follower's index in the upper 24 bits: to whoever asks for later
// commits on the bucket it counts as that index); the other followers are its shadows, done as
// long as that commit stands, pending requests like any other the moment it is taken back.
// (Before: one duplicate per round, the rest -- 37 % of the SETs of a 95/5 Zipf batch -- through the
// tail's grid-wide rounds with a last-writer election there: 72 - 92 us of a 120 us step.)
constexpr uint32_t kFollowBit = 2u, kLeadBit = 4u;
constexpr uint32_t kCheckedBit = 8u;   // tail: the request has looked at the grid-wide rounds' commits on its alternate bucket
constexpr uint32_t kLazyBit = 1u;      // (the number of free slots asked for: Tuning::lazy_alt, 2 by default: 1 / 2 / 3 measured alike)
// The mark pass loads a new request's primary bucket anyway (for the decision above).  It also works out what the
// request would do there as the table stands -- overwrite its duplicate's location, or take the preferred empty
// slot -- and leaves that PLAN where the commit's undo record goes (it IS that record: the words to write and what
// they hold).  Round 1 carries the plan out without reading the bucket again when the request is the bucket's lowest
// registrant -- nobody else may write there before it, and a second registrant that helps itself writes other words
// -- and the table has not changed since the mark pass (no delete of the same cycle in between: the host says so).
// And it knows that without reading the reservation either: every registration of the mark pass is a returning
// atomicMin, and whoever sees two keys of the pass on one word -- its own and the one it displaced or lost to --
// FLAGS the larger one's request (`lost`: a word per request holding the low half of the pass's stamp; never
// cleared).  A request is displaced from a word by exactly one atomic, which then sees its key; a request that never
// got onto the word saw a smaller key itself: so a request nobody flagged is the lowest registrant of every bucket it
// registered on.  (Folded registrations flag on the way: the lanes of a wave that share a bucket, the entries of a
// work-group's table.)  Chains that walk ahead (prewalk_register) register without a returning atomic; a pass in
// which one did says so (Ctl::hint_stamp) and round 1 reads every reservation as before.
// Round 1 of a 52 K-request pass: three dependent round trips (request; reservation + bucket; stores) -> two.
constexpr uint32_t kPlanBit = 16u;

// Reservation entries are 64 bits: a 40-bit round stamp over a 24-bit payload
// (request index in the owner tables, registrant count in the count table).
constexpr int kIdxBits = 24;
constexpr unsigned long long kStampMax = (1ull << 40) - 1;
constexpr unsigned long long kStampReserve = 1ull << 24;   // stamps left when the tables are asked to be cleared
constexpr int64_t kMaxReqCap = (1ll << kIdxBits) - 16;

struct Info {
	uint32_t total, round1, round2, tail_parallel, tail_serial, undone, first_hard, overflow, chains,
			bar_chain, bar_class;   // barriers published by failed chains / by classification
	uint32_t tail_rounds;       // rounds the tail launches ran, summed over the call's passes
	uint32_t helped, losers;    // round 1: second registrants that helped themselves; requests it left to round 2
};

// Log of an optimistic eviction chain: every word it overwrote, in order.
constexpr int kChainLogWords = 14;      // 6 evictions x 2 words, or 5 x 2 + final placement
constexpr uint32_t kChainLogCap = 1u << 16;
constexpr uint32_t kTailLanes = 64u * 1024u;    // lanes of the largest tail launch (Tuning::tail_groups <= 64)
struct ChainLog {
	uint32_t n;
	uint32_t word[kChainLogWords];
	uint32_t old[kChainLogWords];
	uint32_t pad[3];
};

struct Ctl {
	// -- head: what the kernels of a pass read when they start (one 64-byte line)
	// Round stamps.  A pass takes three (round 1, round 2, one spare) plus what its tail's rounds use.  Its MARK
	// pass works out its round-1 stamp -- stamp_base, plus three if the pass before ended without its epilogue
	// (ep_pending: a tail launch that finds nothing to do leaves at once, k_tail) -- and publishes it in stamp_cur,
	// which nobody reads in that kernel; ROUND 1 reads stamp_cur, and its first work-group moves it to stamp_base
	// and sets ep_pending (nobody reads those in that kernel); round 2 and the tail read stamp_base; the tail's
	// epilogue, when it runs, moves stamp_base past the pass and clears ep_pending.  No kernel reads a word that
	// another work-group of the same kernel writes.
	unsigned long long stamp_base;
	unsigned long long stamp_cur;
	uint32_t ep_pending;
	uint32_t total;
	uint32_t overflow;
	uint32_t carry_lo;               // lowest request index carried over from earlier chunk passes (kNone: none)
	// plain flags, set by whoever produces such work: the four words the tail launch decides by what to do (one load)
	uint32_t any_loser;              // round 1 left requests to round 2
	uint32_t any_tail;               // the grid-wide rounds left requests to the tail
	uint32_t loose;                  // round 2: a follower at the head of its bucket did not get to lead -- its
	                                 // shadows are pending requests (the tail's first gather looks for them)
	uint32_t force_tail;             // the tail launch runs its epilogue although nothing was left to it: set by the mark
	                                 // pass (batch dropped, a host flag word that would change, stamps running low)
	uint32_t hint_stamp;             // low word of the round-1 stamp of the last pass whose mark pass walked ahead for a
	                                 // chain (prewalk_register): round 1 then reads every request's reservation
	uint32_t any_hard;               // a request was left to the tail needing an eviction chain
	uint32_t r1_wide;                // a BIG pass's round 1 runs on a lane per request (Ctl::unplanned below)
	uint32_t sorted_fail;            // low word of the round-1 stamp of the last pass whose PARTITION gave up (a bin overflowed,
	                                 // mkv_upd_apply.h): k_apply then leaves EVERY request of the pass to round 2
	uint32_t first_hard;
	uint32_t chain_n;                // chain-log entries handed out this call
	uint32_t chain_ok;               // chains committed optimistically
	uint32_t bar_chain, bar_class;
	uint32_t barrier[kMaxRegions];   // B per closed region (kNone = none)
	Info last;                       // snapshot of the previous call, for diagnostics
	uint32_t sync[2];                // arrival counters of the tail launch's two grid-wide sync points
	uint32_t sync_r2;                // ... and of the meeting behind a round 2 that ran as the launch's first phase
	uint32_t glist_n;                // entries of the tail's global work list
	uint32_t carry_next;             // lowest index the tail leaves pending (next pass's carry_lo)
	uint32_t undone_acc;             // roll-backs done by the tail's helper work-groups
	uint32_t g_sync;                 // arrivals at the sync points of the tail's grid-wide rounds (monotone)
	uint32_t g_rbar[2][kMaxRegions]; // grid-wide rounds: round barrier / lowest pending index per region,
	uint32_t g_rmin[2][kMaxRegions]; //   two sets used alternately (the idle one is reset meanwhile)
	uint32_t g_below[2][kMaxRegions];//   requests before the round's barrier that the round has left pending
	uint32_t g_progress[2], g_left[2];
	uint32_t g_dupw[2];              // a duplicate writer was tagged in the round: the election is worth its meeting
	uint32_t viol[kMaxRegions];      // tail: lowest index per region whose chain met a bucket a LATER request already used
	uint32_t vlist_n;                // ... and the violations themselves: {index, bucket} (the first kViolCap of a turn)
	uint2 vlist[kViolCap];
	uint32_t tflags;                 // tail: bit 0 = a violation was recorded
	uint32_t violations;             // order violations handled this call (each rolls back a region's later commits)
	uint32_t hard_tail;              // hard requests the grid-wide rounds left to the tail this call
	uint32_t tail_rounds_used;       // stamps the tail consumed beyond rounds 1 and 2
	uint32_t hf_last[2];             // what the host-mapped flag words hold (they are written only when they change)
	uint32_t hf_losers;              // ... and word 3: round 1 left many requests to round 2 in the last pass
	uint32_t hf_walk;                // ... and word 4: one request in sixteen (or more) ran an eviction chain in the last pass
	uint32_t dbg[8];                 // diagnostics (in-group rounds): why requests became a round's barrier
	uint32_t tail_phase[16];         // diagnostics: 10 ns ticks per tail phase, summed over the call's passes
	                                 // (8..15: inside the grid-wide rounds, work-group 0's view)
	// Requests rounds 1 and 2 have left to the tail, listed per round and work-group (tail_list_add /
	// tail_list_flush): a tail with few to do need not look for them.  No atomics on global memory: a
	// first version with ONE counter (an atomic per wave that had any) made round 2 of a 50/50 Zipf-SET
	// batch 23 -> 52 us with 650 entries -- returning atomics on one word, beside the words every wave
	// reads when it starts.
	// round 1 on at most kTlGroups work-groups: each leaves ONE 32-byte record in its place -- {entries it left to the
	// tail (may exceed kTlPer), requests it committed, of those by self-help, requests it left to round 2} and the first
	// kLlPer of the latter by index (k_tail: a short list is run by one work-group -- no scan of the pass's tags, no
	// meeting) -- instead of a count here and atomics on the commit counters below (cleared by the mark pass, like those)
	struct alignas(32) WgRec { uint4 stat; uint32_t ll[kLlPer]; } wg_rec[kTlGroups];
	// Requests of the passes so far that got NO plan from their mark pass (primary bucket full as the table stood):
	// counted where they are found, one atomic per wave that has any, never cleared.  The first work-group of a BIG
	// pass's round 1 (bit 8 of its hint) takes what has been added since the last such round (unplanned_base) and
	// leaves its verdict for the NEXT pass in r1_wide_next -- mostly plan-carrying: a lane per request pays, k_round --
	// which that pass's mark pass moves to r1_wide: no kernel reads a word another work-group of it writes, and a
	// table's load factor moves a fraction of a per cent per pass.
	struct { uint32_t n; uint32_t pad[15]; } unplanned[kShards];
	uint32_t unplanned_base;
	uint32_t r1_wide_next;
	uint32_t tl_groups[2];               // work-groups of round 1 / round 2 of this pass
	uint32_t tl_cnt[2][kTlGroups];       // entries a work-group left (may exceed kTlPer: not all listed)
	uint32_t tl[2][kTlGroups][kTlPer];
	// commit counters of rounds 1 / 2, one atomic per wave that committed anything: spread over kShards words
	// on lines of their own -- 800 atomics on the four lines that 64 ADJACENT words make took 2.5 us to get
	// through at the end of a round whose launch takes 9 (same-line atomics go one at a time, 12 ns apiece
	// on an idle chip: profiles/r04_plan_ahead_experiment.txt)
	// (at the end of the block: the fields before it keep offsets a load instruction can hold)
	struct { uint32_t n; uint32_t pad[15]; } committed[4][kShards];     // ([2]: round 1's self-help commits, [3]: its losers)
	// big passes (mkv_upd_apply.h) -- behind everything else: the words the usual kernels read keep the offsets they had
	uint32_t sorted_nlosers;         // requests k_apply left to round 2 (listed in Sorted::losers); reset by the partition
	uint32_t sorted_passes;          // passes k_apply has run so far / passes whose partition gave up (mkv_sorted_stats)
	uint32_t sorted_fallbacks;
	uint32_t hf_sorted;              // what word 5 of the host-mapped flags holds (round_body, sorted)
};

struct Req {
	uint32_t sig, hash, loc;
	uint32_t aux;       // fourth word of the request cache (kLazyBit)
};

__device__ __forceinline__ void reset_tail_scratch(Ctl *ctl);

struct ReqSrc {
	const uint32_t *const *blk_input;   // insert: device array of per-block arrays
	const uint32_t *single;             // delete: one array
	const int *blk_elem_num;
	long long single_n;
	uint32_t nblk;
	uint32_t cnt_stride;                // blk_elem_num[b * cnt_stride] (1 = the ABI's dense int[])
};

// All workers of a scheduler cycle (mkv_cycle_workers): where each worker's requests are -- the insert blocks through
// the worker's own device arrays of block pointers and counts (the ABI's), the delete requests by value.  Passed to
// the fused search launch by value: its mark work-groups read the workers' arrays themselves (mark_body, GATHER).
constexpr int kMaxWorkers = 32;
struct GatherBlocks {
	const uint32_t *const *ins_ptrs[kMaxWorkers];    // worker w's device array of block pointers
	const int *ins_cnts[kMaxWorkers];                // ... and of block counts
	uint32_t ins_first[kMaxWorkers + 1];             // worker w's first block in the cycle's (worker-major) block order
	const uint32_t *del_ptr[kMaxWorkers];
	int del_cnt[kMaxWorkers];
	int nw;
};

// ---- coherent access helpers ---------------------------------------------------
// Inside the tail work-group values written with atomics (owner / count tables)
// or by other waves of the group are read back: those loads bypass the L1.
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t ld_agent32(const uint32_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool COHERENT>
__device__ __forceinline__ void load_sig_row(const uint32_t *table, uint32_t b, uint4 &lo, uint4 &hi)
{
	const uint32_t *w = table + (uint64_t)b * kBucketWords;
	if (COHERENT) {
		const unsigned long long *q = reinterpret_cast<const unsigned long long *>(w);
		const unsigned long long a0 = ld_agent(q), a1 = ld_agent(q + 1), a2 = ld_agent(q + 2),
				a3 = ld_agent(q + 3);
		lo = make_uint4((uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32));
		hi = make_uint4((uint32_t)a2, (uint32_t)(a2 >> 32), (uint32_t)a3, (uint32_t)(a3 >> 32));
	} else {
		const uint4 *p = reinterpret_cast<const uint4 *>(w);
		lo = p[0];
		hi = p[1];
	}
}

template <bool COHERENT>
__device__ __forceinline__ uint32_t load_word(const uint32_t *table, uint64_t word)
{
	return COHERENT ? ld_agent32(table + word) : table[word];
}

__device__ __forceinline__ uint64_t sig_word(uint32_t b, uint32_t slot)
{
	return (uint64_t)b * kBucketWords + slot;
}

__device__ __forceinline__ uint64_t loc_word(uint32_t b, uint32_t slot)
{
	return (uint64_t)b * kBucketWords + kSlots + slot;
}

// A whole 64-byte bucket in registers: one round trip instead of a dependent
// chain of row loads.
struct Bkt {
	uint4 s0, s1, l0, l1;   // sig[0..3], sig[4..7], loc[0..3], loc[4..7]
};

__device__ __forceinline__ Bkt load_bucket(const uint32_t *table, uint32_t b)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(table + (uint64_t)b * kBucketWords);
	Bkt k;
	k.s0 = p[0];
	k.s1 = p[1];
	k.l0 = p[2];
	k.l1 = p[3];
	return k;
}

// ---- a bucket per QUAD of lanes and instruction ---------------------------------------------------------------
// load_bucket above is four 16-byte loads of one lane: four requests to the memory system for one 64-byte line.  Here
// the four lanes of a quad fetch the four pieces of ONE bucket with one instruction -- adjacent 16-byte pieces of a
// line, one request, as in the search kernel -- bucket after bucket of the quad's four, and hand the pieces round
// afterwards (a 4 x 4 transpose among the quad's lanes, two exchanges over DPP).  Every lane of the wave must be
// active; b = kNone: no bucket.  (The mark pass of a 2^19-request batch: 47.7 -> 46.7 us of the fused launch; a pass
// of that size is bound by what the memory side does per request -- atomics, sector writes -- not by their number on
// the way there: profiles/r05_launch_floor.txt, 10.)
template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xF, 0xF, true);
}

template <int BIT>
__device__ __forceinline__ uint32_t quad_swap(uint32_t v)      // BIT 1: lanes 0 <-> 1, 2 <-> 3; BIT 2: lanes 0 <-> 2, 1 <-> 3
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, BIT == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);
}

template <int BIT>
__device__ __forceinline__ void quad_exchange(uint32_t &lo, uint32_t &hi, bool up)
{
	// the lane whose bit BIT is clear gives its `hi` for the partner's `lo`, and the other way round
	const uint32_t y = quad_swap<BIT>(up ? lo : hi);
	lo = up ? y : lo;
	hi = up ? hi : y;
}

template <int BIT>
__device__ __forceinline__ void quad_exchange(uint4 &lo, uint4 &hi, bool up)
{
	quad_exchange<BIT>(lo.x, hi.x, up);
	quad_exchange<BIT>(lo.y, hi.y, up);
	quad_exchange<BIT>(lo.z, hi.z, up);
	quad_exchange<BIT>(lo.w, hi.w, up);
}

// lane q's a[j] <-> lane j's a[q], within each quad
__device__ __forceinline__ void quad_transpose(uint4 &a0, uint4 &a1, uint4 &a2, uint4 &a3)
{
	const uint32_t q = threadIdx.x & 3u;
	quad_exchange<1>(a0, a1, (q & 1u) != 0u);
	quad_exchange<1>(a2, a3, (q & 1u) != 0u);
	quad_exchange<2>(a0, a2, (q & 2u) != 0u);
	quad_exchange<2>(a1, a3, (q & 2u) != 0u);
}

__device__ __forceinline__ Bkt load_bucket_quad(const uint32_t *table, uint32_t b)
{
	const uint32_t q = threadIdx.x & 3u;
	const uint4 z = make_uint4(0u, 0u, 0u, 0u);
	const uint32_t b0 = quad_bcast<0>(b), b1 = quad_bcast<1>(b), b2 = quad_bcast<2>(b), b3 = quad_bcast<3>(b);
	const uint4 *t4 = reinterpret_cast<const uint4 *>(table) + q;
	uint4 a0 = z, a1 = z, a2 = z, a3 = z;
	if (b0 != kNone) a0 = t4[(uint64_t)b0 << 2];
	if (b1 != kNone) a1 = t4[(uint64_t)b1 << 2];
	if (b2 != kNone) a2 = t4[(uint64_t)b2 << 2];
	if (b3 != kNone) a3 = t4[(uint64_t)b3 << 2];
	quad_transpose(a0, a1, a2, a3);
	Bkt k;
	k.s0 = a0;
	k.s1 = a1;
	k.l0 = a2;
	k.l1 = a3;
	return k;
}

__device__ __forceinline__ uint32_t loc_at(const Bkt &k, uint32_t slot)
{
	const uint4 v = (slot & 4u) ? k.l1 : k.l0;
	const uint32_t a = (slot & 1u) ? v.y : v.x, b = (slot & 1u) ? v.w : v.z;
	return (slot & 2u) ? b : a;
}

// signature in slot `slot` of a signature row held in registers
__device__ __forceinline__ uint32_t sig_at(const uint4 &lo, const uint4 &hi, uint32_t slot)
{
	const uint4 v = (slot & 4u) ? hi : lo;
	const uint32_t a = (slot & 1u) ? v.y : v.x, b = (slot & 1u) ? v.w : v.z;
	return (slot & 2u) ? b : a;
}

// What a commit overwrote, so it can be rolled back: up to two words
// (word indices fit 32 bits: mem_p <= 33).
struct Undo {
	uint32_t word0, old0, word1, old1;
};
constexpr uint32_t kNoWord = 0xffffffffu;

// Duplicate -> overwrite loc; else empty -> claim; returns true when finished.
// `pref_sig` is the ORIGINAL request's signature (slot preference, ref :301).
template <bool COHERENT, bool LOG>
__device__ __forceinline__ bool try_place_rows(uint32_t *table, uint32_t b, const uint4 &lo, const uint4 &hi, uint32_t sig,
		uint32_t loc, uint32_t pref_sig, Undo *u)
{
	// (lo, hi: the bucket's signature row as the caller has just loaded it)
	const uint32_t m = eq_mask(lo, hi, sig);
	if (m) {
		const uint64_t w = loc_word(b, (uint32_t)__ffs((int)m) - 1u);
		if (LOG) {
			u->word0 = (uint32_t)w;
			u->old0 = load_word<COHERENT>(table, w);
			u->word1 = kNoWord;
		}
		table[w] = loc;
		return true;
	}
	const uint32_t e = eq_mask(lo, hi, 0u);
	if (e) {
		const uint32_t s = preferred_slot(e, pref_sig);
		const uint64_t ws = sig_word(b, s), wl = loc_word(b, s);
		if (LOG) {
			u->word0 = (uint32_t)ws;
			u->old0 = 0u;
			u->word1 = (uint32_t)wl;
			u->old1 = load_word<COHERENT>(table, wl);   // a deleted entry's loc stays behind (ref :461-463)
		}
		table[ws] = sig;
		table[wl] = loc;
		return true;
	}
	return false;
}
template <bool COHERENT, bool LOG>
__device__ __forceinline__ bool try_place(uint32_t *table, uint32_t b, uint32_t sig, uint32_t loc,
		uint32_t pref_sig, Undo *u)
{
	uint4 lo, hi;
	load_sig_row<COHERENT>(table, b, lo, hi);
	return try_place_rows<COHERENT, LOG>(table, b, lo, hi, sig, loc, pref_sig, u);
}

// delete in one bucket: zero the signature of every slot holding (sig, loc);
// returns whether any matched (ref gpu_hash.cu:459-468).
template <bool COHERENT>
__device__ __forceinline__ bool delete_in(uint32_t *table, uint32_t b, uint32_t sig, uint32_t loc)
{
	uint4 slo, shi;
	load_sig_row<COHERENT>(table, b, slo, shi);
	uint32_t m = eq_mask(slo, shi, sig);
	bool any = false;
	while (m) {
		const uint32_t s = (uint32_t)__ffs((int)m) - 1u;
		m &= m - 1u;
		if (load_word<COHERENT>(table, loc_word(b, s)) == loc) {
			table[sig_word(b, s)] = 0u;
			any = true;
		}
	}
	return any;
}

// What place_loaded would do on bucket b as held in k -- as an undo record (see kPlanBit); false: the bucket
// cannot take the request.
__device__ __forceinline__ bool plan_primary(uint32_t b, const Bkt &k, uint32_t sig, Undo *u)
{
	const uint32_t m = eq_mask(k.s0, k.s1, sig);
	if (m) {
		const uint32_t s = (uint32_t)__ffs((int)m) - 1u;
		u->word0 = (uint32_t)loc_word(b, s);
		u->old0 = loc_at(k, s);
		u->word1 = kNoWord;
		u->old1 = 0u;
		return true;
	}
	const uint32_t e = eq_mask(k.s0, k.s1, 0u);
	if (e) {
		const uint32_t s = preferred_slot(e, sig);
		u->word0 = (uint32_t)sig_word(b, s);
		u->old0 = 0u;
		u->word1 = (uint32_t)loc_word(b, s);
		u->old1 = loc_at(k, s);
		return true;
	}
	return false;
}

// try_place on a bucket already in registers (grid-wide rounds).
template <bool LOG>
__device__ __forceinline__ bool place_loaded(uint32_t *table, uint32_t b, const Bkt &k, uint32_t sig,
		uint32_t loc, uint32_t pref_sig, Undo *u)
{
	const uint32_t m = eq_mask(k.s0, k.s1, sig);
	if (m) {
		const uint32_t s = (uint32_t)__ffs((int)m) - 1u;
		const uint64_t w = loc_word(b, s);
		if (LOG) {
			u->word0 = (uint32_t)w;
			u->old0 = loc_at(k, s);
			u->word1 = kNoWord;
		}
		table[w] = loc;
		return true;
	}
	const uint32_t e = eq_mask(k.s0, k.s1, 0u);
	if (e) {
		const uint32_t s = preferred_slot(e, pref_sig);
		const uint64_t ws = sig_word(b, s), wl = loc_word(b, s);
		if (LOG) {
			u->word0 = (uint32_t)ws;
			u->old0 = 0u;
			u->word1 = (uint32_t)wl;
			u->old1 = loc_at(k, s);
		}
		table[ws] = sig;
		table[wl] = loc;
		return true;
	}
	return false;
}

// delete on a bucket in registers
__device__ __forceinline__ bool delete_loaded(uint32_t *table, uint32_t b, const Bkt &k, uint32_t sig,
		uint32_t loc)
{
	const uint32_t m = eq_mask(k.s0, k.s1, sig) & eq_mask(k.l0, k.l1, loc);
	uint32_t mm = m;
	while (mm) {
		const uint32_t s = (uint32_t)__ffs((int)mm) - 1u;
		mm &= mm - 1u;
		table[sig_word(b, s)] = 0u;
	}
	return m != 0u;
}

// A static-reach request, first half: the primary bucket (already in registers).
// Returns true when the request is finished there.
template <Op OP, bool LOG>
__device__ __forceinline__ bool apply_primary(uint32_t *table, const Req &r, uint32_t P, const Bkt &kP,
		Undo *u)
{
	if (OP == OP_DELETE)
		return delete_loaded(table, P, kP, r.sig, r.loc);
	return place_loaded<LOG>(table, P, kP, r.sig, r.loc, r.sig, u);
}

// Second half: the alternate bucket.  Returns false (nothing written) only for a
// cuckoo request that needs an eviction chain.
template <Op OP, bool LOG>
__device__ __forceinline__ bool apply_alternate(uint32_t *table, const Req &r, uint32_t A, Undo *u)
{
	if (OP == OP_DELETE) {
		delete_in<false>(table, A, r.sig, r.loc);
		return true;
	}
	if (try_place<false, LOG>(table, A, r.sig, r.loc, r.sig, u))
		return true;
	if (OP == OP_2CHOICE) {
		const uint64_t w = sig_word(A, r.sig & 7u);
		if (LOG) {
			u->word0 = (uint32_t)w;
			u->old0 = table[w];
			u->word1 = kNoWord;
		}
		table[w] = r.sig;                           // ref :197-209
		return true;
	}
	return false;
}

// A request whose reach is static.  Returns false (nothing written) only for a
// cuckoo request that needs an eviction chain.
template <Op OP, bool COHERENT, bool LOG>
__device__ __forceinline__ bool apply_static(uint32_t *table, const Req &r, uint32_t P, uint32_t A, Undo *u)
{
	if (OP == OP_DELETE) {
		if (!delete_in<COHERENT>(table, P, r.sig, r.loc))
			delete_in<COHERENT>(table, A, r.sig, r.loc);
		return true;
	}
	if (try_place<COHERENT, LOG>(table, P, r.sig, r.loc, r.sig, u))
		return true;
	if (try_place<COHERENT, LOG>(table, A, r.sig, r.loc, r.sig, u))
		return true;
	if (OP == OP_2CHOICE) {
		// no room: the signature alone replaces slot sig&7, stale loc stays (ref :197-209)
		table[sig_word(A, r.sig & 7u)] = r.sig;
		return true;
	}
	return false;
}

__device__ __forceinline__ Req req_of(const uint4 &rc)
{
	Req r;
	r.sig = rc.x;
	r.hash = rc.y;
	r.loc = rc.z;
	r.aux = rc.w;
	return r;
}

// ---- reservation tables -----------------------------------------------------
__device__ __forceinline__ void bump(unsigned long long *cnt, uint32_t b, unsigned long long stamp)
{
	atomicMax(&cnt[b], stamp << kIdxBits);   // a stale entry restarts at (stamp, 0)
	atomicAdd(&cnt[b], 1ull);
}

// Count entries: stamp | flag "a registrant has this bucket as its ALTERNATE (or only)
// bucket" | registrant count.  Only such a registrant can ever evict from the bucket
// (a chain starts at the alternate bucket; the buckets it goes on to are unregistered).
constexpr unsigned long long kAltFlag = 1ull << (kIdxBits - 1);

__device__ __forceinline__ void bump_alt(unsigned long long *cnt, uint32_t b, unsigned long long stamp)
{
	atomicMax(&cnt[b], stamp << kIdxBits);
	atomicAdd(&cnt[b], 1ull);
	atomicOr(&cnt[b], kAltFlag);
}

__device__ __forceinline__ uint32_t cnt_of(unsigned long long v, unsigned long long stamp)
{
	return (v >> kIdxBits) == stamp ? (uint32_t)(v & (kAltFlag - 1ull)) : 0u;
}

__device__ __forceinline__ bool alt_of(unsigned long long v, unsigned long long stamp)
{
	return (v >> kIdxBits) == stamp && (v & kAltFlag) != 0ull;
}

// Tail rounds also keep, per bucket, the LOWEST index among the registrants that have it as their
// alternate bucket (same keys as the reservations, in the table round 2 used): only a request
// BEFORE idx can evict from the bucket before idx's turn.
__device__ __forceinline__ bool alt_before(const unsigned long long *altmin, uint32_t b, unsigned long long stamp,
		uint32_t idx)
{
	const unsigned long long v = __hip_atomic_load(&altmin[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return (v >> kIdxBits) == kStampMax - stamp && (uint32_t)(v & ((1ull << kIdxBits) - 1ull)) <= idx;
}

// owner key of request idx in the round stamped `stamp`: newer rounds compare lower
__device__ __forceinline__ unsigned long long owner_key(unsigned long long stamp, uint32_t idx)
{
	return ((kStampMax - stamp) << kIdxBits) | idx;
}

// atomicMin(&owner[b], key) for every active lane, but lanes of a wave that target
// the same bucket as the wave's first pending lane are served by ONE atomic carrying
// the smallest key among them (two peeling steps, then one atomic per lane).  Under
// the reference's LOCAL_TEST keying every key of a region has the same alternate
// bucket (hash ^ sig has no region-local bits, src/mega_recv.c:700-704): without
// this all registrations of a batch would queue on 8 words.
__device__ __forceinline__ void atomic_min_dedup(unsigned long long *owner, uint32_t b,
		unsigned long long key, bool active)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		// smallest key of the group (wave reduction over the group's lanes; a group of one
		// needs none)
		unsigned long long k = same ? key : ~0ull;
		if (__popcll(__ballot(same)) > 1) {
			for (int off = 32; off > 0; off >>= 1) {
				const unsigned long long o = __shfl_xor(k, off);
				k = o < k ? o : k;
			}
		}
		if ((int)(threadIdx.x & 63u) == first)
			atomicMin(&owner[fb], k);
		active = active && !same;
	}
	if (active)
		atomicMin(&owner[b], key);
}

__device__ __forceinline__ void reserve(unsigned long long *owner, uint32_t P, uint32_t A,
		unsigned long long key)
{
	atomicMin(&owner[P], key);
	if (A != P)
		atomicMin(&owner[A], key);
}

// wave-convergent form used by the mark pass: every lane of the wave calls it
__device__ __forceinline__ void reserve_wave(unsigned long long *owner, uint32_t P, uint32_t A,
		unsigned long long key, bool active)
{
	atomic_min_dedup(owner, P, key, active);
	atomic_min_dedup(owner, A, key, active && A != P);
}

// wave-convergent registrant count: lanes of a wave that count on the same bucket as the
// wave's first pending lane are served by one add of their number (two peeling steps, then
// one per lane) -- duplicates of a hot key would otherwise queue on one word
__device__ __forceinline__ void bump_wave(unsigned long long *cnt, uint32_t b, unsigned long long stamp,
		bool alt, bool active)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		const unsigned long long group = __ballot(same), alts = __ballot(same && alt);
		if ((int)(threadIdx.x & 63u) == first) {
			atomicMax(&cnt[fb], stamp << kIdxBits);
			atomicAdd(&cnt[fb], (unsigned long long)__popcll(group));
			if (alts)
				atomicOr(&cnt[fb], kAltFlag);
		}
		active = active && !same;
	}
	if (active) {
		if (alt)
			bump_alt(cnt, b, stamp);
		else
			bump(cnt, b, stamp);
	}
}

// wave-convergent atomicMax with the same folding
__device__ __forceinline__ void atomic_max_dedup(unsigned long long *words, uint32_t b, unsigned long long key,
		bool active)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		unsigned long long k = same ? key : 0ull;
		if (__popcll(__ballot(same)) > 1) {
			for (int off = 32; off > 0; off >>= 1) {
				const unsigned long long o = __shfl_xor(k, off);
				k = o > k ? o : k;
			}
		}
		if ((int)(threadIdx.x & 63u) == first)
			atomicMax(&words[fb], k);
		active = active && !same;
	}
	if (active)
		atomicMax(&words[b], key);
}

// Work-group-wide folding of reservation atomics (grid-wide tail rounds): thousands of pending
// SETs of a few hot keys would otherwise queue on the words of those keys' buckets -- a wave
// folds the lanes of its first two buckets only.  The work-group gathers {bucket, smallest or
// largest key, registrant count, alternate flag} in a small hash table in LDS (two probes; an
// entry that finds no slot acts alone on the global word) and sends ONE set of atomics per
// bucket and turn.  The table lives in the LDS of the work list, which grid-wide rounds keep
// in global memory.
constexpr int kFoldLog2 = 11;           // grid-wide tail rounds: 2048 slots for 1024 lanes x 2 buckets
constexpr uint32_t kFoldSlots = 1u << kFoldLog2;
constexpr int kMarkFoldLog2 = 9;        // mark pass: 512 slots for 256 lanes x 2 buckets
constexpr int kMarkFoldBlocks = 256;    // ... for calls with at most this many insert blocks
struct Fold {
	uint32_t *tb;               // bucket of the slot (kNone: free)
	uint32_t *tc;               // registrants entered (bit 31: one of them has the bucket as its alternate); may be null
	unsigned long long *tk;     // smallest (largest) key entered
};

template <bool MAX, int LOG2>
__device__ __forceinline__ void fold_clear(const Fold &f)
{
	for (uint32_t s = threadIdx.x; s < (1u << LOG2); s += blockDim.x) {
		f.tb[s] = kNone;
		if (f.tc)
			f.tc[s] = 0u;
		f.tk[s] = MAX ? 0ull : ~0ull;
	}
	__syncthreads();
}

// true: entered in the table; false: no slot -- the caller acts on the global words itself
template <bool MAX, int LOG2>
__device__ __forceinline__ bool fold_enter(const Fold &f, uint32_t b, unsigned long long key, bool count, bool alt)
{
	uint32_t slot = (b * 2654435761u) >> (32 - LOG2);
	for (int probe = 0; probe < 2; probe++) {
		const uint32_t old = atomicCAS(&f.tb[slot], kNone, b);
		if (old == kNone || old == b) {
			if (MAX)
				atomicMax(&f.tk[slot], key);
			else
				atomicMin(&f.tk[slot], key);
			if (count)
				atomicAdd(&f.tc[slot], 1u);
			if (alt)
				atomicOr(&f.tc[slot], 0x80000000u);
			return true;
		}
		slot = (slot + 1u) & ((1u << LOG2) - 1u);
	}
	return false;
}

// one set of atomics per occupied slot; the table is left clear (NEXT_MAX says for which use)
template <bool MAX, bool NEXT_MAX, int LOG2>
__device__ __forceinline__ void fold_flush(const Fold &f, unsigned long long *words, unsigned long long *cnt,
		unsigned long long stamp)
{
	__syncthreads();
	for (uint32_t s = threadIdx.x; s < (1u << LOG2); s += blockDim.x) {
		const uint32_t b = f.tb[s];
		if (b != kNone) {
			if (MAX)
				atomicMax(&words[b], f.tk[s]);
			else
				atomicMin(&words[b], f.tk[s]);
			const uint32_t c = f.tc ? f.tc[s] : 0u;
			if (c & 0x7fffffffu) {
				atomicMax(&cnt[b], stamp << kIdxBits);   // a stale entry restarts at (stamp, 0)
				atomicAdd(&cnt[b], (unsigned long long)(c & 0x7fffffffu));
				if (c >> 31)
					atomicOr(&cnt[b], kAltFlag);
			}
			f.tb[s] = kNone;
			if (f.tc)
				f.tc[s] = 0u;
		}
		f.tk[s] = NEXT_MAX ? 0ull : ~0ull;
	}
	__syncthreads();
}

// atomic_min_dedup with the fold table behind it: the lanes of a wave on the bucket of the wave's
// first pending lane go as one entry (two peeling steps: a whole wave on one bucket -- the
// reference's LOCAL_TEST keying -- must not queue on one LDS word either), the rest one each.
// (Batches of distinct keys pay nothing measurable for the detour through LDS: A/B on one box,
// tools/gpu_ab_bench.sh.)
template <int LOG2>
__device__ __forceinline__ void fold_min_dedup(const Fold &f, unsigned long long *owner, uint32_t b,
		unsigned long long key, bool active)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		unsigned long long k = same ? key : ~0ull;
		if (__popcll(__ballot(same)) > 1) {
			for (int off = 32; off > 0; off >>= 1) {
				const unsigned long long o = __shfl_xor(k, off);
				k = o < k ? o : k;
			}
		}
		if ((int)(threadIdx.x & 63u) == first && !fold_enter<false, LOG2>(f, fb, k, false, false))
			atomicMin(&owner[fb], k);
		active = active && !same;
	}
	if (active && !fold_enter<false, LOG2>(f, b, key, false, false))
		atomicMin(&owner[b], key);
}

// ... and atomic_max_dedup with the fold table behind it
template <int LOG2>
__device__ __forceinline__ void fold_max_dedup(const Fold &f, unsigned long long *words, uint32_t b,
		unsigned long long key, bool active)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		unsigned long long k = same ? key : 0ull;
		if (__popcll(__ballot(same)) > 1) {
			for (int off = 32; off > 0; off >>= 1) {
				const unsigned long long o = __shfl_xor(k, off);
				k = o > k ? o : k;
			}
		}
		if ((int)(threadIdx.x & 63u) == first && !fold_enter<true, LOG2>(f, fb, k, false, false))
			atomicMax(&words[fb], k);
		active = active && !same;
	}
	if (active && !fold_enter<true, LOG2>(f, b, key, false, false))
		atomicMax(&words[b], key);
}

// ---- the SECOND-lowest registrant of a bucket (mark pass) ---------------------------------------
// Round 1 lets the second-lowest registrant j of a bucket P commit beside the lowest one, c, when it can
// tell what c will do there (self-help, round_body).  For that j must KNOW it is the second: every
// registration of the mark pass on a request's own buckets -- primary, alternate -- is a
// RETURNING atomicMin, and whoever sees two keys of this pass on a bucket (its own and the one it
// displaced or lost to) enters the larger one in a second table, `second`, with another atomicMin.
// Every registrant but the lowest is the larger one of such a pair at some point (it finds a smaller key
// there, or is displaced by one later), so second[b] ends up holding the lowest of them: the second-lowest
// registrant.  Where registrations are folded -- the lanes of a wave on one bucket, the entries of a
// work-group's table -- the fold keeps the two lowest keys it has seen and hands both on.  (The insert
// blocks of a batch are its closed regions: two requests of one bucket are 6 500 indices apart at most,
// not 52 000, and meet in one work-group's turn of 256 four times in a hundred.)
// (Lost: the flags of the requests that are NOT the lowest registrant of a bucket they registered on, kPlanBit)
struct Lost {
	uint32_t *flag;     // a word per request (null: not kept -- deletes)
	uint32_t s32;       // low half of the pass's round-1 stamp
};

__device__ __forceinline__ void flag_lost(const Lost &lost, unsigned long long key)
{
	if (lost.flag)
		lost.flag[(uint32_t)(key & ((1ull << kIdxBits) - 1ull))] = lost.s32;
}

__device__ __forceinline__ void note_second(unsigned long long *second, uint32_t b, unsigned long long old,
		unsigned long long key, const Lost &lost)
{
	// (same stamp field: an entry of this pass)
	if ((old >> kIdxBits) == (key >> kIdxBits) && old != key) {
		const unsigned long long larger = old > key ? old : key;
		atomicMin(&second[b], larger);
		flag_lost(lost, larger);
	}
}

// key2: a key the caller knows to be another, higher registrant of the bucket (~0: none; flagged by the caller)
__device__ __forceinline__ void register_one(unsigned long long *owner, unsigned long long *second, uint32_t b,
		unsigned long long key, const Lost &lost, unsigned long long key2 = ~0ull)
{
	note_second(second, b, atomicMin(&owner[b], key), key, lost);
	if (key2 != ~0ull)
		atomicMin(&second[b], key2);
}

// the two lowest keys among the lanes of the wave for which `same` holds
__device__ __forceinline__ void wave_two_lowest(unsigned long long key, bool same, unsigned long long &k, unsigned long long &k2)
{
	k = same ? key : ~0ull;
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned long long o = __shfl_xor(k, off);
		k = o < k ? o : k;
	}
	k2 = (same && key != k) ? key : ~0ull;
	for (int off = 32; off > 0; off >>= 1) {
		const unsigned long long o = __shfl_xor(k2, off);
		k2 = o < k2 ? o : k2;
	}
}

// atomic_min_dedup with the second table behind it (mark pass without the fold table)
__device__ __forceinline__ void register_dedup(unsigned long long *owner, unsigned long long *second, uint32_t b,
		unsigned long long key, bool active, const Lost &lost)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		unsigned long long k = key, k2 = ~0ull;
		if (__popcll(__ballot(same)) > 1)
			wave_two_lowest(key, same, k, k2);
		if (same && key != k)
			flag_lost(lost, key);                   // (not the lowest of the lanes that go as one)
		if ((int)(threadIdx.x & 63u) == first)
			register_one(owner, second, fb, k, lost, k2);
		active = active && !same;
	}
	if (active)
		register_one(owner, second, b, key, lost);
}

// fold_enter / fold_min_dedup / fold_flush of the mark pass with the second table behind them: a slot
// keeps the lowest key entered (tk) and the lowest of the others (tk2)
template <int LOG2>
__device__ __forceinline__ bool mark_fold_enter(const Fold &f, unsigned long long *tk2, uint32_t b, unsigned long long key,
		unsigned long long key2, const Lost &lost)
{
	uint32_t slot = (b * 2654435761u) >> (32 - LOG2);
	for (int probe = 0; probe < 2; probe++) {
		const uint32_t old = atomicCAS(&f.tb[slot], kNone, b);
		if (old == kNone || old == b) {
			const unsigned long long prev = atomicMin(&f.tk[slot], key);
			unsigned long long other = prev == ~0ull ? ~0ull : (prev > key ? prev : key);
			if (other != ~0ull && prev != key)
				flag_lost(lost, other);             // (the larger of two keys that met in the slot)
			other = key2 < other ? key2 : other;
			if (other != ~0ull)
				atomicMin(&tk2[slot], other);
			return true;
		}
		slot = (slot + 1u) & ((1u << LOG2) - 1u);
	}
	return false;
}

template <int LOG2>
__device__ __forceinline__ void mark_fold_min_dedup(const Fold &f, unsigned long long *tk2, unsigned long long *owner,
		unsigned long long *second, uint32_t b, unsigned long long key, bool active, const Lost &lost)
{
#pragma unroll
	for (int peel = 0; peel < 2; peel++) {
		const unsigned long long pend = __ballot(active);
		if (!pend)
			return;
		const int first = __ffsll((long long)pend) - 1;
		const uint32_t fb = __shfl(b, first);
		const bool same = active && b == fb;
		unsigned long long k = key, k2 = ~0ull;
		if (__popcll(__ballot(same)) > 1)
			wave_two_lowest(key, same, k, k2);
		if (same && key != k)
			flag_lost(lost, key);                   // (not the lowest of the lanes that go as one)
		if ((int)(threadIdx.x & 63u) == first && !mark_fold_enter<LOG2>(f, tk2, fb, k, k2, lost))
			register_one(owner, second, fb, k, lost, k2);
		active = active && !same;
	}
	if (active && !mark_fold_enter<LOG2>(f, tk2, b, key, ~0ull, lost))
		register_one(owner, second, b, key, lost);
}

template <int LOG2>
__device__ __forceinline__ void mark_fold_flush(const Fold &f, unsigned long long *tk2, unsigned long long *owner,
		unsigned long long *second, const Lost &lost)
{
	__syncthreads();
	for (uint32_t s = threadIdx.x; s < (1u << LOG2); s += blockDim.x) {
		const uint32_t b = f.tb[s];
		if (b != kNone) {
			register_one(owner, second, b, f.tk[s], lost, tk2[s]);
			f.tb[s] = kNone;
			tk2[s] = ~0ull;
		}
		f.tk[s] = ~0ull;
	}
	__syncthreads();
}

// statistics: one atomic per wave, spread over kShards words 16 words apart (Ctl::committed)
__device__ __forceinline__ void count_add(uint32_t *shards, uint32_t n)
{
	for (int off = 32; off > 0; off >>= 1)
		n += __shfl_down(n, off);
	if ((threadIdx.x & 63u) == 0u && n)
		atomicAdd(&shards[((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kShards - 1)) * 16u], n);
}

// Mark pass, cuckoo: a request that LOOKS hard in the table as it stands (P and A full,
// no duplicate) also registers on the buckets its eviction chain would go on to -- read-only
// walk, same key, same owner table.  If it is still hard when its turn comes and the chain
// still runs that way, the buckets are its own already (try_chain meets its own key), and the
// requests of this batch that have one of them as their P or A line up behind or before it
// through the ordinary reservation instead of being committed under the chain's feet.  Only
// a hint: nothing depends on the walk being right (the table may change before the turn).
__device__ __forceinline__ void prewalk_register(const uint32_t *table, const DevGeom &g, const Req &r, uint32_t P,
		uint32_t A, unsigned long long *owner, unsigned long long key, uint4 lo, uint4 hi, uint32_t *hint_stamp, const Lost &lost,
		bool flagging)
{
	// (lo, hi: the signature row of P, loaded by the caller)
	if (eq_mask(lo, hi, r.sig) | eq_mask(lo, hi, 0u))
		return;
	if (A != P) {
		load_sig_row<false>(table, A, lo, hi);
		if (eq_mask(lo, hi, r.sig) | eq_mask(lo, hi, 0u))
			return;
	}
	const uint32_t v = r.sig & 7u;              // victim slot: ORIGINAL signature (ref :360)
	// (flagging: the walk's registrations are returning atomics too and flag whom they displace, like all the
	// others -- the plans of the pass stand, kPlanBit; else they are not, and the pass says so: round 1 then reads
	// every reservation.  The host asks for the second form while one request in sixteen or more has been walking a
	// chain (the tail launch's epilogue says so through its host-mapped word 4) -- a well-filled table, thousands of
	// walks per pass: the returning atomic per step made the fused launch a third longer there.)
	if (g.max_cuckoo != 0u && !flagging)
		*hint_stamp = lost.s32;
	for (uint32_t k = 0; k < g.max_cuckoo; k++) {
		const uint4 q = (v & 4u) ? hi : lo;
		const uint32_t a = (v & 1u) ? q.y : q.x, c = (v & 1u) ? q.w : q.z;
		const uint32_t vsig = (v & 2u) ? c : a;
		const uint32_t nb = alt_bucket(r.hash, vsig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (ref :334)
		// (not noted in the table of second registrants: a returning atomic per step made the walk, and with
		// it the fused launch of a well-filled table, a third longer -- 34 -> 46 us at load factor 0.85.
		// A bucket's second registrant that helps itself behind a chain's hint commits where it would
		// have committed in round 2, and the chain, should it ever come, finds it like any later commit.)
		if (flagging) {
			const unsigned long long old = atomicMin(&owner[nb], key);
			load_sig_row<false>(table, nb, lo, hi);
			if ((old >> kIdxBits) == (key >> kIdxBits) && old > key)
				flag_lost(lost, old);
		} else {
			atomicMin(&owner[nb], key);
			load_sig_row<false>(table, nb, lo, hi);
		}
		if (eq_mask(lo, hi, vsig) | eq_mask(lo, hi, 0u))
			return;
	}
}

}  // namespace mkv
