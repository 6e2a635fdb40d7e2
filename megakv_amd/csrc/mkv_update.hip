// mkv_update.hip -- batched insert (cuckoo / two-choice) and delete on gfx950.
//
// Replaces hash_insert_cuckoo (libgpuhash/gpu_hash.cu:231-433),
// hash_insert_2choice (:77-229) and hash_delete (:435-480).
//
// RESULT DEFINITION.  The reference kernels keep many requests in flight with
// no atomics; their outcome is only defined when in-flight requests touch
// disjoint slots.  This implementation produces, for EVERY input, the table
// the reference produces when requests run one at a time in (block, index)
// order -- "canonical order" below -- and does so in parallel.
//
// HOW.  A request's reach is static except for the cuckoo eviction chain: it
// touches its primary bucket P and at most its alternate bucket A, both known
// from the request alone; only a request that finds P and A full with no
// duplicate ("hard") walks on to data-dependent buckets of its region.
// Requests commute unless they share a bucket, so per bucket they must run in
// index order and nothing else has to be ordered.  The work is done in
// reservation rounds over a per-bucket owner table in HBM (64-bit atomicMin of
// {round stamp, canonical index}; stamps make stale entries lose, the table is
// never cleared), committed optimistically:
//
//   K1 mark     every request registers on P -- and on A unless P has room to spare (a
//               new key with two free slots in P registers once; its late use of A is
//               checked like a chain's step) -- and is copied into a 16-byte request
//               cache.  A cuckoo request that looks hard
//               in the table as it stands also registers on the buckets its chain
//               would go on to (read-only walk): later requests of the batch on
//               those buckets then line up behind it.
//   K2 round 1  a request that is the lowest registrant of the buckets it needs
//               executes with plain loads/stores and logs what it overwrote; a hard
//               request runs its chain if every further bucket is its own or
//               nobody's.  Losers re-register for round 2.  A hard request whose
//               chain meets somebody else is left to the tail -- nobody waits for it.
//               Losers whose bucket's lowest registrant holds the SAME key (followers)
//               elect the last of them: only its location survives.
//   K3 round 2  the same over the losers, who registered in a second owner table.  The
//               first follower of a key commits the last one's location for all of
//               them (a group; the others are its shadows).
//   K4 tail     one launch of 32 work-groups of 1024 lanes (the helpers leave at
//               once unless there is much to scan).  Reservation rounds over what
//               is left (register / decide / commit with work-group barriers; the
//               lowest request of a region that is hard, or might turn hard, is the
//               round's barrier for that region: the tail never commits ahead of
//               it, so its commits are final), then the in-order replay per region.
//
// Rounds 1 and 2 commit without asking whether an EARLIER pending request will
// turn hard and walk its chain into the same bucket.  That is checked when the
// chain walks (tail): WHO can have committed on a bucket is in the two owner
// tables (its lowest registrant of either round: two candidates, two loads);
// a later commit there is an order violation, and exactly the commits involved
// -- those later ones on that bucket, and what ran after them on their buckets
// -- are taken back from their logs and run again behind the chain
// (select_from / take_back).  A group's commit counts with the index of its last
// member; it is never taken back for a request above its leader but cut there
// (truncate_groups).
//
// "Holds a duplicate" only counts as "cannot turn hard" while nothing can evict
// that duplicate before the request's turn: never for signature 0 (which matches
// empty slots anybody may fill); only if the request owns the bucket or no
// registrant has it as its alternate bucket; and a chain that takes a bucket
// over from a later pending request bars that request from the round.
//
// Why this equals the canonical order: DESIGN.md 5.2.  No host synchronisation
// anywhere: request counts are read on the device (the reference passes them as
// a device array, gpu_hash.cu:236-237), grids are fixed and stride over them,
// and the round stamps live on the device so a captured hipGraph can be replayed.
//
// The per-lane atomicCAS the north star suggests for slot claims is not needed:
// exclusivity comes from the reservation, so a commit is two plain stores.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include <hip/hip_ext.h>

#include "mkv_common.h"
#include "mkv_search_body.h"

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
constexpr int kTailReplayMax = 24;  // the tail replays this many unfinished requests in order without trying rounds
constexpr int kTailAloneSpan = 131072; // ... or beyond this many request tags to scan
constexpr int kTailAloneMax = 1024; // the tail keeps its helper work-groups only beyond this many unfinished requests
constexpr int kSortCap = 8192;      // tail requests held in LDS; beyond: state scan
constexpr int kViolCap = 128;       // order violations recorded per turn of the tail (more are met again later)
constexpr uint8_t ST_SELECT = 0x80; // tail, transient: flag on a commit tag -- selected for roll-back
constexpr int kTailThreads = 1024;  // lanes per work-group of the tail launch (its helpers are force-inlined: the
                                    // launch stays free of scratch memory and calls)
constexpr uint32_t kLateSticky = 16; // calls for which a call whose chains ran late counts as "chains lately" (diagnostics)
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
	unsigned long long next_stamp;   // persists across calls
	unsigned long long stamp_base;   // stamp of round 1 of the current call
	uint32_t total;
	uint32_t overflow;
	uint32_t any_loser;              // plain flags, set by whoever produces such work
	uint32_t any_tail;
	uint32_t loose;                  // round 2: a follower at the head of its bucket did not get to lead -- its
	                                 // shadows are pending requests (the tail's first gather looks for them)
	uint32_t first_hard;
	uint32_t chain_n;                // chain-log entries handed out this call
	uint32_t chain_ok;               // chains committed optimistically
	uint32_t bar_chain, bar_class;
	uint32_t carry_lo;               // lowest request index carried over from earlier chunk passes (kNone: none)
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
	uint32_t any_hard;               // a request was left to the tail needing an eviction chain
	uint32_t viol[kMaxRegions];      // tail: lowest index per region whose chain met a bucket a LATER request already used
	uint32_t vlist_n;                // ... and the violations themselves: {index, bucket} (the first kViolCap of a turn)
	uint2 vlist[kViolCap];
	uint32_t tflags;                 // tail: bit 0 = a violation was recorded
	uint32_t violations;             // order violations handled this call (each rolls back a region's later commits)
	uint32_t hard_tail;              // hard requests the grid-wide rounds left to the tail this call
	uint32_t late_score;             // > 0: chains ran late in one of the last kLateSticky calls (persists across calls)
	uint32_t tail_rounds_used;       // stamps the tail consumed beyond rounds 1 and 2
	uint32_t hf_last[2];             // what the host-mapped flag words hold (they are written only when they change)
	uint32_t hf_losers;              // ... and word 3: round 1 left many requests to round 2 in the last pass
	uint32_t dbg[8];                 // diagnostics (in-group rounds): why requests became a round's barrier
	uint32_t tail_phase[16];         // diagnostics: 10 ns ticks per tail phase, summed over the call's passes
	                                 // (8..15: inside the grid-wide rounds, work-group 0's view)
	// Requests rounds 1 and 2 have left to the tail, listed per round and work-group (tail_list_add /
	// tail_list_flush): a tail with few to do need not look for them.  No atomics on global memory: a
	// first version with ONE counter (an atomic per wave that had any) made round 2 of a 50/50 Zipf-SET
	// batch 23 -> 52 us with 650 entries -- returning atomics on one word, beside the words every wave
	// reads when it starts.
	uint32_t tl_groups[2];               // work-groups of round 1 / round 2 of this pass
	uint32_t tl_cnt[2][kTlGroups];       // entries a work-group left (may exceed kTlPer: not all listed)
	uint32_t tl[2][kTlGroups][kTlPer];
	// commit counters of rounds 1 / 2, one atomic per wave that committed anything: spread over kShards words
	// on lines of their own -- 800 atomics on the four lines that 64 ADJACENT words make took 2.5 us to get
	// through at the end of a round whose launch takes 9 (same-line atomics go one at a time, 12 ns apiece
	// on an idle chip: profiles/r04_plan_ahead_experiment.txt)
	// (at the end of the block: the fields before it keep offsets a load instruction can hold)
	struct { uint32_t n; uint32_t pad[15]; } committed[4][kShards];     // ([2]: round 1's self-help commits, [3]: its losers)
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
__device__ __forceinline__ void note_second(unsigned long long *second, uint32_t b, unsigned long long old,
		unsigned long long key)
{
	// (same stamp field: an entry of this pass)
	if ((old >> kIdxBits) == (key >> kIdxBits) && old != key)
		atomicMin(&second[b], old > key ? old : key);
}

// key2: a key the caller knows to be another, higher registrant of the bucket (~0: none)
__device__ __forceinline__ void register_one(unsigned long long *owner, unsigned long long *second, uint32_t b,
		unsigned long long key, unsigned long long key2 = ~0ull)
{
	note_second(second, b, atomicMin(&owner[b], key), key);
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
		unsigned long long k = key, k2 = ~0ull;
		if (__popcll(__ballot(same)) > 1)
			wave_two_lowest(key, same, k, k2);
		if ((int)(threadIdx.x & 63u) == first)
			register_one(owner, second, fb, k, k2);
		active = active && !same;
	}
	if (active)
		register_one(owner, second, b, key);
}

// fold_enter / fold_min_dedup / fold_flush of the mark pass with the second table behind them: a slot
// keeps the lowest key entered (tk) and the lowest of the others (tk2)
template <int LOG2>
__device__ __forceinline__ bool mark_fold_enter(const Fold &f, unsigned long long *tk2, uint32_t b, unsigned long long key,
		unsigned long long key2)
{
	uint32_t slot = (b * 2654435761u) >> (32 - LOG2);
	for (int probe = 0; probe < 2; probe++) {
		const uint32_t old = atomicCAS(&f.tb[slot], kNone, b);
		if (old == kNone || old == b) {
			const unsigned long long prev = atomicMin(&f.tk[slot], key);
			unsigned long long other = prev == ~0ull ? ~0ull : (prev > key ? prev : key);
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
		unsigned long long *second, uint32_t b, unsigned long long key, bool active)
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
		if ((int)(threadIdx.x & 63u) == first && !mark_fold_enter<LOG2>(f, tk2, fb, k, k2))
			register_one(owner, second, fb, k, k2);
		active = active && !same;
	}
	if (active && !mark_fold_enter<LOG2>(f, tk2, b, key, ~0ull))
		register_one(owner, second, b, key);
}

template <int LOG2>
__device__ __forceinline__ void mark_fold_flush(const Fold &f, unsigned long long *tk2, unsigned long long *owner,
		unsigned long long *second)
{
	__syncthreads();
	for (uint32_t s = threadIdx.x; s < (1u << LOG2); s += blockDim.x) {
		const uint32_t b = f.tb[s];
		if (b != kNone) {
			register_one(owner, second, b, f.tk[s], tk2[s]);
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
		uint32_t A, unsigned long long *owner, unsigned long long key, uint4 lo, uint4 hi)
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
	for (uint32_t k = 0; k < g.max_cuckoo; k++) {
		const uint4 q = (v & 4u) ? hi : lo;
		const uint32_t a = (v & 1u) ? q.y : q.x, c = (v & 1u) ? q.w : q.z;
		const uint32_t vsig = (v & 2u) ? c : a;
		const uint32_t nb = alt_bucket(r.hash, vsig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (ref :334)
		// (not noted in the table of second registrants: a returning atomic per step made the walk, and with
		// it the fused launch of a well-filled table, a third longer -- 34 -> 46 us at load factor 0.85.
		// A bucket's second registrant that helps itself behind a chain's hint commits where it would
		// have committed in round 2, and the chain, should it ever come, finds it like any later commit.)
		atomicMin(&owner[nb], key);
		load_sig_row<false>(table, nb, lo, hi);
		if (eq_mask(lo, hi, vsig) | eq_mask(lo, hi, 0u))
			return;
	}
}

// ---- K1: mark -----------------------------------------------------------------
template <Op OP>
__device__ __forceinline__ void mark_body(uint32_t block, uint32_t nblocks, Ctl *ctl, const ReqSrc &src,
		const DevGeom &g, unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state, uint32_t req_cap,
		uint32_t lo_idx, uint32_t hi_idx, const uint32_t *table, uint32_t lazy_free)
{
	__shared__ uint32_t s_prefix[kMaxBlocks + 1];
	__shared__ const uint32_t *s_ptr[kMaxBlocks];
	__shared__ uint32_t s_total;
	// Registrations of one turn of the work-group are folded per bucket (Zipf SET keys: the hot
	// keys' buckets; the reference's LOCAL_TEST keying: one alternate bucket per region).  The
	// table lies in the unused end of the block pointers' LDS -- the launch this rides in
	// (k_search_mark) should not cost the search work-groups more LDS -- so a call with more
	// than kMarkFoldBlocks insert blocks (the reference uses 8) goes without it.
	static_assert((kMaxBlocks - kMarkFoldBlocks) * sizeof(uint32_t *) >= (12u << kMarkFoldLog2), "fold table in s_ptr");
	const bool use_fold = src.single || src.nblk <= (uint32_t)kMarkFoldBlocks;
	Fold fold;
	fold.tk = reinterpret_cast<unsigned long long *>(s_ptr + kMarkFoldBlocks);
	fold.tb = reinterpret_cast<uint32_t *>(fold.tk + (1u << kMarkFoldLog2));
	fold.tc = nullptr;
	__shared__ unsigned long long s_tk2[1u << kMarkFoldLog2];      // (the fold's second-lowest keys: mark_fold_enter)
	if (use_fold) {
		for (uint32_t q = threadIdx.x; q < (1u << kMarkFoldLog2); q += blockDim.x)
			s_tk2[q] = ~0ull;
		fold_clear<false, kMarkFoldLog2>(fold);
	}
	const uint32_t t = threadIdx.x;
	const uint32_t nblk = src.nblk;
	const unsigned long long stamp = ctl->stamp_base;
	if (src.single) {
		if (t == 0) {
			const unsigned long long n = src.single_n > 0 ? (unsigned long long)src.single_n : 0ull;
			s_prefix[0] = 0;
			s_total = n > req_cap ? kNone : (uint32_t)n;
		}
	} else {
		for (uint32_t b = t; b < nblk; b += blockDim.x) {
			const int c = src.blk_elem_num[(size_t)b * src.cnt_stride];
			s_ptr[b] = src.blk_input[b];        // fetched beside the counts: one round trip
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
	}
	__syncthreads();
	const bool over = s_total == kNone;
	const uint32_t total = over ? 0u : s_total;
	if (block == 0 && t == 0) {
		ctl->total = total;
		ctl->overflow = over ? 1u : 0u;
	}
	if (block == 0 && t < (uint32_t)kShards) {
		// commit counters of the rounds that follow (the tail reads them: statistics, and how
		// much it will have to gather); cleared here, by the kernel before them in the stream
		ctl->committed[0][t].n = 0;
		ctl->committed[1][t].n = 0;
		ctl->committed[2][t].n = 0;
		ctl->committed[3][t].n = 0;
	}
	const uint32_t end = hi_idx < total ? hi_idx : total;
	// requests an earlier chunk pass of this call left unresolved ride along: they have
	// lower indices, so they win every reservation they share with this chunk's requests
	const uint32_t carry = ctl->carry_lo;
	const uint32_t start = carry < lo_idx ? carry : lo_idx;
	// whole waves iterate together: the reservation de-duplication is a wave operation
	const uint32_t span = nblocks * blockDim.x;
	const uint32_t first = start + block * blockDim.x;          // this block's first index
	for (uint32_t base = first; base < end && base >= start; base += span) {
		const uint32_t idx = base + t;
		bool live = idx < end && idx >= start;
		bool fresh = false;                         // a request of this pass's own range (not carried over)
		uint32_t P = 0, A = 0;
		Req rq;
		rq.sig = rq.hash = rq.loc = rq.aux = 0u;
		if (live && idx < lo_idx) {
			// carried range: unfinished tags ride along (reset to PENDING); commits of the
			// earlier pass are closed here, so that a later roll-back never takes them for
			// commits of its own pass (it starts at the carry index, like this loop)
			const uint8_t st = state[idx];
			live = st == ST_PENDING || st == ST_LOSER || st == ST_TAIL || (st >= ST_CHAIN_OK && st != ST_SHADOW);
			if (live ? st != ST_PENDING : st != ST_DONE)
				state[idx] = live ? ST_PENDING : ST_DONE;
			if (live) {
				const uint4 rc = reqc[idx];
				rq = req_of(rc);
				P = rc.y & g.hash_mask;
				A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
			}
		} else if (live) {
			const uint32_t *p;
			if (src.single) {
				p = src.single + 3ull * idx;
			} else {
				uint32_t lo = 0, hi = nblk;             // block b: prefix[b] <= idx < prefix[b+1]
				while (hi - lo > 1) {
					const uint32_t mid = (lo + hi) >> 1;
					if (s_prefix[mid] <= idx)
						lo = mid;
					else
						hi = mid;
				}
				p = s_ptr[lo] + 3ull * (idx - s_prefix[lo]);
			}
			Req r;
			r.sig = p[0];
			r.hash = p[1];
			r.loc = p[2];
			r.aux = 0u;
			rq = r;
			if (OP != OP_DELETE && r.sig == 0u && r.loc == 0u) {
				reqc[idx] = make_uint4(r.sig, r.hash, r.loc, 0u);
				state[idx] = ST_DONE;               // skipped by the reference too (:259-262)
				live = false;
			} else {
				state[idx] = ST_PENDING;
				P = r.hash & g.hash_mask;
				A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
				fresh = true;
			}
		}
		// the primary bucket's signature row as the table stands (a hint: deletes of the same cycle
		// may still be running): room enough there -> no registration on the alternate bucket
		uint4 plo = make_uint4(1u, 1u, 1u, 1u), phi = plo;
		bool lazy = false;
		if (OP != OP_DELETE && live && table && (fresh || OP == OP_CUCKOO)) {
			load_sig_row<false>(table, P, plo, phi);
			lazy = fresh && lazy_free != 0u && A != P && (uint32_t)__popc(eq_mask(plo, phi, 0u)) >= lazy_free;
		}
		if (fresh)
			reqc[idx] = make_uint4(rq.sig, rq.hash, rq.loc, lazy ? kLazyBit : 0u);
		const unsigned long long key = owner_key(stamp, idx);
		// (every registration notes the bucket's second-lowest registrant: note_second)
		if (use_fold) {
			mark_fold_min_dedup<kMarkFoldLog2>(fold, s_tk2, owner, second, P, key, live);
			mark_fold_min_dedup<kMarkFoldLog2>(fold, s_tk2, owner, second, A, key, live && A != P && !lazy);
			mark_fold_flush<kMarkFoldLog2>(fold, s_tk2, owner, second);
		} else {
			register_dedup(owner, second, P, key, live);
			register_dedup(owner, second, A, key, live && A != P && !lazy);
		}
		if (OP == OP_CUCKOO && live && table && !lazy)
			prewalk_register(table, g, rq, P, A, owner, key, plo, phi);
	}
}

// Optimistic eviction chain of a hard request that owns P and A (grid-wide rounds).
// Every data-dependent bucket is claimed in the round-1 owner table first: all
// requests of the call registered their P and A there in K1, so an entry stamped
// with this call's round-1 stamp means "somebody of this batch may touch this
// bucket" (or another chain claimed it) -> conflict: the chain restores what it
// wrote and reports failure, and the caller falls back to the region barrier.
// Without conflict the chain's touch set is disjoint from every other request's,
// so its position in the order is immaterial.  Everything is logged for the
// roll-back behind a barrier.  Follows ref gpu_hash.cu:357-423.
__device__ bool try_chain(uint32_t *table, const DevGeom &g, const Req &r, uint32_t A, uint32_t idx,
		unsigned long long *owner1, unsigned long long stamp1, Ctl *ctl, ChainLog *logs, uint32_t *log_slot,
		bool own_ok, const unsigned long long *owner2)
{
	// (owner2: in round 2, that round's own table -- a bucket one of ITS requests is registered on is
	// not free either: a request that did not register on its alternate bucket in the mark pass, Tuning::lazy_alt,
	// does so there)
	if (g.max_cuckoo + 1u > (uint32_t)kChainLogWords / 2u)
		return false;
	const uint32_t slot = atomicAdd(&ctl->chain_n, 1u);
	if (slot >= kChainLogCap)
		return false;
	ChainLog *L = &logs[slot];
	const unsigned long long mykey = owner_key(stamp1, idx);
	const unsigned long long stamp_field = kStampMax - stamp1;
	uint32_t n = 0;
	uint32_t sig = r.sig, loc = r.loc, b = A;
	const uint32_t v = r.sig & 7u;              // victim slot: ORIGINAL signature (:360)
	bool ok = true;
	for (uint32_t k = 0;; k++) {
		const uint64_t ws = sig_word(b, v), wl = loc_word(b, v);
		const uint32_t vsig = table[ws], vloc = table[wl];
		L->word[n] = (uint32_t)ws;
		L->old[n++] = vsig;
		L->word[n] = (uint32_t)wl;
		L->old[n++] = vloc;
		table[ws] = sig;
		table[wl] = loc;
		if (k >= g.max_cuckoo)
			break;                                  // victim dropped (:414-423)
		sig = vsig;
		loc = vloc;
		b = alt_bucket(r.hash, sig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (:334)
		// An entry of this pass that is not this request's own: someone of this batch may touch b.
		// Its own (left by the mark pass's walk, prewalk_register) makes b this request's bucket
		// in ROUND 1, where everybody's right to a bucket is read from this table; in round 2 the
		// losers' rights are in the other table -- one of them may own b there -- so only a bucket
		// nobody of the pass has entered will do.
		// The round-1 table also names, per bucket, the one request that can have committed there in
		// this round (its lowest registrant) -- the tail finds order violations by it (later_commit_on)
		// -- so a foreign entry of this pass is left as it is: the claim is a compare-and-swap on what
		// was read, and only over an entry of an earlier pass (or the request's own, from its pre-walk).
		bool mine = false;
		if (owner2 && (ld_agent(&owner2[b]) >> kIdxBits) == stamp_field - 1ull) {
			ok = false;                             // (stamp of round 2 = round 1's + 1: its field is one lower)
			break;
		}
		for (int attempt = 0; attempt < 4 && !mine; attempt++) {
			const unsigned long long cur = ld_agent(&owner1[b]);
			if ((cur >> kIdxBits) == stamp_field) {
				mine = cur == mykey && own_ok;
				break;
			}
			mine = atomicCAS(&owner1[b], cur, mykey) == cur;
		}
		if (!mine) {
			ok = false;
			break;
		}
		uint4 lo, hi;
		load_sig_row<false>(table, b, lo, hi);
		const uint32_t m = eq_mask(lo, hi, sig);
		if (m) {
			const uint64_t w = loc_word(b, (uint32_t)__ffs((int)m) - 1u);
			L->word[n] = (uint32_t)w;
			L->old[n++] = table[w];
			table[w] = loc;
			break;
		}
		const uint32_t e = eq_mask(lo, hi, 0u);
		if (e) {
			const uint32_t s = preferred_slot(e, r.sig);
			const uint64_t ws2 = sig_word(b, s), wl2 = loc_word(b, s);
			L->word[n] = (uint32_t)ws2;
			L->old[n++] = 0u;
			L->word[n] = (uint32_t)wl2;
			L->old[n++] = table[wl2];
			table[ws2] = sig;
			table[wl2] = loc;
			break;
		}
	}
	if (!ok) {
		for (uint32_t i = n; i-- > 0;)
			table[L->word[i]] = L->old[i];
		L->n = 0;
		return false;
	}
	L->n = n;
	*log_slot = slot;
	return true;
}

template <Op OP>
__global__ __launch_bounds__(256) void k_mark(Ctl *ctl, ReqSrc src, DevGeom g,
		unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state, uint32_t req_cap, uint32_t lo_idx,
		uint32_t hi_idx, const uint32_t *table, int lazy_on)
{
	mark_body<OP>(blockIdx.x, gridDim.x, ctl, src, g, owner, second, reqc, state, req_cap, lo_idx, hi_idx, table, (uint32_t)lazy_on);
}

// mkv_cycle: the mark pass never touches the table, so it rides in the search
// launch as extra work-groups (horizontal fusion): its latency-bound atomics hide
// under the bandwidth-bound probes and one kernel boundary disappears.
template <Op OP>
__global__ __launch_bounds__(256) void k_search_mark(const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n, uint32_t search_blocks,
		Ctl *ctl, ReqSrc src, DevGeom g, unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state,
		uint32_t req_cap, uint32_t hi_idx, int lazy_on)
{
	// mark blocks first: they are few and latency-bound, let them start early
	const uint32_t mark_blocks = gridDim.x - search_blocks;
	if (blockIdx.x < mark_blocks)
		mark_body<OP>(blockIdx.x, mark_blocks, ctl, src, g, owner, second, reqc, state, req_cap, 0u, hi_idx,
				reinterpret_cast<const uint32_t *>(table), (uint32_t)lazy_on);
	else
		search_body<4, true, false>(blockIdx.x - mark_blocks, in, out, table, n, g.hash_mask, g.blk_mask);
}

// Sharded index: search of the received slots + mark pass of the inserts they carry,
// one launch (same horizontal fusion as k_search_mark).
template <Op OP>
__global__ __launch_bounds__(256) void k_search_slots_mark(const uint32_t *__restrict__ recvbuf,
		int64_t slot_words, uint32_t cap_get, uint32_t *__restrict__ out, const uint4 *__restrict__ table,
		uint32_t chunks_per_slot, uint32_t search_blocks, Ctl *ctl, ReqSrc src, DevGeom g,
		unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state, uint32_t req_cap, uint32_t hi_idx,
		uint32_t self, uint32_t *__restrict__ self_out, int lazy_on)
{
	const uint32_t mark_blocks = gridDim.x - search_blocks;
	if (blockIdx.x < mark_blocks)
		mark_body<OP>(blockIdx.x, mark_blocks, ctl, src, g, owner, second, reqc, state, req_cap, 0u, hi_idx,
				reinterpret_cast<const uint32_t *>(table), (uint32_t)lazy_on);
	else
		search_slots_body(blockIdx.x - mark_blocks, recvbuf, slot_words, cap_get, out, table, g.hash_mask,
				g.blk_mask, chunks_per_slot, self, self_out);
}

__device__ __forceinline__ bool tag_logged(uint32_t st)
{
	return st == ST_ROUND1 || st == ST_ROUND2 || st == ST_CHAIN1 || st == ST_CHAIN2;
}

__device__ __forceinline__ bool tag_pending(uint32_t st)
{
	return st == ST_TAIL || st == ST_LOSER || st == ST_PENDING || (st >= ST_CHAIN_OK && st != ST_SHADOW);
}

// a shadow whose group commit no longer stands (taken back, or never made): pending again
__device__ __forceinline__ bool shadow_live(const uint8_t *state, const Undo *undo, const uint4 *reqc, uint32_t idx)
{
	const uint32_t c = undo[idx].word0;
	return !(tag_logged(state[c]) && (reqc[c].w & kLeadBit));
}


// did the grid-wide rounds commit request c, and did that commit use bucket b?  (The primary
// bucket counts as used even when full: it was examined.)
// *eff: the index the commit counts as -- its own, or for the one commit of a group of duplicates
// (kLeadBit) the index of the group's last member, whose write it stands for.
// (st, sh: the candidate's tag and cache entry, fetched by the caller -- who asks about up to three
// candidates of a bucket fetches them all at once instead of one after the other's verdict)
__device__ __forceinline__ bool round_commit_used_pre(uint32_t st, const uint4 &sh, const Undo *undo,
		const ChainLog *chain_logs, const DevGeom &g, uint32_t c, uint32_t b, uint32_t *eff)
{
	if (!tag_logged(st))
		return false;                       // pending, done by the tail, or taken back (or selected already)
	*eff = (sh.w & kLeadBit) ? (sh.w >> 8) : c;
	const uint32_t P = sh.y & g.hash_mask;
	if (b == P)
		return true;
	const bool chain = st == ST_CHAIN1 || st == ST_CHAIN2;
	const uint32_t w0 = undo[c].word0;
	if (b == alt_bucket(sh.y, sh.x, g.hash_mask, g.blk_mask))
		return chain || (w0 != kNoWord && (w0 >> 4) != P);      // (kBucketWords = 16)
	if (!chain)
		return false;
	const ChainLog *L = &chain_logs[w0];
	const uint32_t n = L->n < (uint32_t)kChainLogWords ? L->n : (uint32_t)kChainLogWords;
	for (uint32_t i = 0; i < n; i++)
		if ((L->word[i] >> 4) == b)
			return true;
	return false;
}
__device__ __forceinline__ bool round_commit_used(const uint8_t *state, const uint4 *reqc, const Undo *undo,
		const ChainLog *chain_logs, const DevGeom &g, uint32_t c, uint32_t b, uint32_t *eff)
{
	const uint32_t st = state[c];
	if (!tag_logged(st))
		return false;
	return round_commit_used_pre(st, reqc[c], undo, chain_logs, g, c, b, eff);
}

// ---- K2 / K3: rounds 1 and 2 ------------------------------------------------------
// ROUND 1 iterates every request; ROUND 2 the round-1 losers.  A request that is the lowest
// registrant of every bucket it needs executes with plain loads and stores and logs what it
// overwrote.  A hard request (P and A full, no duplicate) runs its eviction chain if every
// further bucket is free of registrants and other chains (try_chain), otherwise it is left
// to the tail -- still registered, so that nothing later passes it on its own buckets.
// No request waits for a request it shares no bucket with: an order violation between a
// pending chain and something committed here is found when that chain walks (tail).
// A request the grid-wide rounds leave to the tail is noted by its work-group (LDS), which writes its
// entries and their count into its own place in the control block when it is through.  The usual
// pass leaves a handful: the tail then starts on them at once instead of scanning the pass's tags
// for them with all its work-groups (5 - 7 us of a launch that takes 20).
__device__ __forceinline__ void tail_list_add(uint32_t *s_tl, uint32_t *s_tln, uint32_t idx, bool add)
{
	if (add) {
		const uint32_t at = atomicAdd(s_tln, 1u);
		if (at < kTlPer)
			s_tl[at] = idx;
	}
}

template <int ROUND>
__device__ __forceinline__ void tail_list_flush(Ctl *ctl, const uint32_t *s_tl, const uint32_t *s_tln, uint32_t block,
		uint32_t nblocks)
{
	__syncthreads();
	if (nblocks <= kTlGroups) {
		if (block == 0 && threadIdx.x == 0)
			ctl->tl_groups[ROUND - 1] = nblocks;
		const uint32_t n = *s_tln;
		if (threadIdx.x < kTlPer && threadIdx.x < n)
			ctl->tl[ROUND - 1][block][threadIdx.x] = s_tl[threadIdx.x];
		if (threadIdx.x == 0)
			ctl->tl_cnt[ROUND - 1][block] = n;
		return;
	}
	// more work-groups than places (round 2 of a large pass runs on up to 2048): several share a place and
	// take their part of it with one atomic each -- only those that leave anything (the places' counts were
	// cleared by round 1's first work-group)
	if (block == 0 && threadIdx.x == 0)
		ctl->tl_groups[ROUND - 1] = kTlGroups;
	const uint32_t n = *s_tln;
	if (n != 0u && threadIdx.x == 0) {
		const uint32_t slot = block & (kTlGroups - 1u);
		const uint32_t at = atomicAdd(&ctl->tl_cnt[ROUND - 1][slot], n);
		for (uint32_t k = 0; k < n && k < kTlPer && at + k < kTlPer; k++)
			ctl->tl[ROUND - 1][slot][at + k] = s_tl[k];
	}
}

// LDS of a round's work-group: the queue of held-back requests (QCAP entries; a turn adds at most one per
// lane), round 1's two fold tables, the list for the tail.  Carved from a buffer the kernel declares (the
// tail launch, which runs round 2 as its first phase, lays it over its work list).
template <int ROUND, int QCAP>
struct RoundLds {
	static constexpr uint32_t kRFold = ROUND == 1 ? (1u << kMarkFoldLog2) : 1u;
	static constexpr size_t q = 0;                                                  // uint4[QCAP]
	static constexpr size_t ftk = q + sizeof(uint4) * QCAP;                         // unsigned long long[2 * kRFold]
	static constexpr size_t qg = ftk + sizeof(unsigned long long) * 2u * kRFold;    // uint32_t[QCAP]
	static constexpr size_t ftb = qg + sizeof(uint32_t) * QCAP;                     // uint32_t[2 * kRFold]
	static constexpr size_t tl = ftb + sizeof(uint32_t) * 2u * kRFold;              // uint32_t[kTlPer]
	static constexpr size_t words = tl + sizeof(uint32_t) * kTlPer;                 // s_qn, s_fused, s_tln
	static constexpr size_t bytes = words + 4u * sizeof(uint32_t);
};

// (block of nblocks: the work-group's place among those that share the round -- the launch's own for
// rounds launched by themselves)
template <Op OP, int ROUND, int QCAP>
__device__ __forceinline__ void round_body(const uint32_t block, const uint32_t nblocks, unsigned char *lds, Ctl *ctl,
		const DevGeom &g, uint32_t *table,
		const unsigned long long *owner, unsigned long long *owner_next, const uint4 *reqc, uint8_t *state,
		Undo *undo, ChainLog *chain_logs, uint32_t lo_idx, uint32_t hi_idx, uint32_t req_cap, int record_hint,
		unsigned long long *emax, const unsigned long long *second)
{
	// (second: the mark pass's second-lowest registrant per bucket, note_second)
	// (emax: a word per bucket -- the followers of a bucket's round-1 owner elect the last of them there,
	// kFollowBit; the tail's elections use the same words under its own, later stamps)
	// The first turn's request and tag are fetched before anything is known from the control block
	// (one round trip less on the critical path of a launch that is all latency); used if the
	// work-group's first index turns out to be what was guessed (no carried requests).
	const uint32_t pre_base = lo_idx + block * blockDim.x;
	const uint32_t pre_idx = pre_base + threadIdx.x;
	const bool pre_ok = pre_idx < req_cap && pre_idx >= lo_idx;
	const uint8_t pre_st = pre_ok ? state[pre_idx] : (uint8_t)ST_DONE;
	const uint4 pre_rc = pre_ok ? reqc[pre_idx] : make_uint4(0u, 0u, 0u, 0u);
	const bool chains_here = (record_hint & 4) == 0;      // tests: every hard request left to the tail
	const bool groups_on = (record_hint & 8) == 0;        // Tuning::dup_groups = 0: duplicates of a key one per round, as before
	const bool help_on = (record_hint & 32) == 0;         // Tuning::self_help = 0: the second registrant waits for round 2, as before
	if (ROUND == 2 && ctl->any_loser == 0u)
		return;
	if (ROUND == 1 && block == 0) {
		if (threadIdx.x == 0)
			reset_tail_scratch(ctl);
		for (uint32_t q = threadIdx.x; q < kTlGroups; q += blockDim.x)
			ctl->tl_cnt[1][q] = 0u;                 // (round 2's places: its work-groups may have to share them, tail_list_flush)
	}
	const uint32_t count = hi_idx < ctl->total ? hi_idx : ctl->total;
	const unsigned long long stamp1 = ctl->stamp_base;
	const unsigned long long stamp = stamp1 + (ROUND - 1);
	const uint32_t span = nblocks * blockDim.x;
	uint32_t ndone = 0, ndone_help = 0, nlosers = 0;
	// owner table holding the round-1 registrations of every request of this pass (chains claim there)
	unsigned long long *owner_r1 = ROUND == 1 ? const_cast<unsigned long long *>(owner) : owner_next;
	const uint32_t carry = ctl->carry_lo;
	const uint32_t start = carry < lo_idx ? carry : lo_idx;
	// Two phases per work-group.  Phase 1, every request: the reservation of P and the primary
	// bucket; a request P can take is finished there (19 of 20 at load factor 0.5, nearly all below
	// 0.3).  One that needs its alternate bucket -- three more dependent round trips, an eviction
	// chain many more -- is put on a queue in LDS instead of holding up the 63 other lanes of its
	// wave for that long in every turn of the loop; phase 2 runs the queue on adjacent lanes, once
	// per work-group (or whenever the queue could overflow).  Nothing else changes: a queued
	// request owns P all along, and nobody may touch A unless he owns it.
	typedef RoundLds<ROUND, QCAP> Lds;
	uint4 *s_q = reinterpret_cast<uint4 *>(lds + Lds::q);
	uint32_t *s_qg = reinterpret_cast<uint32_t *>(lds + Lds::qg);   // round 2: the last member of the group a queued request leads (kNone: none)
	uint32_t *s_words = reinterpret_cast<uint32_t *>(lds + Lds::words);
	uint32_t &s_qn = s_words[0];
	// Round 1, batches with many duplicates of a few keys (SETs drawn from a Zipf law: 36 000 of one key
	// in 2^19): the losers' registrations for round 2 and the followers' election all aim at the words of
	// those keys' buckets -- 226 us of queueing on a handful of addresses.  Waves with many losers
	// send theirs through a hash table in LDS, flushed once per turn: one atomic per bucket and
	// work-group (fold_min_dedup, as the mark pass does).
	constexpr uint32_t kRFold = Lds::kRFold;
	uint32_t *s_ftb = reinterpret_cast<uint32_t *>(lds + Lds::ftb);
	unsigned long long *s_ftk = reinterpret_cast<unsigned long long *>(lds + Lds::ftk);
	uint32_t &s_fused = s_words[1];
	Fold fmin, fmax;
	fmin.tb = s_ftb;
	fmin.tc = nullptr;
	fmin.tk = s_ftk;
	fmax.tb = s_ftb + kRFold;
	fmax.tc = nullptr;
	fmax.tk = s_ftk + kRFold;
	uint32_t *s_tl = reinterpret_cast<uint32_t *>(lds + Lds::tl);   // what this work-group leaves to the tail (tail_list_add)
	uint32_t &s_tln = s_words[2];
	if (threadIdx.x == 0) {
		s_qn = 0u;
		s_fused = 0u;
		s_tln = 0u;
	}
	if (ROUND == 1) {
		for (uint32_t q = threadIdx.x; q < 2u * kRFold; q += blockDim.x) {
			s_ftb[q] = kNone;
			s_ftk[q] = q < kRFold ? ~0ull : 0ull;
		}
	}
	__syncthreads();
	uint4 *reqw = const_cast<uint4 *>(reqc);    // (the flag word of a request's cache entry is written here)
	// whole waves go round together (the re-registration folds the atomics of lanes on one
	// bucket): the loop variable is the work-group's first index
	uint32_t base = start + block * blockDim.x;
	for (bool more = base < count && base >= start; ; ) {
		if (more) {
			const uint32_t idx = base + threadIdx.x;
			const bool in_range = idx < count;
			// first round trip: request and its state -- independent loads
			const bool pre = base == pre_base && pre_ok;
			const uint8_t st = !in_range ? (uint8_t)ST_DONE : pre ? pre_st : state[idx];
			const bool pending = st == (ROUND == 1 ? ST_PENDING : ST_LOSER);
			const uint4 rc = pre ? pre_rc : (in_range && (ROUND == 1 || pending)) ? reqc[idx] : make_uint4(0u, 0u, 0u, 0u);
			const Req r = req_of(rc);
			const uint32_t P = r.hash & g.hash_mask;
			const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
			const unsigned long long key = owner_key(stamp, idx);
			// second round trip: reservation of P and the primary bucket -- independent.  The
			// reservation of A is only read if P cannot take the request: a request needs
			// exclusive use of exactly the buckets it touches.
			bool lose = false, follower = false, shadow = false;
			uint32_t lead_of = kNone;                   // round 2: the request this one is a shadow of
			if (pending) {
				const unsigned long long oP = owner[P];
				const Bkt kP = load_bucket(table, P);
				if (oP == key) {
					// Round 2, a follower at the head of its bucket: it runs with the location of the LAST
					// follower -- the one write that remains of all of theirs -- and is the group's commit.
					Req ra = r;
					uint32_t grp = kNone;
					if (ROUND == 2 && OP != OP_DELETE && (rc.w & kFollowBit)) {
						const unsigned long long ev = emax[P];
						const uint32_t kn = (uint32_t)(ev & ((1ull << kIdxBits) - 1ull));
						if ((ev >> kIdxBits) == stamp1 && kn >= idx) {
							const uint4 rk = kn > idx ? reqc[kn] : rc;
							if (rk.x == r.sig && ((rk.y ^ r.hash) & g.hash_mask) == 0u) {
								ra.loc = rk.z;
								grp = kn;
							}
						}
						if (grp == kNone)
							ctl->loose = ctl->any_tail = 1u;     // (cannot happen: it entered the election itself; its
							                        // shadows would be picked up again by the tail's gather)
					}
					Undo u;
					u.word0 = kNoWord;
					u.word1 = kNoWord;
					u.old0 = 0u;
					u.old1 = 0u;
					if (apply_primary<OP, OP != OP_DELETE>(table, ra, P, kP, &u)) {
						state[idx] = ROUND == 1 ? ST_ROUND1 : ST_ROUND2;
						if (OP != OP_DELETE)
							undo[idx] = u;
						if (grp != kNone)
							reqw[idx].w = (rc.w & 0xffu) | kLeadBit | (grp << 8);     // (in P: a duplicate's or a new entry's location)
						ndone++;
					} else {
						const uint32_t at = atomicAdd(&s_qn, 1u);
						s_q[at] = make_uint4(r.sig, r.hash, ra.loc, idx | ((rc.w & kLazyBit) << 31));
						s_qg[at] = grp;
					}
				} else {
					lose = true;
					// the bucket's lowest registrant holds the same key: round 1, this request is a follower
					// of it; round 2, a shadow of it if that one is a follower too (it will write for both)
					const uint32_t c = (uint32_t)(oP & ((1ull << kIdxBits) - 1ull));
					const bool look = OP != OP_DELETE && r.sig != 0u && c < idx;
					if (look && ((ROUND == 1 && (groups_on || help_on)) || (groups_on && (rc.w & kFollowBit)))) {
						const uint4 rj = reqc[c];
						const bool same = rj.x == r.sig && ((rj.y ^ r.hash) & g.hash_mask) == 0u;   // (the table's notion of a key: signature and bucket)
						follower = ROUND == 1 && groups_on && same;
						shadow = ROUND == 2 && same && (rj.w & kFollowBit);
						lead_of = c;
						// SELF-HELP (round 1).  This request is the bucket's SECOND-lowest registrant (second[P]: nobody
						// of the pass stands between the owner c and it, on any account: note_second), P is the primary
						// bucket of both, and their signatures differ.  Then what c does here is known: it is the lowest
						// registrant of its primary bucket, so it runs apply_primary on P in this very round -- a duplicate
						// of its signature gets its location, else the preferred empty slot gets the entry -- whatever
						// this request stores meanwhile (other words; and taking one other empty slot away does not
						// change which one c prefers).  So this request commits NOW, on P as c will have left it, instead
						// of losing to c and coming again in round 2: the signature row it has loaded holds c's new entry
						// already or not yet, which it can tell (c's signature is in the row or it is not).  It is a
						// round-1 commit like c's (the two write different words: either may be taken back first) and
						// does not stand in the round-2 table: the THIRD registrant of P, a loser of this round, is the
						// lowest one there and commits in round 2 on P as these two left it.  Who asks for commits on P
						// finds this one through second[P] (candidates_of: the bucket's lowest registrant of round 1, of
						// round 2, and this).  Not when c finds P full (it may go on to evict from P), not for signature
						// 0 (it "matches" empty slots: the two would share words).
						// (second[P] is asked for here, with the owner's request, not beside the reservation: fetched
						// by every lane of a 52 K pass it cost 1.2 us more than the losers' extra round trip.)
						if (ROUND == 1 && help_on && !same && rj.x != 0u && rj.x != r.sig && (rj.y & g.hash_mask) == P &&
								second[P] == key) {
							uint4 s0 = kP.s0, s1 = kP.s1;
							bool ok = true;
							if (!eq_mask(s0, s1, rj.x)) {
								const uint32_t e = eq_mask(s0, s1, 0u);
								if (e) {
									// (c's new entry, not in the row yet: its slot is taken)
									const uint32_t sc = preferred_slot(e, rj.x);
									s0.x = sc == 0u ? rj.x : s0.x;
									s0.y = sc == 1u ? rj.x : s0.y;
									s0.z = sc == 2u ? rj.x : s0.z;
									s0.w = sc == 3u ? rj.x : s0.w;
									s1.x = sc == 4u ? rj.x : s1.x;
									s1.y = sc == 5u ? rj.x : s1.y;
									s1.z = sc == 6u ? rj.x : s1.z;
									s1.w = sc == 7u ? rj.x : s1.w;
								} else {
									ok = false;
								}
							}
							if (ok) {
								Undo u;
								u.word1 = kNoWord;
								u.old1 = 0u;
								const uint32_t m = eq_mask(s0, s1, r.sig);
								const uint32_t e = eq_mask(s0, s1, 0u);
								if (m) {
									const uint32_t sl = (uint32_t)__ffs((int)m) - 1u;
									u.word0 = (uint32_t)loc_word(P, sl);
									u.old0 = loc_at(kP, sl);
									table[u.word0] = r.loc;
								} else if (e) {
									const uint32_t sl = preferred_slot(e, r.sig);
									u.word0 = (uint32_t)sig_word(P, sl);
									u.old0 = 0u;
									u.word1 = (uint32_t)loc_word(P, sl);
									u.old1 = loc_at(kP, sl);
									table[u.word0] = r.sig;
									table[u.word1] = r.loc;
								} else {
									ok = false;
								}
								if (ok) {
									undo[idx] = u;
									state[idx] = ST_ROUND1;
									ndone_help++;
									lose = false;
									follower = false;
								}
							}
						}
					}
				}
			}
			if (ROUND == 1) {
				// Losers register for round 2.  (Every lane comes by here.  Many in one wave mean
				// duplicates of a hot key: their atomics on one bucket are folded; the usual two or
				// three go one atomic each.)
				const bool heavy = __popcll(__ballot(lose)) >= 8;
				if (heavy) {
					fold_min_dedup<kMarkFoldLog2>(fmin, owner_next, P, owner_key(stamp + 1, idx), lose);
					fold_min_dedup<kMarkFoldLog2>(fmin, owner_next, A, owner_key(stamp + 1, idx), lose && A != P);
					if ((threadIdx.x & 63u) == 0u)
						s_fused = 1u;
				} else if (lose) {
					reserve(owner_next, P, A, owner_key(stamp + 1, idx));
				}
				// ... and the followers elect the last of them
				if (OP != OP_DELETE && __ballot(follower)) {
					if (heavy)
						fold_max_dedup<kMarkFoldLog2>(fmax, emax, P, (stamp1 << kIdxBits) | idx, follower);
					else
						atomic_max_dedup(emax, P, (stamp1 << kIdxBits) | idx, follower);
					if (follower)
						reqw[idx].w = rc.w | kFollowBit;
				}
			}
			if (lose) {
				if (shadow) {
					state[idx] = ST_SHADOW;
					undo[idx].word0 = lead_of;      // (not counted as finished by this round: it may come back)
				} else {
					state[idx] = ROUND == 1 ? ST_LOSER : ST_TAIL;
					nlosers++;
					if (ROUND == 1)
						ctl->any_loser = 1u;
					else
						ctl->any_tail = 1u;
				}
			}
			if (ROUND == 2)
				tail_list_add(s_tl, &s_tln, idx, lose && !shadow);
			base += span;
			more = base < count && base >= start;
		}
		__syncthreads();
		if (ROUND == 1 && s_fused) {
			fold_flush<false, false, kMarkFoldLog2>(fmin, owner_next, nullptr, 0ull);
			fold_flush<true, true, kMarkFoldLog2>(fmax, emax, nullptr, 0ull);
			if (threadIdx.x == 0)
				s_fused = 0u;
			__syncthreads();
		}
		const uint32_t qn = s_qn;
		if (more && qn + blockDim.x <= (uint32_t)QCAP)
			continue;                                   // room for another turn's worth
		// phase 2: the queued requests, one per lane
		for (uint32_t q0 = 0; q0 < qn; q0 += blockDim.x) {
			const uint32_t q = q0 + threadIdx.x;
			const bool live = q < qn;
			const uint4 e = live ? s_q[q] : make_uint4(0u, 0u, 0u, 0u);
			Req r;
			r.sig = e.x;
			r.hash = e.y;
			r.loc = e.z;
			r.aux = 0u;
			const uint32_t idx = e.w & 0x7fffffffu;
			const bool lazy = (e.w >> 31) != 0u;        // not registered on A in the mark pass (Tuning::lazy_alt)
			const uint32_t grp = live ? s_qg[q] : kNone;    // (round 2: r.loc is then the group's last location)
			const uint32_t P = r.hash & g.hash_mask;
			const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
			const unsigned long long key = owner_key(stamp, idx);
			bool done = false, hard = false;
			// Round 2, a request that comes to its alternate bucket unregistered in round 1: a LATER
			// request may have been committed there by round 1 (the one the round-1 table names; or a
			// chain that claimed the bucket there) -- then this one is for the tail, which takes that
			// commit back.  Nobody else can be on A now: the request is its lowest registrant in this
			// round's table, where every pending request of the pass stands with both its buckets.
			bool late = false;
			if (ROUND == 2 && OP != OP_DELETE && live && lazy && A != P) {
				const unsigned long long v1 = owner_r1[A];
				const uint32_t c1 = (v1 >> kIdxBits) == kStampMax - stamp1 ? (uint32_t)(v1 & ((1ull << kIdxBits) - 1ull)) : kNone;
				uint32_t e1 = 0;
				late = c1 != kNone && round_commit_used(state, reqc, undo, chain_logs, g, c1, A, &e1) && e1 > idx;
				// ... or by the bucket's second registrant helping itself there in round 1 (self-help above)
				const unsigned long long v3 = second[A];
				const uint32_t c3 = (v3 >> kIdxBits) == kStampMax - stamp1 ? (uint32_t)(v3 & ((1ull << kIdxBits) - 1ull)) : kNone;
				if (!late && c3 != kNone && c3 != c1 && c3 > idx)
					late = round_commit_used(state, reqc, undo, chain_logs, g, c3, A, &e1) && e1 > idx;
			}
			if (live && !late && (A == P || owner[A] == key)) {
				Undo u;
				u.word0 = kNoWord;
				u.word1 = kNoWord;
				u.old0 = 0u;
				u.old1 = 0u;
				bool chained = false;
				done = apply_alternate<OP, OP != OP_DELETE>(table, r, A, &u);
				if (OP == OP_CUCKOO && !done) {
					// hard: P and A full without duplicate -> eviction chain
					atomicMin(&ctl->first_hard, idx);
					uint32_t log_slot = 0;
					chained = chains_here &&
						try_chain(table, g, r, A, idx, owner_r1, stamp1, ctl, chain_logs, &log_slot, ROUND == 1,
								ROUND == 2 ? owner : nullptr);
					hard = !chained;
					u.word0 = log_slot;
				}
				if (done || chained) {
					state[idx] = chained ? (ROUND == 1 ? ST_CHAIN1 : ST_CHAIN2) : (ROUND == 1 ? ST_ROUND1 : ST_ROUND2);
					if (OP != OP_DELETE)
						undo[idx] = u;
					// (not when all that was written is a signature -- the two-choice overflow, ref :197-209:
					// the group's location has not been stored, its shadows come back as pending requests)
					if (ROUND == 2 && grp != kNone) {
						if (chained || u.word1 != kNoWord || (u.word0 & 15u) >= kSlots)
							reqw[idx].w = (reqc[idx].w & 0xffu) | kLeadBit | (grp << 8);
						else
							ctl->loose = ctl->any_tail = 1u;    // (work for the tail even if nothing else is)
					}
					if (chained)
						atomicAdd(&ctl->chain_ok, 1u);
					ndone++;
					done = true;
				}
			}
			const bool lose = live && !done && ROUND == 1 && !hard;
			const bool to_tail = live && !done && !lose;
			if (ROUND == 1) {
				// the hard requests left to the tail register for round 2 like the losers: they stay the
				// lowest registrant of their buckets, nothing later passes them there in round 2
				const bool reg = lose || (hard && to_tail);
				if (__popcll(__ballot(reg)) >= 8)
					reserve_wave(owner_next, P, A, owner_key(stamp + 1, idx), reg);
				else if (reg)
					reserve(owner_next, P, A, owner_key(stamp + 1, idx));
			}
			if (lose) {
				state[idx] = ST_LOSER;
				nlosers++;
				ctl->any_loser = 1u;
			}
			if (to_tail) {
				state[idx] = ST_TAIL;
				ctl->any_tail = 1u;
				if (ROUND == 2 && grp != kNone)
					ctl->loose = 1u;                    // (its shadows are on their own)
				if (hard) {
					ctl->any_hard = 1u;
					atomicAdd(&ctl->hard_tail, 1u);
				}
			}
			tail_list_add(s_tl, &s_tln, idx, to_tail);
		}
		if (!more)
			break;
		__syncthreads();
		if (threadIdx.x == 0)
			s_qn = 0u;
		__syncthreads();
	}
	count_add(&ctl->committed[ROUND - 1][0].n, ndone);
	if (ROUND == 1 && help_on) {
		count_add(&ctl->committed[0][0].n, ndone_help);
		count_add(&ctl->committed[2][0].n, ndone_help);
	}
	if (ROUND == 1)
		count_add(&ctl->committed[3][0].n, nlosers);
	tail_list_flush<ROUND>(ctl, s_tl, &s_tln, block, nblocks);
}


template <Op OP, int ROUND>
__global__ __launch_bounds__(256) void k_round(Ctl *ctl, DevGeom g, uint32_t *table,
		const unsigned long long *owner, unsigned long long *owner_next, const uint4 *reqc, uint8_t *state,
		Undo *undo, ChainLog *chain_logs, uint32_t lo_idx, uint32_t hi_idx, uint32_t req_cap, int record_hint,
		unsigned long long *emax, const unsigned long long *second)
{
	__shared__ __align__(16) unsigned char lds[RoundLds<ROUND, kDeferCap>::bytes];
	round_body<OP, ROUND, kDeferCap>(blockIdx.x, gridDim.x, lds, ctl, g, table, owner, owner_next, reqc, state, undo,
			chain_logs, lo_idx, hi_idx, req_cap, record_hint, emax, second);
}

// ---- order validation of eviction chains that run late ------------------------------
// The grid-wide rounds commit a request as soon as it is the lowest registrant of the
// buckets it needs -- without asking whether an EARLIER request that is still pending will
// turn hard and walk its chain into one of those buckets.  That is found out when the chain
// walks.  WHO can have been committed on a bucket by the grid-wide rounds is written in their
// owner tables: round 1 commits a request only on buckets whose lowest registrant in the
// round-1 table it is (or that its chain claimed there, for both rounds' chains), round 2 only
// as the lowest registrant in the round-2 table -- at most two candidates per bucket, read with
// two loads; whether a candidate did commit, and on that bucket, its tag, undo record and chain
// log say (commit_used).  A chain of request j that steps on a bucket with such a commit k > j
// has met an order violation -- k should have run after j.  (Round 2 of this build kept a list
// per bucket, swapped in with one returning atomicExch per commit: 37 of round 2's 55 us at load
// factor 0.5, and every pass of a 50/50 batch paid for it as soon as one chain ran late.)
// Tail commits are never logged or undone: the tail only commits a request when nothing pending
// below it in its region can still run a chain; its rounds keep their reservations in tables of
// their own, so the grid-wide rounds' tables stay as those rounds left them.
struct Tv {
	const unsigned long long *owner1;    // round-1 reservations of this pass (and every chain's claims)
	const unsigned long long *owner2;    // round-2 reservations
	const unsigned long long *second;    // the mark pass's second-lowest registrant per bucket: round 1 may have let it help itself
	unsigned long long stamp1;           // round-1 stamp of this pass (round 2: + 1)
	const uint8_t *state;
	const uint4 *reqc;
	const Undo *undo;
	const ChainLog *chain_logs;          // (behind the kChainLogCap logs of the rounds: one log per lane of the tail launch)
	Ctl *ctl;                            // violations are noted there: viol[] per region (lowest violating index), tflags
	                                     // (bit 0: one was recorded), vlist_n / vlist (this turn's: {index, bucket}); dbg[]
};
constexpr uint32_t kTvViolation = 1u;

// (debug builds, make EXTRA=-DMKV_TRACE: what the tail did, in the order it did it -- tests/debug/fuzz_debug.py prints it)
#ifdef MKV_TRACE
constexpr uint32_t kTraceCap = 1u << 17;
__device__ uint32_t g_trace[4u * kTraceCap + 4u];
__device__ __forceinline__ void trace_ev(uint32_t code, uint32_t a, uint32_t b, uint32_t c)
{
	const uint32_t at = atomicAdd(&g_trace[0], 1u);
	if (at < kTraceCap) {
		uint32_t *e = g_trace + 4u + 4u * at;
		e[0] = code;
		e[1] = a;
		e[2] = b;
		e[3] = c;
	}
}
#define MKV_TR(code, a, b, c) trace_ev(code, a, b, c)
#else
#define MKV_TR(code, a, b, c) ((void)0)
#endif

__device__ __forceinline__ bool commit_used(const Tv &tv, const DevGeom &g, uint32_t c, uint32_t b, uint32_t *eff)
{
	return round_commit_used(tv.state, tv.reqc, tv.undo, tv.chain_logs, g, c, b, eff);
}
// the tags and cache entries of a bucket's (up to three) candidates, asked for together: one round trip
// instead of up to six in a row (a request of the in-order replay spends its time in exactly these)
struct Cands {
	uint32_t st1, st2, st3;
	uint4 sh1, sh2, sh3;
};
__device__ __forceinline__ void load_cands(const Tv &tv, uint32_t c1, uint32_t c2, uint32_t c3, Cands &k)
{
	const uint4 z = make_uint4(0u, 0u, 0u, 0u);
	k.st1 = c1 != kNone ? (uint32_t)tv.state[c1] : (uint32_t)ST_PENDING;
	k.st2 = c2 != kNone ? (uint32_t)tv.state[c2] : (uint32_t)ST_PENDING;
	k.st3 = c3 != kNone ? (uint32_t)tv.state[c3] : (uint32_t)ST_PENDING;
	k.sh1 = c1 != kNone ? tv.reqc[c1] : z;
	k.sh2 = c2 != kNone ? tv.reqc[c2] : z;
	k.sh3 = c3 != kNone ? tv.reqc[c3] : z;
}
__device__ __forceinline__ bool commit_used_pre(const Tv &tv, const DevGeom &g, uint32_t st, const uint4 &sh, uint32_t c,
		uint32_t b, uint32_t *eff)
{
	return round_commit_used_pre(st, sh, tv.undo, tv.chain_logs, g, c, b, eff);
}

// the candidates of bucket b from its reservation words (kNone: none of this pass): its lowest registrant of
// round 1, of round 2, and the mark pass's second-lowest (a self-help commit of round 1)
__device__ __forceinline__ void candidates_of(const Tv &tv, unsigned long long v1, unsigned long long v2,
		unsigned long long v3, uint32_t &c1, uint32_t &c2, uint32_t &c3)
{
	const unsigned long long f1 = kStampMax - tv.stamp1, f2 = kStampMax - (tv.stamp1 + 1ull);
	c1 = (v1 >> kIdxBits) == f1 ? (uint32_t)(v1 & ((1ull << kIdxBits) - 1ull)) : kNone;
	c2 = (v2 >> kIdxBits) == f2 ? (uint32_t)(v2 & ((1ull << kIdxBits) - 1ull)) : kNone;
	c3 = (v3 >> kIdxBits) == f1 ? (uint32_t)(v3 & ((1ull << kIdxBits) - 1ull)) : kNone;
	if (c3 == c1 || c3 == c2)
		c3 = kNone;
}

// a commit later than idx on bucket b (its index), or kNone -- from reservation words the caller
// has loaded beside its other accesses to the bucket
__device__ __forceinline__ uint32_t later_commit_from(const Tv &tv, const DevGeom &g, uint32_t b, unsigned long long v1,
		unsigned long long v2, unsigned long long v3, uint32_t idx)
{
	uint32_t c1, c2, c3, e = 0;
	candidates_of(tv, v1, v2, v3, c1, c2, c3);
	if (c2 == c1)
		c2 = kNone;
	Cands k;
	load_cands(tv, c1, c2, c3, k);
	if (c1 != kNone && commit_used_pre(tv, g, k.st1, k.sh1, c1, b, &e) && e > idx)
		return c1;
	if (c2 != kNone && commit_used_pre(tv, g, k.st2, k.sh2, c2, b, &e) && e > idx)
		return c2;
	if (c3 != kNone && commit_used_pre(tv, g, k.st3, k.sh3, c3, b, &e) && e > idx)
		return c3;
	return kNone;
}

__device__ __forceinline__ uint32_t later_commit_on(const Tv &tv, const DevGeom &g, uint32_t b, uint32_t idx)
{
	return later_commit_from(tv, g, b, ld_agent(&tv.owner1[b]), ld_agent(&tv.owner2[b]), ld_agent(&tv.second[b]), idx);
}

// A commit on bucket b that has to come after request idx AND does not commute with it: any later
// commit (`any`: the request did not register on the bucket, or may go on to replace an entry there),
// or one that carries the request's own signature.  The latter is what a request that stood
// registered on the bucket all along can meet: the commit of a group of duplicates (kLeadBit) led by
// an earlier request, which stands for writes up to the group's last member.  Such a commit only
// ever rewrote the location of ONE entry after its leader's turn; a request that finds its own
// duplicate, or an empty slot, elsewhere in the bucket writes other words -- unless it takes the
// group's entry for its duplicate (same signature from another bucket pair).
__device__ __forceinline__ bool conflict_on(const Tv &tv, const DevGeom &g, uint32_t b, uint32_t idx, uint32_t sig, bool any)
{
	uint32_t c1, c2, c3, e = 0;
	candidates_of(tv, ld_agent(&tv.owner1[b]), ld_agent(&tv.owner2[b]), ld_agent(&tv.second[b]), c1, c2, c3);
	if (c2 == c1)
		c2 = kNone;
	Cands k;
	load_cands(tv, c1, c2, c3, k);
	if (c1 != kNone && commit_used_pre(tv, g, k.st1, k.sh1, c1, b, &e) && e > idx && (any || k.sh1.x == sig))
		return true;
	if (c2 != kNone && commit_used_pre(tv, g, k.st2, k.sh2, c2, b, &e) && e > idx && (any || k.sh2.x == sig))
		return true;
	return c3 != kNone && commit_used_pre(tv, g, k.st3, k.sh3, c3, b, &e) && e > idx && (any || k.sh3.x == sig);
}

// The check a pending request makes ONCE (kCheckedBit), before anything later than it in its region
// can commit in the tail: the bucket with a conflicting commit of the grid-wide rounds (the request
// is then an order violation's seed and the barrier of its region), or kNone.  On P, where every
// request registers in the mark pass, only a group's commit can be later.  On A any later commit
// counts for a request that did not register there (kLazyBit) and, two-choice, for every request:
// it may end up replacing slot sig&7 there (of P, if the two are one bucket), which nobody foresees
// -- found out at commit time, beside the commits of later requests of the same round, it would be
// found out too late (cuckoo: a request that may turn hard is a barrier of the rounds anyway, and
// its chain looks at every bucket it goes on to).
template <Op OP>
__device__ __forceinline__ uint32_t early_conflict(const Tv &tv, const DevGeom &g, uint32_t P, uint32_t A, uint32_t idx,
		uint32_t sig, uint32_t aux)
{
	if (conflict_on(tv, g, P, idx, sig, OP == OP_2CHOICE && A == P))    // (A == P: the slot replaced is one of P's)
		return P;
	if (A != P && conflict_on(tv, g, A, idx, sig, OP == OP_2CHOICE || (aux & kLazyBit) != 0u))
		return A;
	return kNone;
}

__device__ __forceinline__ void report_violation(const Tv &tv, const DevGeom &g, uint32_t P, uint32_t idx,
		uint32_t bucket)
{
	MKV_TR(3u, idx, bucket, P);
	atomicMin(&tv.ctl->viol[P >> g.region_shift], idx);
	const uint32_t at = atomicAdd(&tv.ctl->vlist_n, 1u);
	if (at < (uint32_t)kViolCap)
		tv.ctl->vlist[at] = make_uint2(idx, bucket);
	atomicOr(&tv.ctl->tflags, kTvViolation);
}

// apply_static in the tail, for a request that may not have registered on its alternate bucket in
// the mark pass (kLazyBit): before it touches that bucket, the grid-wide rounds' commits there are
// looked at -- a later one is an order violation like a chain's (reported; the tail takes it back
// and the request comes again).  0 = done, 1 = order violation reported (nothing written),
// 3 = needs an eviction chain (cuckoo; nothing written).
template <Op OP>
__device__ __forceinline__ int apply_static_late(uint32_t *table, const DevGeom &g, const Req &r, uint32_t P, uint32_t A,
		uint32_t idx, const Tv &tv)
{
	if (OP == OP_DELETE) {
		apply_static<OP, true, false>(table, r, P, A, nullptr);
		return 0;
	}
	// (a request that has been through the rounds' early check, kCheckedBit, has looked already)
	const bool unchecked = !(r.aux & kCheckedBit);
	uint4 lo, hi;
	load_sig_row<true>(table, P, lo, hi);       // (on its way while the bucket's candidates are looked at)
	if (unchecked && conflict_on(tv, g, P, idx, r.sig, OP == OP_2CHOICE && A == P)) {
		report_violation(tv, g, P, idx, P);
		return 1;
	}
	if (try_place_rows<true, false>(table, P, lo, hi, r.sig, r.loc, r.sig, nullptr))
		return 0;
	if (unchecked && A != P && conflict_on(tv, g, A, idx, r.sig, OP == OP_2CHOICE || (r.aux & kLazyBit) != 0u)) {
		report_violation(tv, g, P, idx, A);
		return 1;
	}
	if (try_place<true, false>(table, A, r.sig, r.loc, r.sig, nullptr))
		return 0;
	// What follows replaces an entry of A.  The one commit a group of duplicates has (kLeadBit) stands
	// for writes up to its last member: to a request between the two it is a LATER commit on this
	// bucket, and the entry about to go may be the group's.
	if (later_commit_on(tv, g, A, idx) != kNone) {
		report_violation(tv, g, P, idx, A);
		return 1;
	}
	if (OP == OP_2CHOICE) {
		table[sig_word(A, r.sig & 7u)] = r.sig;     // no room: the signature alone replaces slot sig&7 (ref :197-209)
		return 0;
	}
	return 3;
}

// One request run to completion, eviction chain included, at its place in the canonical
// order (nothing else of its region runs meanwhile).  Every bucket the chain goes on to is
// checked for later commits of the grid-wide rounds first (later_commit_on); the chain logs what it overwrites (it may come
// back to a bucket it has used, so a read-only walk would not do) and restores it when it
// has to give up.  0 = done, 1 = order violation (nothing written).
template <Op OP>
__device__ __forceinline__ int apply_serial_checked(uint32_t *table, const DevGeom &g, const Req &r, uint32_t idx, const Tv &tv)
{
	const uint32_t P = r.hash & g.hash_mask;
	const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
	const int rs = apply_static_late<OP>(table, g, r, P, A, idx, tv);
	if (rs != 3)
		return rs;
	// cuckoo, P and A full without duplicate (ref :357-423); A has been looked at for later commits
	// (apply_static_late does before it gives up)
	// (the log lives in this lane's slot of global memory: local arrays would cost the launch its
	// scratch-free dispatch)
	ChainLog *L = const_cast<ChainLog *>(tv.chain_logs) + kChainLogCap + (blockIdx.x * blockDim.x + threadIdx.x);
	uint32_t n = 0;
	uint32_t sig = r.sig, loc = r.loc, b = A;
	const uint32_t v = r.sig & 7u;          // victim slot: ORIGINAL signature (:360)
	for (uint32_t k = 0;; k++) {
		const uint64_t ws = sig_word(b, v), wl = loc_word(b, v);
		const uint32_t vsig = ld_agent32(table + ws);
		const uint32_t vloc = ld_agent32(table + wl);
		L->word[n] = (uint32_t)ws;
		L->old[n++] = vsig;
		L->word[n] = (uint32_t)wl;
		L->old[n++] = vloc;
		table[ws] = sig;
		table[wl] = loc;
		if (k >= g.max_cuckoo)
			return 0;                       // victim dropped (:414-423)
		sig = vsig;
		loc = vloc;
		b = alt_bucket(r.hash, sig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (:334)
		if (later_commit_on(tv, g, b, idx) != kNone) {
			for (uint32_t i = n; i-- > 0;)
				table[L->word[i]] = L->old[i];
			report_violation(tv, g, P, idx, b);
			return 1;
		}
		if (try_place<true, false>(table, b, sig, loc, r.sig, nullptr))
			return 0;
	}
}

// ---- K4: tail -------------------------------------------------------------------
// (an index the compiler takes for new every time: the address built from it is not kept in a register
// pair -- spilled to scratch memory, in this kernel -- across the launch's main loop)
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
	asm volatile("" : "+v"(v));
	return v;
}

__device__ __forceinline__ void wg_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__syncthreads();
}

// All work-groups of the launch meet here.  They are few (Tuning::tail_groups) and stay resident
// while they wait, so the others always get to run.  Release before, acquire after.
__device__ __forceinline__ void grid_sync(uint32_t *counter, uint32_t nwg)
{
	if (nwg == 1u) {
		wg_sync();
		return;
	}
	wg_sync();                     // the group's stores have been issued and are visible to thread 0
	if (threadIdx.x == 0) {
		__threadfence();           // ... which releases them to the other groups (cumulative),
		atomicAdd(counter, 1u);
		while (ld_agent32(counter) < nwg)
			__builtin_amdgcn_s_sleep(2);
		__threadfence();           // acquires theirs, and hands them on through the barrier below
	}
	wg_sync();
}

// The tail launch's scratch in the control block.  Reset by the kernel BEFORE it in the
// stream (round 1) -- never by the tail itself: a helper group may still be polling a
// counter when work-group 0 is already through.
__device__ __forceinline__ void reset_tail_scratch(Ctl *ctl)
{
	ctl->sync[0] = ctl->sync[1] = 0;
	ctl->sync_r2 = 0;
	ctl->g_sync = 0;
	ctl->glist_n = 0;
	ctl->loose = 0;
	ctl->carry_next = kNone;
	ctl->undone_acc = 0;
	ctl->g_progress[0] = ctl->g_progress[1] = 0;
	ctl->g_left[0] = ctl->g_left[1] = 0;
	ctl->g_dupw[0] = ctl->g_dupw[1] = 0;
	ctl->tail_rounds_used = 0;
	ctl->tflags = 0;
	ctl->vlist_n = 0;
	for (int k = 0; k < kMaxRegions; k++) {
		ctl->g_rbar[0][k] = ctl->g_rbar[1][k] = kNone;
		ctl->g_rmin[0][k] = ctl->g_rmin[1][k] = kNone;
		ctl->g_below[0][k] = ctl->g_below[1][k] = 0u;
		ctl->viol[k] = kNone;
	}
}

// like grid_sync, on a counter that only grows: the k-th meeting waits for k * nwg arrivals
__device__ __forceinline__ void grid_sync_nth(uint32_t *counter, uint32_t nwg, uint32_t *nth)
{
	const uint32_t target = ++*nth * nwg;
	wg_sync();
	if (threadIdx.x == 0) {
		__threadfence();
		atomicAdd(counter, 1u);
		while (ld_agent32(counter) < target)
			__builtin_amdgcn_s_sleep(2);
		__threadfence();
	}
	wg_sync();
}

// bitonic sort of n2 (power of two) keys in LDS, ascending
__device__ __forceinline__ void lds_sort(uint32_t *keys, uint32_t n2)
{
	const uint32_t t = threadIdx.x;
	for (uint32_t k = 2; k <= n2; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t i = t; i < n2; i += blockDim.x) {
				const uint32_t ixj = i ^ j;
				if (ixj > i) {
					const uint32_t a = keys[i], b = keys[ixj];
					if ((a > b) == ((i & k) == 0)) {
						keys[i] = b;
						keys[ixj] = a;
					}
				}
			}
			__syncthreads();
		}
	}
}

// Replay of the sorted requests lkeys[0..n) one at a time per closed region, regions side
// by side (one wave each): the region of every request is fetched by all threads at once,
// then a region's wave walks the list with ballots and its first lane replays.  A region
// stops at the first request whose chain meets an order violation: everything from there on
// stays pending.  Returns the number replayed.
template <Op OP>
__device__ __forceinline__ uint32_t replay_regions(uint32_t *table, const DevGeom &g, const uint4 *reqc, uint8_t *state,
		const uint32_t *lkeys, uint8_t *lreg, uint32_t n, const Tv &tv, uint32_t *s_count)
{
	const uint32_t t = threadIdx.x;
	for (uint32_t i = t; i < n; i += blockDim.x) {
		const uint32_t idx = lkeys[i];
		const uint4 rc = reqc[idx];
		const uint32_t P = rc.y & g.hash_mask, A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
		lreg[i] = (uint8_t)(P >> g.region_shift);
		// touch both buckets: the replaying lane then finds them in L2
		const uint32_t a = ld_agent32(table + sig_word(P, 0u)), b = ld_agent32(table + sig_word(A, 0u));
		asm volatile("" ::"v"(a), "v"(b));
	}
	__syncthreads();
	const uint32_t wave = t >> 6, lane = t & 63u, nwaves = blockDim.x >> 6;
	for (uint32_t region = wave; region < g.nregions; region += nwaves) {
		uint32_t d = 0;
		bool stop = false;
		for (uint32_t base = 0; base < n && !stop; base += 64u) {
			const uint32_t i = base + lane;
			unsigned long long todo = __ballot(i < n && lreg[i] == (uint8_t)region);
			while (todo && !stop) {
				const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u;
				todo &= todo - 1ull;
				int rc = 0;
				if (lane == 0) {
					const uint32_t idx = lkeys[base + j];
					rc = apply_serial_checked<OP>(table, g, req_of(reqc[idx]), idx, tv);
					if (rc == 0) {
						state[idx] = ST_DONE;
						MKV_TR(2u, idx, 0u, 0u);
						d++;
					}
				}
				stop = __shfl(rc, 0) != 0;
			}
		}
		if (lane == 0 && d)
			atomicAdd(s_count, d);
	}
	wg_sync();
	return *s_count;
}

// The same for a list too long for LDS (one work-group): the pending requests of
// [lo_idx, total) are first compacted IN INDEX ORDER into the global list (each thread takes
// a contiguous run of tags: count, scan, write), then every region's wave walks that list 64
// entries at a time -- all lanes fetch their entry's hash and touch its buckets (so the
// replaying lane finds them in L2), lane 0 replays its region's, stopping like replay_regions.
template <Op OP>
__device__ __forceinline__ uint32_t replay_pending_scan(uint32_t *table, const DevGeom &g, const uint4 *reqc, uint8_t *state,
		uint32_t *glist, uint32_t *scratch1024, uint32_t lo_idx, uint32_t total, const Tv &tv, uint32_t *s_count)
{
	const uint32_t t = threadIdx.x;
	const uint32_t lo16 = lo_idx & ~15u;
	const uint32_t span = total > lo16 ? total - lo16 : 0u;
	const uint32_t per = (((span + blockDim.x - 1u) / blockDim.x) + 15u) & ~15u;
	const uint32_t first = lo16 + t * per;
	const uint32_t last = first + per < total ? first + per : total;
	const uint4 *st4 = reinterpret_cast<const uint4 *>(state);
	uint32_t c = 0;
	for (uint32_t base = first; base < last && base >= lo16; base += 16u) {
		const uint4 v = st4[base >> 4];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint32_t st = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
			const uint32_t idx = base + (uint32_t)k;
			c += (idx >= lo_idx && idx < last && (tag_pending(st) || (st == ST_SHADOW && shadow_live(state, tv.undo, reqc, idx)))) ? 1u : 0u;
		}
	}
	scratch1024[t] = c;
	__syncthreads();
	for (uint32_t off = 1; off < blockDim.x; off <<= 1) {      // inclusive scan
		const uint32_t add = t >= off ? scratch1024[t - off] : 0u;
		__syncthreads();
		scratch1024[t] += add;
		__syncthreads();
	}
	const uint32_t n = scratch1024[blockDim.x - 1u];
	uint32_t at = scratch1024[t] - c;
	for (uint32_t base = first; base < last && base >= lo16; base += 16u) {
		const uint4 v = st4[base >> 4];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint32_t st = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
			const uint32_t idx = base + (uint32_t)k;
			if (idx >= lo_idx && idx < last && (tag_pending(st) || (st == ST_SHADOW && shadow_live(state, tv.undo, reqc, idx))))
				glist[at++] = idx;
		}
	}
	wg_sync();
	const uint32_t *reqw = reinterpret_cast<const uint32_t *>(reqc);
	const uint32_t wave = t >> 6, lane = t & 63u, nwaves = blockDim.x >> 6;
	for (uint32_t region = wave; region < g.nregions; region += nwaves) {
		uint32_t d = 0;
		bool stop = false;
		for (uint32_t base = 0; base < n && !stop; base += 64u) {
			const uint32_t i = base + lane;
			uint32_t idx = 0;
			bool mine = false;
			if (i < n) {
				idx = glist[i];
				const uint32_t sig = reqw[(size_t)idx * 4u], hash = reqw[(size_t)idx * 4u + 1u];
				const uint32_t P = hash & g.hash_mask;
				mine = (P >> g.region_shift) == region;
				if (mine) {
					// touch both buckets: the replay below then finds them in L2
					const uint32_t A = alt_bucket(hash, sig, g.hash_mask, g.blk_mask);
					const uint32_t a = ld_agent32(table + sig_word(P, 0u)), b = ld_agent32(table + sig_word(A, 0u));
					asm volatile("" ::"v"(a), "v"(b));
				}
			}
			unsigned long long todo = __ballot(mine);
			while (todo && !stop) {
				const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u;
				todo &= todo - 1ull;
				const uint32_t ridx = __shfl(idx, (int)j);
				int rc = 0;
				if (lane == 0) {
					rc = apply_serial_checked<OP>(table, g, req_of(reqc[ridx]), ridx, tv);
					if (rc == 0) {
						state[ridx] = ST_DONE;
						MKV_TR(2u, ridx, 1u, 0u);
						d++;
					}
				}
				stop = __shfl(rc, 0) != 0;
			}
		}
		if (lane == 0 && d)
			atomicAdd(s_count, d);
	}
	wg_sync();
	return n;
}

// Tail rounds, decide phase: read-only walk of the eviction chain of a hard request
// that owns P and A, claiming every further bucket in the round's owner table.  A bucket an
// EARLIER pending request is on stops the walk (that request goes first).  From a LATER one
// the bucket is taken over, and the later request is made a barrier of this round: the
// classification "cannot turn hard" of pending requests counts the free slots and duplicates
// of the buckets they registered, which a chain may only change under a request if neither
// that request nor anything behind it commits in the same round.  True = the walk got
// through, the chain never returns to a bucket it already used (so the walk saw what the real
// run will see), and no later request of this pass was COMMITTED on one of the buckets by the
// grid-wide rounds (later_commit_from; else: order violation, reported) -- the request may run
// its chain in this round's commit phase if it still owns everything then (run_owned_chain:
// an earlier chain may come by after it).
__device__ __forceinline__ bool walk_and_claim(const uint32_t *table, const DevGeom &g, const Req &r, uint32_t P, uint32_t A,
		uint32_t idx, unsigned long long *owner, unsigned long long stamp, const Tv &tv, uint32_t *rbar, uint32_t vsig)
{
	// vsig: the signature in the victim slot of A (the caller holds A's signature row).  Every
	// further bucket costs ONE round trip: its claim, its two reservation words of the grid-wide
	// rounds and its signature row -- which also holds the next victim -- go out together.
	const unsigned long long mykey = owner_key(stamp, idx);
	const unsigned long long stamp_field = kStampMax - stamp;
	const uint32_t v = r.sig & 7u;
	// (the chain begins with an eviction from A: the grid-wide rounds' commits there are looked at
	// first, like apply_static_late does -- a request that did not register on A in the mark pass may
	// find a later request's there, any request the commit of a group of duplicates that reaches
	// beyond it)
	if (later_commit_on(tv, g, A, idx) != kNone) {
		report_violation(tv, g, P, idx, A);
		return false;
	}
	uint32_t seen[8];
	uint32_t nseen = 0;
	for (uint32_t k = 0;; k++) {
		if (k >= g.max_cuckoo)
			return true;                            // last eviction drops its victim: no further bucket
		const uint32_t nb = alt_bucket(r.hash, vsig, g.hash_mask, g.blk_mask);
		if (nb == P || nb == A) {
			atomicAdd(tv.ctl->dbg + 7, 1u);
			return false;
		}
		for (uint32_t i = 0; i < nseen; i++)
			if (seen[i] == nb) {
				atomicAdd(tv.ctl->dbg + 7, 1u);
				return false;
			}
		const unsigned long long old = atomicMin(&owner[nb], mykey);
		const unsigned long long v1 = ld_agent(&tv.owner1[nb]), v2 = ld_agent(&tv.owner2[nb]), v3 = ld_agent(&tv.second[nb]);
		uint4 lo, hi;
		load_sig_row<true>(table, nb, lo, hi);
		if ((old >> kIdxBits) == stamp_field && old != mykey) {
			if (old < mykey) {
				atomicAdd(tv.ctl->dbg + 3, 1u);
				return false;                       // an earlier pending request is on nb: it goes first
			}
			// A LATER pending request was the lowest on nb: the bucket is taken over (the atomicMin
			// has done that), and that request becomes a barrier of this round -- neither it nor
			// anything behind it in the region commits now, whatever it concluded from the bucket's
			// free slots and duplicates; it finds at commit time that it no longer owns the bucket.
			atomicAdd(tv.ctl->dbg + 0, 1u);
			atomicMin(&rbar[P >> g.region_shift], (uint32_t)(old & ((1ull << kIdxBits) - 1ull)));
		}
		// a later request of this pass was committed on nb by the grid-wide rounds: order violation
		if (later_commit_from(tv, g, nb, v1, v2, v3, idx) != kNone) {
			report_violation(tv, g, P, idx, nb);
			return false;
		}
		seen[nseen++] = nb;
		if (eq_mask(lo, hi, vsig) || eq_mask(lo, hi, 0u))
			return true;                            // the victim finds a duplicate or room here
		vsig = sig_at(lo, hi, v);
	}
}

// Commit phase: the chain of a request tagged CHAIN_OK runs while it is checked that the
// request still owns P, A and every bucket of the walk (an earlier chain that came by after
// this request's claim has taken the bucket over) -- one round trip per bucket (its reservation, its signature row and the location of the next
// victim go out together) instead of a read-only walk followed by the run.  A bucket is only
// written once it is known to be the request's own, and such a bucket is written by nobody
// else in this phase; if a later one turns out to be taken over, what was written is put
// back from the lane's log.  The chain never returns to a bucket (walk_and_claim saw to
// that), so each bucket reads as the walk read it.  False: nothing is changed.
__device__ __forceinline__ bool run_owned_chain(uint32_t *table, const DevGeom &g, const Req &r, uint32_t P, uint32_t A,
		uint32_t idx, const unsigned long long *owner, unsigned long long stamp, const Tv &tv)
{
	const unsigned long long mykey = owner_key(stamp, idx);
	const uint32_t v = r.sig & 7u;          // victim slot: ORIGINAL signature (ref :360)
	const unsigned long long oP = ld_agent(&owner[P]);
	const unsigned long long oA = A != P ? ld_agent(&owner[A]) : mykey;
	uint32_t vsig = ld_agent32(table + sig_word(A, v));
	uint32_t vloc = ld_agent32(table + loc_word(A, v));
	if (oP != mykey || oA != mykey)
		return false;
	ChainLog *L = const_cast<ChainLog *>(tv.chain_logs) + kChainLogCap + (blockIdx.x * blockDim.x + threadIdx.x);
	uint32_t n = 0;
	uint32_t sig = r.sig, loc = r.loc, b = A;
	for (uint32_t k = 0;; k++) {
		const uint64_t ws = sig_word(b, v), wl = loc_word(b, v);
		L->word[n] = (uint32_t)ws;
		L->old[n++] = vsig;
		L->word[n] = (uint32_t)wl;
		L->old[n++] = vloc;
		table[ws] = sig;
		table[wl] = loc;
		if (k >= g.max_cuckoo)
			return true;                    // victim dropped (ref :414-423)
		sig = vsig;
		loc = vloc;
		b = alt_bucket(r.hash, sig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (ref :334)
		const unsigned long long ob = ld_agent(&owner[b]);
		uint4 lo, hi;
		load_sig_row<true>(table, b, lo, hi);
		const uint32_t nloc = ld_agent32(table + loc_word(b, v));
		if (ob != mykey) {
			for (uint32_t i = n; i-- > 0;)
				table[L->word[i]] = L->old[i];
			return false;
		}
		const uint32_t m = eq_mask(lo, hi, sig);
		if (m) {
			table[loc_word(b, (uint32_t)__ffs((int)m) - 1u)] = loc;
			return true;
		}
		const uint32_t e = eq_mask(lo, hi, 0u);
		if (e) {
			const uint32_t s2 = preferred_slot(e, r.sig);
			table[sig_word(b, s2)] = sig;
			table[loc_word(b, s2)] = loc;
			return true;
		}
		vsig = sig_at(lo, hi, v);
		vloc = nloc;
	}
}

// Tail rounds, duplicate writers.  A request tagged DUPW holds a duplicate in its primary
// bucket that nothing can evict before its turn: all it will do is overwrite that slot's
// location, which commutes with every structural request -- only the order among the
// writers of one slot matters (the last one wins).  So the writers of a bucket below the
// round's barrier elect the highest index among them (atomicMax on a word per bucket);
// the winner writes, the others with the winner's signature are dead writes and done,
// writers of another slot of that bucket try again next round.  Nobody else may write the
// bucket in that round: the election only counts when the bucket's owner (its lowest
// registrant) is a duplicate writer too -- any other request needs to own what it writes.
// Thousands of SETs of one hot key thus take a round or two instead of one round each.
__device__ __forceinline__ void elect_last_writer(const DevGeom &g, const unsigned long long *owner,
		unsigned long long *emax, const uint4 *reqc, uint8_t *state, uint32_t idx, uint32_t P, uint32_t bar,
		unsigned long long stamp, bool active)
{
	// every lane of the wave calls this; `active` = the lane holds a pending request
	bool take = active && state[idx] == ST_DUPW && idx < bar;
	if (take) {
		const uint32_t o = (uint32_t)(ld_agent(&owner[P]) & ((1ull << kIdxBits) - 1ull));
		const uint8_t so = state[o];
		take = so == ST_DUPW || so == ST_DUPW_OK;
	}
	atomic_max_dedup(emax, P, (stamp << kIdxBits) | idx, take);
	if (take)
		state[idx] = ST_DUPW_OK;
}

// commit of a request tagged DUPW_OK: 1 = done (wrote, or a dead write), 0 = try again
__device__ __forceinline__ int commit_last_writer(uint32_t *table, const DevGeom &g, const unsigned long long *emax,
		const uint4 *reqc, const Req &r, uint32_t P, uint32_t idx)
{
	const uint32_t w = (uint32_t)(ld_agent(&emax[P]) & ((1ull << kIdxBits) - 1ull));
	if (w == idx) {
		uint4 lo, hi;
		load_sig_row<true>(table, P, lo, hi);
		const uint32_t m = eq_mask(lo, hi, r.sig);
		if (!m)
			return 0;                                   // (cannot happen: the duplicate was stable)
		table[loc_word(P, (uint32_t)__ffs((int)m) - 1u)] = r.loc;
		return 1;
	}
	return reqc[w].x == r.sig ? 1 : 0;
}

// The same election inside ONE work-group (in-group rounds), per ENTRY instead of per bucket: a hash
// table in LDS keyed by {bucket, slot of the duplicate}, the value the highest index entered.  Every
// key of a bucket with pending duplicates is settled in one round -- the per-bucket word settles one
// key per round, and a bucket that holds two hot keys of a Zipf batch sends four rounds' worth of
// requests round the loop.  (A request that finds no slot stays tagged DUPW: next round.)
constexpr uint32_t kElectSlots = 1024;       // (keys and values share the 8 KB of the replay's region bytes)
constexpr int kElectProbes = 16;             // (linear probing: with two probes one entry in twenty found no slot at 650 requests)
// the slot of entry {P, slot of the duplicate} in the election table, claimed if it is not there yet
// (kNone: the request's signature is not in P -- cannot happen, the duplicate was stable -- or no
// room in the table)
__device__ __forceinline__ uint32_t elect_claim(uint32_t *etb, const uint32_t *table, uint32_t P, uint32_t sig)
{
	uint4 lo, hi;
	load_sig_row<true>(table, P, lo, hi);
	const uint32_t m = eq_mask(lo, hi, sig);
	if (!m)
		return kNone;
	const uint32_t key = (P << 3) | ((uint32_t)__ffs((int)m) - 1u);
	uint32_t slot = (key * 2654435761u) >> 22;
	for (int probe = 0; probe < kElectProbes; probe++) {
		const uint32_t old = atomicCAS(&etb[slot], kNone, key);
		if (old == kNone || old == key)
			return slot;
		slot = (slot + 1u) & (kElectSlots - 1u);
	}
	return kNone;
}

// (the region's barrier is final when this runs: every request entered is below it, so the entry's
// writer commits in this round and the others' writes are dead in this round)
__device__ __forceinline__ void elect_entry_lds(uint32_t *etb, uint32_t *etv, uint32_t *table, const DevGeom &g,
		const unsigned long long *owner, const uint4 &rc, uint8_t *state, uint32_t idx, uint32_t bar)
{
	if (state[idx] != ST_DUPW || idx >= bar)
		return;
	const uint32_t P = rc.y & g.hash_mask;
	const uint32_t o = (uint32_t)(ld_agent(&owner[P]) & ((1ull << kIdxBits) - 1ull));
	const uint8_t so = state[o];
	if (so != ST_DUPW && so != ST_DUPW_OK)
		return;
	const uint32_t slot = elect_claim(etb, table, P, rc.x);
	if (slot == kNone)
		return;                                     // (a request that has not looked at A yet holds its slot already)
	atomicMax(&etv[slot], idx);
	state[idx] = ST_DUPW_OK;
}

// commit of a request tagged DUPW_OK by elect_entry_lds: the entry's highest index writes, the
// others' writes are dead (same bucket and slot: the same key) -- provided the writer is below the
// region's barrier as it stands now (a request without a slot in the table may have lowered it during
// the election: the writer then waits, and so do they).  1 = done, 0 = try again
__device__ __forceinline__ int commit_entry_lds(const uint32_t *etb, const uint32_t *etv, uint32_t *table,
		const Req &r, uint32_t P, uint32_t idx, uint32_t bar)
{
	uint4 lo, hi;
	load_sig_row<true>(table, P, lo, hi);
	const uint32_t m = eq_mask(lo, hi, r.sig);
	if (!m)
		return 0;
	const uint32_t sl = (uint32_t)__ffs((int)m) - 1u;
	const uint32_t key = (P << 3) | sl;
	uint32_t slot = (key * 2654435761u) >> 22;
	for (int probe = 0; probe < kElectProbes; probe++) {
		if (etb[slot] == key) {
			const uint32_t w = etv[slot];
			if (w == idx)
				table[loc_word(P, sl)] = r.loc;
			return w < bar ? 1 : 0;
		}
		slot = (slot + 1u) & (kElectSlots - 1u);
	}
	return 0;
}

// Reservation rounds inside the tail work-group over keys[0..m) (kNone = done).
// Same rule as the grid-wide rounds, with decide-then-commit separated by
// work-group barriers.  A region whose lowest pending request is itself the
// round's barrier (hard, or possibly hard) gets that one request replayed
// exactly -- it is first in canonical order and nothing else of its region runs
// in that round -- after which the requests behind it are parallel work again.
// Returns the number of requests still pending.
template <Op OP>
__device__ __forceinline__ uint32_t tail_rounds(Ctl *ctl, const DevGeom &g, uint32_t *table, unsigned long long *owner,
		unsigned long long *cnt, unsigned long long *altmin, const uint4 *reqc, uint8_t *state, uint32_t *keys, uint32_t m,
		uint32_t *s_rbar, uint32_t *s_rmin, uint32_t *s_below, uint32_t *s_progress, uint32_t *s_par, uint32_t *s_left,
		uint32_t *s_dupw, int max_rounds, uint32_t *rounds_used, const Tv &tv, uint32_t stamp_skip, uint32_t *etb)
{
	const uint32_t t = threadIdx.x;
	const unsigned long long stamp0 = ctl->stamp_base + 2 + stamp_skip;   // (stamps earlier calls of this launch used)
	uint32_t *etv = etb + kElectSlots;          // the duplicates' election (elect_entry_lds)
	uint32_t left = m, recent = 0;
	// A list of at most one request per lane (the usual case): the lane keeps its request's cache
	// entry in registers over the phases and rounds -- every phase otherwise begins with that load,
	// one dependent round trip of the three or four a phase consists of.  (Only the lane itself
	// writes the entry's flag word in here; list positions do not move.)
	const bool one = m <= blockDim.x;
	uint4 rc0 = make_uint4(0u, 0u, 0u, 0u);
	if (one && max_rounds > 0 && t < m && keys[t] != kNone)
		rc0 = reqc[keys[t]];
	for (int round = 0; round < max_rounds && left != 0; round++) {
		*rounds_used = (uint32_t)round + 1u;
		const unsigned long long stamp = stamp0 + (unsigned)round;
		if (t < kMaxRegions) {
			s_rbar[t] = kNone;
			s_rmin[t] = kNone;
			s_below[t] = 0u;
		}
		if (t == 0) {
			*s_progress = 0;
			*s_left = 0;
			*s_dupw = 0;
		}
		__syncthreads();
		// register; the count decides who might turn hard (whole waves go round together: the
		// atomics of lanes on one bucket are folded)
		for (uint32_t base = 0; base < m; base += blockDim.x) {
			const uint32_t i = base + t;
			const uint32_t idx = i < m ? keys[i] : kNone;
			const bool live = idx != kNone;
			const uint4 rc = !live ? make_uint4(0u, 0u, 0u, 0u) : one ? rc0 : reqc[idx];
			const uint32_t P = rc.y & g.hash_mask;
			const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
			const bool reg = live && state[idx] != ST_DEFER;      // (a deferred request sits this round out)
			reserve_wave(owner, P, A, owner_key(stamp, live ? idx : 0u), reg);
			if (OP != OP_DELETE) {
				bump_wave(cnt, P, stamp, A == P, reg);
				bump_wave(cnt, A, stamp, true, reg && A != P);
				atomic_min_dedup(altmin, A, owner_key(stamp, live ? idx : 0u), reg);
				if (live)
					atomicMin(&s_rmin[P >> g.region_shift], idx);
			}
		}
		wg_sync();
		// decide: the lowest index per region that is hard or might turn hard
		if (OP != OP_DELETE) {
			for (uint32_t i = t; i < m; i += blockDim.x) {
				const uint32_t idx = keys[i];
				if (idx == kNone)
					continue;
				const uint4 rc = one ? rc0 : reqc[idx];
				const uint32_t P = rc.y & g.hash_mask;
				const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
				const uint32_t region = P >> g.region_shift;
				if (state[idx] == ST_DEFER) {
					atomicMin(&s_rbar[region], idx);        // not registered: nothing of its region passes it
					state[idx] = ST_TAIL;                   // (back in line next round)
					continue;
				}
				if (state[idx] >= ST_DUPW)
					state[idx] = ST_TAIL;                   // last round's tag
				if (idx >= s_rbar[region])
					continue;
				// The grid-wide rounds' commits on the request's alternate bucket, looked at ONCE, in the
				// first round in which the request is not behind a barrier -- before anything later than it
				// in its region can commit in the tail, where commits are final.  A later commit there is an
				// order violation (the request is the round's barrier of its region, the commit is taken
				// back, select_from): a request that did not register on the bucket in the mark pass
				// (kLazyBit) may find a later request's; any request may find the commit of a group of
				// duplicates that reaches beyond it (kLeadBit), whose entry it might replace.  None: an
				// ordinary request from here on -- it registers on both buckets in every round of the tail.
				// The part of the check that concerns A is put off for a request that holds a duplicate
				// nothing can evict (below): it will never go to A -- IF it is through by the end of this
				// round.  Whether it will be is known once every request has been classified: the second
				// pass below makes the check for those that will not.  (Zipf SET keys: the duplicates of a
				// key that shares its bucket with a hotter one all come here, registered on P only, and a
				// later request's commit on their alternate bucket, which they never visit, cost one to
				// seven order violations per 50/50 batch.)
				const bool unchecked = OP != OP_DELETE && !(rc.w & kCheckedBit);
				if (unchecked && conflict_on(tv, g, P, idx, rc.x, OP == OP_2CHOICE && A == P)) {
					report_violation(tv, g, P, idx, P);
					atomicMin(&s_rbar[region], idx);
					continue;
				}
				uint4 lo, hi;
				load_sig_row<true>(table, P, lo, hi);
				const unsigned long long key = owner_key(stamp, idx);
				const bool ownP = ld_agent(&owner[P]) == key;
				const bool ownA = A == P || ld_agent(&owner[A]) == key;
				const unsigned long long cP = ld_agent(&cnt[P]);
				// A duplicate only counts while nothing can evict it before this request's turn
				// (and never for signature 0, which "matches" empty slots anybody may fill):
				// the request owns the bucket (every other registrant comes later), or no
				// registrant BEFORE it has the bucket as its alternate (a later one does not own
				// the bucket while this request is registered on it).
				const bool dupw = rc.x != 0u && eq_mask(lo, hi, rc.x) && (ownP || !alt_before(altmin, P, stamp, idx));
				if (unchecked && !dupw) {
					if (A != P && conflict_on(tv, g, A, idx, rc.x, OP == OP_2CHOICE || (rc.w & kLazyBit) != 0u)) {
						report_violation(tv, g, P, idx, A);
						atomicMin(&s_rbar[region], idx);
						continue;
					}
					const_cast<uint4 *>(reqc)[idx].w = (rc.w & ~kLazyBit) | kCheckedBit;
					rc0.w = one ? (rc.w & ~kLazyBit) | kCheckedBit : rc0.w;
				}
				if (dupw) {
					state[idx] = ST_DUPW;                   // a pure loc write: see elect_entry_lds
					*s_dupw = 1u;                           // (still unchecked: the second pass below)
					continue;
				}
				if (OP != OP_CUCKOO)
					continue;                               // two-choice: nothing turns hard
				uint32_t room = (uint32_t)__popc(eq_mask(lo, hi, 0u));
				const bool clean = ownP && ownA;
				if (clean && room)
					continue;
				uint32_t others = cnt_of(cP, stamp) - 1u;
				if (A != P) {
					load_sig_row<true>(table, A, lo, hi);
					if (rc.x != 0u && eq_mask(lo, hi, rc.x) && ownA)
						continue;
					room += (uint32_t)__popc(eq_mask(lo, hi, 0u));
					others += cnt_of(ld_agent(&cnt[A]), stamp) - 1u;
				}
				bool bad = clean ? room == 0u : room < 1u + others;
				if (bad && !clean)
					atomicAdd(&ctl->dbg[2], 1u);
				if (bad && clean) {
					// hard right now: it may still run this round if its chain is all its own
					if (walk_and_claim(table, g, req_of(rc), P, A, idx, owner, stamp, tv, s_rbar, sig_at(lo, hi, rc.x & 7u))) {
						state[idx] = ST_CHAIN_OK;
						bad = false;
					} else {
						atomicAdd(&ctl->dbg[1], 1u);
					}
				}
				if (bad)
					atomicMin(&s_rbar[region], idx);
			}
			wg_sync();
			// the duplicate writers below the round's barrier elect the last one per bucket
			if (*s_dupw) {
				// Second pass of the check: a duplicate holder that has not looked at A yet and will NOT be
				// through by the end of this round -- it is beyond its region's barrier as it stands now,
				// the head of its bucket is no duplicate writer, or the election table has no room for its
				// entry, so it will not take part in the election -- looks now, and is the barrier of its
				// region if it finds a later commit there: a pending request that has not looked must never
				// see anything above it commit.  (One that is beyond a barrier set in THIS pass stays as it
				// is: nothing at or above that barrier commits either.  No barrier moves after this pass:
				// whoever enters the election is below the final one.)
				for (uint32_t q = t; q < kElectSlots; q += blockDim.x) {
					etb[q] = kNone;
					etv[q] = 0u;
				}
				__syncthreads();
				for (uint32_t i = t; i < m; i += blockDim.x) {
					const uint32_t idx = keys[i];
					if (idx == kNone || state[idx] != ST_DUPW)
						continue;
					const uint4 rc = one ? rc0 : reqc[idx];
					if (rc.w & kCheckedBit)
						continue;
					const uint32_t P = rc.y & g.hash_mask;
					const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
					const uint32_t region = P >> g.region_shift;
					const uint32_t o = (uint32_t)(ld_agent(&owner[P]) & ((1ull << kIdxBits) - 1ull));
					// (it takes part in the election -- its entry's slot in the table is claimed here and now,
					// the election only has to find it -- and is through this round)
					if (idx < s_rbar[region] && state[o] == ST_DUPW && elect_claim(etb, table, P, rc.x) != kNone)
						continue;
					if (A != P && conflict_on(tv, g, A, idx, rc.x, OP == OP_2CHOICE || (rc.w & kLazyBit) != 0u)) {
						report_violation(tv, g, P, idx, A);
						atomicMin(&s_rbar[region], idx);
						continue;
					}
					const_cast<uint4 *>(reqc)[idx].w = (rc.w & ~kLazyBit) | kCheckedBit;
					rc0.w = one ? (rc.w & ~kLazyBit) | kCheckedBit : rc0.w;
				}
				wg_sync();
				for (uint32_t i = t; i < m; i += blockDim.x) {
					const uint32_t idx = keys[i];
					if (idx != kNone) {
						const uint4 rc = one ? rc0 : reqc[idx];
						elect_entry_lds(etb, etv, table, g, owner, rc, state, idx,
								s_rbar[(rc.y & g.hash_mask) >> g.region_shift]);
					}
				}
				wg_sync();
			}
		}
		// commit
		uint32_t my_bar = kNone;                    // list position of a round barrier this lane holds back (see below)
		for (uint32_t i = t; i < m; i += blockDim.x) {
			const uint32_t idx = keys[i];
			if (idx == kNone)
				continue;
			const Req r = req_of(one ? rc0 : reqc[idx]);
			const uint32_t P = r.hash & g.hash_mask;
			const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
			const unsigned long long key = owner_key(stamp, idx);
			const uint32_t region = P >> g.region_shift;
			const uint32_t bar = OP != OP_DELETE ? s_rbar[region] : kNone;
			bool done = false, settled = false;         // settled: nothing more to try this round
			if (OP != OP_DELETE) {
				const uint8_t st = state[idx];
				if (idx == bar) {
					if (st >= ST_CHAIN_OK)
						state[idx] = ST_TAIL;
					if (idx == s_rmin[region]) {
						// first of its region: exact, alone (its chain checked for later commits on its way)
						done = apply_serial_checked<OP>(table, g, r, idx, tv) == 0;
					} else if (my_bar == kNone) {
						my_bar = i;                         // its turn may come at the end of this round
						continue;
					}
					settled = true;
				} else if (idx > bar || st == ST_DUPW) {
					if (st >= ST_CHAIN_OK)
						state[idx] = ST_TAIL;
					settled = true;
				} else if (st == ST_CHAIN_OK) {
					done = run_owned_chain(table, g, r, P, A, idx, owner, stamp, tv);   // chain over buckets only it holds
					if (!done)
						state[idx] = ST_TAIL;           // an earlier chain took one of its buckets over
					settled = true;
				} else if (st == ST_DUPW_OK) {
					done = commit_entry_lds(etb, etv, table, r, P, idx, bar) != 0;
					if (!done)
						state[idx] = ST_TAIL;
					settled = true;
				}
			}
			if (!settled && ld_agent(&owner[P]) == key && (A == P || ld_agent(&owner[A]) == key))
				done = apply_static_late<OP>(table, g, r, P, A, idx, tv) == 0;
			if (done) {
				state[idx] = ST_DONE;
				MKV_TR(1u, idx, (uint32_t)round, 0u);
				keys[i] = kNone;
				atomicAdd(s_progress, 1u);
			} else {
				atomicAdd(s_left, 1u);
				if (idx < bar)
					atomicAdd(&s_below[region], 1u);
			}
		}
		if (OP != OP_DELETE) {
			// The round's barrier of a region -- hard, or not provably harmless -- has had to wait for
			// the requests before it.  If they are all through now, it is the first pending request of
			// its region: it runs exactly, alone, at the end of this round instead of the next.
			wg_sync();
			if (my_bar != kNone) {
				const uint32_t idx = keys[my_bar];
				const Req r = req_of(one ? rc0 : reqc[idx]);
				const uint32_t region = (r.hash & g.hash_mask) >> g.region_shift;
				if (s_below[region] == 0u && apply_serial_checked<OP>(table, g, r, idx, tv) == 0) {
					state[idx] = ST_DONE;
					MKV_TR(1u, idx, (uint32_t)round, 1u);
					keys[my_bar] = kNone;
					atomicAdd(s_progress, 1u);
				} else {
					atomicAdd(s_left, 1u);
				}
			}
		}
		wg_sync();
		const uint32_t progress = *s_progress;
		left = *s_left;
		if (t == 0) {
			*s_par += progress;
			ctl->dbg[5] += progress + left;
			ctl->dbg[6] += 1u;
			ctl->dbg[4] += progress;
		}
		__syncthreads();
		// a round that resolves less than the one-at-a-time replay would in its time
		// (duplicates of one key, chains in a row) is not worth another
		recent = round == 0 ? progress : (3u * recent + progress) / 4u;
		// an order violation blocks its region until the
		// caller has dealt with it
		if (progress == 0 || (round >= 3 && recent < kMinRoundProgress) || ld_agent32(&tv.ctl->tflags) != 0u)
			break;
	}
	return left;
}

// The same rounds run by every work-group of the tail launch over the global list
// keys[0..m): for a final pass that is left with more than one work-group can chew.
// Round state lives in the control block; same-address atomics are folded per
// work-group in LDS first.  Four grid-wide meetings per round.  All groups read the
// same totals after the third and so leave the loop together.
template <Op OP>
__device__ __forceinline__ uint32_t grid_rounds(Ctl *ctl, const DevGeom &g, uint32_t *table, unsigned long long *owner,
		unsigned long long *cnt, unsigned long long *altmin, const uint4 *reqc, uint8_t *state, uint32_t *keys, uint32_t m,
		uint32_t *s_rbar, uint32_t *s_rmin, uint32_t *s_below, uint32_t *s_progress, uint32_t *s_left, uint32_t *s_dupw,
		int max_rounds, uint32_t *rounds_used, uint32_t *par_total, uint32_t *ticks, const Tv &tv, uint32_t stamp_skip,
		const Fold &fold, uint32_t min_progress)
{
	const uint32_t t = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
	const uint32_t first = wg * blockDim.x + t, span = nwg * blockDim.x;
	fold_clear<false, kFoldLog2>(fold);
	// ticks[0..7]: register, its meeting, decide, its meeting, election + meeting, commit, its meeting
#ifdef MKV_DIAGNOSTICS
	unsigned long long tick = wall_clock64();
#define MKV_GTICK(k) do { const unsigned long long now_ = wall_clock64(); ticks[k] += (uint32_t)(now_ - tick); tick = now_; } while (0)
#else
#define MKV_GTICK(k) ((void)0)
#endif
	const unsigned long long stamp0 = ctl->stamp_base + 2 + stamp_skip;
	unsigned long long *emax = cnt + ((size_t)g.hash_mask + 1u);   // election words behind the counts
	uint32_t left = m, nth = 0, used = 0, par = 0, recent = 0;
	for (int round = 0; round < max_rounds && left != 0; round++) {
		const unsigned long long stamp = stamp0 + (unsigned)round;
		const uint32_t set = (uint32_t)round & 1u;
		used = (uint32_t)round + 1u;
		if (t < kMaxRegions) {
			s_rbar[t] = kNone;
			s_rmin[t] = kNone;
			s_below[t] = 0u;
		}
		if (t == 0) {
			*s_progress = 0;
			*s_left = 0;
			*s_dupw = 0;
		}
		__syncthreads();
		// register (whole waves go round together: atomics of lanes on one bucket are folded)
		for (uint32_t base = wg * blockDim.x; base < m; base += span) {
			const uint32_t i = base + t;
			const uint32_t idx = i < m ? keys[i] : kNone;
			const bool live = idx != kNone;
			const uint4 rc = live ? reqc[idx] : make_uint4(0u, 0u, 0u, 0u);
			const uint32_t P = rc.y & g.hash_mask;
			const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
			const bool reg = live && state[idx] != ST_DEFER;      // (a deferred request sits this round out)
			// (folded per work-group: fold_enter; an entry without a slot goes alone)
			const unsigned long long key = owner_key(stamp, live ? idx : 0u);
			if (reg && !fold_enter<false, kFoldLog2>(fold, P, key, OP != OP_DELETE, A == P)) {
				atomicMin(&owner[P], key);
				if (OP != OP_DELETE) {
					if (A == P)
						bump_alt(cnt, P, stamp);
					else
						bump(cnt, P, stamp);
				}
			}
			if (reg && A != P && !fold_enter<false, kFoldLog2>(fold, A, key, OP != OP_DELETE, true)) {
				atomicMin(&owner[A], key);
				if (OP != OP_DELETE)
					bump_alt(cnt, A, stamp);
			}
			if (OP != OP_DELETE && live)
				atomicMin(&s_rmin[P >> g.region_shift], idx);
			fold_flush<false, false, kFoldLog2>(fold, owner, cnt, stamp);
			if (OP != OP_DELETE) {
				// the lowest registrant that has the bucket as its alternate (alt_before)
				if (reg && !fold_enter<false, kFoldLog2>(fold, A, key, false, false))
					atomicMin(&altmin[A], key);
				fold_flush<false, false, kFoldLog2>(fold, altmin, cnt, stamp);
			}
		}
		__syncthreads();
		if (OP == OP_CUCKOO && t < g.nregions && s_rmin[t] != kNone)
			atomicMin(&ctl->g_rmin[set][t], s_rmin[t]);
		MKV_GTICK(0);
		grid_sync_nth(&ctl->g_sync, nwg, &nth);
		MKV_GTICK(1);
		// decide: the lowest index per region that is hard or might turn hard
		if (OP != OP_DELETE) {
			for (uint32_t i = first; i < m; i += span) {
				const uint32_t idx = keys[i];
				if (idx == kNone)
					continue;
				const uint4 rc = reqc[idx];
				const uint32_t P = rc.y & g.hash_mask;
				const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
				const uint32_t region = P >> g.region_shift;
				if (state[idx] == ST_DEFER) {
					atomicMin(&s_rbar[region], idx);        // not registered: nothing of its region passes it
					state[idx] = ST_TAIL;                   // (back in line next round)
					continue;
				}
				if (state[idx] >= ST_DUPW)
					state[idx] = ST_TAIL;                   // last round's tag
				if (idx >= s_rbar[region])
					continue;
				// The grid-wide rounds' commits on the request's alternate bucket, looked at ONCE, in the
				// first round in which the request is not behind a barrier -- before anything later than it
				// in its region can commit in the tail, where commits are final.  A later commit there is an
				// order violation (the request is the round's barrier of its region, the commit is taken
				// back, select_from): a request that did not register on the bucket in the mark pass
				// (kLazyBit) may find a later request's; any request may find the commit of a group of
				// duplicates that reaches beyond it (kLeadBit), whose entry it might replace.  None: an
				// ordinary request from here on -- it registers on both buckets in every round of the tail.
				if (OP != OP_DELETE && !(rc.w & kCheckedBit)) {
					const uint32_t vb = early_conflict<OP>(tv, g, P, A, idx, rc.x, rc.w);
					if (vb != kNone) {
						report_violation(tv, g, P, idx, vb);
						atomicMin(&s_rbar[region], idx);
						continue;
					}
					const_cast<uint4 *>(reqc)[idx].w = (rc.w & ~kLazyBit) | kCheckedBit;
				}
				uint4 lo, hi;
				load_sig_row<true>(table, P, lo, hi);
				const unsigned long long key = owner_key(stamp, idx);
				const bool ownP = ld_agent(&owner[P]) == key;
				const bool ownA = A == P || ld_agent(&owner[A]) == key;
				const unsigned long long cP = ld_agent(&cnt[P]);
				// A duplicate only counts while nothing can evict it before this request's turn
				// (and never for signature 0, which "matches" empty slots anybody may fill):
				// the request owns the bucket (every other registrant comes later), or no
				// registrant BEFORE it has the bucket as its alternate (a later one does not own
				// the bucket while this request is registered on it).
				if (rc.x != 0u && eq_mask(lo, hi, rc.x) && (ownP || !alt_before(altmin, P, stamp, idx))) {
					state[idx] = ST_DUPW;                   // a pure loc write: see elect_last_writer
					*s_dupw = 1u;
					continue;
				}
				if (OP != OP_CUCKOO)
					continue;                               // two-choice: nothing turns hard
				uint32_t room = (uint32_t)__popc(eq_mask(lo, hi, 0u));
				const bool clean = ownP && ownA;
				if (clean && room)
					continue;
				uint32_t others = cnt_of(cP, stamp) - 1u;
				if (A != P) {
					load_sig_row<true>(table, A, lo, hi);
					if (rc.x != 0u && eq_mask(lo, hi, rc.x) && ownA)
						continue;
					room += (uint32_t)__popc(eq_mask(lo, hi, 0u));
					others += cnt_of(ld_agent(&cnt[A]), stamp) - 1u;
				}
				bool bad = clean ? room == 0u : room < 1u + others;
				if (bad && clean) {
					if (walk_and_claim(table, g, req_of(rc), P, A, idx, owner, stamp, tv, s_rbar, sig_at(lo, hi, rc.x & 7u))) {
						state[idx] = ST_CHAIN_OK;
						bad = false;
					}
				}
				if (bad)
					atomicMin(&s_rbar[region], idx);
			}
			__syncthreads();
			if (t < g.nregions && s_rbar[t] != kNone)
				atomicMin(&ctl->g_rbar[set][t], s_rbar[t]);
			if (t == 0 && *s_dupw)
				atomicOr(&ctl->g_dupw[set], 1u);
			MKV_GTICK(2);
			grid_sync_nth(&ctl->g_sync, nwg, &nth);
			MKV_GTICK(3);
			if (t < kMaxRegions) {
				s_rbar[t] = t < g.nregions ? ld_agent32(&ctl->g_rbar[set][t]) : kNone;
				s_rmin[t] = t < g.nregions ? ld_agent32(&ctl->g_rmin[set][t]) : kNone;
			}
			__syncthreads();
			// the duplicate writers below the round's barrier elect the last one per bucket
			// (all work-groups read the same flag: they take or skip the meeting together)
			if (ld_agent32(&ctl->g_dupw[set])) {
				fold_clear<true, kFoldLog2>(fold);
				for (uint32_t base = wg * blockDim.x; base < m; base += span) {
					const uint32_t i = base + t;
					const uint32_t idx = i < m ? keys[i] : kNone;
					const bool live = idx != kNone;
					const uint32_t P = live ? reqc[idx].y & g.hash_mask : 0u;
					// (elect_last_writer, its atomicMax folded per work-group)
					bool take = live && state[idx] == ST_DUPW && idx < s_rbar[P >> g.region_shift];
					if (take) {
						const uint32_t o = (uint32_t)(ld_agent(&owner[P]) & ((1ull << kIdxBits) - 1ull));
						const uint8_t so = state[o];
						take = so == ST_DUPW || so == ST_DUPW_OK;
					}
					const unsigned long long ekey = (stamp << kIdxBits) | idx;
					if (take && !fold_enter<true, kFoldLog2>(fold, P, ekey, false, false))
						atomicMax(&emax[P], ekey);
					if (take)
						state[idx] = ST_DUPW_OK;
					fold_flush<true, true, kFoldLog2>(fold, emax, cnt, stamp);
				}
				fold_clear<false, kFoldLog2>(fold);
				grid_sync_nth(&ctl->g_sync, nwg, &nth);
				MKV_GTICK(4);
			}
		}
		// the idle set is made ready for the next round
		if (wg == 0 && t < kMaxRegions) {
			ctl->g_rbar[set ^ 1u][t] = kNone;
			ctl->g_rmin[set ^ 1u][t] = kNone;
			ctl->g_below[set ^ 1u][t] = 0u;
			if (t == 0)
				ctl->g_progress[set ^ 1u] = ctl->g_left[set ^ 1u] = ctl->g_dupw[set ^ 1u] = 0;
		}
		// commit
		uint32_t my_done = 0, my_left = 0;
		uint32_t my_bar = kNone;                    // list position of a round barrier this lane holds back (see below)
		for (uint32_t i = first; i < m; i += span) {
			const uint32_t idx = keys[i];
			if (idx == kNone)
				continue;
			const Req r = req_of(reqc[idx]);
			const uint32_t P = r.hash & g.hash_mask;
			const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
			const unsigned long long key = owner_key(stamp, idx);
			const uint32_t region = P >> g.region_shift;
			const uint32_t bar = OP != OP_DELETE ? s_rbar[region] : kNone;
			bool done = false, settled = false;         // settled: nothing more to try this round
			if (OP != OP_DELETE) {
				const uint8_t st = state[idx];
				if (idx == bar) {
					if (st >= ST_CHAIN_OK)
						state[idx] = ST_TAIL;
					if (idx == s_rmin[region]) {
						// first of its region: exact, alone (its chain checked for later commits on its way)
						done = apply_serial_checked<OP>(table, g, r, idx, tv) == 0;
					} else if (my_bar == kNone) {
						my_bar = i;                         // its turn may come at the end of this round
						continue;
					}
					settled = true;
				} else if (idx > bar || st == ST_DUPW) {
					if (st >= ST_CHAIN_OK)
						state[idx] = ST_TAIL;
					settled = true;
				} else if (st == ST_CHAIN_OK) {
					done = run_owned_chain(table, g, r, P, A, idx, owner, stamp, tv);   // chain over buckets only it holds
					if (!done)
						state[idx] = ST_TAIL;           // an earlier chain took one of its buckets over
					settled = true;
				} else if (st == ST_DUPW_OK) {
					done = commit_last_writer(table, g, emax, reqc, r, P, idx) != 0;
					if (!done)
						state[idx] = ST_TAIL;
					settled = true;
				}
			}
			if (!settled && ld_agent(&owner[P]) == key && (A == P || ld_agent(&owner[A]) == key))
				done = apply_static_late<OP>(table, g, r, P, A, idx, tv) == 0;
			if (done) {
				state[idx] = ST_DONE;
				MKV_TR(1u, idx, (uint32_t)round, 2u);
				keys[i] = kNone;
				my_done++;
			} else {
				my_left++;
				if (idx < bar)
					atomicAdd(&s_below[region], 1u);
			}
		}
		if (OP != OP_DELETE) {
			// The round's barrier of a region has had to wait for the requests before it; if they
			// are all through now it is the first pending request of its region and runs exactly,
			// alone, at the end of this round instead of the next.  (One more meeting -- taken by
			// all work-groups or none: they hold the same barriers and minima.)
			bool late_bar = false;
			for (uint32_t k = 0; k < g.nregions; k++)
				late_bar = late_bar || (s_rbar[k] != kNone && s_rbar[k] != s_rmin[k]);
			if (late_bar) {
				__syncthreads();
				if (t < g.nregions && s_below[t])
					atomicAdd(&ctl->g_below[set][t], s_below[t]);
				grid_sync_nth(&ctl->g_sync, nwg, &nth);
				if (my_bar != kNone) {
					const uint32_t idx = keys[my_bar];
					const Req r = req_of(reqc[idx]);
					const uint32_t region = (r.hash & g.hash_mask) >> g.region_shift;
					if (ld_agent32(&ctl->g_below[set][region]) == 0u && apply_serial_checked<OP>(table, g, r, idx, tv) == 0) {
						state[idx] = ST_DONE;
						MKV_TR(1u, idx, (uint32_t)round, 3u);
						keys[my_bar] = kNone;
						my_done++;
					} else {
						my_left++;
					}
				}
			} else if (my_bar != kNone) {
				my_left++;
			}
		}
		if (my_done)
			atomicAdd(s_progress, my_done);
		if (my_left)
			atomicAdd(s_left, my_left);
		__syncthreads();
		if (t == 0) {
			if (*s_progress)
				atomicAdd(&ctl->g_progress[set], *s_progress);
			if (*s_left)
				atomicAdd(&ctl->g_left[set], *s_left);
		}
		MKV_GTICK(5);
		grid_sync_nth(&ctl->g_sync, nwg, &nth);
		MKV_GTICK(6);
		const uint32_t progress = ld_agent32(&ctl->g_progress[set]);
		left = ld_agent32(&ctl->g_left[set]);
		par += progress;
		// rounds go on while they resolve more than the one-at-a-time replay would in their
		// time, judged over the last few (the round after a chain is often a thin one)
		recent = round == 0 ? progress : (3u * recent + progress) / 4u;
		// (the flags were last written before the meeting above: every work-group reads the same)
		// (so few are left that one work-group takes them in one go: it continues alone, with barriers
		// in place of the grid-wide meetings)
		if (progress == 0 || (round >= 3 && recent < min_progress) || ld_agent32(&tv.ctl->tflags) != 0u ||
				left <= kGridHandover)
			break;
	}
	*rounds_used = used;
	*par_total = par;
	return left;
}

// An order violation {j, X}: the chain of the pending request j steps on bucket X, which
// a LATER commit of the grid-wide rounds has used.  What has to go:
//   (a) every logged commit k > j on X (at most two: the candidates of its reservation words);
//   (b) for every commit u taken back, every logged commit k' > u on one of the buckets of u's
//       reach (primary, alternate -- used or not: u will run again and may need it --, chain
//       buckets): it ran there after u.
// Commits that keep their place ran on their buckets before everything taken back there -- or
// on other buckets -- so they stand.  One lane per violation follows the buckets, with a stack
// of {bucket, index} in LDS; a commit is selected by flagging its tag ST_SELECT
// and noting it in the list `sel`.  Returns false if its stack or the list overflowed (the caller then takes back
// the whole region suffix).  Nothing is written to the table here.
constexpr uint32_t kSelCap = 2048, kLaneStack = 24;

// A group's commit (kLeadBit) by a request BELOW the asking index that reaches beyond it is not taken
// back -- what lies below the asking index may already have final commits of the tail behind it --
// but cut: noted in trunc[] as {leader, asking index}; the caller lets the group end before that
// index (truncate_groups).
constexpr uint32_t kTruncCap = 128;

__device__ __forceinline__ bool select_from(const DevGeom &g, uint8_t *state, const uint4 *reqc, const Undo *undo,
		const ChainLog *chain_logs, const Tv &tv, uint32_t j, uint32_t X, uint2 *stack, uint32_t *sel,
		uint32_t *nsel, uint2 *trunc, uint32_t *ntrunc)
{
	uint32_t sp = 0;
	stack[sp++] = make_uint2(X, j);
	bool ok = true;
	while (sp != 0u) {
		const uint2 e = stack[--sp];
		uint32_t cand[3];
		candidates_of(tv, ld_agent(&tv.owner1[e.x]), ld_agent(&tv.owner2[e.x]), ld_agent(&tv.second[e.x]), cand[0], cand[1], cand[2]);
		for (int ci = 0; ci < 3; ci++) {
			const uint32_t u = cand[ci];
			uint32_t eff = 0;
			if (u == kNone || (ci == 1 && u == cand[0]))
				continue;
			if (!commit_used(tv, g, u, e.x, &eff) || eff <= e.y)
				continue;                           // no commit there, not a later one, or selected already
			if (u < e.y) {                          // a group that begins below the asking index: cut, not taken back
				const uint32_t at = atomicAdd(ntrunc, 1u);
				MKV_TR(6u, u, e.y, 0u);
				if (at < kTruncCap)
					trunc[at] = make_uint2(u, e.y);
				else
					ok = false;
				continue;
			}
			const uint32_t st = state[u];
			if (!tag_logged(st))
				continue;
			// (the flag is set with an atomic on the tags' word: exactly one lane selects a commit,
			// the list holds it once)
			const uint32_t sh = (u & 3u) * 8u;
			if ((atomicOr(reinterpret_cast<uint32_t *>(state) + (u >> 2), (uint32_t)ST_SELECT << sh) >> sh) & ST_SELECT)
				continue;
			const uint32_t at = atomicAdd(nsel, 1u);
			if (at < kSelCap)
				sel[at] = u;
			else
				ok = false;
			const uint4 rc = reqc[u];
			const uint32_t P = rc.y & g.hash_mask;
			const uint32_t A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
			const bool chain = st == ST_CHAIN1 || st == ST_CHAIN2;
			const ChainLog *L = chain ? &chain_logs[undo[u].word0] : nullptr;
			const uint32_t nl = chain ? (L->n < (uint32_t)kChainLogWords ? L->n : (uint32_t)kChainLogWords) : 0u;
			if (sp + 2u + (nl + 1u) / 2u > kLaneStack) {
				ok = false;
				continue;
			}
			stack[sp++] = make_uint2(P, u);
			if (A != P)
				stack[sp++] = make_uint2(A, u);
			uint32_t last = A;
			for (uint32_t i = 0; i < nl; i++) {
				const uint32_t b = L->word[i] >> 4;
				if (b != last && b != P && b != A)
					stack[sp++] = make_uint2(b, u);
				last = b;
			}
		}
	}
	return ok;
}

// takes the selected commit u (tag already checked) back: its words restored, its tag ST_DEFER
// (pending, sitting out one round -- and no longer a commit to anybody who asks commit_used)
__device__ __forceinline__ void take_back(const DevGeom &g, uint32_t *table, uint8_t *state, const uint4 *reqc,
		const Undo *undo, const ChainLog *chain_logs, uint32_t u, bool chain)
{
	const Undo ur = undo[u];
	if (chain) {
		const ChainLog *L = &chain_logs[ur.word0];
		const uint32_t n = L->n < (uint32_t)kChainLogWords ? L->n : (uint32_t)kChainLogWords;
		for (uint32_t k = n; k-- > 0;)
			table[L->word[k]] = L->old[k];
	} else {
		if (ur.word1 != kNoWord)
			table[ur.word1] = ur.old1;
		if (ur.word0 != kNoWord)
			table[ur.word0] = ur.old0;
	}
	state[u] = ST_DEFER;
	MKV_TR(4u, u, 0u, 0u);
}

// Roll back the commits of the grid-wide rounds that are flagged ST_SELECT (select_from)
// or lie at or beyond thr[region of the request] (kNone: none by index in that region), newest
// round first; the requests become pending again (tag TAIL).  A thread owns the 16 tags it loads; the loads of a step are issued together.
// `nwg` work-groups share the scan and meet between the two rounds on `sync_counter`.
// Returns the number this thread rolled back.
__device__ __forceinline__ uint32_t roll_back_from(const DevGeom &g, uint32_t *table, uint8_t *state, const uint4 *reqc,
		const Undo *undo, const ChainLog *chain_logs, const uint32_t *thr, uint32_t lo_idx, uint32_t total,
		uint32_t wg, uint32_t nwg, uint32_t *sync_counter, uint2 *trunc, uint32_t *ntrunc)
{
	// (a group's commit by a request below its region's threshold that reaches to or beyond it is cut
	// there, like select_from cuts it: trunc[], the caller runs truncate_groups)
	const uint32_t *reqw = reinterpret_cast<const uint32_t *>(reqc);
	uint32_t undone = 0;
	for (int pass = 0; pass < 2; pass++) {
		const uint32_t tag = pass == 0 ? ST_ROUND2 : ST_ROUND1;
		const uint32_t ctag = pass == 0 ? ST_CHAIN2 : ST_CHAIN1;
		for (uint32_t idx = lo_idx + wg * blockDim.x + threadIdx.x; idx < total && idx >= lo_idx;
				idx += nwg * blockDim.x) {
			const uint32_t st = state[idx];
			if ((st & 0x7fu) != tag && (st & 0x7fu) != ctag)
				continue;
			if (!(st & (uint32_t)ST_SELECT)) {
				const uint32_t hash = reqw[(size_t)idx * 4u + 1u], aux = reqw[(size_t)idx * 4u + 3u];
				// (the one commit of a group of duplicates stands for writes up to the group's last member)
				const uint32_t eff = (aux & kLeadBit) ? (aux >> 8) : idx;
				const uint32_t th = thr[(hash & g.hash_mask) >> g.region_shift];
				if (eff < th)
					continue;
				if (idx < th) {                         // (only a group's leader has eff > idx)
					const uint32_t at = atomicAdd(ntrunc, 1u);
					if (at < kTruncCap)
						MKV_TR(6u, idx, th - 1u, 1u);
						trunc[at] = make_uint2(idx, th - 1u);   // (members at or beyond the threshold are pending again;
						                                        // beyond kTruncCap: met again in the next turn)
					continue;
				}
			}
			const Undo u = undo[idx];
			if ((st & 0x7fu) == ctag) {
				const ChainLog *L = &chain_logs[u.word0];
				const uint32_t n = L->n < (uint32_t)kChainLogWords ? L->n : (uint32_t)kChainLogWords;
				for (uint32_t k = n; k-- > 0;)
					table[L->word[k]] = L->old[k];
			} else {
				if (u.word1 != kNoWord)
					table[u.word1] = u.old1;
				if (u.word0 != kNoWord)
					table[u.word0] = u.old0;
			}
			state[idx] = ST_TAIL;
			MKV_TR(5u, idx, thr[(reqw[(size_t)idx * 4u + 1u] & g.hash_mask) >> g.region_shift], st);
			if (reqw[(size_t)idx * 4u + 3u] & kLeadBit) {       // its shadows are requests of their own again
				const uint32_t at = atomicAdd(ntrunc, 1u);
				if (at < kTruncCap)
					trunc[at] = make_uint2(idx, idx);
			}
			undone++;
		}
		if (pass == 0)
			grid_sync(sync_counter, nwg);   // every round-2 undo lands before any round-1 undo
		else
			wg_sync();
	}
	return undone;
}

// Gather the unfinished requests of [lo_idx, total) into list[] (any order), counting in
// *counter (LDS for one work-group, the control block for several).
// (shadows: only looked at when a leader may have let its shadows down without reviving them itself)
__device__ __forceinline__ void gather_pending(uint8_t *state, const Undo *undo, const uint4 *reqc, uint32_t *list,
		uint32_t *counter, uint32_t lo_idx, uint32_t total, uint32_t wg, uint32_t nwg, bool shadows)
{
	const uint32_t t = threadIdx.x;
	const uint32_t lo16 = lo_idx & ~15u;
	const uint4 *st4 = reinterpret_cast<const uint4 *>(state);
	for (uint32_t base = lo16 + (t * nwg + wg) * 16u; base < total && base >= lo16; base += nwg * blockDim.x * 16u) {
		const uint4 v = st4[base >> 4];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		uint32_t take = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint32_t st = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
			const uint32_t idx = base + (uint32_t)k;
			if (idx >= lo_idx && idx < total && tag_pending(st))
				take |= 1u << k;
			if (shadows && idx >= lo_idx && idx < total && st == ST_SHADOW && shadow_live(state, undo, reqc, idx)) {
				take |= 1u << k;                    // its group's commit does not stand (any more): a request like any other
				MKV_TR(9u, idx, undo[idx].word0, 0u);
				state[idx] = ST_TAIL;
			}
		}
		if (!take)
			continue;
		uint32_t at = atomicAdd(counter, (uint32_t)__popc(take));
#pragma unroll
		for (int k = 0; k < 16; k++)
			if ((take >> k) & 1u)
				list[at++] = base + (uint32_t)k;
	}
}

// The location word a group's commit stored the group's last location in.
__device__ __forceinline__ uint32_t group_loc_word(const Undo &u, uint32_t st, const ChainLog *chain_logs)
{
	if (st == ST_CHAIN1 || st == ST_CHAIN2)
		return chain_logs[u.word0].word[1];         // (the chain's first eviction put the request itself into A)
	return u.word1 != kNoWord ? u.word1 : u.word0;  // a new entry: {signature word, location word}; a duplicate: its location word
}

// Cuts groups of duplicates (select_from, roll-back by region): the group of leader c ends before
// index `cut` -- its members beyond are pending requests again (tag TAIL: they lie beyond the
// violation's index, nothing there has been committed by the tail; appended to list[] if given,
// counted in *revived), the entry's location is that of the last member that remains, and the
// commit counts as that member's index from here on.  An entry {c, c}: the leader's commit itself
// has been taken back, all its shadows are on their own.
// One work-group; trunc[0 .. n) in LDS, m1[] scratch of n words.  A thread takes 16 tags per load
// and issues the loads of their leaders together (a Zipf batch is half shadows).
__device__ __forceinline__ void truncate_groups(uint32_t *table, uint8_t *state, const uint4 *reqc, const Undo *undo,
		const ChainLog *chain_logs, const uint2 *trunc, uint32_t n, uint32_t *m1, uint32_t lo_idx, uint32_t total,
		uint32_t *list, uint32_t *revived)
{
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x)
		m1[k] = trunc[k].x;
	__syncthreads();
	const uint32_t lo16 = lo_idx & ~15u;
	const uint4 *st4 = reinterpret_cast<const uint4 *>(state);
	for (uint32_t base = lo16 + threadIdx.x * 16u; base < total && base >= lo16; base += blockDim.x * 16u) {
		const uint4 v = st4[base >> 4];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
		uint32_t sh = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint32_t idx = base + (uint32_t)k;
			if (((w[k >> 2] >> ((k & 3) * 8)) & 0xffu) == (uint32_t)ST_SHADOW && idx >= lo_idx && idx < total)
				sh |= 1u << k;
		}
		if (!sh)
			continue;
		uint32_t lead[16];
#pragma unroll
		for (int k = 0; k < 16; k++)
			lead[k] = ((sh >> k) & 1u) ? undo[base + (uint32_t)k].word0 : kNone;
		uint32_t take = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (!((sh >> k) & 1u))
				continue;
			const uint32_t i = base + (uint32_t)k, c = lead[k];
			uint32_t cut = kNone;
			for (uint32_t q = 0; q < n; q++)
				if (trunc[q].x == c && trunc[q].y < cut)
					cut = trunc[q].y;
			if (cut == kNone)
				continue;
			if (i > cut) {
				MKV_TR(7u, i, c, cut);
				state[i] = ST_TAIL;
				take |= 1u << k;
			} else {
				for (uint32_t q = 0; q < n; q++)
					if (trunc[q].x == c)
						atomicMax(&m1[q], i);
			}
		}
		if (take) {
			uint32_t at = atomicAdd(revived, (uint32_t)__popc(take));
			if (list) {
#pragma unroll
				for (int k = 0; k < 16; k++)
					if ((take >> k) & 1u)
						list[at++] = base + (uint32_t)k;
			}
		}
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
		const uint32_t c = trunc[k].x;
		bool first = true;                          // (one lane per leader: the first entry that names it)
		uint32_t last = m1[k];
		for (uint32_t q = 0; q < n; q++)
			if (trunc[q].x == c) {
				first = first && q >= k;
				last = m1[q] > last ? m1[q] : last;
			}
		const uint32_t st = state[c] & 0x7fu;
		if (!first || !tag_logged(st))              // (not logged: the leader's commit is gone altogether)
			continue;
		MKV_TR(8u, c, last, st);
		table[group_loc_word(undo[c], st, chain_logs)] = reqc[last].z;
		const_cast<uint4 *>(reqc)[c].w = (reqc[c].w & 0xffu) | ((uint32_t)last << 8);
	}
	__syncthreads();
}

// What the tail launch leaves in the control block for the diagnostics and for the next pass: run
// by work-group 0 when everything is done -- or at once, when the rounds left nothing to do.
template <bool IDLE>     // (IDLE: the launch found nothing to do -- no phase has taken any time)
__device__ __forceinline__ void tail_epilogue(Ctl *ctl, const uint32_t *phase_ticks, uint32_t rounds_total,
		uint32_t total_all, uint32_t par, uint32_t serial, uint32_t undone, uint32_t viols, int first_chunk, int last_chunk,
		uint32_t *host_flag)
{
	const uint32_t t = threadIdx.x;
	if (t < 64) {
		// statistics shards: one wave sums them
		uint32_t c1 = ctl->committed[0][t].n, c2 = ctl->committed[1][t].n;      // (cleared by the next mark pass)
		for (int off = 32; off > 0; off >>= 1) {
			c1 += __shfl_down(c1, off);
			c2 += __shfl_down(c2, off);
		}
		uint32_t c3 = ctl->committed[2][t].n, c4 = ctl->committed[3][t].n;
		for (int off = 32; off > 0; off >>= 1) {
			c3 += __shfl_down(c3, off);
			c4 += __shfl_down(c4, off);
		}
		if (t == 0) {
			ctl->last.round1 = (first_chunk ? 0u : ctl->last.round1) + c1;
			ctl->last.round2 = (first_chunk ? 0u : ctl->last.round2) + c2;
			ctl->last.helped = (first_chunk ? 0u : ctl->last.helped) + c3;
			ctl->last.losers = (first_chunk ? 0u : ctl->last.losers) + c4;
			// word 3 of the host's flags: round 1 left round 2 more than a handful -- the host then gives round 2
			// a launch of its own (more lanes than this launch has) until a pass leaves few again
			const uint32_t many = c4 > kMergeLosersMax ? 1u : 0u;
			if (host_flag && many != ctl->hf_losers) {
				host_flag[3] = many;
				ctl->hf_losers = many;
			}
		}
	}
	if (t == 0) {
		// diagnostics of this call (summed over its chunks), then the control block of the next pass
		for (int k = 0; k < 16; k++)
			ctl->tail_phase[k] = (first_chunk ? 0u : ctl->tail_phase[k]) + ((IDLE || !phase_ticks) ? 0u : phase_ticks[k]);
		for (int k = 0; k < 8; k++) {                  // (diagnostic counts ride in the slots of the grid-wide rounds' ticks)
			ctl->tail_phase[8 + k] += ctl->dbg[k] * 100u;
			ctl->dbg[k] = 0;
		}
		ctl->last.total = total_all;
		ctl->last.tail_parallel = (first_chunk ? 0u : ctl->last.tail_parallel) + par;
		ctl->last.tail_serial = (first_chunk ? 0u : ctl->last.tail_serial) + serial;
		ctl->last.undone = (first_chunk ? 0u : ctl->last.undone) + undone;
		ctl->last.tail_rounds = (first_chunk ? 0u : ctl->last.tail_rounds) + rounds_total;
		const uint32_t fh = ctl->first_hard;
		ctl->last.first_hard = first_chunk ? fh : (fh < ctl->last.first_hard ? fh : ctl->last.first_hard);
		ctl->last.overflow = ctl->overflow;
		ctl->last.chains = (first_chunk ? 0u : ctl->last.chains) + ctl->chain_ok;
		// (the two "barrier" fields of the diagnostics now count: order violations handled, and hard
		// requests the grid-wide rounds left to the tail)
		ctl->last.bar_chain = (first_chunk ? 0u : ctl->last.bar_chain) + viols;
		ctl->last.bar_class = (first_chunk ? 0u : ctl->last.bar_class) + ctl->hard_tail;
		// word 0: chains ran late in this call (the next hinted batch is chunked finely: the chance of
		// an order violation grows with the square of the pass size); word 1: the 40-bit round stamps
		// are nearly used up -- the host clears the reservation tables before a later call
		// (ensure_ws); word 2: batches that exceeded the workspace and were dropped, so far.  Only
		// this kernel writes the words; the host only reads them.
		if (last_chunk)
			ctl->late_score = (ctl->last.bar_chain + ctl->last.bar_class) != 0u ? kLateSticky :
					(ctl->late_score ? ctl->late_score - 1u : 0u);
		if (last_chunk && host_flag) {
			// (a store to host memory makes the end of this kernel wait for the PCIe write: 4 us between
			// this launch and the next search, every step -- the words are written when they change)
			const uint32_t f0 = (ctl->last.bar_chain + ctl->last.bar_class) != 0u ? 1u : 0u;
			const uint32_t f1 = ctl->stamp_base > kStampMax - kStampReserve ? 1u : 0u;
			if (f0 != ctl->hf_last[0]) {
				host_flag[0] = f0;
				ctl->hf_last[0] = f0;
			}
			if (f1 != ctl->hf_last[1]) {
				host_flag[1] = f1;
				ctl->hf_last[1] = f1;
			}
			if (ctl->overflow)
				host_flag[2] = host_flag[2] + 1u;
		}
		ctl->hard_tail = 0;
		ctl->chain_n = 0;
		ctl->chain_ok = 0;
		ctl->any_loser = 0;
		ctl->any_tail = 0;
		ctl->any_hard = 0;
		ctl->first_hard = kNone;
		ctl->total = 0;
		// stamps: rounds 1 and 2, what the tail's rounds took, one spare
		const unsigned long long st = ctl->stamp_base + 3ull + rounds_total;
		ctl->stamp_base = st;
		ctl->next_stamp = st;
		ctl->carry_lo = kNone;
	}
}

// ---- the tail launch ----------------------------------------------------------------------
// Finishes what rounds 1 and 2 left: losers of both rounds and hard requests whose chain
// could not run beside the other requests.  All work-groups share the scans when there is
// much to scan (gather) and the rounds when much is left (grid_rounds);
// otherwise the helpers leave at once and work-group 0 does the rest: reservation rounds,
// then the in-order replay per closed region.  A chain that meets an order violation makes
// the launch roll back the later commits of that region and take them on as well.
template <Op OP>
__global__ __launch_bounds__(kTailThreads) void k_tail(Ctl *ctl, DevGeom g, uint32_t *table,
		unsigned long long *owner, unsigned long long *altmin, unsigned long long *cnt, const unsigned long long *owner1,
		const unsigned long long *owner2, const unsigned long long *second, const uint4 *reqc, uint8_t *state, const Undo *undo,
		const ChainLog *chain_logs,
		uint32_t *glist, int max_rounds,
		uint32_t lo_idx, uint32_t hi_idx, int first_chunk, int last_chunk, uint32_t *host_flag, uint32_t grid_min, uint32_t grid_progress,
		int dbg_path, int round2_hint, uint32_t req_cap)
{
	// (round2_hint >= 0: ROUND 2 runs here, as the launch's first phase, with that record_hint -- a launch that
	// finds nothing to do costs 4.6 - 5 us in this chain, and since a bucket's second registrant helps itself
	// in round 1 the usual batch leaves round 2 nothing: what is left of its losers takes the lanes of this
	// launch and a meeting of its work-groups instead of a launch of its own.  A host that announces a pass
	// of very many requests launches round 2 by itself as before: it wants more lanes than these.)
	__shared__ __align__(16) uint32_t lkeys[kSortCap];
	__shared__ __align__(16) uint8_t lreg[kSortCap];    // region of a listed request (in-order replay); in-group rounds: the duplicates' election
	static_assert(2u * kElectSlots * sizeof(uint32_t) <= (size_t)kSortCap, "the election table lives in lreg");
	__shared__ uint32_t s_thr[kMaxRegions];     // roll-back thresholds when a violation takes a whole region suffix
	__shared__ uint32_t s_dflag, s_nsel, s_ntrunc;
	__shared__ uint32_t s_loose, s_revived;
	__shared__ uint2 s_trunc[kTruncCap];        // groups of duplicates to be cut (select_from, roll_back_from)
	__shared__ uint32_t s_tm1[kTruncCap];
	__shared__ uint32_t s_rbar[kMaxRegions];    // barrier of the current in-group round
	__shared__ uint32_t s_rmin[kMaxRegions];    // lowest pending index per region
	__shared__ uint32_t s_below[kMaxRegions];   // requests before the round's barrier that the round has left pending
	__shared__ uint32_t s_n, s_progress, s_undone, s_par, s_serial, s_left, s_dupw, s_done12, s_viols;
	const uint32_t t = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
	// (time per phase: diagnostics builds only, make DIAG=1 -- sixteen registers held for the whole
	// launch are more than this kernel has to spare)
#ifdef MKV_DIAGNOSTICS
	unsigned long long tick = wall_clock64();
	uint32_t phase_ticks[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#define MKV_PHASE(k) do { const unsigned long long now_ = wall_clock64(); phase_ticks[k] += (uint32_t)(now_ - tick); tick = now_; } while (0)
	uint32_t *const grid_ticks = phase_ticks + 8;
#else
	uint32_t *const phase_ticks = nullptr, *const grid_ticks = nullptr;
#define MKV_PHASE(k) ((void)0)
#endif
	static_assert(kTailThreads == 2 * (int)kTlGroups, "a lane of the tail's first work-group per round and work-group of the rounds");
	const uint32_t my_round = t < kTlGroups ? 0u : 1u, my_wg = t & (kTlGroups - 1u);
	const uint32_t total_all = ctl->total;
	if (round2_hint >= 0 && OP != OP_DELETE && ctl->any_loser != 0u) {
		// (every work-group reads the same flag: round 1, the kernel before, set it)
		static_assert(RoundLds<2, kTailThreads>::bytes <= sizeof(lkeys), "round 2's LDS in the work list's");
		round_body<OP, 2, kTailThreads>(wg, nwg, reinterpret_cast<unsigned char *>(lkeys), ctl, g, table, owner2,
				const_cast<unsigned long long *>(owner1), reqc, state, const_cast<Undo *>(undo), const_cast<ChainLog *>(chain_logs),
				lo_idx, hi_idx, req_cap, round2_hint, cnt + ((size_t)g.hash_mask + 1u), second);
		grid_sync(&ctl->sync_r2, nwg);
	}
	// The rounds left nothing (the usual batch, three passes in four): no lists, no scans, no meetings --
	// every work-group reads the one flag (final: only kernels before this one in the stream -- or the
	// phase above, behind its meeting -- set it) and work-group 0 leaves the control block ready for the
	// next pass.
	const uint32_t p_any_tail = ld_agent32(&ctl->any_tail);
	if (p_any_tail == 0u) {
		if (wg == 0)
			tail_epilogue<true>(ctl, nullptr, 0u, total_all, 0u, 0u, 0u, 0u, first_chunk, last_chunk, host_flag);
		return;
	}
	const uint32_t total = hi_idx < total_all ? hi_idx : total_all;   // this chunk ends here
	const uint32_t carry_in = ctl->carry_lo;
	// what the grid-wide rounds left, listed by themselves (tail_list_add) -- unless a work-group left
	// more than its place holds, shadows without a leader are about (not listed), or the call runs in
	// several passes.  Every work-group of this launch reads the counts and decides alike.
	__shared__ uint32_t s_tl_total, s_tl_over;
	if (t == 0) {
		s_tl_total = 0u;
		s_tl_over = 0u;
	}
	__syncthreads();
	const bool list_ok = first_chunk && last_chunk && carry_in >= lo_idx && ctl->loose == 0u && p_any_tail != 0u;
	const uint32_t tl_g1 = list_ok ? ctl->tl_groups[0] : 0u;
	const uint32_t tl_g2 = list_ok && ctl->any_loser != 0u ? ctl->tl_groups[1] : 0u;    // (round 2 leaves at once without losers)
	uint32_t my_cnt = 0, my_base = 0;
	if (list_ok) {
		if (tl_g1 > kTlGroups || tl_g2 > kTlGroups) {
			s_tl_over = 1u;
		} else if (my_wg < (my_round ? tl_g2 : tl_g1)) {
			my_cnt = ctl->tl_cnt[my_round][my_wg];
			if (my_cnt > kTlPer)
				s_tl_over = 1u;
			else if (my_cnt)
				my_base = atomicAdd(&s_tl_total, my_cnt);
		}
	}
	__syncthreads();
	const bool listed = list_ok && s_tl_over == 0u;
	const uint32_t tln = s_tl_total;
	if (carry_in < lo_idx)
		lo_idx = carry_in;                        // this pass also covered the carried requests
	const bool work_left = p_any_tail != 0u;      // round 2 has retagged every LOSER
	const unsigned long long stamp1 = ctl->stamp_base;    // round-1 stamp of this pass
	if (t == 0) {
		s_n = 0;
		s_undone = 0;
		s_par = 0;
		s_serial = 0;
		s_viols = 0;
		s_loose = 0;
		s_revived = 0;
	}
	// The scans over the request tags can be shared by all work-groups of the launch;
	// everything after them is work-group 0's (unless much is left).  The helpers only stay when
	// that pays -- judged by what every work-group reads alike: the rounds' flags and commit
	// counters (final: only the kernels before this one in the stream change them).
	if (t < 64u) {
		uint32_t c = ctl->committed[0][t].n + ctl->committed[1][t].n;
		for (int off = 32; off > 0; off >>= 1)
			c += __shfl_down(c, off);
		if (t == 0)
			s_done12 = c;
	}
	__syncthreads();
	const uint32_t span = total > lo_idx ? total - lo_idx : 0u;
	const uint32_t pending_est = span > s_done12 ? span - s_done12 : 0u;
	const bool use_helpers = !listed && nwg > 1u && (pending_est > (uint32_t)kTailAloneMax || span > (uint32_t)kTailAloneSpan);
			// (a long tag range is worth sharing even if little is left in it)
	if (wg != 0 && (!work_left || !use_helpers))
		return;
	const uint32_t nwg_eff = use_helpers ? nwg : 1u;     // work-groups taking part from here on
	bool grid_mode = false;
	bool lds_ready = false;                   // the first work list is in lkeys already
	if (work_left && listed) {
		MKV_PHASE(0);
		// (the list goes into LDS as well: the work list of a short list is read from there at once)
		for (uint32_t k = 0; k < my_cnt; k++) {
			const uint32_t e = ctl->tl[my_round][my_wg][k];
			glist[my_base + k] = e;
			lkeys[my_base + k] = e;
		}
		if (t == 0)
			s_n = tln;
		lds_ready = true;
		MKV_PHASE(6);
		wg_sync();
	} else if (work_left) {
		MKV_PHASE(0);
		gather_pending(state, undo, reqc, glist, &ctl->glist_n, lo_idx, total, wg, nwg_eff, ctl->loose != 0u);
		MKV_PHASE(6);
		grid_sync(&ctl->sync[1], nwg_eff);
		// a launch left with a long list keeps every work-group for its rounds
		const uint32_t m_all = ld_agent32(&ctl->glist_n);
		grid_mode = max_rounds > 0 && nwg_eff > 1u && m_all > grid_min;
		if (wg != 0 && !grid_mode)
			return;
		MKV_PHASE(7);
		if (t == 0)
			s_n = m_all;
		__syncthreads();
	}
	MKV_PHASE(1);
	Tv tv;
	tv.owner1 = owner1;
	tv.owner2 = owner2;
	tv.second = second;
	tv.stamp1 = stamp1;
	tv.state = state;
	tv.reqc = reqc;
	tv.undo = undo;
	tv.chain_logs = chain_logs;
	tv.ctl = ctl;
	uint32_t rounds_total = 0;
	while (work_left) {
		const uint32_t m = s_n;
		if (m == 0)
			break;
		uint32_t used = 0, left = 0;
		bool in_lds = false;
		uint32_t n2 = 1;
		if (grid_mode) {
			uint32_t par = 0;
			Fold fold;                             // (the work list of grid-wide rounds is the global one: its LDS is free)
			fold.tb = lkeys;
			fold.tc = lkeys + kFoldSlots;
			fold.tk = reinterpret_cast<unsigned long long *>(lkeys + 2u * kFoldSlots);
			static_assert(4u * kFoldSlots <= (uint32_t)kSortCap, "fold tables in the LDS of the work list");
			left = grid_rounds<OP>(ctl, g, table, owner, cnt, altmin, reqc, state, glist, m, s_rbar, s_rmin, s_below, &s_progress,
					&s_left, &s_dupw, kGridRoundsMax, &used, &par, grid_ticks, tv, rounds_total, fold, grid_progress);
			if (wg != 0)
				return;
			grid_mode = false;                     // (later turns of this loop are work-group 0's alone)
			if (t == 0)
				s_par += par;
			__syncthreads();
			if (left != 0 && left <= kGridHandover && max_rounds > 0 && ld_agent32(&ctl->tflags) == 0u) {
				// handed over: what is left goes into the LDS list and through the in-group rounds
				if (t == 0)
					s_left = 0;
				__syncthreads();
				for (uint32_t i = t; i < m; i += blockDim.x) {
					const uint32_t k = glist[i];
					if (k != kNone)
						lkeys[atomicAdd(&s_left, 1u)] = k;
				}
				__syncthreads();
				const uint32_t m2 = s_left;
				while (n2 < m2)
					n2 <<= 1;
				for (uint32_t i = m2 + t; i < n2; i += blockDim.x)
					lkeys[i] = kNone;
				in_lds = true;
				rounds_total += used;
				used = 0;
				wg_sync();
				left = tail_rounds<OP>(ctl, g, table, owner, cnt, altmin, reqc, state, lkeys, m2, s_rbar, s_rmin, s_below, &s_progress,
						&s_par, &s_left, &s_dupw, max_rounds, &used, tv, rounds_total, reinterpret_cast<uint32_t *>(lreg));
			}
		} else {
			// work list: LDS when it fits, else the global list it was gathered into
			in_lds = m <= (uint32_t)kSortCap;
			uint32_t *keys = in_lds ? lkeys : glist;
			while (n2 < m)
				n2 <<= 1;
			if (in_lds) {
				if (!lds_ready)
					for (uint32_t i = t; i < m; i += blockDim.x)
						lkeys[i] = glist[i];
				for (uint32_t i = m + t; i < n2; i += blockDim.x)
					lkeys[i] = kNone;
			}
			lds_ready = false;
			wg_sync();
			// a handful of requests: the in-order replay is done before a round would be
			left = m;
			if (m > (uint32_t)kTailReplayMax && max_rounds > 0)
				left = tail_rounds<OP>(ctl, g, table, owner, cnt, altmin, reqc, state, keys, m, s_rbar, s_rmin, s_below, &s_progress,
						&s_par, &s_left, &s_dupw, max_rounds, &used, tv, rounds_total, reinterpret_cast<uint32_t *>(lreg));
		}
		rounds_total += used;
		MKV_PHASE(4);
		if (left != 0 && ld_agent32(&ctl->tflags) == 0u) {
			// what the rounds left: canonical order, one request at a time per closed region
			if (!in_lds) {
				// (the global list, finished entries marked kNone): into LDS if it fits
				if (t == 0)
					s_left = 0;
				__syncthreads();
				for (uint32_t i = t; i < m; i += blockDim.x) {
					const uint32_t k = glist[i];
					if (k != kNone) {
						const uint32_t at = atomicAdd(&s_left, 1u);
						if (at < (uint32_t)kSortCap)
							lkeys[at] = k;
					}
				}
				__syncthreads();
				left = s_left;
				n2 = 1;
				while (n2 < left)
					n2 <<= 1;
				if (left <= (uint32_t)kSortCap)
					for (uint32_t i = left + t; i < n2; i += blockDim.x)
						lkeys[i] = kNone;
				__syncthreads();
			}
			if (left <= (uint32_t)kSortCap) {
				lds_sort(lkeys, n2);
				replay_regions<OP>(table, g, reqc, state, lkeys, lreg, left, tv, &s_serial);
			} else {
				replay_pending_scan<OP>(table, g, reqc, state, glist, lkeys, lo_idx, total, tv, &s_serial);
			}
		}
		wg_sync();
		MKV_PHASE(5);
		const uint32_t flags = ld_agent32(&ctl->tflags);
		if (flags == 0u)
			break;                                 // everything is done
		// -- a chain met an order violation -----------------------------------------------------------
		uint32_t relisted = kNone;                 // entries of the next work list already in glist (kNone: gather anew)
		if (OP != OP_DELETE && (flags & kTvViolation)) {
			uint32_t nseeds = ld_agent32(&ctl->vlist_n);
			nseeds = nseeds < (uint32_t)kViolCap ? nseeds : (uint32_t)kViolCap;
			if (in_lds) {
				// The next work list is what this one has left plus what is taken back below -- no
				// need to scan the tags of the whole pass for it.  The first part goes to the global
				// list now: the LDS of the work list is about to hold the stacks.
				if (t == 0)
					s_left = 0;
				__syncthreads();
				for (uint32_t i = t; i < n2; i += blockDim.x) {
					const uint32_t k = lkeys[i];
					if (k != kNone && tag_pending(state[k]))
						glist[atomicAdd(&s_left, 1u)] = k;
				}
				__syncthreads();
				relisted = s_left;
			}
			// one lane per violation selects what has to go (select_from); the work list's LDS holds
			// the lanes' stacks and the list of what they selected (it is gathered anew afterwards)
			static_assert((uint32_t)kViolCap * kLaneStack * 2u + kSelCap <= (uint32_t)kSortCap, "LDS of the work list");
			uint32_t *sel = lkeys + (uint32_t)kViolCap * kLaneStack * 2u;
			if (t == 0) {
				s_ntrunc = 0u;
				s_dflag = dbg_path == 2 ? 1u : 0u;      // (tests: always the region-suffix fall-back)
				s_nsel = 0;
				s_viols += nseeds;
			}
			__syncthreads();
			if (t < nseeds && dbg_path != 2) {
				const uint32_t *sw = reinterpret_cast<const uint32_t *>(ctl->vlist + opaque(t));
				const uint32_t sj = ld_agent32(sw), sX = ld_agent32(sw + 1);
				if (!select_from(g, state, reqc, undo, chain_logs, tv, sj, sX,
						reinterpret_cast<uint2 *>(lkeys) + opaque(t) * kLaneStack, sel, &s_nsel, s_trunc, &s_ntrunc))
					s_dflag = 1u;
			}
			wg_sync();
			uint32_t u = 0;
			if (!s_dflag) {
				// take them back, newest round first (two commits of one round never share a bucket)
				const uint32_t nsel = s_nsel;
				for (int pass = 0; pass < 2; pass++) {
					const uint32_t tag = (pass == 0 ? ST_ROUND2 : ST_ROUND1) | ST_SELECT;
					const uint32_t ctag = (pass == 0 ? ST_CHAIN2 : ST_CHAIN1) | ST_SELECT;
					for (uint32_t i = t; i < nsel; i += blockDim.x) {
						const uint32_t k = sel[i];
						const uint32_t st = state[k];
						if (st != tag && st != ctag)
							continue;
						take_back(g, table, state, reqc, undo, chain_logs, k, st == ctag);
						if (reqc[k].w & kLeadBit) {         // its shadows are requests of their own again
							const uint32_t at = atomicAdd(&s_ntrunc, 1u);
							if (at < kTruncCap)
								s_trunc[at] = make_uint2(k, k);
						}
						u++;
					}
					wg_sync();
				}
				if (relisted != kNone) {
					for (uint32_t i = t; i < nsel; i += blockDim.x)
						glist[relisted + i] = sel[i];
					relisted += nsel;
				}
			} else {
				relisted = kNone;
				// a stack or the list overflowed: the whole suffix of the regions concerned
				// (roll_back_from also takes back what is flagged already)
				if (t < kMaxRegions)
					s_thr[t] = t < g.nregions ? ld_agent32(&ctl->viol[opaque(t)]) : kNone;
				__syncthreads();
				if (t == 0)
					s_ntrunc = 0u;                      // (what select_from noted is found again by index)
				__syncthreads();
				u = roll_back_from(g, table, state, reqc, undo, chain_logs, s_thr, lo_idx, total, 0u, 1u, nullptr, s_trunc,
						&s_ntrunc);
			}
			// groups of duplicates that begin below a violation's index and reach beyond it: cut there
			__syncthreads();
			{
				const uint32_t ntr = s_ntrunc < kTruncCap ? s_ntrunc : kTruncCap;
				if (t == 0) {
					s_loose = s_ntrunc > kTruncCap ? 1u : 0u;   // (not all noted: the gather looks at every shadow's leader)
					s_revived = 0u;
				}
				__syncthreads();
				if (ntr) {
					// (their members beyond the cut are pending again: appended to the next work list)
					truncate_groups(table, state, reqc, undo, chain_logs, s_trunc, ntr, s_tm1, lo_idx, total,
							relisted != kNone ? glist + relisted : nullptr, &s_revived);
					if (relisted != kNone)
						relisted += s_revived;
				}
				if (s_loose)
					relisted = kNone;
			}
			if (u)
				atomicAdd(&s_undone, u);
		}
		wg_sync();
		MKV_PHASE(2);
		if (t == 0)
			s_n = 0;
		if (t < kMaxRegions)
			atomicExch(&ctl->viol[opaque(t)], kNone);
		if (t == 0) {
			atomicExch(&ctl->tflags, 0u);
			atomicExch(&ctl->vlist_n, 0u);
		}
		__syncthreads();
		if (relisted == kNone)
			gather_pending(state, undo, reqc, glist, &s_n, lo_idx, total, 0u, 1u, s_loose != 0u);
		else if (t == 0)
			s_n = relisted;
		wg_sync();
		MKV_PHASE(3);
	}
	__syncthreads();
	tail_epilogue<false>(ctl, phase_ticks, rounds_total, total_all, s_par, s_serial, s_undone, s_viols, first_chunk, last_chunk,
			host_flag);
}

__global__ void k_init_ctl(Ctl *ctl, unsigned long long first_stamp, uint32_t *host_flag)
{
	const uint32_t t = threadIdx.x;
	if (t == 0) {
		if (host_flag)
			host_flag[0] = host_flag[1] = host_flag[3] = 0;     // (in step with hf_last / hf_losers below)
		ctl->stamp_base = first_stamp;
		ctl->next_stamp = first_stamp + kStampsPerCall;
		ctl->total = 0;
		ctl->overflow = 0;
		ctl->any_loser = ctl->any_tail = 0;
		for (int k = 0; k < kShards; k++)
			ctl->committed[0][k].n = ctl->committed[1][k].n = ctl->committed[2][k].n = ctl->committed[3][k].n = 0;
		ctl->first_hard = kNone;
		ctl->last = Info{0, 0, 0, 0, 0, 0, kNone, 0, 0, 0, 0, 0, 0, 0};
		ctl->bar_chain = ctl->bar_class = 0;
		ctl->carry_lo = kNone;
		ctl->chain_n = 0;
		ctl->chain_ok = 0;
		ctl->any_hard = 0;
		ctl->violations = 0;
		ctl->hard_tail = 0;
		ctl->late_score = 0;
		ctl->hf_last[0] = ctl->hf_last[1] = 0;
		ctl->hf_losers = 0;
		for (int k = 0; k < 8; k++)
			ctl->dbg[k] = 0;
		reset_tail_scratch(ctl);
	}
	if (t < kMaxRegions)
		ctl->barrier[t] = kNone;
}

struct Chunking {
	int passes;         // data chunks + flush passes (when more than one data chunk)
	uint32_t step;      // canonical indices per data chunk
	int flush;          // trailing passes over an empty range (carried requests only)
};

// ---- per-stream workspace -----------------------------------------------------
struct Workspace {
	int device = -1;
	uint64_t nbuckets = 0;
	int64_t req_cap = 0;
	// reservation tables, 8 bytes per bucket each: [0] round 1 (and every chain's claims), [1] round 2 --
	// left as those rounds wrote them until the pass is over: the tail reads from them who can have
	// been committed where (later_commit_on) --, [2] the tail's rounds, [3] the tail's "lowest
	// registrant that has the bucket as its alternate"
	unsigned long long *owner[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // ([4]: the mark pass's second-lowest registrants)
	unsigned long long *cnt = nullptr;
	uint8_t *state = nullptr;
	uint4 *reqc = nullptr;
	Undo *undo = nullptr;
	uint32_t *glist = nullptr;
	ChainLog *chain_logs = nullptr;        // kChainLogCap logs of the grid-wide rounds' chains, then one per lane
	                                       // of the tail launch (kTailLanes)
	Ctl *ctl = nullptr;
	// mkv_cycle only: side stream on which the table-independent mark pass of the
	// insert overlaps the search kernel, and the fork/join events
	hipStream_t aux = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// Three words of pinned host memory that only the tail kernel writes (the host never does):
	//   [0] 1 = eviction chains ran late in the call (read without synchronising when the next
	//       call on this workspace is set up -- possibly a call late, which is fine: it only steers
	//       how finely the next hinted batch is chunked, never the result);
	//   [1] 1 = the 40-bit round stamps are nearly used up (ensure_ws clears the reservation tables);
	//   [2] number of batches dropped so far because they exceeded the workspace (report_dropped).
	volatile uint32_t *host_flag = nullptr;
	uint32_t *host_flag_dev = nullptr;
	uint32_t dropped_seen = 0;         // value of word [2] already reported
	int reset_cooldown = 0;            // calls for which word [1] is ignored after a stamp reset was enqueued
	Chunking cycle_ck = {1, 65536, 0};    // chunking chosen by the first half of an mkv_cycle insert
	int barrier_score = 0;             // > 0: eviction chains ran late within the last few calls
	int stamp_resets = 0;              // times the reservation tables were cleared because the stamps ran low
};

static std::mutex g_ws_mutex;
// ONE pair of workspaces per device, shared by every stream and every table of the process
// (key: device << 1 | which): [0] serves insert / delete calls, [1] the insert half of a cycle
// (mkv_cycle, mkv_slots_cycle), whose mark pass rides in the search launch while a delete of the
// same cycle may still be using [0].  The reference's scheduler drives up to 16 worker streams
// over one table (src/mega_scheduler.c:273-280): a workspace per stream would cost it 16 x 0.9
// GiB for nothing, since updates are put in issue order across streams anyway (order_updates).
static std::unordered_map<uint64_t, Workspace *> g_ws;
static unsigned long long g_dropped_total = 0;   // batches dropped because they exceeded a workspace

static void free_ws(Workspace *w)
{
	for (int i = 0; i < 5; i++)
		if (w->owner[i]) (void)hipFree(w->owner[i]);
	if (w->cnt) (void)hipFree(w->cnt);
	if (w->state) (void)hipFree(w->state);
	if (w->reqc) (void)hipFree(w->reqc);
	if (w->undo) (void)hipFree(w->undo);
	if (w->glist) (void)hipFree(w->glist);
	if (w->chain_logs) (void)hipFree(w->chain_logs);
	if (w->ctl) (void)hipFree(w->ctl);
	if (w->host_flag) (void)hipHostFree((void *)w->host_flag);
	if (w->aux) (void)hipStreamDestroy(w->aux);
	if (w->ev_fork) (void)hipEventDestroy(w->ev_fork);
	if (w->ev_join) (void)hipEventDestroy(w->ev_join);
	*w = Workspace();
}

constexpr uint64_t kCycleSalt = 1ull;   // the device's second workspace (insert half of a cycle)

// An earlier insert/delete on this workspace carried more requests than it holds: the counts are
// device memory, so the library only learns this after the fact, and that batch was NOT applied.
// Nothing can be returned through the reference's void ABI (and that call has long returned), and
// killing a key-value server over it would be worse than the loss: it is reported on stderr, once
// per dropped batch, counted (mkv_dropped_batches), and the current call goes on normally.
static void report_dropped(Workspace *w)
{
	if (!w->host_flag)
		return;
	const uint32_t now = w->host_flag[2];
	if (now == w->dropped_seen)
		return;
	g_dropped_total += (uint32_t)(now - w->dropped_seen);
	w->dropped_seen = now;
	fprintf(stderr, "libgpuhash (MI355X): an insert/delete call carried more requests than the update "
			"workspace holds (%lld); that batch was dropped (%llu so far).  Size the workspace with "
			"mkv_reserve(geom, max_requests, stream), pass the batch size to mkv_insert_hinted, or set "
			"MKV_REQ_CAP (limit %lld requests per call).\n", (long long)w->req_cap, g_dropped_total,
			(long long)kMaxReqCap);
}

static int ensure_ws(Workspace **out, uint64_t nbuckets, int64_t req_cap, hipStream_t stream,
		uint64_t salt = 0)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	const uint64_t key = ((uint64_t)dev << 1) | (salt ? 1ull : 0ull);
	std::lock_guard<std::mutex> lock(g_ws_mutex);
	Workspace *&w = g_ws[key];
	if (!w)
		w = new Workspace();
	report_dropped(w);
	if (w->device == dev && w->nbuckets >= nbuckets && w->req_cap >= req_cap) {
		if (w->reset_cooldown > 0) {
			w->reset_cooldown--;
		} else if (w->host_flag && w->host_flag[1]) {
			// round stamps nearly used up (tail kernel's flag, read a call or two late: the
			// reserve covers that): start over with cleared reservation tables, in stream order.
			// The flag is the kernel's to write; it reads 0 again once a call has run after the
			// reset, until then it is not looked at.
			bool ok = true;
			for (int i = 0; i < 5; i++)
				ok = ok && hipMemsetAsync(w->owner[i], 0xff, w->nbuckets * 8, stream) == hipSuccess;
			ok = ok && hipMemsetAsync(w->cnt, 0, w->nbuckets * 16, stream) == hipSuccess;
			if (!ok)
				return MKV_EHIP;
			hipLaunchKernelGGL(k_init_ctl, dim3(1), dim3(64), 0, stream, w->ctl, 1ull, w->host_flag_dev);
			w->reset_cooldown = 64;
			w->stamp_resets++;
		}
		*out = w;
		return MKV_OK;
	}
	const uint64_t nb = nbuckets > w->nbuckets ? nbuckets : w->nbuckets;
	const int64_t rc = req_cap > w->req_cap ? req_cap : w->req_cap;
	// growing: finish what the old buffers may still be used by (any stream of the device)
	if (w->device >= 0)
		(void)hipDeviceSynchronize();
	const uint32_t dropped_seen = w->dropped_seen;
	free_ws(w);
	w->device = dev;
	w->nbuckets = nb;
	w->req_cap = rc;
	bool ok = true;
	for (int i = 0; i < 5 && ok; i++)
		ok = ok && hipMalloc(&w->owner[i], nb * 8) == hipSuccess;
	ok = ok && hipMalloc(&w->cnt, nb * 16) == hipSuccess;     // registrant counts, then the election words
	ok = ok && hipMalloc(&w->state, ((size_t)rc + 15) & ~(size_t)15) == hipSuccess;
	ok = ok && hipMalloc(&w->reqc, (size_t)rc * sizeof(uint4)) == hipSuccess;
	ok = ok && hipMalloc(&w->undo, (size_t)rc * sizeof(Undo)) == hipSuccess;
	ok = ok && hipMalloc(&w->glist, (size_t)rc * 4) == hipSuccess;
	ok = ok && hipMalloc(&w->chain_logs, ((size_t)kChainLogCap + kTailLanes) * sizeof(ChainLog)) == hipSuccess;
	ok = ok && hipMalloc(&w->ctl, sizeof(Ctl)) == hipSuccess;
	if (ok) {
		void *hf = nullptr;
		ok = hipHostMalloc(&hf, 64, hipHostMallocMapped) == hipSuccess;
		if (ok) {
			w->host_flag = reinterpret_cast<volatile uint32_t *>(hf);
			w->host_flag[0] = w->host_flag[1] = 0;
			w->host_flag[2] = dropped_seen;          // (the count goes on over a regrown workspace)
			w->dropped_seen = dropped_seen;
			void *dp = nullptr;
			ok = hipHostGetDevicePointer(&dp, hf, 0) == hipSuccess;
			w->host_flag_dev = reinterpret_cast<uint32_t *>(dp);
		}
	}
	if (ok) {
		// owner entries start at "no owner" (all ones); counters at zero
		for (int i = 0; i < 5; i++)
			ok = ok && hipMemsetAsync(w->owner[i], 0xff, nb * 8, stream) == hipSuccess;
		ok = ok && hipMemsetAsync(w->cnt, 0, nb * 16, stream) == hipSuccess;
		if (ok)
			hipLaunchKernelGGL(k_init_ctl, dim3(1), dim3(64), 0, stream, w->ctl,
					(unsigned long long)(tuning().debug_first_stamp > 0 ? tuning().debug_first_stamp : 1), w->host_flag_dev);
		// the buffers are shared by all streams: they are ready before any other stream uses them
		ok = ok && hipStreamSynchronize(stream) == hipSuccess;
	}
	if (ok && salt == kCycleSalt) {
		ok = ok && hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) {
		free_ws(w);
		return MKV_ENOMEM;
	}
	*out = w;
	return MKV_OK;
}

// ---- several streams ----------------------------------------------------------------------
// The reference's scheduler issues the deletes and inserts of all its workers on ONE table,
// a stream per worker (src/mega_scheduler.c:393-500); its kernels then race benignly.  Here an
// update keeps its reservation state in the device's workspace, and a roll-back restores words
// -- two updates running side by side would not be anybody's serialisation.  So the updates of
// a device are put in the order the host issued them: an update on another stream than the last
// one waits for it.  No handle of the earlier stream is touched for that (it may have been
// destroyed since): while updates have been changing streams lately, every update call records
// an event of the library's own behind its last kernel (updates_issued) and the next stream
// waits for that; a change of stream that finds no such event -- the first one, or one after a
// long run on one stream -- synchronises the device instead.  A host that keeps to one stream
// pays nothing: an event behind every call costs 5.7 us per call on this runtime (the marker
// packet holds up the next launch; kernel timeline of bench.py, whose pre-load runs on another
// stream than its steps), so recording stops kUpdateSwitchMemory calls after the last change.
// Searches are not ordered against updates of other streams (neither are the reference's).
// Not called by mkv_slots_cycle: the sharded engines order their local phases themselves.
constexpr int kUpdateSwitchMemory = 4;
struct UpdateOrder {
	hipStream_t last = nullptr;
	bool any = false;
	bool ev_valid = false;          // `ev` was recorded behind the last update issued
	int active = 0;                 // update calls for which an event is still recorded behind each
	hipEvent_t ev = nullptr;
};
static std::unordered_map<int, UpdateOrder> g_order;   // key: device

// An update call enqueues several kernels that share the device's workspace, between order_updates
// and updates_issued.  Host threads that update concurrently (each on its own stream) take turns
// for that stretch -- microseconds of enqueueing; the kernels themselves are ordered by the event.
// (The reference's scheduler issues from one thread, src/mega_scheduler.c:306-532; recursive: the
// cycle calls the delete entry point.)
static std::recursive_mutex g_call_mutex;
struct CallGuard {
	std::lock_guard<std::recursive_mutex> lock{g_call_mutex};
};
}  // namespace mkv

// mkv_cycle (mkv_abi.hip) holds the turn over ALL its parts: its fused search launch fills the
// device's cycle workspace and its rounds, enqueued two calls later, run over it -- another thread's
// cycle must not get in between (nor grow the workspace under it).
extern "C" void mkv_call_lock(void)
{
	mkv::g_call_mutex.lock();
}

extern "C" void mkv_call_unlock(void)
{
	mkv::g_call_mutex.unlock();
}

namespace mkv {

static int order_updates(hipStream_t stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	std::lock_guard<std::mutex> lock(g_ws_mutex);
	UpdateOrder &o = g_order[dev];
	if (o.any && o.last != stream) {
		if (!o.ev && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess)
			return MKV_EHIP;
		if (o.ev_valid) {
			if (hipStreamWaitEvent(stream, o.ev, 0) != hipSuccess)
				return MKV_EHIP;
		} else if (hipDeviceSynchronize() != hipSuccess) {
			return MKV_EHIP;
		}
		o.active = kUpdateSwitchMemory;
	} else if (o.active > 0) {
		o.active--;
	}
	o.last = stream;
	o.any = true;
	o.ev_valid = false;             // (what is enqueued from here on lies behind any event recorded so far)
	return MKV_OK;
}

// behind the last kernel of an update call
static int updates_issued(hipStream_t stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	std::lock_guard<std::mutex> lock(g_ws_mutex);
	UpdateOrder &o = g_order[dev];
	if (o.active > 0 && o.ev) {
		if (hipEventRecord(o.ev, stream) != hipSuccess)
			return MKV_EHIP;
		o.ev_valid = true;
	}
	return MKV_OK;
}

static int64_t ws_need(const ReqSrc &src, long long hint = 0)
{
	int64_t need = (int64_t)tuning().req_cap;
	if (src.single_n > need)
		need = src.single_n;
	if (hint > need)
		need = hint;                   // the caller's upper bound on the batch
	if (need > kMaxReqCap)
		need = kMaxReqCap;
	return need;
}

// K1 alone: touches the requests and the workspace, never the table.
template <Op OP>
static void launch_mark(Workspace *w, const DevGeom &g, const ReqSrc &src, hipStream_t stream,
		const uint32_t *table, uint32_t lo = 0, uint32_t hi = kNone)
{
	hipLaunchKernelGGL((k_mark<OP>), dim3((unsigned)tuning().mark_alone_blocks), dim3(256), 0, stream, w->ctl, src, g,
			w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, lo, hi, table, tuning().lazy_alt);
}

template <Op OP>
static int launch_rounds(Workspace *w, const DevGeom &g, uint32_t *table, hipStream_t stream,
		uint32_t lo = 0, uint32_t hi = kNone, int first_chunk = 1, int last_chunk = 1, hipEvent_t done_ev = nullptr,
		long long hint = 0);

// Canonical-index passes of one call.  With a size hint from the caller a large batch runs
// pass after pass (stream order = canonical order), Tuning::insert_chunk requests each; no
// hint (the plain ABI, whose callers stay below 32 K requests): one pass over everything.
// Every pass finishes its own requests.  Two-choice insert and delete: always one pass.
//
// The tail kernel leaves a word in pinned host memory (bit 0: eviction chains had to run
// late in that call).  It is read here without synchronising -- possibly a call late, which
// is fine: with calm_factor > 1 it steers the pass size (larger passes while no chain ran late
// within the last Tuning::barrier_sticky calls) -- never the result.
// work-groups of the mark pass inside a fused search launch: Tuning::mark_blocks for the usual
// few per cent of SETs, more when the caller announces many (one per 512 requests, up to 1024)
// (busy: eviction chains have been running late lately, chains_lately -- a well-filled table: most
// requests register on both their buckets and many walk ahead, and the mark pass, whose registrations
// wait for their atomics to return, is what the fused launch ends with: a lane per announced request
// then -- load factor 0.85: launch 40 -> 34 us; at load factor 0.2 the extra work-groups cost the
// search half a microsecond)
static uint32_t mark_blocks_for(long long hint, bool busy = false)
{
	long long b = tuning().mark_blocks;
	if (hint / 512 > b)
		b = hint / 512;
	if (busy && (hint + 255) / 256 > b)
		b = (hint + 255) / 256;
	return (uint32_t)(b > 1024 ? 1024 : b);
}

static bool chains_lately(Workspace *w)
{
	if (w && w->host_flag) {
		if (w->host_flag[0] & 1u)
			w->barrier_score = tuning().barrier_sticky;
		else if (w->barrier_score > 0)
			w->barrier_score--;
	}
	return w && w->barrier_score > 0;
}

static Chunking pick_chunking(Workspace *w, long long hint, bool cuckoo)
{
	Chunking k;
	k.passes = 1;
	k.flush = 0;
	k.step = (uint32_t)tuning().insert_chunk;
	const bool lately = cuckoo && chains_lately(w);
	if (!cuckoo || hint <= 0)
		return k;
	long long c = tuning().insert_chunk;
	if (!lately && tuning().calm_factor > 1)
		c *= tuning().calm_factor;
	if (hint <= c)
		return k;
	const long long n = (hint + c - 1) / c;
	k.passes = n > 250 ? 250 : (int)n;
	k.step = (uint32_t)c;
	return k;
}

static void chunk_range(int c, const Chunking &k, uint32_t *lo, uint32_t *hi)
{
	const uint32_t step = k.step;
	const int nchunks = k.passes;
	if (nchunks == 1) {
		*lo = 0;
		*hi = kNone;
		return;
	}
	const int data = nchunks - k.flush;            // the last passes are flush passes
	if (c >= data) {
		*lo = kNone - 15u;                          // empty range: carried requests only
		*hi = kNone;
		return;
	}
	*lo = (uint32_t)c * step;
	*hi = c == data - 1 ? kNone : (uint32_t)(c + 1) * step;   // the last data chunk takes any excess
}

template <Op OP>
static int run_update(const DevGeom &g, uint32_t *table, const ReqSrc &src, hipStream_t stream,
		long long hint = 0)
{
	Workspace *w = nullptr;
	if (src.single_n > kMaxReqCap)
		return MKV_ETOOBIG;
	if (hint > kMaxReqCap)
		return MKV_ETOOBIG;
	CallGuard call;
	int rc = order_updates(stream);
	if (rc != MKV_OK)
		return rc;
	rc = ensure_ws(&w, (uint64_t)g.hash_mask + 1, ws_need(src, hint), stream);
	if (rc != MKV_OK)
		return rc;
	if (src.single_n > 0 && hint <= 0)
		hint = src.single_n;               // delete: the count is a host value
	const Chunking ck = pick_chunking(w, hint, OP == OP_CUCKOO);
	const int nchunks = ck.passes;
	for (int c = 0; c < nchunks; c++) {
		uint32_t lo, hi;
		chunk_range(c, ck, &lo, &hi);
		launch_mark<OP>(w, g, src, stream, table, lo, hi);
		rc = launch_rounds<OP>(w, g, table, stream, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : hint);
		if (rc != MKV_OK)
			return rc;
	}
	return updates_issued(stream);
}

// The work-groups of the tail launch meet at spin barriers (grid_sync): every one of them must be
// resident at once.  Asked once per device and kernel (a CU-masked or partitioned device holds
// fewer than Tuning::tail_groups may ask for).
template <Op OP>
static unsigned tail_groups_for()
{
	static int cap[64] = {0};
	int dev = 0;
	(void)hipGetDevice(&dev);
	int g = tuning().tail_groups;
	if (dev >= 0 && dev < 64) {
		if (cap[dev] == 0) {
			int per_cu = 0, cus = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail<OP>, kTailThreads, 0) != hipSuccess ||
					hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
				per_cu = cus = 0;
			cap[dev] = per_cu * cus > 0 ? per_cu * cus : -1;
		}
		if (cap[dev] > 0 && g > cap[dev])
			g = cap[dev];
	}
	return (unsigned)(g < 1 ? 1 : g);
}

// K2..K4
// done_ev: an event that completes with the LAST kernel of the pass (the tail launch delivers its
// end into it, hipExtLaunchKernelGGL): what a hipEventRecord behind the launch would give, without
// the marker packet of its own (3 us between two kernels of a stream, tools/probes/event_probe.hip)
template <Op OP>
static int launch_rounds(Workspace *w, const DevGeom &g, uint32_t *table, hipStream_t stream,
		uint32_t lo, uint32_t hi, int first_chunk, int last_chunk, hipEvent_t done_ev, long long hint)
{
	// (hint: the caller's upper bound on the pass's requests, 0 = unknown)
	int tail_rounds = tuning().tail_rounds;
	if (tail_rounds > kMaxTailRounds)
		tail_rounds = kMaxTailRounds;
	// what the rounds are told about the tuning: bit 2 -- every eviction chain left to the tail (tests);
	// bit 3 -- no groups of duplicates; bit 5 -- no self-help of a bucket's second registrant
	const int rhint = (tuning().debug_skip & 4) | (tuning().dup_groups ? 0 : 8) | (tuning().self_help ? 0 : 32);
	hipLaunchKernelGGL((k_round<OP, 1>), dim3((unsigned)tuning().round_blocks), dim3(256), 0, stream, w->ctl, g, table,
			w->owner[0], w->owner[1], w->reqc, w->state, w->undo, w->chain_logs, lo, hi, (uint32_t)w->req_cap,
			rhint, w->cnt + ((size_t)g.hash_mask + 1u), w->owner[4]);
	// round 2: inside the tail launch (k_tail's first phase), unless the pass is announced with more
	// requests than that launch has lanes for in a few turns, or is a delete
	// (word 3 of the tail kernel's flags, read without synchronising -- a pass late, which only costs time:
	// round 1 has been leaving round 2 a lot: duplicates of hot keys, say)
	const bool merged = OP != OP_DELETE && tuning().merge_round2 != 0 && hint <= (long long)tuning().merge_max &&
			!(w->host_flag && w->host_flag[3] && tuning().merge_round2 == 1);
	// (round 2 looks at every request's tag to find its few: a lane per announced request, within 2048
	// work-groups -- 524 K tags on the 512 work-groups of round 1 took 15 us, on 2048 they take 6.5)
	long long r2_blocks = (hint + 255) / 256;
	r2_blocks = r2_blocks < tuning().round_blocks ? tuning().round_blocks : (r2_blocks > 2048 ? 2048 : r2_blocks);
	if (!merged)
		hipLaunchKernelGGL((k_round<OP, 2>), dim3((unsigned)r2_blocks), dim3(256), 0, stream, w->ctl, g, table,
				w->owner[1], w->owner[0], w->reqc, w->state, w->undo, w->chain_logs, lo, hi, (uint32_t)w->req_cap,
				rhint, w->cnt + ((size_t)g.hash_mask + 1u), w->owner[4]);
	MKV_LAUNCH_TIMED((k_tail<OP>), dim3(tail_groups_for<OP>()), dim3(kTailThreads), stream, (hipEvent_t) nullptr, done_ev, w->ctl, g, table, w->owner[2],
			w->owner[3], w->cnt, w->owner[0], w->owner[1], w->owner[4], w->reqc, w->state, w->undo, w->chain_logs, w->glist,
			tail_rounds, lo, hi, first_chunk,
			last_chunk, w->host_flag_dev, (uint32_t)tuning().grid_min, (uint32_t)tuning().grid_progress, tuning().debug_skip & 3,
			merged ? rhint : -1, (uint32_t)w->req_cap);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

}  // namespace mkv

extern "C" int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);

extern "C" int mkv_insert(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, void *stream)
{
	return mkv_insert_hinted(geom, table, blk_input, blk_elem_num, num_blks, 0, stream);
}

extern "C" int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || num_blks < 0 || num_blks > mkv::kMaxBlocks)
		return MKV_EINVAL;
	if (num_blks == 0)
		return MKV_OK;
	if (!table || !blk_input || !blk_elem_num)
		return MKV_EINVAL;
	mkv::ReqSrc src;
	src.blk_input = reinterpret_cast<const uint32_t *const *>(blk_input);
	src.single = nullptr;
	src.blk_elem_num = blk_elem_num;
	src.single_n = 0;
	src.nblk = (uint32_t)num_blks;
	src.cnt_stride = 1;
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	if (d.scheme == MKV_SCHEME_2CHOICE)
		return mkv::run_update<mkv::OP_2CHOICE>(d, t, src, (hipStream_t)stream, size_hint);
	return mkv::run_update<mkv::OP_CUCKOO>(d, t, src, (hipStream_t)stream, size_hint);
}

// Insert of mkv_cycle, in two halves around the search launch: the mark pass is
// enqueued on the workspace's side stream right after `fork` (everything already
// on `stream` is complete before it starts, this cycle's search is not waited
// for), the rounds follow on `stream` once mark has joined.
extern "C" int mkv_cycle_insert_begin(const mkv_geom_t *geom, const bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || num_blks <= 0 || num_blks > mkv::kMaxBlocks || !blk_input ||
			!blk_elem_num)
		return MKV_EINVAL;
	mkv::ReqSrc src;
	src.blk_input = reinterpret_cast<const uint32_t *const *>(blk_input);
	src.single = nullptr;
	src.blk_elem_num = blk_elem_num;
	src.single_n = 0;
	src.nblk = (uint32_t)num_blks;
	src.cnt_stride = 1;
	mkv::Workspace *w = nullptr;
	hipStream_t s = (hipStream_t)stream;
	if (size_hint > mkv::kMaxReqCap)
		return MKV_ETOOBIG;
	mkv::CallGuard call;
	int rc = mkv::order_updates(s);
	if (rc != MKV_OK)
		return rc;
	rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, mkv::ws_need(src, size_hint), s, mkv::kCycleSalt);
	if (rc != MKV_OK)
		return rc;
	if (hipEventRecord(w->ev_fork, s) != hipSuccess || hipStreamWaitEvent(w->aux, w->ev_fork, 0) != hipSuccess)
		return MKV_EHIP;
	uint32_t lo, hi;
	w->cycle_ck = mkv::pick_chunking(w, size_hint, d.scheme == MKV_SCHEME_CUCKOO);
	mkv::chunk_range(0, w->cycle_ck, &lo, &hi);
	if (d.scheme == MKV_SCHEME_2CHOICE)
		mkv::launch_mark<mkv::OP_2CHOICE>(w, d, src, w->aux, nullptr, lo, hi);
	else
		mkv::launch_mark<mkv::OP_CUCKOO>(w, d, src, w->aux, reinterpret_cast<const uint32_t *>(table), lo, hi);
	if (hipEventRecord(w->ev_join, w->aux) != hipSuccess)
		return MKV_EHIP;
	return MKV_OK;
}

// Search of a cycle and mark pass of its insert in ONE launch; the rounds follow
// with mkv_cycle_insert_end(..., joined = 0).
extern "C" int mkv_cycle_search_mark(const mkv_geom_t *geom, const selem_t *in, loc_t *out,
		const bucket_t *table, int64_t num_search, ielem_t **blk_input, int *blk_elem_num, int num_blks,
		long long size_hint, void *stream, void *ev_start, void *ev_stop)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || num_blks <= 0 || num_blks > mkv::kMaxBlocks || !blk_input ||
			!blk_elem_num || num_search <= 0 || !in || !out || !table)
		return MKV_EINVAL;
	mkv::ReqSrc src;
	src.blk_input = reinterpret_cast<const uint32_t *const *>(blk_input);
	src.single = nullptr;
	src.blk_elem_num = blk_elem_num;
	src.single_n = 0;
	src.nblk = (uint32_t)num_blks;
	src.cnt_stride = 1;
	mkv::Workspace *w = nullptr;
	hipStream_t s = (hipStream_t)stream;
	if (size_hint > mkv::kMaxReqCap)
		return MKV_ETOOBIG;
	mkv::CallGuard call;
	int rc = mkv::order_updates(s);
	if (rc != MKV_OK)
		return rc;
	rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, mkv::ws_need(src, size_hint), s, mkv::kCycleSalt);
	if (rc != MKV_OK)
		return rc;
	const int64_t search_blocks = (num_search + 255) / 256;      // U = 4: 256 requests per block
	const uint2 *in2 = reinterpret_cast<const uint2 *>(in);
	const uint4 *t4 = reinterpret_cast<const uint4 *>(table);
	// the fused launch marks the first chunk only
	uint32_t lo, hi;
	w->cycle_ck = mkv::pick_chunking(w, size_hint, d.scheme == MKV_SCHEME_CUCKOO);
	mkv::chunk_range(0, w->cycle_ck, &lo, &hi);
	const uint32_t mark_blocks = mkv::mark_blocks_for(size_hint, w->barrier_score > 0);
	if (search_blocks + mark_blocks > 0x7fffffffLL)
		return MKV_ETOOBIG;
	const dim3 grid((unsigned)(search_blocks + mark_blocks)), block(256);
	// (ev_start / ev_stop: HIP events that receive the START and END time of this very dispatch --
	// hipExtLaunchKernelGGL -- i.e. the kernel's duration as a profiler sees it, without the
	// launch latency a pair of hipEventRecord calls around the launch includes)
	hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
	if (d.scheme == MKV_SCHEME_2CHOICE)
		MKV_LAUNCH_TIMED((mkv::k_search_mark<mkv::OP_2CHOICE>), grid, block, s, e0, e1, in2, out, t4, num_search,
				(uint32_t)search_blocks, w->ctl, src, d, w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi,
				mkv::tuning().lazy_alt);
	else
		MKV_LAUNCH_TIMED((mkv::k_search_mark<mkv::OP_CUCKOO>), grid, block, s, e0, e1, in2, out, t4, num_search,
				(uint32_t)search_blocks, w->ctl, src, d, w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi,
				mkv::tuning().lazy_alt);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" int mkv_cycle_insert_end(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream, int joined)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || !table)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	hipStream_t s = (hipStream_t)stream;
	int rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, 0, s, mkv::kCycleSalt);
	if (rc != MKV_OK)
		return rc;
	if (joined && hipStreamWaitEvent(s, w->ev_join, 0) != hipSuccess)
		return MKV_EHIP;
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	mkv::ReqSrc src;
	src.blk_input = reinterpret_cast<const uint32_t *const *>(blk_input);
	src.single = nullptr;
	src.blk_elem_num = blk_elem_num;
	src.single_n = 0;
	src.nblk = (uint32_t)num_blks;
	src.cnt_stride = 1;
	// chunk 0 was marked beside (or inside) the search launch; later chunks are marked here
	const mkv::Chunking ck = w->cycle_ck;
	const int nchunks = ck.passes;
	for (int c = 0; c < nchunks; c++) {
		uint32_t lo, hi;
		mkv::chunk_range(c, ck, &lo, &hi);
		if (d.scheme == MKV_SCHEME_2CHOICE) {
			if (c > 0)
				mkv::launch_mark<mkv::OP_2CHOICE>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_2CHOICE>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : size_hint);
		} else {
			if (c > 0)
				mkv::launch_mark<mkv::OP_CUCKOO>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_CUCKOO>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : size_hint);
		}
		if (rc != MKV_OK)
			return rc;
	}
	return mkv::updates_issued(s);
}

// Local phase of a sharded step in one call: search every received slot and insert
// what the slots carry (one block per source rank, counts read from the slot headers
// on the device).  set_ptrs: device array of the nslots device pointers to the slots'
// insert regions (fixed per receive buffer).
extern "C" int mkv_slots_cycle_self(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out_slots, bucket_t *table, ielem_t **set_ptrs, long long size_hint,
		void *stream, void *timer, int slot, int self, loc_t *self_out, void *done_event);

extern "C" int mkv_slots_cycle(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out_slots, bucket_t *table, ielem_t **set_ptrs, long long size_hint,
		void *stream, void *timer, int slot)
{
	return mkv_slots_cycle_self(geom, recvbuf, nslots, cap_get, cap_set, out_slots, table, set_ptrs, size_hint, stream,
			timer, slot, -1, nullptr, nullptr);
}

// The same with the results of slot `self` written to self_out (cap_get x 2 loc_t) instead of their
// place in out_slots (the sharded engine: a rank's answers to itself go straight into its buffer of
// returned results).  self < 0: none.  done_event (a hipEvent_t, or NULL): completes with the call's
// last kernel, as if recorded behind it.
extern "C" int mkv_slots_cycle_self(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out_slots, bucket_t *table, ielem_t **set_ptrs, long long size_hint,
		void *stream, void *timer, int slot, int self, loc_t *self_out, void *done_event)
{
	if (self >= nslots || (self >= 0 && !self_out))
		return MKV_EINVAL;
	const uint32_t self_u = self >= 0 ? (uint32_t)self : 0xffffffffu;
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || nslots < 1 || nslots > mkv::kMaxBlocks || cap_get < 0 ||
			cap_get > 0x7fffffffLL || cap_set < 0 || !recvbuf || !out_slots || !table)
		return MKV_EINVAL;
	MkvTimer *tm = reinterpret_cast<MkvTimer *>(timer);
	if (tm && (slot < 0 || slot >= tm->slots))
		return MKV_EINVAL;
	const int64_t slot_words = mkv::kSlotHdrWords + cap_get * 2 + cap_set * 3;
	hipStream_t s = (hipStream_t)stream;
	const uint32_t chunks = (uint32_t)((cap_get + 255) / 256);
	const uint32_t search_blocks = (chunks ? chunks : 1u) * (uint32_t)nslots;
	const uint4 *t4 = reinterpret_cast<const uint4 *>(table);
	if (cap_set == 0 || !set_ptrs) {
		// nothing to insert: plain slot search (mark_blocks = 0)
		mkv::ReqSrc none = {};
		MKV_LAUNCH_TIMED((mkv::k_search_slots_mark<mkv::OP_CUCKOO>), dim3(search_blocks), dim3(256), s, tm ? tm->ev[2 * slot] : nullptr, tm ? tm->ev[2 * slot + 1] : nullptr,
				recvbuf, slot_words, (uint32_t)cap_get, out_slots, t4, chunks ? chunks : 1u, search_blocks,
				(mkv::Ctl *)nullptr, none, d, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint4 *)nullptr,
				(uint8_t *)nullptr, 0u, 0u, self_u, self_out, 0);
		if (done_event && hipEventRecord((hipEvent_t)done_event, s) != hipSuccess)
			return MKV_EHIP;
		return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
	}
	mkv::ReqSrc src;
	src.blk_input = reinterpret_cast<const uint32_t *const *>(set_ptrs);
	src.single = nullptr;
	src.blk_elem_num = reinterpret_cast<const int *>(recvbuf + 1);     // n_set of slot b: header word 1
	src.single_n = 0;
	src.nblk = (uint32_t)nslots;
	src.cnt_stride = (uint32_t)slot_words;
	mkv::Workspace *w = nullptr;
	if (size_hint > mkv::kMaxReqCap)
		return MKV_ETOOBIG;
	// (no ordering against other streams here: the callers of this entry point -- the sharded
	// engines -- run the local phases of their lanes' streams in batch order with events of their own;
	// host threads take turns for the enqueueing like everywhere else)
	mkv::CallGuard call;
	int rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, mkv::ws_need(src, size_hint), s, mkv::kCycleSalt);
	if (rc != MKV_OK)
		return rc;
	const bool cuckoo = d.scheme == MKV_SCHEME_CUCKOO;
	const mkv::Chunking ck = mkv::pick_chunking(w, size_hint, cuckoo);
	const int nchunks = ck.passes;
	uint32_t lo, hi;
	mkv::chunk_range(0, ck, &lo, &hi);
	const uint32_t mark_blocks = mkv::mark_blocks_for(size_hint, w->barrier_score > 0);
	const dim3 grid(search_blocks + mark_blocks), block(256);
	hipEvent_t e0 = tm ? tm->ev[2 * slot] : nullptr, e1 = tm ? tm->ev[2 * slot + 1] : nullptr;
	if (cuckoo)
		MKV_LAUNCH_TIMED((mkv::k_search_slots_mark<mkv::OP_CUCKOO>), grid, block, s, e0, e1, recvbuf, slot_words,
				(uint32_t)cap_get, out_slots, t4, chunks ? chunks : 1u, search_blocks, w->ctl, src, d,
				w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi, self_u, self_out, mkv::tuning().lazy_alt);
	else
		MKV_LAUNCH_TIMED((mkv::k_search_slots_mark<mkv::OP_2CHOICE>), grid, block, s, e0, e1, recvbuf, slot_words,
				(uint32_t)cap_get, out_slots, t4, chunks ? chunks : 1u, search_blocks, w->ctl, src, d,
				w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi, self_u, self_out, mkv::tuning().lazy_alt);
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	for (int c = 0; c < nchunks; c++) {
		mkv::chunk_range(c, ck, &lo, &hi);
		hipEvent_t dev = c == nchunks - 1 ? (hipEvent_t)done_event : nullptr;
		if (cuckoo) {
			if (c > 0)
				mkv::launch_mark<mkv::OP_CUCKOO>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_CUCKOO>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, dev, nchunks > 1 ? (long long)ck.step : size_hint);
		} else {
			if (c > 0)
				mkv::launch_mark<mkv::OP_2CHOICE>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_2CHOICE>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, dev, nchunks > 1 ? (long long)ck.step : size_hint);
		}
		if (rc != MKV_OK)
			return rc;
	}
	return MKV_OK;
}

// ---- one call for ALL workers of a scheduler cycle ------------------------------------------------
// The reference's scheduler runs, per cycle, memset + gpu_hash_search + gpu_hash_delete + gpu_hash_insert
// for each of up to 16 workers on a stream each (src/mega_scheduler.c:393-502) with at most 32 767
// searches and 8 x 4 096 inserts per worker (src/mega.c:136,143-144): 16 x 9 launches of a few
// microseconds of work each, the updates serialised.  Here the whole cycle is four pieces of work:
// ONE search launch over every worker's requests (a miss writes 0: no memset), one tiny launch that
// gathers the workers' block pointers and counts (device arrays the host rewrites per cycle, as the
// ABI has them) into one array, ONE delete pass over the workers' delete arrays as its blocks and ONE
// insert pass over all the workers' blocks, worker-major.  Order: every search sees the table as the
// cycle found it; then the deletes in (worker, index) order; then the inserts in (worker, block, index)
// order -- what the per-worker calls give when the workers' keys do not meet within a cycle (per worker
// they run delete then insert: a worker deleting what an EARLIER worker inserts in the same cycle is
// the one case that differs).
namespace mkv {

constexpr int kMaxWorkers = 32;

struct MultiSearch {
	const uint2 *in[kMaxWorkers];
	uint32_t *out[kMaxWorkers];
	uint32_t n[kMaxWorkers];
	uint32_t first[kMaxWorkers + 1];     // first work-group of worker w; [nw] = all
	int nw;
};

__global__ __launch_bounds__(256) void k_search_multi(MultiSearch a, const uint4 *__restrict__ table, uint32_t hash_mask,
		uint32_t blk_mask)
{
	int w = 0;
	while (w + 1 < a.nw && blockIdx.x >= a.first[w + 1])
		w++;
	search_body<4, true, false>(blockIdx.x - a.first[w], a.in[w], a.out[w], table, (int64_t)a.n[w], hash_mask, blk_mask);
}

struct GatherBlocks {
	const uint32_t *const *ins_ptrs[kMaxWorkers];    // worker w's device array of block pointers
	const int *ins_cnts[kMaxWorkers];                // ... and of block counts
	uint32_t ins_first[kMaxWorkers + 1];             // worker w's first block in the gathered array
	const uint32_t *del_ptr[kMaxWorkers];
	int del_cnt[kMaxWorkers];
	int nw;
};

__global__ void k_gather_blocks(GatherBlocks a, const uint32_t **ins_ptrs, int *ins_cnts, const uint32_t **del_ptrs,
		int *del_cnts)
{
	const uint32_t t = threadIdx.x;
	if (t < a.ins_first[a.nw]) {
		int w = 0;
		while (w + 1 < a.nw && t >= a.ins_first[w + 1])
			w++;
		const uint32_t b = t - a.ins_first[w];
		ins_ptrs[t] = a.ins_ptrs[w][b];
		ins_cnts[t] = a.ins_cnts[w][b];
	}
	if (t < (uint32_t)a.nw) {
		del_ptrs[t] = a.del_ptr[t];
		del_cnts[t] = a.del_cnt[t];
	}
}

// the gathered arrays: one set per device, made on first use (a cycle's calls are one turn: CallGuard)
struct WorkerArrays {
	const uint32_t **ins_ptrs = nullptr, **del_ptrs = nullptr;
	int *ins_cnts = nullptr, *del_cnts = nullptr;
};
static WorkerArrays g_wk[64];

}  // namespace mkv

extern "C" int mkv_cycle_workers(const mkv_geom_t *geom, bucket_t *table, const mkv_worker_t *workers, int nworkers,
		void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || !table || !workers || nworkers < 1 || nworkers > mkv::kMaxWorkers)
		return MKV_EINVAL;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
		return MKV_EHIP;
	hipStream_t s = (hipStream_t)stream;
	mkv::MultiSearch ms;
	mkv::GatherBlocks gb;
	memset(&ms, 0, sizeof(ms));
	memset(&gb, 0, sizeof(gb));
	ms.nw = gb.nw = nworkers;
	uint32_t blocks = 0, nblk = 0;
	long long ndel = 0;
	for (int w = 0; w < nworkers; w++) {
		const mkv_worker_t &k = workers[w];
		if (k.num_search < 0 || k.num_search > 0x7fffffffLL || k.num_delete < 0 || k.num_delete > 0x7fffffffLL ||
				k.num_blks < 0 || (k.num_search && (!k.search_in || !k.search_out)) || (k.num_delete && !k.delete_in) ||
				(k.num_blks && (!k.blk_input || !k.blk_elem_num)))
			return MKV_EINVAL;
		ms.in[w] = reinterpret_cast<const uint2 *>(k.search_in);
		ms.out[w] = reinterpret_cast<uint32_t *>(k.search_out);
		ms.n[w] = (uint32_t)k.num_search;
		ms.first[w] = blocks;
		blocks += (uint32_t)((k.num_search + 255) / 256);
		gb.ins_ptrs[w] = reinterpret_cast<const uint32_t *const *>(k.blk_input);
		gb.ins_cnts[w] = k.blk_elem_num;
		gb.ins_first[w] = nblk;
		nblk += (uint32_t)k.num_blks;
		gb.del_ptr[w] = reinterpret_cast<const uint32_t *>(k.delete_in);
		gb.del_cnt[w] = (int)k.num_delete;
		ndel += k.num_delete;
	}
	ms.first[nworkers] = blocks;
	gb.ins_first[nworkers] = nblk;
	if (nblk > (uint32_t)mkv::kMaxBlocks)
		return MKV_EINVAL;
	mkv::CallGuard call;                        // (the parts below are one turn)
	mkv::WorkerArrays &wa = mkv::g_wk[dev];
	if (!wa.ins_ptrs) {
		const size_t n = (size_t)mkv::kMaxBlocks;
		if (hipMalloc(&wa.ins_ptrs, n * sizeof(void *)) != hipSuccess || hipMalloc(&wa.ins_cnts, n * sizeof(int)) != hipSuccess ||
				hipMalloc(&wa.del_ptrs, mkv::kMaxWorkers * sizeof(void *)) != hipSuccess ||
				hipMalloc(&wa.del_cnts, mkv::kMaxWorkers * sizeof(int)) != hipSuccess)
			return MKV_ENOMEM;
	}
	if (blocks)
		hipLaunchKernelGGL(mkv::k_search_multi, dim3(blocks), dim3(256), 0, s, ms, reinterpret_cast<const uint4 *>(table),
				d.hash_mask, d.blk_mask);
	if (nblk || ndel)
		hipLaunchKernelGGL(mkv::k_gather_blocks, dim3(1), dim3(1024), 0, s, gb, wa.ins_ptrs, wa.ins_cnts, wa.del_ptrs,
				wa.del_cnts);
	if (hipGetLastError() != hipSuccess)
		return MKV_EHIP;
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	int rc = MKV_OK;
	if (ndel) {
		mkv::ReqSrc src;
		src.blk_input = wa.del_ptrs;
		src.single = nullptr;
		src.blk_elem_num = wa.del_cnts;
		src.single_n = 0;
		src.nblk = (uint32_t)nworkers;
		src.cnt_stride = 1;
		rc = mkv::run_update<mkv::OP_DELETE>(d, t, src, s, ndel);
	}
	if (rc == MKV_OK && nblk) {
		mkv::ReqSrc src;
		src.blk_input = wa.ins_ptrs;
		src.single = nullptr;
		src.blk_elem_num = wa.ins_cnts;
		src.single_n = 0;
		src.nblk = nblk;
		src.cnt_stride = 1;
		rc = d.scheme == MKV_SCHEME_2CHOICE ? mkv::run_update<mkv::OP_2CHOICE>(d, t, src, s, 0)
		                                    : mkv::run_update<mkv::OP_CUCKOO>(d, t, src, s, 0);
	}
	return rc;
}

extern "C" int mkv_delete(const mkv_geom_t *geom, delem_t *in, bucket_t *table,
		int64_t num_elem, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || num_elem < 0)
		return MKV_EINVAL;
	if (num_elem == 0)
		return MKV_OK;
	if (!table || !in)
		return MKV_EINVAL;
	mkv::ReqSrc src;
	src.blk_input = nullptr;
	src.single = reinterpret_cast<const uint32_t *>(in);
	src.blk_elem_num = nullptr;
	src.single_n = num_elem;
	src.nblk = 1;
	src.cnt_stride = 1;
	return mkv::run_update<mkv::OP_DELETE>(d, reinterpret_cast<uint32_t *>(table), src,
			(hipStream_t)stream);
}

// A stream is about to be destroyed (the library's own: pipeline, shard engine; or the host's):
// its work is waited for, and it is no longer named as "where the last update was issued".
// (Nothing else refers to a stream: the update workspaces belong to the device, and the
// cross-stream order is kept with an event of the library's own, never with a stream handle.)
extern "C" void mkv_forget_stream(void *stream)
{
	(void)hipStreamSynchronize((hipStream_t)stream);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return;
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	auto it = mkv::g_order.find(dev);
	if (it != mkv::g_order.end() && it->second.any && it->second.last == (hipStream_t)stream)
		it->second.any = false;
}

extern "C" int mkv_reserve(const mkv_geom_t *geom, int64_t max_requests, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || max_requests < 0)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	int64_t need = (int64_t)mkv::tuning().req_cap;
	if (max_requests > need)
		need = max_requests;
	if (need > mkv::kMaxReqCap)
		return MKV_ETOOBIG;
	int rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, need, (hipStream_t)stream);
	if (rc == MKV_OK)
		rc = mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, need, (hipStream_t)stream, mkv::kCycleSalt);
	return rc;
}

// Frees the update workspaces of the calling process (both of every device).  Synchronises.
extern "C" void mkv_release_all(void)
{
	(void)hipDeviceSynchronize();
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	for (auto &kv : mkv::g_ws) {
		mkv::free_ws(kv.second);
		delete kv.second;
	}
	mkv::g_ws.clear();
	for (auto &kv : mkv::g_order)
		if (kv.second.ev)
			(void)hipEventDestroy(kv.second.ev);
	mkv::g_order.clear();
}

// Batches dropped so far because they exceeded the update workspace (see mkv_reserve); the
// count is current up to the last insert/delete call that has COMPLETED before this one.
extern "C" unsigned long long mkv_dropped_batches(void)
{
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	for (auto &kv : mkv::g_ws)
		mkv::report_dropped(kv.second);
	return mkv::g_dropped_total;
}

// Bytes of device memory the update workspaces of the current device hold.
extern "C" long long mkv_workspace_bytes(void)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return -1;
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	long long total = 0;
	for (auto &kv : mkv::g_ws) {
		const mkv::Workspace *w = kv.second;
		if (!w || w->device != dev)
			continue;
		total += (long long)w->nbuckets * 56 +
				(long long)w->req_cap * (1 + 16 + 16 + 4) +
				(long long)(((size_t)mkv::kChainLogCap + mkv::kTailLanes) * sizeof(mkv::ChainLog)) + (long long)sizeof(mkv::Ctl);
	}
	return total;
}

extern "C" int mkv_last_update_info_ex(void *stream, int cycle_insert, mkv_update_info_t *info);

extern "C" int mkv_last_update_info(void *stream, mkv_update_info_t *info)
{
	return mkv_last_update_info_ex(stream, 0, info);
}

extern "C" int mkv_last_update_info_ex(void *stream, int cycle_insert, mkv_update_info_t *info)
{
	if (!info)
		return MKV_EINVAL;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
		if (it == mkv::g_ws.end() || !it->second->ctl)
			return MKV_EINVAL;
		w = it->second;
	}
	mkv::Ctl h;
	if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
		return MKV_EHIP;
	if (hipMemcpy(&h, w->ctl, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
	info->total = h.last.overflow ? 0xffffffffu : h.last.total;
	info->round1 = h.last.round1;
	info->round2 = h.last.round2;
	info->tail_parallel = h.last.tail_parallel;
	info->tail_serial = h.last.tail_serial;
	info->undone = h.last.undone;
	info->first_hard = h.last.first_hard;
	info->chains = h.last.chains;
	info->bar_chain = h.last.bar_chain;
	info->bar_class = h.last.bar_class;
	info->tail_rounds = h.last.tail_rounds;
	info->helped = h.last.helped;
	info->losers = h.last.losers;
	return MKV_OK;
}

// ---- diagnostics builds only (make DIAG=1: ../_lib/libmegakv_hip_diag.so) ----
#ifdef MKV_DIAGNOSTICS
// diagnostics (tests/debug/streams_probe.py): what the workspace holds for request idx after the last
// pass -- out[0] = tag, out[1..4] = request cache {sig, hash, loc, flags}, out[5..8] = undo record
extern "C" int mkv_debug_request(int cycle_insert, uint32_t idx, uint32_t out[9])
{
	int dev = 0;
	if (!out || hipGetDevice(&dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
		if (it == mkv::g_ws.end() || !it->second->ctl || (int64_t)idx >= it->second->req_cap)
			return MKV_EINVAL;
		w = it->second;
	}
	uint8_t st = 0;
	if (hipMemcpy(&st, w->state + idx, 1, hipMemcpyDeviceToHost) != hipSuccess ||
			hipMemcpy(out + 1, w->reqc + idx, 16, hipMemcpyDeviceToHost) != hipSuccess ||
			hipMemcpy(out + 5, w->undo + idx, 16, hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
	out[0] = st;
	return MKV_OK;
}

#ifdef MKV_TRACE
// debug builds: the tail's trace (words 0..3: count; then {code, a, b, c} per event); reset != 0 clears it
extern "C" int mkv_debug_trace(uint32_t *out, uint32_t cap_words, int reset)
{
	if (hipDeviceSynchronize() != hipSuccess)
		return MKV_EHIP;
	size_t n = sizeof(mkv::g_trace);
	if ((size_t)cap_words * 4u < n)
		n = (size_t)cap_words * 4u;
	if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mkv::g_trace), n) != hipSuccess)
		return MKV_EHIP;
	if (reset) {
		const uint32_t z = 0;
		if (hipMemcpyToSymbol(HIP_SYMBOL(mkv::g_trace), &z, 4) != hipSuccess)
			return MKV_EHIP;
	}
	return MKV_OK;
}
#endif

// diagnostics: how often this stream's workspace cleared its reservation tables because
// the round stamps ran low (-1: no such workspace)
extern "C" int mkv_debug_stamp_resets(void *stream, int cycle_insert)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return -1;
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
	return it == mkv::g_ws.end() ? -1 : it->second->stamp_resets;
}

// diagnostics: microseconds the last call's tail kernels spent per phase
// (roll-back, gather, exact replay, retag, in-group rounds, serial), summed over its passes
extern "C" int mkv_debug_tail_phases(void *stream, int cycle_insert, float out_us[16])
{
	int dev = 0;
	if (!out_us || hipGetDevice(&dev) != hipSuccess)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
		if (it == mkv::g_ws.end() || !it->second->ctl)
			return MKV_EINVAL;
		w = it->second;
	}
	mkv::Ctl h;
	if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
			hipMemcpy(&h, w->ctl, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
	for (int k = 0; k < 16; k++)
		out_us[k] = (float)h.tail_phase[k] * 0.01f;
	return MKV_OK;
}
#endif  // MKV_DIAGNOSTICS
