// mkv_upd_tail.h -- insert / delete, K4: order validation of late chains, the tail's rounds and replays, k_tail, k_init_ctl
// Part of the insert / delete translation unit (mkv_update.hip includes it; the kernels share the control
// block, the reservation helpers and the request cache, and the tail launch runs round 2's body, so they
// are compiled together).  The design is described at the top of mkv_update.hip and in DESIGN.md 5.2.
#pragma once
#include "mkv_upd_common.h"
#include "mkv_upd_round.h"

namespace mkv {

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
	Sorted sorted;                       // a BIG pass's bins (mkv_upd_sorted.h): k_apply's commits are found there, a fourth
	bool sorted_on;                      // source beside the three reservation words -- any number of commits per bucket
};

// a commit of k_apply later than idx on bucket b (its index; kNone: none) -- `sig_only`: only one that carries that signature
__device__ __forceinline__ uint32_t sorted_later_commit(const Tv &tv, const DevGeom &g, uint32_t b, uint32_t idx, bool any,
		uint32_t sig)
{
	uint32_t found = kNone;
	if (tv.sorted_on)
		sorted_for_each_on(tv.sorted, g.hash_mask, b, [&](const uint4 &r, uint32_t) {
			// (tag ST_ROUND1 exactly: pending again, selected for take-back or committed by round 2 -- then the
			// reservation words name it -- it is not k_apply's commit any more)
			if (r.w > idx && (any || r.x == sig) && tv.state[r.w] == ST_ROUND1) {
				found = r.w;
				return true;
			}
			return false;
		});
	return found;
}
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
	return round_commit_used(tv.state, tv.reqc, tv.undo, tv.chain_logs, g, c, b, eff, tv.sorted_on);
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
	return round_commit_used_pre(st, sh, tv.undo, tv.chain_logs, g, c, b, eff, tv.sorted_on);
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
	return sorted_later_commit(tv, g, b, idx, true, 0u);
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
	if (c3 != kNone && commit_used_pre(tv, g, k.st3, k.sh3, c3, b, &e) && e > idx && (any || k.sh3.x == sig))
		return true;
	return sorted_later_commit(tv, g, b, idx, any, sig) != kNone;
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

// a meeting of the work-group for what it keeps in LDS only: stores to global memory stay on their way (wg_sync and
// __syncthreads wait until every one of them has been acknowledged -- a round trip)
__device__ __forceinline__ void lds_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
	// (called by every lane of a work-group: the per-region words a lane each)
	const uint32_t t = threadIdx.x;
	if (t == 0) {
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
	}
	if (t < (uint32_t)kMaxRegions) {
		ctl->g_rbar[0][t] = ctl->g_rbar[1][t] = kNone;
		ctl->g_rmin[0][t] = ctl->g_rmin[1][t] = kNone;
		ctl->g_below[0][t] = ctl->g_below[1][t] = 0u;
		ctl->viol[t] = kNone;
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
		const uint32_t *lkeys, uint8_t *lreg, uint32_t n, const Tv &tv, uint32_t *s_count, uint32_t *s_viol)
{
	// (*s_viol, zero on entry: set when a request of the list reported an order violation -- the caller need not ask
	// the control block; only then does the work-group wait for its stores before it goes on)
	const uint32_t t = threadIdx.x;
	for (uint32_t i = t; i < n; i += blockDim.x) {
		const uint32_t idx = lkeys[i];
		const uint4 rc = reqc[idx];
		const uint32_t P = rc.y & g.hash_mask, A = alt_bucket(rc.y, rc.x, g.hash_mask, g.blk_mask);
		lreg[i] = (uint8_t)(P >> g.region_shift);
		// touch both buckets: the replaying lane then finds them in L2
		const uint32_t a = ld_agent32(table + sig_word(P, 0u)), b = ld_agent32(table + sig_word(A, 0u));
		asm volatile("" ::"v"(a), "v"(b));
		if (OP != OP_DELETE) {
			// ... and what it will ask about the primary bucket's earlier commits (conflict_on): the three
			// reservation words, then the candidates' tags and cache entries -- every request of the list
			// at once here, one after the other's answer there
			uint32_t c1, c2, c3;
			candidates_of(tv, ld_agent(&tv.owner1[P]), ld_agent(&tv.owner2[P]), ld_agent(&tv.second[P]), c1, c2, c3);
			Cands k;
			load_cands(tv, c1, c2 == c1 ? kNone : c2, c3, k);
			asm volatile("" ::"v"(k.st1), "v"(k.st2), "v"(k.st3), "v"(k.sh1.x), "v"(k.sh2.x), "v"(k.sh3.x));
		}
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
					} else {
						*s_viol = 1u;
					}
				}
				stop = __shfl(rc, 0) != 0;
			}
		}
		if (lane == 0 && d)
			atomicAdd(s_count, d);
	}
	lds_sync();
	if (*s_viol)
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
#ifdef MKV_DIAGNOSTICS
				if (bad && !clean)
					atomicAdd(&ctl->dbg[2], 1u);
#endif
				if (bad && clean) {
					// hard right now: it may still run this round if its chain is all its own
					if (walk_and_claim(table, g, req_of(rc), P, A, idx, owner, stamp, tv, s_rbar, sig_at(lo, hi, rc.x & 7u))) {
						state[idx] = ST_CHAIN_OK;
						bad = false;
					} else {
#ifdef MKV_DIAGNOSTICS
						atomicAdd(&ctl->dbg[1], 1u);
#endif
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
#ifdef MKV_DIAGNOSTICS
			ctl->dbg[5] += progress + left;
			ctl->dbg[6] += 1u;
			ctl->dbg[4] += progress;
#endif
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
	const uint32_t t = threadIdx.x, wg = blockIdx.x;
	uint32_t nwg = gridDim.x;
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
		// ... and k_apply's commits on the bucket (a big pass: any number of them, all with this bucket as their primary
		// one, found in the bin): every one later than the asking index goes -- which covers (b) for this bucket, so
		// only its alternate bucket, which it may need when it runs again, is followed
		if (tv.sorted_on) {
			sorted_for_each_on(tv.sorted, g.hash_mask, e.x, [&](const uint4 &r, uint32_t) {
				const uint32_t u = r.w;
				if (u <= e.y || state[u] != ST_ROUND1)
					return false;
				const uint32_t sh = (u & 3u) * 8u;
				if ((atomicOr(reinterpret_cast<uint32_t *>(state) + (u >> 2), (uint32_t)ST_SELECT << sh) >> sh) & ST_SELECT)
					return false;
				const uint32_t at = atomicAdd(nsel, 1u);
				if (at < kSelCap)
					sel[at] = u;
				else
					ok = false;
				const uint32_t A = alt_bucket(r.y, r.x, g.hash_mask, g.blk_mask);
				if (A != e.x) {
					if (sp + 1u > kLaneStack)
						ok = false;
					else
						stack[sp++] = make_uint2(A, u);
				}
				return false;
			});
		}
	}
	return ok;
}

// takes the selected commit u (tag already checked) back: its words restored, its tag ST_DEFER
// (pending, sitting out one round -- and no longer a commit to anybody who asks commit_used)
__device__ __forceinline__ void take_back(const DevGeom &g, uint32_t *table, uint8_t *state, const uint4 *reqc,
		const Undo *undo, const ChainLog *chain_logs, uint32_t u, bool chain, const Tv &tv, uint32_t st)
{
	// (a commit of k_apply -- a big pass -- has its record in its bin)
	const Undo ur = undo_record(undo, tv.sorted, tv.sorted_on && !chain, g.hash_mask, u, reqc[u], st);
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
		uint32_t wg, uint32_t nwg, uint32_t *sync_counter, uint2 *trunc, uint32_t *ntrunc, const Tv &tv)
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
			const Undo u = undo_record(undo, tv.sorted, tv.sorted_on, g.hash_mask, idx, reqc[idx], st);
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
		uint32_t *host_flag, const uint32_t *rec_sums)
{
	// (rec_sums: round 1's per-work-group records summed -- commits, of those by self-help, losers: Ctl::wg_rec)
	const uint32_t t = threadIdx.x;
	if (t >= 64)
		return;
	// Every word the epilogue reads is asked for HERE, before anything waits for an answer: one round trip.  (A lane
	// that loads a word, decides, stores and loads the next made five to seven of them in a row -- 3 us of a launch
	// that had replayed one request.)
	// statistics shards: one wave sums them (cleared by the next mark pass)
	uint32_t c1 = ctl->committed[0][t].n, c2 = ctl->committed[1][t].n;
	uint32_t c3 = ctl->committed[2][t].n, c4 = ctl->committed[3][t].n;
	uint32_t w_hfl = 0, w_hf0 = 0, w_hf1 = 0, w_hfw = 0, w_fh = 0, w_ov = 0, w_cok = 0, w_ht = 0, w_cn = 0;
	unsigned long long w_stamp = 0;
	if (t == 0) {
		w_hfl = ld_agent32(&ctl->hf_losers);
		w_hf0 = ld_agent32(&ctl->hf_last[0]);
		w_hf1 = ld_agent32(&ctl->hf_last[1]);
		w_hfw = ld_agent32(&ctl->hf_walk);
		w_fh = ld_agent32(&ctl->first_hard);
		w_ov = ld_agent32(&ctl->overflow);
		w_cok = ld_agent32(&ctl->chain_ok);
		w_ht = ld_agent32(&ctl->hard_tail);
		w_cn = ld_agent32(&ctl->chain_n);
		w_stamp = ld_agent(&ctl->stamp_base);
	}
	for (int off = 32; off > 0; off >>= 1) {
		c1 += __shfl_down(c1, off);
		c2 += __shfl_down(c2, off);
		c3 += __shfl_down(c3, off);
		c4 += __shfl_down(c4, off);
	}
	if (t != 0)
		return;
	c1 += rec_sums[0];
	c3 += rec_sums[1];
	c4 += rec_sums[2];
	// (a call in several passes: the earlier passes' counts are added to -- one more round trip, for those calls only)
	Info acc{};
	acc.first_hard = kNone;
	if (!first_chunk)
		acc = ctl->last;
	ctl->last.round1 = acc.round1 + c1;
	ctl->last.round2 = acc.round2 + c2;
	ctl->last.helped = acc.helped + c3;
	ctl->last.losers = acc.losers + c4;
	// word 3 of the host's flags: round 1 left round 2 more than a handful -- the host then gives round 2
	// a launch of its own (more lanes than this launch has) until a pass leaves few again
	const uint32_t many = c4 > kMergeLosersMax ? 1u : 0u;
	if (host_flag && many != w_hfl) {
		host_flag[3] = many;
		ctl->hf_losers = many;
	}
#ifdef MKV_DIAGNOSTICS
	// diagnostics of this call (summed over its chunks)
	for (int k = 0; k < 16; k++)
		ctl->tail_phase[k] = (first_chunk ? 0u : ctl->tail_phase[k]) + ((IDLE || !phase_ticks) ? 0u : phase_ticks[k]);
	for (int k = 0; k < 8; k++) {                  // (diagnostic counts ride in the slots of the grid-wide rounds' ticks)
		ctl->tail_phase[8 + k] += ctl->dbg[k] * 100u;
		ctl->dbg[k] = 0;
	}
#else
	(void)phase_ticks;
#endif
	// ... then the control block of the next pass
	ctl->last.total = total_all;
	ctl->last.tail_parallel = acc.tail_parallel + par;
	ctl->last.tail_serial = acc.tail_serial + serial;
	ctl->last.undone = acc.undone + undone;
	ctl->last.tail_rounds = acc.tail_rounds + rounds_total;
	ctl->last.first_hard = w_fh < acc.first_hard ? w_fh : acc.first_hard;
	ctl->last.overflow = w_ov;
	ctl->last.chains = acc.chains + w_cok;
	// (the two "barrier" fields of the diagnostics now count: order violations handled, and hard
	// requests the grid-wide rounds left to the tail)
	const uint32_t bar_chain = acc.bar_chain + viols, bar_class = acc.bar_class + w_ht;
	ctl->last.bar_chain = bar_chain;
	ctl->last.bar_class = bar_class;
	// word 0: chains ran late in this call (the next hinted batch is chunked finely: the chance of
	// an order violation grows with the square of the pass size); word 1: the 40-bit round stamps
	// are nearly used up -- the host clears the reservation tables before a later call
	// (ensure_ws); word 2: batches that exceeded the workspace and were dropped, so far.  Only
	// this kernel writes the words; the host only reads them.
	if (last_chunk && host_flag) {
		// (a store to host memory makes the end of this kernel wait for the PCIe write: 4 us between
		// this launch and the next search, every step -- the words are written when they change)
		const uint32_t f0 = (bar_chain + bar_class) != 0u ? 1u : 0u;
		const uint32_t f1 = w_stamp > kStampMax - kStampReserve ? 1u : 0u;
		if (f0 != w_hf0) {
			host_flag[0] = f0;
			ctl->hf_last[0] = f0;
		}
		if (f1 != w_hf1) {
			host_flag[1] = f1;
			ctl->hf_last[1] = f1;
		}
		if (w_ov)
			host_flag[2] = host_flag[2] + 1u;
		// word 4: the pass walked many eviction chains (a well-filled table) -- the host then has the next mark
		// passes' walks ahead register without returning atomics (prewalk_register)
		const uint32_t f4 = w_cn * 16u > total_all ? 1u : 0u;
		if (f4 != w_hfw) {
			host_flag[4] = f4;
			ctl->hf_walk = f4;
		}
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
	ctl->stamp_base = w_stamp + 3ull + rounds_total;
	ctl->ep_pending = 0u;
	ctl->carry_lo = kNone;
}

// ---- the tail launch ----------------------------------------------------------------------
// Finishes what rounds 1 and 2 left: losers of both rounds and hard requests whose chain
// could not run beside the other requests.  All work-groups share the scans when there is
// much to scan (gather) and the rounds when much is left (grid_rounds);
// otherwise the helpers leave at once and work-group 0 does the rest: reservation rounds,
// then the in-order replay per closed region.  A chain that meets an order violation makes
// the launch roll back the later commits of that region and take them on as well.
template <Op OP, bool SORTED = false>
__global__ __launch_bounds__(kTailThreads) void k_tail(Ctl *ctl, DevGeom g, uint32_t *table,
		unsigned long long *owner, unsigned long long *altmin, unsigned long long *cnt, const unsigned long long *owner1,
		const unsigned long long *owner2, const unsigned long long *second, const uint4 *reqc, uint8_t *state, const Undo *undo,
		const ChainLog *chain_logs,
		uint32_t *glist, int max_rounds,
		uint32_t lo_idx, uint32_t hi_idx, int first_chunk, int last_chunk, uint32_t *host_flag, uint32_t grid_min, uint32_t grid_progress,
		int dbg_path, int round2_hint, uint32_t req_cap, typename SortedArg<SORTED>::type Sarg, int sorted_hint)
{
	const Sorted S = sorted_of_arg(Sarg);
	// (S, sorted_hint: a BIG pass -- partition, then apply per bucket, mkv_upd_apply.h: the commits of k_apply are looked
	// for in its bins, Tv::sorted)
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
	__shared__ uint32_t s_n, s_progress, s_undone, s_par, s_serial, s_left, s_dupw, s_done12, s_viols, s_rviol;
	const uint32_t t = threadIdx.x, wg = blockIdx.x;
	uint32_t nwg = gridDim.x;
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
	// The rounds left nothing and the epilogue has nothing to tell anybody (the usual batch, three passes in
	// four): every work-group reads the four flag words -- one load; final: only kernels before this one in the
	// stream set them -- and leaves.  Nothing is written: the pass's epilogue stays pending (Ctl::ep_pending), the
	// next pass's mark pass steps over its stamps and resets what is to be reset, and whoever asks for the pass's
	// statistics meanwhile works them out from the rounds' counters (mkv_last_update_info_ex).  An extra launch of
	// this grid with nothing to do costs 1.3 - 1.5 us in this chain (profiles/r05_launch_floor.txt); the epilogue --
	// five dependent round trips on one lane -- made it 4.1 - 4.8.
	// (round 1's record for this lane -- what it counted, the first of what it left to round 2: asked for beside the
	// flags, one round trip for both)
	uint4 rec_stat = make_uint4(0u, 0u, 0u, 0u), rec_ll = rec_stat;
	if (t < kTlGroups) {
		const uint4 *rp = reinterpret_cast<const uint4 *>(&ctl->wg_rec[t]);
		rec_stat = rp[0];
		rec_ll = rp[1];
	}
	static_assert(sizeof(Ctl::WgRec) == 32 && offsetof(Ctl, wg_rec) % 32 == 0, "two 16-byte loads");
	const uint4 flags = *reinterpret_cast<const uint4 *>(&ctl->any_loser);     // {any_loser, any_tail, loose, force_tail}
	static_assert(offsetof(Ctl, any_loser) % 16 == 0 && offsetof(Ctl, force_tail) == offsetof(Ctl, any_loser) + 12, "one load");
	if ((flags.x | flags.y | flags.z | flags.w) == 0u && first_chunk && last_chunk)
		return;
	// What the launch goes on to read from the control block is asked for together, HERE: one round trip where the
	// words read one after the other's answer (the pass's size, the carried index, the stamp, the lists' counts, the
	// commit counters) made five.  (Round 2 as this launch's first phase writes the lists' counts and its commit
	// counters: those are read behind it then, as before.)
	// (... and with round 2 ahead nothing is asked for that it does not need: its first meeting would wait for the answers)
	static_assert(offsetof(Ctl, ep_pending) % 16 == 0 && offsetof(Ctl, total) == offsetof(Ctl, ep_pending) + 4 &&
			offsetof(Ctl, carry_lo) == offsetof(Ctl, ep_pending) + 12, "one load");
	const bool r2_here = round2_hint >= 0 && OP != OP_DELETE && flags.x != 0u;   // (round 2 runs in this launch)
	uint4 head1 = make_uint4(0u, 0u, 0u, 0u);                                    // {ep_pending, total, overflow, carry_lo}
	unsigned long long stamp1 = 0ull;                                            // round-1 stamp of this pass
	uint32_t pre_g1 = 0u, pre_g2 = 0u, pre_cnt = 0u, pre_done = 0u;
	uint2 pre_e = make_uint2(kNone, kNone);       // the first two entries this lane would list (work-group 0 lists)
	static_assert(offsetof(Ctl, tl) % 8 == 0 && kTlPer >= 2u, "one 8-byte load");
	if (!r2_here) {
		if (wg == 0)
			pre_e = *reinterpret_cast<const uint2 *>(&ctl->tl[my_round][my_wg][0]);
		head1 = *reinterpret_cast<const uint4 *>(&ctl->ep_pending);
		stamp1 = ctl->stamp_base;
		pre_g1 = ctl->tl_groups[0];
		pre_g2 = ctl->tl_groups[1];
		pre_cnt = ctl->tl_cnt[my_round][my_wg];
		if (t < 64u)
			pre_done = ctl->committed[0][t].n + ctl->committed[1][t].n;
	}
	// Round 1 left round 2 a handful (the usual pass has one loser in two cases of three: a third registrant of a
	// bucket, a request behind one that holds the bucket as its alternate, a second registrant of a full bucket) and
	// listed them (Ctl::ll): this work-group runs round 2 over the list and goes on alone -- no scan of the pass's
	// tags by 32 work-groups, no meeting behind it; the others read the same records, decide alike and leave.
	__shared__ uint32_t s_few_n, s_few_bad, s_few[kFewLosers], s_rec[3];
	bool few = false;
	// the records' sums (lanes beyond the records hold zeros): added up where the launch has a barrier anyway
#define MKV_SUM_RECS()                                                                  \
	do {                                                                                \
		uint32_t a_ = rec_stat.y, b_ = rec_stat.z, c_ = rec_stat.w;                     \
		if (__ballot(a_ | b_ | c_)) {                                                   \
			for (int off = 32; off > 0; off >>= 1) {                                    \
				a_ += __shfl_down(a_, off);                                             \
				b_ += __shfl_down(b_, off);                                             \
				c_ += __shfl_down(c_, off);                                             \
			}                                                                           \
			if ((t & 63u) == 0u) {                                                      \
				if (a_) atomicAdd(&s_rec[0], a_);                                       \
				if (b_) atomicAdd(&s_rec[1], b_);                                       \
				if (c_) atomicAdd(&s_rec[2], c_);                                       \
			}                                                                           \
		}                                                                               \
	} while (0)
	const bool try_few = r2_here && first_chunk && last_chunk;
	if (try_few) {
		if (t == 0) {
			s_few_n = 0u;
			s_few_bad = 0u;
		}
		__syncthreads();
		if (t < kTlGroups) {
			const uint32_t nl = rec_stat.w;
			if (nl > kLlPer) {
				s_few_bad = 1u;
			} else if (nl) {
				const uint32_t at = atomicAdd(&s_few_n, nl);
				if (at + nl > kFewLosers) {
					s_few_bad = 1u;
				} else {
					s_few[at] = rec_ll.x;
					if (nl > 1u) s_few[at + 1u] = rec_ll.y;
					if (nl > 2u) s_few[at + 2u] = rec_ll.z;
					if (nl > 3u) s_few[at + 3u] = rec_ll.w;
				}
			}
		}
		__syncthreads();
	}
	few = try_few && s_few_bad == 0u && s_few_n != 0u;      // (none listed: round 1 ran on more work-groups than it has records for)
	uint32_t p_any_tail = flags.y;      // (no round 2 in this launch: only kernels before this one set the flag)
	if (few) {
		if (wg != 0)
			return;
		nwg = 1u;
		round_body<OP, 2, kTailThreads, SORTED>(0u, 1u, reinterpret_cast<unsigned char *>(lkeys), ctl, g, table, owner2,
				const_cast<unsigned long long *>(owner1), reqc, state, const_cast<Undo *>(undo), const_cast<ChainLog *>(chain_logs),
				lo_idx, hi_idx, req_cap, round2_hint, cnt + ((size_t)g.hash_mask + 1u), second, nullptr, S, s_few, s_few_n);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (every lane's stores have left before anybody reads on)
		wg_sync();
		// ... and that was it, as a rule: the pass's epilogue stays pending then, as on the way out above
		// (the next mark pass resets the round's flag; its commits are counted where the host finds them)
		p_any_tail = ld_agent32(&ctl->any_tail);
		if (p_any_tail == 0u && flags.w == 0u)
			return;
	} else if (r2_here) {
		// (every work-group reads the same flag: round 1, the kernel before, set it)
		static_assert(RoundLds<2, kTailThreads>::bytes <= sizeof(lkeys), "round 2's LDS in the work list's");
		round_body<OP, 2, kTailThreads, SORTED>(wg, nwg, reinterpret_cast<unsigned char *>(lkeys), ctl, g, table, owner2,
				const_cast<unsigned long long *>(owner1), reqc, state, const_cast<Undo *>(undo), const_cast<ChainLog *>(chain_logs),
				lo_idx, hi_idx, req_cap, round2_hint, cnt + ((size_t)g.hash_mask + 1u), second, nullptr, S);
		grid_sync(&ctl->sync_r2, nwg);
		p_any_tail = ld_agent32(&ctl->any_tail);
	}
	if (r2_here) {
		head1 = *reinterpret_cast<const uint4 *>(&ctl->ep_pending);
		stamp1 = ctl->stamp_base;
	}
	const uint32_t total_all = head1.y;
	// Round 2 left nothing (or the epilogue is wanted for its own sake): no lists, no scans, no meetings --
	// every work-group reads the one flag (final: only kernels before this one in the stream -- or the
	// phase above, behind its meeting -- set it) and work-group 0 leaves the control block ready for the
	// next pass.
	if (p_any_tail == 0u) {
		if (wg == 0) {
			if (t == 0)
				s_rec[0] = s_rec[1] = s_rec[2] = 0u;
			__syncthreads();
			MKV_SUM_RECS();
			__syncthreads();
			tail_epilogue<true>(ctl, nullptr, 0u, total_all, 0u, 0u, 0u, 0u, first_chunk, last_chunk, host_flag, s_rec);
		}
		return;
	}
	const uint32_t total = hi_idx < total_all ? hi_idx : total_all;   // this chunk ends here
	const uint32_t carry_in = head1.w;
	// what the grid-wide rounds left, listed by themselves (tail_list_add) -- unless a work-group left
	// more than its place holds, shadows without a leader are about (not listed), or the call runs in
	// several passes.  Every work-group of this launch reads the counts and decides alike.
	__shared__ uint32_t s_tl_total, s_tl_over;
	if (t == 0) {
		s_tl_total = 0u;
		s_tl_over = 0u;
		s_rec[0] = s_rec[1] = s_rec[2] = 0u;
	}
	__syncthreads();
	MKV_SUM_RECS();          // (read behind the next barrier: s_done12 below, the epilogue at the end)
	const uint32_t p_loose = r2_here ? ctl->loose : flags.z;
	const bool list_ok = first_chunk && last_chunk && carry_in >= lo_idx && p_loose == 0u && p_any_tail != 0u;
	const uint32_t tl_g1 = !list_ok ? 0u : r2_here ? ctl->tl_groups[0] : pre_g1;
	const uint32_t tl_g2 = !list_ok || flags.x == 0u ? 0u : r2_here ? ctl->tl_groups[1] : pre_g2;    // (round 2 leaves at once without losers)
	uint32_t my_cnt = 0, my_base = 0;
	if (list_ok) {
		if (tl_g1 > kTlGroups || tl_g2 > kTlGroups) {
			s_tl_over = 1u;
		} else if (my_wg < (my_round ? tl_g2 : tl_g1)) {
			// (round 1: one or the other, round_body -- its record is in this lane's registers already)
			my_cnt = (r2_here ? ctl->tl_cnt[my_round][my_wg] : pre_cnt) + (my_round == 0u ? rec_stat.x : 0u);
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
	if (t == 0) {
		s_n = 0;
		s_undone = 0;
		s_par = 0;
		s_serial = 0;
		s_viols = 0;
		s_rviol = 0;
		s_loose = 0;
		s_revived = 0;
	}
	// The scans over the request tags can be shared by all work-groups of the launch;
	// everything after them is work-group 0's (unless much is left).  The helpers only stay when
	// that pays -- judged by what every work-group reads alike: the rounds' flags and commit
	// counters (final: only the kernels before this one in the stream change them).
	if (t < 64u) {
		uint32_t c = r2_here ? ctl->committed[0][t].n + ctl->committed[1][t].n : pre_done;
		for (int off = 32; off > 0; off >>= 1)
			c += __shfl_down(c, off);
		if (t == 0)
			s_done12 = c + s_rec[0];
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
		// (LDS only: a listed work list always fits there -- kTlPer entries per lane of this work-group at most -- and
		// nothing reads the global list before it is written anew; a store on its way to it made the meeting
		// below wait a round trip)
		static_assert(2u * kTlGroups * kTlPer <= (uint32_t)kSortCap, "a listed work list fits the LDS list");
		if (!r2_here && my_cnt <= 2u) {
			// (a lane's first two entries came with the counts: the usual short list costs no round trip of its own)
			if (my_cnt > 0u)
				lkeys[my_base] = pre_e.x;
			if (my_cnt > 1u)
				lkeys[my_base + 1u] = pre_e.y;
		} else {
			for (uint32_t k = 0; k < my_cnt; k++)
				lkeys[my_base + k] = ctl->tl[my_round][my_wg][k];
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
	tv.sorted = S;
	tv.sorted_on = SORTED && sorted_hint != 0;      // (a constant false in the usual instantiation: the bins' code is not in it)
	uint32_t rounds_total = 0;
	// (flags_clean: nothing in this launch has run yet that could have recorded an order violation since the control
	// block's word was last known to be clear -- round 1 cleared it, nobody but this launch sets it: the word need
	// not be asked for, a round trip each time, by a launch that only replays a handful)
	bool flags_clean = true;
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
			flags_clean = false;
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
			if (m > (uint32_t)kTailReplayMax && max_rounds > 0) {
				flags_clean = false;
				left = tail_rounds<OP>(ctl, g, table, owner, cnt, altmin, reqc, state, keys, m, s_rbar, s_rmin, s_below, &s_progress,
						&s_par, &s_left, &s_dupw, max_rounds, &used, tv, rounds_total, reinterpret_cast<uint32_t *>(lreg));
			}
		}
		rounds_total += used;
		MKV_PHASE(4);
		bool replayed_list = false;
		if (left != 0 && (flags_clean || ld_agent32(&ctl->tflags) == 0u)) {
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
				replay_regions<OP>(table, g, reqc, state, lkeys, lreg, left, tv, &s_serial, &s_rviol);
				replayed_list = true;
			} else {
				flags_clean = false;
				replay_pending_scan<OP>(table, g, reqc, state, glist, lkeys, lo_idx, total, tv, &s_serial);
			}
		}
		// (a replay of the list alone, and none of its requests reported a violation -- the work-group has met behind
		// it, replay_regions: everything is done, and the stores of the replay need not have arrived for the epilogue)
		if (replayed_list && flags_clean && s_rviol == 0u) {
			MKV_PHASE(5);
			break;
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
						take_back(g, table, state, reqc, undo, chain_logs, k, st == ctag, tv, st);
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
						&s_ntrunc, tv);
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
			s_rviol = 0u;
		}
		flags_clean = true;
		__syncthreads();
		if (relisted == kNone)
			gather_pending(state, undo, reqc, glist, &s_n, lo_idx, total, 0u, 1u, s_loose != 0u);
		else if (t == 0)
			s_n = relisted;
		wg_sync();
		MKV_PHASE(3);
	}
	lds_sync();        // (the counts below are in LDS; what the replay stored to the table is on its way)
	tail_epilogue<false>(ctl, phase_ticks, rounds_total, total_all, s_par, s_serial, s_undone, s_viols, first_chunk, last_chunk,
			host_flag, s_rec);
}

__global__ void k_init_ctl(Ctl *ctl, unsigned long long first_stamp, uint32_t *host_flag)
{
	const uint32_t t = threadIdx.x;
	if (t == 0) {
		if (host_flag)
			host_flag[0] = host_flag[1] = host_flag[3] = host_flag[4] = host_flag[5] = 0;     // (in step with hf_last / hf_losers / hf_walk / hf_sorted below)
		ctl->stamp_base = ctl->stamp_cur = first_stamp;
		ctl->ep_pending = 0;
		ctl->total = 0;
		ctl->overflow = 0;
		ctl->any_loser = ctl->any_tail = 0;
		ctl->force_tail = 0;
		ctl->hint_stamp = 0;
		ctl->sorted_fail = 0;
		ctl->sorted_nlosers = 0;
		ctl->sorted_passes = ctl->sorted_fallbacks = 0;
		ctl->hf_sorted = 0;
		for (int k = 0; k < kShards; k++)
			ctl->committed[0][k].n = ctl->committed[1][k].n = ctl->committed[2][k].n = ctl->committed[3][k].n = 0;
		for (uint32_t k = 0; k < kTlGroups; k++)
			ctl->wg_rec[k].stat = make_uint4(0u, 0u, 0u, 0u);
		ctl->first_hard = kNone;
		ctl->last = Info{0, 0, 0, 0, 0, 0, kNone, 0, 0, 0, 0, 0, 0, 0};
		ctl->bar_chain = ctl->bar_class = 0;
		ctl->carry_lo = kNone;
		ctl->chain_n = 0;
		ctl->chain_ok = 0;
		ctl->any_hard = 0;
		ctl->violations = 0;
		ctl->hard_tail = 0;
		ctl->hf_last[0] = ctl->hf_last[1] = 0;
		ctl->hf_losers = 0;
		ctl->hf_walk = 0;
		for (int k = 0; k < 8; k++)
			ctl->dbg[k] = 0;
	}
	reset_tail_scratch(ctl);
	if (t < kMaxRegions)
		ctl->barrier[t] = kNone;
	if (t < (uint32_t)kShards)
		ctl->unplanned[t].n = 0u;
	if (t == 0)
		ctl->unplanned_base = ctl->r1_wide = ctl->r1_wide_next = 0u;
}

}  // namespace mkv
