// mkv_upd_round.h -- insert / delete, K2 / K3: the grid-wide rounds 1 and 2 (k_round; round 2 also as the tail launch's first phase)
// Part of the insert / delete translation unit (mkv_update.hip includes it; the kernels share the control
// block, the reservation helpers and the request cache, and the tail launch runs round 2's body, so they
// are compiled together).  The design is described at the top of mkv_update.hip and in DESIGN.md 5.2.
#pragma once
#include "mkv_upd_common.h"
#include "mkv_upd_sorted.h"

namespace mkv {

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

// ... and round 1 what it leaves to round 2: the first kLlPer per work-group (the count is in its record, Ctl::wg_rec)
__device__ __forceinline__ void loser_list_add(uint32_t *s_ll, volatile uint32_t *s_lln, uint32_t idx, bool add)
{
	// (once the list is full nobody queues on its counter: a pass of duplicates leaves a work-group a hundred losers)
	if (add && *s_lln < kLlPer) {
		const uint32_t at = atomicAdd(const_cast<uint32_t *>(s_lln), 1u);
		if (at < kLlPer)
			s_ll[at] = idx;
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
	static constexpr size_t ll = tl + sizeof(uint32_t) * kTlPer;                    // uint32_t[kLlPer]
	static constexpr size_t words = ll + sizeof(uint32_t) * kLlPer;                 // s_qn, s_fused, s_tln, s_lln, then the work-group's three counts
	static constexpr size_t bytes = words + 8u * sizeof(uint32_t);
};

// (block of nblocks: the work-group's place among those that share the round -- the launch's own for
// rounds launched by themselves)
// (SORTED: the instantiation for big passes, mkv_upd_apply.h -- the usual launches carry none of its code: a round 2
// that only scans 524 K tags took 8.3 us with it compiled in, 6.6 without, and the tail launch 18.1 against 17.0)
template <Op OP, int ROUND, int QCAP, bool SORTED = false>
__device__ __forceinline__ void round_body(const uint32_t block, const uint32_t nblocks_launched, unsigned char *lds, Ctl *ctl,
		const DevGeom &g, uint32_t *table,
		const unsigned long long *owner, unsigned long long *owner_next, const uint4 *reqc, uint8_t *state,
		Undo *undo, ChainLog *chain_logs, uint32_t lo_idx, uint32_t hi_idx, uint32_t req_cap, int record_hint,
		unsigned long long *emax, const unsigned long long *second, const uint32_t *lost, const Sorted S, const uint32_t *list_arg = nullptr,
		uint32_t list_n_arg = 0u)
{
	// A BIG pass (bit 9 of the hint: its requests went through the partition and k_apply, mkv_upd_apply.h; no round 1
	// ran): round 2 runs over the LIST of what k_apply left over, on as many work-groups as it takes.  Nobody of the
	// pass but those requests stands in a reservation table (they stand in this round's, on both their buckets); what
	// k_apply committed on a bucket is in its bin.  No groups of duplicates.
	const uint32_t *list = list_arg;
	uint32_t list_n = list_n_arg;
	const bool sorted = SORTED && ROUND == 2 && (record_hint & 512) != 0 && list == nullptr;
	if (sorted) {
		list = S.losers;
		list_n = ctl->sorted_nlosers;
		// (word 5 of the host's flags: the pass fell back -- a bin overflowed, everything was left over -- or left
		// over more than one request in 256: the host then runs the next passes the old way for a while.  Written
		// when it has something new to say, with the pass's number so that the host tells a new verdict from an old one)
		if (block == 0 && threadIdx.x == 0 && S.host_flag) {
			const uint32_t bad = (unsigned long long)list_n * 256ull > ctl->total ? 1u : 0u;
			if (bad || ctl->hf_sorted != 0u) {
				const uint32_t word = bad ? (((ctl->sorted_passes + ctl->sorted_fallbacks) << 1) | 1u) : 0u;
				S.host_flag[5] = word;
				ctl->hf_sorted = word;
			}
		}
	}
	// (list: the round's requests by index, at most a lane each -- one turn over them instead of the scan of the
	// pass's tags: round 2 run by the tail launch's first work-group over the few losers round 1 listed)
	// (lost: the mark pass's flags of the requests that are not the lowest registrant of their buckets, kPlanBit)
	// (second: the mark pass's second-lowest registrant per bucket, note_second)
	// (emax: a word per bucket -- the followers of a bucket's round-1 owner elect the last of them there,
	// kFollowBit; the tail's elections use the same words under its own, later stamps)
	// The first turn's request and tag are fetched before anything is known from the control block
	// (one round trip less on the critical path of a launch that is all latency); used if the
	// work-group's first index turns out to be what was guessed (no carried requests).
	const uint32_t pre_base = lo_idx + block * blockDim.x;
	const uint32_t pre_idx = pre_base + threadIdx.x;
	const bool pre_ok = pre_idx < req_cap && pre_idx >= lo_idx && list == nullptr;
	const uint8_t pre_st = pre_ok ? state[pre_idx] : (uint8_t)ST_DONE;
	const uint4 pre_rc = pre_ok ? reqc[pre_idx] : make_uint4(0u, 0u, 0u, 0u);
	// (round 1: the mark pass's plans stand -- bit 6 of the hint: the host knows of nothing that wrote the
	// table since -- so the request's plan and its flag travel with the request)
	const bool plans = ROUND == 1 && OP != OP_DELETE && (record_hint & 64) != 0 && lost != nullptr;
	const uint32_t pre_lost = plans && pre_ok ? lost[pre_idx] : 0u;
	Undo pre_plan;
	if (plans && pre_ok)
		pre_plan = undo[pre_idx];
	else
		pre_plan.word0 = pre_plan.old0 = pre_plan.word1 = pre_plan.old1 = kNoWord;
	const bool chains_here = (record_hint & 4) == 0;      // tests: every hard request left to the tail
	const bool groups_on = (record_hint & 8) == 0 && !sorted;        // Tuning::dup_groups = 0: duplicates of a key one per round, as before
	const bool help_on = (record_hint & 32) == 0;         // Tuning::self_help = 0: the second registrant waits for round 2, as before
	if (ROUND == 2 && ctl->any_loser == 0u)
		return;
	// (round 1 reads the stamp its mark pass published; its first work-group makes it the pass's stamp for the
	// kernels that follow and notes that the pass's epilogue is still to come: Ctl, head)
	const unsigned long long stamp1 = ROUND == 1 ? ctl->stamp_cur : ctl->stamp_base;
	if (ROUND == 1 && block == 0) {
		reset_tail_scratch(ctl);
		if (threadIdx.x == 0) {
			ctl->stamp_base = stamp1;
			ctl->ep_pending = 1u;
		}
		for (uint32_t q = threadIdx.x; q < kTlGroups; q += blockDim.x)
			ctl->tl_cnt[1][q] = 0u;                 // (round 2's places: its work-groups may have to share them, tail_list_flush)
	}
	const uint32_t count = hi_idx < ctl->total ? hi_idx : ctl->total;
	const unsigned long long stamp = stamp1 + (ROUND - 1);
	// (... unless a chain walked ahead in this pass's mark pass: it registers without flagging, prewalk_register)
	const bool plans_on = plans && ctl->hint_stamp != (uint32_t)stamp1;
	const uint32_t s32 = (uint32_t)stamp1;
	// A BIG pass is launched on a lane per request (bit 8 of the hint, launch_rounds): a turn that only carries out plans
	// is one round trip whatever it holds, and 2^19 requests on kTlGroups work-groups are four of them (zipf50 at load
	// factor 0.2: round 1 29.3 -> 25.0 us).  Four times the work-groups walking to alternate buckets and along chains at
	// once are SLOWER (load factor 0.38: 33.5 -> 49.6 us; profiles/r05_launch_floor.txt 12): unless the last such pass
	// was mostly plan-carrying (Ctl::r1_wide, moved here by this pass's mark pass) and this one's plans stand, the round
	// keeps to kTlGroups work-groups and the others leave.  Every work-group reads the same words and decides alike.
	uint32_t nblocks = nblocks_launched;
	if (ROUND == 1 && (record_hint & 256) != 0) {
		if (block == 0 && threadIdx.x < (uint32_t)kShards) {
			// (the verdict for the NEXT big pass: more than one request in 128 of this one without a plan -- its primary
			// bucket was full -- and a lane per request does not pay)
			static_assert(kShards == 64, "a shard a lane of a wave");
			uint32_t un = ctl->unplanned[threadIdx.x].n;
			for (int off = 32; off > 0; off >>= 1)
				un += __shfl_xor(un, off);
			if (threadIdx.x == 0) {
				const uint32_t added = un - ctl->unplanned_base;
				ctl->unplanned_base = un;
				ctl->r1_wide_next = plans_on && (unsigned long long)added * 128ull <= ctl->total ? 1u : 0u;
			}
		}
		if (nblocks > kTlGroups && !(ctl->r1_wide != 0u && plans_on)) {
			nblocks = kTlGroups;
			if (block >= nblocks)
				return;
		}
	}
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
	uint32_t *s_ll = reinterpret_cast<uint32_t *>(lds + Lds::ll);   // round 1: what this work-group leaves to round 2 (the first kLlPer)
	uint32_t &s_lln = s_words[3];
	if (threadIdx.x == 0) {
		s_qn = 0u;
		s_fused = 0u;
		s_tln = 0u;
		s_lln = 0u;
		s_words[4] = s_words[5] = s_words[6] = 0u;
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
#ifdef MKV_DIAGNOSTICS
	const unsigned long long dg_t0 = wall_clock64();
	unsigned long long dg_p2acc = 0;
	uint32_t dg_qmax = 0;
#endif
	uint32_t base = start + block * blockDim.x;
	// (list mode: this turn's first entry of the list.  Only the instantiation for big passes walks a list on several
	// work-groups and turns; the usual one takes its handful in ONE turn of one work-group, as it always did -- the
	// expressions are kept apart so that the usual kernels compile to what they were)
	uint32_t lpos = SORTED ? block * blockDim.x : 0u;
	uint32_t pf_a = 0u, pf_b = 0u;
	// (Fetching a turn's requests and tags a turn ahead was built and measured: no difference -- the turns of a
	// long pass are bound by the memory system's rate of random accesses, not by their round trips.)
	for (bool more = list ? (SORTED ? lpos < list_n : list_n != 0u) : (base < count && base >= start); ; ) {
		asm volatile("" ::"v"(pf_a), "v"(pf_b));       // (last turn's touches have long arrived)
		if (more) {
			const bool listed = list != nullptr && (SORTED ? lpos + threadIdx.x < list_n : threadIdx.x < list_n);
			const uint32_t idx = list ? (listed ? list[SORTED ? lpos + threadIdx.x : threadIdx.x] : 0u) : base + threadIdx.x;
			const bool in_range = list ? listed : idx < count;
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
			// Round 1, the usual request: it has a plan and nobody flagged it -- the lowest registrant of its
			// primary bucket, which is as the mark pass saw it: the plan's stores, no further load (kPlanBit)
			bool planned = false;
			if (ROUND == 1 && plans_on && pending && (rc.w & kPlanBit)) {
				const uint32_t lf = pre ? pre_lost : lost[idx];
				if (lf != s32) {
					Undo u;
					if (pre)
						u = pre_plan;
					else
						u = undo[idx];
					if (u.word1 == kNoWord) {
						table[u.word0] = r.loc;
					} else {
						table[u.word0] = r.sig;
						table[u.word1] = r.loc;
					}
					state[idx] = ST_ROUND1;
					ndone++;
					planned = true;
				}
			}
			if (pending && !planned) {
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
						// what phase 2 will ask about the alternate bucket first -- its reservation word, its
						// signature row -- is asked for NOW and looked at a turn later (pf_a / pf_b are only
						// there to keep the loads alive): beside the other work-groups' traffic a dependent
						// access costs 2 - 3 us, and phase 2 is a chain of them at the very end of the
						// work-group (a quarter of round 1's time for a handful of queued requests: diagnostics
						// build, phases 8 / 11 of mkv_debug_tail_phases).  Worth 0.5 - 0.7 % of a 95/5 step.
						pf_a = (uint32_t)owner[A];
						pf_b = table[sig_word(A, 0u)];
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
			else
				loser_list_add(s_ll, &s_lln, idx, lose);
			base += span;
			if (SORTED) {
				lpos += span;
				more = list ? lpos < list_n : (base < count && base >= start);
			} else {
				more = list == nullptr && base < count && base >= start;
			}
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
#ifdef MKV_DIAGNOSTICS
		const unsigned long long dg_p2 = wall_clock64();
#endif
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
				late = c1 != kNone && round_commit_used(state, reqc, undo, chain_logs, g, c1, A, &e1, sorted) && e1 > idx;
				// ... or by the bucket's second registrant helping itself there in round 1 (self-help above)
				const unsigned long long v3 = second[A];
				const uint32_t c3 = (v3 >> kIdxBits) == kStampMax - stamp1 ? (uint32_t)(v3 & ((1ull << kIdxBits) - 1ull)) : kNone;
				if (!late && c3 != kNone && c3 != c1 && c3 > idx)
					late = round_commit_used(state, reqc, undo, chain_logs, g, c3, A, &e1, sorted) && e1 > idx;
			}
			const bool own = live && !late && (A == P || owner[A] == key);
			// ... or by k_apply (a big pass): whatever it committed there of later requests is in the bin -- taken out,
			// the request put in its place and they behind it again, in registers (resolve_late_alternate)
			int res = 0;
			Undo ur;
			uint32_t dr0 = kNone, dr1 = kNone, dr2 = kNone, dr3 = kNone;
			ur.word0 = ur.word1 = kNoWord;
			ur.old0 = ur.old1 = 0u;
			if (ROUND == 2 && OP != OP_DELETE && sorted && own && lazy && A != P)
				res = resolve_late_alternate(table, g, S, state, r.sig, r.loc, idx, A, ur, dr0, dr1, dr2, dr3);
			late = late || res == 2;
			if (res == 1) {
				state[idx] = ST_ROUND2;
				undo[idx] = ur;
				ndone++;
				done = true;
				if ((dr0 & dr1 & dr2 & dr3) != kNone)
					ctl->any_tail = 1u;
			}
			if (sorted) {
				tail_list_add(s_tl, &s_tln, dr0, dr0 != kNone);
				tail_list_add(s_tl, &s_tln, dr1, dr1 != kNone);
				tail_list_add(s_tl, &s_tln, dr2, dr2 != kNone);
				tail_list_add(s_tl, &s_tln, dr3, dr3 != kNone);
			}
			if (own && res == 0) {
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
						try_chain<SORTED>(table, g, r, A, idx, owner_r1, stamp1, ctl, chain_logs, &log_slot, ROUND == 1,
								ROUND == 2 ? owner : nullptr, S.rec, S.heads, S.next, S.bin_shift, sorted, state);
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
					if (chained) {
						atomicAdd(&ctl->chain_ok, 1u);
						ctl->force_tail = 1u;           // (the epilogue reports the pass's share of chains to the host)
					}
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
			if (ROUND == 1)
				loser_list_add(s_ll, &s_lln, idx, lose);
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
#ifdef MKV_DIAGNOSTICS
		dg_p2acc += wall_clock64() - dg_p2;
		dg_qmax = qn > dg_qmax ? qn : dg_qmax;
#endif
		if (!more)
			break;
		__syncthreads();
		if (threadIdx.x == 0)
			s_qn = 0u;
		__syncthreads();
	}
#ifdef MKV_DIAGNOSTICS
	// (round 1: the slowest work-group's time in all, its share in phase 2, the longest queue -- in ticks of
	// 10 ns / requests; they come out as phases 8, 11, 15 of mkv_debug_tail_phases)
	if (ROUND == 1 && threadIdx.x == 0) {
		atomicMax(&ctl->dbg[0], (uint32_t)(wall_clock64() - dg_t0));
		atomicMax(&ctl->dbg[3], (uint32_t)dg_p2acc);
		atomicMax(&ctl->dbg[7], dg_qmax);
	}
#endif
	if (ROUND == 1 && nblocks <= kTlGroups) {
		// Round 1 with a place of its own per work-group (the usual launch): what it counted and what it leaves to
		// the tail go there in one record, no atomics on global memory -- 800 waves adding to the sharded counters and
		// 512 work-groups each storing a count cost a 52 K-request pass 0.9 us (profiles/r05_launch_floor.txt)
		uint32_t a = ndone + ndone_help, b = ndone_help, c = nlosers;
		for (int off = 32; off > 0; off >>= 1) {
			a += __shfl_down(a, off);
			b += __shfl_down(b, off);
			c += __shfl_down(c, off);
		}
		if ((threadIdx.x & 63u) == 0u) {
			if (a) atomicAdd(&s_words[4], a);
			if (b) atomicAdd(&s_words[5], b);
			if (c) atomicAdd(&s_words[6], c);
		}
		__syncthreads();
		const uint32_t n = s_tln;
		if (threadIdx.x < kTlPer && threadIdx.x < n)
			ctl->tl[0][block][threadIdx.x] = s_tl[threadIdx.x];
		if (threadIdx.x < kLlPer && threadIdx.x < s_lln)          // (s_lln may run a little past kLlPer; the count is in the record)
			ctl->wg_rec[block].ll[threadIdx.x] = s_ll[threadIdx.x];
		if (threadIdx.x == 0) {
			ctl->wg_rec[block].stat = make_uint4(n, s_words[4], s_words[5], s_words[6]);
			if (block == 0)
				ctl->tl_groups[0] = nblocks;
		}
		return;
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


template <Op OP, int ROUND, bool SORTED = false>
__global__ __launch_bounds__(256) void k_round(Ctl *ctl, DevGeom g, uint32_t *table,
		const unsigned long long *owner, unsigned long long *owner_next, const uint4 *reqc, uint8_t *state,
		Undo *undo, ChainLog *chain_logs, uint32_t lo_idx, uint32_t hi_idx, uint32_t req_cap, int record_hint,
		unsigned long long *emax, const unsigned long long *second, const uint32_t *lost, typename SortedArg<SORTED>::type Sarg)
{
	const Sorted S = sorted_of_arg(Sarg);
	__shared__ __align__(16) unsigned char lds[RoundLds<ROUND, kDeferCap>::bytes];
	round_body<OP, ROUND, kDeferCap, SORTED>(blockIdx.x, gridDim.x, lds, ctl, g, table, owner, owner_next, reqc, state, undo,
			chain_logs, lo_idx, hi_idx, req_cap, record_hint, emax, second, lost, S);
}

}  // namespace mkv
