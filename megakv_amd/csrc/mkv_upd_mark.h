// mkv_upd_mark.h -- insert / delete, K1: the mark pass (k_mark, and as extra work-groups of the search launches)
// Part of the insert / delete translation unit (mkv_update.hip includes it; the kernels share the control
// block, the reservation helpers and the request cache, and the tail launch runs round 2's body, so they
// are compiled together).  The design is described at the top of mkv_update.hip and in DESIGN.md 5.2.
#pragma once
#include "mkv_upd_common.h"
#include "mkv_upd_sorted.h"
#include "mkv_search_body.h"

namespace mkv {

// ---- K1: mark -----------------------------------------------------------------
// (undo, lost_flags: where the plans and the flags of kPlanBit go -- null: none are made)
// (lazy_free: Tuning::lazy_alt; bit 8: the last passes walked many eviction chains -- the walks ahead do not flag, prewalk_register)
// (GATHER: the blocks are the workers' of a scheduler cycle, found through gb -- src.nblk says how many)
template <Op OP, bool GATHER = false>
__device__ __forceinline__ void mark_body(uint32_t block, uint32_t nblocks, Ctl *ctl, const ReqSrc &src,
		const DevGeom &g, unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state, uint32_t req_cap,
		uint32_t lo_idx, uint32_t hi_idx, const uint32_t *table, uint32_t lazy_free, Undo *undo, uint32_t *lost_flags,
		const GatherBlocks &gb)
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
	// this pass's round-1 stamp (Ctl, head: the pass before may have ended without its epilogue)
	const unsigned long long stamp_prev = ctl->stamp_base;
	const uint32_t ep_pending = ctl->ep_pending;
	const unsigned long long stamp = stamp_prev + (ep_pending ? 3ull : 0ull);
	Lost lost;
	lost.flag = OP != OP_DELETE && undo ? lost_flags : nullptr;
	lost.s32 = (uint32_t)stamp;
	if (src.single) {
		if (t == 0) {
			const unsigned long long n = src.single_n > 0 ? (unsigned long long)src.single_n : 0ull;
			s_prefix[0] = 0;
			s_total = n > req_cap ? kNone : (uint32_t)n;
		}
	} else {
		for (uint32_t b = t; b < nblk; b += blockDim.x) {
			int c;
			if (GATHER && OP == OP_DELETE) {
				c = gb.del_cnt[b];                  // (a delete "block" is a worker's delete array)
				s_ptr[b] = gb.del_ptr[b];
			} else if (GATHER) {
				int w = 0;
				while (w + 1 < gb.nw && b >= gb.ins_first[w + 1])
					w++;
				const uint32_t k = b - gb.ins_first[w];
				c = gb.ins_cnts[w][k];
				s_ptr[b] = gb.ins_ptrs[w][k];
			} else {
				c = src.blk_elem_num[(size_t)b * src.cnt_stride];
				s_ptr[b] = src.blk_input[b];        // fetched beside the counts: one round trip
			}
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
		ctl->r1_wide = ctl->r1_wide_next;      // (the verdict of the last big pass's round 1, for this pass's: Ctl::unplanned)
		ctl->stamp_cur = stamp;
		ctl->total = total;
		ctl->overflow = over ? 1u : 0u;
		// The tail launch of this pass leaves at once if the rounds leave it nothing -- unless its epilogue has
		// something to tell the host (a dropped batch to count, a flag word that may have to change back, round
		// stamps running low)
		ctl->force_tail = (over || ctl->hf_last[0] != 0u || ctl->hf_last[1] != 0u || ctl->hf_losers != 0u || ctl->hf_walk != 0u ||
				stamp > kStampMax - kStampReserve) ? 1u : 0u;
		if (ep_pending) {
			// ... in which case what the epilogue would have reset is reset here
			ctl->first_hard = kNone;
			ctl->chain_n = 0u;
			ctl->chain_ok = 0u;
			ctl->any_loser = 0u;        // (a handful of losers that round 2 saw through, k_tail)
		}
	}
	if (block == 0 && t < (uint32_t)kShards) {
		// commit counters of the rounds that follow (the tail reads them: statistics, and how
		// much it will have to gather); cleared here, by the kernel before them in the stream
		ctl->committed[0][t].n = 0;
		ctl->committed[1][t].n = 0;
		ctl->committed[2][t].n = 0;
		ctl->committed[3][t].n = 0;
	}
	// (round 1's places in the rounds' lists for the tail: its work-groups share them, adding to the counts,
	// when the round runs on more than kTlGroups of them -- tuning "round_blocks"; round 2's are cleared by
	// round 1.  Same rule: by a kernel before the one that counts.)
	if (block == 0)
		for (uint32_t q = t; q < kTlGroups; q += blockDim.x) {
			ctl->tl_cnt[0][q] = 0u;
			ctl->wg_rec[q].stat = make_uint4(0u, 0u, 0u, 0u);
		}
	const uint32_t end = hi_idx < total ? hi_idx : total;
	// requests an earlier chunk pass of this call left unresolved ride along: they have
	// lower indices, so they win every reservation they share with this chunk's requests
	const uint32_t carry = ctl->carry_lo;
	const uint32_t start = carry < lo_idx ? carry : lo_idx;
	// whole waves iterate together: the reservation de-duplication is a wave operation
	const uint32_t span = nblocks * blockDim.x;
	const uint32_t first = start + block * blockDim.x;          // this block's first index
	uint32_t nun = 0u;      // this lane's requests without a plan
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
		bool lazy = false, planned = false;
		// (the whole bucket: the same 64-byte request as its signature row -- one request, the quad's four lanes a piece each)
		const bool look = OP != OP_DELETE && live && table && (fresh || OP == OP_CUCKOO);
		const Bkt kP = OP != OP_DELETE && table ? load_bucket_quad(table, look ? P : kNone) : Bkt();
		if (look) {
			plo = kP.s0;
			phi = kP.s1;
			lazy = fresh && (lazy_free & 0xffu) != 0u && A != P && (uint32_t)__popc(eq_mask(plo, phi, 0u)) >= (lazy_free & 0xffu);
			// what the request would do in P as the table stands (kPlanBit)
			Undo u;
			if (fresh && undo && plan_primary(P, kP, rq.sig, &u)) {
				undo[idx] = u;
				planned = true;
			}
			nun += fresh && undo && !planned ? 1u : 0u;
		}
		if (fresh)
			reqc[idx] = make_uint4(rq.sig, rq.hash, rq.loc, (lazy ? kLazyBit : 0u) | (planned ? kPlanBit : 0u));
		const unsigned long long key = owner_key(stamp, idx);
		// (every registration notes the bucket's second-lowest registrant, note_second, and flags whoever is
		// not a bucket's lowest one, flag_lost)
		if (use_fold) {
			mark_fold_min_dedup<kMarkFoldLog2>(fold, s_tk2, owner, second, P, key, live, lost);
			mark_fold_min_dedup<kMarkFoldLog2>(fold, s_tk2, owner, second, A, key, live && A != P && !lazy, lost);
			mark_fold_flush<kMarkFoldLog2>(fold, s_tk2, owner, second, lost);
		} else {
			register_dedup(owner, second, P, key, live, lost);
			register_dedup(owner, second, A, key, live && A != P && !lazy, lost);
		}
		if (OP == OP_CUCKOO && live && table && !lazy)
			prewalk_register(table, g, rq, P, A, owner, key, plo, phi, &ctl->hint_stamp, lost, (lazy_free & 0x100u) == 0u && lost.flag);
	}
	if (OP != OP_DELETE && undo)
		count_add(&ctl->unplanned[0].n, nun);       // (an atomic per wave that has any: next to none below load factor 0.3)
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
template <bool SORTED>
__device__ bool try_chain(uint32_t *table, const DevGeom &g, const Req &r, uint32_t A, uint32_t idx,
		unsigned long long *owner1, unsigned long long stamp1, Ctl *ctl, ChainLog *logs, uint32_t *log_slot,
		bool own_ok, const unsigned long long *owner2, const uint4 *s_rec, const uint16_t *s_heads, const uint16_t *s_next,
		uint32_t s_bin_shift, bool sorted_flag, const uint8_t *state)
{
	const bool sorted_on = SORTED && sorted_flag;
	// (the bins by their parts: a reference to the kernel's argument would put it into scratch memory for this call)
	Sorted S = {};
	S.rec = const_cast<uint4 *>(s_rec);
	S.heads = const_cast<uint16_t *>(s_heads);
	S.next = const_cast<uint16_t *>(s_next);
	S.bin_shift = s_bin_shift;
	// (sorted_on: a big pass, mkv_upd_apply.h -- nobody but the requests k_apply left over stands in a reservation
	// table: they all stand in round 2's, owner2; what k_apply COMMITTED on a bucket the chain goes on to is in its bin,
	// and a commit of a later request there is an order violation: the chain gives up and is left to the tail)
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
		if (sorted_on) {
			bool later = false;
			sorted_for_each_on(S, g.hash_mask, b, [&](const uint4 &rs, uint32_t) {
				later = rs.w > idx && state[rs.w] == ST_ROUND1;
				return later;
			});
			if (later) {
				ok = false;
				break;
			}
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
		uint32_t hi_idx, const uint32_t *table, int lazy_on, Undo *undo, uint32_t *lost)
{
	mark_body<OP>(blockIdx.x, gridDim.x, ctl, src, g, owner, second, reqc, state, req_cap, lo_idx, hi_idx, table, (uint32_t)lazy_on,
			undo, lost, GatherBlocks());
}

// mkv_cycle: the mark pass never touches the table, so it rides in the search
// launch as extra work-groups (horizontal fusion): its latency-bound atomics hide
// under the bandwidth-bound probes and one kernel boundary disappears.
template <Op OP>
__global__ __launch_bounds__(256) void k_search_mark(const uint2 *__restrict__ in,
		uint32_t *__restrict__ out, const uint4 *__restrict__ table, int64_t n, uint32_t search_blocks,
		Ctl *ctl, ReqSrc src, DevGeom g, unsigned long long *owner, unsigned long long *second, uint4 *reqc, uint8_t *state,
		uint32_t req_cap, uint32_t hi_idx, int lazy_on, Undo *undo, uint32_t *lost)
{
	// mark blocks first: they are few and latency-bound, let them start early
	const uint32_t mark_blocks = gridDim.x - search_blocks;
	if (blockIdx.x < mark_blocks)
		mark_body<OP>(blockIdx.x, mark_blocks, ctl, src, g, owner, second, reqc, state, req_cap, 0u, hi_idx,
				reinterpret_cast<const uint32_t *>(table), (uint32_t)lazy_on, undo, lost, GatherBlocks());
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
		uint32_t self, uint32_t *__restrict__ self_out, int lazy_on, Undo *undo, uint32_t *lost)
{
	const uint32_t mark_blocks = gridDim.x - search_blocks;
	if (blockIdx.x < mark_blocks)
		mark_body<OP>(blockIdx.x, mark_blocks, ctl, src, g, owner, second, reqc, state, req_cap, 0u, hi_idx,
				reinterpret_cast<const uint32_t *>(table), (uint32_t)lazy_on, undo, lost, GatherBlocks());
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
		const ChainLog *chain_logs, const DevGeom &g, uint32_t c, uint32_t b, uint32_t *eff, bool sorted_on = false)
{
	// (sorted_on: the pass may hold commits of k_apply -- a constant in every instantiation, so the usual ones carry no check)
	if (!tag_logged(st))
		return false;                       // pending, done by the tail, or taken back (or selected already)
	*eff = (sh.w & kLeadBit) ? (sh.w >> 8) : c;
	const uint32_t P = sh.y & g.hash_mask;
	if (b == P)
		return true;
	if (sorted_on && sorted_commit(st, sh.w))
		return false;                       // (a commit of k_apply, big passes: settled in its primary bucket; its record is in its bin)
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
		const ChainLog *chain_logs, const DevGeom &g, uint32_t c, uint32_t b, uint32_t *eff, bool sorted_on = false)
{
	const uint32_t st = state[c];
	if (!tag_logged(st))
		return false;
	return round_commit_used_pre(st, reqc[c], undo, chain_logs, g, c, b, eff, sorted_on);
}

}  // namespace mkv
