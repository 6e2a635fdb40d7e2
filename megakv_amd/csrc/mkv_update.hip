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
// duplicate ("hard") walks on to data-dependent buckets.  Requests therefore
// commute unless they share P/A, and the work proceeds in reservation rounds:
//
//   mark    every pending request writes its canonical index into owner[P] and
//           owner[A] with a 64-bit atomicMin (per-bucket side table in HBM, one
//           random atomic per bucket touched; a round stamp in the high word
//           makes stale entries lose, so the tables are never cleared);
//   decide  (cuckoo only) finds, per closed region, the lowest pending index
//           that is hard, or that might become hard before its turn (it lost a
//           reservation, holds no duplicate, and the empty slots of P and A do
//           not outnumber the other requests registered there).  Nothing at or
//           beyond that index may commit this round: an eviction chain may
//           reach any bucket of its region;
//   commit  every request that owns all its buckets (no earlier pending
//           request touches them) and lies below the region's barrier executes
//           against the table with plain loads/stores -- it has exclusive
//           use of those buckets in this round -- and the rest re-register
//           for the next round in the other owner table;
//   tail    after a fixed number of rounds one work-group sorts what is left by
//           canonical index (LDS bitonic sort) and replays it in order,
//           eviction chains included.
//
// Why that equals the canonical order is argued in DESIGN.md ("insert:
// proof sketch").  No host synchronisation: the request counts are read on
// the device (the reference passes them as a device array, gpu_hash.cu:236-237)
// and every kernel grid-strides over them.
//
// The per-lane atomicCAS the north star suggests for slot claims is not needed:
// exclusivity comes from the reservation, so a commit is two plain stores.
#include <mutex>
#include <unordered_map>

#include "mkv_common.h"

namespace mkv {

constexpr int kMaxBlocks = 1024;   // insert blocks per call
constexpr int kMaxRegions = 32;
constexpr int kMaxRounds = 6;
constexpr int kSortCap = 8192;     // tail requests sorted in LDS; beyond: state scan
constexpr uint32_t kGridAll = 1024;
constexpr uint32_t kGridList = 512;

enum Op : uint32_t { OP_CUCKOO = 0, OP_2CHOICE = 1, OP_DELETE = 2 };

struct Ctl {
	uint32_t total;
	uint32_t overflow;
	uint32_t list_n[2];
	uint32_t committed[kMaxRounds];
	uint32_t drained;
	uint32_t first_hard;
	uint32_t barrier[kMaxRounds][kMaxRegions];
	uint32_t blk_prefix[kMaxBlocks + 1];
};

struct Req {
	uint32_t sig, hash, loc;
};

// Where the requests live: insert = device array of per-block device arrays
// (reference ABI), delete = one array.
struct ReqSrc {
	const uint32_t *const *blk_input;
	const uint32_t *single;
	uint32_t nblk;
};

__device__ __forceinline__ Req load_req(const ReqSrc &src, const Ctl *ctl, uint32_t idx)
{
	const uint32_t *p;
	if (src.single) {
		p = src.single + 3ull * idx;
	} else {
		uint32_t b = 0;
		while (b + 1 < src.nblk && ctl->blk_prefix[b + 1] <= idx)
			b++;
		p = src.blk_input[b] + 3ull * (idx - ctl->blk_prefix[b]);
	}
	Req r;
	r.sig = p[0];
	r.hash = p[1];
	r.loc = p[2];
	return r;
}

// ---- table access helpers ---------------------------------------------------
__device__ __forceinline__ void load_sig_row(const uint32_t *table, uint32_t b, uint4 &lo, uint4 &hi)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(table + (uint64_t)b * kBucketWords);
	lo = p[0];
	hi = p[1];
}

__device__ __forceinline__ void load_loc_row(const uint32_t *table, uint32_t b, uint4 &lo, uint4 &hi)
{
	const uint4 *p = reinterpret_cast<const uint4 *>(table + (uint64_t)b * kBucketWords);
	lo = p[2];
	hi = p[3];
}

__device__ __forceinline__ void store_sig(uint32_t *table, uint32_t b, uint32_t slot, uint32_t v)
{
	table[(uint64_t)b * kBucketWords + slot] = v;
}

__device__ __forceinline__ void store_loc(uint32_t *table, uint32_t b, uint32_t slot, uint32_t v)
{
	table[(uint64_t)b * kBucketWords + kSlots + slot] = v;
}

// Duplicate -> overwrite loc; else empty -> claim; returns true when finished.
// `pref_sig` is the ORIGINAL request's signature (slot preference, ref :301).
__device__ __forceinline__ bool try_place(uint32_t *table, uint32_t b, uint32_t sig, uint32_t loc,
		uint32_t pref_sig)
{
	uint4 lo, hi;
	load_sig_row(table, b, lo, hi);
	uint32_t m = eq_mask(lo, hi, sig);
	if (m) {
		store_loc(table, b, (uint32_t)__ffs((int)m) - 1u, loc);
		return true;
	}
	uint32_t e = eq_mask(lo, hi, 0u);
	if (e) {
		uint32_t s = preferred_slot(e, pref_sig);
		store_sig(table, b, s, sig);
		store_loc(table, b, s, loc);
		return true;
	}
	return false;
}

// delete in one bucket: zero the signature of every slot holding (sig, loc);
// returns whether any matched (ref gpu_hash.cu:459-468).
__device__ __forceinline__ bool delete_in(uint32_t *table, uint32_t b, uint32_t sig, uint32_t loc)
{
	uint4 slo, shi, llo, lhi;
	load_sig_row(table, b, slo, shi);
	uint32_t m = eq_mask(slo, shi, sig);
	if (!m)
		return false;
	load_loc_row(table, b, llo, lhi);
	m &= eq_mask(llo, lhi, loc);
	uint32_t mm = m;
	while (mm) {
		uint32_t s = (uint32_t)__ffs((int)mm) - 1u;
		store_sig(table, b, s, 0u);
		mm &= mm - 1u;
	}
	return m != 0u;
}

// A request whose reach is static.  Returns false (nothing written) only for a
// cuckoo request that turns out to need an eviction chain.
template <Op OP>
__device__ __forceinline__ bool apply_static(uint32_t *table, const DevGeom &g, const Req &r,
		uint32_t P, uint32_t A)
{
	if (OP == OP_DELETE) {
		if (!delete_in(table, P, r.sig, r.loc))
			delete_in(table, A, r.sig, r.loc);
		return true;
	}
	if (try_place(table, P, r.sig, r.loc, r.sig))
		return true;
	if (try_place(table, A, r.sig, r.loc, r.sig))
		return true;
	if (OP == OP_2CHOICE) {
		// no room: signature alone replaces slot sig&7, stale loc stays (ref :197-209)
		store_sig(table, A, r.sig & 7u, r.sig);
		return true;
	}
	return false;
}

// One request run to completion, eviction chain included (tail only).
template <Op OP>
__device__ void apply_serial(uint32_t *table, const DevGeom &g, const Req &r)
{
	const uint32_t P = r.hash & g.hash_mask;
	const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
	if (apply_static<OP>(table, g, r, P, A))
		return;
	// cuckoo, P and A full without duplicate (ref :357-423)
	uint32_t sig = r.sig, loc = r.loc, b = A;
	const uint32_t v = r.sig & 7u;          // victim slot: ORIGINAL signature (:360)
	for (uint32_t k = 0;; k++) {
		const uint32_t vsig = table[(uint64_t)b * kBucketWords + v];
		const uint32_t vloc = table[(uint64_t)b * kBucketWords + kSlots + v];
		store_sig(table, b, v, sig);
		store_loc(table, b, v, loc);
		if (k >= g.max_cuckoo)
			return;                         // victim dropped (:414-423)
		sig = vsig;
		loc = vloc;
		b = alt_bucket(r.hash, sig, g.hash_mask, g.blk_mask);   // ORIGINAL hash (:334)
		if (try_place(table, b, sig, loc, r.sig))
			return;
	}
}

// ---- reservation tables -----------------------------------------------------
__device__ __forceinline__ void bump(unsigned long long *cnt, uint32_t b, uint32_t stamp_inc)
{
	atomicMax(&cnt[b], (unsigned long long)stamp_inc << 32);
	atomicAdd(&cnt[b], 1ull);
}

__device__ __forceinline__ uint32_t read_cnt(const unsigned long long *cnt, uint32_t b, uint32_t stamp_inc)
{
	unsigned long long v = cnt[b];
	return (uint32_t)(v >> 32) == stamp_inc ? (uint32_t)v : 0u;
}

template <Op OP>
__device__ __forceinline__ void register_req(unsigned long long *owner, unsigned long long *cnt,
		uint32_t P, uint32_t A, uint32_t idx, uint32_t stamp_min, uint32_t stamp_inc)
{
	const unsigned long long key = ((unsigned long long)stamp_min << 32) | idx;
	atomicMin(&owner[P], key);
	if (A != P)
		atomicMin(&owner[A], key);
	if (OP == OP_CUCKOO) {
		bump(cnt, P, stamp_inc);
		if (A != P)
			bump(cnt, A, stamp_inc);
	}
}

__device__ __forceinline__ bool is_clean(const unsigned long long *owner, uint32_t P, uint32_t A,
		uint32_t idx, uint32_t stamp_min)
{
	const unsigned long long key = ((unsigned long long)stamp_min << 32) | idx;
	return owner[P] == key && (A == P || owner[A] == key);
}

// wave-aggregated append; returns nothing, order inside the list is arbitrary
__device__ __forceinline__ void list_append(uint32_t *list, uint32_t *list_n, bool want, uint32_t idx)
{
	const unsigned long long mask = __ballot(want);
	if (!mask)
		return;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
	uint32_t base = 0;
	if (lane == leader)
		base = atomicAdd(list_n, (uint32_t)__popcll(mask));
	base = __shfl(base, (int)leader);
	if (want)
		list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = idx;
}

__device__ __forceinline__ void count_add(uint32_t *ctr, bool flag)
{
	const unsigned long long mask = __ballot(flag);
	if (mask && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)mask) - 1u)
		atomicAdd(ctr, (uint32_t)__popcll(mask));
}

// ---- kernels ----------------------------------------------------------------
__global__ void k_prep(Ctl *ctl, const int *blk_elem_num, int nblk, long long single_n,
		uint32_t req_cap)
{
	const uint32_t t = threadIdx.x;
	if (t == 0) {
		unsigned long long sum = 0;
		ctl->blk_prefix[0] = 0;
		if (blk_elem_num) {
			for (int b = 0; b < nblk; b++) {
				int c = blk_elem_num[b];
				sum += c > 0 ? (unsigned long long)c : 0ull;
				ctl->blk_prefix[b + 1] = (uint32_t)(sum > 0xfffffffeull ? 0xfffffffeull : sum);
			}
		} else {
			sum = single_n > 0 ? (unsigned long long)single_n : 0ull;
		}
		const bool over = sum > req_cap;
		ctl->overflow = over ? 1u : 0u;
		ctl->total = over ? 0u : (uint32_t)sum;
		ctl->list_n[0] = 0;
		ctl->list_n[1] = 0;
		ctl->drained = 0;
		ctl->first_hard = kNone;
	}
	for (uint32_t i = t; i < kMaxRounds * kMaxRegions; i += blockDim.x)
		(&ctl->barrier[0][0])[i] = kNone;
	if (t < kMaxRounds)
		ctl->committed[t] = 0;
}

// round 1: every request registers
template <Op OP>
__global__ __launch_bounds__(256) void k_mark_all(Ctl *ctl, ReqSrc src, DevGeom g,
		unsigned long long *owner, unsigned long long *cnt, uint8_t *state,
		uint32_t stamp_min, uint32_t stamp_inc)
{
	const uint32_t total = ctl->total;
	for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total;
			idx += gridDim.x * blockDim.x) {
		const Req r = load_req(src, ctl, idx);
		if (OP != OP_DELETE && r.sig == 0u && r.loc == 0u) {
			state[idx] = 1;                 // skipped by the reference too (:259-262)
			continue;
		}
		state[idx] = 0;
		const uint32_t P = r.hash & g.hash_mask;
		const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
		register_req<OP>(owner, cnt, P, A, idx, stamp_min, stamp_inc);
	}
}

// cuckoo: per-region barrier index of this round
template <bool FROM_LIST>
__global__ __launch_bounds__(256) void k_decide(Ctl *ctl, ReqSrc src, DevGeom g,
		const uint32_t *table, const unsigned long long *owner, const unsigned long long *cnt,
		const uint8_t *state, const uint32_t *list, int list_sel, int round,
		uint32_t stamp_min, uint32_t stamp_inc)
{
	const uint32_t count = FROM_LIST ? ctl->list_n[list_sel] : ctl->total;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count;
			i += gridDim.x * blockDim.x) {
		uint32_t idx = i;
		if (FROM_LIST)
			idx = list[i];
		else if (state[idx])
			continue;
		const Req r = load_req(src, ctl, idx);
		const uint32_t P = r.hash & g.hash_mask;
		const uint32_t A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
		uint4 lo, hi;
		load_sig_row(table, P, lo, hi);
		if (eq_mask(lo, hi, r.sig))
			continue;                       // duplicate in P: finishes in P whenever it runs
		const uint32_t eP = (uint32_t)__popc(eq_mask(lo, hi, 0u));
		const bool clean = is_clean(owner, P, A, idx, stamp_min);
		if (clean && eP)
			continue;
		uint32_t eA = 0;
		if (A != P) {
			load_sig_row(table, A, lo, hi);
			if (eq_mask(lo, hi, r.sig))
				continue;                   // duplicate in A: never goes past A
			eA = (uint32_t)__popc(eq_mask(lo, hi, 0u));
		}
		bool bad;
		if (clean) {
			bad = (eP + eA) == 0u;          // hard right now
			if (bad)
				atomicMin(&ctl->first_hard, idx);
		} else {
			// other requests registered on P / A can each take one empty slot
			uint32_t others = read_cnt(cnt, P, stamp_inc) - 1u;
			if (A != P)
				others += read_cnt(cnt, A, stamp_inc) - 1u;
			bad = (eP + eA) < 1u + others;
		}
		if (bad)
			atomicMin(&ctl->barrier[round][P >> g.region_shift], idx);
	}
}

template <Op OP, bool FROM_LIST, bool MARK_NEXT>
__global__ __launch_bounds__(256) void k_commit(Ctl *ctl, ReqSrc src, DevGeom g, uint32_t *table,
		const unsigned long long *owner, unsigned long long *owner_next, unsigned long long *cnt,
		uint8_t *state, const uint32_t *list, uint32_t *list_next, int list_sel, int round,
		uint32_t stamp_min, uint32_t stamp_min_next, uint32_t stamp_inc_next)
{
	const uint32_t count = FROM_LIST ? ctl->list_n[list_sel] : ctl->total;
	const uint32_t span = gridDim.x * blockDim.x;
	// all lanes of a wave iterate together so the wave-wide appends stay convergent
	const uint32_t iters = (count + span - 1) / span;
	for (uint32_t it = 0; it < iters; it++) {
		const uint32_t i = it * span + blockIdx.x * blockDim.x + threadIdx.x;
		bool pending = i < count;
		uint32_t idx = i;
		if (pending) {
			if (FROM_LIST)
				idx = list[i];
			else
				pending = state[idx] == 0;
		}
		bool done = false;
		uint32_t P = 0, A = 0;
		if (pending) {
			const Req r = load_req(src, ctl, idx);
			P = r.hash & g.hash_mask;
			A = alt_bucket(r.hash, r.sig, g.hash_mask, g.blk_mask);
			bool ok = is_clean(owner, P, A, idx, stamp_min);
			if (OP == OP_CUCKOO && ok)
				ok = idx < ctl->barrier[round][P >> g.region_shift];
			if (ok)
				done = apply_static<OP>(table, g, r, P, A);
			if (done)
				state[idx] = 1;
		}
		const bool left = pending && !done;
		if (MARK_NEXT && left)
			register_req<OP>(owner_next, cnt, P, A, idx, stamp_min_next, stamp_inc_next);
		list_append(list_next, &ctl->list_n[list_sel ^ 1], left, idx);
		count_add(&ctl->committed[round], done);
	}
}

// resets the size of the list the NEXT commit will append to
__global__ void k_reset_list(Ctl *ctl, int which)
{
	ctl->list_n[which] = 0;
}

template <Op OP>
__global__ __launch_bounds__(1024) void k_tail(Ctl *ctl, ReqSrc src, DevGeom g, uint32_t *table,
		uint8_t *state, const uint32_t *list, int list_sel)
{
	__shared__ uint32_t keys[kSortCap];
	const uint32_t m = ctl->list_n[list_sel];
	if (m == 0)
		return;
	const uint32_t t = threadIdx.x;
	if (m <= (uint32_t)kSortCap) {
		uint32_t n2 = 1;
		while (n2 < m)
			n2 <<= 1;
		for (uint32_t i = t; i < n2; i += blockDim.x)
			keys[i] = i < m ? list[i] : kNone;
		__syncthreads();
		for (uint32_t k = 2; k <= n2; k <<= 1) {
			for (uint32_t j = k >> 1; j > 0; j >>= 1) {
				for (uint32_t i = t; i < n2; i += blockDim.x) {
					const uint32_t ixj = i ^ j;
					if (ixj > i) {
						const uint32_t a = keys[i], b = keys[ixj];
						const bool up = (i & k) == 0;
						if ((a > b) == up) {
							keys[i] = b;
							keys[ixj] = a;
						}
					}
				}
				__syncthreads();
			}
		}
		if (t == 0) {
			for (uint32_t i = 0; i < m; i++) {
				const uint32_t idx = keys[i];
				const Req r = load_req(src, ctl, idx);
				apply_serial<OP>(table, g, r);
				state[idx] = 1;
			}
			ctl->drained = m;
		}
	} else if (t == 0) {
		// too many leftovers for LDS: walk the state array in canonical order
		const uint32_t total = ctl->total;
		uint32_t d = 0;
		for (uint32_t idx = 0; idx < total; idx++) {
			if (state[idx])
				continue;
			const Req r = load_req(src, ctl, idx);
			apply_serial<OP>(table, g, r);
			state[idx] = 1;
			d++;
		}
		ctl->drained = d;
	}
}

// ---- per-stream workspace -----------------------------------------------------
struct Workspace {
	int device = -1;
	uint64_t nbuckets = 0;
	int64_t req_cap = 0;
	unsigned long long *owner[2] = {nullptr, nullptr};
	unsigned long long *cnt = nullptr;
	uint8_t *state = nullptr;
	uint32_t *list[2] = {nullptr, nullptr};
	Ctl *ctl = nullptr;
	uint32_t round_counter = 1;
	int last_rounds = 0;
};

static std::mutex g_ws_mutex;
static std::unordered_map<uint64_t, Workspace *> g_ws;   // key: device<<56 ^ stream

static void free_ws(Workspace *w)
{
	for (int i = 0; i < 2; i++) {
		if (w->owner[i]) (void)hipFree(w->owner[i]);
		if (w->list[i]) (void)hipFree(w->list[i]);
	}
	if (w->cnt) (void)hipFree(w->cnt);
	if (w->state) (void)hipFree(w->state);
	if (w->ctl) (void)hipFree(w->ctl);
	*w = Workspace();
}

static int ensure_ws(Workspace **out, uint64_t nbuckets, int64_t req_cap, hipStream_t stream)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	const uint64_t key = ((uint64_t)dev << 56) ^ (uint64_t)(uintptr_t)stream;
	std::lock_guard<std::mutex> lock(g_ws_mutex);
	Workspace *&w = g_ws[key];
	if (!w)
		w = new Workspace();
	if (w->device == dev && w->nbuckets >= nbuckets && w->req_cap >= req_cap) {
		*out = w;
		return MKV_OK;
	}
	const uint64_t nb = nbuckets > w->nbuckets ? nbuckets : w->nbuckets;
	const int64_t rc = req_cap > w->req_cap ? req_cap : w->req_cap;
	// growing: finish what the old buffers may still be used by
	if (w->device >= 0)
		(void)hipStreamSynchronize(stream);
	free_ws(w);
	w->device = dev;
	w->nbuckets = nb;
	w->req_cap = rc;
	bool ok = true;
	for (int i = 0; i < 2 && ok; i++) {
		ok = ok && hipMalloc(&w->owner[i], nb * 8) == hipSuccess;
		ok = ok && hipMalloc(&w->list[i], (size_t)rc * 4) == hipSuccess;
	}
	ok = ok && hipMalloc(&w->cnt, nb * 8) == hipSuccess;
	ok = ok && hipMalloc(&w->state, (size_t)rc) == hipSuccess;
	ok = ok && hipMalloc(&w->ctl, sizeof(Ctl)) == hipSuccess;
	if (ok) {
		// owner entries start at "no owner" (all ones); counters at zero
		ok = ok && hipMemsetAsync(w->owner[0], 0xff, nb * 8, stream) == hipSuccess;
		ok = ok && hipMemsetAsync(w->owner[1], 0xff, nb * 8, stream) == hipSuccess;
		ok = ok && hipMemsetAsync(w->cnt, 0, nb * 8, stream) == hipSuccess;
		ok = ok && hipMemsetAsync(w->ctl, 0, sizeof(Ctl), stream) == hipSuccess;
	}
	if (!ok) {
		free_ws(w);
		return MKV_ENOMEM;
	}
	w->round_counter = 1;
	*out = w;
	return MKV_OK;
}

static int default_rounds()
{
	return tuning().update_rounds;
}

static int64_t default_req_cap()
{
	return (int64_t)tuning().req_cap;
}

template <Op OP>
static int run_update(const DevGeom &g, uint32_t *table, const ReqSrc &src, const int *blk_elem_num,
		int64_t single_n, hipStream_t stream)
{
	Workspace *w = nullptr;
	int64_t need = default_req_cap();
	if (single_n > need)
		need = single_n;
	int rc = ensure_ws(&w, (uint64_t)g.hash_mask + 1, need, stream);
	if (rc != MKV_OK)
		return rc;
	const int rounds = default_rounds();
	w->last_rounds = rounds;

	hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, stream, w->ctl, blk_elem_num, (int)src.nblk,
			(long long)single_n, (uint32_t)(w->req_cap > 0xfffffff0ll ? 0xfffffff0ll : w->req_cap));

	uint32_t stamp_inc = w->round_counter;
	uint32_t stamp_min = kNone - stamp_inc;
	hipLaunchKernelGGL((k_mark_all<OP>), dim3(kGridAll), dim3(256), 0, stream, w->ctl, src, g,
			w->owner[0], w->cnt, w->state, stamp_min, stamp_inc);

	for (int r = 0; r < rounds; r++) {
		const bool first = r == 0;
		const bool last = r == rounds - 1;
		const int set = r & 1;          // owner table of this round
		const int lsel = r & 1;         // list read by this round (r > 0); appends go to lsel^1
		const uint32_t next_inc = stamp_inc + 1;
		const uint32_t next_min = kNone - next_inc;
		const dim3 grid(first ? kGridAll : kGridList), block(256);
		if (OP == OP_CUCKOO) {
			if (first)
				hipLaunchKernelGGL((k_decide<false>), grid, block, 0, stream, w->ctl, src, g, table,
						w->owner[set], w->cnt, w->state, w->list[lsel], lsel, r, stamp_min, stamp_inc);
			else
				hipLaunchKernelGGL((k_decide<true>), grid, block, 0, stream, w->ctl, src, g, table,
						w->owner[set], w->cnt, w->state, w->list[lsel], lsel, r, stamp_min, stamp_inc);
		}
		if (!first)
			hipLaunchKernelGGL(k_reset_list, dim3(1), dim3(1), 0, stream, w->ctl, lsel ^ 1);
#define MKV_COMMIT(FL, MN)                                                                          \
	hipLaunchKernelGGL((k_commit<OP, FL, MN>), grid, block, 0, stream, w->ctl, src, g, table,        \
			w->owner[set], w->owner[set ^ 1], w->cnt, w->state, w->list[lsel], w->list[lsel ^ 1],   \
			lsel, r, stamp_min, next_min, next_inc)
		if (first && !last) MKV_COMMIT(false, true);
		else if (first && last) MKV_COMMIT(false, false);
		else if (!first && !last) MKV_COMMIT(true, true);
		else MKV_COMMIT(true, false);
#undef MKV_COMMIT
		stamp_inc = next_inc;
		stamp_min = next_min;
	}
	w->round_counter = stamp_inc + 1;
	if (w->round_counter > 0xfff00000u) {
		// stamps about to wrap: start a fresh epoch
		(void)hipMemsetAsync(w->owner[0], 0xff, w->nbuckets * 8, stream);
		(void)hipMemsetAsync(w->owner[1], 0xff, w->nbuckets * 8, stream);
		(void)hipMemsetAsync(w->cnt, 0, w->nbuckets * 8, stream);
		w->round_counter = 1;
	}
	const int tail_sel = rounds & 1;    // list the last commit appended to
	hipLaunchKernelGGL((k_tail<OP>), dim3(1), dim3(1024), 0, stream, w->ctl, src, g, table, w->state,
			w->list[tail_sel], tail_sel);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

}  // namespace mkv

extern "C" int mkv_insert(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, void *stream)
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
	src.nblk = (uint32_t)num_blks;
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	if (d.scheme == MKV_SCHEME_2CHOICE)
		return mkv::run_update<mkv::OP_2CHOICE>(d, t, src, blk_elem_num, 0, (hipStream_t)stream);
	return mkv::run_update<mkv::OP_CUCKOO>(d, t, src, blk_elem_num, 0, (hipStream_t)stream);
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
	if (num_elem > 0xfffffff0ll)
		return MKV_ETOOBIG;
	mkv::ReqSrc src;
	src.blk_input = nullptr;
	src.single = reinterpret_cast<const uint32_t *>(in);
	src.nblk = 1;
	return mkv::run_update<mkv::OP_DELETE>(d, reinterpret_cast<uint32_t *>(table), src, nullptr,
			num_elem, (hipStream_t)stream);
}

extern "C" int mkv_reserve(const mkv_geom_t *geom, int64_t max_requests, void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || max_requests < 0)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	int64_t need = mkv::default_req_cap();
	if (max_requests > need)
		need = max_requests;
	return mkv::ensure_ws(&w, (uint64_t)d.hash_mask + 1, need, (hipStream_t)stream);
}

extern "C" void mkv_release_all(void)
{
	std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
	for (auto &kv : mkv::g_ws) {
		mkv::free_ws(kv.second);
		delete kv.second;
	}
	mkv::g_ws.clear();
}

extern "C" int mkv_last_update_info(void *stream, mkv_update_info_t *info)
{
	if (!info)
		return MKV_EINVAL;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		return MKV_EHIP;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 56) ^ (uint64_t)(uintptr_t)stream);
		if (it == mkv::g_ws.end() || !it->second->ctl)
			return MKV_EINVAL;
		w = it->second;
	}
	mkv::Ctl h;
	if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
		return MKV_EHIP;
	if (hipMemcpy(&h, w->ctl, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
	info->total = h.overflow ? 0xffffffffu : h.total;
	for (int i = 0; i < 6; i++)
		info->committed[i] = i < mkv::kMaxRounds ? h.committed[i] : 0;
	info->drained = h.drained;
	info->first_hard = h.first_hard;
	info->rounds = (uint32_t)w->last_rounds;
	return MKV_OK;
}
