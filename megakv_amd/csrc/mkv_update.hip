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

#include "mkv_upd_common.h"
#include "mkv_upd_mark.h"
#include "mkv_upd_round.h"
#include "mkv_upd_tail.h"
#include "mkv_upd_apply.h"

// ---- host side: chunking, workspaces, ordering across streams, launches, the C entry points ----
namespace mkv {

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
	uint32_t *lost = nullptr;              // the mark pass's flags: not the lowest registrant of a bucket (kPlanBit)
	uint32_t *glist = nullptr;
	ChainLog *chain_logs = nullptr;        // kChainLogCap logs of the grid-wide rounds' chains, then one per lane
	                                       // of the tail launch (kTailLanes)
	Sorted sorted = {};                    // big passes (mkv_upd_apply.h): the bins, their hash tables and logs, the list for round 2
	bool cycle_sorted = false;             // the first half of an mkv_cycle insert enqueued the partition, not the mark pass
	uint32_t sorted_acted = 0;             // the verdict of the big-pass path on itself (host flag word 5) last acted on
	int sorted_cooldown = 0;               // big passes still to be run the old way because of it
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
	if (w->lost) (void)hipFree(w->lost);
	if (w->glist) (void)hipFree(w->glist);
	if (w->chain_logs) (void)hipFree(w->chain_logs);
	if (w->sorted.rec) (void)hipFree(w->sorted.rec);
	if (w->sorted.cursor) (void)hipFree(w->sorted.cursor);
	if (w->sorted.heads) (void)hipFree(w->sorted.heads);
	if (w->sorted.next) (void)hipFree(w->sorted.next);
	if (w->sorted.undo) (void)hipFree(w->sorted.undo);
	if (w->sorted.losers) (void)hipFree(w->sorted.losers);
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
	ok = ok && hipMalloc(&w->lost, (size_t)rc * 4) == hipSuccess;
	ok = ok && hipMalloc(&w->chain_logs, ((size_t)kChainLogCap + kTailLanes) * sizeof(ChainLog)) == hipSuccess;
	ok = ok && hipMalloc(&w->ctl, sizeof(Ctl)) == hipSuccess;
	// big passes: at most 2^kSortBinsLog2 bins of kSortBinCap records whatever the table (30 MiB), a list entry per request
	constexpr size_t kBinsMax = (size_t)1 << kSortBinsLog2;
	ok = ok && hipMalloc(&w->sorted.rec, kBinsMax * kSortBinCap * sizeof(uint4)) == hipSuccess;
	ok = ok && hipMalloc(&w->sorted.cursor, (kSortCursorStride + 1) * kBinsMax * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMalloc(&w->sorted.heads, kBinsMax * kSortHeads * sizeof(uint16_t)) == hipSuccess;
	ok = ok && hipMalloc(&w->sorted.next, kBinsMax * kSortBinCap * sizeof(uint16_t)) == hipSuccess;
	ok = ok && hipMalloc(&w->sorted.undo, kBinsMax * kSortBinCap * sizeof(uint2)) == hipSuccess;
	ok = ok && hipMalloc(&w->sorted.losers, (size_t)rc * 4) == hipSuccess;
	if (ok) {
		w->sorted.count = w->sorted.cursor + kSortCursorStride * kBinsMax;
		ok = hipMemsetAsync(w->sorted.cursor, 0, (kSortCursorStride + 1) * kBinsMax * sizeof(uint32_t), stream) == hipSuccess;
	}
	if (ok) {
		void *hf = nullptr;
		ok = hipHostMalloc(&hf, 64, hipHostMallocMapped) == hipSuccess;
		if (ok) {
			w->host_flag = reinterpret_cast<volatile uint32_t *>(hf);
			w->host_flag[0] = w->host_flag[1] = w->host_flag[3] = w->host_flag[4] = w->host_flag[5] = 0;
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
		ok = ok && hipMemsetAsync(w->lost, 0, (size_t)rc * 4, stream) == hipSuccess;
		if (ok)
			hipLaunchKernelGGL(k_init_ctl, dim3(1), dim3(64), 0, stream, w->ctl,
					(unsigned long long)(MKV_DIAG(debug_first_stamp) > 0 ? MKV_DIAG(debug_first_stamp) : 1), w->host_flag_dev);
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
	// A stream that is being captured into a hipGraph takes neither a device synchronisation nor the
	// library's event: the capture holds the calls of ONE stream (what a replay does is what was captured,
	// examples/graph_cycle.c), and an update that would have to be ordered behind another stream's cannot
	// be part of it -- refused, nothing enqueued.
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
		if (o.any && o.last != stream) {
			fprintf(stderr, "megakv_hip: update on a stream under capture while the device's last update ran on another stream -- "
					"not enqueued (synchronise, or run one update on this stream, before capturing)\n");
			return MKV_EINVAL;
		}
		return MKV_OK;
	}
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
	hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
	if (stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
		return MKV_OK;                  // (the library's event does not go into a captured graph)
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
	hipLaunchKernelGGL((k_mark<OP>), dim3((unsigned)MKV_DIAG(mark_alone_blocks)), dim3(256), 0, stream, w->ctl, src, g,
			w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, lo, hi, table,
			tuning().lazy_alt | (w->host_flag && w->host_flag[4] ? 0x100 : 0),
			tuning().plans ? w->undo : nullptr, w->lost);
}

// ---- big passes: partition, then apply per bucket (mkv_upd_apply.h) ----------------------------------------------
static Sorted sorted_of(Workspace *w, const DevGeom &g)
{
	Sorted S = w->sorted;
	uint32_t bucket_bits = 0;
	while (((uint64_t)1 << bucket_bits) <= g.hash_mask)
		bucket_bits++;
	S.bin_shift = bucket_bits > kSortBinsLog2 ? bucket_bits - kSortBinsLog2 : 0u;
	S.nbins = (uint32_t)(((uint64_t)g.hash_mask + 1u) >> S.bin_shift);
	S.host_flag = w->host_flag_dev;
	return S;
}

// A pass goes that way when it is announced big enough to pay for the extra launches (Tuning::sorted_min), is the
// call's only pass, and an even spread over the bins would fill them to two thirds at most (a skewed one overflows
// a bin, which the device finds out: Ctl::sorted_fail -- k_apply then leaves the whole pass to round 2 and the tail).
static bool sorted_wanted(Workspace *w, const DevGeom &g, long long hint, bool is_delete, int nchunks)
{
	const long long smin = tuning().sorted_min;
	if (is_delete || smin <= 0 || hint < smin || nchunks != 1 || !w->sorted.rec)
		return false;
	const Sorted S = sorted_of(w, g);
	if (hint > (long long)S.nbins * (long long)(kSortBinCap * 2u / 3u))
		return false;
	// The path's verdict on itself (word 5 of the host-mapped flags, written by round 2 of such a pass and read here
	// without synchronising -- a pass or two late, which only costs time): a pass that fell back, or left over more
	// than one request in 256 (a well-filled table: the left-overs are what such a pass costs), sends the next
	// Tuning::sorted_cooldown big passes the old way; then one is tried again.
	if (w->host_flag) {
		const uint32_t word = w->host_flag[5];
		if ((word & 1u) && word != w->sorted_acted) {
			w->sorted_acted = word;
			w->sorted_cooldown = tuning().sorted_cooldown;
		}
	}
	if (w->sorted_cooldown > 0) {
		w->sorted_cooldown--;
		return false;
	}
	return true;
}

static uint32_t part_blocks_for(long long hint)
{
	const long long b = (hint + kPartTile - 1) / kPartTile;
	return (uint32_t)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

template <Op OP>
static int launch_rounds(Workspace *w, const DevGeom &g, uint32_t *table, hipStream_t stream,
		uint32_t lo = 0, uint32_t hi = kNone, int first_chunk = 1, int last_chunk = 1, hipEvent_t done_ev = nullptr,
		long long hint = 0, bool plans_ok = true, bool sorted = false);

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
	if (!lately && MKV_DIAG(calm_factor) > 1)
		c *= MKV_DIAG(calm_factor);
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
	const bool sorted = sorted_wanted(w, g, hint, OP == OP_DELETE, nchunks);
	for (int c = 0; c < nchunks; c++) {
		uint32_t lo, hi;
		chunk_range(c, ck, &lo, &hi);
		if constexpr (OP != OP_DELETE) {
			if (sorted)
				hipLaunchKernelGGL((k_part<OP>), dim3(part_blocks_for(hint)), dim3(kPartThreads), 0, stream, w->ctl, src, g,
						sorted_of(w, g), w->reqc, w->state, (uint32_t)w->req_cap, hi);
		}
		if (!sorted)
			launch_mark<OP>(w, g, src, stream, table, lo, hi);
		rc = launch_rounds<OP>(w, g, table, stream, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : hint,
				true, sorted);
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
		uint32_t lo, uint32_t hi, int first_chunk, int last_chunk, hipEvent_t done_ev, long long hint, bool plans_ok, bool sorted)
{
	// (sorted: the pass's requests went through the partition, k_part / k_search_part -- no mark pass, no round 1:
	// k_apply, then round 2 over what it left over, bit 9 of the hint, then the tail)
	Sorted S = {};
	if constexpr (OP != OP_DELETE) {
		if (sorted) {
			S = sorted_of(w, g);
			hipLaunchKernelGGL((k_apply<OP>), dim3(S.nbins), dim3(kApplyThreads), 0, stream, w->ctl, g, table, S, w->state, w->owner[1],
					w->reqc);
		}
	}
	// (hint: the caller's upper bound on the pass's requests, 0 = unknown)
	// (plans_ok: nothing has written the table since this pass's mark pass read it -- false for the insert of a
	// cycle whose mark pass rode in the search launch ahead of the cycle's deletes)
	int tail_rounds = tuning().tail_rounds;
	if (tail_rounds > kMaxTailRounds)
		tail_rounds = kMaxTailRounds;
	// what the rounds are told about the tuning: bit 2 -- every eviction chain left to the tail (tests);
	// bit 3 -- no groups of duplicates; bit 5 -- no self-help of a bucket's second registrant
	// bit 6 -- round 1 carries out the mark pass's plans (kPlanBit)
	const int rhint = (MKV_DIAG(debug_skip) & 4) | (tuning().dup_groups ? 0 : 8) | (tuning().self_help ? 0 : 32) |
			(plans_ok && tuning().plans ? 64 : 0) | (sorted ? 512 : 0);
	// (a pass announced with fewer requests than round 1's grid has lanes: as many work-groups as it takes --
	// 208 instead of 512 are 0.5 us of a 52 K-request pass)
	// (not while round 1 has been leaving round 2 a lot, or chains have been running late: what a work-group
	// holds back for its second phase is then the round's critical path, and more work-groups share it)
	long long r1_blocks = tuning().round_blocks;
	const bool busy = (w->host_flag && w->host_flag[3]) || w->barrier_score > 0;
	if (hint > 0 && !busy && (hint + 255) / 256 < r1_blocks)
		r1_blocks = (hint + 255) / 256;
	// (... and a BIG pass a lane per request, within 2048 work-groups -- the round itself keeps to kTlGroups of them
	// unless the passes are mostly plan-carrying, bit 8 of its hint: round_body.  Only the default of
	// Tuning::round_blocks is scaled: a host that set it gets what it set.)
	bool big_grid = false;
	if (hint > r1_blocks * 256 && r1_blocks == 512 && OP != OP_DELETE && plans_ok && tuning().plans) {
		r1_blocks = (hint + 255) / 256 > 2048 ? 2048 : (hint + 255) / 256;
		big_grid = true;
	}
	if (!sorted)
		hipLaunchKernelGGL((k_round<OP, 1>), dim3((unsigned)r1_blocks), dim3(256), 0, stream, w->ctl, g, table,
				w->owner[0], w->owner[1], w->reqc, w->state, w->undo, w->chain_logs, lo, hi, (uint32_t)w->req_cap,
				rhint | (big_grid ? 256 : 0), w->cnt + ((size_t)g.hash_mask + 1u), w->owner[4], w->lost, NoSorted());
	// round 2: inside the tail launch (k_tail's first phase), unless the pass is announced with more
	// requests than that launch has lanes for in a few turns, or is a delete
	// (word 3 of the tail kernel's flags, read without synchronising -- a pass late, which only costs time:
	// round 1 has been leaving round 2 a lot: duplicates of hot keys, say)
	// (a big pass's round 2 runs over a LIST -- what k_apply left over -- in a launch of its own: inside the tail
	// launch, 32 work-groups, it took 40 us where the two launches take 15 + 23, and 71 against 15 + 37 once a
	// thousand requests are left over: profiles/r06_sorted_path.txt)
	const bool merged = OP != OP_DELETE && tuning().merge_round2 != 0 && hint <= (long long)tuning().merge_max && !sorted &&
			!(w->host_flag && w->host_flag[3] && tuning().merge_round2 == 1);
	// (round 2 looks at every request's tag to find its few: a lane per announced request, within 2048
	// work-groups -- 524 K tags on the 512 work-groups of round 1 took 15 us, on 2048 they take 6.5)
	// (... and a pass announced small as many work-groups as it has requests for: a delete pass of a scheduler cycle,
	// 16 K requests, leaves round 2 nothing as a rule -- the launch is its cost)
	long long r2_blocks = (hint + 255) / 256;
	if (hint <= 0 || r2_blocks > tuning().round_blocks)
		r2_blocks = r2_blocks < tuning().round_blocks ? tuning().round_blocks : (r2_blocks > 2048 ? 2048 : r2_blocks);
	if (!merged && !sorted)
		hipLaunchKernelGGL((k_round<OP, 2>), dim3((unsigned)r2_blocks), dim3(256), 0, stream, w->ctl, g, table,
				w->owner[1], w->owner[0], w->reqc, w->state, w->undo, w->chain_logs, lo, hi, (uint32_t)w->req_cap,
				rhint, w->cnt + ((size_t)g.hash_mask + 1u), w->owner[4], (const uint32_t *)nullptr, NoSorted());
	if constexpr (OP != OP_DELETE) {
		if (sorted)     // (the instantiations that know of the bins: k_round<.., 2, true>, k_tail<.., true>)
			hipLaunchKernelGGL((k_round<OP, 2, true>), dim3((unsigned)r2_blocks), dim3(256), 0, stream, w->ctl, g, table,
					w->owner[1], w->owner[0], w->reqc, w->state, w->undo, w->chain_logs, lo, hi, (uint32_t)w->req_cap,
					rhint, w->cnt + ((size_t)g.hash_mask + 1u), w->owner[4], (const uint32_t *)nullptr, S);
	}
	bool tail_done = false;
	if constexpr (OP != OP_DELETE) {
		if (sorted) {
			MKV_LAUNCH_TIMED((k_tail<OP, true>), dim3(tail_groups_for<OP>()), dim3(kTailThreads), stream, (hipEvent_t) nullptr, done_ev, w->ctl, g, table, w->owner[2],
					w->owner[3], w->cnt, w->owner[0], w->owner[1], w->owner[4], w->reqc, w->state, w->undo, w->chain_logs, w->glist,
					tail_rounds, lo, hi, first_chunk,
					last_chunk, w->host_flag_dev, (uint32_t)MKV_DIAG(grid_min), (uint32_t)MKV_DIAG(grid_progress), MKV_DIAG(debug_skip) & 3,
					-1, (uint32_t)w->req_cap, S, 1);
			tail_done = true;
		}
	}
	if (!tail_done)
		MKV_LAUNCH_TIMED((k_tail<OP>), dim3(tail_groups_for<OP>()), dim3(kTailThreads), stream, (hipEvent_t) nullptr, done_ev, w->ctl, g, table, w->owner[2],
				w->owner[3], w->cnt, w->owner[0], w->owner[1], w->owner[4], w->reqc, w->state, w->undo, w->chain_logs, w->glist,
				tail_rounds, lo, hi, first_chunk,
				last_chunk, w->host_flag_dev, (uint32_t)MKV_DIAG(grid_min), (uint32_t)MKV_DIAG(grid_progress), MKV_DIAG(debug_skip) & 3,
				merged ? rhint : -1, (uint32_t)w->req_cap, NoSorted(), 0);
	return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
}

}  // namespace mkv

extern "C" int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);
extern "C" int mkv_defer_barrier(void);

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
	const int brc = mkv_defer_barrier();
	if (brc != MKV_OK)
		return brc;
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
	w->cycle_sorted = false;
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
	// a big pass: the partition rides in the search launch in the mark pass's place (mkv_upd_apply.h)
	w->cycle_sorted = mkv::sorted_wanted(w, d, size_hint, false, w->cycle_ck.passes);
	const uint32_t mark_blocks = w->cycle_sorted ? mkv::part_blocks_for(size_hint) : mkv::mark_blocks_for(size_hint, w->barrier_score > 0);
	if (search_blocks + mark_blocks > 0x7fffffffLL)
		return MKV_ETOOBIG;
	const dim3 grid((unsigned)(search_blocks + mark_blocks)), block(256);
	// (ev_start / ev_stop: HIP events that receive the START and END time of this very dispatch --
	// hipExtLaunchKernelGGL -- i.e. the kernel's duration as a profiler sees it, without the
	// launch latency a pair of hipEventRecord calls around the launch includes)
	hipEvent_t e0 = (hipEvent_t)ev_start, e1 = (hipEvent_t)ev_stop;
	if (w->cycle_sorted) {
		const mkv::Sorted S = mkv::sorted_of(w, d);
		if (d.scheme == MKV_SCHEME_2CHOICE)
			MKV_LAUNCH_TIMED((mkv::k_search_part<mkv::OP_2CHOICE>), grid, block, s, e0, e1, in2, out, t4, num_search,
					(uint32_t)search_blocks, w->ctl, src, d, S, w->reqc, w->state, (uint32_t)w->req_cap, hi);
		else
			MKV_LAUNCH_TIMED((mkv::k_search_part<mkv::OP_CUCKOO>), grid, block, s, e0, e1, in2, out, t4, num_search,
					(uint32_t)search_blocks, w->ctl, src, d, S, w->reqc, w->state, (uint32_t)w->req_cap, hi);
		return hipGetLastError() == hipSuccess ? MKV_OK : MKV_EHIP;
	}
	if (d.scheme == MKV_SCHEME_2CHOICE)
		MKV_LAUNCH_TIMED((mkv::k_search_mark<mkv::OP_2CHOICE>), grid, block, s, e0, e1, in2, out, t4, num_search,
				(uint32_t)search_blocks, w->ctl, src, d, w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi,
				mkv::tuning().lazy_alt | (w->host_flag && w->host_flag[4] ? 0x100 : 0), mkv::tuning().plans ? w->undo : nullptr, w->lost);
	else
		MKV_LAUNCH_TIMED((mkv::k_search_mark<mkv::OP_CUCKOO>), grid, block, s, e0, e1, in2, out, t4, num_search,
				(uint32_t)search_blocks, w->ctl, src, d, w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi,
				mkv::tuning().lazy_alt | (w->host_flag && w->host_flag[4] ? 0x100 : 0), mkv::tuning().plans ? w->undo : nullptr, w->lost);
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
	// (joined bit 0: the mark pass ran on the side stream; bit 1: a delete of the cycle ran since the mark pass
	// read the table -- the plans it made for chunk 0 do not hold)
	const bool stale = (joined & 2) != 0;
	if ((joined & 1) && hipStreamWaitEvent(s, w->ev_join, 0) != hipSuccess)
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
		// (a big pass, partitioned in the search launch: the partition never looked at the table, and k_apply reads it
		// behind the cycle's deletes)
		const bool sorted = c == 0 && w->cycle_sorted;
		if (d.scheme == MKV_SCHEME_2CHOICE) {
			if (c > 0)
				mkv::launch_mark<mkv::OP_2CHOICE>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_2CHOICE>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : size_hint,
					c > 0 || !stale || sorted, sorted);
		} else {
			if (c > 0)
				mkv::launch_mark<mkv::OP_CUCKOO>(w, d, src, s, t, lo, hi);
			rc = mkv::launch_rounds<mkv::OP_CUCKOO>(w, d, t, s, lo, hi, c == 0, c == nchunks - 1, nullptr, nchunks > 1 ? (long long)ck.step : size_hint,
					c > 0 || !stale || sorted, sorted);
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
				(uint8_t *)nullptr, 0u, 0u, self_u, self_out, 0, (mkv::Undo *)nullptr, (uint32_t *)nullptr);
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
				w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi, self_u, self_out, mkv::tuning().lazy_alt | (w->host_flag && w->host_flag[4] ? 0x100 : 0),
				mkv::tuning().plans ? w->undo : nullptr, w->lost);
	else
		MKV_LAUNCH_TIMED((mkv::k_search_slots_mark<mkv::OP_2CHOICE>), grid, block, s, e0, e1, recvbuf, slot_words,
				(uint32_t)cap_get, out_slots, t4, chunks ? chunks : 1u, search_blocks, w->ctl, src, d,
				w->owner[0], w->owner[4], w->reqc, w->state, (uint32_t)w->req_cap, hi, self_u, self_out, mkv::tuning().lazy_alt | (w->host_flag && w->host_flag[4] ? 0x100 : 0),
				mkv::tuning().plans ? w->undo : nullptr, w->lost);
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
// microseconds of work each, the updates serialised.  Here the whole cycle is three pieces of work:
// ONE launch that searches every worker's requests (a miss writes 0: no memset) and marks the cycle's
// deletes and inserts beside the searches (k_search_multi_mark: the mark work-groups read the workers'
// block pointers and counts -- device arrays the host rewrites per cycle, as the ABI has them --
// themselves), then the rounds of ONE delete pass over the workers' delete arrays as its blocks and of
// ONE insert pass over all the workers' blocks, worker-major.  Order: every search sees the table as the
// cycle found it; then the deletes in (worker, index) order; then the inserts in (worker, block, index)
// order -- what the per-worker calls give when the workers' keys do not meet within a cycle (per worker
// they run delete then insert: a worker deleting what an EARLIER worker inserts in the same cycle is
// the one case that differs).
namespace mkv {

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

// the searches of all workers, the mark pass of the cycle's deletes (on the device's first workspace) and the mark
// pass of its inserts (on the second) in ONE launch: the mark work-groups come first, they are few and latency-bound.
// (Before: search, a one-work-group launch gathering the workers' block lists, a mark launch for the deletes and one
// for the inserts -- 8 + 5 + 11 + 20 us of a 96-us cycle: profiles/r06_scheduler_regime.txt.)
struct MarkArgs {
	Ctl *ctl;
	unsigned long long *owner, *second;
	uint4 *reqc;
	uint8_t *state;
	Undo *undo;
	uint32_t *lost;
	uint32_t req_cap;
	uint32_t nblk;
	uint32_t blocks;        // work-groups of this mark pass in the launch
	int lazy_on;
};

template <Op OP>
__global__ __launch_bounds__(256) void k_search_multi_mark(MultiSearch a, const uint4 *__restrict__ table, DevGeom g, GatherBlocks gb,
		MarkArgs del, MarkArgs ins)
{
	const uint32_t b = blockIdx.x;
	ReqSrc src = {};
	src.cnt_stride = 1;
	if (b < del.blocks) {
		src.nblk = del.nblk;
		mark_body<OP_DELETE, true>(b, del.blocks, del.ctl, src, g, del.owner, del.second, del.reqc, del.state, del.req_cap, 0u, kNone,
				nullptr, 0u, nullptr, nullptr, gb);
		return;
	}
	if (b < del.blocks + ins.blocks) {
		src.nblk = ins.nblk;
		mark_body<OP, true>(b - del.blocks, ins.blocks, ins.ctl, src, g, ins.owner, ins.second, ins.reqc, ins.state, ins.req_cap, 0u,
				kNone, reinterpret_cast<const uint32_t *>(table), (uint32_t)ins.lazy_on, ins.undo, ins.lost, gb);
		return;
	}
	const uint32_t sb = b - del.blocks - ins.blocks;
	int w = 0;
	while (w + 1 < a.nw && sb >= a.first[w + 1])
		w++;
	search_body<4, true, false>(sb - a.first[w], a.in[w], a.out[w], table, (int64_t)a.n[w], g.hash_mask, g.blk_mask);
}

}  // namespace mkv

extern "C" void mkv_after_search(void *stream);
extern "C" int mkv_defer_barrier(void);

extern "C" int mkv_cycle_workers(const mkv_geom_t *geom, bucket_t *table, const mkv_worker_t *workers, int nworkers,
		void *stream)
{
	mkv::DevGeom d;
	if (!mkv::make_dev_geom(geom, &d) || !table || !workers || nworkers < 1 || nworkers > mkv::kMaxWorkers)
		return MKV_EINVAL;
	const int brc = mkv_defer_barrier();        // (the drop-in side's queued request copies, deferred mode: mkv_abi.hip)
	if (brc != MKV_OK)
		return brc;
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
		if (k.num_blks > mkv::kMaxBlocks)
			return MKV_EINVAL;                 // (before it is summed: the sum must not wrap)
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
	mkv::Workspace *w0 = nullptr, *w1 = nullptr;
	if (nblk || ndel) {
		// A call that is refused enqueues nothing: whatever can refuse it comes before the launch.  (The updates take
		// their place in the device's issue order now; the deletes use the device's first workspace, the inserts its
		// second one -- the two mark passes run side by side.)
		const int orc = mkv::order_updates(s);
		if (orc != MKV_OK)
			return orc;
		mkv::ReqSrc probe = {};
		probe.single_n = ndel;
		int wrc = ndel ? mkv::ensure_ws(&w0, (uint64_t)d.hash_mask + 1, mkv::ws_need(probe, 0), s) : MKV_OK;
		if (wrc == MKV_OK && nblk) {
			probe.single_n = 0;
			wrc = mkv::ensure_ws(&w1, (uint64_t)d.hash_mask + 1, mkv::ws_need(probe, 0), s, mkv::kCycleSalt);
		}
		if (wrc != MKV_OK)
			return wrc;
	}
	const bool cuckoo = d.scheme != MKV_SCHEME_2CHOICE;
	if (w0 || w1) {
		mkv::MarkArgs del = {}, ins = {};
		if (w0) {
			del.ctl = w0->ctl;
			del.owner = w0->owner[0];
			del.second = w0->owner[4];
			del.reqc = w0->reqc;
			del.state = w0->state;
			del.req_cap = (uint32_t)w0->req_cap;
			del.nblk = (uint32_t)nworkers;
			del.blocks = (uint32_t)((ndel + 255) / 256 > 256 ? 256 : (ndel + 255) / 256);
		}
		if (w1) {
			ins.ctl = w1->ctl;
			ins.owner = w1->owner[0];
			ins.second = w1->owner[4];
			ins.reqc = w1->reqc;
			ins.state = w1->state;
			ins.undo = mkv::tuning().plans ? w1->undo : nullptr;
			ins.lost = w1->lost;
			ins.req_cap = (uint32_t)w1->req_cap;
			ins.nblk = nblk;
			// (the size of the pass is the device's to know: a lane per request of what the reference's scheduler hands
			// a worker at most -- 8 x 4 096 -- would be 2 048 work-groups with 16 workers; four per block is one turn
			// for the usual 6 000 inserts per worker and costs the launch nothing)
			ins.blocks = mkv::mark_blocks_for(0, w1->barrier_score > 0);
			if (ins.blocks < 4u * nblk)
				ins.blocks = 4u * nblk > 1024u ? 1024u : 4u * nblk;
			ins.lazy_on = mkv::tuning().lazy_alt | (w1->host_flag && w1->host_flag[4] ? 0x100 : 0);
			w1->cycle_sorted = false;
		}
		const dim3 grid(del.blocks + ins.blocks + blocks), blk(256);
		if (cuckoo)
			hipLaunchKernelGGL((mkv::k_search_multi_mark<mkv::OP_CUCKOO>), grid, blk, 0, s, ms, reinterpret_cast<const uint4 *>(table), d, gb,
					del, ins);
		else
			hipLaunchKernelGGL((mkv::k_search_multi_mark<mkv::OP_2CHOICE>), grid, blk, 0, s, ms, reinterpret_cast<const uint4 *>(table), d, gb,
					del, ins);
	} else if (blocks) {
		hipLaunchKernelGGL(mkv::k_search_multi, dim3(blocks), dim3(256), 0, s, ms, reinterpret_cast<const uint4 *>(table),
				d.hash_mask, d.blk_mask);
	}
	mkv_after_search(s);        // (the deferred mode's queued result copies: mkv_abi.hip)
	if (hipGetLastError() != hipSuccess)
		return MKV_EHIP;
	uint32_t *t = reinterpret_cast<uint32_t *>(table);
	int rc = MKV_OK;
	if (w0)
		rc = mkv::launch_rounds<mkv::OP_DELETE>(w0, d, t, s, 0, mkv::kNone, 1, 1, nullptr, ndel);
	// (the inserts' mark pass read the table ahead of the cycle's deletes: its plans do not stand if there were any)
	if (rc == MKV_OK && w1)
		rc = cuckoo ? mkv::launch_rounds<mkv::OP_CUCKOO>(w1, d, t, s, 0, mkv::kNone, 1, 1, nullptr, 0, ndel == 0)
		            : mkv::launch_rounds<mkv::OP_2CHOICE>(w1, d, t, s, 0, mkv::kNone, 1, 1, nullptr, 0, ndel == 0);
	if (rc == MKV_OK && (w0 || w1))
		rc = mkv::updates_issued(s);
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
	const int brc = mkv_defer_barrier();
	if (brc != MKV_OK)
		return brc;
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
				(long long)w->req_cap * (1 + 16 + 16 + 4 + 4) +
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
	if (h.ep_pending) {
		// the pass's tail launch found nothing to do and left without its epilogue (k_tail): the statistics
		// are what the rounds counted -- nothing reached round 2 or the tail
		memset(info, 0, sizeof(*info));
		info->total = h.total;
		for (int k = 0; k < mkv::kShards; k++) {
			info->round1 += h.committed[0][k].n;
			info->round2 += h.committed[1][k].n;
			info->helped += h.committed[2][k].n;
			info->losers += h.committed[3][k].n;
		}
		for (uint32_t k = 0; k < mkv::kTlGroups; k++) {
			info->round1 += h.wg_rec[k].stat.y;
			info->helped += h.wg_rec[k].stat.z;
			info->losers += h.wg_rec[k].stat.w;
		}
		info->first_hard = h.first_hard;
		info->chains = h.chain_ok;
		return MKV_OK;
	}
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

// Big insert passes (partition, then apply per bucket: mkv_upd_apply.h) on the device's workspace so far:
// out[0] = passes that ran that way, out[1] = passes whose partition gave up (a bin overflowed: they ran the
// reservation rounds), out[2] = requests the last such pass left over for round 2 and the tail, out[3] = 0.
// Waits for `stream`.
extern "C" int mkv_sorted_stats(void *stream, int cycle_insert, uint32_t out[4])
{
	if (!out)
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
	if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
		return MKV_EHIP;
	uint32_t h[3];
	static_assert(offsetof(mkv::Ctl, sorted_fallbacks) == offsetof(mkv::Ctl, sorted_nlosers) + 8, "three words in a row");
	if (hipMemcpy(h, &w->ctl->sorted_nlosers, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
	out[0] = h[1];
	out[1] = h[2];
	out[2] = h[0];
	out[3] = 0u;
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

// diagnostics (tests/test_gpu_parity.py): what the last BIG pass's round 1 left for the next one -- 1: mostly plan-carrying, a
// lane per request pays (Ctl::r1_wide_next); 0: not; < 0: no such workspace
extern "C" int mkv_debug_wide_round(int cycle_insert)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
		return -2;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
		if (it == mkv::g_ws.end() || !it->second->ctl)
			return -1;
		w = it->second;
	}
	uint32_t v = 0;
	if (hipMemcpy(&v, &w->ctl->r1_wide_next, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
		return -2;
	return (int)v;
}

// diagnostics (tools/loser_probe.py): the tags and cache entries of requests 0 .. n-1 after the last pass
extern "C" int mkv_debug_dump(int cycle_insert, uint8_t *state_out, uint32_t *reqc_out, uint32_t n)
{
	int dev = 0;
	if (!state_out || !reqc_out || hipGetDevice(&dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
		return MKV_EINVAL;
	mkv::Workspace *w = nullptr;
	{
		std::lock_guard<std::mutex> lock(mkv::g_ws_mutex);
		auto it = mkv::g_ws.find(((uint64_t)dev << 1) | (cycle_insert ? 1ull : 0ull));
		if (it == mkv::g_ws.end() || !it->second->ctl || (int64_t)n > it->second->req_cap)
			return MKV_EINVAL;
		w = it->second;
	}
	if (hipMemcpy(state_out, w->state, n, hipMemcpyDeviceToHost) != hipSuccess ||
			hipMemcpy(reqc_out, w->reqc, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess)
		return MKV_EHIP;
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
	if (h.ep_pending)
		// (the tail launch left at once, without its epilogue: no phase took any time; the rounds'
		// diagnostic counts are still where they put them)
		for (int k = 0; k < 16; k++)
			out_us[k] = k < 8 ? 0.0f : (float)h.dbg[k - 8];
	return MKV_OK;
}
#endif  // MKV_DIAGNOSTICS
