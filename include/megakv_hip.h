/*
 * megakv_hip.h -- run-time-geometry entry points of the MI355X hash index.
 *
 * libgpuhash.h is the drop-in surface (MEM_P and scheme baked in, exactly as
 * the reference bakes them into libgpuhash.a).  The functions below are the
 * same operations with the geometry passed at run time; gpu_hash_search /
 * gpu_hash_insert / gpu_hash_delete are thin calls of them.  They exist so
 * that one library build can be parity-tested on small tables and both insert
 * schemes, and so the multi-GPU sharding (local tables with a different
 * region split) needs no recompilation.
 *
 * Plain C: device pointers, sizes, and the HIP stream as an opaque pointer
 * (a hipStream_t; NULL = default stream).  Every call is asynchronous on
 * `stream` and returns 0, or a negative MKV_E* code when nothing was enqueued.
 */
#ifndef MEGAKV_HIP_H
#define MEGAKV_HIP_H

#include <stdint.h>
#include "gpu_hash.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MKV_SCHEME_CUCKOO  0	/* reference HASH_CUCKOO  (gpu_hash.cu:231-433) */
#define MKV_SCHEME_2CHOICE 1	/* reference HASH_2CHOICE (gpu_hash.cu:77-229)  */

#define MKV_OK          0
#define MKV_EINVAL     -1	/* bad geometry / arguments */
#define MKV_ENOMEM     -2	/* workspace allocation failed */
#define MKV_ETOOBIG    -3	/* more requests than the reserved workspace holds */
#define MKV_EHIP       -4	/* a HIP runtime call failed */

typedef struct mkv_geom_s {
	uint32_t mem_p;			/* log2(table bytes); reference MEM_P (gpu_hash.h:55) */
	uint32_t iblock_p;		/* log2(closed regions); reference IBLOCK_P (gpu_hash.h:67) */
	uint32_t scheme;		/* MKV_SCHEME_* */
	uint32_t max_cuckoo;	/* reference MAX_CUCKOO_NUM = 5 (gpu_hash.h:75) */
} mkv_geom_t;

/* search flags */
#define MKV_SEARCH_KEEP_MISS 0u	/* reference behaviour: out untouched on a miss */
#define MKV_SEARCH_ZERO_MISS 1u	/* write 0 on a miss: caller needs no memset */

int mkv_search(const mkv_geom_t *geom, const selem_t *in, loc_t *out,
		const bucket_t *table, int64_t num_elem, uint32_t flags, void *stream);

int mkv_insert(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, void *stream);

/* Same, with an upper bound on the total number of requests (the counts are device
 * memory, so the library cannot know it).  With a hint above the "insert_chunk"
 * tuning value (64 K) the batch is processed in canonical-index chunks, which keeps
 * the redo work behind an eviction barrier small; results do not depend on it.
 * size_hint <= 0: unknown, one pass (what mkv_insert / gpu_hash_insert do). */
int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);

int mkv_delete(const mkv_geom_t *geom, delem_t *in, bucket_t *table,
		int64_t num_elem, void *stream);

/* One scheduler cycle of one worker in a single call, in the reference's order
 * (src/mega_scheduler.c:393-502): clear out -> search -> delete -> insert, all
 * asynchronous and ordered on `stream`; a part is skipped when its count is 0.
 * Results are those of the three separate calls.  Two things differ in how:
 * the result buffer is cleared by the search kernel itself (a miss writes 0;
 * MKV_CYCLE_MEMSET asks for the separate memset instead), and the insert's first
 * pass, which reads only the requests and never the table, is launched as extra
 * work-groups of the search kernel.  MKV_CYCLE_SIDE_STREAM runs that pass on an
 * internal side stream instead (it starts after everything enqueued on `stream`
 * before the call); MKV_CYCLE_NO_OVERLAP issues the plain three-call sequence.
 * insert_hint: upper bound on the insert requests (see mkv_insert_hinted), 0 = unknown. */
#define MKV_CYCLE_MEMSET      0x100u
#define MKV_CYCLE_NO_OVERLAP  0x200u
#define MKV_CYCLE_SIDE_STREAM 0x400u
int mkv_cycle(const mkv_geom_t *geom, bucket_t *table, const selem_t *search_in,
		loc_t *search_out, int64_t num_search, delem_t *delete_in, int64_t num_delete,
		ielem_t **blk_input, int *blk_elem_num, int num_blks, int64_t insert_hint, uint32_t search_flags,
		void *stream, void *timer, int slot);

/* One call for ALL workers of a scheduler cycle (src/mega_scheduler.c:393-502 runs memset, search,
 * delete, insert per worker on a stream each -- up to 16 x 9 launches of a few microseconds of work).
 * Per worker exactly what the three entry points take: device arrays of search requests and results,
 * of delete requests, and the device array of block pointers with its device int[] of block counts
 * (rewritten by the host every cycle, as in the ABI).  Work: one search launch over every worker's
 * requests (a miss writes 0: no memset), one delete pass, one insert pass.  Order: every search sees
 * the table as the cycle found it; then the deletes in (worker, index) order; then the inserts in
 * (worker, block, index) order -- the result of the per-worker calls whenever no worker deletes what an
 * earlier worker inserts in the same cycle.  At most 32 workers and 1024 insert blocks in all.
 * Asynchronous on `stream`. */
typedef struct mkv_worker_s {
	const selem_t *search_in;	/* device, num_search requests */
	loc_t *search_out;			/* device, 2 * num_search results */
	int64_t num_search;
	delem_t *delete_in;			/* device, num_delete requests */
	int64_t num_delete;
	ielem_t **blk_input;		/* device array of num_blks device pointers */
	int *blk_elem_num;			/* device int[num_blks] */
	int num_blks;
} mkv_worker_t;
int mkv_cycle_workers(const mkv_geom_t *geom, bucket_t *table, const mkv_worker_t *workers, int nworkers, void *stream);

/* Deferred launches for a host that keeps to the three drop-in entry points (MKV_DEFER=1 in the
 * environment, or mkv_set_defer(1); off by default).  gpu_hash_search / gpu_hash_delete /
 * gpu_hash_insert then only record their arguments, per stream; a device-to-host cudaMemcpyAsync issued
 * through the library's cuda_runtime.h names behind recorded work is queued; cudaDeviceSynchronize,
 * cudaStreamSynchronize, the synchronous copies and frees, and mkv_flush() issue everything recorded as
 * ONE cycle -- mkv_cycle for one worker, mkv_cycle_workers for several (every search sees the table as
 * the cycle found it, then the deletes, then the inserts, worker by worker in the order the workers'
 * streams first appeared: reference src/mega_scheduler.c:393-504 issues exactly that sequence and
 * synchronises once) -- and then the queued copies.  mkv_flush() does not wait; a host that synchronises
 * with the HIP runtime's own calls instead of the cuda* names calls it first. */
void mkv_set_defer(int on);
int mkv_flush(void);

/* Measurement aid: `slots` HIP event pairs; mkv_cycle(timer, slot) records pair
 * `slot` on the caller's stream around its search launch.  timer may be NULL. */
void *mkv_timer_create(int slots);
int mkv_timer_elapsed_ms(void *timer, int slot, float *ms);
void mkv_timer_destroy(void *timer);

/* Insert/delete keep their reservation state (conflict tables, request state) in a device
 * workspace owned by the library: ONE pair per device, shared by every stream and table of
 * the process (one for insert / delete calls, one for the insert half of mkv_cycle) -- about
 * 0.9 GiB each at MEM_P = 30 with the default request capacity, whatever the number of worker
 * streams (the reference's scheduler runs up to 16 over one table, src/mega_scheduler.c:273-280).
 * They are created on the first call that needs them; call this beforehand to size them
 * explicitly (and to keep allocation out of a region being captured into a hipGraph).
 * max_requests bounds the requests of one insert/delete call.  `stream`: where the
 * initialisation is enqueued (it is waited for). */
int mkv_reserve(const mkv_geom_t *geom, int64_t max_requests, void *stream);

/* Frees the workspaces of the calling process (synchronises the device). */
void mkv_release_all(void);

/* Device memory the workspaces of the current device hold, in bytes. */
long long mkv_workspace_bytes(void);

/* An insert/delete call WITHOUT a size hint whose device-side counts exceed the workspace is
 * not applied (the library only learns of it after the fact): reported on stderr by the next
 * insert/delete call, and counted here.  Nothing aborts. */
unsigned long long mkv_dropped_batches(void);

/* Big insert passes -- a pass announced (mkv_insert_hinted, mkv_cycle's insert_hint) with at least
 * MKV_SORTED_MIN requests (tuning "sorted_min"; default 0: never -- the path is exact but measured no
 * faster than the reservation rounds on MI355X) is split into bins by bucket
 * range and applied per bucket with one line read and one whole-line write, no reservation atomics;
 * what is not settled in its primary bucket goes through round 2 and the tail as ever.  Statistics of
 * the current device's workspace (cycle_insert: the one mkv_cycle's inserts use): out[0] = passes that
 * ran that way so far, out[1] = passes that fell back to the reservation rounds because a bin
 * overflowed (many duplicates of one key), out[2] = requests the last such pass left over, out[3] = 0.
 * Waits for `stream`. */
int mkv_sorted_stats(void *stream, int cycle_insert, uint32_t out[4]);

/* Updates issued on different streams are put in issue order (while updates have been changing
 * streams lately every update call records an event of the library's own behind its last kernel
 * and an update on another stream waits for it; a change of stream after a long run on one
 * stream synchronises the device instead).  No stream handle is kept beyond the name of the last
 * one used: a host that destroys a stream may call this first (the stream is synchronised and
 * forgotten); it is not required. */
void mkv_forget_stream(void *stream);

/* Diagnostics of the most recent insert/delete on `stream` (synchronises the
 * stream).  Not part of the hot path. */
typedef struct mkv_update_info_s {
	uint32_t total;			/* requests in the call (0xffffffff: over the reserved capacity, nothing done) */
	uint32_t round1;		/* finished by the first grid-wide round */
	uint32_t round2;		/* finished by the second grid-wide round */
	uint32_t tail_parallel;	/* finished by the rounds inside the tail work-group */
	uint32_t tail_serial;	/* replayed one by one in canonical order */
	uint32_t undone;		/* optimistic commits rolled back behind an eviction barrier */
	uint32_t first_hard;	/* lowest request index found needing an eviction chain, or 0xffffffff */
	uint32_t chains;		/* eviction chains committed in the grid-wide rounds (conflict-free) */
	uint32_t bar_chain;		/* region barriers raised by chains that met another request's bucket */
	uint32_t bar_class;		/* region barriers raised for requests that might turn hard */
	uint32_t tail_rounds;	/* reservation rounds the tail launches ran */
	uint32_t helped;		/* round 1: second registrants of a bucket that committed beside its owner (self-help; counted in round1) */
	uint32_t losers;		/* round 1: requests it left to round 2 */
} mkv_update_info_t;

int mkv_last_update_info(void *stream, mkv_update_info_t *info);
/* cycle_insert != 0: the insert half of the last mkv_cycle on `stream` */
int mkv_last_update_info_ex(void *stream, int cycle_insert, mkv_update_info_t *info);
/* ---- diagnostics build only ---------------------------------------------------
 * (make -C megakv_amd/csrc DIAG=1 -> libmegakv_hip_diag.so, compiled with -DMKV_DIAGNOSTICS.  The
 * product library and lib/libgpuhash.a export none of these and refuse the tunings listed under
 * "diagnostics build" below.) */
#ifdef MKV_DIAGNOSTICS
/* diagnostics: microseconds the last call's tail kernels spent per phase (roll-back, gather,
 * exact replay below a barrier, retag, in-group rounds, serial replay, two gather details; then
 * inside the grid-wide rounds: register, meeting, decide, meeting, election + meeting, commit,
 * meeting, one spare), summed over the call's passes.  Synchronises the stream. */
int mkv_debug_tail_phases(void *stream, int cycle_insert, float out_us[16]);
/* diagnostics: the random-read rate of the memory system -- groups of 4 lanes read
 * pieces_per_group (a multiple of 8) pseudo-random gran-byte pieces (32 / 64 / 128 / 256) of buf
 * (buf_bytes a power of two), 4 pieces in flight per group as in the search kernel; returns
 * the bytes read (negative: error).  tools/random_read_probe.py times it. */
long long mkv_debug_random_read(const void *buf, long long buf_bytes, int gran, long long groups,
		long long pieces_per_group, uint32_t *sink, void *stream);
/* diagnostics: times this stream's workspace cleared its reservation tables because the
 * 40-bit round stamps ran low (the tail kernel flags it, a later call acts on it); -1: no
 * such workspace. */
int mkv_debug_stamp_resets(void *stream, int cycle_insert);
/* what the workspace holds for request idx after the last pass: out[0] = tag, out[1..4] = the
 * request cache {sig, hash, loc, flags}, out[5..8] = its undo record.  Synchronises the device. */
int mkv_debug_request(int cycle_insert, uint32_t idx, uint32_t out[9]);
/* what the last big pass's round 1 left for the next one: 1 = mostly plan-carrying (it runs on a lane per request),
 * 0 = not; negative: no such workspace.  Synchronises the device. */
int mkv_debug_wide_round(int cycle_insert);
#endif

/* ---- request preparation on the device ------------------
 * What the reference's receiver does per packet on the CPU, for a batch of
 * fixed-stride keys: sig64 = first 8 key bytes (src/mega_recv.c:349-350), with
 * fold != 0 XOR-folded over the rest of the key like the SIGNATURE build
 * (:351-360); out[i] = {sig = low word, hash = high word[, loc = locs[i]]}
 * (:361-362).  out_words: 2 = selem_t, 3 = ielem_t.  Bin the insert requests
 * into blocks with mkv_route_requests: (mask, shift, n) = (0xffffffff,
 * 32-IBLOCK_P, 2^IBLOCK_P) is the receiver's rule (:476-477), (HASH_MASK,
 * MEM_P-6-IBLOCK_P, 2^IBLOCK_P) the region rule of libgpuhash/test/insert_test.c. */
int mkv_derive_requests(const void *keys, int64_t num_keys, int key_len, int64_t key_stride, int fold,
		const loc_t *locs, void *out, int out_words, void *stream);

/* ---- result consumption on the device ---------------------
 * loc[i] = pairs[2i] ? pairs[2i] : pairs[2i+1]  (src/mega_send.c:408-414; 0 = miss,
 * :464-466).  counters: device uint64[66], scratch + results: counters[64] = hits,
 * counters[65] = misses after the call. */
int mkv_resolve_results(const loc_t *pairs, int64_t num_elem, loc_t *loc, uint64_t *counters, void *stream);

/* The same with the sender's KEY_MATCH check (src/mega_send.c:543-553): the item the picked
 * location names must hold the request's key -- a 32-bit signature has false positives.
 * compare: MKV_KEYCMP_STRNCMP = the reference's `strncmp(ITEM_key(it), key, key_len) == 0`
 * (:544), which stops at the first NUL -- the reference's own keys are binary and NUL-laden, so
 * keys that agree up to their first NUL are equal to it; MKV_KEYCMP_BYTES = all key_len bytes.
 * A flat item arena stands in for the slab store: item `l` starts at arena + l * item_stride
 * (l < arena_items), its uint16 key length at byte nkey_off and its key bytes at key_off
 * (the reference's item header: nkey_off 10, key_off 24, src/include/items.h:37-50).
 * keys: the requests' keys, key_stride bytes apart, key_len bytes each.  loc[i] = the
 * location if the keys agree, else 0.  counters: device uint64[68], scratch + results:
 * [64] hits, [65] misses (no location, or keys differ -- the reference counts both as a miss),
 * [66] false positives (a location whose item holds another key) after the call. */
#define MKV_KEYCMP_STRNCMP 0
#define MKV_KEYCMP_BYTES   1
int mkv_resolve_verify(const loc_t *pairs, int64_t num_elem, const void *keys, int64_t key_stride,
		int key_len, const void *arena, int64_t item_stride, int64_t arena_items, int nkey_off, int key_off,
		int compare, loc_t *loc, uint64_t *counters, void *stream);

/* ---- request routing for the sharded (multi-GPU) index, SURVEY 8(e) -------
 * shard(e) = ((e.hash & hash_mask_agg) >> shard_shift); requests are grouped by
 * shard, stably, on the device.  `perm[j]` = original index of the j-th
 * routed request; counts[s] = requests of shard s (device int64[nshards]). */
int mkv_route_requests(const void *in, int elem_words /* 2 = selem, 3 = ielem */,
		int64_t num_elem, uint32_t hash_mask_agg, uint32_t shard_shift, int nshards,
		void *routed_out, int32_t *perm, int64_t *counts, void *scratch,
		int64_t scratch_bytes, void *stream);

/* out[perm[j]] = routed[j] for 8-byte loc pairs (undo of the routing). */
int mkv_unroute_results(const loc_t *routed, const int32_t *perm, int64_t num_elem,
		loc_t *out, void *stream);

/* ---- padded exchange format of the sharded index (no host sync in a step) ----
 * A rank packs its batch into nshards equal slots, one per peer:
 *   slot = [n_get, n_set, 0, 0 | cap_get x selem_t | cap_set x ielem_t]  (uint32 words)
 * SETs keep batch order, so the canonical insert order of a shard is (source rank, index);
 * GETs of a slot are in no particular order.  perm_get[e] (n_get entries) = s * cap_get + r:
 * GET e went to shard s at position r (-1: dropped, its slot was full) -- its result
 * comes back at the same position (mkv_unpack_results).  `scratch` must be zero before the
 * first call (the kernels leave it zero).
 * *overflow is set to 1 when a slot cannot hold its requests (they are dropped). */
int64_t mkv_pack_slot_words(int64_t cap_get, int64_t cap_set);
int64_t mkv_pack_scratch_bytes(int64_t n_get, int64_t n_set, int nshards);
int mkv_pack_requests(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		void *stream);
/* Search every received slot (counts read from the slot headers on the device);
 * out is [nslots][cap_get][2]; misses are written as 0.  Also copies the slots'
 * n_set into set_counts[nslots] for the insert that follows. */
int mkv_search_slots(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out, const bucket_t *table, int *set_counts, void *stream);
/* Local phase of a sharded step in one call: search every received slot and insert
 * what the slots carry -- one insert block per source rank, counts read from the slot
 * headers on the device, the insert's mark pass fused into the search launch.
 * set_ptrs: device array of nslots device pointers to the slots' insert regions
 * (recvbuf + s*slot_words + 4 + 2*cap_get words).  size_hint as in mkv_insert_hinted. */
int mkv_slots_cycle(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out_slots, bucket_t *table, ielem_t **set_ptrs, long long size_hint,
		void *stream, void *timer, int slot);
/* out[e] = back[where[e]] for the n_get requests of the batch that was packed with
 * perm_get = where (a request that was dropped by an overflowing slot gets 0). */
int mkv_unpack_results(const loc_t *back, const int32_t *where, int64_t n_get, loc_t *out, void *stream);
/* The pack with a rank's own slot written elsewhere (self_slot: its receive buffer; self < 0: none) and,
 * with tile_meta (mkv_pack_meta_words(n_get) words, or NULL), a note of where every tile of GETs went -- one
 * contiguous run per shard: mkv_unpack_results_tiled then brings a tile's results back run by run, streamed
 * through LDS, instead of gathering 8 bytes per request.  tile_meta is written only by the pack that is ONE
 * launch (mkv_pack_one_launch(n_set, nshards, scratch_bytes) != 0: ask first); the un-pack must see the
 * same tuning "pack_threads" as the pack. */
int mkv_pack_one_launch(int64_t n_set, int nshards, int64_t scratch_bytes);
int64_t mkv_pack_meta_words(int64_t n_get);
int mkv_pack_requests_meta(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, uint32_t *tile_meta, void *stream);
int mkv_unpack_results_tiled(const loc_t *back, const int32_t *where, int64_t n_get, loc_t *out,
		const uint32_t *tile_meta, int nshards, int64_t cap_get, void *stream);

/* Request packets of the reference's protocol on the device (src/mega_recv.c:292-545,
 * README.md:70-81): payloads = npackets UDP payloads at `stride` bytes (a multiple of 4,
 * at most 2048 are looked at), payload_len = their lengths (device int32[], or NULL: all
 * `stride`).  GET requests go to gets ({sig, hash}, arrival order: packet, then job) with
 * get_packet[i] = their packet (may be NULL), SET requests to sets with loc = packet *
 * 2048 + byte offset of the key in its payload (the host replaces it by the item it
 * allocates).  Jobs whose key/value length differs from key_len/value_len are skipped
 * like the reference does.  counts (device u64[4]) = n_get, n_set, bad packets, skipped
 * jobs; requests beyond cap_get / cap_set are dropped (compare counts with the caps).
 * fold: XOR-fold key bytes beyond the first 8 (the reference's SIGNATURE build). */
int64_t mkv_parse_scratch_bytes(int64_t npackets);
int mkv_parse_packets(const void *payloads, int64_t stride, const int32_t *payload_len, int64_t npackets,
		int key_len, int value_len, int fold, selem_t *gets, int32_t *get_packet, ielem_t *sets,
		int64_t cap_get, int64_t cap_set, unsigned long long *counts, void *scratch, int64_t scratch_bytes,
		void *stream);

/* ---- one rank of the sharded index, driven from C (libmegakv_hip.so only; RCCL is
 * opened with dlopen(rccl_path), default "librccl.so") -------------------------------
 * geom: the LOCAL table's geometry, i.e. IBLOCK_P already reduced by log2(world);
 * n_get / n_set: the largest batch this rank hands in; cap_get / cap_set: slot capacity
 * per peer.  Create on every rank, make the id on rank 0 (mkv_shard_unique_id), hand the
 * 128 bytes to all ranks, connect (collective).  mkv_shard_run enqueues a whole run of
 * batches, two in flight (exchanges of one overlap the local kernels of the other; local
 * phases in batch order): gets[i] / sets[i] / outs[i] are device pointers (host arrays
 * of them), outs[i] receives n_get[i] x 2 loc_t in request order.  Nothing synchronises;
 * the lanes start after what is queued on `stream`, which waits for them at the end.
 * timer: mkv_timer_* handle bracketing the local search launch of every timer_every-th
 * batch (slot i / timer_every), or NULL. */
typedef struct mkv_shard mkv_shard_t;
int mkv_shard_create(mkv_shard_t **out, const mkv_geom_t *geom, int world, int rank, bucket_t *table,
		int64_t n_get, int64_t n_set, int64_t cap_get, int64_t cap_set, const char *rccl_path);
int mkv_shard_unique_id(mkv_shard_t *e, void *id128);
int mkv_shard_connect(mkv_shard_t *e, const void *id128);
int mkv_shard_run(mkv_shard_t *e, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int64_t *n_set, loc_t *const *outs, void *stream, void *timer,
		int timer_every);
int mkv_shard_overflowed(mkv_shard_t *e, int *flag);   /* synchronises the device */
/* Device time per phase of a batch (pack, exchange, local, un-pack; microseconds, mean over the spans
 * seen): mkv_shard_trace(n) arms events around every operation of n batches near the end of the next
 * runs, mkv_shard_phases() reads the means and disarms.  A pass of its own: the events cost time. */
void mkv_shard_trace(int nbatches);
int mkv_shard_phases(float out_us[4], int counts[4]);
const char *mkv_shard_error(const mkv_shard_t *e);
void mkv_shard_destroy(mkv_shard_t *e);

/* ---- the worker cycle as a three-stage pipeline over HOST buffers ----
 * What the reference's scheduler does per worker and batch on one stream -- requests host ->
 * device, search / insert, results device -> host (src/mega_scheduler.c:393-504, buffers
 * src/mega_recv.c:120-213, src/include/mega_batch.h:74-97) -- with the three stages on three
 * streams and `depth` sets of device buffers: the copy-in of batch i+1 and the copy-out of
 * batch i-1 run beside the kernels of batch i (PCIe is full duplex; pinned hipMemcpyAsync runs
 * on the DMA engines).  Results are those of the cycles run one after the other.
 * mkv_pipe_run: gets[i] / sets[i] / outs[i] are HOST pointers (pinned: cudaHostAlloc /
 * mkv_host_alloc), sets[i] holds the insert requests block after block with
 * blk_counts[i][0 .. num_blks) per block; outs[i] receives n_get[i] x 2 loc_t (resolve = 0)
 * or n_get[i] x loc_t (resolve = 1: loc = pair[0] ? pair[0] : pair[1], src/mega_send.c:408-414
 * done on the device, which halves the copy-out).  The host separates the pipeline's steps (one
 * synchronisation per step, as the reference's scheduler has one per cycle): the call waits for
 * what is queued on `stream`, and RETURNS WHEN THE LAST RESULT IS IN HOST MEMORY. */
typedef struct mkv_pipe mkv_pipe_t;
int mkv_pipe_create(mkv_pipe_t **out, const mkv_geom_t *geom, bucket_t *table, int64_t max_get, int64_t max_set,
		int num_blks, int depth);
int mkv_pipe_run(mkv_pipe_t *p, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int *const *blk_counts, loc_t *const *outs, int resolve, void *stream);
const char *mkv_pipe_error(const mkv_pipe_t *p);
void mkv_pipe_destroy(mkv_pipe_t *p);
void *mkv_host_alloc(int64_t bytes);    /* pinned host memory (hipHostMalloc) */
void mkv_host_free(void *ptr);

/* Tuning knobs (also environment variables MKV_<NAME>, read once at start-up):
 *   search:  "search_u" (1/2/4/8 requests in flight per lane group), "search_nt" (non-temporal
 *            table loads);
 *   update:  "tail_rounds" (0..48 reservation rounds of the tail), "tail_groups" (work-groups of
 *            the tail launch, 1..64), "grid_min" (a final tail list longer than this runs its
 *            rounds on all of them), "insert_chunk" (requests per pass of a hinted insert),
 *            "calm_factor" (passes that many times larger while no barrier was met lately),
 *            "barrier_sticky", "barrier_split", "mark_blocks" (mark work-groups inside the fused
 *            search launch), "mark_alone_blocks" (of a mark launch of its own), "round_blocks"
 *            (work-groups of rounds 1 and 2; 512: more lanes in flight were slower),
 *            "grid_progress" (grid-wide tail rounds go on while they resolve at least this many
 *            requests), "req_cap" (default request capacity of an update workspace), "dup_groups",
 *            "lazy_alt", "self_help" (round 1: a bucket's second registrant commits beside its
 *            owner), "merge_round2" (round 2 inside the tail launch: 0 never, 1 while round 1
 *            leaves it next to nothing, 2 always), "merge_max" (... and the batch is no larger);
 *   sharded: "pack_threads" (lanes per work-group of the one-launch pack: 256 / 512 / 1024).
 * Diagnostics build only (MKV_EINVAL in the product library):
 *            "search_wide" (8 lanes per request, one 64-byte access per bucket), "search_lds"
 *            (signature rows staged in LDS and probed with a wave ballot, the location fetched by
 *            a second load: the reference's structure) -- the two search layouts that were built,
 *            measured and withdrawn; "pack_three_pass" (forces the pack's
 *            fall-back); "debug_first_stamp"; "debug_skip" (1 / 2: the tail takes the general way /
 *            the region-suffix fall-back for every order violation of an eviction chain; + 4:
 *            the grid-wide rounds leave every eviction chain to the tail).
 * Results never depend on them; they exist for measurement and tests. */
int mkv_set_tuning(const char *name, long long value);

const char *mkv_version(void);

#ifdef __cplusplus
}
#endif
#endif
