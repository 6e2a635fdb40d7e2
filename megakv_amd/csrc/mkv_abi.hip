// mkv_abi.hip -- the drop-in entry points: same names and signatures as the
// reference's extern "C" wrappers (libgpuhash/gpu_hash.cu:482-593), geometry
// and insert scheme baked in from gpu_hash.h at build time like the reference.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "libgpuhash.h"
#include "megakv_hip.h"
#include "mkv_common.h"

static const mkv_geom_t kGeom = {
	MEM_P, IBLOCK_P,
#if defined(HASH_2CHOICE)
	MKV_SCHEME_2CHOICE, 0
#else
	MKV_SCHEME_CUCKOO, MAX_CUCKOO_NUM
#endif
};

// The reference never checks a launch; a failure here cannot be reported
// through the void ABI, so it is made loud instead of silent.
static void die_on(int rc, const char *what)
{
	if (rc != MKV_OK) {
		fprintf(stderr, "libgpuhash (MI355X): %s failed with code %d\n", what, rc);
		abort();
	}
}

extern "C" void gpu_hash_search(selem_t *in, loc_t *out, bucket_t *hash_table, int num_elem,
		int num_thread, int threads_per_blk, cudaStream_t stream)
{
	// argument checks of the reference wrapper (gpu_hash.cu:492-498); the
	// values themselves are CUDA launch hints and do not shape the HIP launch
	assert(num_thread > threads_per_blk);
	assert(threads_per_blk <= 1024);
	(void)num_thread;
	(void)threads_per_blk;
	die_on(mkv_search(&kGeom, in, out, hash_table, num_elem, MKV_SEARCH_KEEP_MISS, (void *)stream),
			"gpu_hash_search");
}

extern "C" void gpu_hash_insert(bucket_t *hash_table, ielem_t **blk_input, int *blk_elem_num,
		int num_blks, cudaStream_t stream)
{
	assert(ELEM_NUM_P < 5 && ELEM_NUM_P > 0);   // gpu_hash.cu:535
	die_on(mkv_insert(&kGeom, hash_table, blk_input, blk_elem_num, num_blks, (void *)stream),
			"gpu_hash_insert");
}

extern "C" void gpu_hash_delete(delem_t *in, bucket_t *hash_table, int num_elem, int num_thread,
		int threads_per_blk, cudaStream_t stream)
{
	assert(num_thread >= threads_per_blk);      // gpu_hash.cu:567-568
	assert(threads_per_blk <= 1024);
	(void)num_thread;
	(void)threads_per_blk;
	die_on(mkv_delete(&kGeom, in, hash_table, num_elem, (void *)stream), "gpu_hash_delete");
}

extern "C" void gpu_delete_insert(bucket_t *hash_table, delem_t *delete_in, uint32_t num_delete_job,
		ielem_t **insert_blk_input, int *insert_blk_elem_num, int num_insert_blks,
		uint32_t num_delete_thread, uint32_t threads_per_blk, cudaStream_t stream)
{
	(void)num_delete_thread;
	(void)threads_per_blk;
	die_on(mkv_delete(&kGeom, delete_in, hash_table, num_delete_job, (void *)stream),
			"gpu_delete_insert/delete");
	die_on(mkv_insert(&kGeom, hash_table, insert_blk_input, insert_blk_elem_num, num_insert_blks,
				(void *)stream), "gpu_delete_insert/insert");
}

// HIP event pairs that receive the start and end time of the search dispatch of mkv_cycle
// (hipExtLaunchKernelGGL on the caller's stream): how bench.py times the dominant kernel.
extern "C" void *mkv_timer_create(int slots)
{
	if (slots <= 0)
		return nullptr;
	MkvTimer *t = new MkvTimer;
	t->slots = slots;
	t->ev = new hipEvent_t[2 * slots];
	for (int i = 0; i < 2 * slots; i++) {
		if (hipEventCreate(&t->ev[i]) != hipSuccess) {
			for (int j = 0; j < i; j++)
				(void)hipEventDestroy(t->ev[j]);
			delete[] t->ev;
			delete t;
			return nullptr;
		}
	}
	return t;
}

extern "C" int mkv_timer_elapsed_ms(void *timer, int slot, float *ms)
{
	MkvTimer *t = reinterpret_cast<MkvTimer *>(timer);
	if (!t || !ms || slot < 0 || slot >= t->slots)
		return MKV_EINVAL;
	return hipEventElapsedTime(ms, t->ev[2 * slot], t->ev[2 * slot + 1]) == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" void mkv_timer_destroy(void *timer)
{
	MkvTimer *t = reinterpret_cast<MkvTimer *>(timer);
	if (!t)
		return;
	for (int i = 0; i < 2 * t->slots; i++)
		(void)hipEventDestroy(t->ev[i]);
	delete[] t->ev;
	delete t;
}

extern "C" int mkv_tuning_is_default_search(void);
extern "C" void mkv_call_lock(void);
extern "C" void mkv_call_unlock(void);
extern "C" int mkv_cycle_insert_begin(const mkv_geom_t *geom, const bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);
extern "C" int mkv_cycle_insert_end(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream, int joined);
extern "C" int mkv_cycle_search_mark(const mkv_geom_t *geom, const selem_t *in, loc_t *out,
		const bucket_t *table, int64_t num_search, ielem_t **blk_input, int *blk_elem_num, int num_blks,
		long long size_hint, void *stream, void *ev_start, void *ev_stop);
extern "C" int mkv_search_timed(const mkv_geom_t *geom, const selem_t *in, loc_t *out, const bucket_t *table,
		int64_t num_elem, uint32_t flags, void *stream, void *ev_start, void *ev_stop);
extern "C" int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);

// One scheduler cycle for one worker in a single call: clear the result buffer,
// search, delete, insert, all on `stream` (reference src/mega_scheduler.c:393-502
// does exactly this sequence per worker).  Saves host round trips for callers
// that are not bound to the three-call ABI.
extern "C" int mkv_cycle(const mkv_geom_t *geom, bucket_t *table, const selem_t *search_in,
		loc_t *search_out, int64_t num_search, delem_t *delete_in, int64_t num_delete,
		ielem_t **blk_input, int *blk_elem_num, int num_blks, int64_t insert_hint, uint32_t search_flags,
		void *stream, void *timer, int slot)
{
	int rc = MKV_OK;
	MkvTimer *tm = reinterpret_cast<MkvTimer *>(timer);
	if (tm && (slot < 0 || slot >= tm->slots))
		return MKV_EINVAL;
	// host threads that run cycles at once (each on its own stream) take turns for the whole call: the
	// parts below share the device's cycle workspace from the fused launch to the rounds
	struct Turn {
		Turn() { mkv_call_lock(); }
		~Turn() { mkv_call_unlock(); }
	} turn;
	// The insert's mark pass never writes the table (its pre-walk reads it: a hint only, so a
	// delete of the same cycle running beside it is harmless).  Default: it rides in the
	// search launch (extra work-groups of one kernel).  MKV_CYCLE_SIDE_STREAM: it
	// runs on a side stream beside the search kernel.  MKV_CYCLE_NO_OVERLAP: plain
	// sequence of the three entry points.
	const bool can = num_blks > 0 && !(search_flags & MKV_CYCLE_NO_OVERLAP);
	const bool side = can && (search_flags & MKV_CYCLE_SIDE_STREAM);
	const bool fused = can && !side && num_search > 0 && !(search_flags & MKV_CYCLE_MEMSET) &&
			mkv_tuning_is_default_search();
	const bool overlap = side;
	if (side) {
		rc = mkv_cycle_insert_begin(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream);
		if (rc != MKV_OK)
			return rc;
	}
	if (fused) {
		rc = mkv_cycle_search_mark(geom, search_in, search_out, table, num_search, blk_input, blk_elem_num,
				num_blks, insert_hint, stream, tm ? tm->ev[2 * slot] : nullptr, tm ? tm->ev[2 * slot + 1] : nullptr);
		if (rc != MKV_OK)
			return rc;
	} else if (num_search > 0) {
		if (search_flags & MKV_CYCLE_MEMSET) {
			if (hipMemsetAsync(search_out, 0, (size_t)num_search * 2 * sizeof(loc_t),
						(hipStream_t)stream) != hipSuccess)
				return MKV_EHIP;
		}
		// memset + write-on-hit and write-always give the same buffer: fused unless asked
		rc = mkv_search_timed(geom, search_in, search_out, table, num_search,
				(search_flags & MKV_CYCLE_MEMSET) ? MKV_SEARCH_KEEP_MISS : MKV_SEARCH_ZERO_MISS, stream,
				tm ? tm->ev[2 * slot] : nullptr, tm ? tm->ev[2 * slot + 1] : nullptr);
		if (rc != MKV_OK)
			return rc;
	}
	if (num_delete > 0) {
		rc = mkv_delete(geom, delete_in, table, num_delete, stream);
		if (rc != MKV_OK)
			return rc;
	}
	if (overlap || fused)
		rc = mkv_cycle_insert_end(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream,
				(overlap ? 1 : 0) | (num_delete > 0 ? 2 : 0));
	else if (num_blks > 0)
		rc = mkv_insert_hinted(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream);
	return rc;
}

extern "C" const char *mkv_version(void)
{
	return "megakv_amd libgpuhash gfx950 r4";
}
