// mkv_abi.hip -- the drop-in entry points: same names and signatures as the
// reference's extern "C" wrappers (libgpuhash/gpu_hash.cu:482-593), geometry
// and insert scheme baked in from gpu_hash.h at build time like the reference.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>

#include "libgpuhash.h"
#include "megakv_hip.h"

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

extern "C" const char *mkv_version(void)
{
	return "megakv_amd libgpuhash gfx950 r1";
}
