/*
 * cpu_abi.c -- the CPU oracle behind the reference's own C boundary: libgpuhash_cpu.a.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  It exists so that programs
 * written against the reference's library -- first of all the reference's own test driver
 * libgpuhash/test/insert_test.c:45-244, compiled unchanged where it lies -- can run against
 * the ORACLE on the host cores, with no GPU: the one check of the oracle by execution of
 * reference-held code that this image allows (make -C oracle ref_cpu; tests/test_oracle_ref_cpu.py).
 * Nothing under megakv_amd/ links it, and it never travels as product: the archive is written
 * to oracle/_build/, the driver built from the reference's source to oracle/_ref/.
 *
 * What it provides:
 *   gpu_hash_search / gpu_hash_insert / gpu_hash_delete / gpu_delete_insert
 *       -- the entry points of reference libgpuhash/libgpuhash.h:29-62 (wrappers
 *       libgpuhash/gpu_hash.cu:482-593), every "device" pointer being a host pointer, run by
 *       orc_search / orc_insert / orc_delete in canonical order; the wrappers'
 *       argument asserts (gpu_hash.cu:492-493, 567-568) are kept;
 *   the dozen CUDA host names include/cuda_runtime.h declares, on host memory: cudaMalloc is an
 *       aligned allocation, copies are memcpy, streams are tokens, every call is synchronous
 *       (which satisfies "asynchronous on the stream": nothing can observe the difference).
 * MEM_P and HASH_CUCKOO / HASH_2CHOICE are baked in at build time like in the reference
 * (gpu_hash.h:54-79); IBLOCK_P and MAX_CUCKOO_NUM come from the same header.
 */
#include <assert.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/libgpuhash.h"
#include "gpuhash_oracle.h"

static orc_geom_t baked_geometry(void)
{
	orc_geom_t g;
	g.mem_p = MEM_P;
	g.iblock_p = IBLOCK_P;
#if defined(HASH_2CHOICE)
	g.scheme = ORC_SCHEME_2CHOICE;
	g.max_cuckoo = 5;
#else
	g.scheme = ORC_SCHEME_CUCKOO;
	g.max_cuckoo = MAX_CUCKOO_NUM;
#endif
	return g;
}

/* ref gpu_hash.cu:482-518: `out` is written on hits only -- the caller has zeroed it. */
void gpu_hash_search(selem_t *requests, loc_t *results, bucket_t *table, int count, int num_thread,
		int threads_per_blk, cudaStream_t stream)
{
	(void)stream;
	assert(num_thread > threads_per_blk);	/* ref :492 */
	assert(threads_per_blk <= 1024);		/* ref :493 */
	orc_geom_t g = baked_geometry();
	orc_search(&g, (const orc_bucket_t *)table, (const orc_selem_t *)requests, results, count);
}

/* ref gpu_hash.cu:521-556: one CUDA block per insert block; canonical order = block, then index. */
void gpu_hash_insert(bucket_t *table, ielem_t **block_requests, int *block_counts, int blocks,
		cudaStream_t stream)
{
	(void)stream;
	orc_geom_t g = baked_geometry();
	orc_insert(&g, (orc_bucket_t *)table, (const orc_ielem_t *const *)block_requests, block_counts, blocks, NULL);
}

/* ref gpu_hash.cu:558-593 */
void gpu_hash_delete(delem_t *requests, bucket_t *table, int count, int num_thread, int threads_per_blk,
		cudaStream_t stream)
{
	(void)stream;
	assert(num_thread >= threads_per_blk);	/* ref :567 */
	assert(threads_per_blk <= 1024);		/* ref :568 */
	orc_geom_t g = baked_geometry();
	orc_delete(&g, (orc_bucket_t *)table, (const orc_ielem_t *)requests, count, NULL);
}

/* declared by the reference (libgpuhash.h:53-62), defined nowhere: delete, then insert, as the
 * product library defines it */
void gpu_delete_insert(bucket_t *table, delem_t *deletes, uint32_t delete_count, ielem_t **block_requests,
		int *block_counts, int blocks, uint32_t num_delete_thread, uint32_t threads_per_blk,
		cudaStream_t stream)
{
	if (delete_count)
		gpu_hash_delete(deletes, table, (int)delete_count, (int)num_delete_thread, (int)threads_per_blk, stream);
	gpu_hash_insert(table, block_requests, block_counts, blocks, stream);
}

/* ---- the CUDA host names on host memory ------------------------------------------------- */

enum { kOutOfMemory = 2 };	/* the value cudaErrorMemoryAllocation carries */

cudaError_t cudaMalloc(void **dev_ptr, size_t size)
{
	/* bucket_t is 128-byte aligned in the reference's header (gpu_hash.h:82) */
	size_t rounded = (size + 127) & ~(size_t)127;
	*dev_ptr = aligned_alloc(128, rounded ? rounded : 128);
	return *dev_ptr ? cudaSuccess : kOutOfMemory;
}

cudaError_t cudaFree(void *dev_ptr)
{
	free(dev_ptr);
	return cudaSuccess;
}

cudaError_t cudaHostAlloc(void **host_ptr, size_t size, unsigned int flags)
{
	(void)flags;
	return cudaMalloc(host_ptr, size);
}

cudaError_t cudaFreeHost(void *host_ptr)
{
	free(host_ptr);
	return cudaSuccess;
}

cudaError_t cudaMemset(void *dev_ptr, int value, size_t count)
{
	memset(dev_ptr, value, count);
	return cudaSuccess;
}

cudaError_t cudaMemsetAsync(void *dev_ptr, int value, size_t count, cudaStream_t stream)
{
	(void)stream;
	return cudaMemset(dev_ptr, value, count);
}

cudaError_t cudaMemcpy(void *dst, const void *src, size_t count, enum cudaMemcpyKind kind)
{
	(void)kind;
	memmove(dst, src, count);
	return cudaSuccess;
}

cudaError_t cudaMemcpyAsync(void *dst, const void *src, size_t count, enum cudaMemcpyKind kind, cudaStream_t stream)
{
	(void)stream;
	return cudaMemcpy(dst, src, count, kind);
}

cudaError_t cudaStreamCreate(cudaStream_t *stream)
{
	static uintptr_t next_token = 1;
	*stream = (cudaStream_t)(next_token++ << 4);	/* a token: never dereferenced */
	return cudaSuccess;
}

cudaError_t cudaStreamDestroy(cudaStream_t stream)
{
	(void)stream;
	return cudaSuccess;
}

cudaError_t cudaStreamSynchronize(cudaStream_t stream)
{
	(void)stream;
	return cudaSuccess;
}

cudaError_t cudaDeviceSynchronize(void)
{
	return cudaSuccess;
}

cudaError_t cudaGetLastError(void)
{
	return cudaSuccess;
}

cudaError_t cudaMemGetInfo(size_t *free_bytes, size_t *total_bytes)
{
	*free_bytes = *total_bytes = (size_t)1 << 40;
	return cudaSuccess;
}

const char *cudaGetErrorString(cudaError_t error)
{
	return error == cudaSuccess ? "no error" : error == kOutOfMemory ? "out of memory" : "error";
}
