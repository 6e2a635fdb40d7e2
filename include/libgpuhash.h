/*
 * libgpuhash.h -- C ABI of the batched GPU hash index (MI355X build).
 *
 * Each entry point replaces the function of the same name exported by the
 * reference's CUDA library; the signatures are identical, so an existing C
 * host links unchanged against lib/libgpuhash.a:
 *
 *   gpu_hash_search   reference libgpuhash/libgpuhash.h:29-36, gpu_hash.cu:482-518
 *   gpu_hash_insert   reference libgpuhash/libgpuhash.h:38-43, gpu_hash.cu:521-556
 *   gpu_hash_delete   reference libgpuhash/libgpuhash.h:45-51, gpu_hash.cu:558-593
 *
 * Contract (reference behaviour, kept):
 *   - every pointer argument is a DEVICE pointer, including blk_input (device
 *     array of device pointers) and blk_elem_num (device int[]);
 *   - calls are asynchronous: they return after enqueueing work on `stream`
 *     (0 = default stream); ordering is stream order; nothing is allocated or
 *     freed by the caller-visible path and no status is returned;
 *   - gpu_hash_search writes out[2i] / out[2i+1] only on a hit in the primary /
 *     alternate bucket: the caller zeroes `out` first (ref gpu_hash.cu:50-53);
 *   - num_thread / threads_per_blk are launch hints of the CUDA version.  They
 *     do not affect results here, but the reference's argument checks are kept:
 *     search aborts unless num_thread > threads_per_blk && threads_per_blk <= 1024
 *     (ref :492-493), delete unless num_thread >= threads_per_blk (ref :567-568);
 *   - MEM_P and HASH_CUCKOO / HASH_2CHOICE are baked in at library build time.
 *
 * Where this build differs from "the library is stateless and allocates nothing":
 *   - gpu_hash_insert / gpu_hash_delete keep a device workspace (reservation tables, request
 *     state), created on the first call that needs it -- call mkv_reserve() (megakv_hip.h) at
 *     start-up to keep the allocation out of the request path;
 *   - the workspace holds MKV_REQ_CAP requests (environment variable, default 4 194 304; the
 *     reference's hosts stay below 32 768 per call, src/mega.c:143-144,163-165) unless the library
 *     knows the size of a batch.  gpu_hash_delete's count is an argument.  gpu_hash_insert's counts
 *     are device memory -- but a host that wrote them the reference's way, with cudaMemcpy /
 *     cudaMemcpyAsync from host memory through this library (src/mega_scheduler.c:487-494), has
 *     shown them to it on the way: a batch of any size up to 16 777 200 requests is then sized, the
 *     workspace grown and the batch run in passes.  Only counts the library never saw (written by a
 *     kernel of the host's) leave it blind: such a batch beyond the workspace is detected on the
 *     device and NOT applied; the next insert/delete call reports it on stderr, once per dropped
 *     batch, and mkv_dropped_batches() (megakv_hip.h) counts them -- nothing aborts.
 *     mkv_insert_hinted takes the batch size as an argument.
 *
 * Result definition: search output and final table state are bit-identical to
 * the reference kernels executed one request at a time in (block, index) order
 * -- the only order for which the reference itself is deterministic.
 */
#ifndef _LIBGPUHASH_H_
#define _LIBGPUHASH_H_

#include <stdio.h>
#include <stdlib.h>
#include <cuda_runtime.h>
#include "gpu_hash.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Parameter names below describe what is passed; the reference's names are given in brackets
 * where they differ (ref libgpuhash.h:29-62).  Types and order are the reference's. */

/* Probe `count` [num_elem] requests: results[2i], results[2i+1] <- location found in the
 * primary / alternate bucket of request i [in, out, hash_table]. */
void gpu_hash_search(selem_t *requests, loc_t *results, bucket_t *table, int count, int num_thread,
		int threads_per_blk, cudaStream_t stream);

/* Insert `blocks` [num_blks] blocks: block_requests[b] holds block_counts[b] requests
 * [blk_input, blk_elem_num] -- both arrays live in device memory. */
void gpu_hash_insert(bucket_t *table, ielem_t **block_requests, int *block_counts, int blocks,
		cudaStream_t stream);

/* Empty the slots that hold exactly (sig, loc) of each of the `count` requests. */
void gpu_hash_delete(delem_t *requests, bucket_t *table, int count, int num_thread, int threads_per_blk,
		cudaStream_t stream);

/* Declared by the reference (libgpuhash.h:53-62) but defined nowhere and never called.
 * Provided here as delete-then-insert on the same stream so that a host referencing it links. */
void gpu_delete_insert(bucket_t *table, delem_t *deletes, uint32_t delete_count, ielem_t **block_requests,
		int *block_counts, int blocks, uint32_t num_delete_thread, uint32_t threads_per_blk,
		cudaStream_t stream);

#ifdef __cplusplus
}
#endif

/* The reference's hosts wrap their runtime calls in CUDA_SAFE_CALL (ref libgpuhash.h:64-70):
 * on failure, the same message on stderr and exit(EXIT_FAILURE). */
static inline void libgpuhash_check_(cudaError_t status, const char *file, int line)
{
	if (status == cudaSuccess)
		return;
	fprintf(stderr, "Cuda error in file '%s' in line %i : %s.\n", file, line, cudaGetErrorString(status));
	exit(EXIT_FAILURE);
}
#define CUDA_SAFE_CALL(call) libgpuhash_check_((call), __FILE__, __LINE__)

/* the call, then a device synchronisation, both checked (the reference's macro of this name,
 * ref :72-79, does not compile and is used nowhere) */
static inline void libgpuhash_check_sync_(cudaError_t status, const char *file, int line)
{
	libgpuhash_check_(status, file, line);
	libgpuhash_check_(cudaDeviceSynchronize(), file, line);
}
#define CUDA_SAFE_CALL_SYNC(call) libgpuhash_check_sync_((call), __FILE__, __LINE__)

#endif
