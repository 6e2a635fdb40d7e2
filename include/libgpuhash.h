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

void gpu_hash_search(
		selem_t 	*in,
		loc_t		*out,
		bucket_t	*hash_table,
		int			num_elem,
		int			num_thread,
		int			threads_per_blk,
		cudaStream_t stream);

void gpu_hash_insert(
		bucket_t	*hash_table,
		ielem_t		**blk_input,
		int			*blk_elem_num,
		int			num_blks,
		cudaStream_t stream);

void gpu_hash_delete(
		delem_t 	*in,
		bucket_t	*hash_table,
		int			num_elem,
		int 		num_thread,
		int			threads_per_blk,
		cudaStream_t stream);

/* Declared by the reference (libgpuhash.h:53-62) but defined nowhere and never
 * called.  Provided here as delete-then-insert on the same stream so that a
 * host referencing it links. */
void gpu_delete_insert(
		bucket_t		*hash_table,
		delem_t			*delete_in,
		uint32_t		num_delete_job,
		ielem_t			**insert_blk_input,
		int				*insert_blk_elem_num,
		int				num_insert_blks,
		uint32_t		num_delete_thread,
		uint32_t		threads_per_blk,
		cudaStream_t	stream);

#ifdef __cplusplus
}
#endif

#define CUDA_SAFE_CALL(call) do {                                            \
	cudaError_t err = call;                                                    \
	if(cudaSuccess != err) {                                                \
		fprintf(stderr, "Cuda error in file '%s' in line %i : %s.\n",        \
				 __FILE__, __LINE__, cudaGetErrorString( err) );             \
		exit(EXIT_FAILURE);                                                  \
	} } while (0)

#define CUDA_SAFE_CALL_SYNC(call) do {                                       \
	cudaError_t err_s = call;                                                  \
	if (cudaSuccess == err_s) err_s = cudaDeviceSynchronize();                \
	if(cudaSuccess != err_s) {                                              \
		fprintf(stderr, "Cuda error in file '%s' in line %i : %s.\n",        \
				__FILE__, __LINE__, cudaGetErrorString( err_s) );            \
		exit(EXIT_FAILURE);                                                  \
	} } while (0)
#endif
