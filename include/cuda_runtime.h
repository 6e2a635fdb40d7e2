/*
 * cuda_runtime.h -- the dozen CUDA runtime names Mega-KV's C host uses,
 * declared for a plain C compiler and implemented over HIP inside
 * lib/libgpuhash.a (megakv_amd/csrc/cuda_names.hip).
 *
 * The reference's libgpuhash.h does `#include <cuda_runtime.h>`
 * (libgpuhash/libgpuhash.h:26) and its host files call:
 *   cudaMalloc / cudaMemset            src/mega_scheduler.c:273-274, src/mega.c:290-291
 *   cudaHostAlloc                      src/mega_recv.c:132-178
 *   cudaStreamCreate                   src/mega_scheduler.c:279
 *   cudaMemcpy                         src/mega_recv.c:186-191
 *   cudaMemcpyAsync / cudaMemsetAsync  src/mega_scheduler.c:404-419,452,487-494
 *   cudaDeviceSynchronize              src/mega_scheduler.c:504
 *   cudaGetErrorString                 libgpuhash/libgpuhash.h:64-70 (CUDA_SAFE_CALL)
 * With this header on the include path those files compile with gcc as they
 * are.  This is not a CUDA compatibility layer: no kernel-launch syntax, no
 * device-side names, only the host calls listed above (plus their free /
 * destroy / query counterparts), each a direct call of the HIP runtime.
 * cudaStream_t IS a hipStream_t, so streams made here can be handed to HIP
 * code and vice versa.
 */
#ifndef MEGAKV_AMD_CUDA_RUNTIME_H
#define MEGAKV_AMD_CUDA_RUNTIME_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *cudaStream_t;	/* same object as hipStream_t */
typedef int cudaError_t;					/* carries the hipError_t value */

enum { cudaSuccess = 0 };

enum cudaMemcpyKind {
	cudaMemcpyHostToHost = 0,
	cudaMemcpyHostToDevice = 1,
	cudaMemcpyDeviceToHost = 2,
	cudaMemcpyDeviceToDevice = 3,
	cudaMemcpyDefault = 4
};

#define cudaHostAllocDefault 0x00

cudaError_t cudaMalloc(void **dev_ptr, size_t size);
cudaError_t cudaFree(void *dev_ptr);
cudaError_t cudaHostAlloc(void **host_ptr, size_t size, unsigned int flags);
cudaError_t cudaFreeHost(void *host_ptr);
cudaError_t cudaMemset(void *dev_ptr, int value, size_t count);
cudaError_t cudaMemsetAsync(void *dev_ptr, int value, size_t count, cudaStream_t stream);
cudaError_t cudaMemcpy(void *dst, const void *src, size_t count, enum cudaMemcpyKind kind);
cudaError_t cudaMemcpyAsync(void *dst, const void *src, size_t count,
		enum cudaMemcpyKind kind, cudaStream_t stream);
cudaError_t cudaStreamCreate(cudaStream_t *stream);
cudaError_t cudaStreamDestroy(cudaStream_t stream);
cudaError_t cudaStreamSynchronize(cudaStream_t stream);
cudaError_t cudaDeviceSynchronize(void);
cudaError_t cudaGetLastError(void);
cudaError_t cudaMemGetInfo(size_t *free_bytes, size_t *total_bytes);	/* (examples/scheduler_cycle.c: the library's memory) */
const char *cudaGetErrorString(cudaError_t error);

#ifdef __cplusplus
}
#endif

#endif
