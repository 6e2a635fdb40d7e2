// cuda_names.hip -- the host-side CUDA runtime names Mega-KV's C host calls
// (listed in include/cuda_runtime.h), each a direct call of the HIP runtime.
// Part of the drop-in boundary only; no kernel in this library is written
// against these names.
#include <hip/hip_runtime.h>

#include "cuda_runtime.h"

static hipMemcpyKind kind_of(enum cudaMemcpyKind k)
{
	switch (k) {
	case cudaMemcpyHostToHost: return hipMemcpyHostToHost;
	case cudaMemcpyHostToDevice: return hipMemcpyHostToDevice;
	case cudaMemcpyDeviceToHost: return hipMemcpyDeviceToHost;
	case cudaMemcpyDeviceToDevice: return hipMemcpyDeviceToDevice;
	default: return hipMemcpyDefault;
	}
}

// The deferred mode of the drop-in entry points (mkv_abi.hip, MKV_DEFER=1): what the three entry points have
// recorded is issued by whatever makes the host wait, and a device-to-host copy behind recorded work waits
// its turn.  Without the mode mkv_flush() finds nothing and mkv_defer_copy() says "now".
extern "C" int mkv_flush(void);
extern "C" int mkv_defer_copy(void *dst, const void *src, size_t n, void *stream);
extern "C" int mkv_defer_copy_in(void *dst, const void *src, size_t n, void *stream);
extern "C" void mkv_pinned_note(void *p, size_t n, int add);
extern "C" void mkv_note_h2d(void *dst, const void *src, size_t n);
extern "C" void mkv_note_write(void *dst, size_t n);
extern "C" int mkv_defer_memset(void *p, int value, size_t n, void *stream);
extern "C" void mkv_defer_touch(void);

static cudaError_t flushed(void)
{
	return mkv_flush() == 0 ? (cudaError_t)hipSuccess : (cudaError_t)hipErrorLaunchFailure;
}

extern "C" {

cudaError_t cudaMalloc(void **p, size_t n) { return (cudaError_t)hipMalloc(p, n); }
cudaError_t cudaFree(void *p)
{
	const cudaError_t f = flushed();
	return f ? f : (cudaError_t)hipFree(p);
}
cudaError_t cudaHostAlloc(void **p, size_t n, unsigned int flags)
{
	// (pinned buffers made here are known to the deferred mode: copies to and from them can wait for the cycle
	// and go as ONE launch, mkv_abi.hip)
	const cudaError_t e = (cudaError_t)hipHostMalloc(p, n, flags);
	if (!e && p && *p)
		mkv_pinned_note(*p, n, 1);
	return e;
}
cudaError_t cudaFreeHost(void *p)
{
	const cudaError_t f = flushed();
	mkv_pinned_note(p, 0, 0);
	return f ? f : (cudaError_t)hipHostFree(p);
}
cudaError_t cudaMemset(void *p, int v, size_t n)
{
	const cudaError_t f = flushed();
	mkv_note_write(p, n);
	return f ? f : (cudaError_t)hipMemset(p, v, n);
}
cudaError_t cudaMemsetAsync(void *p, int v, size_t n, cudaStream_t s)
{
	mkv_note_write(p, n);
	if (mkv_defer_memset(p, v, n, (void *)s))
		return (cudaError_t)hipSuccess;     // (held: the search that follows may make it superfluous, mkv_abi.hip)
	return (cudaError_t)hipMemsetAsync(p, v, n, (hipStream_t)s);
}
// (what the host copies into small device arrays is remembered -- gpu_hash_insert's block counts: mkv_abi.hip)
static void note_copy(void *d, const void *s, size_t n, enum cudaMemcpyKind k)
{
	if (k == cudaMemcpyHostToDevice)
		mkv_note_h2d(d, s, n);
	else if (k == cudaMemcpyDeviceToDevice || k == cudaMemcpyDefault)
		mkv_note_write(d, n);
}
cudaError_t cudaMemcpy(void *d, const void *s, size_t n, enum cudaMemcpyKind k)
{
	const cudaError_t f = flushed();
	note_copy(d, s, n, k);
	return f ? f : (cudaError_t)hipMemcpy(d, s, n, kind_of(k));
}
cudaError_t cudaMemcpyAsync(void *d, const void *s, size_t n, enum cudaMemcpyKind k, cudaStream_t st)
{
	note_copy(d, s, n, k);
	if (k == cudaMemcpyDeviceToHost) {
		if (mkv_defer_copy(d, s, n, (void *)st))
			return (cudaError_t)hipSuccess;
	} else if (k == cudaMemcpyHostToDevice && mkv_defer_copy_in(d, s, n, (void *)st)) {
		return (cudaError_t)hipSuccess;     // (from a pinned buffer, deferred mode: it goes with the cycle's other copies)
	} else {
		mkv_defer_touch();
	}
	return (cudaError_t)hipMemcpyAsync(d, s, n, kind_of(k), (hipStream_t)st);
}
cudaError_t cudaStreamCreate(cudaStream_t *s) { return (cudaError_t)hipStreamCreate((hipStream_t *)s); }
cudaError_t cudaStreamDestroy(cudaStream_t s)
{
	const cudaError_t f = flushed();
	return f ? f : (cudaError_t)hipStreamDestroy((hipStream_t)s);
}
cudaError_t cudaStreamSynchronize(cudaStream_t s)
{
	const cudaError_t f = flushed();
	return f ? f : (cudaError_t)hipStreamSynchronize((hipStream_t)s);
}
cudaError_t cudaDeviceSynchronize(void)
{
	const cudaError_t f = flushed();
	return f ? f : (cudaError_t)hipDeviceSynchronize();
}
cudaError_t cudaGetLastError(void) { return (cudaError_t)hipGetLastError(); }
cudaError_t cudaMemGetInfo(size_t *f, size_t *t) { return (cudaError_t)hipMemGetInfo(f, t); }
const char *cudaGetErrorString(cudaError_t e) { return hipGetErrorString((hipError_t)e); }

}
