// Would it pay to let the LAST work-group of round 1 do the little that rounds 2 and the tail
// do on a 52 K batch, instead of launching them?  Stand-ins: a "search" of 27 us, then
//   A) three launches: 512 work-groups with three dependent random loads per lane; 512 work-groups of
//      which one does three more; 32 x 1024 lanes of which work-group 0 does four more
//   B) one launch: the same first kernel, whose last work-group to arrive (fence, counter) goes on
//      with the seven dependent loads
// Time per step over 300 steps on one stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_busy(unsigned *p, unsigned us)
{
	const unsigned long long t0 = wall_clock64();
	unsigned acc = 0;
	while (wall_clock64() - t0 < (unsigned long long)us * 100ull)
		acc += p[(threadIdx.x + acc) & 1023];
	if (acc == 0xdeadbeef) p[0] = acc;
}
__device__ __forceinline__ unsigned chase(const unsigned *tab, unsigned mask, unsigned x, int hops)
{
	for (int h = 0; h < hops; h++)
		x = mix(x + __hip_atomic_load(&tab[x & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	return x;
}
__global__ __launch_bounds__(256) void k_first(const unsigned *tab, unsigned mask, unsigned *out, unsigned n, unsigned *counter, int fused)
{
	const unsigned i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		out[i] = chase(tab, mask, i * 2654435761u, 3);
	if (!fused)
		return;
	__shared__ unsigned s_last;
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0)
		s_last = atomicAdd(counter, 1u) == gridDim.x - 1u;
	__syncthreads();
	if (!s_last)
		return;
	__threadfence();
	if (threadIdx.x == 0)
		*counter = 0;
	out[n + threadIdx.x] = chase(tab, mask, out[threadIdx.x] + 17u, 7);
}
__global__ __launch_bounds__(256) void k_second(const unsigned *tab, unsigned mask, unsigned *out, unsigned n)
{
	if (blockIdx.x == 0)
		out[n + threadIdx.x] = chase(tab, mask, out[threadIdx.x] + 17u, 3);
}
__global__ __launch_bounds__(1024) void k_third(const unsigned *tab, unsigned mask, unsigned *out, unsigned n)
{
	if (blockIdx.x == 0)
		out[n + 256 + threadIdx.x] = chase(tab, mask, out[n + (threadIdx.x & 255u)] + 5u, 4);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const unsigned words = (getenv("PROBE_WORDS_LOG2") ? 1u << atoi(getenv("PROBE_WORDS_LOG2")) : 1u << 26), n = 52428;
	unsigned *tab, *out, *counter, *busy;
	CK(hipMalloc(&tab, (size_t)words * 4)); CK(hipMemset(tab, 1, (size_t)words * 4));
	CK(hipMalloc(&out, (n + 4096) * 4)); CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
	CK(hipMalloc(&busy, 4096)); CK(hipMemset(busy, 0, 4096));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const int steps = 300;
	for (int rep = 0; rep < 3; rep++) {
		for (int fused = 0; fused < 2; fused++) {
			CK(hipDeviceSynchronize());
			const double t0 = now_us();
			for (int i = 0; i < steps; i++) {
				hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s, busy, 27u);
				hipLaunchKernelGGL(k_first, dim3(512), dim3(256), 0, s, tab, words - 1, out, n, counter, fused);
				if (!fused) {
					hipLaunchKernelGGL(k_second, dim3(512), dim3(256), 0, s, tab, words - 1, out, n);
					hipLaunchKernelGGL(k_third, dim3(32), dim3(1024), 0, s, tab, words - 1, out, n);
				}
			}
			CK(hipDeviceSynchronize());
			printf("%s %6.1f us per step\n", fused ? "B) one launch, last work-group goes on:  " : "A) three launches:                       ",
					(now_us() - t0) / steps);
		}
	}
	return 0;
}
