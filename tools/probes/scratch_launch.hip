// Does a kernel that needs scratch memory cost more to launch?  Two kernels that do the same
// nothing (one forced to keep a dynamically indexed local array), alternated with a small
// scratch-free kernel like the rounds of the insert pipeline.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_plain(int *p, int n) { if (n == 12345) p[threadIdx.x] = n; }
__global__ __launch_bounds__(1024) void k_noscratch(int *p, int n) { if (n == 12345) p[threadIdx.x] = n; }
__global__ __launch_bounds__(1024) void k_scratch(int *p, int n)
{
	volatile int a[64];
	if (n == 12345) {
		for (int i = 0; i < 64; i++) a[i] = p[i];
		p[threadIdx.x] = a[(n + threadIdx.x) & 63];
	}
}
__global__ __launch_bounds__(1024) void k_lds(int *p, int n)
{
	__shared__ int s[12000];
	if (n == 12345) { s[threadIdx.x] = n; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) & 1023]; }
}
int main()
{
	int *p; hipMalloc(&p, 1 << 20);
	hipStream_t s; hipStreamCreate(&s);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int variant = 0; variant < 3; variant++) {
		for (int rep = 0; rep < 3; rep++) {
			hipEventRecord(a, s);
			for (int i = 0; i < 200; i++) {
				hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, s, p, i);
				hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, s, p, i);
				if (variant == 0) hipLaunchKernelGGL(k_noscratch, dim3(32), dim3(1024), 0, s, p, i);
				if (variant == 1) hipLaunchKernelGGL(k_scratch, dim3(32), dim3(1024), 0, s, p, i);
				if (variant == 2) hipLaunchKernelGGL(k_lds, dim3(32), dim3(1024), 0, s, p, i);
			}
			hipEventRecord(b, s); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			printf("variant %d (%s): %.2f us per triple\n", variant, variant == 0 ? "no scratch" : variant == 1 ? "scratch" : "48 KB LDS", ms * 1000 / 200);
		}
	}
	return 0;
}
