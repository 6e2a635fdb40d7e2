// What moves 8 MiB between pinned host memory and HBM fastest, and what overlaps with what?
//   a) hipMemcpyAsync H2D / D2H alone, and both at once on two streams
//   b) the same while a long kernel runs on a third stream
//   c) copy KERNELS (16 bytes per lane, grid-stride) reading / writing the pinned host memory
//      directly (it is mapped into the device's address space), alone / both / beside a kernel
// hipcc --offload-arch=gfx950 -O2 tools/probes/pcie_probe.hip -o tools/probes/pcie_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = src[i];
}

__global__ void k_copy_nt(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src) + i);
		__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst) + i);
	}
}

__global__ void k_busy(unsigned *p, unsigned us)
{
	const unsigned long long t0 = wall_clock64();
	unsigned acc = 0;
	while (wall_clock64() - t0 < (unsigned long long)us * 100ull)
		acc += p[(threadIdx.x + acc) & 1023];
	if (acc == 0xdeadbeef) p[0] = acc;
}

static double now_us()
{
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
	const size_t bytes = 8u << 20, n16 = bytes / 16;
	void *h_in, *h_out, *d_in, *d_out;
	unsigned *d_busy;
	CK(hipHostMalloc(&h_in, bytes, hipHostMallocDefault));
	CK(hipHostMalloc(&h_out, bytes, hipHostMallocDefault));
	CK(hipMalloc(&d_in, bytes));
	CK(hipMalloc(&d_out, bytes));
	CK(hipMalloc(&d_busy, 4096));
	CK(hipMemset(d_busy, 0, 4096));
	hipStream_t s1, s2, s3;
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
	const int reps = 20;
	auto timeit = [&](const char *what, auto fn) {
		fn(); CK(hipDeviceSynchronize());
		const double t0 = now_us();
		for (int r = 0; r < reps; r++) fn();
		CK(hipDeviceSynchronize());
		const double dt = (now_us() - t0) / reps;
		printf("%-58s %8.1f us  (%5.1f GB/s per 8 MiB)\n", what, dt, bytes / dt / 1e3);
	};
	timeit("a) memcpyAsync H2D", [&] { CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
	timeit("a) memcpyAsync D2H", [&] { CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
	timeit("a) memcpyAsync H2D + D2H on two streams", [&] {
		CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1));
		CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2));
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
	timeit("b) busy kernel 150 us alone (256 x 256 lanes)", [&] { hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s3, d_busy, 150u); CK(hipStreamSynchronize(s3)); });
	timeit("b) busy kernel 150 us + memcpyAsync H2D + D2H", [&] {
		hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s3, d_busy, 150u);
		CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1));
		CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2));
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s3)); });
	for (int blocks : {16, 64, 256, 1024}) {
		char name[128];
		snprintf(name, sizeof(name), "c) copy kernel H2D, %d x 256 lanes", blocks);
		timeit(name, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_in, n16); CK(hipStreamSynchronize(s1)); });
		snprintf(name, sizeof(name), "c) copy kernel D2H, %d x 256 lanes", blocks);
		timeit(name, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16); CK(hipStreamSynchronize(s2)); });
	}
	timeit("c) copy kernel nt H2D, 256 x 256", [&] { hipLaunchKernelGGL(k_copy_nt, dim3(256), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_in, n16); CK(hipStreamSynchronize(s1)); });
	timeit("c) copy kernel nt D2H, 256 x 256", [&] { hipLaunchKernelGGL(k_copy_nt, dim3(256), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16); CK(hipStreamSynchronize(s2)); });
	timeit("c) copy kernels H2D + D2H on two streams (256 blocks each)", [&] {
		hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_in, n16);
		hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16);
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
	timeit("c) busy kernel 150 us + copy kernels H2D + D2H", [&] {
		hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s3, d_busy, 150u);
		hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_in, n16);
		hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16);
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s3)); });
	timeit("d) memcpyAsync H2D + copy kernel D2H", [&] {
		CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1));
		hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16);
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
	// e) the three-stage pipeline itself: copy-in | 55 us kernel | copy-out on three streams, three
	//    buffer sets, events between the stages
	{
		const int depth = 3, steps = 48;
		hipEvent_t ev_in[3], ev_comp[3], ev_out[3];
		for (int k = 0; k < depth; k++) {
			CK(hipEventCreateWithFlags(&ev_in[k], hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&ev_comp[k], hipEventDisableTiming));
			CK(hipEventCreateWithFlags(&ev_out[k], hipEventDisableTiming));
		}
		for (int variant = 0; variant < 4; variant++) {
			CK(hipDeviceSynchronize());
			const double t0 = now_us();
			for (int i = 0; i < steps; i++) {
				const int k = i % depth;
				if (i >= depth) CK(hipStreamWaitEvent(s1, ev_out[k], 0));
				if (variant == 0 || variant == 2) CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1));
				else hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s1, (const uint4 *)h_in, (uint4 *)d_in, n16);
				CK(hipEventRecord(ev_in[k], s1));
				CK(hipStreamWaitEvent(s3, ev_in[k], 0));
				hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s3, d_busy, 55u);
				CK(hipEventRecord(ev_comp[k], s3));
				CK(hipStreamWaitEvent(s2, ev_comp[k], 0));
				if (variant == 0 || variant == 3) CK(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, s2));
				else hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_out, n16);
				CK(hipEventRecord(ev_out[k], s2));
			}
			CK(hipDeviceSynchronize());
			const char *names[4] = {"memcpyAsync both ways", "copy kernels both ways", "memcpyAsync in, copy kernel out", "copy kernel in, memcpyAsync out"};
			printf("e) pipeline, %-34s %8.1f us per step\n", names[variant], (now_us() - t0) / steps);
		}
	}
	// f) the same stages, but the HOST separates the steps (one synchronisation per step, like the
	//    reference's scheduler): no copy ever waits on the device for another stream
	for (int variant = 0; variant < 2; variant++) {
		const int steps = 48;
		CK(hipDeviceSynchronize());
		const double t0 = now_us();
		for (int i = 0; i < steps + 2; i++) {
			if (i < steps) CK(hipMemcpyAsync(d_in, h_in, bytes, hipMemcpyHostToDevice, s1));
			if (i >= 1 && i <= steps) hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s3, d_busy, 55u);
			if (i >= 2) CK(hipMemcpyAsync(h_out, d_out, variant ? bytes / 2 : bytes, hipMemcpyDeviceToHost, s2));
			CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s3)); CK(hipStreamSynchronize(s2));
		}
		printf("f) host-separated steps, %s out %25.1f us per step\n", variant ? "4 MiB" : "8 MiB", (now_us() - t0) / steps);
	}
	return 0;
}
