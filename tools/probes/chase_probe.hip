// chase_probe.hip -- what one DEPENDENT access costs a lone lane on this device: a pointer chase (every load's
// address comes out of the load before it) through a random cycle laid over a working set of a given size, with
// plain loads, agent-scope loads (the tail launch's ld_agent) and returning atomics; then the same chase by 256
// lanes of four waves side by side (the tail's in-group rounds at load factor 0.85 have ~200 chains under way).
// One launch per line; the set is touched by a streaming kernel in between so that nothing of it is left in the
// caches or the TLBs from the launch before.   make -C tools/probes chase_probe && ./chase_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: plain load, 1: agent-scope load, 2: returning atomic (atomicMax with 0: leaves the word as it is)
template <int MODE>
__global__ void k_chase(unsigned long long *buf, uint32_t start_stride, int steps, unsigned long long *out_ticks, uint32_t *sink)
{
	// (two passes over the same elements: the first finds nothing of them in the caches or the TLBs -- what a kernel's
	// first touch of a line costs --, the second finds what the first brought in)
	for (int pass = 0; pass < 2; pass++) {
		uint32_t at = (blockIdx.x * blockDim.x + threadIdx.x) * start_stride;
		const unsigned long long t0 = wall_clock64();
		for (int k = 0; k < steps; k++) {
			unsigned long long v;
			if (MODE == 0)
				v = buf[at];
			else if (MODE == 1)
				v = __hip_atomic_load(buf + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else
				v = atomicMax(buf + at, 0ull);
			at = (uint32_t)v;
		}
		const unsigned long long t1 = wall_clock64();
		if (threadIdx.x == 0 && blockIdx.x == 0)
			out_ticks[pass] = t1 - t0;
		if (at == 0xffffffffu)
			*sink = at;
	}
}

__global__ void k_stream(const uint4 *p, size_t n, uint32_t *sink)
{
	uint32_t acc = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		acc ^= p[i].x;
	if (acc == 0x12345u)
		*sink = acc;
}

int main()
{
	const size_t kMaxBytes = 4ull << 30;
	unsigned long long *buf, *ticks;
	uint32_t *sink;
	uint4 *flush;
	const size_t flush_bytes = 1ull << 30;
	CK(hipMalloc(&buf, kMaxBytes));
	CK(hipMalloc(&flush, flush_bytes));
	CK(hipMemset(flush, 1, flush_bytes));
	CK(hipMalloc(&ticks, 16));
	CK(hipMalloc(&sink, 4));
	std::mt19937_64 rng(7);
	printf("dependent accesses of ONE lane (and of 256 lanes side by side), ns per access; wall_clock64 ticks of 10 ns\n");
	printf("first: the launch's first touch of the elements (caches and TLBs hold nothing of them); again: the same elements once more\n");
	printf("%-28s %10s %10s %10s   %10s %10s\n", "working set (an element per", "plain", "agent", "atomic", "agent", "atomic");
	printf("%-28s %10s %10s %10s   %10s %10s\n", "64-byte line, random cycle)", "1 lane", "1 lane", "1 lane", "256 lanes", "256 lanes");
	for (size_t bytes : {64ull << 10, 2ull << 20, 32ull << 20, 128ull << 20, 1ull << 30, 4ull << 30}) {
		// elements 64 bytes apart; a random cycle over them
		const size_t n = bytes / 64;
		std::vector<uint32_t> perm(n);
		std::iota(perm.begin(), perm.end(), 0u);
		for (size_t i = n - 1; i > 0; i--)
			std::swap(perm[i], perm[rng() % i]);       // (Sattolo: one cycle)
		std::vector<unsigned long long> host(n * 8, 0ull);
		for (size_t i = 0; i < n; i++)
			host[i * 8] = (unsigned long long)perm[i] * 8ull;
		CK(hipMemcpy(buf, host.data(), n * 64, hipMemcpyHostToDevice));
		const int steps = 64;
		double ns[5][2];
		for (int col = 0; col < 5; col++) {
			const int mode = col < 3 ? col : col - 2;          // 0 1 2 | 1 2
			const int lanes = col < 3 ? 1 : 256;
			const uint32_t stride = (uint32_t)((n / 256) ? (n / 256) * 8 : 8);
			double best[2] = {1e30, 1e30};
			for (int rep = 0; rep < 3; rep++) {
				hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, flush, flush_bytes / 16, sink);
				if (mode == 0)
					hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(lanes), 0, 0, buf, stride, steps, ticks, sink);
				else if (mode == 1)
					hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(lanes), 0, 0, buf, stride, steps, ticks, sink);
				else
					hipLaunchKernelGGL(k_chase<2>, dim3(1), dim3(lanes), 0, 0, buf, stride, steps, ticks, sink);
				unsigned long long t[2] = {0, 0};
				CK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
				for (int p = 0; p < 2; p++) {
					const double v = (double)t[p] * 10.0 / steps;
					best[p] = v < best[p] ? v : best[p];
				}
			}
			ns[col][0] = best[0];
			ns[col][1] = best[1];
		}
		char label[64];
		snprintf(label, sizeof label, "%zu %s", bytes >= (1ull << 20) ? bytes >> 20 : bytes >> 10, bytes >= (1ull << 20) ? "MiB" : "KiB");
		printf("%-22s first", label);
		for (int col = 0; col < 5; col++)
			printf(" %10.0f%s", ns[col][0], col == 2 ? "  " : "");
		printf("\n%-22s again", "");
		for (int col = 0; col < 5; col++)
			printf(" %10.0f%s", ns[col][1], col == 2 ? "  " : "");
		printf("\n");
	}
	return 0;
}
