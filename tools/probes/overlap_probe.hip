// Would overlapping the insert rounds of batch i with the search of batch i+1 pay?  The
// schedule of such a pipeline with stand-in kernels of the measured durations:
//   stream S:  A (search + mark, 30 us, whole chip) | wait insert(i-1) | F (correction pass, 8 us) | record
//   stream T:  wait F(i) | R1 8 us, R2 5 us, tail 6 us (small grids) | record
// against the same five kernels back to back on one stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_busy(unsigned *p, unsigned us)
{
	const unsigned long long t0 = wall_clock64();
	unsigned acc = 0;
	while (wall_clock64() - t0 < (unsigned long long)us * 100ull)
		acc += p[(threadIdx.x + acc) & 1023];
	if (acc == 0xdeadbeef) p[0] = acc;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	unsigned *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
	hipStream_t S, T; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&T, hipStreamNonBlocking));
	hipEvent_t evF[4], evI[4];
	for (int k = 0; k < 4; k++) { CK(hipEventCreateWithFlags(&evF[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evI[k], hipEventDisableTiming)); }
	const int steps = 400;
	for (int rep = 0; rep < 3; rep++) {
		CK(hipDeviceSynchronize());
		double t0 = now_us();
		for (int i = 0; i < steps; i++) {
			hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, S, d, 30u);
			hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, S, d, 8u);
			hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, S, d, 5u);
			hipLaunchKernelGGL(k_busy, dim3(32), dim3(1024), 0, S, d, 6u);
		}
		CK(hipDeviceSynchronize());
		printf("one stream, A R1 R2 tail back to back:        %6.1f us per step\n", (now_us() - t0) / steps);
		t0 = now_us();
		for (int i = 0; i < steps; i++) {
			const int k = i & 3, kp = (i + 3) & 3;
			hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, S, d, 30u);          // A(i)
			if (i) CK(hipStreamWaitEvent(S, evI[kp], 0));
			hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, S, d, 8u);           // F(i)
			CK(hipEventRecord(evF[k], S));
			CK(hipStreamWaitEvent(T, evF[k], 0));
			hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, T, d, 8u);           // R1
			hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, T, d, 5u);            // R2
			hipLaunchKernelGGL(k_busy, dim3(32), dim3(1024), 0, T, d, 6u);            // tail
			CK(hipEventRecord(evI[k], T));
		}
		CK(hipDeviceSynchronize());
		printf("two streams, rounds of i beside A of i+1 (+ F):  %6.1f us per step\n", (now_us() - t0) / steps);
	}
	return 0;
}
