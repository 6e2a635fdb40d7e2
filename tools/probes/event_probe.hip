// event_probe.hip -- what the cross-stream primitives cost between two kernels of ONE stream on
// this runtime: a busy kernel of ~60 us launched back to back on stream C, with, between two
// launches, (a) nothing, (b) hipEventRecord on C, (c) hipStreamWaitEvent(C, e) on an event of
// another stream that completed long ago, (d) both, (e) a wait on an event recorded on the other
// stream a moment ago behind a short kernel.  Time per iteration (events around 200 iterations).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/event_probe.hip -o tools/probes/event_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_busy(unsigned us, unsigned *sink)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (unsigned long long)us * 100ull)
		__builtin_amdgcn_s_sleep(4);
	if (sink && threadIdx.x == 0 && blockIdx.x == 0)
		*sink = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
	hipStream_t C, S;
	CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
	hipEvent_t t0, t1, old_ev, ev[64], evc[64];
	CK(hipEventCreate(&t0));
	CK(hipEventCreate(&t1));
	CK(hipEventCreateWithFlags(&old_ev, hipEventDisableTiming));
	for (int i = 0; i < 64; i++) {
		CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
		CK(hipEventCreateWithFlags(&evc[i], hipEventDisableTiming));
	}
	hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, S, 5u, (unsigned *)nullptr);
	CK(hipEventRecord(old_ev, S));
	CK(hipDeviceSynchronize());
	const int N = 200;
	const unsigned busy = 60, grid = 2048;
	for (int mode = 0; mode < 7; mode++) {
		for (int rep = 0; rep < 2; rep++) {
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(t0, C));
			for (int i = 0; i < N; i++) {
				if (mode == 1 || mode == 3)
					CK(hipEventRecord(evc[i & 63], C));
				if (mode == 2 || mode == 3)
					CK(hipStreamWaitEvent(C, old_ev, 0));
				if (mode == 4) {
					// S: short kernel behind C's previous kernel's event, then C waits for it (a real hand-over both ways)
					CK(hipEventRecord(evc[i & 63], C));
					CK(hipStreamWaitEvent(S, evc[i & 63], 0));
					hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, S, 2u, (unsigned *)nullptr);
					CK(hipEventRecord(ev[i & 63], S));
					CK(hipStreamWaitEvent(C, ev[i & 63], 0));
				}
				if (mode == 5) {
					// S runs a kernel of its own per iteration, independent; C waits for the one of TWO iterations ago
					hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, S, 20u, (unsigned *)nullptr);
					CK(hipEventRecord(ev[i & 63], S));
					if (i >= 2)
						CK(hipStreamWaitEvent(C, ev[(i - 2) & 63], 0));
				}
				if (mode == 6) {
					// like 5, and S waits for C's kernel of the previous iteration before its own (both directions, slack of one)
					CK(hipEventRecord(evc[i & 63], C));
					if (i >= 1)
						CK(hipStreamWaitEvent(S, evc[(i - 1) & 63], 0));
					hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, S, 20u, (unsigned *)nullptr);
					CK(hipEventRecord(ev[i & 63], S));
					if (i >= 2)
						CK(hipStreamWaitEvent(C, ev[(i - 2) & 63], 0));
				}
				hipLaunchKernelGGL(k_busy, dim3(grid), dim3(256), 0, C, busy, (unsigned *)nullptr);
			}
			CK(hipEventRecord(t1, C));
			CK(hipDeviceSynchronize());
			float ms = 0;
			CK(hipEventElapsedTime(&ms, t0, t1));
			if (rep == 1) {
				const char *what[] = {"nothing between the launches", "hipEventRecord on C", "wait on an event long complete",
					"record + wait (long complete)", "hand-over to S and back (2 us kernel there)",
					"S busy 20 us per iteration, C waits for S's kernel of two iterations ago",
					"the same, S also waiting for C's previous kernel"};
				printf("mode %d %-78s %.1f us per iteration (kernel %u us)\n", mode, what[mode], ms * 1e3 / N, busy);
			}
		}
	}
	return 0;
}
