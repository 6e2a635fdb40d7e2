/*
 * graph_cycle.c -- one scheduler cycle (mkv_cycle: search -> insert of one worker,
 * reference src/mega_scheduler.c:393-502) captured into a hipGraph THROUGH THE C API and
 * replayed, against the same cycle issued eagerly.  Plain C (gcc), the HIP runtime's C
 * header for the stream / graph / event calls, the library through include/megakv_hip.h:
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/graph_cycle.c \
 *       -Llib -lgpuhash -L/opt/rocm/lib -lamdhip64 -lstdc++ -o graph_cycle
 *   usage: graph_cycle [mem_p, default 30] [batch log2, default 20] [set per mille, default 50] [replays, default 100]
 *
 * The cycle is capturable because nothing in it needs the host: request counts are read on
 * the device, round stamps live on the device, the update workspaces are reserved before the
 * capture (mkv_reserve).  Every replay runs the SAME batch (its SETs are then updates with
 * the same value), so eager and replay do identical work; the table is compared at the end
 * (a digest over its words) between an eager run and a replayed run from the same start.
 * Prints microseconds per cycle for both (HIP events around the whole run).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "megakv_hip.h"

#define CHECK(expr)                                                                         \
	do {                                                                                    \
		hipError_t e_ = (expr);                                                             \
		if (e_ != hipSuccess) {                                                             \
			fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
			exit(2);                                                                        \
		}                                                                                   \
	} while (0)

static uint64_t mix64(uint64_t x)
{
	x += 0x9e3779b97f4a7c15ULL;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	return x ^ (x >> 31);
}

static uint64_t digest(const uint32_t *w, size_t n)
{
	uint64_t h = 1469598103934665603ULL;
	size_t i;
	for (i = 0; i < n; i++)
		if (w[i])
			h = (h ^ (w[i] + (uint64_t)i * 0x9e3779b97f4a7c15ULL)) * 1099511628211ULL;
	return h;
}

int main(int argc, char **argv)
{
	const int mem_p = argc > 1 ? atoi(argv[1]) : 30;
	const int blog = argc > 2 ? atoi(argv[2]) : 20;
	const int set_pm = argc > 3 ? atoi(argv[3]) : 50;
	const int replays = argc > 4 ? atoi(argv[4]) : 100;
	const mkv_geom_t geom = {(uint32_t)mem_p, 3, MKV_SCHEME_CUCKOO, 5};
	const size_t ht_bytes = (size_t)1 << mem_p, ht_words = ht_bytes / 4;
	const int64_t n = (int64_t)1 << blog, n_set = n * set_pm / 1000, n_get = n - n_set;
	const uint32_t bucket_bits = (uint32_t)mem_p - 6, region_shift = bucket_bits - 3;
	const uint32_t hash_mask = (uint32_t)(((uint64_t)1 << bucket_bits) - 1);
	int64_t i;
	int b, pass;

	/* requests: GETs of keys 1 .. n_get (pre-loaded below), SETs of fresh keys, binned by region */
	selem_t *gets = (selem_t *)malloc((size_t)n_get * sizeof(selem_t));
	ielem_t *sets = (ielem_t *)malloc((size_t)(n_set ? n_set : 1) * sizeof(ielem_t));
	ielem_t *pre = (ielem_t *)malloc((size_t)n_get * sizeof(ielem_t));
	int cnt[8] = {0}, pcnt[8] = {0}, off[9], poff[9], fill[8];
	for (i = 0; i < n_get; i++) {
		uint64_t z = mix64((uint64_t)i + 1);
		gets[i].sig = (uint32_t)z ? (uint32_t)z : 1;
		gets[i].hash = (uint32_t)(z >> 32);
		pcnt[(gets[i].hash & hash_mask) >> region_shift]++;
	}
	for (i = 0; i < n_set; i++) {
		uint64_t z = mix64((uint64_t)n_get + 1 + (uint64_t)i);
		cnt[(((uint32_t)(z >> 32)) & hash_mask) >> region_shift]++;
	}
	off[0] = poff[0] = 0;
	for (b = 0; b < 8; b++) {
		off[b + 1] = off[b] + cnt[b];
		poff[b + 1] = poff[b] + pcnt[b];
	}
	memset(fill, 0, sizeof(fill));
	for (i = 0; i < n_get; i++) {
		b = (int)((gets[i].hash & hash_mask) >> region_shift);
		ielem_t *e = &pre[poff[b] + fill[b]++];
		e->sig = gets[i].sig;
		e->hash = gets[i].hash;
		e->loc = (uint32_t)i + 1;
	}
	memset(fill, 0, sizeof(fill));
	for (i = 0; i < n_set; i++) {
		uint64_t z = mix64((uint64_t)n_get + 1 + (uint64_t)i);
		uint32_t sig = (uint32_t)z ? (uint32_t)z : 1, hash = (uint32_t)(z >> 32);
		b = (int)((hash & hash_mask) >> region_shift);
		ielem_t *e = &sets[off[b] + fill[b]++];
		e->sig = sig;
		e->hash = hash;
		e->loc = (uint32_t)(n_get + 1 + i);
	}

	uint32_t *table;
	selem_t *d_gets;
	ielem_t *d_sets, *d_pre;
	loc_t *d_out;
	ielem_t **d_ptrs, *ptrs[8], *pptrs[8];
	int *d_cnt;
	hipStream_t s;
	hipEvent_t e0, e1;
	CHECK(hipMalloc((void **)&table, ht_bytes));
	CHECK(hipMalloc((void **)&d_gets, (size_t)n_get * sizeof(selem_t)));
	CHECK(hipMalloc((void **)&d_sets, (size_t)(n_set ? n_set : 1) * sizeof(ielem_t)));
	CHECK(hipMalloc((void **)&d_pre, (size_t)n_get * sizeof(ielem_t)));
	CHECK(hipMalloc((void **)&d_out, (size_t)n_get * 2 * sizeof(loc_t)));
	CHECK(hipMalloc((void **)&d_ptrs, 8 * sizeof(ielem_t *)));
	CHECK(hipMalloc((void **)&d_cnt, 8 * sizeof(int)));
	CHECK(hipMemcpy(d_gets, gets, (size_t)n_get * sizeof(selem_t), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_sets, sets, (size_t)n_set * sizeof(ielem_t), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_pre, pre, (size_t)n_get * sizeof(ielem_t), hipMemcpyHostToDevice));
	CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	if (mkv_reserve(&geom, n_get + 1024, s) != MKV_OK) {     /* before any capture: allocation is not capturable */
		fprintf(stderr, "mkv_reserve failed\n");
		return 2;
	}
	uint32_t *host_table = (uint32_t *)malloc(ht_bytes);
	uint64_t dig[2] = {0, 0};
	float us[2] = {0, 0};
	uint32_t *out_h = (uint32_t *)malloc((size_t)n_get * 8);
	int errors = 0;

	for (pass = 0; pass < 2; pass++) {          /* 0: eager, 1: graph replay -- from the same start */
		CHECK(hipMemsetAsync(table, 0, ht_bytes, s));
		for (b = 0; b < 8; b++) {
			pptrs[b] = d_pre + poff[b];
			ptrs[b] = d_sets + off[b];
		}
		CHECK(hipMemcpyAsync(d_ptrs, pptrs, sizeof(pptrs), hipMemcpyHostToDevice, s));
		CHECK(hipMemcpyAsync(d_cnt, pcnt, sizeof(pcnt), hipMemcpyHostToDevice, s));
		if (mkv_insert_hinted(&geom, (bucket_t *)table, d_ptrs, d_cnt, 8, n_get, s) != MKV_OK)
			return 2;
		CHECK(hipStreamSynchronize(s));
		CHECK(hipMemcpyAsync(d_ptrs, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, s));
		CHECK(hipMemcpyAsync(d_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, s));
		CHECK(hipStreamSynchronize(s));
		hipGraph_t graph = NULL;
		hipGraphExec_t exec = NULL;
		if (pass == 1) {
			CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
			if (mkv_cycle(&geom, (bucket_t *)table, d_gets, d_out, n_get, NULL, 0, n_set ? d_ptrs : NULL,
						n_set ? d_cnt : NULL, n_set ? 8 : 0, n_set, 0u, s, NULL, 0) != MKV_OK) {
				fprintf(stderr, "mkv_cycle failed under capture\n");
				return 2;
			}
			CHECK(hipStreamEndCapture(s, &graph));
			CHECK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
		}
		int r, warm = 10;
		for (r = 0; r < warm + replays; r++) {
			if (r == warm)
				CHECK(hipEventRecord(e0, s));
			if (pass == 1) {
				CHECK(hipGraphLaunch(exec, s));
			} else if (mkv_cycle(&geom, (bucket_t *)table, d_gets, d_out, n_get, NULL, 0, n_set ? d_ptrs : NULL,
						n_set ? d_cnt : NULL, n_set ? 8 : 0, n_set, 0u, s, NULL, 0) != MKV_OK) {
				return 2;
			}
		}
		CHECK(hipEventRecord(e1, s));
		CHECK(hipStreamSynchronize(s));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		us[pass] = ms * 1e3f / (float)replays;
		CHECK(hipMemcpy(host_table, table, ht_bytes, hipMemcpyDeviceToHost));
		dig[pass] = digest(host_table, ht_words);
		CHECK(hipMemcpy(out_h, d_out, (size_t)n_get * 8, hipMemcpyDeviceToHost));
		for (i = 0; i < n_get; i++) {
			uint32_t loc = out_h[2 * i] ? out_h[2 * i] : out_h[2 * i + 1];
			if (loc != (uint32_t)i + 1)
				errors++;
		}
		if (exec)
			CHECK(hipGraphExecDestroy(exec));
		if (graph)
			CHECK(hipGraphDestroy(graph));
	}
	printf("mem_p %d batch 2^%d (%lld GET + %lld SET): eager %.1f us per cycle, hipGraph replay %.1f us per cycle, "
			"tables %s, wrong results %d\n", mem_p, blog, (long long)n_get, (long long)n_set, us[0], us[1],
			dig[0] == dig[1] ? "identical" : "DIFFER", errors);
	return (dig[0] == dig[1] && errors == 0) ? 0 : 1;
}
