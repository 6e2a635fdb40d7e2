/*
 * scheduler_cycle.c -- a plain-C host (gcc, no HIP headers) that drives the
 * library exactly the way Mega-KV's GPU scheduler does
 * (reference src/mega_scheduler.c:273-280 set-up, :393-504 one cycle):
 *
 *   per worker stream: H2D search requests, memsetAsync(out), gpu_hash_search,
 *   D2H results; H2D delete requests, gpu_hash_delete; H2D the 8 insert blocks
 *   and their counts (device int[8]), gpu_hash_insert; one
 *   cudaDeviceSynchronize per cycle.
 *
 * It is the build's own counterpart of the caller (the reference's .c files
 * are not copied) and doubles as the link test of lib/libgpuhash.a:
 *   gcc -Iinclude examples/scheduler_cycle.c -Llib -lgpuhash -lamdhip64 -lstdc++
 * usage: scheduler_cycle [cycles] [worker streams, default 4] [mapped, default 0] [one call, default 0] [digest, default 0] [per-stream wait, default 0]
 *   per-stream wait = 1: instead of the reference's one cudaDeviceSynchronize per cycle the host waits for
 *   every worker's stream by itself, the LAST worker's first (cudaStreamSynchronize) -- each stream must then
 *   have that worker's results behind it on its own (the library's deferred mode issues a cycle on one stream).
 *   digest = 1: a checksum of the whole table is printed at the end (runs with and without MKV_DEFER=1 --
 *   the library's deferred mode, megakv_hip.h -- must print the same).
 *   mapped = 1: no staging copies; the library gets the pinned host buffers themselves
 *   (mapped into the device's address space), same calls, same checks.
 *   one call = 1: the cycle's library work -- every worker's search, delete and insert -- is ONE
 *   mkv_cycle_workers call on the first worker's stream (megakv_hip.h) instead of three calls per
 *   worker; same buffers, same copies, same checks.
 * Self-checking: every key inserted in cycle c is found in cycle c+1 with its
 * loc (property P1 of libgpuhash/test/insert_test.c:178-195), deleted keys are
 * gone (P2, :237-244).  Exit code 0 = all checks passed.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "libgpuhash.h"
#include "megakv_hip.h"

#define MAX_WORKERS 16		/* assert(config->cpu_worker_num <= 16), src/mega_scheduler.c:276 */
#define MAX_SEARCH 32768	/* config->batch_max_search_job, src/mega.c:136 */
#define MAX_INSERT 4096		/* per block, src/mega.c:143 */
#define MAX_DELETE 4096		/* src/mega.c:144 */

static uint64_t mix64(uint64_t x)
{
	x += 0x9e3779b97f4a7c15ULL;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	return x ^ (x >> 31);
}

/* key -> (sig, hash): low/high word of a 64-bit digest (src/mega_recv.c:361-362) */
static void key_elem(uint32_t key, uint32_t *sig, uint32_t *hash)
{
	uint64_t z = mix64(key);
	*sig = (uint32_t)z ? (uint32_t)z : 1;
	*hash = (uint32_t)(z >> 32);
}

typedef struct worker_s {
	cudaStream_t stream;
	selem_t *search_in, *search_in_d;
	loc_t *search_out, *search_out_d;
	delem_t *delete_in, *delete_in_d;
	ielem_t *insert_in[INSERT_BLOCK];	/* pinned host, one per block */
	ielem_t *insert_in_d;				/* device, INSERT_BLOCK slices */
	ielem_t **insert_ptrarray_d;		/* device array of device pointers */
	int *insert_job_num_d;
	int *insert_job_num_p;				/* mapped mode: the counts in pinned host memory */
	int insert_job_num[INSERT_BLOCK];
	int num_search, num_delete, num_gone;
	uint32_t *search_keys;
} worker_t;

int main(int argc, char **argv)
{
	int cycles = argc > 1 ? atoi(argv[1]) : 20;
	/* Worker streams.  Like the reference, inserts of different workers are not ordered
	 * against each other (SURVEY B-1): with several workers a lost update between two
	 * concurrent inserts into one bucket is possible by design.  1 worker = fully defined. */
	int WORKERS = argc > 2 ? atoi(argv[2]) : 4;
	/* 1: no staging copies -- the pinned host buffers are mapped into the device's address space
	 * (same addresses), so the library is handed the host buffers themselves and its kernels read
	 * the requests and write the results over PCIe.  0: the reference's copies. */
	int MAPPED = argc > 3 ? atoi(argv[3]) : 0;
	int ONECALL = argc > 4 ? atoi(argv[4]) : 0;
	int DIGEST = argc > 5 ? atoi(argv[5]) : 0;
	int STREAMWAIT = argc > 6 ? atoi(argv[6]) : 0;
	mkv_geom_t geom = {MEM_P, IBLOCK_P,
#if defined(HASH_2CHOICE)
		MKV_SCHEME_2CHOICE, 0
#else
		MKV_SCHEME_CUCKOO, MAX_CUCKOO_NUM
#endif
	};
	uint8_t *table;
	worker_t w[MAX_WORKERS];
	if (WORKERS < 1 || WORKERS > MAX_WORKERS)
		WORKERS = 4;
	int i, b, c, id, errors = 0;

	CUDA_SAFE_CALL(cudaMalloc((void **)&table, HT_SIZE));
	CUDA_SAFE_CALL(cudaMemset(table, 0, HT_SIZE));

	for (id = 0; id < WORKERS; id++) {
		worker_t *k = &w[id];
		ielem_t *ptrs[INSERT_BLOCK];
		CUDA_SAFE_CALL(cudaStreamCreate(&k->stream));
		CUDA_SAFE_CALL(cudaHostAlloc((void **)&k->search_in, MAX_SEARCH * sizeof(selem_t), cudaHostAllocDefault));
		CUDA_SAFE_CALL(cudaHostAlloc((void **)&k->search_out, 2 * MAX_SEARCH * sizeof(loc_t), cudaHostAllocDefault));
		CUDA_SAFE_CALL(cudaHostAlloc((void **)&k->delete_in, MAX_DELETE * sizeof(delem_t), cudaHostAllocDefault));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->search_in_d, MAX_SEARCH * sizeof(selem_t)));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->search_out_d, 2 * MAX_SEARCH * sizeof(loc_t)));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->delete_in_d, MAX_DELETE * sizeof(delem_t)));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->insert_in_d, INSERT_BLOCK * MAX_INSERT * sizeof(ielem_t)));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->insert_ptrarray_d, INSERT_BLOCK * sizeof(ielem_t *)));
		CUDA_SAFE_CALL(cudaMalloc((void **)&k->insert_job_num_d, INSERT_BLOCK * sizeof(int)));
		for (b = 0; b < INSERT_BLOCK; b++) {
			CUDA_SAFE_CALL(cudaHostAlloc((void **)&k->insert_in[b], MAX_INSERT * sizeof(ielem_t), cudaHostAllocDefault));
			ptrs[b] = MAPPED ? k->insert_in[b] : k->insert_in_d + (size_t)b * MAX_INSERT;
		}
		CUDA_SAFE_CALL(cudaHostAlloc((void **)&k->insert_job_num_p, INSERT_BLOCK * sizeof(int), cudaHostAllocDefault));
		/* written once at init, like src/mega_recv.c:186-191 */
		CUDA_SAFE_CALL(cudaMemcpy(k->insert_ptrarray_d, ptrs, sizeof(ptrs), cudaMemcpyHostToDevice));
		k->search_keys = (uint32_t *)malloc(MAX_SEARCH * sizeof(uint32_t));
		k->num_search = k->num_delete = k->num_gone = 0;
	}

	/* device memory the library takes for itself (update workspaces): free memory once the host's
	 * own buffers are there, against free memory at the end */
	size_t free_setup = 0, free_end = 0, total_mem = 0;
	CUDA_SAFE_CALL(cudaDeviceSynchronize());
	CUDA_SAFE_CALL(cudaMemGetInfo(&free_setup, &total_mem));

	uint32_t next_key = 1;
	uint64_t total_search = 0, total_insert = 0, total_delete = 0;
	struct timespec t0, t1, c0, c1;
	double sched_us = 0;
	int sched_n = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);

	for (c = 0; c < cycles; c++) {
		/* --- receivers: fill this cycle's batches --- */
		for (id = 0; id < WORKERS; id++) {
			worker_t *k = &w[id];
			/* search what this worker inserted last cycle; delete (then expect a
			 * miss for) what it inserted two cycles ago */
			memset(k->insert_job_num, 0, sizeof(k->insert_job_num));
			for (i = 0; i < 6000; i++) {
				uint32_t key = next_key++, sig, hash;
				key_elem(key, &sig, &hash);
				b = (int)(hash >> (32 - IBLOCK_P));		/* src/mega_recv.c:476-477 */
				if (k->insert_job_num[b] + 1 >= MAX_INSERT)
					continue;
				ielem_t *e = &k->insert_in[b][k->insert_job_num[b]++];
				e->sig = sig;
				e->hash = hash;
				e->loc = key;							/* loc = key: easy to verify */
			}
		}

		/* --- scheduler: one cycle --- */
		clock_gettime(CLOCK_MONOTONIC, &c0);
		if (ONECALL) {
			/* the copies of every worker on one stream, then ONE library call for the whole cycle */
			mkv_worker_t mw[MAX_WORKERS];
			cudaStream_t st = w[0].stream;
			for (id = 0; id < WORKERS; id++) {
				worker_t *k = &w[id];
				if (!MAPPED) {
					if (k->num_search)
						CUDA_SAFE_CALL(cudaMemcpyAsync(k->search_in_d, k->search_in, k->num_search * sizeof(selem_t),
									cudaMemcpyHostToDevice, st));
					if (k->num_delete)
						CUDA_SAFE_CALL(cudaMemcpyAsync(k->delete_in_d, k->delete_in, k->num_delete * sizeof(delem_t),
									cudaMemcpyHostToDevice, st));
					for (b = 0; b < INSERT_BLOCK; b++)
						if (k->insert_job_num[b])
							CUDA_SAFE_CALL(cudaMemcpyAsync(k->insert_in_d + (size_t)b * MAX_INSERT, k->insert_in[b],
										k->insert_job_num[b] * sizeof(ielem_t), cudaMemcpyHostToDevice, st));
					CUDA_SAFE_CALL(cudaMemcpyAsync(k->insert_job_num_d, k->insert_job_num, sizeof(k->insert_job_num),
								cudaMemcpyHostToDevice, st));
				} else {
					for (b = 0; b < INSERT_BLOCK; b++)
						k->insert_job_num_p[b] = k->insert_job_num[b];
				}
				for (b = 0; b < INSERT_BLOCK; b++)
					total_insert += k->insert_job_num[b];
				total_search += k->num_search;
				total_delete += k->num_delete;
				mw[id].search_in = MAPPED ? k->search_in : k->search_in_d;
				mw[id].search_out = MAPPED ? k->search_out : k->search_out_d;
				mw[id].num_search = k->num_search;
				mw[id].delete_in = MAPPED ? k->delete_in : k->delete_in_d;
				mw[id].num_delete = k->num_delete;
				mw[id].blk_input = k->insert_ptrarray_d;
				mw[id].blk_elem_num = MAPPED ? k->insert_job_num_p : k->insert_job_num_d;
				mw[id].num_blks = INSERT_BLOCK;
			}
			if (mkv_cycle_workers(&geom, (bucket_t *)table, mw, WORKERS, st) != MKV_OK) {
				fprintf(stderr, "mkv_cycle_workers failed\n");
				return 2;
			}
			if (!MAPPED)
				for (id = 0; id < WORKERS; id++)
					if (w[id].num_search)
						CUDA_SAFE_CALL(cudaMemcpyAsync(w[id].search_out, w[id].search_out_d,
									2 * w[id].num_search * sizeof(loc_t), cudaMemcpyDeviceToHost, st));
		}
		for (id = 0; id < WORKERS && !ONECALL; id++) {	/* 1. search */
			worker_t *k = &w[id];
			if (k->num_search == 0)
				continue;
			if (MAPPED) {
				memset(k->search_out, 0, 2 * k->num_search * sizeof(loc_t));
				gpu_hash_search(k->search_in, k->search_out, (bucket_t *)table, k->num_search, 24576, 256, k->stream);
				total_search += k->num_search;
				continue;
			}
			CUDA_SAFE_CALL(cudaMemcpyAsync(k->search_in_d, k->search_in, k->num_search * sizeof(selem_t),
						cudaMemcpyHostToDevice, k->stream));
			CUDA_SAFE_CALL(cudaMemsetAsync(k->search_out_d, 0, 2 * k->num_search * sizeof(loc_t), k->stream));
			gpu_hash_search(k->search_in_d, k->search_out_d, (bucket_t *)table, k->num_search,
					24576, 256, k->stream);				/* src/mega.c:163-165 */
			CUDA_SAFE_CALL(cudaMemcpyAsync(k->search_out, k->search_out_d, 2 * k->num_search * sizeof(loc_t),
						cudaMemcpyDeviceToHost, k->stream));
			total_search += k->num_search;
		}
		for (id = 0; id < WORKERS && !ONECALL; id++) {	/* 2. delete */
			worker_t *k = &w[id];
			if (k->num_delete == 0)
				continue;
			if (MAPPED) {
				gpu_hash_delete(k->delete_in, (bucket_t *)table, k->num_delete, 16384, 256, k->stream);
				total_delete += k->num_delete;
				continue;
			}
			CUDA_SAFE_CALL(cudaMemcpyAsync(k->delete_in_d, k->delete_in, k->num_delete * sizeof(delem_t),
						cudaMemcpyHostToDevice, k->stream));
			gpu_hash_delete(k->delete_in_d, (bucket_t *)table, k->num_delete, 16384, 256, k->stream);
			total_delete += k->num_delete;
		}
		for (id = 0; id < WORKERS && !ONECALL; id++) {	/* 3. insert */
			worker_t *k = &w[id];
			if (MAPPED) {
				for (b = 0; b < INSERT_BLOCK; b++) {
					k->insert_job_num_p[b] = k->insert_job_num[b];
					total_insert += k->insert_job_num[b];
				}
				gpu_hash_insert((bucket_t *)table, k->insert_ptrarray_d, k->insert_job_num_p, INSERT_BLOCK, k->stream);
				continue;
			}
			for (b = 0; b < INSERT_BLOCK; b++) {
				if (k->insert_job_num[b] == 0)
					continue;
				CUDA_SAFE_CALL(cudaMemcpyAsync(k->insert_in_d + (size_t)b * MAX_INSERT, k->insert_in[b],
							k->insert_job_num[b] * sizeof(ielem_t), cudaMemcpyHostToDevice, k->stream));
				total_insert += k->insert_job_num[b];
			}
			CUDA_SAFE_CALL(cudaMemcpyAsync(k->insert_job_num_d, k->insert_job_num, sizeof(k->insert_job_num),
						cudaMemcpyHostToDevice, k->stream));
			gpu_hash_insert((bucket_t *)table, k->insert_ptrarray_d, k->insert_job_num_d, INSERT_BLOCK, k->stream);
		}
		if (STREAMWAIT)
			for (id = WORKERS - 1; id >= 0; id--)
				CUDA_SAFE_CALL(cudaStreamSynchronize(w[id].stream));
		else
			CUDA_SAFE_CALL(cudaDeviceSynchronize());
		clock_gettime(CLOCK_MONOTONIC, &c1);
		if (c >= cycles / 4)	/* (the first cycles create the library's workspaces) */
			sched_us += 1e6 * (c1.tv_sec - c0.tv_sec) + (c1.tv_nsec - c0.tv_nsec) / 1e3, sched_n++;

		/* --- senders: consume results (src/mega_send.c:408-414) --- */
		for (id = 0; id < WORKERS; id++) {
			worker_t *k = &w[id];
			for (i = 0; i < k->num_search; i++) {
				loc_t loc = k->search_out[2 * i];
				if (loc == 0)
					loc = k->search_out[2 * i + 1];
				/* the last num_gone requests repeat keys deleted one cycle ago */
				int expect_hit = i < k->num_search - k->num_gone;
				if (expect_hit ? loc != k->search_keys[i] : loc == k->search_keys[i]) {
					if (errors++ < 10)
						fprintf(stderr, "cycle %d worker %d key %u: loc %u (%s expected)\n", c, id,
								k->search_keys[i], loc, expect_hit ? "hit" : "miss");
				}
			}
			/* Next cycle this worker
			 *   searches every key inserted now (hit expected) and, at the end of
			 *   the batch, the keys deleted in THIS cycle (miss expected);
			 *   deletes the first 1000 keys inserted now -- delete runs after
			 *   search on the same stream, so they are still found next cycle. */
			int n = 0, gone = 0;
			for (b = 0; b < INSERT_BLOCK; b++) {
				for (i = 0; i < k->insert_job_num[b] && n < MAX_SEARCH - 1 - MAX_DELETE; i++) {
					ielem_t *e = &k->insert_in[b][i];
					k->search_in[n].sig = e->sig;
					k->search_in[n].hash = e->hash;
					k->search_keys[n] = e->loc;
					n++;
				}
			}
			for (i = 0; i < k->num_delete; i++) {		/* deleted this cycle */
				k->search_in[n].sig = k->delete_in[i].sig;
				k->search_in[n].hash = k->delete_in[i].hash;
				k->search_keys[n] = k->delete_in[i].loc;
				n++;
				gone++;
			}
			int nd = n - gone < 1000 ? n - gone : 1000;
			for (i = 0; i < nd; i++) {
				k->delete_in[i].sig = k->search_in[i].sig;
				k->delete_in[i].hash = k->search_in[i].hash;
				k->delete_in[i].loc = k->search_keys[i];
			}
			k->num_search = n;
			k->num_gone = gone;
			k->num_delete = nd;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	double us = 1e6 * (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) / 1e3;
	CUDA_SAFE_CALL(cudaMemGetInfo(&free_end, &total_mem));
	printf("cycles %d workers %d%s%s: search %llu insert %llu delete %llu in %.0f us -> %.2f Mops end to end, errors %d, library memory %.0f MiB\n",
			cycles, WORKERS, MAPPED ? " (host buffers mapped, no copies)" : "", ONECALL ? " (one library call per cycle)" : "",
			(unsigned long long)total_search, (unsigned long long)total_insert,
			(unsigned long long)total_delete, us, (total_search + total_insert + total_delete) / us, errors,
			free_setup > free_end ? (double)(free_setup - free_end) / 1048576.0 : 0.0);
	/* the scheduler's own share: from its first call of a cycle to the return of the cycle's synchronisation (the rest
	 * is this program's receivers and senders: making the next batches, checking the results) */
	if (sched_n)
		printf("scheduler section: %.1f us per cycle over the last %d cycles -> %.1f Mops\n", sched_us / sched_n, sched_n,
				(double)(total_search + total_insert + total_delete) / cycles / (sched_us / sched_n));
	if (DIGEST) {
		uint64_t *host = (uint64_t *)malloc(HT_SIZE), h = 1469598103934665603ULL;
		size_t q;
		if (!host)
			return 3;
		CUDA_SAFE_CALL(cudaMemcpy(host, table, HT_SIZE, cudaMemcpyDeviceToHost));
		for (q = 0; q < HT_SIZE / 8; q++)
			h = (h ^ host[q]) * 1099511628211ULL;
		printf("table digest %016llx\n", (unsigned long long)h);
		free(host);
	}
	return errors ? 1 : 0;
}
