// mkv_pipe.hip -- the scheduler's batch pipeline when the boundary hands over HOST buffers.
//
// The reference's worker cycle stages a batch in pinned host memory, copies the requests to
// the device, runs search / delete / insert, and copies the results back, all on the
// worker's one stream (src/mega_scheduler.c:393-504; buffers src/mega_recv.c:120-213,
// triple-buffered between receiver, scheduler and sender: src/include/mega_batch.h:74-97).
// On one stream the three stages add up (here: 160 us in + 55 us kernels + 150 us out per
// 2^20-request batch).  This file is that cycle as a three-stage pipeline:
//
//   copy-in stream    requests of batch i+1   host -> device     (hipMemcpyAsync, DMA engine)
//   compute stream    mkv_cycle of batch i    (search, insert)
//   copy-out stream   results of batch i-1    device -> host
//
// with `depth` (3) sets of device buffers; the host separates the steps with one
// synchronisation each (mkv_pipe_run).  Measured on MI355X (tools/probes/pcie_probe.hip): a
// pinned 8 MiB copy takes 157 us either way, both directions at once 183 us, and neither is
// slowed by kernels running beside them -- PCIe is full duplex and hipMemcpyAsync from
// hipHostMalloc memory runs on the DMA engines.  (Copy KERNELS reading and writing the mapped
// host memory reach the same rate one way but only 260 us with both directions at once, and
// they occupy CUs.)  So a step costs what the slower direction costs plus the
// synchronisation, ~190 us, instead of the sum.
//
// Results are those of the cycles run one after the other: the compute stream runs them in
// batch order.  Optionally the results leave the device resolved (loc = out[2i] ?:
// out[2i+1], 4 bytes per GET instead of 8: what the reference's sender computes from the
// pair, src/mega_send.c:408-414), which halves the copy-out.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "megakv_hip.h"

namespace {

constexpr int kMaxDepth = 8;

struct Slot {
	selem_t *q = nullptr;
	loc_t *out = nullptr;          // pairs
	loc_t *loc = nullptr;          // resolved (optional mode)
	uint64_t *counters = nullptr;  // mkv_resolve_results scratch
	// The insert side of a batch travels as ONE copy (small copies cost 10 - 20 us each here): a
	// staging area [nblk block pointers | nblk counts | pad | requests], pinned on the host and
	// mirrored on the device; the pointers are the device addresses of the blocks in the mirror.
	char *stage_dev = nullptr, *stage_host = nullptr;
	int64_t nset = 0;              // requests staged
	hipEvent_t ev_in = nullptr, ev_comp = nullptr, ev_out = nullptr;   // the HOST waits on these
};

}  // namespace

struct mkv_pipe {
	mkv_geom_t geom;
	bucket_t *table = nullptr;
	int64_t max_get = 0, max_set = 0;
	int depth = 3, nblk = 8;
	Slot slot[kMaxDepth];
	hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
	size_t hdr_bytes = 0;          // bytes of [pointers | counts | pad] in a staging area
	char err[200] = {0};
};

static int pfail(mkv_pipe *p, int code, const char *what, hipError_t e)
{
	snprintf(p->err, sizeof(p->err), "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
	return code;
}

#define PIPE_HIP(p, expr)                                             \
	do {                                                              \
		hipError_t e_ = (expr);                                       \
		if (e_ != hipSuccess)                                         \
			return pfail(p, MKV_EHIP, #expr, e_);                     \
	} while (0)

extern "C" const char *mkv_pipe_error(const mkv_pipe *p)
{
	return p ? p->err : "no pipeline";
}

extern "C" void mkv_pipe_destroy(mkv_pipe *p)
{
	if (!p)
		return;
	(void)hipDeviceSynchronize();
	for (int k = 0; k < kMaxDepth; k++) {
		Slot &s = p->slot[k];
		if (s.q) (void)hipFree(s.q);
		if (s.out) (void)hipFree(s.out);
		if (s.loc) (void)hipFree(s.loc);
		if (s.counters) (void)hipFree(s.counters);
		if (s.stage_dev) (void)hipFree(s.stage_dev);
		if (s.stage_host) (void)hipHostFree(s.stage_host);
		if (s.ev_in) (void)hipEventDestroy(s.ev_in);
		if (s.ev_comp) (void)hipEventDestroy(s.ev_comp);
		if (s.ev_out) (void)hipEventDestroy(s.ev_out);
	}
	if (p->s_in) (void)hipStreamDestroy(p->s_in);
	if (p->s_comp) {
		mkv_forget_stream(p->s_comp);
		(void)hipStreamDestroy(p->s_comp);
	}
	if (p->s_out) (void)hipStreamDestroy(p->s_out);
	delete p;
}

// max_get / max_set: the largest batch; num_blks: insert blocks per batch (2^IBLOCK_P for the
// reference's hosts); depth: sets of device buffers (3 = one per stage)
extern "C" int mkv_pipe_create(mkv_pipe **out, const mkv_geom_t *geom, bucket_t *table, int64_t max_get,
		int64_t max_set, int num_blks, int depth)
{
	if (!out || !geom || !table || max_get < 0 || max_set < 0 || num_blks < 1 || num_blks > 1024 || depth < 2 ||
			depth > kMaxDepth)
		return MKV_EINVAL;
	mkv_pipe *p = new (std::nothrow) mkv_pipe();
	if (!p)
		return MKV_ENOMEM;
	*out = p;
	p->geom = *geom;
	p->table = table;
	p->max_get = max_get;
	p->max_set = max_set;
	p->nblk = num_blks;
	p->depth = depth;
	// The runtime multiplexes streams over a handful of hardware queues (four by default), least
	// loaded first -- how many streams the host has made before decides whether two of these three
	// end up on one queue, and a copy-in that shares a queue with the cycles' kernels runs in turn
	// with them instead of beside them (258 instead of 190 us per 2^20-request step, measured when
	// the benchmark's stream count changed; GPU_MAX_HW_QUEUES=6 restored it).  The copy streams are
	// therefore made with the highest priority: queues are pooled per priority, so they never share
	// one with the compute stream (nor with each other: made one after the other in their pool).
	int prio_least = 0, prio_greatest = 0;
	PIPE_HIP(p, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
	PIPE_HIP(p, hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, prio_greatest));
	PIPE_HIP(p, hipStreamCreateWithFlags(&p->s_comp, hipStreamNonBlocking));
	PIPE_HIP(p, hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, prio_greatest));
	p->hdr_bytes = ((size_t)num_blks * (sizeof(ielem_t *) + sizeof(int)) + 63) & ~(size_t)63;
	for (int k = 0; k < depth; k++) {
		Slot &s = p->slot[k];
		PIPE_HIP(p, hipMalloc(&s.q, (size_t)(max_get ? max_get : 1) * sizeof(selem_t)));
		PIPE_HIP(p, hipMalloc(&s.out, (size_t)(max_get ? max_get : 1) * 2 * sizeof(loc_t)));
		PIPE_HIP(p, hipMalloc(&s.loc, (size_t)(max_get ? max_get : 1) * sizeof(loc_t)));
		PIPE_HIP(p, hipMalloc(&s.counters, 68 * sizeof(uint64_t)));
		const size_t stage = p->hdr_bytes + (size_t)(max_set ? max_set : 1) * sizeof(ielem_t);
		PIPE_HIP(p, hipMalloc(reinterpret_cast<void **>(&s.stage_dev), stage));
		PIPE_HIP(p, hipHostMalloc(reinterpret_cast<void **>(&s.stage_host), stage, hipHostMallocDefault));
		PIPE_HIP(p, hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
		PIPE_HIP(p, hipEventCreateWithFlags(&s.ev_comp, hipEventDisableTiming));
		PIPE_HIP(p, hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
	}
	// the device's update workspaces now, not inside the first cycle
	if (max_set) {
		const int rc = mkv_reserve(geom, max_set + 8, p->s_comp);
		if (rc != MKV_OK)
			return pfail(p, rc, "mkv_reserve", hipSuccess);
	}
	PIPE_HIP(p, hipDeviceSynchronize());
	return MKV_OK;
}

// nbatches batches from PINNED host memory (hipHostMalloc / cudaHostAlloc; pageable memory
// works but serialises the copies): gets[i] = n_get[i] x selem_t, sets[i] = the insert
// requests of batch i block after block with blk_counts[i][0 .. num_blks) of them per block,
// outs[i] receives n_get[i] x 2 loc_t (resolve = 0: the ABI's pairs) or n_get[i] x loc_t
// (resolve = 1: loc = pair[0] ? pair[0] : pair[1]).
//
// The HOST joins the stages: it waits for an event (copy-in of batch t done, cycle of batch
// t-1 done, ...) and only then issues the stage that needs it, so no stream ever waits on the
// device for another -- as the reference's scheduler, which synchronises with the device once
// per cycle (src/mega_scheduler.c:504).  Measured (tools/probes/pcie_probe.hip): chaining the
// stages with hipStreamWaitEvent instead makes the dependent copies slow and erratic (240 -
// 390 us per step against 192 us with the host in between; 384 us on one stream).  The copy-in
// stream is kept `depth - 1` batches ahead, so the bottleneck direction never idles.  The call
// BLOCKS until the last result is in host memory; it starts after what is queued on `stream`.
// fills the slot's pinned staging area with batch i's insert side (host work, done while the
// device is busy with the step before)
static int stage_inserts(mkv_pipe *p, int i, const ielem_t *const *sets, const int *const *blk_counts)
{
	Slot &s = p->slot[i % p->depth];
	ielem_t **hp = reinterpret_cast<ielem_t **>(s.stage_host);
	int *hc = reinterpret_cast<int *>(hp + p->nblk);
	ielem_t *dev_sets = reinterpret_cast<ielem_t *>(s.stage_dev + p->hdr_bytes);
	int64_t nset = 0;
	for (int b = 0; b < p->nblk; b++) {
		if (blk_counts[i][b] < 0)
			return pfail(p, MKV_EINVAL, "negative block count", hipSuccess);
		hp[b] = dev_sets + nset;
		hc[b] = blk_counts[i][b];
		nset += blk_counts[i][b];
	}
	if (nset > p->max_set)
		return pfail(p, MKV_ETOOBIG, "batch larger than the pipeline was created for", hipSuccess);
	if (nset)
		memcpy(s.stage_host + p->hdr_bytes, sets[i], (size_t)nset * sizeof(ielem_t));
	s.nset = nset;
	return MKV_OK;
}

static int pipe_run(mkv_pipe *p, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int *const *blk_counts, loc_t *const *outs, int resolve, void *stream);

// Every way out of a failed run drains the device first: copies on the caller's host buffers may
// still be in flight on the pipeline's streams, and the caller is about to free those buffers.
extern "C" int mkv_pipe_run(mkv_pipe *p, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int *const *blk_counts, loc_t *const *outs, int resolve, void *stream)
{
	const int rc = pipe_run(p, nbatches, gets, n_get, sets, blk_counts, outs, resolve, stream);
	if (rc != MKV_OK && p)
		(void)hipDeviceSynchronize();
	return rc;
}

static int pipe_run(mkv_pipe *p, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int *const *blk_counts, loc_t *const *outs, int resolve, void *stream)
{
	if (!p || nbatches < 0 || (nbatches && (!gets || !n_get || !sets || !blk_counts || !outs)))
		return MKV_EINVAL;
	if (nbatches == 0)
		return MKV_OK;
	for (int i = 0; i < nbatches; i++)
		if (n_get[i] < 0 || n_get[i] > p->max_get)
			return pfail(p, MKV_ETOOBIG, "batch larger than the pipeline was created for", hipSuccess);
	PIPE_HIP(p, hipStreamSynchronize((hipStream_t)stream));
	const int D = p->depth;
	int rc = MKV_OK;
	// copy-in of batch i into its slot: the requests straight from the caller's buffer, the insert
	// side through the slot's staging area (filled here, on the host)
	auto copy_in = [&](int i) -> int {
		Slot &s = p->slot[i % D];
		const int r = stage_inserts(p, i, sets, blk_counts);
		if (r != MKV_OK)
			return r;
		if (n_get[i])
			PIPE_HIP(p, hipMemcpyAsync(s.q, gets[i], (size_t)n_get[i] * sizeof(selem_t), hipMemcpyHostToDevice, p->s_in));
		if (s.nset)
			PIPE_HIP(p, hipMemcpyAsync(s.stage_dev, s.stage_host, p->hdr_bytes + (size_t)s.nset * sizeof(ielem_t),
					hipMemcpyHostToDevice, p->s_in));
		PIPE_HIP(p, hipEventRecord(s.ev_in, p->s_in));
		return MKV_OK;
	};
	// The copy-in stream runs ahead by itself: D - 1 batches are queued on it before the first
	// cycle, and every cycle that finishes frees a slot for the next one -- the bottleneck
	// direction never waits for the host.  The HOST waits for events (hipEventSynchronize), the
	// streams never wait for each other.
	for (int i = 0; i < D - 1 && i < nbatches; i++)
		if ((rc = copy_in(i)) != MKV_OK)
			return rc;
	for (int t = 0; t < nbatches; t++) {
		Slot &s = p->slot[t % D];
		// the cycle of batch t-1 is done: its results can leave, and the slot of batch t-1-(D-2) ...
		if (t >= 1) {
			Slot &sp = p->slot[(t - 1) % D];
			PIPE_HIP(p, hipEventSynchronize(sp.ev_comp));
			if (n_get[t - 1])
				PIPE_HIP(p, hipMemcpyAsync(outs[t - 1], resolve ? (const void *)sp.loc : (const void *)sp.out,
						(size_t)n_get[t - 1] * (resolve ? 1 : 2) * sizeof(loc_t), hipMemcpyDeviceToHost, p->s_out));
			PIPE_HIP(p, hipEventRecord(sp.ev_out, p->s_out));
		}
		// ... the request buffers of the slot that batch t+D-1 will use (last used by the cycle of
		// batch t-1, just waited for) take the next copy-in
		if (t + D - 1 < nbatches)
			if ((rc = copy_in(t + D - 1)) != MKV_OK)
				return rc;
		// the cycle of batch t: its requests have arrived, and the result buffers of its slot have
		// been copied out (batch t-D)
		PIPE_HIP(p, hipEventSynchronize(s.ev_in));
		if (t >= D)
			PIPE_HIP(p, hipEventSynchronize(s.ev_out));
		{
			ielem_t **ptrs = reinterpret_cast<ielem_t **>(s.stage_dev);
			int *cnts = reinterpret_cast<int *>(ptrs + p->nblk);
			const int64_t nset = s.nset;
			rc = mkv_cycle(&p->geom, p->table, s.q, s.out, n_get[t], nullptr, 0, nset ? ptrs : nullptr,
					nset ? cnts : nullptr, nset ? p->nblk : 0, nset, 0u, p->s_comp, nullptr, 0);
			if (rc == MKV_OK && resolve && n_get[t])
				rc = mkv_resolve_results(s.out, n_get[t], s.loc, s.counters, p->s_comp);
			if (rc != MKV_OK)
				return pfail(p, rc, "mkv_cycle", hipSuccess);
			PIPE_HIP(p, hipEventRecord(s.ev_comp, p->s_comp));
		}
	}
	{
		Slot &sl = p->slot[(nbatches - 1) % D];
		PIPE_HIP(p, hipEventSynchronize(sl.ev_comp));
		if (n_get[nbatches - 1])
			PIPE_HIP(p, hipMemcpyAsync(outs[nbatches - 1], resolve ? (const void *)sl.loc : (const void *)sl.out,
					(size_t)n_get[nbatches - 1] * (resolve ? 1 : 2) * sizeof(loc_t), hipMemcpyDeviceToHost, p->s_out));
	}
	PIPE_HIP(p, hipStreamSynchronize(p->s_in));
	PIPE_HIP(p, hipStreamSynchronize(p->s_comp));
	PIPE_HIP(p, hipStreamSynchronize(p->s_out));
	return MKV_OK;
}

// pinned host memory for the pipeline's callers that have no allocator of their own for it
extern "C" void *mkv_host_alloc(int64_t bytes)
{
	void *ptr = nullptr;
	if (bytes <= 0 || hipHostMalloc(&ptr, (size_t)bytes, hipHostMallocDefault) != hipSuccess)
		return nullptr;
	return ptr;
}

extern "C" void mkv_host_free(void *ptr)
{
	if (ptr)
		(void)hipHostFree(ptr);
}
