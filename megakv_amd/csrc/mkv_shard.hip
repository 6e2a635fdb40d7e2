// megakv_amd -- sharded index, one rank's engine in C.
//
// A step of the sharded index is  pack -> exchange -> search+insert of the received
// slots -> exchange of the results -> unpack  (mkv_route.hip, mkv_update.hip).  This file
// drives that sequence for a whole run of batches from one C call, with ONE schedule
// (run_batches): the local phases, which update one table in batch order, run back to
// back on a compute stream of their own; everything else of a batch runs on a service
// stream; B buffer sets keep as many batches in flight.  Exchanges go straight to RCCL
// (grouped ncclSend / ncclRecv per peer, or ncclAllToAll) over a communicator of the
// engine's own; the library is opened with dlopen -- the C ABI of the hot path
// (libgpuhash.a) does not depend on it.  Collectives are issued in the same order on every
// rank: fwd(0) .. fwd(B-1), {back(0), fwd(B)}, {back(1), fwd(B+1)}, ...
//
// New design: the reference is single-GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <new>

#include "megakv_hip.h"

extern "C" int mkv_pack_one_launch(int64_t n_set, int nshards, int64_t scratch_bytes);
extern "C" int64_t mkv_pack_meta_words(int64_t n_get);
extern "C" int mkv_pack_requests_meta(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, uint32_t *tile_meta, void *stream);
extern "C" int mkv_unpack_results_tiled(const loc_t *back, const int32_t *where, int64_t n_get, loc_t *out,
		const uint32_t *tile_meta, int nshards, int64_t cap_get, void *stream);

namespace {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllToAll)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
};

// one buffer set: a batch in flight
struct Lane {
	uint32_t *sendbuf = nullptr, *recvbuf = nullptr;
	loc_t *out_slots = nullptr;
	// Returned results and the batch's `where` array, one of each per parity of the set's use: the
	// local phase of batch i+B writes this rank's own answers straight into the results buffer while
	// the un-pack of batch i, which nobody waits for, may still be reading the other one.
	loc_t *back_slots[2] = {nullptr, nullptr};
	int32_t *perm2[2] = {nullptr, nullptr};
	uint32_t *meta2[2] = {nullptr, nullptr};    // where each GET tile's runs went (mkv_pack_requests_meta), beside perm2
	uint32_t *scratch = nullptr;
	ielem_t **set_ptrs = nullptr;
	hipEvent_t ev_local[2] = {nullptr, nullptr};   // alternated so a pending wait keeps its event
	hipEvent_t ev_recv[2] = {nullptr, nullptr};    // the set's requests have arrived (and its result slots have been sent)
	hipEvent_t ev_done = nullptr;
};

}  // namespace

struct mkv_shard {
	mkv_geom_t geom;
	int world = 1, rank = 0;
	bucket_t *table = nullptr;
	int64_t n_get = 0, n_set = 0, cap_get = 0, cap_set = 0, slot_words = 0, scratch_bytes = 0;
	uint32_t agg_mask = 0, shift = 0;
	int *overflow = nullptr;
	int loopback = 0;               // rccl_path "loopback": the exchanges are local copies (tests)
	bool tiled_unpack = false;      // the un-pack goes by the pack's tiles (k_unpack_tiled)
	int self_direct = 0;            // what a rank sends to itself bypasses the exchange (the pack writes that slot
	                                // into the receive buffer, the slot search its results into the buffer of
	                                // returned results); the peers' slots go by grouped send / receive.  Settled
	                                // when the engine is created, from its sizes alone: alike on every rank.
	                                // MKV_SHARD_SELF=0: everything through ncclAllToAll (which copies the own slot itself)
	int nlanes = 2;                 // buffer sets = batches in flight (MKV_SHARD_LANES = 2..4)
	Lane lane[4];
	hipEvent_t ev_start = nullptr;
	hipStream_t cstream = nullptr;  // every local phase, one after the other
	hipEvent_t ev_cdone = nullptr;
	int host_paced = 1;             // the host, not the stream, waits for a local phase's requests (MKV_SHARD_PACED=0)
	int nsvc = 1;                   // service streams (MKV_SHARD_SVC = 1..3; default 1 rank: 1, else 3)
	hipStream_t svc[3] = {nullptr, nullptr, nullptr};
	int delay_unpack_us = 0;        // tests, diagnostics build (MKV_SHARD_DELAY_UNPACK_US): every un-pack starts that much later
	const char *phase = "idle";     // what the engine was doing when a call failed (mkv_shard_error)
	Rccl api;
	ncclComm_t comm = nullptr;
	char err[320] = {0};
};

// MKV_SHARD_TRACE=n (diagnostics): device timestamps around every operation of n batches
// near the end of a run (events with timing; the run is synchronised at its end and the
// table goes to stderr)
namespace {
struct TraceRec { const char *what; int lane, batch; hipEvent_t a, b; };
struct Trace {
	int on = -1, n = 0;
	bool quiet = false;             // mkv_shard_trace: the spans are summed per phase instead of printed
	double sum_us[4] = {0, 0, 0, 0};
	int cnt[4] = {0, 0, 0, 0};
	TraceRec rec[512];
};
Trace g_trace;
const char *const kPhaseNames[4] = {"pack", "exchange", "local", "unpack"};

struct Span {
	TraceRec *r = nullptr;
	hipStream_t s;
	Span(const char *what, int lane, int batch, hipStream_t stream, bool active) : s(stream)
	{
		if (!active || g_trace.n >= 512)
			return;
		r = &g_trace.rec[g_trace.n++];
		r->what = what;
		r->lane = lane;
		r->batch = batch;
		(void)hipEventCreate(&r->a);
		(void)hipEventCreate(&r->b);
		(void)hipEventRecord(r->a, s);
	}
	~Span()
	{
		if (r)
			(void)hipEventRecord(r->b, s);
	}
};
}  // namespace

// ---- diagnostics build only (make DIAG=1): the stand-ins a one-GPU box measures and tests W-rank schedules with ----
// (the product library has neither: tests/test_gpu_sharded.py and tools/loopback_trace.py load libmegakv_hip_diag.so)
#ifdef MKV_DIAGNOSTICS
// MKV_SHARD_FAKE_XFER_US=n (diagnostics): every exchange is followed by a one-wave kernel that
// waits n microseconds on the lane's stream -- a stand-in for the time the exchange takes over
// xGMI when only one GPU is there to measure schedules on
__global__ void k_wait_us(unsigned us)
{
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (unsigned long long)us * 100ull)
		__builtin_amdgcn_s_sleep(8);
}
static int g_fake_xfer_us = -1;
static void fake_xfer(hipStream_t s)
{
	if (g_fake_xfer_us < 0) {
		const char *v = getenv("MKV_SHARD_FAKE_XFER_US");
		g_fake_xfer_us = v ? atoi(v) : 0;
	}
	if (g_fake_xfer_us > 0)
		hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, s, (unsigned)g_fake_xfer_us);
}

// rccl_path "loopback": the schedule of a W-rank run on ONE process and one GPU, for tests
// (RCCL wants a GPU per rank, so a one-GPU box can only form a one-rank communicator, where
// every request goes to slot 0 and the same place every time).  Every peer is this process: the
// slot packed for shard s comes back as the slot received from source rank s, so the local
// table is served W slots with W different `where` arrays per batch -- pack, slot search,
// one insert block per source rank, result exchange and unpack run exactly as with W ranks,
// only the wire is a device copy.  Canonical order of the local table = (slot, batch index).
namespace {
int g_loop_world = 1;
ncclResult_t loop_id(ncclUniqueId *id) { memset(id, 0, sizeof(*id)); return ncclSuccess; }
ncclResult_t loop_init(ncclComm_t *c, int, ncclUniqueId, int) { *c = reinterpret_cast<ncclComm_t>(&g_loop_world); return ncclSuccess; }
ncclResult_t loop_destroy(ncclComm_t) { return ncclSuccess; }
ncclResult_t loop_a2a(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t, hipStream_t s)
{
	return hipMemcpyAsync(recv, send, (size_t)g_loop_world * count * 4u, hipMemcpyDeviceToDevice, s) == hipSuccess
			? ncclSuccess : ncclUnhandledCudaError;
}
const char *loop_err(ncclResult_t) { return "loopback exchange failed"; }
// grouped send / receive: what is sent to peer p comes back as what is received from peer p --
// the k-th send to a peer in a group meets the k-th receive from it, as in RCCL
struct LoopOp { const void *send; void *recv; size_t sbytes, rbytes; hipStream_t s; };
constexpr int kLoopOps = 4;                 // sends (receives) per peer in one group
LoopOp g_loop_ops[64][kLoopOps];
int g_loop_ns[64], g_loop_nr[64];
bool g_loop_open = false;
ncclResult_t loop_group_start()
{
	memset(g_loop_ops, 0, sizeof(g_loop_ops));
	memset(g_loop_ns, 0, sizeof(g_loop_ns));
	memset(g_loop_nr, 0, sizeof(g_loop_nr));
	g_loop_open = true;
	return ncclSuccess;
}
ncclResult_t loop_send(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t s)
{
	if (!g_loop_open || peer < 0 || peer >= 64 || g_loop_ns[peer] >= kLoopOps)
		return ncclInvalidUsage;
	LoopOp &o = g_loop_ops[peer][g_loop_ns[peer]++];
	o.send = buf;
	o.sbytes = count * 4u;
	o.s = s;
	return ncclSuccess;
}
ncclResult_t loop_recv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t s)
{
	if (!g_loop_open || peer < 0 || peer >= 64 || g_loop_nr[peer] >= kLoopOps)
		return ncclInvalidUsage;
	LoopOp &o = g_loop_ops[peer][g_loop_nr[peer]++];
	o.recv = buf;
	o.rbytes = count * 4u;
	o.s = s;
	return ncclSuccess;
}
// MKV_SHARD_LOOP_WGS=n (default 0: a device copy per pair): a group leaves as ONE launch of n work-groups
// instead, like RCCL's grouped send / receive kernel with n channels.  Measured with eight looped-back ranks
// (profiles/r04_loopback.txt, one box): 94 us per batch with the copies, 197 / 136 / 105 / 88 us with 8 / 16 /
// 32 / 64 work-groups -- a few work-groups beside a search launch that saturates the memory system get a
// small share of it, and here both ends of every transfer are this GPU's memory.  The copies stay the
// default stand-in for the wire; MKV_SHARD_FAKE_XFER_US adds wire time behind them.
constexpr int kLoopGroupOps = 32;
struct LoopGroup { const uint4 *src[kLoopGroupOps]; uint4 *dst[kLoopGroupOps]; uint32_t n16[kLoopGroupOps]; int nops; };
__global__ __launch_bounds__(256) void k_loop_group(LoopGroup a)
{
	for (int k = 0; k < a.nops; k++) {
		const uint4 *__restrict__ s = a.src[k];
		uint4 *__restrict__ d = a.dst[k];
		const uint32_t n = a.n16[k], stride = gridDim.x * 256u;
		uint32_t i = blockIdx.x * 256u + threadIdx.x;
		for (; i + 7u * stride < n; i += 8u * stride) {
			uint4 v[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
				v[u] = s[i + (uint32_t)u * stride];
#pragma unroll
			for (int u = 0; u < 8; u++)
				d[i + (uint32_t)u * stride] = v[u];
		}
		for (; i < n; i += stride)
			d[i] = s[i];
	}
}
ncclResult_t loop_group_end()
{
	g_loop_open = false;
	static int wgs = -1;
	if (wgs < 0) {
		const char *v = getenv("MKV_SHARD_LOOP_WGS");
		wgs = v ? atoi(v) : 0;
		if (wgs < 0 || wgs > 4096)
			wgs = 0;
	}
	LoopGroup grp;
	grp.nops = 0;
	hipStream_t gs = nullptr;
	bool one_launch = wgs > 0;
	for (int p = 0; p < 64; p++) {
		if (g_loop_ns[p] != g_loop_nr[p])
			return ncclInvalidUsage;         // a send without its receive (or the other way round)
		for (int k = 0; k < g_loop_ns[p]; k++) {
			const LoopOp &o = g_loop_ops[p][k];
			if (o.sbytes != o.rbytes)
				return ncclInvalidUsage;
			const bool fits = grp.nops < kLoopGroupOps && (o.sbytes & 15u) == 0u && (o.sbytes >> 4) < 0xffffffffull &&
					((reinterpret_cast<uintptr_t>(o.send) | reinterpret_cast<uintptr_t>(o.recv)) & 15u) == 0u &&
					(grp.nops == 0 || gs == o.s);
			if (one_launch && fits) {
				grp.src[grp.nops] = reinterpret_cast<const uint4 *>(o.send);
				grp.dst[grp.nops] = reinterpret_cast<uint4 *>(o.recv);
				grp.n16[grp.nops] = (uint32_t)(o.sbytes >> 4);
				grp.nops++;
				gs = o.s;
			} else if (hipMemcpyAsync(o.recv, o.send, o.sbytes, hipMemcpyDeviceToDevice, o.s) != hipSuccess) {
				return ncclUnhandledCudaError;
			}
		}
	}
	if (grp.nops) {
		hipLaunchKernelGGL(k_loop_group, dim3((unsigned)wgs), dim3(256), 0, gs, grp);
		if (hipGetLastError() != hipSuccess)
			return ncclUnhandledCudaError;
	}
	return ncclSuccess;
}
Rccl loopback_api()
{
	Rccl a;
	a.GetUniqueId = loop_id;
	a.CommInitRank = loop_init;
	a.CommDestroy = loop_destroy;
	a.AllToAll = loop_a2a;
	a.GetErrorString = loop_err;
	a.GroupStart = loop_group_start;
	a.GroupEnd = loop_group_end;
	a.Send = loop_send;
	a.Recv = loop_recv;
	return a;
}
}  // namespace
#else
static inline void fake_xfer(hipStream_t) {}
#endif

static int fail(mkv_shard *e, int code, const char *what, const char *detail)
{
	snprintf(e->err, sizeof(e->err), "[rank %d of %d, %s] %s%s%s", e->rank, e->world, e->phase, what, detail ? ": " : "",
			detail ? detail : "");
	return code;
}

#define SH_HIP(e, expr)                                                       \
	do {                                                                      \
		hipError_t r_ = (expr);                                               \
		if (r_ != hipSuccess)                                                 \
			return fail(e, MKV_EHIP, #expr, hipGetErrorString(r_));           \
	} while (0)

#define SH_NCCL(e, expr)                                                                          \
	do {                                                                                          \
		ncclResult_t r_ = (expr);                                                                 \
		if (r_ != ncclSuccess)                                                                    \
			return fail(e, MKV_EHIP, #expr, e->api.GetErrorString ? e->api.GetErrorString(r_) : nullptr); \
	} while (0)

extern "C" const char *mkv_shard_error(const mkv_shard *e)
{
	return e ? e->err : "no engine";
}

extern "C" void mkv_shard_destroy(mkv_shard *e)
{
	if (!e)
		return;
	(void)hipDeviceSynchronize();
	if (e->comm && e->api.CommDestroy)
		(void)e->api.CommDestroy(e->comm);
	for (hipStream_t sv : e->svc)
		if (sv)
			(void)hipStreamDestroy(sv);
	for (Lane &l : e->lane) {
		if (l.sendbuf) (void)hipFree(l.sendbuf);
		if (l.recvbuf) (void)hipFree(l.recvbuf);
		if (l.out_slots) (void)hipFree(l.out_slots);
		for (loc_t *b : l.back_slots)
			if (b) (void)hipFree(b);
		for (uint32_t *m : l.meta2)
			if (m) (void)hipFree(m);
		for (int32_t *w : l.perm2)
			if (w) (void)hipFree(w);
		if (l.scratch) (void)hipFree(l.scratch);
		if (l.set_ptrs) (void)hipFree(l.set_ptrs);
		for (hipEvent_t ev : l.ev_local)
			if (ev) (void)hipEventDestroy(ev);
		for (hipEvent_t ev : l.ev_recv)
			if (ev) (void)hipEventDestroy(ev);
		if (l.ev_done) (void)hipEventDestroy(l.ev_done);
	}
	if (e->overflow) (void)hipFree(e->overflow);
	if (e->ev_start) (void)hipEventDestroy(e->ev_start);
	if (e->ev_cdone) (void)hipEventDestroy(e->ev_cdone);
	if (e->cstream) {
		mkv_forget_stream(e->cstream);
		(void)hipStreamDestroy(e->cstream);
	}
	// the RCCL handle stays open: the host framework usually holds the same library
	delete e;
}

// geom: the LOCAL table's geometry (region split = aggregate IBLOCK_P - log2 world);
// n_get / n_set: the largest batch a rank hands in; cap_*: slot capacities per peer.
extern "C" int mkv_shard_create(mkv_shard **out, const mkv_geom_t *geom, int world, int rank, bucket_t *table,
		int64_t n_get, int64_t n_set, int64_t cap_get, int64_t cap_set, const char *rccl_path)
{
	if (!out || !geom || !table || world < 1 || world > 64 || (world & (world - 1)) || rank < 0 || rank >= world ||
			n_get < 0 || n_set < 0 || cap_get < 0 || cap_set < 0)
		return MKV_EINVAL;
	mkv_shard *e = new (std::nothrow) mkv_shard();
	if (!e)
		return MKV_ENOMEM;
	*out = e;                       // the caller reads mkv_shard_error() and destroys it on failure
	e->geom = *geom;
	e->world = world;
	e->rank = rank;
	e->table = table;
	e->n_get = n_get;
	e->n_set = n_set;
	e->cap_get = cap_get;
	e->cap_set = cap_set;
	int k = 0;
	while ((1 << k) < world)
		k++;
	e->agg_mask = (uint32_t)((1ull << (geom->mem_p + k - 6)) - 1ull);
	e->shift = (uint32_t)(geom->mem_p - 6);
	e->phase = "create";
	// buffer sets = batches in flight: two with one rank (67.9 - 69.7 us per step against 70.3 - 71.7
	// with three), three with several (the exchanges take their time on the service streams)
	e->nlanes = world > 1 ? 3 : 2;
	if (const char *v = getenv("MKV_SHARD_LANES")) {
		const int n = atoi(v);
		if (n >= 2 && n <= 4)
			e->nlanes = n;
	}
	if (const char *v = getenv("MKV_SHARD_PACED"))
		e->host_paced = atoi(v) != 0;
	// (one service stream per buffer set with several ranks: with two, the stream that serves two of the
	// three sets -- pack 30, exchange, un-pack 10, three hand-overs -- was what a batch waited for:
	// eight looped-back ranks 103 -> 87 us per batch, two 73.5 -> 66.2; profiles/r04_loopback.txt)
	e->nsvc = world > 1 ? 3 : 1;
	if (const char *v = getenv("MKV_SHARD_SVC")) {
		const int n = atoi(v);
		if (n >= 1 && n <= 3)
			e->nsvc = n;
	}
	// (the epilogue of a run records an event of lane k behind service stream k: never more service streams
	// than buffer sets -- MKV_SHARD_LANES=2 with several ranks)
	if (e->nsvc > e->nlanes)
		e->nsvc = e->nlanes;
	if (const char *v = getenv("MKV_SHARD_DELAY_UNPACK_US"))
		e->delay_unpack_us = atoi(v);
	e->slot_words = mkv_pack_slot_words(cap_get, cap_set);
	e->scratch_bytes = mkv_pack_scratch_bytes(n_get ? n_get : 1, n_set ? n_set : 1, world);
	// What a rank sends to itself bypasses the exchange -- if the one-launch pack, which alone can
	// write the own slot elsewhere, will do for the LARGEST batch the engine is made for (a smaller one
	// has fewer tiles): settled here once, from arguments every rank passes alike, so that no rank
	// changes the form of its exchanges in the middle of a run or apart from its peers.
	e->self_direct = mkv_pack_one_launch(n_set, world, e->scratch_bytes) != 0;
	// ... and the one-launch pack notes where its tiles' runs went: the un-pack can stream the results back run by run
	// instead of gathering them request by request (k_unpack_tiled).  Measured on one MI355X, 2^20-request batches, 95/5
	// Zipf (profiles/r06_loopback.txt): one rank 57.1 -> 54.5 us per batch through bench.py, 59.7 -> 58.0 looped back;
	// EIGHT looped-back ranks 79.9 -> 82.5 us -- a tile then holds eight runs of a thousand results, and 122 work-groups
	// of 1 024 lanes with two meetings each do not keep up with 4 096 independent ones.  So: up to two ranks by default
	// (MKV_SHARD_TILED_UNPACK=0 / 1: never / whenever the pack is one launch).
	e->tiled_unpack = e->self_direct && cap_get > 0 && world <= 2;
	if (const char *v = getenv("MKV_SHARD_TILED_UNPACK"))
		e->tiled_unpack = e->self_direct && cap_get > 0 && atoi(v) != 0;
	if (const char *v = getenv("MKV_SHARD_SELF"))
		e->self_direct = e->self_direct && atoi(v) != 0;

	if (rccl_path && !strcmp(rccl_path, "loopback")) {
#ifdef MKV_DIAGNOSTICS
		// test shape (tests/test_gpu_sharded.py): every peer is this process -- see loopback_api()
		e->api = loopback_api();
		g_loop_world = world;
		e->loopback = 1;
#else
		return fail(e, MKV_EINVAL, "create", "rccl_path \"loopback\" (every peer looped back) exists in the diagnostics build only");
#endif
	} else {
	e->api.lib = dlopen(rccl_path && *rccl_path ? rccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!e->api.lib)
		return fail(e, MKV_EHIP, "dlopen(rccl)", dlerror());
#define SH_SYM(field, name)                                                              \
	do {                                                                                 \
		*reinterpret_cast<void **>(&e->api.field) = dlsym(e->api.lib, name);             \
		if (!e->api.field)                                                               \
			return fail(e, MKV_EHIP, "dlsym", name);                                     \
	} while (0)
	SH_SYM(GetUniqueId, "ncclGetUniqueId");
	SH_SYM(CommInitRank, "ncclCommInitRank");
	SH_SYM(CommDestroy, "ncclCommDestroy");
	SH_SYM(AllToAll, "ncclAllToAll");
	SH_SYM(GetErrorString, "ncclGetErrorString");
	SH_SYM(GroupStart, "ncclGroupStart");
	SH_SYM(GroupEnd, "ncclGroupEnd");
	SH_SYM(Send, "ncclSend");
	SH_SYM(Recv, "ncclRecv");
#undef SH_SYM
	}

	SH_HIP(e, hipMalloc(&e->overflow, sizeof(int)));
	SH_HIP(e, hipMemset(e->overflow, 0, sizeof(int)));
	SH_HIP(e, hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming));
	const size_t slot_bytes = (size_t)world * (size_t)e->slot_words * 4u;
	const size_t res_bytes = (size_t)world * (size_t)(cap_get ? cap_get : 1) * sizeof(loc_t) * 2u;
	for (int li = 0; li < e->nlanes; li++) {
		Lane &l = e->lane[li];
		SH_HIP(e, hipMalloc(&l.sendbuf, slot_bytes));
		SH_HIP(e, hipMalloc(&l.recvbuf, slot_bytes));
		SH_HIP(e, hipMalloc(&l.out_slots, res_bytes));
		SH_HIP(e, hipMemset(l.sendbuf, 0, slot_bytes));
		SH_HIP(e, hipMemset(l.recvbuf, 0, slot_bytes));
		SH_HIP(e, hipMemset(l.out_slots, 0, res_bytes));
		const size_t where_bytes = (size_t)(n_get ? n_get : 1) * sizeof(int32_t);
		for (int k2 = 0; k2 < 2; k2++) {
			SH_HIP(e, hipMalloc(&l.back_slots[k2], res_bytes));
			SH_HIP(e, hipMemset(l.back_slots[k2], 0, res_bytes));
			SH_HIP(e, hipMalloc(&l.perm2[k2], where_bytes));   // where each GET went
			SH_HIP(e, hipMalloc(&l.meta2[k2], (size_t)mkv_pack_meta_words(n_get) * 4));
			SH_HIP(e, hipMemset(l.meta2[k2], 0, (size_t)mkv_pack_meta_words(n_get) * 4));
		}
		SH_HIP(e, hipMalloc(&l.scratch, (size_t)e->scratch_bytes + 64));
		SH_HIP(e, hipMemset(l.scratch, 0, (size_t)e->scratch_bytes + 64));   // the pack's control words start at zero
		SH_HIP(e, hipMalloc(&l.set_ptrs, (size_t)world * sizeof(ielem_t *)));
		ielem_t *ptrs[64];
		for (int s = 0; s < world; s++)
			ptrs[s] = reinterpret_cast<ielem_t *>(l.recvbuf + (size_t)s * (size_t)e->slot_words + 4 + 2 * (size_t)cap_get);
		SH_HIP(e, hipMemcpy(l.set_ptrs, ptrs, (size_t)world * sizeof(ielem_t *), hipMemcpyHostToDevice));
		for (hipEvent_t &ev : l.ev_local)
			SH_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		for (hipEvent_t &ev : l.ev_recv)
			SH_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		SH_HIP(e, hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming));
	}
	// (streams are multiplexed over a handful of hardware queues: two streams that share one wait
	// for each other's kernels -- only the streams the schedule uses are created)
	for (int k2 = 0; k2 < e->nsvc; k2++)
		SH_HIP(e, hipStreamCreateWithFlags(&e->svc[k2], hipStreamNonBlocking));
	{
		// The local phases are the critical path.  Their stream gets the highest priority: the
		// runtime pools its hardware queues per priority, so this stream never shares a queue with a
		// service stream (which would make pack, exchange and un-pack take turns with the local
		// kernels instead of running beside them) however many streams the host and RCCL have made
		// before -- on a one-GPU box the counts happen to fall right, with torch.distributed's own
		// streams in the process they need not (the host-buffer pipeline lost 35 % that way,
		// mkv_pipe.hip).  MKV_SHARD_CPRIO=0: normal priority.
		const char *v = getenv("MKV_SHARD_CPRIO");
		int prio_least = 0, prio_greatest = 0;
		SH_HIP(e, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
		if (v && atoi(v) == 0)
			SH_HIP(e, hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking));
		else
			SH_HIP(e, hipStreamCreateWithPriority(&e->cstream, hipStreamNonBlocking, prio_greatest));
	}
	SH_HIP(e, hipEventCreateWithFlags(&e->ev_cdone, hipEventDisableTiming));
	if (cap_set) {
		const int rc = mkv_reserve(&e->geom, (long long)world * cap_set + 1024, e->cstream);
		if (rc != MKV_OK)
			return fail(e, rc, "mkv_reserve", nullptr);
	}
	SH_HIP(e, hipDeviceSynchronize());
	e->phase = "created";
	return MKV_OK;
}

extern "C" int mkv_pack_requests_self(const selem_t *gets, int64_t n_get, const ielem_t *sets, int64_t n_set,
		uint32_t hash_mask_agg, uint32_t shard_shift, int nshards, uint32_t *sendbuf, int64_t cap_get,
		int64_t cap_set, int32_t *perm_get, void *scratch, int64_t scratch_bytes, int *overflow,
		int self, uint32_t *self_slot, void *stream);
extern "C" int mkv_pack_one_launch(int64_t n_set, int nshards, int64_t scratch_bytes);
extern "C" int mkv_slots_cycle_self(const mkv_geom_t *geom, const uint32_t *recvbuf, int nslots, int64_t cap_get,
		int64_t cap_set, loc_t *out_slots, bucket_t *table, ielem_t **set_ptrs, long long size_hint,
		void *stream, void *timer, int slot, int self, loc_t *self_out, void *done_event);

// Block p of `send` (count words each) to peer p, block p of `recv` from peer p, for every peer but
// this rank: the all-to-all as RCCL itself composes it (grouped ncclSend / ncclRecv), minus the copy
// of the own block, which the producing kernels have already put in place.  One rank: nothing.
static int exchange_peers(mkv_shard *e, const void *send, void *recv, size_t count, hipStream_t s)
{
	if (e->world == 1)
		return MKV_OK;
	const uint32_t *sp = static_cast<const uint32_t *>(send);
	uint32_t *rp = static_cast<uint32_t *>(recv);
	SH_NCCL(e, e->api.GroupStart());
	ncclResult_t bad = ncclSuccess;
	for (int p = 0; p < e->world; p++) {
		if (p == e->rank)
			continue;
		ncclResult_t r = e->api.Send(sp + (size_t)p * count, count, ncclInt32, p, e->comm, s);
		if (r == ncclSuccess)
			r = e->api.Recv(rp + (size_t)p * count, count, ncclInt32, p, e->comm, s);
		if (r != ncclSuccess && bad == ncclSuccess)
			bad = r;
	}
	SH_NCCL(e, e->api.GroupEnd());
	SH_NCCL(e, bad);
	return MKV_OK;
}

// rank 0 makes the 128-byte id; the host framework hands it to every rank
extern "C" int mkv_shard_unique_id(mkv_shard *e, void *id128)
{
	if (!e || !id128)
		return MKV_EINVAL;
	ncclUniqueId id;
	SH_NCCL(e, e->api.GetUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return MKV_OK;
}

// collective over all ranks of the job
extern "C" int mkv_shard_connect(mkv_shard *e, const void *id128)
{
	if (!e || !id128 || e->comm)
		return MKV_EINVAL;
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	e->phase = "connect (ncclCommInitRank)";
	SH_NCCL(e, e->api.CommInitRank(&e->comm, e->world, id, e->rank));
	e->phase = "connected";
	if (e->loopback)
		return MKV_OK;                 // (block j comes back as block j: nothing to check)
	e->phase = "first exchange (all-to-all self-check)";
	// Self-check of the exchange before anything relies on it: block j of rank i must arrive
	// as block i of rank j.  (With one rank every layout looks right, so this is the only
	// place a wrong assumption about the collective would show before the results do.)
	const int w = e->world;
	uint32_t host[64 * 4], *dsend = nullptr, *drecv = nullptr;
	for (int j = 0; j < w; j++)
		for (int k = 0; k < 4; k++)
			host[j * 4 + k] = ((uint32_t)e->rank << 16) | ((uint32_t)j << 8) | (uint32_t)k;
	SH_HIP(e, hipMalloc(&dsend, (size_t)w * 16));
	SH_HIP(e, hipMalloc(&drecv, (size_t)w * 16));
	SH_HIP(e, hipMemcpy(dsend, host, (size_t)w * 16, hipMemcpyHostToDevice));
	SH_HIP(e, hipMemset(drecv, 0xff, (size_t)w * 16));
	hipStream_t cs0 = e->svc[0];
	SH_NCCL(e, e->api.AllToAll(dsend, drecv, 4, ncclInt32, e->comm, cs0));
	SH_HIP(e, hipStreamSynchronize(cs0));
	SH_HIP(e, hipMemcpy(host, drecv, (size_t)w * 16, hipMemcpyDeviceToHost));
	(void)hipFree(dsend);
	(void)hipFree(drecv);
	for (int i = 0; i < w; i++)
		for (int k = 0; k < 4; k++)
			if (host[i * 4 + k] != (((uint32_t)i << 16) | ((uint32_t)e->rank << 8) | (uint32_t)k))
				return fail(e, MKV_EHIP, "all-to-all self-check", "block from a peer is not where it belongs");
	// ... and the grouped send / receive form the runs use for the peers' slots, once, the same way
	if (e->self_direct && w > 1) {
		e->phase = "first exchange (grouped send / receive self-check)";
		SH_HIP(e, hipMalloc(&dsend, (size_t)w * 16));
		SH_HIP(e, hipMalloc(&drecv, (size_t)w * 16));
		for (int j = 0; j < w; j++)
			for (int k = 0; k < 4; k++)
				host[j * 4 + k] = 0x5a000000u | ((uint32_t)e->rank << 16) | ((uint32_t)j << 8) | (uint32_t)k;
		SH_HIP(e, hipMemcpy(dsend, host, (size_t)w * 16, hipMemcpyHostToDevice));
		SH_HIP(e, hipMemset(drecv, 0xff, (size_t)w * 16));
		const int rc = exchange_peers(e, dsend, drecv, 4, cs0);
		if (rc != MKV_OK)
			return rc;
		SH_HIP(e, hipStreamSynchronize(cs0));
		SH_HIP(e, hipMemcpy(host, drecv, (size_t)w * 16, hipMemcpyDeviceToHost));
		(void)hipFree(dsend);
		(void)hipFree(drecv);
		for (int i = 0; i < w; i++)
			for (int k = 0; k < 4 && i != e->rank; k++)
				if (host[i * 4 + k] != (0x5a000000u | ((uint32_t)i << 16) | ((uint32_t)e->rank << 8) | (uint32_t)k))
					return fail(e, MKV_EHIP, "send / receive self-check", "block from a peer is not where it belongs");
	}
	e->phase = "connected";
	return MKV_OK;
}

static void print_trace()
{
	if (g_trace.n) {
		(void)hipDeviceSynchronize();
		for (int k = 0; k < g_trace.n; k++) {
			float t0 = 0, t1 = 0;
			(void)hipEventElapsedTime(&t0, g_trace.rec[0].a, g_trace.rec[k].a);
			(void)hipEventElapsedTime(&t1, g_trace.rec[0].a, g_trace.rec[k].b);
			if (g_trace.quiet) {
				for (int p = 0; p < 4; p++)
					if (!strcmp(g_trace.rec[k].what, kPhaseNames[p])) {
						g_trace.sum_us[p] += (double)(t1 - t0) * 1e3;
						g_trace.cnt[p]++;
					}
				continue;
			}
			fprintf(stderr, "%8.1f -> %8.1f  lane %d %*s%-8s batch %d (%.1f us)\n", t0 * 1e3, t1 * 1e3,
					g_trace.rec[k].lane, g_trace.rec[k].lane * 6, "", g_trace.rec[k].what, g_trace.rec[k].batch,
					(t1 - t0) * 1e3);
		}
		for (int k = 0; k < g_trace.n; k++) {
			(void)hipEventDestroy(g_trace.rec[k].a);
			(void)hipEventDestroy(g_trace.rec[k].b);
		}
		g_trace.n = 0;
	}
}

// The schedule.  The local phases update ONE table in batch order: they run back to back on a
// stream of their own (`cstream`) and never wait for an event that has not long been signalled.
// Everything else of a batch -- pack, exchanges, un-pack -- runs on a service stream; B buffer sets
// keep as many batches in flight.  For batch i (set b = i mod B, which always uses the same service
// stream, parity k = (i / B) mod 2 of the set's use) the service stream does, once local(i) is done:
//     pack(i+B) -> { results of batch i back, requests of batch i+B out } -> un-pack(i)
// the exchange as ONE grouped launch (or nothing at all with one rank), the pack of batch i+B ahead of
// it into the set's other `where` array, so that what local(i+B) needs is on its way first, and the
// un-pack, which nobody waits for, last.  What local(i+B) may write while un-pack(i) still reads --
// this rank's own answers, which bypass the exchange -- goes into the returned-results buffer of the
// other parity; the buffers of parity k are next written by local(i+2B) and the exchange behind it,
// both behind un-pack(i) in the order of the service stream.  Streams are multiplexed over a handful
// of hardware queues, and two streams sharing one wait for each other's kernels (MKV_SHARD_TRACE
// showed 20 - 60 us holes between local phases with 3 lanes x 2 streams + 1): the schedule keeps 2
// streams busy with one rank, 3 with several.
static int exchange_pair(mkv_shard *e, Lane &l, int k, bool back, bool fwd, hipStream_t s)
{
	// (k: parity of the batch whose results come back)
	if (!back && !fwd)
		return MKV_OK;
	if (e->self_direct && e->world == 1)
		return MKV_OK;
	SH_NCCL(e, e->api.GroupStart());
	ncclResult_t bad = ncclSuccess;
	auto note = [&](ncclResult_t r) { if (r != ncclSuccess && bad == ncclSuccess) bad = r; };
	if (e->self_direct) {
		for (int p = 0; p < e->world; p++) {
			if (p == e->rank)
				continue;
			if (back) {
				const size_t c = (size_t)e->cap_get * 2u;
				note(e->api.Send(l.out_slots + (size_t)p * c, c, ncclInt32, p, e->comm, s));
				note(e->api.Recv(l.back_slots[k] + (size_t)p * c, c, ncclInt32, p, e->comm, s));
			}
			if (fwd) {
				const size_t c = (size_t)e->slot_words;
				note(e->api.Send(l.sendbuf + (size_t)p * c, c, ncclInt32, p, e->comm, s));
				note(e->api.Recv(l.recvbuf + (size_t)p * c, c, ncclInt32, p, e->comm, s));
			}
		}
	} else {
		if (back)
			note(e->api.AllToAll(l.out_slots, l.back_slots[k], (size_t)e->cap_get * 2u, ncclInt32, e->comm, s));
		if (fwd)
			note(e->api.AllToAll(l.sendbuf, l.recvbuf, (size_t)e->slot_words, ncclInt32, e->comm, s));
	}
	SH_NCCL(e, e->api.GroupEnd());
	SH_NCCL(e, bad);
	fake_xfer(s);
	return MKV_OK;
}

static double host_us()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static int run_batches(mkv_shard *e, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int64_t *n_set, loc_t *const *outs, hipStream_t main, void *timer,
		int timer_every, int trace_from)
{
	const int B = e->nlanes, K = e->nsvc;
	static int host_trace = -1;                 // MKV_SHARD_HOSTTIME=1 (diagnostics): host clock at the call's milestones
	if (host_trace < 0)
		host_trace = getenv("MKV_SHARD_HOSTTIME") ? 1 : 0;
	const double h0 = host_trace ? host_us() : 0.0;
	double h_pro = 0, h_first = 0, h_sync_sum = 0, h_sync_max = 0;
	hipStream_t C = e->cstream;
	e->phase = "run: prologue (first packs and exchanges)";
	SH_HIP(e, hipEventRecord(e->ev_start, main));
	SH_HIP(e, hipStreamWaitEvent(C, e->ev_start, 0));
	for (int k = 0; k < K; k++)
		SH_HIP(e, hipStreamWaitEvent(e->svc[k], e->ev_start, 0));
	auto pack = [&](Lane &l, int j, hipStream_t s, int set, bool tr) -> int {
		if (n_get[j] > e->n_get || n_set[j] > e->n_set)
			return fail(e, MKV_ETOOBIG, "batch larger than the engine was created for", nullptr);
		Span sp("pack", set, j, s, tr);
		int32_t *where = l.perm2[(j / B) & 1];
		uint32_t *meta = e->tiled_unpack ? l.meta2[(j / B) & 1] : nullptr;
		const int rc = mkv_pack_requests_meta(gets[j], n_get[j], sets[j], n_set[j], e->agg_mask, e->shift, e->world, l.sendbuf,
				e->cap_get, e->cap_set, where, l.scratch, e->scratch_bytes, e->overflow, e->self_direct ? e->rank : -1,
				e->self_direct ? l.recvbuf + (size_t)e->rank * (size_t)e->slot_words : nullptr, meta, s);
		if (rc != MKV_OK)
			return fail(e, rc, "mkv_pack_requests", nullptr);
		return MKV_OK;
	};
	int rc = MKV_OK;
	for (int j = 0; j < B && j < nbatches; j++) {
		Lane &l = e->lane[j];
		hipStream_t S = e->svc[j % K];
		if ((rc = pack(l, j, S, j, false)) != MKV_OK)
			return rc;
		if ((rc = exchange_pair(e, l, 0, false, true, S)) != MKV_OK)
			return rc;
		SH_HIP(e, hipEventRecord(l.ev_recv[0], S));
	}
	if (host_trace)
		h_pro = host_us();
	e->phase = "run: batches";
	for (int i = 0; i < nbatches; i++) {
		const int b = i % B, k = (i / B) & 1, k2 = k ^ 1;
		Lane &l = e->lane[b];
		hipStream_t S = e->svc[b % K];
		const bool tr = i >= trace_from && i < trace_from + g_trace.on;
		const bool more = i + B < nbatches;
		// Local phase: its requests have arrived (and, earlier on the same service stream, the results
		// of batch i-B have left the set's result buffers).  The HOST waits for that -- it happened a
		// local phase or two ago on the device -- and then enqueues the local phase with no wait on
		// the stream at all: on this runtime a kernel that follows a cross-stream wait starts ~10 us
		// late even when the event completed long before (tools/probes/event_probe.hip: 72.3 against
		// 61.5 us per 60-us kernel), while a wait on an event the host already sees complete costs
		// nothing.  So the call runs at most B - 1 local phases ahead of the device.
		if (e->host_paced) {
			const double a = host_trace ? host_us() : 0.0;
			SH_HIP(e, hipEventSynchronize(l.ev_recv[k]));
			if (host_trace) {
				const double d = host_us() - a;
				h_sync_sum += d;
				h_sync_max = d > h_sync_max ? d : h_sync_max;
				if (i == 0)
					h_first = host_us();
			}
		} else
			SH_HIP(e, hipStreamWaitEvent(C, l.ev_recv[k], 0));
		const bool timed = timer && timer_every > 0 && i % timer_every == 0;
		{
			Span sp("local", b, i, C, tr);
			rc = mkv_slots_cycle_self(&e->geom, l.recvbuf, e->world, e->cap_get, e->cap_set, l.out_slots, e->table,
					e->cap_set ? l.set_ptrs : nullptr, (long long)e->n_set, C, timed ? timer : nullptr,
					timed ? i / timer_every : 0, e->self_direct ? e->rank : -1,
					l.back_slots[k] + (size_t)e->rank * (size_t)e->cap_get * 2u, tr ? nullptr : l.ev_local[k]);
		}
		if (rc != MKV_OK)
			return fail(e, rc, "mkv_slots_cycle", nullptr);
		if (tr)                                          // (a traced span puts an event of its own behind the phase)
			SH_HIP(e, hipEventRecord(l.ev_local[k], C));
		// service stream: the set's next batch first, the results of this one behind it
		SH_HIP(e, hipStreamWaitEvent(S, l.ev_local[k], 0));
		if (more && (rc = pack(l, i + B, S, b, tr)) != MKV_OK)
			return rc;
		{
			Span sp("exchange", b, i, S, tr);
			if ((rc = exchange_pair(e, l, k, true, more, S)) != MKV_OK)
				return rc;
		}
		if (more)
			SH_HIP(e, hipEventRecord(l.ev_recv[k2], S));
#ifdef MKV_DIAGNOSTICS
		if (e->delay_unpack_us > 0)
			hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, S, (unsigned)e->delay_unpack_us);
#endif
		{
			Span sp("unpack", b, i, S, tr);
			rc = e->tiled_unpack ? mkv_unpack_results_tiled(l.back_slots[k], l.perm2[k], n_get[i], outs[i], l.meta2[k], e->world,
							e->cap_get, S)
					: mkv_unpack_results(l.back_slots[k], l.perm2[k], n_get[i], outs[i], S);
		}
		if (rc != MKV_OK)
			return fail(e, rc, "mkv_unpack_results", nullptr);
	}
	e->phase = "run: epilogue";
	for (int k = 0; k < K; k++) {
		SH_HIP(e, hipEventRecord(e->lane[k].ev_done, e->svc[k]));
		SH_HIP(e, hipStreamWaitEvent(main, e->lane[k].ev_done, 0));
	}
	SH_HIP(e, hipEventRecord(e->ev_cdone, C));
	SH_HIP(e, hipStreamWaitEvent(main, e->ev_cdone, 0));
	if (host_trace) {
		const double h1 = host_us();
		(void)hipStreamSynchronize(main);
		const double h2 = host_us();
		fprintf(stderr, "run %d batches: prologue enqueued %.0f us, first requests in %.0f us, loop done %.0f us, "
				"device done %.0f us; host waits %.0f us in all, longest %.0f us\n", nbatches, h_pro - h0, h_first - h0,
				h1 - h0, h2 - h0, h_sync_sum, h_sync_max);
	}
	e->phase = "idle";
	return MKV_OK;
}

// nbatches batches of this rank: gets[i] (n_get[i] x selem_t), sets[i] (n_set[i] x ielem_t),
// outs[i] (n_get[i] x 2 loc_t, request order) -- all device pointers, host arrays of them.
// The service and compute streams start after what is queued on `stream` and `stream` waits for
// them at the end; the host paces the local phases (see run_batches).  timer (mkv_timer_*)
// brackets the local search launch of every timer_every-th batch, slot = i / timer_every.
// Every way out of a failed run drains the device first: kernels and exchanges on the caller's
// request and result buffers may still be in flight on the engine's streams.
extern "C" int mkv_shard_run(mkv_shard *e, int nbatches, const selem_t *const *gets, const int64_t *n_get,
		const ielem_t *const *sets, const int64_t *n_set, loc_t *const *outs, void *stream, void *timer,
		int timer_every)
{
	if (!e || !e->comm || nbatches < 0 || (nbatches && (!gets || !n_get || !sets || !n_set || !outs)))
		return MKV_EINVAL;
	if (nbatches == 0)
		return MKV_OK;
	if (g_trace.on < 0) {
		const char *v = getenv("MKV_SHARD_TRACE");
		g_trace.on = v ? atoi(v) : 0;
	}
	const int trace_from = g_trace.on > 0 && nbatches > g_trace.on + 16 ? nbatches - g_trace.on - 8 : nbatches + 1;
	g_trace.n = 0;
	const int rc = run_batches(e, nbatches, gets, n_get, sets, n_set, outs, (hipStream_t)stream, timer, timer_every, trace_from);
	if (rc == MKV_OK)
		print_trace();
	else
		(void)hipDeviceSynchronize();
	return rc;
}

// Device time per phase of a batch, for a host that wants to read a run against a prediction without a
// profiler (bench.py: "phases_us").  mkv_shard_trace(n) arms the NEXT runs of the process: events around every
// operation of n batches near each run's end (as MKV_SHARD_TRACE=n does, without the table on stderr);
// mkv_shard_phases() returns the mean duration of the spans seen since -- pack, exchange (one grouped launch:
// results of batch i back, requests of batch i+B out), local (slot search + mark, rounds, tail), un-pack -- in
// microseconds, and disarms.  The events themselves cost 3 - 5 us each between two kernels of a stream
// (tools/probes/event_probe.hip): run it on a pass of its own, never inside a timed region.
extern "C" void mkv_shard_trace(int nbatches)
{
	g_trace.on = nbatches > 0 ? (nbatches > 64 ? 64 : nbatches) : 0;
	g_trace.quiet = nbatches > 0;
	for (int p = 0; p < 4; p++) {
		g_trace.sum_us[p] = 0;
		g_trace.cnt[p] = 0;
	}
}

extern "C" int mkv_shard_phases(float out_us[4], int counts[4])
{
	if (!out_us)
		return MKV_EINVAL;
	for (int p = 0; p < 4; p++) {
		out_us[p] = g_trace.cnt[p] ? (float)(g_trace.sum_us[p] / g_trace.cnt[p]) : 0.0f;
		if (counts)
			counts[p] = g_trace.cnt[p];
	}
	g_trace.on = 0;
	g_trace.quiet = false;
	return MKV_OK;
}

// synchronises; *flag = a slot overflowed since the engine was created (results invalid)
extern "C" int mkv_shard_overflowed(mkv_shard *e, int *flag)
{
	if (!e || !flag)
		return MKV_EINVAL;
	SH_HIP(e, hipDeviceSynchronize());
	SH_HIP(e, hipMemcpy(flag, e->overflow, sizeof(int), hipMemcpyDeviceToHost));
	return MKV_OK;
}
