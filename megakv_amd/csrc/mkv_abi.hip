// mkv_abi.hip -- the drop-in entry points: same names and signatures as the
// reference's extern "C" wrappers (libgpuhash/gpu_hash.cu:482-593), geometry
// and insert scheme baked in from gpu_hash.h at build time like the reference.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include "libgpuhash.h"
#include "megakv_hip.h"
#include "mkv_common.h"

static const mkv_geom_t kGeom = {
	MEM_P, IBLOCK_P,
#if defined(HASH_2CHOICE)
	MKV_SCHEME_2CHOICE, 0
#else
	MKV_SCHEME_CUCKOO, MAX_CUCKOO_NUM
#endif
};

// The reference never checks a launch; a failure here cannot be reported
// through the void ABI, so it is made loud instead of silent.
static void die_on(int rc, const char *what)
{
	if (rc != MKV_OK) {
		fprintf(stderr, "libgpuhash (MI355X): %s failed with code %d\n", what, rc);
		abort();
	}
}

// ---- deferred launches: the fused cycle for a host that keeps to the three entry points ---------------
// The reference's scheduler issues, per worker and stream, memset + gpu_hash_search + a copy of the
// results, then gpu_hash_delete, then gpu_hash_insert, and synchronises the device ONCE per cycle
// (src/mega_scheduler.c:393-504).  Launched as they come that is three to nine small launches per worker,
// the updates of all workers serialised -- 188 Mops device-resident with 16 workers at the scheduler's batch
// sizes, where ONE mkv_cycle_workers call for the whole cycle reaches 1 728.  With
// MKV_DEFER=1 in the environment (or mkv_set_defer(1)) the three entry points only RECORD their arguments,
// per stream; device-to-host copies issued through the library's cudaMemcpyAsync on a stream with recorded
// work are queued behind it; and whatever makes the host wait -- cudaDeviceSynchronize,
// cudaStreamSynchronize, a synchronous copy, cudaFree -- or mkv_flush() issues everything recorded as one
// cycle: one worker -> mkv_cycle on its stream (search and mark pass in one launch); several -> the streams'
// earlier work (the host's request copies) is waited for, then ONE mkv_cycle_workers call on the first
// worker's stream -- every search sees the table as the cycle found it, then all deletes, then all inserts,
// worker by worker in the order their streams first appeared: the order the library gives the eager calls'
// updates anyway (they are serialised in issue order) -- then the queued copies.  Host-to-device copies and
// memsets go out at once, as they do without the mode.  Off by default: the eager calls are the reference's
// semantics to the letter (a search on one stream may see another stream's earlier update, as there).
namespace {

struct DeferredCopy {
	void *dst;
	const void *src;
	size_t n;
	hipStream_t stream;         // the stream the host issued it on (the flush makes that stream wait for it)
};

// Pinned host buffers the host got from the library's cudaHostAlloc (cuda_names.hip): device-accessible at the same
// address, and -- unlike pageable memory, which the runtime stages before an asynchronous copy returns -- not the
// host's to change between cudaMemcpyAsync and the next synchronisation.
std::mutex g_pinned_mutex;
std::map<const char *, size_t> g_pinned;    // by first byte (a scheduler with 16 workers holds 200 of them, and asks 200 times a cycle)

bool pinned_holds(const void *p, size_t n)
{
	std::lock_guard<std::mutex> lock(g_pinned_mutex);
	const char *c = reinterpret_cast<const char *>(p);
	auto it = g_pinned.upper_bound(c);
	if (it == g_pinned.begin())
		return false;
	--it;
	return c >= it->first && c + n <= it->first + it->second;
}

struct DeferredWorker {
	hipStream_t stream;
	bucket_t *table;
	mkv_worker_t w;
	bool has_search, has_delete, has_insert;
};

std::mutex g_defer_mutex;
std::atomic<int> g_defer_queued{0};         // something is recorded, queued or held: what mkv_defer_barrier looks at first
int g_defer_mode = -1;                     // -1: not looked up yet
std::vector<DeferredWorker> g_defer;
std::vector<DeferredCopy> g_defer_copies;
std::vector<DeferredCopy> g_defer_in;       // host-to-device copies from pinned buffers, waiting for the cycle (copy_segments)

// The scheduler clears a worker's result buffer right before its search (cudaMemsetAsync, src/mega_scheduler.c:
// 406-407): the cycle's search writes every result, zeros for misses, so that launch is for nothing.  A memset to
// zero is HELD until the next call of the library says what it is for: a gpu_hash_search on the same stream into
// exactly that buffer drops it, anything else issues it first.
struct HeldMemset {
	void *p;
	size_t n;
	hipStream_t stream;
	bool held;
};
HeldMemset g_held = {nullptr, 0, nullptr, false};

int release_memset_locked()
{
	if (!g_held.held)
		return MKV_OK;
	g_held.held = false;
	return hipMemsetAsync(g_held.p, 0, g_held.n, g_held.stream) == hipSuccess ? MKV_OK : MKV_EHIP;
}

bool defer_on_locked()
{
	if (g_defer_mode < 0) {
		const char *e = getenv("MKV_DEFER");
		g_defer_mode = e && atoi(e) != 0 ? 1 : 0;
	}
	return g_defer_mode == 1;
}

// the entry of `stream` that has no call of this kind yet (a second search on a stream before the
// cycle is issued starts a new entry: a later worker)
int defer_flush_locked();

DeferredWorker *defer_slot(hipStream_t stream, bucket_t *table, int kind)
{
	// A search recorded behind a delete or insert of its OWN stream must see that update (stream order): a flush
	// runs every search of the cycle ahead of every update, so what is recorded goes out first.  (Two searches in a
	// row on one stream see the same table either way and share a cycle.)
	if (kind == 0) {
		for (const DeferredWorker &d : g_defer)
			if (d.stream == stream && (d.has_delete || d.has_insert)) {
				die_on(defer_flush_locked(), "gpu_hash_search/flush of the stream's earlier cycle");
				break;
			}
	}
	for (auto it = g_defer.rbegin(); it != g_defer.rend(); ++it) {
		if (it->stream != stream)
			continue;
		const bool taken = kind == 0 ? (it->has_search || it->has_delete || it->has_insert) :
				kind == 1 ? (it->has_delete || it->has_insert) : it->has_insert;
		// (a worker's calls come search, delete, insert: one out of that order belongs to a new entry)
		if (!taken && it->table == table)
			return &*it;
		break;
	}
	DeferredWorker d;
	memset(&d, 0, sizeof(d));
	d.stream = stream;
	d.table = table;
	g_defer.push_back(d);
	g_defer_queued.store(1, std::memory_order_relaxed);
	return &g_defer.back();
}

// the queued device-to-host copies go out behind the cycle's SEARCH, ahead of its deletes and inserts -- where the
// host issued them (mkv_cycle / mkv_cycle_workers call this hook once the search launch is enqueued)
thread_local hipStream_t t_copy_stream = nullptr;
thread_local bool t_copies_pending = false;
thread_local int t_copy_rc = MKV_OK;

// Small copies from pageable memory are snapshot into a pinned arena (mkv_defer_copy_in): kSnapArenas of them, one per
// flush in turn (the launch that reads an arena is long over when its turn comes again).
constexpr size_t kSnapMax = 4096, kSnapArena = 256 * 1024;
constexpr int kSnapArenas = 4;
char *g_snap = nullptr;
size_t g_snap_used = 0;
int g_snap_cur = 0;

bool snap_room(size_t n)
{
	if (!g_snap && hipHostMalloc(reinterpret_cast<void **>(&g_snap), kSnapArena * kSnapArenas, hipHostMallocDefault) != hipSuccess)
		g_snap = nullptr;
	return g_snap && g_snap_used + ((n + 15u) & ~(size_t)15u) <= kSnapArena;
}

const void *snap_take(const void *src, size_t n)
{
	char *at = g_snap + (size_t)g_snap_cur * kSnapArena + g_snap_used;
	memcpy(at, src, n);
	g_snap_used += (n + 15u) & ~(size_t)15u;
	return at;
}

// The reference's scheduler stages every worker's requests and results with a copy each (src/mega_scheduler.c:404-419,
// 452, 487-494): 12 copies per worker and cycle, 192 with 16 workers, of 32 bytes to 100 KB -- launch-bound, 1.4 ms of
// a cycle whose kernels take a tenth of that.  In deferred mode the copies between DEVICE memory and PINNED host
// buffers of the library's cudaHostAlloc wait for the cycle and go as ONE launch each way: a kernel that reads (writes)
// the pinned buffers over PCIe, a segment per copy.  What the host asked for is what it gets -- the same bytes in the
// same places by the time it synchronises; copies from pageable memory go out at once as before.
struct CopySeg {
	void *dst;
	const void *src;
	unsigned long long n;
};
constexpr int kSegTables = 8, kSegCap = 1024;
CopySeg *g_seg_tables = nullptr;            // kSegTables tables of kSegCap segments, pinned (the kernel reads them over PCIe)
int g_seg_next = 0;

__global__ __launch_bounds__(256) void k_copy_segments(const CopySeg *__restrict__ segs)
{
	const CopySeg sg = segs[blockIdx.y];
	const unsigned long long n = sg.n;
	char *d = reinterpret_cast<char *>(sg.dst);
	const char *s = reinterpret_cast<const char *>(sg.src);
	const unsigned long long lane = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, lanes = (unsigned long long)gridDim.x * blockDim.x;
	if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15u) == 0u) {
		const unsigned long long n16 = n >> 4;
		const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
		uint4 *d4 = reinterpret_cast<uint4 *>(d);
		for (unsigned long long i = lane; i < n16; i += lanes)
			d4[i] = s4[i];
		for (unsigned long long i = (n16 << 4) + lane; i < n; i += lanes)
			d[i] = s[i];
	} else if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 3u) == 0u) {
		const unsigned long long n4 = n >> 2;
		const uint32_t *s1 = reinterpret_cast<const uint32_t *>(s);
		uint32_t *d1 = reinterpret_cast<uint32_t *>(d);
		for (unsigned long long i = lane; i < n4; i += lanes)
			d1[i] = s1[i];
		for (unsigned long long i = (n4 << 2) + lane; i < n; i += lanes)
			d[i] = s[i];
	} else {
		for (unsigned long long i = lane; i < n; i += lanes)
			d[i] = s[i];
	}
}

// the copies of `v` as one launch on `s` (several if there are more than a table holds); individual copies if the
// tables cannot be had
int copy_segments(std::vector<DeferredCopy> &v, hipMemcpyKind kind, hipStream_t s)
{
	int rc = MKV_OK;
	if (v.empty())
		return rc;
	if (!g_seg_tables && hipHostMalloc(reinterpret_cast<void **>(&g_seg_tables), sizeof(CopySeg) * kSegTables * kSegCap,
			hipHostMallocDefault) != hipSuccess)
		g_seg_tables = nullptr;
	if (!g_seg_tables || v.size() < 4) {
		for (const DeferredCopy &c : v)
			if (hipMemcpyAsync(c.dst, c.src, c.n, kind, s) != hipSuccess)
				rc = MKV_EHIP;
		v.clear();
		return rc;
	}
	for (size_t at = 0; at < v.size(); at += kSegCap) {
		// (a table is the host's to rewrite once the launch that read it is over: kSegTables launches ago at the
		// least, each followed by whatever made the host flush -- a wait, as a rule; the ring is insurance)
		CopySeg *tab = g_seg_tables + (size_t)g_seg_next * kSegCap;
		g_seg_next = (g_seg_next + 1) % kSegTables;
		const size_t m = v.size() - at < (size_t)kSegCap ? v.size() - at : (size_t)kSegCap;
		unsigned long long longest = 0;
		for (size_t k = 0; k < m; k++) {
			tab[k].dst = v[at + k].dst;
			tab[k].src = v[at + k].src;
			tab[k].n = v[at + k].n;
			longest = v[at + k].n > longest ? v[at + k].n : longest;
		}
		// (work-groups per segment by the longest one: 16 bytes a lane, ONE turn -- a turn is a round trip over PCIe --
		// within 64 work-groups; the work-groups of a shorter segment find nothing and leave)
		unsigned gx = (unsigned)((longest + 256u * 16u - 1u) / (256u * 16u));
		gx = gx < 1u ? 1u : (gx > 64u ? 64u : gx);
		hipLaunchKernelGGL(k_copy_segments, dim3(gx, (unsigned)m), dim3(256), 0, s, tab);
		if (hipGetLastError() != hipSuccess)
			rc = MKV_EHIP;
	}
	v.clear();
	return rc;
}

void issue_deferred_copies(hipStream_t s)
{
	if (!t_copies_pending)
		return;
	t_copies_pending = false;
	// results into pinned buffers of the library's cudaHostAlloc: one launch; the others a copy each
	std::vector<DeferredCopy> batch;
	for (const DeferredCopy &c : g_defer_copies) {
		if (pinned_holds(c.dst, c.n))
			batch.push_back(c);
		else if (hipMemcpyAsync(c.dst, c.src, c.n, hipMemcpyDeviceToHost, s) != hipSuccess)
			t_copy_rc = MKV_EHIP;
	}
	if (copy_segments(batch, hipMemcpyDeviceToHost, s) != MKV_OK)
		t_copy_rc = MKV_EHIP;
	g_defer_copies.clear();
}

// the queued host-to-device copies: behind what their own streams hold already? -- no: they go on ONE stream, the
// flush's, ahead of the cycle, and every stream they were issued on is made to wait for the flush (join_streams_locked)
int issue_deferred_in(hipStream_t s)
{
	const int rc = copy_segments(g_defer_in, hipMemcpyHostToDevice, s);
	if (g_snap_used) {
		g_snap_used = 0;
		g_snap_cur = (g_snap_cur + 1) % kSnapArenas;
	}
	return rc;
}

// A flush issues the whole cycle and the queued result copies on ONE stream, s0 (the first worker's).  The
// host, though, issued them on a stream per worker and may wait for any of those alone (cudaStreamSynchronize,
// a later copy or kernel of its own on that stream): every OTHER stream that has recorded work or a queued
// copy is made to wait for an event behind the flush, so that stream order holds on each of them as if the
// work had been issued there.
hipEvent_t g_flush_ev = nullptr;

int join_streams_locked(hipStream_t s0, const std::vector<hipStream_t> &others)
{
	if (others.empty())
		return MKV_OK;
	if (!g_flush_ev && hipEventCreateWithFlags(&g_flush_ev, hipEventDisableTiming) != hipSuccess)
		return MKV_EHIP;
	if (hipEventRecord(g_flush_ev, s0) != hipSuccess)
		return MKV_EHIP;
	for (hipStream_t o : others)
		if (hipStreamWaitEvent(o, g_flush_ev, 0) != hipSuccess)
			return MKV_EHIP;
	return MKV_OK;
}

thread_local bool t_in_flush = false;       // (the flush itself calls mkv_cycle / mkv_cycle_workers)

int defer_flush_locked_inner();
int defer_flush_locked()
{
	t_in_flush = true;
	const int rc = defer_flush_locked_inner();
	t_in_flush = false;
	g_defer_queued.store(0, std::memory_order_relaxed);
	return rc;
}

int defer_flush_locked_inner()
{
	int rc = release_memset_locked();
	if (g_defer.empty() && g_defer_copies.empty() && g_defer_in.empty())
		return rc;
	hipStream_t s0 = !g_defer.empty() ? g_defer[0].stream : !g_defer_in.empty() ? g_defer_in[0].stream : g_defer_copies[0].stream;
	std::vector<hipStream_t> others;
	auto note_stream = [&](hipStream_t o) {
		if (o == s0)
			return;
		for (hipStream_t k : others)
			if (k == o)
				return;
		others.push_back(o);
	};
	for (const DeferredWorker &d : g_defer)
		note_stream(d.stream);
	for (const DeferredCopy &c : g_defer_copies)
		note_stream(c.stream);
	for (const DeferredCopy &c : g_defer_in)
		note_stream(c.stream);
	const bool had_in = !g_defer_in.empty();
	size_t i = 0;
	if (g_defer.size() > 1 || (had_in && !others.empty())) {
		// What the host queued on the workers' own streams (request copies, memsets) is done first: ONE wait for
		// the device, not one per stream (a blocking wait is tens of microseconds of wake-up each -- sixteen of
		// them per cycle were a quarter of a staged cycle's time)
		if (hipDeviceSynchronize() != hipSuccess)
			rc = MKV_EHIP;
	}
	// the request copies that waited for the cycle: one launch, ahead of it
	if (rc == MKV_OK)
		rc = issue_deferred_in(s0);
	// (one worker: the copies go out behind its search, where the host issued them; several: behind the whole
	// cycle -- sixteen copies between the search and the updates on the one stream hold the updates up)
	t_copies_pending = !g_defer_copies.empty() && g_defer.size() == 1;
	t_copy_rc = MKV_OK;
	t_copy_stream = s0;
	while (i < g_defer.size() && rc == MKV_OK) {
		// a run of entries on one table, within what one call takes
		size_t j = i;
		int blocks = 0;
		while (j < g_defer.size() && g_defer[j].table == g_defer[i].table && j - i < 32 &&
				blocks + g_defer[j].w.num_blks <= 1024) {
			blocks += g_defer[j].w.num_blks;
			j++;
		}
		if (j == i)
			j = i + 1;
		if (j - i == 1 && g_defer.size() == 1) {
			const DeferredWorker &d = g_defer[i];
			rc = mkv_cycle(&kGeom, d.table, d.w.search_in, d.w.search_out, d.w.num_search, d.w.delete_in, d.w.num_delete,
					d.w.blk_input, d.w.blk_elem_num, d.w.num_blks, 0, 0u, (void *)d.stream, nullptr, 0);
		} else {
			mkv_worker_t ws[32];
			for (size_t k = i; k < j; k++)
				ws[k - i] = g_defer[k].w;
			if (rc == MKV_OK)
				rc = mkv_cycle_workers(&kGeom, g_defer[i].table, ws, (int)(j - i), (void *)s0);
		}
		i = j;
	}
	// (copies the cycle's search did not take with it: no search in the cycle, or no cycle at all)
	t_copies_pending = !g_defer_copies.empty();
	issue_deferred_copies(s0);
	if (rc == MKV_OK)
		rc = t_copy_rc;
	if (rc == MKV_OK)
		rc = join_streams_locked(s0, others);
	g_defer.clear();
	g_defer_copies.clear();
	g_defer_in.clear();
	return rc;
}

}  // namespace

// ---- what the host copied into small device arrays: the block counts of gpu_hash_insert ----------------------
// The ABI hands gpu_hash_insert its block counts in DEVICE memory (ref libgpuhash/gpu_hash.cu:236-237), so the
// library does not know how big a batch is -- and an update workspace holds MKV_REQ_CAP requests unless told
// otherwise (mkv_insert_hinted).  The reference's hosts write those counts with a copy from host memory right before
// the call (src/mega_scheduler.c:487-494), through the library's own cudaMemcpy / cudaMemcpyAsync: small
// host-to-device copies are remembered here (the bytes, by destination), and gpu_hash_insert looks its count array
// up -- a batch of any size is then sized and run in passes like a hinted one, not dropped.  Anything else the
// library sees written over such an array (memset, another copy) forgets it; counts a kernel of the host's wrote
// are not seen at all -- such a batch is bounded by the workspace as before.
namespace {
struct NotedCopy {
	const char *dst = nullptr;
	size_t n = 0;
	int words[1024];
};
constexpr int kNoted = 32;
std::mutex g_noted_mutex;
NotedCopy *g_noted = nullptr;       // kNoted entries, round robin
int g_noted_next = 0;

void forget_noted_locked(const char *d, size_t n)
{
	if (!g_noted)
		return;
	for (int k = 0; k < kNoted; k++)
		if (g_noted[k].dst && d < g_noted[k].dst + g_noted[k].n && g_noted[k].dst < d + n)
			g_noted[k].dst = nullptr;
}
}  // namespace

// a host-to-device copy through the library's cudaMemcpy / cudaMemcpyAsync (src: host memory, still valid)
extern "C" void mkv_note_h2d(void *dst, const void *src, size_t n)
{
	std::lock_guard<std::mutex> lock(g_noted_mutex);
	forget_noted_locked(reinterpret_cast<const char *>(dst), n);
	if (n == 0 || n > sizeof(NotedCopy::words) || (n & 3u))
		return;
	if (!g_noted)
		g_noted = new NotedCopy[kNoted];
	NotedCopy &e = g_noted[g_noted_next];
	g_noted_next = (g_noted_next + 1) % kNoted;
	e.dst = reinterpret_cast<const char *>(dst);
	e.n = n;
	memcpy(e.words, src, n);
}

// device memory written some other way the library sees (memset, device-to-device copy)
extern "C" void mkv_note_write(void *dst, size_t n)
{
	std::lock_guard<std::mutex> lock(g_noted_mutex);
	forget_noted_locked(reinterpret_cast<const char *>(dst), n);
}

// the sum of the `count` ints at device address p if the host is known to have copied them there; -1: unknown
static long long noted_total(const int *p, int count)
{
	std::lock_guard<std::mutex> lock(g_noted_mutex);
	if (!g_noted || count <= 0)
		return -1;
	const char *c = reinterpret_cast<const char *>(p);
	for (int k = 0; k < kNoted; k++) {
		const NotedCopy &e = g_noted[k];
		if (e.dst && c >= e.dst && c + (size_t)count * 4u <= e.dst + e.n) {
			const int *w = e.words + (c - e.dst) / 4;
			long long sum = 0;
			for (int b = 0; b < count; b++)
				sum += w[b] > 0 ? w[b] : 0;
			return sum;
		}
	}
	return -1;
}

// for the library's cudaHostAlloc / cudaFreeHost
extern "C" void mkv_pinned_note(void *p, size_t n, int add)
{
	std::lock_guard<std::mutex> lock(g_pinned_mutex);
	if (add)
		g_pinned[reinterpret_cast<const char *>(p)] = n;
	else
		g_pinned.erase(reinterpret_cast<const char *>(p));
}

// for the library's cudaMemcpyAsync, host to device: a copy from a pinned buffer of the library's cudaHostAlloc waits
// for the cycle in deferred mode (1: queued; 0: issue it now)
extern "C" int mkv_defer_copy_in(void *dst, const void *src, size_t n, void *stream)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	if (!defer_on_locked() || !n)
		return 0;
	if (!pinned_holds(src, n)) {
		// Pageable memory is the host's again the moment the call returns: a SMALL copy from it (the scheduler's
		// block counts, 32 bytes a worker and cycle -- each a launch of its own and 10 - 15 us of the host's time
		// otherwise) is taken NOW, into a pinned arena of the library's, and goes with the cycle's other copies.
		if (n > kSnapMax || !snap_room(n))
			return 0;
		src = snap_take(src, n);
	}
	die_on(release_memset_locked(), "held memset");
	g_defer_in.push_back(DeferredCopy{dst, src, n, (hipStream_t)stream});
	g_defer_queued.store(1, std::memory_order_relaxed);
	return 1;
}

// called by mkv_cycle / mkv_cycle_workers behind their search launch (nothing to do outside a deferred flush)
extern "C" void mkv_after_search(void *stream)
{
	if (t_copies_pending)
		issue_deferred_copies((hipStream_t)stream);
}

extern "C" void mkv_set_defer(int on)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	if (!on)
		die_on(defer_flush_locked(), "mkv_set_defer/flush");
	g_defer_mode = on ? 1 : 0;
}

// For the library's OWN entry points (mkv_search, mkv_insert, mkv_delete, mkv_cycle, mkv_cycle_workers ...) when a host
// mixes them with the drop-in names in deferred mode: what the drop-in side has recorded or queued -- request copies
// from pinned buffers above all -- goes out first, so that the call finds its inputs where the host put them.  Nothing
// to do (one relaxed load) unless something is queued; a no-op inside the flush itself.
extern "C" int mkv_defer_barrier(void)
{
	if (t_in_flush || g_defer_queued.load(std::memory_order_relaxed) == 0)
		return MKV_OK;
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	return defer_flush_locked();
}

// issues whatever the deferred mode has recorded (no wait); MKV_OK when there is nothing
extern "C" int mkv_flush(void)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	return defer_flush_locked();
}

// for the library's cudaMemsetAsync: 1 = held back (see HeldMemset), 0 = issue it now
extern "C" int mkv_defer_memset(void *p, int value, size_t n, void *stream)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	if (!defer_on_locked())
		return 0;
	die_on(release_memset_locked(), "cudaMemsetAsync/held memset");
	if (value != 0 || !p || !n)
		return 0;
	g_held = HeldMemset{p, n, (hipStream_t)stream, true};
	g_defer_queued.store(1, std::memory_order_relaxed);
	return 1;
}

// ... and for its other asynchronous calls: whatever is held goes out first
extern "C" void mkv_defer_touch(void)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	die_on(release_memset_locked(), "held memset");
	// (an asynchronous call that goes out at once -- a copy from pageable memory, a device-to-device copy -- while
	// copies from pinned buffers wait for the cycle and nothing is recorded yet that they could belong to: they go
	// first, in the order the host issued them)
	if (!g_defer_in.empty() && g_defer.empty())
		die_on(defer_flush_locked(), "queued copies");
}

// for the library's cudaMemcpyAsync: a device-to-host copy on a stream with recorded work waits its turn
// (1: queued; 0: issue it now)
extern "C" int mkv_defer_copy(void *dst, const void *src, size_t n, void *stream)
{
	std::lock_guard<std::mutex> lock(g_defer_mutex);
	die_on(release_memset_locked(), "held memset");
	if (g_defer.empty())
		return 0;
	// (any stream: the recorded cycle may write what the copy reads whatever stream it names)
	g_defer_copies.push_back(DeferredCopy{dst, src, n, (hipStream_t)stream});
	g_defer_queued.store(1, std::memory_order_relaxed);
	return 1;
}

extern "C" void gpu_hash_search(selem_t *in, loc_t *out, bucket_t *hash_table, int num_elem,
		int num_thread, int threads_per_blk, cudaStream_t stream)
{
	// argument checks of the reference wrapper (gpu_hash.cu:492-498); the
	// values themselves are CUDA launch hints and do not shape the HIP launch
	assert(num_thread > threads_per_blk);
	assert(threads_per_blk <= 1024);
	(void)num_thread;
	(void)threads_per_blk;
	{
		std::lock_guard<std::mutex> lock(g_defer_mutex);
		if (defer_on_locked()) {
			if (g_held.held && g_held.p == (void *)out && g_held.stream == (hipStream_t)stream && num_elem > 0 &&
					g_held.n == (size_t)num_elem * 2u * sizeof(loc_t))
				g_held.held = false;            // (the cycle's search writes the zeros itself)
			die_on(release_memset_locked(), "gpu_hash_search/held memset");
			if (num_elem <= 0)
				return;
			DeferredWorker *d = defer_slot((hipStream_t)stream, hash_table, 0);
			d->w.search_in = in;
			d->w.search_out = out;
			d->w.num_search = num_elem;
			d->has_search = true;
			return;
		}
		die_on(defer_flush_locked(), "gpu_hash_search/flush");     // (eager: behind whatever is still recorded)
	}
	die_on(mkv_search(&kGeom, in, out, hash_table, num_elem, MKV_SEARCH_KEEP_MISS, (void *)stream),
			"gpu_hash_search");
}

extern "C" void gpu_hash_insert(bucket_t *hash_table, ielem_t **blk_input, int *blk_elem_num,
		int num_blks, cudaStream_t stream)
{
	assert(ELEM_NUM_P < 5 && ELEM_NUM_P > 0);   // gpu_hash.cu:535
	{
		std::lock_guard<std::mutex> lock(g_defer_mutex);
		die_on(release_memset_locked(), "gpu_hash_insert/held memset");
		if (defer_on_locked() && num_blks > 0 && num_blks <= 1024) {
			DeferredWorker *d = defer_slot((hipStream_t)stream, hash_table, 2);
			d->w.blk_input = blk_input;
			d->w.blk_elem_num = blk_elem_num;
			d->w.num_blks = num_blks;
			d->has_insert = true;
			return;
		}
		// (eager -- the mode is off, or the call has more blocks than a recorded cycle takes: behind whatever is
		// recorded, in the order the host issued the calls)
		die_on(defer_flush_locked(), "gpu_hash_insert/flush");
	}
	// (the size of the batch, if the host's copy of the counts came by the library: a batch beyond the workspace is
	// then run in passes; 0 = unknown, bounded by MKV_REQ_CAP)
	const long long known = noted_total(blk_elem_num, num_blks);
	die_on(mkv_insert_hinted(&kGeom, hash_table, blk_input, blk_elem_num, num_blks,
				known > mkv::tuning().req_cap && known <= (1ll << 24) - 16 ? known : 0, (void *)stream), "gpu_hash_insert");
}

extern "C" void gpu_hash_delete(delem_t *in, bucket_t *hash_table, int num_elem, int num_thread,
		int threads_per_blk, cudaStream_t stream)
{
	assert(num_thread >= threads_per_blk);      // gpu_hash.cu:567-568
	assert(threads_per_blk <= 1024);
	(void)num_thread;
	(void)threads_per_blk;
	{
		std::lock_guard<std::mutex> lock(g_defer_mutex);
		die_on(release_memset_locked(), "gpu_hash_delete/held memset");
		if (defer_on_locked()) {
			if (num_elem <= 0)
				return;
			DeferredWorker *d = defer_slot((hipStream_t)stream, hash_table, 1);
			d->w.delete_in = in;
			d->w.num_delete = num_elem;
			d->has_delete = true;
			return;
		}
		die_on(defer_flush_locked(), "gpu_hash_delete/flush");
	}
	die_on(mkv_delete(&kGeom, in, hash_table, num_elem, (void *)stream), "gpu_hash_delete");
}

extern "C" void gpu_delete_insert(bucket_t *hash_table, delem_t *delete_in, uint32_t num_delete_job,
		ielem_t **insert_blk_input, int *insert_blk_elem_num, int num_insert_blks,
		uint32_t num_delete_thread, uint32_t threads_per_blk, cudaStream_t stream)
{
	(void)num_delete_thread;
	(void)threads_per_blk;
	die_on(mkv_flush(), "gpu_delete_insert/flush");     // (issued as it comes, behind whatever is recorded)
	die_on(mkv_delete(&kGeom, delete_in, hash_table, num_delete_job, (void *)stream),
			"gpu_delete_insert/delete");
	die_on(mkv_insert(&kGeom, hash_table, insert_blk_input, insert_blk_elem_num, num_insert_blks,
				(void *)stream), "gpu_delete_insert/insert");
}

// HIP event pairs that receive the start and end time of the search dispatch of mkv_cycle
// (hipExtLaunchKernelGGL on the caller's stream): how bench.py times the dominant kernel.
extern "C" void *mkv_timer_create(int slots)
{
	if (slots <= 0)
		return nullptr;
	MkvTimer *t = new MkvTimer;
	t->slots = slots;
	t->ev = new hipEvent_t[2 * slots];
	for (int i = 0; i < 2 * slots; i++) {
		if (hipEventCreate(&t->ev[i]) != hipSuccess) {
			for (int j = 0; j < i; j++)
				(void)hipEventDestroy(t->ev[j]);
			delete[] t->ev;
			delete t;
			return nullptr;
		}
	}
	return t;
}

extern "C" int mkv_timer_elapsed_ms(void *timer, int slot, float *ms)
{
	MkvTimer *t = reinterpret_cast<MkvTimer *>(timer);
	if (!t || !ms || slot < 0 || slot >= t->slots)
		return MKV_EINVAL;
	return hipEventElapsedTime(ms, t->ev[2 * slot], t->ev[2 * slot + 1]) == hipSuccess ? MKV_OK : MKV_EHIP;
}

extern "C" void mkv_timer_destroy(void *timer)
{
	MkvTimer *t = reinterpret_cast<MkvTimer *>(timer);
	if (!t)
		return;
	for (int i = 0; i < 2 * t->slots; i++)
		(void)hipEventDestroy(t->ev[i]);
	delete[] t->ev;
	delete t;
}

extern "C" int mkv_tuning_is_default_search(void);
extern "C" void mkv_after_search(void *stream);
extern "C" void mkv_call_lock(void);
extern "C" void mkv_call_unlock(void);
extern "C" int mkv_cycle_insert_begin(const mkv_geom_t *geom, const bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);
extern "C" int mkv_cycle_insert_end(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream, int joined);
extern "C" int mkv_cycle_search_mark(const mkv_geom_t *geom, const selem_t *in, loc_t *out,
		const bucket_t *table, int64_t num_search, ielem_t **blk_input, int *blk_elem_num, int num_blks,
		long long size_hint, void *stream, void *ev_start, void *ev_stop);
extern "C" int mkv_search_timed(const mkv_geom_t *geom, const selem_t *in, loc_t *out, const bucket_t *table,
		int64_t num_elem, uint32_t flags, void *stream, void *ev_start, void *ev_stop);
extern "C" int mkv_insert_hinted(const mkv_geom_t *geom, bucket_t *table, ielem_t **blk_input,
		int *blk_elem_num, int num_blks, long long size_hint, void *stream);

// One scheduler cycle for one worker in a single call: clear the result buffer,
// search, delete, insert, all on `stream` (reference src/mega_scheduler.c:393-502
// does exactly this sequence per worker).  Saves host round trips for callers
// that are not bound to the three-call ABI.
extern "C" int mkv_cycle(const mkv_geom_t *geom, bucket_t *table, const selem_t *search_in,
		loc_t *search_out, int64_t num_search, delem_t *delete_in, int64_t num_delete,
		ielem_t **blk_input, int *blk_elem_num, int num_blks, int64_t insert_hint, uint32_t search_flags,
		void *stream, void *timer, int slot)
{
	int rc = mkv_defer_barrier();
	if (rc != MKV_OK)
		return rc;
	MkvTimer *tm = reinterpret_cast<MkvTimer *>(timer);
	if (tm && (slot < 0 || slot >= tm->slots))
		return MKV_EINVAL;
	// host threads that run cycles at once (each on its own stream) take turns for the whole call: the
	// parts below share the device's cycle workspace from the fused launch to the rounds
	struct Turn {
		Turn() { mkv_call_lock(); }
		~Turn() { mkv_call_unlock(); }
	} turn;
	// The insert's mark pass never writes the table (its pre-walk reads it: a hint only, so a
	// delete of the same cycle running beside it is harmless).  Default: it rides in the
	// search launch (extra work-groups of one kernel).  MKV_CYCLE_SIDE_STREAM: it
	// runs on a side stream beside the search kernel.  MKV_CYCLE_NO_OVERLAP: plain
	// sequence of the three entry points.
	const bool can = num_blks > 0 && !(search_flags & MKV_CYCLE_NO_OVERLAP);
	const bool side = can && (search_flags & MKV_CYCLE_SIDE_STREAM);
	const bool fused = can && !side && num_search > 0 && !(search_flags & MKV_CYCLE_MEMSET) &&
			mkv_tuning_is_default_search();
	const bool overlap = side;
	if (side) {
		rc = mkv_cycle_insert_begin(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream);
		if (rc != MKV_OK)
			return rc;
	}
	if (fused) {
		rc = mkv_cycle_search_mark(geom, search_in, search_out, table, num_search, blk_input, blk_elem_num,
				num_blks, insert_hint, stream, tm ? tm->ev[2 * slot] : nullptr, tm ? tm->ev[2 * slot + 1] : nullptr);
		if (rc != MKV_OK)
			return rc;
	} else if (num_search > 0) {
		if (search_flags & MKV_CYCLE_MEMSET) {
			if (hipMemsetAsync(search_out, 0, (size_t)num_search * 2 * sizeof(loc_t),
						(hipStream_t)stream) != hipSuccess)
				return MKV_EHIP;
		}
		// memset + write-on-hit and write-always give the same buffer: fused unless asked
		rc = mkv_search_timed(geom, search_in, search_out, table, num_search,
				(search_flags & MKV_CYCLE_MEMSET) ? MKV_SEARCH_KEEP_MISS : MKV_SEARCH_ZERO_MISS, stream,
				tm ? tm->ev[2 * slot] : nullptr, tm ? tm->ev[2 * slot + 1] : nullptr);
		if (rc != MKV_OK)
			return rc;
	}
	mkv_after_search(stream);
	if (num_delete > 0) {
		rc = mkv_delete(geom, delete_in, table, num_delete, stream);
		if (rc != MKV_OK)
			return rc;
	}
	if (overlap || fused)
		rc = mkv_cycle_insert_end(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream,
				(overlap ? 1 : 0) | (num_delete > 0 ? 2 : 0));
	else if (num_blks > 0)
		rc = mkv_insert_hinted(geom, table, blk_input, blk_elem_num, num_blks, insert_hint, stream);
	return rc;
}

extern "C" const char *mkv_version(void)
{
#ifndef MKV_BUILD_ID
#define MKV_BUILD_ID "unknown"
#endif
	return "megakv_amd libgpuhash gfx950 r6 " MKV_BUILD_ID;       // (the last word: a digest of the library's sources, csrc/Makefile)
}
