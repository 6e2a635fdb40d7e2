// mkv_common.h -- device-side geometry and bucket helpers shared by the HIP
// kernels of the MI355X hash index (gfx950 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "megakv_hip.h"

// HIP event pairs recorded on the caller's stream around a search launch
// (mkv_timer_create / mkv_cycle / mkv_slots_cycle): bench.py's kernel timing.
struct MkvTimer {
	int slots;
	hipEvent_t *ev;
};

namespace mkv {

constexpr int kWave = 64;
constexpr uint32_t kSlots = 8;          // ELEM_NUM
constexpr uint32_t kBucketWords = 16;   // 8 sig + 8 loc dwords
constexpr uint32_t kNone = 0xffffffffu;

// Geometry as the kernels consume it (derived once on the host).
struct DevGeom {
	uint32_t hash_mask;     // HASH_MASK        (ref gpu_hash.h:61)
	uint32_t blk_mask;      // BLOCK_HASH_MASK  (ref gpu_hash.h:69)
	uint32_t region_shift;  // bucket >> region_shift = closed region id
	uint32_t nregions;
	uint32_t max_cuckoo;
	uint32_t scheme;
};

inline bool make_dev_geom(const mkv_geom_t *g, DevGeom *d)
{
	if (!g || g->mem_p < 6 + g->iblock_p || g->mem_p > 33 || g->iblock_p > 5)
		return false;
	if (g->scheme != MKV_SCHEME_CUCKOO && g->scheme != MKV_SCHEME_2CHOICE)
		return false;
	if (g->max_cuckoo > 6)          // eviction-chain logs hold 7 evictions (reference MAX_CUCKOO_NUM = 5)
		return false;
	uint32_t bucket_bits = g->mem_p - 6;
	if (bucket_bits > 31)
		return false;
	d->hash_mask = (uint32_t)(((uint64_t)1 << bucket_bits) - 1);
	d->blk_mask = (uint32_t)(((uint64_t)1 << (bucket_bits - g->iblock_p)) - 1);
	d->region_shift = bucket_bits - g->iblock_p;
	d->nregions = 1u << g->iblock_p;
	d->max_cuckoo = g->max_cuckoo;
	d->scheme = g->scheme;
	return true;
}

// Alternate bucket: the low (region-local) bits are XORed with the signature,
// the region bits are kept (ref gpu_hash.cu:66-67).
__device__ __forceinline__ uint32_t alt_bucket(uint32_t hash, uint32_t sig,
		uint32_t hash_mask, uint32_t blk_mask)
{
	return (((hash ^ sig) & blk_mask) | (hash & ~blk_mask)) & hash_mask;
}

// 8-bit mask of the slots of a signature row equal to `sig`.
__device__ __forceinline__ uint32_t eq_mask(const uint4 &lo, const uint4 &hi, uint32_t sig)
{
	return (uint32_t)(lo.x == sig) | ((uint32_t)(lo.y == sig) << 1) |
		((uint32_t)(lo.z == sig) << 2) | ((uint32_t)(lo.w == sig) << 3) |
		((uint32_t)(hi.x == sig) << 4) | ((uint32_t)(hi.y == sig) << 5) |
		((uint32_t)(hi.z == sig) << 6) | ((uint32_t)(hi.w == sig) << 7);
}

// Slot an insert takes among the empty slots E: the lowest one at or above
// (orig_sig & 7), else the lowest (ref gpu_hash.cu:301-309).
__device__ __forceinline__ uint32_t preferred_slot(uint32_t empties, uint32_t orig_sig)
{
	uint32_t ml = (1u << (orig_sig & 7u)) - 1u;
	uint32_t ballot = ((empties & ml) << 16) | (empties & ~ml);
	return (uint32_t)(__ffs((int)ballot) - 1) & 7u;
}

// Process-wide tuning knobs (defaults = measured best on MI355X).  Initialised from the environment
// (MKV_<NAME>) on first use, changeable through mkv_set_tuning().  Eighteen of them; what only tests and probes
// ever set -- how the tail takes an order violation back, layouts kept for the record, sizes whose sweeps are
// flat -- exists in the diagnostics build only (DiagTuning below) and is a constant in the product.
struct Tuning {
	int search_u;           // requests in flight per 4-lane group: 1, 2, 4 or 8
	int search_nt;          // non-temporal table loads in search
	int tail_rounds;        // reservation rounds inside the tail work-group (0..48)
	long long req_cap;      // default request capacity of an update workspace
	long long insert_chunk; // requests per canonical-index chunk when the caller hints the batch size
	int mark_blocks;        // work-groups of the insert mark pass inside the fused search launch
	int pack_threads;       // one-launch pack: lanes per work-group (256, 512 or 1024)
	int barrier_sticky;     // calls for which a call whose chains ran late counts as "chains lately"
	int round_blocks;       // work-groups of the round-1 / round-2 launches of an insert or delete (at most)
	int tail_groups;        // work-groups of the tail launch (all share its tag scans; group 0 does the rest)
	int dup_groups;         // duplicates of a key inside one pass are settled as a group in round 2 (kFollowBit); 0: off
	int lazy_alt;           // a request whose primary bucket shows at least this many free slots at mark time does
	                        // not register on its alternate bucket (0: every request registers on both)
	int self_help;          // round 1: a bucket's second-lowest registrant commits beside the lowest one when it can tell
	                        // what that one will do there (self-help, mkv_update.hip); 0: it waits for round 2
	int plans;              // the mark pass plans what a request does in its primary bucket and flags who is not a bucket's lowest
	                        // registrant; round 1 carries the plans out without reading bucket or reservation (kPlanBit); 0: as before
	int merge_round2;       // round 2 runs as the first phase of the tail launch (0: a launch of its own; 1: unless round 1 has
	                        // been leaving it a lot; 2: always)
	long long merge_max;    // ... for passes announced with at most this many requests
	long long sorted_min;   // an insert pass announced with at least this many requests (one pass, not a chunk of several) is
	                        // partitioned into bins and applied per bucket, whole-bucket write-back, no reservation atomics
	                        // (mkv_upd_apply.h); 0: never
	int sorted_cooldown;    // ... and after such a pass that fell back or left over more than one request in 256, this many big
	                        // passes run the old way before one is tried again (0: never back off -- tests)
};
Tuning &tuning();

// Diagnostics build only (make DIAG=1): knobs of tests and probes.  The product uses the defaults.
struct DiagTuning {
	int search_wide = 0;        // 8 lanes per request, one 64-byte access per bucket (search_body_wide)
	int search_lds = 0;         // signature rows staged in LDS, ballot/ffs probe, dependent location load (search_body_lds)
	int pack_three_pass = 0;    // sharded pack: the three-kernel stable path instead of the one-launch pack
	int grid_min = 2048;        // a final tail list longer than this runs its rounds on all work-groups of the launch
	int grid_progress = 96;     // grid-wide tail rounds go on while they resolve at least this many requests (else: in-order replay)
	int calm_factor = 1;        // chunk size multiplier while no chain ran late lately
	int mark_alone_blocks = 1024;   // work-groups of the stand-alone mark launch
	long long debug_first_stamp = 0; // first round stamp of a new workspace (exercises the stamp wrap guard)
	int debug_skip = 0;         // (results unchanged) how the tail deals with an order violation -- 0 as designed, 1 always the
	                            // general way (resolve_violations), 2 always the region-suffix fall-back; +4: rounds 1 and 2
	                            // leave every eviction chain to the tail
};
#ifdef MKV_DIAGNOSTICS
DiagTuning &diag_tuning();
#define MKV_DIAG(field) (::mkv::diag_tuning().field)
#else
#define MKV_DIAG(field) (::mkv::DiagTuning().field)
#endif


// A launch whose start and end time are delivered into HIP events when given (the dispatch's
// own timestamps, hipExtLaunchKernelGGL); a plain launch otherwise.
#define MKV_LAUNCH_TIMED(kernel, grid, block, stream, e0, e1, ...)                                  \
	do {                                                                                            \
		if ((e0) || (e1))                                                                           \
			hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, e0, e1, 0, __VA_ARGS__);           \
		else                                                                                        \
			hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                         \
	} while (0)

#define MKV_HIP_TRY(expr)                                   \
	do {                                                    \
		hipError_t e_ = (expr);                             \
		if (e_ != hipSuccess) return MKV_EHIP;              \
	} while (0)

}  // namespace mkv
