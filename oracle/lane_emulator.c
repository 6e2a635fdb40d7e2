/*
 * lane_emulator.c -- a second, structurally independent reading of the reference's insert
 * kernels: eight lanes per request, every lane-level step spelled out.
 *
 * TEST INFRASTRUCTURE ONLY (tests/test_lane_emulator.py).  Not the reference, not the
 * product, and not what the parity tests compare the HIP library with -- that is
 * gpuhash_oracle.c, which states the kernels' canonical serial outcome in collapsed form
 * ("duplicate? else lowest preferred empty slot? else evict ...").  This file instead walks
 * down libgpuhash/gpu_hash.cu:77-229 (hash_insert_2choice) and :231-433
 * (hash_insert_cuckoo) the way the hardware would for ONE 8-lane group at a time:
 *
 *   - a warp of 32 lanes holds four groups; the group at position g sees its slice of every
 *     ballot at bit offset bit_move = 8 g (ref :96, :255); the other groups are idle here
 *     (one request in flight -- the only schedule for which the reference is deterministic,
 *     SURVEY Appendix B-1), so their ballot bits are 0;
 *   - a lane's predicate is evaluated per lane, the ballot assembled bit by bit, the
 *     reference's shifts, masks, the 16-bit "major location" rotation (ref :139-147,
 *     :301-309) and __ffs applied to it as written there;
 *   - writes happen lane by lane under the reference's `simd_lane == chosen_simd` guards,
 *     the write / __syncthreads / re-read claim loops (ref :138-165, :303-329, :352-430)
 *     are kept as loops -- with one group in flight they run once, which is the point of
 *     checking them against the collapsed form.
 *
 * Two independent readings agreeing over a fuzz campaign (lane_fuzz below) is the best
 * evidence this repository can give for slot choice and the eviction chain, which the
 * reference's own tests do not pin.
 */
#include <stdint.h>
#include <string.h>

#include "gpuhash_oracle.h"

#define LANES 8			/* ELEM_NUM = 1 << ELEM_NUM_P, libgpuhash/gpu_hash.h:71-73 */
#define LANE_BITS 3

typedef struct group_s {
	uint32_t bit_move;	/* where this group's lanes sit in the warp-wide ballot */
} group_t;

/* __ffs: 1-based position of the lowest set bit, 0 for 0 */
static int ffs32(uint32_t v)
{
	int n = 1;

	if (!v)
		return 0;
	while (!(v & 1u)) {
		v >>= 1;
		n++;
	}
	return n;
}

/* __ballot over the warp with only this group's lanes active: bit (bit_move + lane) is the
 * lane's predicate; then the reference's extraction of the group's byte (ref :120-123) */
static uint32_t group_ballot(const group_t *grp, const int pred[LANES])
{
	uint32_t warp = 0;
	int lane;

	for (lane = 0; lane < LANES; lane++)
		if (pred[lane])
			warp |= 1u << (grp->bit_move + (uint32_t)lane);
	return (warp >> grp->bit_move) & ((1u << LANES) - 1u);
}

/* the empty-slot ballot rotated around the major location: bits below it go to the upper
 * half-word, so __ffs finds the lowest empty lane at or above it first (ref :145, :307) */
static uint32_t rotate_major(uint32_t ballot, uint32_t ml_mask)
{
	return ((ballot & ml_mask) << 16) | (ballot & ~ml_mask);
}

/* "ballot = __ballot(b->sig[simd_lane] == x)" */
static uint32_t ballot_sig_equals(const group_t *grp, const orc_bucket_t *b, uint32_t x)
{
	int pred[LANES], lane;

	for (lane = 0; lane < LANES; lane++)
		pred[lane] = b->sig[lane] == x;
	return group_ballot(grp, pred);
}

/* "if (simd_lane == chosen_simd) b->FIELD[simd_lane] = v" executed by all eight lanes */
static void lanes_store_sig(orc_bucket_t *b, int chosen, uint32_t v)
{
	int lane;

	for (lane = 0; lane < LANES; lane++)
		if (lane == chosen)
			b->sig[lane] = v;
}

static void lanes_store_loc(orc_bucket_t *b, int chosen, uint32_t v)
{
	int lane;

	for (lane = 0; lane < LANES; lane++)
		if (lane == chosen)
			b->loc[lane] = v;
}

static uint32_t alt_index(const orc_geom_t *g, uint32_t hash, uint32_t sig)
{
	const uint32_t hm = orc_hash_mask(g), bm = orc_block_mask(g);

	return (((hash ^ sig) & bm) | (hash & ~bm)) & hm;	/* ref :171-172, :334-335 */
}

/* one request through hash_insert_2choice, ref :97-226 */
static void lane_insert_2choice_one(const orc_geom_t *g, orc_bucket_t *table, const orc_ielem_t *elem,
		const group_t *grp)
{
	const uint32_t sig = elem->sig;
	orc_bucket_t *b;
	uint32_t ballot, ml_mask;
	int chosen;

	if (elem->sig == 0 && elem->loc == 0)		/* :101-104 */
		return;
	b = &table[elem->hash & orc_hash_mask(g)];
	ballot = ballot_sig_equals(grp, b, elem->sig);	/* :118-123 */
	if (ballot != 0) {
		chosen = (ffs32(ballot) - 1) & (LANES - 1);
		lanes_store_loc(b, chosen, elem->loc);
		return;					/* "continue" */
	}
	ml_mask = (1u << (elem->sig & (LANES - 1))) - 1u;	/* :139 */
	for (;;) {					/* :141-165 */
		ballot = rotate_major(ballot_sig_equals(grp, b, 0u), ml_mask);
		chosen = 0;
		if (ballot != 0) {
			chosen = (ffs32(ballot) - 1) & (LANES - 1);
			lanes_store_sig(b, chosen, sig);
		}
		/* __syncthreads() */
		if (ballot != 0) {
			if (b->sig[chosen] == sig) {
				lanes_store_loc(b, chosen, elem->loc);
				return;			/* goto finish */
			}
		} else {
			break;
		}
	}
	b = &table[alt_index(g, elem->hash, sig)];	/* :171-173 */
	ballot = ballot_sig_equals(grp, b, elem->sig);	/* :176-186 */
	if (ballot != 0) {
		chosen = (ffs32(ballot) - 1) & (LANES - 1);
		lanes_store_loc(b, chosen, elem->loc);
		return;
	}
	for (;;) {					/* :188-222 */
		ballot = rotate_major(ballot_sig_equals(grp, b, 0u), ml_mask);
		if (ballot != 0) {
			chosen = (ffs32(ballot) - 1) & (LANES - 1);
			lanes_store_sig(b, chosen, sig);
		} else {
			/* no room: the major location takes the signature, nothing else (:197-209) */
			chosen = (int)(elem->sig & (LANES - 1));
			lanes_store_sig(b, chosen, sig);
			break;
		}
		/* __syncthreads() */
		if (b->sig[chosen] == sig) {
			lanes_store_loc(b, chosen, elem->loc);
			return;
		}
	}
}

/* one request through hash_insert_cuckoo, ref :256-430 */
static void lane_insert_cuckoo_one(const orc_geom_t *g, orc_bucket_t *table, const orc_ielem_t *elem,
		const group_t *grp)
{
	uint32_t sig, loc, hash, new_sig = 0, new_loc = 0;
	uint32_t ballot, ml_mask, cuckoo_num;
	orc_bucket_t *b;
	int chosen;

	if (elem->sig == 0 && elem->loc == 0)		/* :259-262 */
		return;
	sig = elem->sig;
	hash = elem->hash;
	loc = elem->loc;
	b = &table[hash & orc_hash_mask(g)];
	ballot = ballot_sig_equals(grp, b, sig);	/* :277-287 */
	if (ballot != 0) {
		chosen = (ffs32(ballot) - 1) & (LANES - 1);
		lanes_store_loc(b, chosen, loc);
		return;
	}
	ml_mask = (1u << (sig & (LANES - 1))) - 1u;	/* :301 -- never recomputed below */
	for (;;) {					/* :303-329 */
		ballot = rotate_major(ballot_sig_equals(grp, b, 0u), ml_mask);
		chosen = 0;
		if (ballot != 0) {
			chosen = (ffs32(ballot) - 1) & (LANES - 1);
			lanes_store_sig(b, chosen, sig);
		}
		/* __syncthreads() */
		if (ballot != 0) {
			if (b->sig[chosen] == sig) {
				lanes_store_loc(b, chosen, loc);
				return;
			}
		} else {
			break;
		}
	}
	cuckoo_num = 0;					/* :331 */
cuckoo_evict:
	b = &table[alt_index(g, hash, sig)];		/* :334-336: `hash` is the request's, `sig` the travelling one */
	ballot = ballot_sig_equals(grp, b, sig);	/* :339-349 */
	if (ballot != 0) {
		chosen = (ffs32(ballot) - 1) & (LANES - 1);
		lanes_store_loc(b, chosen, loc);
		return;
	}
	for (;;) {					/* :351-427 */
		ballot = rotate_major(ballot_sig_equals(grp, b, 0u), ml_mask);
		if (ballot != 0) {
			chosen = (ffs32(ballot) - 1) & (LANES - 1);
		} else {
			chosen = (int)(elem->sig & (LANES - 1));	/* :360: the REQUEST's signature */
			if (cuckoo_num < g->max_cuckoo) {
				new_sig = b->sig[chosen];		/* :361-365 */
				new_loc = b->loc[chosen];
			}
		}
		/* __syncthreads() */
		lanes_store_sig(b, chosen, sig);	/* :371-384, both branches */
		/* __syncthreads() */
		if (ballot != 0) {
			if (b->sig[chosen] == sig) {	/* :388-395 */
				lanes_store_loc(b, chosen, loc);
				return;
			}
		} else if (cuckoo_num < g->max_cuckoo) {	/* :397-413 */
			cuckoo_num++;
			if (b->sig[chosen] == sig) {
				lanes_store_loc(b, chosen, loc);
				sig = new_sig;
				loc = new_loc;
				goto cuckoo_evict;
			}
			return;				/* lost the slot to somebody: gives up */
		} else {				/* :414-423 */
			if (b->sig[chosen] == sig)
				lanes_store_loc(b, chosen, loc);
			return;
		}
	}
}

/* the blocks in index order, the requests of a block in index order; request i of a block sits
 * in group (i mod 4) of its warp, as elem_id = idx >> ELEM_NUM_P puts it (ref :94, :253) */
void lane_insert(const orc_geom_t *g, orc_bucket_t *table, const orc_ielem_t *const *blk_input,
		const int *blk_elem_num, int num_blks)
{
	int blk, i;

	for (blk = 0; blk < num_blks; blk++) {
		for (i = 0; i < blk_elem_num[blk]; i++) {
			group_t grp;

			grp.bit_move = (uint32_t)(i & 3) << LANE_BITS;
			if (g->scheme == ORC_SCHEME_2CHOICE)
				lane_insert_2choice_one(g, table, &blk_input[blk][i], &grp);
			else
				lane_insert_cuckoo_one(g, table, &blk_input[blk][i], &grp);
		}
	}
}

/* ---- fuzz driver: both readings on the same random tables and request sequences --------- */
static uint64_t rng_next(uint64_t *s)
{
	uint64_t x = *s;

	x ^= x >> 12;
	x ^= x << 25;
	x ^= x >> 27;
	*s = x;
	return x * 0x2545f4914f6cdd1dull;
}

/* One sequence: a table of 2^(mem_p-6) buckets filled to `fill_pct`, then nreq requests from a
 * small signature space (duplicates, full buckets and eviction chains everywhere), a share of
 * them with signature 0 (and loc 0), with no region-local signature bits (alternate ==
 * primary), and -- the case the task names -- built so that the first victim's alternate is
 * the bucket it was evicted from.  Returns 0 when lane_insert and orc_insert leave the same
 * table, else 1.  scratch: 2 tables + the requests. */
#define FUZZ_MAX_BUCKETS 256
#define FUZZ_MAX_REQ 512

int lane_fuzz_one(uint64_t seed, uint32_t *shape_out)
{
	static orc_bucket_t ta[FUZZ_MAX_BUCKETS], tb[FUZZ_MAX_BUCKETS];
	static orc_ielem_t req[FUZZ_MAX_REQ];
	uint64_t s = seed * 0x9e3779b97f4a7c15ull + 0x1234567ull;
	orc_geom_t g;
	const orc_ielem_t *blocks[4];
	int counts[4];
	orc_stats_t st;
	uint32_t nb, nreq, sig_space, fill, i, k;

	memset(&st, 0, sizeof(st));
	g.mem_p = 8u + (uint32_t)(rng_next(&s) % 7u);			/* 4 .. 256 buckets */
	g.iblock_p = (uint32_t)(rng_next(&s) % 4u);
	if (g.iblock_p > g.mem_p - 6u)
		g.iblock_p = g.mem_p - 6u;
	g.scheme = (uint32_t)(rng_next(&s) & 1u);
	g.max_cuckoo = (uint32_t)(rng_next(&s) % 7u);
	nb = 1u << (g.mem_p - 6u);
	sig_space = (rng_next(&s) & 1u) ? 24u : 4000u;
	fill = (uint32_t)(rng_next(&s) % 101u);
	nreq = 1u + (uint32_t)(rng_next(&s) % FUZZ_MAX_REQ);
	for (i = 0; i < nb; i++)
		for (k = 0; k < 8; k++) {
			const int used = rng_next(&s) % 100u < fill;

			ta[i].sig[k] = used ? 1u + (uint32_t)(rng_next(&s) % sig_space) : 0u;
			ta[i].loc[k] = (used || (rng_next(&s) & 3u) == 0u) ? (uint32_t)rng_next(&s) | 1u : 0u;
		}
	memcpy(tb, ta, sizeof(orc_bucket_t) * nb);
	for (i = 0; i < nreq; i++) {
		const uint32_t kind = (uint32_t)(rng_next(&s) % 16u);

		req[i].sig = 1u + (uint32_t)(rng_next(&s) % sig_space);
		req[i].hash = (uint32_t)rng_next(&s);
		req[i].loc = (uint32_t)rng_next(&s) | 1u;
		if (kind == 0) {
			req[i].sig = 0;				/* matches empty slots */
			if (rng_next(&s) & 1u)
				req[i].loc = 0;			/* ... and this one is skipped */
		} else if (kind == 1) {
			req[i].sig &= ~orc_block_mask(&g);	/* alternate == primary */
			if (req[i].sig == 0)
				req[i].sig = orc_block_mask(&g) + 1u;
		} else if (kind == 2) {
			/* the victim of the alternate bucket sends the chain back to where it came from:
			 * victim signature v with ALT(hash, v) == ALT(hash, sig), i.e. v == sig in the
			 * region-local bits */
			const uint32_t a = alt_index(&g, req[i].hash, req[i].sig);

			ta[a].sig[req[i].sig & 7u] = tb[a].sig[req[i].sig & 7u] =
					(req[i].sig & orc_block_mask(&g)) | ((uint32_t)rng_next(&s) & ~orc_block_mask(&g)) | 0x80000000u;
		}
	}
	/* blocks: up to four, cut anywhere (the insert order is block, then index) */
	{
		uint32_t cut[5], nblk = 1u + (uint32_t)(rng_next(&s) % 4u);

		cut[0] = 0;
		for (i = 1; i < nblk; i++)
			cut[i] = (uint32_t)(rng_next(&s) % (nreq + 1u));
		cut[nblk] = nreq;
		for (i = 1; i < nblk; i++)			/* sort the few cuts */
			for (k = i; k > 0 && cut[k] < cut[k - 1]; k--) {
				const uint32_t t = cut[k];

				cut[k] = cut[k - 1];
				cut[k - 1] = t;
			}
		for (i = 0; i < nblk; i++) {
			blocks[i] = req + cut[i];
			counts[i] = (int)(cut[i + 1] - cut[i]);
		}
		lane_insert(&g, ta, blocks, counts, (int)nblk);
		orc_insert(&g, tb, blocks, counts, (int)nblk, &st);
	}
	if (shape_out) {
		shape_out[0] = g.mem_p;
		shape_out[1] = g.scheme;
		shape_out[2] = g.max_cuckoo;
		shape_out[3] = nreq;
		shape_out[4] = (uint32_t)st.evictions;
		shape_out[5] = (uint32_t)st.dropped;
		shape_out[6] = (uint32_t)st.skipped_zero;
	}
	return memcmp(ta, tb, sizeof(orc_bucket_t) * nb) != 0;
}

/* seeds first .. first+count-1; returns the number of sequences on which the two readings
 * differ, the first such seed in *first_bad, and totals of what the sequences exercised */
int64_t lane_fuzz(uint64_t first, int64_t count, uint64_t *first_bad, uint64_t totals[4])
{
	int64_t bad = 0, i;

	totals[0] = totals[1] = totals[2] = totals[3] = 0;
	for (i = 0; i < count; i++) {
		uint32_t shape[7];

		if (lane_fuzz_one(first + (uint64_t)i, shape)) {
			if (!bad && first_bad)
				*first_bad = first + (uint64_t)i;
			bad++;
		}
		totals[0] += shape[3];
		totals[1] += shape[4];
		totals[2] += shape[5];
		totals[3] += shape[6];
	}
	return bad;
}
