/*
 * gpuhash_oracle.h -- CPU restatement of Mega-KV's libgpuhash kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under megakv_amd/ (the product) may
 * include, link or call this; it is used by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker / CPU yardstick.
 */
#ifndef GPUHASH_ORACLE_H
#define GPUHASH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SCHEME_CUCKOO  0
#define ORC_SCHEME_2CHOICE 1

/* Table geometry: the reference fixes these at compile time
 * (libgpuhash/gpu_hash.h:54-69); the oracle takes them at run time so one
 * build checks every configuration. */
typedef struct orc_geom_s {
	uint32_t mem_p;       /* log2(table bytes); bucket count = 2^(mem_p-6) */
	uint32_t iblock_p;    /* reference IBLOCK_P = 3 */
	uint32_t scheme;      /* ORC_SCHEME_* */
	uint32_t max_cuckoo;  /* reference MAX_CUCKOO_NUM = 5 */
} orc_geom_t;

typedef struct orc_selem_s { uint32_t sig, hash; } orc_selem_t;
typedef struct orc_ielem_s { uint32_t sig, hash, loc; } orc_ielem_t;
typedef struct orc_bucket_s { uint32_t sig[8]; uint32_t loc[8]; } orc_bucket_t;

/* Work counters used for the algorithmic-bytes figures:
 * bytes = 12*elems + 64*buckets_visited + 4*dword_stores. */
typedef struct orc_stats_s {
	uint64_t elems;
	uint64_t buckets_visited;
	uint64_t dword_stores;
	uint64_t evictions;     /* cuckoo relocations performed */
	uint64_t dropped;       /* victims dropped after max_cuckoo relocations */
	uint64_t skipped_zero;  /* elements with sig==0 && loc==0 */
} orc_stats_t;

uint32_t orc_hash_mask(const orc_geom_t *g);
uint32_t orc_block_mask(const orc_geom_t *g);
uint32_t orc_primary(const orc_geom_t *g, uint32_t hash);
uint32_t orc_alternate(const orc_geom_t *g, uint32_t hash, uint32_t sig);

void orc_search(const orc_geom_t *g, const orc_bucket_t *table,
		const orc_selem_t *in, uint32_t *out, int64_t n);

void orc_insert(const orc_geom_t *g, orc_bucket_t *table,
		const orc_ielem_t *const *blk_input, const int *blk_elem_num,
		int num_blks, orc_stats_t *stats);

void orc_delete(const orc_geom_t *g, orc_bucket_t *table,
		const orc_ielem_t *in, int64_t n, orc_stats_t *stats);

/* Multi-threaded search over [0,n) split in nthreads contiguous chunks (search
 * is read-only, so any split is exact).  Used only for the CPU baseline. */
void orc_search_mt(const orc_geom_t *g, const orc_bucket_t *table,
		const orc_selem_t *in, uint32_t *out, int64_t n, int nthreads);

/* orc_insert with one host thread per block, for blocks binned by closed region (each block
 * one region, no region twice); returns 1, or 0 (nothing done) if they are not.  CPU baseline only. */
int orc_insert_mt(const orc_geom_t *g, orc_bucket_t *table, const orc_ielem_t *const *blk_input,
		const int *blk_elem_num, int num_blks);

/* One UDP payload of the reference's request protocol (src/mega_recv.c:292-545,
 * README.md:70-81): appends its GETs ({sig, hash}) and SETs ({sig, hash, packet * 2048 + key
 * offset}); returns 0 or -1 like the reference's parser. */
int orc_parse_packet(const uint8_t *payload, uint32_t len, uint32_t packet, uint32_t key_len,
		uint32_t value_len, int fold, uint32_t *gets, int64_t *n_get, uint32_t *sets, int64_t *n_set,
		int64_t *n_bad);

/* What the reference's sender does with the pair a search returned, KEY_MATCH build
 * (src/mega_send.c:408-414 pick, :543-553 verify): loc = pair[0] ? pair[0] : pair[1]; the item it
 * names must hold the request's key -- `it->nkey == key_len && strncmp(ITEM_key(it), key, key_len)
 * == 0` -- else the request counts as a miss.  A flat arena stands in for the slab store (item
 * `l` at arena + l * item_stride, l < arena_items; header src/include/items.h:37-50: nkey at byte
 * nkey_off, key at key_off).  bytes_mode != 0 compares all key_len bytes (memcmp) instead of the
 * reference's strncmp, which stops at the first NUL.  out_loc[i] = loc or 0;
 * totals[0..2] = hits, misses, false positives (a location whose item fails the compare). */
void orc_resolve_verify(const uint32_t *pairs, int64_t n, const uint8_t *keys, int64_t key_stride, int key_len,
		const uint8_t *arena, int64_t item_stride, int64_t arena_items, int nkey_off, int key_off, int bytes_mode,
		uint32_t *out_loc, uint64_t totals[3]);

#ifdef __cplusplus
}
#endif
#endif
