/*
 * gpu_hash.h -- table/request layouts and geometry macros of the batched GPU
 * hash index, MI355X build.
 *
 * This header is the compile-time half of the drop-in boundary.  It keeps the
 * type names, field order, sizes and macro names a C host written against
 * Mega-KV's libgpuhash expects (reference: libgpuhash/gpu_hash.h:40-104), so
 * src/mega_scheduler.c / src/mega_recv.c style callers compile unchanged:
 *
 *   bucket_t  64 B : sig[8] (u32, 0 = empty) @0, loc[8] (u32) @32   (ref :79-82)
 *   selem_t    8 B : {sig, hash}                                    (ref :85-89)
 *   ielem_t   12 B : {sig, hash, loc}; delem_t is the same struct   (ref :98-104)
 *
 * Differences from the reference header, all deliberate:
 *   - MEM_P and the insert scheme can be chosen with -DMEM_P=.. / -DHASH_2CHOICE
 *     (the reference requires editing the file, ref :54-55,72-73);
 *   - HT_SIZE is 64-bit so MEM_P >= 31 is well defined (ref :64 shifts an int);
 *   - the layouts are checked at compile time (sizes and offsets below);
 *   - the kernel-internal helper type of the CUDA version (ref :91-95) is not part
 *     of this boundary.
 * Host and library must still be built with the same MEM_P / scheme, exactly
 * as in the reference: both are baked into libgpuhash.a.
 */
#ifndef _GPU_HASH_H_
#define _GPU_HASH_H_

#include <stddef.h>
#include <stdint.h>

/* ---- geometry: what the library and its host must agree on ------------------------- */

#ifndef MEM_P
#define MEM_P (30)                       /* log2 of the table size in bytes: 1 GiB */
#endif

/* a slot is a 32-bit signature plus a 32-bit location; a bucket holds 2^ELEM_NUM_P of them */
#define ELEM_NUM_P   3
#define ELEM_SIZE_P  3
#define ELEM_NUM     8
#define ELEM_SIG_SIZE 8
#define BUC_P        6                   /* log2 of the bucket size: ELEM_NUM_P + ELEM_SIZE_P */

/* launch-shape constants of the CUDA kernels (two lanes per bucket half); hosts may print them */
#define UNIT_THREAD_NUM_P 1
#define UNIT_THREAD_NUM   2

#define MKV_POW2_U64(p)  ((uint64_t)1 << (p))
#define HT_SIZE          MKV_POW2_U64(MEM_P)                          /* bytes */
#define BUC_NUM          MKV_POW2_U64(MEM_P - BUC_P)                  /* buckets */
#define HASH_MASK        ((uint32_t)(BUC_NUM - 1u))                   /* hash -> primary bucket */

/* Insert batches come in 2^IBLOCK_P blocks.  An entry's alternate bucket differs from its
 * primary bucket only in the bits under BLOCK_HASH_MASK, i.e. the table falls into
 * 2^IBLOCK_P closed regions (ref libgpuhash/gpu_hash.cu:66-67). */
#define IBLOCK_P         3
#define INSERT_BLOCK     8
#define BLOCK_HASH_MASK  ((uint32_t)(MKV_POW2_U64(MEM_P - BUC_P - IBLOCK_P) - 1u))

/* insert scheme: cuckoo unless the build asks for two-choice */
#if defined(HASH_2CHOICE) && defined(HASH_CUCKOO)
#error "define only one of HASH_2CHOICE / HASH_CUCKOO"
#endif
#if !defined(HASH_2CHOICE) && !defined(HASH_CUCKOO)
#define HASH_CUCKOO 1
#endif
#if defined(HASH_CUCKOO)
#define MAX_CUCKOO_NUM 5                 /* evictions of a chain before the last victim is dropped */
#endif

/* ---- layouts --------------------------------------------------------------------------- */

#define MY_ALIGN(n) __attribute__((aligned(n)))

typedef uint32_t sign_t;                 /* signature; 0 marks an empty slot */
typedef uint32_t hash_t;
typedef uint32_t loc_t;                  /* opaque to the index; 0 reads as "miss" to the caller */

struct bucket_s {                        /* one table cell */
	sign_t sig[ELEM_NUM];
	loc_t  loc[ELEM_NUM];
} MY_ALIGN(64);
typedef struct bucket_s bucket_t;

struct selem_s {                         /* search request */
	sign_t sig;
	hash_t hash;
} MY_ALIGN(8);
typedef struct selem_s selem_t;

struct ielem_s {                         /* insert request; a delete request looks the same */
	sign_t sig;
	hash_t hash;
	loc_t  loc;
};
typedef struct ielem_s ielem_t;
typedef struct ielem_s delem_t;

/* the reference's layouts, as its own header yields them with gcc */
#if defined(__cplusplus)
#define MKV_LAYOUT_CHECK(cond, what) static_assert(cond, what)
#else
#define MKV_LAYOUT_CHECK(cond, what) _Static_assert(cond, what)
#endif
MKV_LAYOUT_CHECK(ELEM_NUM == (1 << ELEM_NUM_P) && INSERT_BLOCK == (1 << IBLOCK_P) &&
		BUC_P == ELEM_NUM_P + ELEM_SIZE_P && UNIT_THREAD_NUM == (1 << UNIT_THREAD_NUM_P), "derived constants");
MKV_LAYOUT_CHECK(sizeof(bucket_t) == 64 && offsetof(bucket_t, sig) == 0 && offsetof(bucket_t, loc) == 32,
		"bucket_t: sig row at 0, loc row at 32, 64 bytes");
MKV_LAYOUT_CHECK(sizeof(selem_t) == 8 && offsetof(selem_t, hash) == 4, "selem_t: 8 bytes");
MKV_LAYOUT_CHECK(sizeof(ielem_t) == 12 && offsetof(ielem_t, loc) == 8 && sizeof(delem_t) == 12, "ielem_t: 12 bytes");

#endif
