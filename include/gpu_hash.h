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
 *   - MY_ALIGN knows hipcc.
 * Host and library must still be built with the same MEM_P / scheme, exactly
 * as in the reference: both are baked into libgpuhash.a.
 */
#ifndef _GPU_HASH_H_
#define _GPU_HASH_H_

#include <stdint.h>

#if defined(__HIPCC__) || defined(__GNUC__) || defined(__clang__)
	#define MY_ALIGN(n) __attribute__((aligned(n)))
#else
	#error "Please provide a definition for MY_ALIGN macro for your host compiler!"
#endif

typedef uint32_t sign_t;
typedef uint32_t loc_t;
typedef uint32_t hash_t;

#define ELEM_SIG_SIZE	8
#define ELEM_SIZE_P		3	/* 2^3 bytes per (sig, loc) slot */
#define ELEM_NUM_P		3	/* 2^3 slots per bucket */
#define ELEM_NUM		(1 << ELEM_NUM_P)
#define UNIT_THREAD_NUM_P 1
#define UNIT_THREAD_NUM (1 << UNIT_THREAD_NUM_P)

#ifndef MEM_P
#define MEM_P			(30)	/* 2^30 B = 1 GiB table */
#endif

#define BUC_P			(ELEM_NUM_P + ELEM_SIZE_P)	/* 64-byte bucket */
#define HASH_MASK		((uint32_t)(((uint64_t)1 << (MEM_P - BUC_P)) - 1))
#define BUC_NUM			((uint64_t)1 << (MEM_P - BUC_P))
#define HT_SIZE			((uint64_t)1 << (MEM_P))

/* Insert batches are binned into 2^IBLOCK_P blocks; the alternate bucket of
 * an entry only differs from its primary in the bits under BLOCK_HASH_MASK, so
 * the table decomposes into 2^IBLOCK_P closed regions. */
#define IBLOCK_P		3
#define INSERT_BLOCK	(1 << IBLOCK_P)
#define BLOCK_HASH_MASK	((uint32_t)(((uint64_t)1 << (MEM_P - BUC_P - IBLOCK_P)) - 1))

#if !defined(HASH_2CHOICE) && !defined(HASH_CUCKOO)
#define HASH_CUCKOO		1
#endif
#if defined(HASH_2CHOICE) && defined(HASH_CUCKOO)
#error "define only one of HASH_2CHOICE / HASH_CUCKOO"
#endif
#ifdef HASH_CUCKOO
	#define MAX_CUCKOO_NUM	5	/* relocations before the last victim is dropped */
#endif

typedef struct bucket_s {
	sign_t sig[ELEM_NUM];
	loc_t loc[ELEM_NUM];
} MY_ALIGN(64) bucket_t;

/* search request */
typedef struct selem_s {
	sign_t	sig;
	hash_t	hash;
} MY_ALIGN(8) selem_t;

typedef union selem_shared_s {
	selem_t elem_s;
	long long elem_u;
} MY_ALIGN(8) selem_shared_t;

/* insert and delete request */
typedef struct ielem_s {
	sign_t	sig;
	hash_t	hash;
	loc_t   loc;
} ielem_t;

typedef struct ielem_s delem_t;

#endif
