/*
 * workload.c -- synthetic request streams for the benchmark and the tests
 * (host-only C, no GPU).  It restates the workload definition of the
 * reference's LOCAL_TEST generator so batches here are the batches the
 * reference would have produced:
 *
 *   Zipf ranks     src/zipf.h:40-185 (CMU mehcached generator after Gray et
 *                  al., SIGMOD'94: approximate pow, 48-bit LCG) -- restated
 *                  exactly; known answers in tests/golden/zipf_kat.json
 *   GET key        rank + 1                         src/mega_recv.c:691
 *   SET key        fresh sequential ids             src/mega_recv.c:727
 *   key -> elem    8-byte key {k, ((k&0xff)<<24 << (8-IBLOCK_P)) | k};
 *                  sig = low word, hash = high word src/mega_recv.c:700-704,361-362
 *   K-mix keying   build-added: splitmix64(k) split the same hi/lo way, so
 *                  alternates and eviction chains are exercised
 *   block binning  region of the bucket (libgpuhash/test/insert_test.c:117-124)
 *                  or hash >> (32-IBLOCK_P)        (src/mega_recv.c:476-477)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct mkv_zipf_s {
	uint64_t n;
	double theta;
	double alpha;
	double thres;
	uint64_t last_n;
	double dbl_n;
	double zetan;
	double eta;
	uint64_t rand_state;
} mkv_zipf_t;

/* src/zipf.h:40-70 */
static inline double pow_approx(double a, double b)
{
	int e = (int)b;
	union { double d; int x[2]; } u = { a };
	u.x[1] = (int)((b - (double)e) * (double)(u.x[1] - 1072632447) + 1072632447.);
	u.x[0] = 0;
	double r = 1.;
	while (e) {
		if (e & 1)
			r *= a;
		a *= a;
		e >>= 1;
	}
	return r * u.d;
}

/* src/zipf.h:110-123 */
static double zeta(uint64_t last_n, double last_sum, uint64_t n, double theta)
{
	if (last_n > n) {
		last_n = 0;
		last_sum = 0.;
	}
	while (last_n < n) {
		last_sum += 1. / pow_approx((double)last_n + 1., theta);
		last_n++;
	}
	return last_sum;
}

/* src/zipf.h:125-132 */
static inline double rand_d(uint64_t *state)
{
	*state = (*state * 0x5deece66dUL + 0xbUL) & ((1UL << 48) - 1);
	return (double)*state / (double)((1UL << 48) - 1);
}

/* src/zipf.h:72-108; returns 0, or -1 for a theta the generator rejects */
int mkv_zipf_init(mkv_zipf_t *s, uint64_t n, double theta, uint64_t seed)
{
	if (n == 0 || seed >= (1UL << 48))
		return -1;
	if (!(theta == -1. || (theta >= 0. && theta < 1.) || theta >= 40.))
		return -1;
	memset(s, 0, sizeof(*s));
	s->n = n;
	s->theta = theta;
	if (theta == -1.)
		seed = seed % n;
	else if (theta > 0. && theta < 1.) {
		s->alpha = 1. / (1. - theta);
		s->thres = 1. + pow_approx(0.5, theta);
	}
	s->rand_state = seed;
	return 0;
}

/* src/zipf.h:134-185 */
uint64_t mkv_zipf_next(mkv_zipf_t *s)
{
	if (s->last_n != s->n) {
		if (s->theta > 0. && s->theta < 1.) {
			s->zetan = zeta(s->last_n, s->zetan, s->n, s->theta);
			s->eta = (1. - pow_approx(2. / (double)s->n, 1. - s->theta)) /
				(1. - zeta(0, 0., 2, s->theta) / s->zetan);
		}
		s->last_n = s->n;
		s->dbl_n = (double)s->n;
	}
	if (s->theta == -1.) {
		uint64_t v = s->rand_state;
		if (++s->rand_state >= s->n)
			s->rand_state = 0;
		return v;
	} else if (s->theta == 0.) {
		double u = rand_d(&s->rand_state);
		return (uint64_t)(s->dbl_n * u);
	} else if (s->theta >= 40.) {
		return 0UL;
	} else {
		double u = rand_d(&s->rand_state);
		double uz = u * s->zetan;
		if (uz < 1.)
			return 0UL;
		else if (uz < s->thres)
			return 1UL;
		else
			return (uint64_t)(s->dbl_n * pow_approx(s->eta * (u - 1.) + 1., s->alpha));
	}
}

void mkv_zipf_params(const mkv_zipf_t *s, double out[3])
{
	out[0] = s->zetan;
	out[1] = s->eta;
	out[2] = s->thres;
}

/* GET keys: rank + 1 (src/mega_recv.c:691) */
void mkv_zipf_keys(mkv_zipf_t *s, uint32_t *keys, int64_t count)
{
	for (int64_t i = 0; i < count; i++)
		keys[i] = (uint32_t)mkv_zipf_next(s) + 1;
}

#define MKV_KEYING_REF 0
#define MKV_KEYING_MIX 1

static inline uint64_t splitmix64(uint64_t x)
{
	x += 0x9e3779b97f4a7c15ULL;
	x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
	x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
	return x ^ (x >> 31);
}

static inline void key_to_sig_hash(uint32_t k, int keying, uint32_t iblock_p,
		uint32_t *sig, uint32_t *hash)
{
	if (keying == MKV_KEYING_REF) {
		/* high word = (bswap32(k & 0xff) << (8 - bits)) | k, low word = k
		 * (src/mega_recv.c:700-704); hash = high, sig = low (:361-362) */
		uint32_t hi = k;
		if (iblock_p)
			hi = (uint32_t)(((k & 0xffu) << 24) << (8 - iblock_p)) | k;
		*sig = k;
		*hash = hi;
	} else {
		uint64_t z = splitmix64(k);
		uint32_t s = (uint32_t)z;
		*sig = s ? s : 1;	/* 0 marks an empty slot (src/mega_common.c:58-59) */
		*hash = (uint32_t)(z >> 32);
	}
}

void mkv_make_selem(const uint32_t *keys, int64_t n, int keying, uint32_t iblock_p,
		uint32_t *selem /* [n][2] */)
{
	for (int64_t i = 0; i < n; i++)
		key_to_sig_hash(keys[i], keying, iblock_p, &selem[2 * i], &selem[2 * i + 1]);
}

void mkv_make_ielem(const uint32_t *keys, const uint32_t *locs, int64_t n, int keying,
		uint32_t iblock_p, uint32_t *ielem /* [n][3] */)
{
	for (int64_t i = 0; i < n; i++) {
		key_to_sig_hash(keys[i], keying, iblock_p, &ielem[3 * i], &ielem[3 * i + 1]);
		ielem[3 * i + 2] = locs ? locs[i] : keys[i];
	}
}

#define MKV_BIN_REGION 0	/* libgpuhash/test/insert_test.c:117-124 */
#define MKV_BIN_APP    1	/* src/mega_recv.c:476-477 */

/* Stable partition of insert requests into 2^iblock_p blocks.  out is [n][3],
 * counts is [2^iblock_p]; block b occupies out[offset_b .. offset_b+counts[b]). */
void mkv_bin_inserts(const uint32_t *ielem, int64_t n, uint32_t mem_p, uint32_t iblock_p,
		int convention, uint32_t *out, int64_t *counts)
{
	uint32_t nblk = 1u << iblock_p;
	uint32_t hash_mask = (uint32_t)(((uint64_t)1 << (mem_p - 6)) - 1);
	uint32_t shift = mem_p - 6 - iblock_p;
	int64_t *off = (int64_t *)calloc(nblk + 1, sizeof(int64_t));
	for (uint32_t b = 0; b < nblk; b++)
		counts[b] = 0;
	for (int64_t i = 0; i < n; i++) {
		uint32_t h = ielem[3 * i + 1];
		uint32_t b = convention == MKV_BIN_REGION ? (h & hash_mask) >> shift
			: (iblock_p ? h >> (32 - iblock_p) : 0);
		counts[b]++;
	}
	for (uint32_t b = 0; b < nblk; b++)
		off[b + 1] = off[b] + counts[b];
	for (int64_t i = 0; i < n; i++) {
		uint32_t h = ielem[3 * i + 1];
		uint32_t b = convention == MKV_BIN_REGION ? (h & hash_mask) >> shift
			: (iblock_p ? h >> (32 - iblock_p) : 0);
		memcpy(&out[3 * off[b]], &ielem[3 * i], 12);
		off[b]++;
	}
	free(off);
}

/* ---- LOCAL_TEST packet synthesis ---------------------------------
 * The reference's receiver builds its own request packets when compiled with LOCAL_TEST
 * and feeds them to its parser (src/mega_recv.c:633-768).  This restates the payload it
 * writes -- 16-bit magic 0x1234, jobs, 16-bit end mark 0xFFFF (README.md:70-81):
 *   GET: u16 0x2, u16 key_len, key            SET: u16 0x3, u16 key_len, u32 value_len, key, value
 * Keys: first 4 bytes = key id, next 4 = (bswap32(id & 0xff) << (8 - bits_insert_buf)) | id
 * (:700-704, :733-737); the rest of a longer key and the value are not written by the
 * reference (stale buffer contents) and are zero here.  GET ids come from the Zipf stream
 * (rank + 1, :691), SET ids count up (:727); NUM_DEFINED_GET GETs alternate with
 * NUM_DEFINED_SET SETs across packet borders (:681-749); a packet is closed when the next
 * job would push the frame beyond 1514 bytes including the 42-byte Ethernet/IP/UDP header. */
typedef struct mkv_synth_s {
	mkv_zipf_t zipf;
	uint32_t set_key;          /* last SET id handed out (the next one is +1) */
	int num_get, num_set;      /* jobs left in the current run of GETs / SETs */
	int def_get, def_set;      /* NUM_DEFINED_GET / NUM_DEFINED_SET (src/macros.h:88-97) */
	int key_len, value_len;    /* KEY_LEN / VALUE_LEN (src/macros.h:58-74) */
	int bits_insert_buf;       /* IBLOCK_P */
} mkv_synth_t;

int mkv_synth_init(mkv_synth_t *s, uint64_t zipf_n, double theta, uint64_t seed, uint32_t first_set_key,
		int def_get, int def_set, int key_len, int value_len, int bits_insert_buf)
{
	if (key_len < 8 || value_len < 0 || def_get < 0 || def_set < 0 || def_get + def_set == 0 ||
			bits_insert_buf < 0 || bits_insert_buf > 8)
		return -1;
	if (mkv_zipf_init(&s->zipf, zipf_n, theta, seed) != 0)
		return -1;
	s->set_key = first_set_key - 1u;
	s->def_get = def_get;
	s->def_set = def_set;
	s->num_get = def_get;              /* :645-646 */
	s->num_set = def_set;
	s->key_len = key_len;
	s->value_len = value_len;
	s->bits_insert_buf = bits_insert_buf;
	return 0;
}

static void synth_key(uint8_t *p, uint32_t id, int key_len, int bits)
{
	uint32_t hi = id;
	if (bits)
		hi = (uint32_t)(((id & 0xffu) << 24) << (8 - bits)) | id;
	memset(p, 0, (size_t)key_len);
	memcpy(p, &id, 4);
	memcpy(p + 4, &hi, 4);
}

/* Writes npackets payloads of at most `stride` bytes each (stride >= 1472); payload_len[p] =
 * bytes of packet p (magic + jobs + end mark).  get_keys / set_keys, if not NULL, receive the
 * ids in job order (capacity: the caller's; counts are returned through n_get / n_set). */
int64_t mkv_synth_packets(mkv_synth_t *s, uint8_t *payloads, int64_t stride, int64_t npackets,
		int32_t *payload_len, uint32_t *get_keys, uint32_t *set_keys, int64_t *n_get, int64_t *n_set)
{
	const int hdr = 42, frame_max = 1514;
	const int get_len = s->key_len + 4, set_len = s->key_len + s->value_len + 8;
	int64_t ng = 0, ns = 0;
	if (stride < frame_max - hdr)
		return -1;
	for (int64_t p = 0; p < npackets; p++) {
		uint8_t *base = payloads + p * stride, *ptr = base;
		int frame = hdr + 2 + 2;                   /* reserved_len (:639) */
		const uint16_t magic = 0x1234, end = 0xFFFF, jget = 0x2, jset = 0x3;
		memcpy(ptr, &magic, 2);
		ptr += 2;
		for (;;) {
			if (s->num_get > 0) {
				if (frame + get_len > frame_max)
					break;
				const uint16_t kl = (uint16_t)s->key_len;
				memcpy(ptr, &jget, 2);
				memcpy(ptr + 2, &kl, 2);
				ptr += 4;
				const uint32_t id = (uint32_t)mkv_zipf_next(&s->zipf) + 1u;
				synth_key(ptr, id, s->key_len, s->bits_insert_buf);
				if (get_keys)
					get_keys[ng] = id;
				ng++;
				ptr += s->key_len;
				frame += get_len;
				if (--s->num_get == 0) {
					if (s->def_set > 0)
						s->num_set = s->def_set;
					else
						s->num_get = s->def_get;
				}
			} else if (s->num_set > 0) {
				if (frame + set_len > frame_max)
					break;
				const uint16_t kl = (uint16_t)s->key_len;
				const uint32_t vl = (uint32_t)s->value_len;
				memcpy(ptr, &jset, 2);
				memcpy(ptr + 2, &kl, 2);
				memcpy(ptr + 4, &vl, 4);
				ptr += 8;
				const uint32_t id = ++s->set_key;
				synth_key(ptr, id, s->key_len, s->bits_insert_buf);
				memset(ptr + s->key_len, 0, (size_t)s->value_len);
				if (set_keys)
					set_keys[ns] = id;
				ns++;
				ptr += s->key_len + s->value_len;
				frame += set_len;
				if (--s->num_set == 0) {
					if (s->def_get > 0)
						s->num_get = s->def_get;
					else
						s->num_set = s->def_set;
				}
			} else {
				return -1;
			}
		}
		memcpy(ptr, &end, 2);
		ptr += 2;
		if (payload_len)
			payload_len[p] = (int32_t)(ptr - base);
	}
	if (n_get)
		*n_get = ng;
	if (n_set)
		*n_set = ns;
	return ng + ns;
}
