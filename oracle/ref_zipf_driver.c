/*
 * ref_zipf_driver.c -- prints known answers of the REFERENCE's own Zipf
 * generator (src/zipf.h, included where it lies under /root/reference; never
 * copied).  TEST INFRASTRUCTURE: built into oracle/_ref/ by `make ref` when the
 * reference tree is present, used by tests/golden/make_golden.py to produce
 * tests/golden/zipf_kat.json and by tests/test_workload.py to cross-check the
 * restatement in megakv_amd/csrc/workload.c.
 *
 * usage: ref_zipf N THETA SEED COUNT  -> COUNT ranks, one per line, then
 *        "zetan eta thres"
 */
#define LOCAL_TEST 1
#include <assert.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zipf.h"

int main(int argc, char **argv)
{
	if (argc != 5)
		return 2;
	uint64_t n = strtoull(argv[1], NULL, 10);
	double theta = atof(argv[2]);
	uint64_t seed = strtoull(argv[3], NULL, 10);
	long count = atol(argv[4]);
	struct zipf_gen_state st;
	mehcached_zipf_init(&st, n, theta, seed);
	for (long i = 0; i < count; i++)
		printf("%llu\n", (unsigned long long)mehcached_zipf_next(&st));
	printf("%.17g %.17g %.17g\n", st.zetan, st.eta, st.thres);
	return 0;
}
