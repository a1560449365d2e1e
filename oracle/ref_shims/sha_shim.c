/*
 * sha_shim.c -- builds the REFERENCE sha256_hash (tests/sha256_common/sha256_common_tmr.c) from the source where
 * it lies under /root/reference; supplies the data symbols its includer normally provides (sha_data.inc).
 * Exports ref_sha256_hash() with the reference signature.  Test infrastructure only.
 */
#include <stdint.h>
#include "COAST.h" /* the reference's tests/COAST.h (-I) */

#define LEN 1
#define hash_data ref_sha_hash_data
#define golden ref_sha_golden
#define k ref_sha_k
#define data ref_sha_data
#define state ref_sha_state
#define bitlen ref_sha_bitlen
#define hashGlbl ref_sha_hashGlbl
#define sha256_transform ref_sha256_transform
#define sha256_hash ref_sha256_hash
#define sha_run_test ref_sha_run_test
#define checkGolden ref_sha_checkGolden

uint8_t hash_data[LEN];
uint8_t golden[32];

#include "sha256_common/sha256_common_tmr.c"
