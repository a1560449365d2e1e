/*
 * chsha_glue.c -- the strong sha_stream for the CHStone sha benchmark.  Its input lives in globals whose shape is fixed by
 * macros of the reference's own header (tests/chstone/sha/sha.h:59-60: BLOCK_SIZE, VSIZE; indata, in_i, sha_info_digest), so
 * this TU is compiled against that header where it lies:   gcc -fcommon -I<reference>/tests/chstone/sha -c chsha_glue.c
 */
#include "sha.h"

void coast_dropin_sha_stream(const unsigned char *indata, const int *in_i, int vsize, int block_size, unsigned int *digest);

void sha_stream(void)
{
    coast_dropin_sha_stream(&indata[0][0], in_i, VSIZE, BLOCK_SIZE, sha_info_digest);
}
