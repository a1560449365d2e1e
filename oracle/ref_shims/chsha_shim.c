/*
 * chsha_shim.c -- builds the REFERENCE CHStone sha (tests/chstone/sha/sha.c + sha_data.c) from the sources where they lie
 * under /root/reference.  sha.c defines its own memset / memcpy with non-libc signatures and sha.h defines its globals in
 * the header, so everything is renamed into a ref_chsha_ namespace and pulled into this one TU.
 * Exports ref_chsha(data, len, digest) = sha_init + sha_update + sha_final, ref_chsha_stream(digest) = the benchmark's own
 * sha_stream over its built-in 2 x 8192-byte vectors, and the vectors themselves.  Test infrastructure only.
 */
#define memset ref_chsha_memset
#define memcpy ref_chsha_memcpy
#define sha_info_digest ref_chsha_info_digest
#define sha_info_count_lo ref_chsha_info_count_lo
#define sha_info_count_hi ref_chsha_info_count_hi
#define sha_info_data ref_chsha_info_data
#define sha_init ref_chsha_init
#define sha_update ref_chsha_update
#define sha_final ref_chsha_final
#define sha_stream ref_chsha_sha_stream
#define indata ref_chsha_indata
#define in_i ref_chsha_in_i

#include "chstone/sha/sha.c"
#include "chstone/sha/sha_data.c"

void ref_chsha(const unsigned char *data, int len, unsigned int digest[5])
{
    sha_init();
    sha_update(data, len);
    sha_final();
    for (int i = 0; i < 5; ++i)
        digest[i] = sha_info_digest[i];
}

void ref_chsha_stream(unsigned int digest[5])
{
    sha_stream();
    for (int i = 0; i < 5; ++i)
        digest[i] = sha_info_digest[i];
}
