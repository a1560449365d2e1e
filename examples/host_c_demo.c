/*
 * host_c_demo.c -- plain C host program on the MI355X backend, written against the reference's own call shapes:
 *   crc16(str, 13)                        tests/crc16/crc16.c:38-40      -> "result: 5ba3"
 *   sha256_hash(..., "abc", 3, hash)      tests/sha256_common/sha256_common_tmr.c:101
 *   aes_enc_dec(state, key, 0/1)          tests/aes/aes.c:91-97 (FIPS-197 appendix B vector)
 *   matrix_multiply(f, s, r) + XOR golden tests/mm_common/mm_common_tmr.c:3-32
 * and prints the campaign line `C: E: F: T:` (sha256_tmr.c:30) from TMR_ERROR_CNT.
 * Build: gcc -std=gnu11 -Dside=4 host_c_demo.c ../coast_amd/csrc/mm_glue.c ../coast_amd/lib/coast_dropin.o \
 *            -L../coast_amd/lib -lcoast_hip -Wl,-rpath,'$ORIGIN/../coast_amd/lib' -o host_c_demo
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

extern uint32_t TMR_ERROR_CNT;
extern uint64_t __SYNC_COUNT;
unsigned short crc16(const unsigned char *data_p, unsigned char length);
void aes_enc_dec(unsigned char *state, unsigned char *key, unsigned char dir);
void sha256_hash(unsigned char ctx_data[], uint32_t ctx_bitlen[], uint32_t ctx_state[], unsigned char data[],
                 uint32_t len, unsigned char hash[]);
void matrix_multiply(uint32_t f[][side], uint32_t s[][side], uint32_t r[][side]);

int main(void)
{
    int errors = 0;
    unsigned char str[] = "Automated TMR";
    const unsigned short crc = crc16(str, 13);
    printf("result: %hx\n", crc);
    errors += crc != 0x5ba3;

    unsigned char ctx_data[64], hash[32], msg[] = "abc";
    uint32_t bitlen[2], state[8];
    sha256_hash(ctx_data, bitlen, state, msg, 3, hash);
    static const unsigned char abc[32] = {0xba, 0x78, 0x16, 0xbf, 0x8f, 0x01, 0xcf, 0xea, 0x41, 0x41, 0x40,
                                          0xde, 0x5d, 0xae, 0x22, 0x23, 0xb0, 0x03, 0x61, 0xa3, 0x96, 0x17,
                                          0x7a, 0x9c, 0xb4, 0x10, 0xff, 0x61, 0xf2, 0x00, 0x15, 0xad};
    errors += memcmp(hash, abc, 32) != 0 || bitlen[0] != 24 || state[0] != 0xba7816bfu;

    unsigned char st[16] = {0x32, 0x43, 0xf6, 0xa8, 0x88, 0x5a, 0x30, 0x8d, 0x31, 0x31, 0x98, 0xa2, 0xe0, 0x37, 0x07, 0x34};
    unsigned char key[16] = {0x2b, 0x7e, 0x15, 0x16, 0x28, 0xae, 0xd2, 0xa6, 0xab, 0xf7, 0x15, 0x88, 0x09, 0xcf, 0x4f, 0x3c};
    unsigned char key2[16], pt[16];
    static const unsigned char ct[16] = {0x39, 0x25, 0x84, 0x1d, 0x02, 0xdc, 0x09, 0xfb, 0xdc, 0x11, 0x85, 0x97, 0x19, 0x6a, 0x0b, 0x32};
    memcpy(key2, key, 16);
    memcpy(pt, st, 16);
    aes_enc_dec(st, key, 0);
    errors += memcmp(st, ct, 16) != 0;
    aes_enc_dec(st, key2, 1);
    errors += memcmp(st, pt, 16) != 0;

    uint32_t f[side][side], s[side][side], r[side][side], x = 0, want = 0;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            f[i][j] = 0x9e3779b9u * (uint32_t)(i * side + j + 1);
            s[i][j] = 0x85ebca6bu ^ (uint32_t)(j * side + i);
        }
    matrix_multiply(f, s, r);
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            uint32_t acc = 0;
            for (int k = 0; k < side; ++k)
                acc += f[i][k] * s[k][j];
            want ^= acc;
            x ^= r[i][j];
        }
    errors += x != want;
    printf("C:0 E:%d F:%u T:0us\n", errors, TMR_ERROR_CNT);
    printf("syncs: %llu\n", (unsigned long long)__SYNC_COUNT);
    return errors;
}
