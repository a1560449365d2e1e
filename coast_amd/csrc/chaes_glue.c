/*
 * chaes_glue.c -- the strong encrypt / decrypt for the CHStone aes benchmark (tests/chstone/aes/aes_enc.c:67-134,
 * aes_dec.c:66-140).  Both reference functions do three things: run the cipher on `statemt`, print the block as hex, and
 * add the number of bytes that differ from a built-in test vector to the program's `main_result`.  The cipher runs protected
 * on the GPU (coast_dropin_chstone_aes); the printing and the check are replayed here so that the unmodified aes.c prints
 * what it always prints.  The vectors are FIPS-197 Appendix B (aes_enc.c:77-80, aes_dec.c:76-79).
 * Compiled against the reference's header where it lies:   gcc -I<reference>/tests/chstone/aes -c chaes_glue.c
 */
#include <stdio.h>

#include "aes.h"

extern int main_result;
int coast_dropin_chstone_aes(int *statemt, const int *key, int type, int dir);

int encrypt(int statemt[32], int key[32], int type)
{
    static const int out_enc_statemt[16] = {0x39, 0x25, 0x84, 0x1d, 0x02, 0xdc, 0x09, 0xfb,
                                            0xdc, 0x11, 0x85, 0x97, 0x19, 0x6a, 0x0b, 0x32};
    int i;
    if (coast_dropin_chstone_aes(statemt, key, type, 0))
        return -1;
    nb = (type % 1000) / 32; /* the globals encrypt leaves behind (aes_enc.c:85-111): Nb, and Nr - 10 */
    round_val = (type / 1000 > type % 1000 ? type / 1000 : type % 1000) / 32 - 4;
    printf("encrypted message \t");
    for (i = 0; i < nb * 4; ++i) {
        if (statemt[i] < 16)
            printf("0");
        printf("%x", statemt[i]);
    }
    for (i = 0; i < 16; i++)
        main_result += (statemt[i] != out_enc_statemt[i]);
    return 0;
}

int decrypt(int statemt[32], int key[32], int type)
{
    static const int out_dec_statemt[16] = {0x32, 0x43, 0xf6, 0xa8, 0x88, 0x5a, 0x30, 0x8d,
                                            0x31, 0x31, 0x98, 0xa2, 0xe0, 0x37, 0x07, 0x34};
    int i;
    if (coast_dropin_chstone_aes(statemt, key, type, 1))
        return -1;
    nb = (type % 1000) / 32; /* aes_dec.c:83-113: Nb, and Nr */
    round_val = (type / 1000 > type % 1000 ? type / 1000 : type % 1000) / 32 + 6;
    printf("\ndecrypto message\t");
    for (i = 0; i < ((type % 1000) / 8); ++i) {
        if (statemt[i] < 16)
            printf("0");
        printf("%x", statemt[i]);
    }
    for (i = 0; i < 16; i++)
        main_result += (statemt[i] != out_dec_statemt[i]);
    return 0;
}
