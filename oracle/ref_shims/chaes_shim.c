/*
 * chaes_shim.c -- builds the REFERENCE CHStone aes (tests/chstone/aes: aes_enc.c, aes_dec.c, aes_key.c, aes_func.c) from
 * the sources where they lie under /root/reference, as one callable per direction.  The four files share globals that the
 * benchmark's aes.c defines (type, nb, round_val, key, statemt, word, main_result): defined here instead; everything is renamed
 * into a ref_chaes_ namespace, and the hex dump encrypt / decrypt print is swallowed.  Test infrastructure only.
 */
#define type ref_chaes_type
#define nb ref_chaes_nb
#define round_val ref_chaes_round_val
#define key ref_chaes_key
#define statemt ref_chaes_statemt
#define word ref_chaes_word
#define main_result ref_chaes_main_result
#define KeySchedule ref_chaes_KeySchedule
#define SubByte ref_chaes_SubByte
#define ByteSub_ShiftRow ref_chaes_ByteSub_ShiftRow
#define InversShiftRow_ByteSub ref_chaes_InversShiftRow_ByteSub
#define MixColumn_AddRoundKey ref_chaes_MixColumn_AddRoundKey
#define AddRoundKey_InversMixColumn ref_chaes_AddRoundKey_InversMixColumn
#define AddRoundKey ref_chaes_AddRoundKey
#define encrypt ref_chaes_encrypt
#define decrypt ref_chaes_decrypt
#define Sbox ref_chaes_Sbox
#define invSbox ref_chaes_invSbox
#define Rcon0 ref_chaes_Rcon0
#define printf(...) ((void)0)

int ref_chaes_type, ref_chaes_nb, ref_chaes_round_val, ref_chaes_main_result;
int ref_chaes_key[32], ref_chaes_statemt[32], ref_chaes_word[4][120];

#include "aes_func.c"
#include "aes_key.c"
#include "aes_enc.c"
#include "aes_dec.c"

#undef type
#undef key
#undef statemt

/* state / key: one byte per element as in the benchmark, 4 Nb / 4 Nk of them; returns KeySchedule's verdict on `type` */
int ref_chaes(int *state, int *k, int type, int dir)
{
    if (ref_chaes_KeySchedule(type, k))
        return -1;
    if (dir)
        ref_chaes_decrypt(state, k, type);
    else
        ref_chaes_encrypt(state, k, type);
    return 0;
}
