// chaes_kernel.hip -- CHStone aes (tests/chstone/aes; unittest/cfg/full.yml:6): Rijndael with 128 / 192 / 256-bit keys AND
// 128 / 192 / 256-bit blocks, the nine `type`s of KeySchedule's switch (aes_key.c:83-134), encrypt (aes_enc.c:67-134) and
// decrypt (aes_dec.c:66-140), over a batch of independent blocks.
//
// Work item = one block with its own key, run by a lane group (NREP adjacent lanes, one per replica).  The benchmark keeps one
// byte per int (statemt[32], word[4][120]); here a state column is one packed dword (row r in byte r, statemt[r + 4 c]) held in a
// register -- Nb of them, Nb a template parameter so that the ShiftRow gather is a fixed register permutation -- and the
// expanded key is the lane's own strip of LDS (120 packed columns, lane-interleaved: conflict-free).  Both are replica-private.
//   KeySchedule                  one column per step: W[j] = W[j-Nk] ^ f(W[j-1]); RotByte+SubByte+Rcon every Nk-th column, SubByte
//                                at j % Nk == 4 for Nk = 8 (aes_key.c:139-163); Rcon0[] = successive GF doublings (:64-73)
//   ByteSub_ShiftRow             row r of the state rotates by C_r columns, C = {0,1,2,3} (Nb = 4, 6), {0,1,3,4} (Nb = 8)
//                                (aes_func.c:137-262; inverse :271-396)
//   MixColumn_AddRoundKey        (2 3 1 1) circulant on a packed column + the round key (:399-432)
//   AddRoundKey_InversMixColumn  round key, then (14 11 13 9) (:435-512)
// Rounds Nr = max(Nk, Nb) + 6 (round_val + 9 main rounds + the last one, aes_enc.c:85-111; aes_dec.c:83-113).
//
// Sync points: the result block's stores into statemt, one vote per packed column; sync_every = 1 also votes the state after
// every round.  Injector hooks: a state column at a round boundary, an expanded-key column right after it was produced.
#include "xmr.hpp"

namespace coast {

enum { SITE_CHAES_STATE = 64, SITE_CHAES_WORD = 65 };
constexpr int kChaesMaxCols = 120; // word[4][120]: Nb (Nr + 1) <= 8 * 15

__device__ __forceinline__ uint32_t chaes_sub4(uint32_t w, const uint8_t *sb)
{
    return (uint32_t)sb[w & 0xffu] | ((uint32_t)sb[(w >> 8) & 0xffu] << 8) | ((uint32_t)sb[(w >> 16) & 0xffu] << 16) |
           ((uint32_t)sb[w >> 24] << 24);
}

template <int NB> struct ChaesShift {
    static constexpr int c(int row) { return NB == 8 ? (row == 0 ? 0 : row == 1 ? 1 : row == 2 ? 3 : 4) : row; }
};

// one wave (64-thread workgroup) per tile of IPW blocks
template <int NREP, int NB>
__global__ __launch_bounds__(64) void chaes_kernel(uint8_t *__restrict__ states, const uint8_t *__restrict__ keys, uint64_t nblocks,
                                                   int nk, int nr, int dirFlag, uint32_t syncEvery, Counters ctr, FaultTab ft,
                                                   uint8_t *__restrict__ detected)
{
    __shared__ __attribute__((aligned(16))) uint8_t sSb[256];
    __shared__ __attribute__((aligned(16))) uint8_t sRsb[256];
    __shared__ uint32_t sW[kChaesMaxCols][64];
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nblocks;
    const bool cnt = live && lm.r == 0;
    const bool dir = dirFlag != 0;
    reinterpret_cast<uint32_t *>(sSb)[lane] = reinterpret_cast<const uint32_t *>(gAesSbox)[lane];
    reinterpret_cast<uint32_t *>(sRsb)[lane] = reinterpret_cast<const uint32_t *>(gAesRsbox)[lane];
    if (lane < 4)
        sCnt[lane] = 0;
    wave_lds_sync();

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    Tally tl;
    const uint64_t it = live ? item : 0;
    const uint32_t *kp = reinterpret_cast<const uint32_t *>(keys) + it * (uint64_t)nk;
    uint32_t *sp = reinterpret_cast<uint32_t *>(states) + it * (uint64_t)NB;

    auto word_hook = [&](uint32_t w, uint32_t j) __attribute__((always_inline)) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.site == SITE_CHAES_WORD && df.step == j && (int)df.local == slot && (int)df.replica == lm.r && lm.live)
                w ^= 1u << (df.bit & 31u);
        }
        return w;
    };
    // ---- KeySchedule: the lane's own copy of word[][] ----
    const int ncols = NB * (nr + 1);
    for (int j = 0; j < nk; ++j) {
        uint32_t w = kp[j];
        if (fr.y)
            w = word_hook(w, (uint32_t)j);
        sW[j][lane] = w;
    }
    {
        uint32_t rcon = 1u, prev = sW[nk - 1][lane];
        int jm = 0; // j % nk
        for (int j = nk; j < ncols; ++j) {
            uint32_t t = prev;
            if (jm == 0) {
                t = chaes_sub4(__builtin_amdgcn_alignbit(t, t, 8), sSb) ^ rcon; // rows 1,2,3,0 of the previous column
                rcon = ((rcon << 1) ^ ((rcon & 0x80u) ? 0x11bu : 0u)) & 0xffu;
            } else if (nk > 6 && jm == 4)
                t = chaes_sub4(t, sSb);
            uint32_t w = sW[j - nk][lane] ^ t;
            if (fr.y)
                w = word_hook(w, (uint32_t)j);
            sW[j][lane] = w;
            prev = w;
            jm = jm + 1 == nk ? 0 : jm + 1;
        }
    }

    uint32_t s[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c)
        s[c] = sp[c];
    auto state_hook = [&](uint32_t step) __attribute__((always_inline)) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.site != SITE_CHAES_STATE || df.step != step || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            const uint32_t m = 1u << (df.bit & 31u);
#pragma unroll
            for (int c = 0; c < NB; ++c)
                if ((int)(df.index & 7u) == c)
                    s[c] ^= m;
        }
    };
    auto add_round_key = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
            s[c] ^= sW[NB * n + c][lane];
    };

    if (fr.y)
        state_hook(0u);
    add_round_key(dir ? nr : 0);
#pragma unroll 1
    for (int rd = 1; rd <= nr; ++rd) {
        if (fr.y)
            state_hook((uint32_t)rd);
        // (Invers)ShiftRow + ByteSub: column c takes row r from column c +- C_r
        uint32_t g[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            uint32_t v = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int from = dir ? (c - ChaesShift<NB>::c(r) + NB) % NB : (c + ChaesShift<NB>::c(r)) % NB;
                v |= s[from] & (0xffu << (8 * r));
            }
            g[c] = chaes_sub4(v, dir ? sRsb : sSb);
        }
        if (!dir) {
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                uint32_t v = g[c];
                if (rd < nr) { // MixColumn: 2 v ^ 3 rot(v) ^ rot2(v) ^ rot3(v)
                    const uint32_t r8 = __builtin_amdgcn_alignbit(v, v, 8);
                    v = aes_xtime4(v ^ r8) ^ aes_xor3(r8, __builtin_amdgcn_alignbit(v, v, 16), __builtin_amdgcn_alignbit(v, v, 24));
                }
                s[c] = v ^ sW[NB * rd + c][lane];
            }
        } else {
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const uint32_t v = g[c] ^ sW[NB * (nr - rd) + c][lane];
                s[c] = rd < nr ? aes_imc_col(v) : v;
            }
        }
        if (syncEvery && rd < nr) {
#pragma unroll
            for (int c = 0; c < NB; ++c)
                s[c] = xmr_store_sync<NREP>(s[c], lm, cnt, tl);
        }
    }
    if (fr.y)
        state_hook((uint32_t)nr + 1u);
#pragma unroll
    for (int c = 0; c < NB; ++c)
        s[c] = xmr_store_sync<NREP>(s[c], lm, cnt, tl); // statemt[i] = ...: store-data sync

    uint32_t detItems = 0;
    if (cnt) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
            sp[c] = s[c];
        if (tl.det) {
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
