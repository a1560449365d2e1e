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

enum { SITE_CHAES_STATE = 64, SITE_CHAES_WORD = 65, SITE_CHAES_RND = 66, SITE_CHAES_J = 67, SITE_CHAES_I = 68 };
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

// CHStone aes with its loops, switches and array indices as written, for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the walk follows the
// -O0 IR of the benchmark's files statement by statement (KeySchedule aes_key.c:77-165, encrypt aes_enc.c:67-119, decrypt
// aes_dec.c:66-118, SubByte / ByteSub_ShiftRow / InversShiftRow_ByteSub / MixColumn_AddRoundKey / AddRoundKey_InversMixColumn /
// AddRoundKey aes_func.c:135-545) and adds to the frozen schedule (the Nb packed columns of the result block, voted as stored data)
//   COAST_F_BRANCH_SYNC  every evaluated conditional branch -- loop conditions, the `if ((x >> 8) == 1)` tests of the two MixColumn
//                        functions, `(j % nk) == 0`, `(j % nk) != 0`, `nk > 6 && j % nk == 4` in short-circuit order --, every switch
//                        (on type: KeySchedule, encrypt / decrypt, AddRoundKey; on nb: the two ShiftRow functions) and every return of
//                        a computed value (SubByte's; KeySchedule's, through %retval)                 synchronization.cpp:741-949
//   COAST_F_ADDR_SYNC    every GEP whose last index is not a constant: both levels of word[i][j] and of Sbox[a][b], the one level of
//                        word[1][j - 1], key[i + j * 4], statemt[..], ret[..], temp[i], Rcon0[..]; `statemt[k] ^= w` loads first, so
//                        its address is a LOAD address (:341-367)                                                       :413-474
//   COAST_F_LOCAL_STORE_SYNC (with the two) the data of every store of a computed value, in place and into the locals' allocas  :197-224, 476-561
// 3 971 / 6 722 sync points per 128-bit encryption / decryption (5 853 / 12 367 with the stores), 11 407 / 19 328 with 256-bit key and block -- the counts of
// tools/ir_sync_counts.py `chaes` on the reference's IR (tests/test_ir_counts_cpu.py).  The region ends where encrypt / decrypt start
// printing.  The locals (the round counter, the callees' j and i, x, temp[4], ret[32]) and statemt[] / word[][] are replica-private: the
// lane's own strips of LDS, because they are indexed at run time; a voted (or, unvoted, replica 0's) offset selects the element every
// copy accesses.  word[][] is kept packed (byte `row` of column dword j); statemt[], ret[], temp[] are ints.  Fault sites:
// SITE_CHAES_RND / _J / _I of a replica (32 bits live), `step` = how many LOOP conditions the call has evaluated; SITE_CHAES_STATE before
// the `step`-th (Invers)ShiftRow call (0: entry, Nr + 1: exit); SITE_CHAES_WORD after the `step`-th key-schedule column.  A wild index
// reads 0 / stores nothing; a walk that a corrupted counter keeps alive is cut after 8192 loop conditions.  Oracle: chaes_item_indexed
// (oracle/chaes_indexed.inc).  The sync-point-parity form of the kernel, not the throughput form.
constexpr uint32_t kChaesWalkCap = 8192u;
template <int NREP>
__global__ __launch_bounds__(64) void chaes_indexed_kernel(uint8_t *__restrict__ states, const uint8_t *__restrict__ keys,
                                                           uint64_t nblocks, int type, int nk, int nb, int nr, int dirFlag, Counters ctr,
                                                           FaultTab ft, uint8_t *__restrict__ detected)
{
    __shared__ __attribute__((aligned(16))) uint8_t sSb[256];
    __shared__ __attribute__((aligned(16))) uint8_t sRsb[256];
    __shared__ uint32_t sW[kChaesMaxCols][64];
    __shared__ int32_t sSt[32][64];
    __shared__ int32_t sRet[32][64];
    __shared__ int32_t sOut[32][64]; // the ShiftRow functions' result on its way back into statemt[] (ret[] keeps its stale content)
    __shared__ int32_t sTemp[4][64];
    __shared__ int32_t sRc[32];
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    // COAST_F_LOCAL_STORE_SYNC: the data of every store of a computed value of the -O0 IR -- the counters' ++ / --, x, the parameters'
    // spills into their allocas, and every int stored into statemt[] / ret[] / temp[] / word[][] in place (1170 + 712 votes per 128/128
    // encryption, 1392 + 4253 per decryption: tools/ir_sync_counts.py chaes)
    const bool lss = xmr_local_sync_on(ctr.flags);
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
    if (lane == 0) { // Rcon0[30] (aes_key.c:64-73): successive doublings in GF(2^8)
        int32_t v = 1;
        for (int k = 0; k < 30; ++k) {
            sRc[k] = v;
            v = ((v << 1) ^ ((v & 0x80) ? 0x11b : 0)) & 0xff;
        }
    }
    const uint64_t it = live ? item : 0;
    const uint8_t *kp = keys + it * (uint64_t)(4 * nk);
    uint8_t *sp = states + it * (uint64_t)(4 * nb);
    for (int k = 0; k < 32; ++k)
        sSt[k][lane] = k < 4 * nb ? (int32_t)sp[k] : 0, sRet[k][lane] = 0;
    for (int k = 0; k < kChaesMaxCols; ++k)
        sW[k][lane] = 0u;
    for (int k = 0; k < 4; ++k)
        sTemp[k][lane] = 0;
    wave_lds_sync();

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    Tally tl;
    int32_t rnd = 0, i = 0, j = 0;
    uint32_t tick = 0u, nsub = 0u, ncol = 0u;
    enum { LT, LE, GE };
    if (lm.live) { // (the idle lane of a TMR wave has no block of its own: its replica group would wrap to lanes 0, 1)
        auto loopc = [&](const int32_t &reg, int32_t lim, int cmp) { // one evaluated loop condition
            for (uint32_t q = 0; q < fr.y; ++q) { // the counters' upsets land right before the condition reads them
                const DevFault df = ft.list[fr.x + q];
                if (df.step != tick || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const int32_t m = (int32_t)(1u << (df.bit & 31u));
                if (df.site == SITE_CHAES_RND)
                    rnd ^= m;
                else if (df.site == SITE_CHAES_J)
                    j ^= m;
                else if (df.site == SITE_CHAES_I)
                    i ^= m;
            }
            if (tick >= kChaesWalkCap)
                return false;
            ++tick;
            const bool c = cmp == LT ? reg < lim : cmp == LE ? reg <= lim : reg >= lim;
            return xmr_steer<NREP>(c ? 1u : 0u, lm, bs, cnt, tl) != 0u;
        };
        auto ifc = [&](bool c) { return xmr_steer<NREP>(c ? 1u : 0u, lm, bs, cnt, tl) != 0u; };
        auto sw = [&](int32_t v) { (void)xmr_steer<NREP>((uint32_t)v, lm, bs, cnt, tl); };       // a switch votes its operand (:761-767)
        auto retv = [&](int32_t v) { return bs ? (int32_t)xmr_sync<NREP>((uint32_t)v, lm, cnt, tl) : v; }; // `ret` of a computed value
        auto lsy = [&](int32_t v) { return (int32_t)xmr_local_sync<NREP>((uint32_t)v, lm, lss, cnt, tl); }; // the data of a store
        auto off = [&](int32_t idx, bool store) { return xmr_steer<NREP>((uint32_t)idx, lm, store ? ss : ls, cnt, tl); };
        auto ld = [&](int32_t (*arr)[64], int32_t idx) -> int32_t {
            const uint32_t o = off(idx, false);
            return o < 32u ? arr[o][lane] : 0;
        };
        auto stv = [&](int32_t (*arr)[64], int32_t idx, int32_t v) {
            const uint32_t o = off(idx, true);
            const int32_t d = lsy(v);
            if (o < 32u)
                arr[o][lane] = d;
        };
        auto xr = [&](int32_t (*arr)[64], int32_t idx, int32_t v) { // arr[idx] ^= v: a LOAD address
            const uint32_t o = off(idx, false);
            const int32_t d = lsy((o < 32u ? arr[o][lane] : 0) ^ v);
            if (o < 32u)
                arr[o][lane] = d;
        };
        auto wget = [&](uint32_t f) -> int32_t { return f < 480u ? (int32_t)((sW[f % 120u][lane] >> (8u * (f / 120u))) & 0xffu) : 0; };
        auto wput = [&](uint32_t f, int32_t v) {
            if (f < 480u) {
                const uint32_t sh = 8u * (f / 120u);
                uint32_t &w = sW[f % 120u][lane];
                w = (w & ~(0xffu << sh)) | (((uint32_t)v & 0xffu) << sh);
            }
        };
        auto ldw = [&](int row, int32_t idx) -> int32_t { return wget((uint32_t)row * 120u + off(idx, false)); }; // word[row][idx], constant row
        auto box = [&](const uint8_t *tab, int32_t a, int32_t b) -> int32_t { // Sbox[a][b] / invSbox[a][b]: two voted levels
            const uint32_t o1 = off(a, false), o2 = off(b, false), f = o1 * 16u + o2;
            return f < 256u ? (int32_t)tab[f] : 0;
        };
        auto subbyte = [&](int32_t in) -> int32_t { // aes_func.c:248-252
            in = lsy(in); // the parameter into its alloca
            return retv(box(sSb, in / 16, in % 16));
        };
        auto wordHook = [&]() {
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.site == SITE_CHAES_WORD && df.step == ncol && (int)df.local == slot && (int)df.replica == lm.r && ncol < 120u)
                    sW[ncol][lane] ^= 1u << (df.bit & 31u);
            }
            ++ncol;
        };
        auto stateHook = [&](uint32_t step) {
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.site == SITE_CHAES_STATE && df.step == step && (int)df.local == slot && (int)df.replica == lm.r)
                    sSt[4 * (df.index & 7u) + ((df.bit & 31u) >> 3)][lane] ^= (int32_t)(1u << (df.bit & 7u));
            }
        };
        auto arkLoop = [&](int32_t n) { // statemt[c + j*4] ^= word[c][j + nb*n], c = 0..3, every column j (aes_func.c:535-542, :443-449)
            for (j = 0; loopc(j, nb, LT); j = lsy((int32_t)((uint32_t)j + 1u)))
                for (int cc = 0; cc < 4; ++cc) {
                    const int32_t w = ldw(cc, (int32_t)((uint32_t)j + (uint32_t)nb * (uint32_t)n));
                    xr(sSt, (int32_t)((uint32_t)cc + (uint32_t)j * 4u), w);
                }
        };
        auto addRoundKey = [&](int32_t n) { // aes_func.c:514-545
            (void)lsy(type), n = lsy(n); // the parameters type and n into their allocas
            sw(type);
            arkLoop(n);
        };
        auto subShift = [&](bool inverse) { // aes_func.c:135-246, :254-366: 4 nb lookups, statemt's own indices are constants
            // every looked-up byte is stored once -- into statemt[] or into `temp` --, and the bytes that went through `temp` a second time
            // (statemt[k] = temp): one per cycle of the row's rotation, 1 / 2 / 1 cycles in rows 1..3 for Nb = 4, 1 / 2 / 3 for 6, 1 / 1 / 4 for 8
            (void)lsy(nb); // the parameter nb into its alloca
            sw(nb);
            stateHook(++nsub);
            const uint8_t *tab = inverse ? sRsb : sSb;
            for (int jj = 0; jj < nb; ++jj)
                for (int ii = 0; ii < 4; ++ii) {
                    const int sh = nb == 8 ? (ii == 0 ? 0 : ii == 1 ? 1 : ii == 2 ? 3 : 4) : ii;
                    const int from = inverse ? (jj - sh + nb) % nb : (jj + sh) % nb;
                    const int32_t v = sSt[ii + 4 * from][lane];
                    const int cyc = ii == 0 ? 0 : nb == 4 ? (ii == 2 ? 2 : 1) : nb == 6 ? ii : (ii == 3 ? 4 : 1);
                    int32_t b = lsy(box(tab, v >> 4, v & 0xf));
                    if (jj < cyc)
                        b = lsy(b);
                    sOut[ii + 4 * jj][lane] = b;
                }
            for (int k = 0; k < 4 * nb; ++k)
                sSt[k][lane] = sOut[k][lane];
        };
        auto reduce = [&](int32_t x) -> int32_t { return ifc((x >> 8) == 1) ? lsy(x ^ 283) : x; }; // if ((x >> 8) == 1) x ^= 283;
        auto mixColArk = [&](int32_t n) { // MixColumn_AddRoundKey, aes_func.c:368-432
            (void)lsy(nb), n = lsy(n); // the parameters nb and n into their allocas
            for (j = 0; loopc(j, nb, LT); j = lsy((int32_t)((uint32_t)j + 1u)))
                for (int cc = 0; cc < 4; ++cc) {
                    const uint32_t j4 = (uint32_t)j * 4u;
                    const int32_t id = (int32_t)((uint32_t)cc + j4);
                    stv(sRet, id, (int32_t)((uint32_t)ld(sSt, id) << 1));             // ret[c + j*4] = statemt[c + j*4] << 1
                    if (ifc((ld(sRet, id) >> 8) == 1))                                  // if ((ret[..] >> 8) == 1) ret[..] ^= 283
                        xr(sRet, id, 283);
                    int32_t x = lsy(ld(sSt, (int32_t)((uint32_t)((cc + 1) & 3) + j4))); // x = statemt[(c+1)%4 + j*4]; x ^= x << 1
                    x = lsy(x ^ (int32_t)((uint32_t)x << 1));
                    if (ifc((x >> 8) == 1))                                             // if ((x >> 8) == 1) ret ^= x ^ 283; else ret ^= x
                        xr(sRet, id, x ^ 283);
                    else
                        xr(sRet, id, x);
                    const int32_t p = ld(sSt, (int32_t)((uint32_t)((cc + 2) & 3) + j4)); // ret ^= statemt[..] ^ statemt[..] ^ word[c][j + nb*n]
                    const int32_t q = ld(sSt, (int32_t)((uint32_t)((cc + 3) & 3) + j4));
                    const int32_t w = ldw(cc, (int32_t)((uint32_t)j + (uint32_t)nb * (uint32_t)n));
                    xr(sRet, id, p ^ q ^ w);
                }
            for (j = 0; loopc(j, nb, LT); j = lsy((int32_t)((uint32_t)j + 1u)))             // statemt[c + j*4] = ret[c + j*4]
                for (int cc = 0; cc < 4; ++cc) {
                    const int32_t id = (int32_t)((uint32_t)cc + (uint32_t)j * 4u);
                    stv(sSt, id, ld(sRet, id));
                }
        };
        auto arkInvMix = [&](int32_t n) { // AddRoundKey_InversMixColumn, aes_func.c:434-511
            (void)lsy(nb), n = lsy(n); // the parameters nb and n into their allocas
            arkLoop(n);
            for (j = 0; loopc(j, nb, LT); j = lsy((int32_t)((uint32_t)j + 1u)))
                for (i = 0; loopc(i, 4, LT); i = lsy((int32_t)((uint32_t)i + 1u))) {
                    const uint32_t j4 = (uint32_t)j * 4u;
                    const int32_t id = (int32_t)((uint32_t)i + j4);
                    auto lds = [&](int k) { return ld(sSt, (int32_t)((uint32_t)((int32_t)((uint32_t)i + (uint32_t)k) % 4) + j4)); };
                    auto shl = [&](int32_t x) { return reduce(lsy((int32_t)((uint32_t)x << 1))); };  // x = x << 1; if (..) x ^= 283
                    auto xk = [&](int32_t x, int k) { return lsy(x ^ lds(k)); };                    // x ^= statemt[(i + k) % 4 + j * 4]
                    int32_t x;
                    x = shl(lds(0)), x = xk(x, 0), x = shl(x), x = xk(x, 0), x = shl(x);     // 14 x                       :454-463
                    stv(sRet, id, x);
                    x = shl(lds(1)), x = shl(x), x = xk(x, 1), x = shl(x), x = xk(x, 1);     // 11 x                       :465-475
                    xr(sRet, id, x);
                    x = shl(lds(2)), x = xk(x, 2), x = shl(x), x = shl(x), x = xk(x, 2);     // 13 x                       :477-487
                    xr(sRet, id, x);
                    x = shl(lds(3)), x = shl(x), x = shl(x), x = xk(x, 3);                   //  9 x                       :489-498
                    xr(sRet, id, x);
                }
            for (i = 0; loopc(i, nb, LT); i = lsy((int32_t)((uint32_t)i + 1u)))             // statemt[c + i*4] = ret[c + i*4]    :503-509
                for (int cc = 0; cc < 4; ++cc) {
                    const int32_t id = (int32_t)((uint32_t)cc + (uint32_t)i * 4u);
                    stv(sSt, id, ld(sRet, id));
                }
        };

        // ---- KeySchedule, aes_key.c:77-165 ----
        (void)lsy(type);                                                               // (encrypt's / decrypt's parameter `type` into its alloca)
        (void)lsy(type);                                                               // (KeySchedule's)
        sw(type);                                                                      // switch (type)                       :83
        for (j = 0; loopc(j, nk, LT); j = lsy((int32_t)((uint32_t)j + 1u))) {               // for (j = 0; j < nk; ++j)            :135
            for (i = 0; loopc(i, 4, LT); i = lsy((int32_t)((uint32_t)i + 1u))) {            //   for (i = 0; i < 4; ++i)           :136
                const uint32_t ok = off((int32_t)((uint32_t)i + (uint32_t)j * 4u), false); // word[i][j] = key[i + j * 4]     :138
                const int32_t v = ok < 4u * (uint32_t)nk ? (int32_t)kp[ok] : 0;
                const uint32_t o1 = off(i, true), o2 = off(j, true);
                wput(o1 * 120u + o2, lsy(v));
            }
            wordHook();
        }
        (void)lsy(nk);                                                                 // j = nk: a loaded value              :141
        for (j = nk; loopc(j, nb * (nr + 1), LT); j = lsy((int32_t)((uint32_t)j + 1u))) {   // the expanded key                    :141
            const int32_t jm = j % nk, jm1 = (int32_t)((uint32_t)j - 1u);
            if (ifc(jm == 0)) {                                                        //   if ((j % nk) == 0): RotByte, SubByte :145-151
                for (int cc = 0; cc < 4; ++cc) {
                    int32_t s = subbyte(ldw((cc + 1) & 3, jm1));
                    if (cc == 0) {
                        const uint32_t o = off(j / nk - 1, false);                     //     ^ Rcon0[(j / nk) - 1]
                        s ^= o < 30u ? sRc[o] : 0;
                    }
                    sTemp[cc][lane] = lsy(s);
                }
            }
            if (ifc(jm != 0))                                                          //   if ((j % nk) != 0)                :152-158
                for (int cc = 0; cc < 4; ++cc)
                    sTemp[cc][lane] = lsy(ldw(cc, jm1));
            if (ifc(nk > 6)) {                                                         //   if (nk > 6 && j % nk == 4)        :159-161
                if (ifc(jm == 4))
                    for (i = 0; loopc(i, 4, LT); i = lsy((int32_t)((uint32_t)i + 1u))) {
                        const uint32_t ol = off(i, false);
                        const int32_t s = subbyte(ol < 4u ? sTemp[ol][lane] : 0);
                        const uint32_t os = off(i, true);
                        const int32_t sd = lsy(s);
                        if (os < 4u)
                            sTemp[os][lane] = sd;
                    }
            }
            for (i = 0; loopc(i, 4, LT); i = lsy((int32_t)((uint32_t)i + 1u))) {            //   word[i][j] = word[i][j - nk] ^ temp[i] :162-163
                const uint32_t a1 = off(i, false), a2 = off((int32_t)((uint32_t)j - (uint32_t)nk), false);
                const uint32_t ot = off(i, false);
                const int32_t v = wget(a1 * 120u + a2) ^ (ot < 4u ? sTemp[ot][lane] : 0);
                const uint32_t o1 = off(i, true), o2 = off(j, true);
                wput(o1 * 120u + o2, lsy(v));
            }
            wordHook();
        }
        (void)retv(0);                                                                 // return 0 (through %retval)          :164
        // ---- encrypt aes_enc.c:82-119 / decrypt aes_dec.c:81-127 ----
        sw(type);                                                                      // switch (type): round_val, nb
        stateHook(0u);
        if (!dir) {
            addRoundKey(0);                                                            // AddRoundKey (statemt, type, 0)      :112
            for (rnd = 1; loopc(rnd, nr - 1, LE); rnd = lsy((int32_t)((uint32_t)rnd + 1u))) { // i = 1 .. round_val + 9           :113-117
                subShift(false);
                mixColArk(rnd);
            }
            subShift(false);                                                           // ByteSub_ShiftRow; AddRoundKey (.., i) :118-119
            addRoundKey(rnd);
        } else {
            addRoundKey(nr);                                                           // AddRoundKey (statemt, type, round_val) :117
            subShift(true);                                                            // InversShiftRow_ByteSub               :119
            for (rnd = lsy(nr - 1); loopc(rnd, 1, GE); rnd = lsy((int32_t)((uint32_t)rnd - 1u))) { // i = round_val - 1 .. 1             :121-125
                arkInvMix(rnd);
                subShift(true);
            }
            addRoundKey(0);                                                            // AddRoundKey (statemt, type, 0)       :127
        }
        stateHook((uint32_t)nr + 1u);
    }
    // the frozen schedule's votes: the result block, one packed column per sync point
    uint32_t detItems = 0;
    for (int c = 0; c < nb; ++c) {
        uint32_t v = ((uint32_t)sSt[4 * c][lane] & 0xffu) | (((uint32_t)sSt[4 * c + 1][lane] & 0xffu) << 8) |
                     (((uint32_t)sSt[4 * c + 2][lane] & 0xffu) << 16) | (((uint32_t)sSt[4 * c + 3][lane] & 0xffu) << 24);
        v = xmr_store_sync<NREP>(v, lm, cnt, tl);
        if (cnt)
            reinterpret_cast<uint32_t *>(sp)[c] = v;
    }
    if (cnt && tl.det) {
        if (NREP == 2)
            detItems = 1;
        if (detected)
            detected[item] = 1;
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
