// aes_kernel.hip -- protected aes_enc_dec (AES-128, one block, on-the-fly key schedule) for gfx950.
//
// Replaces: the DWC/TMR-transformed aes_enc_dec of tests/aes/TI_aes_128.c:107-231, including its data contract: the
// 16-byte state AND the 16-byte key are updated in place (encrypt leaves the last round key, decrypt first rolls the key
// forward 10 rounds, :110-123, then walks it back to the cipher key).
// Logical work item = one 16-byte block; lane NREP*q + r holds replica r of the wave's q-th block: state and running
// round key as 16 + 16 byte values in VGPRs.  S-box / inverse S-box (:44,:64) sit in LDS -- a single shared copy, i.e.
// read-only memory outside the sphere of replication, like the reference's const tables under -noMemReplication.
// Sync points (frozen in oracle/coast_oracle.c): the 4 state dwords and 4 key dwords at the end (they are stored back),
// and with sync_every != 0 also after every main-loop round.
//
// The kernels, one wave per tile of IPW blocks (the persistent bank-replicated forms of the two lean ones: further down):
//   aes128_enc_fast_kernel  encryption, mandatory sync points only (a tile that owns an armed upset applies it itself): state and key as four
//                           little-endian column dwords; SubBytes + ShiftRows + MixColumns of one round are 16 lookups in
//                           four 1-KiB LDS tables (Te_r[v] = MixColumns column r scaled by S[v]) folded with v_bitop3
//                           xors -- the same bytes as TI_aes_128.c:142-185 computes one at a time.
//   aes128_dec_fast_kernel  decryption likewise: InvMixColumns is linear, so the state chain is 16 Td lookups per round
//                           and InvMix(round key) follows the inverse key schedule through a second table set.
//   aes128_xmr_kernel       byte-at-a-time exactly as written in the reference, both directions, injector hooks and
//                           per-round sync points; runs sync_every != 0 and -noStoreDataSync (round 3: armed tiles stay in the
//                           lean kernels).  aes128_indexed_kernel: the same walk with `round` / `i` inside the sphere of replication.
#include <type_traits>

#include "xmr.hpp"

namespace coast {

enum { SITE_AES_STATE = 16, SITE_AES_KEY = 17, SITE_AES_ROUND = 18, SITE_AES_I = 19 };

__constant__ uint8_t kAesRcon[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36}; // :83

// FIPS-197 S-box generated from its definition at load time by aes_tables_kernel (no pasted table)
__device__ uint8_t gAesSbox[256];
__device__ uint8_t gAesRsbox[256];
__device__ uint32_t gAesTe[4][256]; // Te_r[v]: bytes (out row 0..3) = MixColumns matrix column r times S[v]
__device__ uint32_t gAesTd[4][256]; // Td_r[v]:  InvMixColumns matrix column r times rsbox[v]
__device__ uint32_t gAesTis[4][256]; // Tis_r[v]: InvMixColumns matrix column r times sbox[v] (key-schedule word)
__device__ uint32_t gAesImcRcon[10]; // InvMixColumns of the column (Rcon[j], 0, 0, 0)

__device__ __forceinline__ uint32_t gf_mul_dev(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 8; ++i) {
        if (b & 1u)
            p ^= a;
        const uint32_t hi = a & 0x80u;
        a = (a << 1) & 0xffu;
        if (hi)
            a ^= 0x1bu;
        b >>= 1;
    }
    return p;
}

__global__ void aes_tables_kernel()
{
    const uint32_t x = threadIdx.x; // 256 threads
    uint32_t inv = 0;
    if (x)
        for (uint32_t y = 1; y < 256; ++y)
            if (gf_mul_dev(x, y) == 1u) {
                inv = y;
                break;
            }
    uint32_t s = inv;
    for (int k = 1; k <= 4; ++k)
        s ^= ((inv << k) | (inv >> (8 - k))) & 0xffu;
    s ^= 0x63u;
    gAesSbox[x] = (uint8_t)s;
    gAesRsbox[s] = (uint8_t)x;
    const uint32_t s2 = ((s << 1) ^ ((s & 0x80u) ? 0x1bu : 0u)) & 0xffu, s3 = s2 ^ s;
    // matrix [2 3 1 1; 1 2 3 1; 1 1 2 3; 3 1 1 2] (TI_aes_128.c:176-185): column r lists the weights of input row r
    gAesTe[0][x] = s2 | (s << 8) | (s << 16) | (s3 << 24);
    gAesTe[1][x] = s3 | (s2 << 8) | (s << 16) | (s << 24);
    gAesTe[2][x] = s | (s3 << 8) | (s2 << 16) | (s << 24);
    gAesTe[3][x] = s | (s << 8) | (s3 << 16) | (s2 << 24);
    __syncthreads(); // gAesRsbox complete (single workgroup)
    // inverse matrix [14 11 13 9; 9 14 11 13; 13 9 14 11; 11 13 9 14] (the :172-175 pre-step composed with :176-185)
    const uint32_t w[2] = {(uint32_t)gAesRsbox[x], s};
    for (int t = 0; t < 2; ++t) {
        const uint32_t v = w[t];
        const uint32_t e = gf_mul_dev(v, 14), b = gf_mul_dev(v, 11), d = gf_mul_dev(v, 13), n = gf_mul_dev(v, 9);
        uint32_t(*T)[256] = t ? gAesTis : gAesTd;
        T[0][x] = e | (n << 8) | (d << 16) | (b << 24);
        T[1][x] = b | (e << 8) | (n << 16) | (d << 24);
        T[2][x] = d | (b << 8) | (e << 16) | (n << 24);
        T[3][x] = n | (d << 8) | (b << 16) | (e << 24);
    }
    if (x < 10) {
        const uint32_t rc = kAesRcon[x];
        gAesImcRcon[x] = gf_mul_dev(rc, 14) | (gf_mul_dev(rc, 9) << 8) | (gf_mul_dev(rc, 13) << 16) |
                         (gf_mul_dev(rc, 11) << 24);
    }
}

// The LDS images of the persistent kernels' bank-replicated tables (layout: aes128_enc_rep_kernel / aes128_dec_rep_kernel below), built once per
// device behind aes_tables_kernel: a workgroup fills its LDS with a straight 16-byte copy of the image (4 / 8 pieces per thread, coalesced,
// L2-resident) instead of assembling it from the five tables word by word -- 16 dependent-latency iterations per thread, with a divergent
// branch per iteration in the decryption kernel: a launch with nothing to do took 9.8 / 15.0 us (profiles/r06_aes_fixed_cost.txt).
__device__ uint32_t gAesEncImage[65536 / 4];     // block {Te_0..Te_3}: entry v = row v of 256 B, slot r at 64 r, copy c at 4 c
__device__ uint32_t gAesDecImage[2 * 65536 / 4]; // block 0 {Td_0..Td_3}; block 1: pairs {Tis_0[v], S[v] x 4} at 8 c, rsbox[v] x 4 at 128 + 4 c
__global__ void aes_images_kernel() // 64 workgroups x 256 threads: thread = one dword of a 64 KiB block
{
    const uint32_t d = blockIdx.x * 256u + threadIdx.x; // dword of the block
    const uint32_t v = d >> 6, w = d & 63u;              // row (entry value), dword inside the row
    gAesEncImage[d] = gAesTe[w >> 4][v];
    gAesDecImage[d] = gAesTd[w >> 4][v];
    uint32_t x = 0u;
    if (w < 32u)
        x = (w & 1u) ? (uint32_t)gAesSbox[v] * 0x01010101u : gAesTis[0][v];
    else if (w < 48u)
        x = (uint32_t)gAesRsbox[v] * 0x01010101u;
    gAesDecImage[16384u + d] = x;
}

__device__ __forceinline__ uint32_t xtime(uint32_t v) { return ((v << 1) ^ ((v & 0x80u) ? 0x1bu : 0u)) & 0xffu; } // :88-99

__device__ __forceinline__ void aes_mix_col(uint32_t *c, bool inverse) // :168-185
{
    if (inverse) {
        const uint32_t u = xtime(xtime(c[0] ^ c[2])), v = xtime(xtime(c[1] ^ c[3]));
        c[0] ^= u;
        c[1] ^= v;
        c[2] ^= u;
        c[3] ^= v;
    }
    const uint32_t t = c[0] ^ c[1] ^ c[2] ^ c[3], c0 = c[0];
    c[0] ^= xtime(c[0] ^ c[1]) ^ t;
    c[1] ^= xtime(c[1] ^ c[2]) ^ t;
    c[2] ^= xtime(c[2] ^ c[3]) ^ t;
    c[3] ^= xtime(c[3] ^ c0) ^ t;
}

__device__ __forceinline__ void aes_key_fwd(uint32_t k[16], const uint8_t *sb, int rd) // :115-121, :220-226
{
    k[0] ^= (uint32_t)sb[k[13]] ^ (uint32_t)kAesRcon[rd];
    k[1] ^= sb[k[14]];
    k[2] ^= sb[k[15]];
    k[3] ^= sb[k[12]];
#pragma unroll
    for (int i = 4; i < 16; ++i)
        k[i] ^= k[i - 4];
}

__device__ __forceinline__ void aes_key_inv(uint32_t k[16], const uint8_t *sb, int rd) // :134-141
{
#pragma unroll
    for (int i = 15; i > 3; --i)
        k[i] ^= k[i - 4];
    k[0] ^= (uint32_t)sb[k[13]] ^ (uint32_t)kAesRcon[rd];
    k[1] ^= sb[k[14]];
    k[2] ^= sb[k[15]];
    k[3] ^= sb[k[12]];
}

// one main-loop iteration (:131-227) on one replica
__device__ __forceinline__ void aes_round(uint32_t s[16], uint32_t k[16], const uint8_t *sb, const uint8_t *rsb,
                                          bool dir, int rd)
{
    uint32_t t[16];
    if (dir) {
        aes_key_inv(k, sb, 9 - rd);
        if (rd > 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                aes_mix_col(s + 4 * c, true);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                t[4 * ((c + r) & 3) + r] = s[4 * c + r]; // inverse shift rows
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s[i] = (uint32_t)rsb[t[i]] ^ k[i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            t[i] = sb[s[i] ^ k[i]];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s[4 * c + r] = t[4 * ((c + r) & 3) + r]; // shift rows
        if (rd < 9) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                aes_mix_col(s + 4 * c, false);
        }
        aes_key_fwd(k, sb, rd);
    }
}

__device__ __forceinline__ uint32_t pack4(const uint32_t *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24); }
__device__ __forceinline__ void unpack4(uint32_t w, uint32_t *b)
{
    b[0] = w & 0xffu;
    b[1] = (w >> 8) & 0xffu;
    b[2] = (w >> 16) & 0xffu;
    b[3] = w >> 24;
}

template <int NREP>
__device__ __forceinline__ void aes_sync(uint32_t s[16], uint32_t k[16], const LaneMap<NREP> &lm, bool cnt, Tally &tl)
{
#pragma unroll
    for (int w = 0; w < 4; ++w)
        unpack4(xmr_store_sync<NREP>(pack4(s + 4 * w), lm, cnt, tl), s + 4 * w);
#pragma unroll
    for (int w = 0; w < 4; ++w)
        unpack4(xmr_store_sync<NREP>(pack4(k + 4 * w), lm, cnt, tl), k + 4 * w);
}

// ------------------------------------------------------------------------------------------------ fast encryption
__device__ __forceinline__ uint32_t aes_xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Injector hook of the lean kernels: XOR masks of the upsets that are due at the start of main-loop round `rd` (10: after the
// loop) in this lane's copy of the state / running key, as column dwords (byte r of column c = state[4c + r], so bit b of dword
// `index` is bit b % 8 of byte 4 index + b / 8: the same flip aes128_xmr_kernel applies to its byte registers).  Only tiles that
// own an armed fault come here (wave-uniform test on the range table).
struct AesDue { // XOR masks for the four state columns and the four key columns
    uint32_t s0 = 0u, s1 = 0u, s2 = 0u, s3 = 0u, k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;
};
__device__ __forceinline__ AesDue aes_due_scan(const DevFault *list, uint2 fr, uint32_t rd, int slot, int rep, bool laneLive)
{
    AesDue d;
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault *fp = list + fr.x + q;
        const uint32_t packed = *reinterpret_cast<const uint32_t *>(&fp->replica); // replica, site, bit, index
        if (fp->step != rd || (int)fp->local != slot || (int)(packed & 0xffu) != rep || !laneLive)
            continue;
        const uint32_t site = (packed >> 8) & 0xffu, m = 1u << ((packed >> 16) & 31u), c = (packed >> 24) & 3u;
        const uint32_t ms = site == (uint32_t)SITE_AES_STATE ? m : 0u, mk = site == (uint32_t)SITE_AES_KEY ? m : 0u;
        d.s0 ^= c == 0u ? ms : 0u, d.s1 ^= c == 1u ? ms : 0u, d.s2 ^= c == 2u ? ms : 0u, d.s3 ^= c == 3u ? ms : 0u;
        d.k0 ^= c == 0u ? mk : 0u, d.k1 ^= c == 1u ? mk : 0u, d.k2 ^= c == 2u ? mk : 0u, d.k3 ^= c == 3u ? mk : 0u;
    }
    return d;
}
// This lane's upsets, read from the table (HBM) ONCE per tile: up to four of them as 16-bit records {round : 4, column code : 3,
// bit : 5} in the lane's own 8-byte LDS slot -- a table scan per round (11 dependent trips to L2 per tile) made an armed tile cost
// several clean ones, and a persistent workgroup waits for its slowest wave; registers would cost the encryption kernels their
// 64-register budget.  More than four on one lane (wave-uniform through the ballot): the scan, as before.
typedef uint32_t aes_rec2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) aes_rec2 *aes_lds_rec_p; // a 32-bit LDS pointer: one register
struct AesLaneFaults {
    aes_lds_rec_p slot = nullptr; // this thread's LDS slot (written and read by this lane only: program order is enough)
    bool over = false;
};
__device__ __forceinline__ AesLaneFaults aes_gather_faults(aes_lds_rec_p slotLds, const FaultTab &ft, uint2 fr, int slot, int rep,
                                                           bool laneLive)
{
    AesLaneFaults lf;
    lf.slot = slotLds;
    uint32_t n = 0u, p01 = 0xffffffffu, p23 = 0xffffffffu; // round nibble 15 = empty
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault *fp = ft.list + fr.x + q;
        const uint32_t packed = *reinterpret_cast<const uint32_t *>(&fp->replica); // replica, site, bit, index
        const uint32_t site = (packed >> 8) & 0xffu;
        if ((int)fp->local != slot || (int)(packed & 0xffu) != rep || !laneLive || fp->step > 10u ||
            (site != (uint32_t)SITE_AES_STATE && site != (uint32_t)SITE_AES_KEY))
            continue;
        const uint32_t code = ((packed >> 24) & 3u) | (site == (uint32_t)SITE_AES_KEY ? 4u : 0u);
        const uint32_t rec = fp->step | (code << 4) | (((packed >> 16) & 31u) << 7);
        if (n == 0u)
            p01 = (p01 & 0xffff0000u) | rec;
        else if (n == 1u)
            p01 = (p01 & 0x0000ffffu) | (rec << 16);
        else if (n == 2u)
            p23 = (p23 & 0xffff0000u) | rec;
        else if (n == 3u)
            p23 = (p23 & 0x0000ffffu) | (rec << 16);
        n += 1u;
    }
    *slotLds = aes_rec2{p01, p23};
    lf.over = __builtin_amdgcn_ballot_w64(n > 4u) != 0ull;
    return lf;
}
// `any` (wave-uniform): some lane of the wave has an upset due this round -- the callers skip their mask arithmetic otherwise (the
// decryption hooks run InvMixColumns over the eight masks: done blindly at every hook it made an armed tile cost two clean ones)
__device__ __forceinline__ AesDue aes_due_masks(const AesLaneFaults &lf, const FaultTab &ft, uint2 fr, uint32_t rd, int slot, int rep,
                                                bool laneLive, bool &any)
{
    any = true;
    if (lf.over)
        return aes_due_scan(ft.list, fr, rd, slot, rep, laneLive);
    AesDue d;
    aes_rec2 p = *lf.slot;
    // opaque per round: otherwise the eleven decodes of the unrolled rounds are merged and computed up front (their masks then live
    // through the whole block: 30 spilled registers in the 64-register encryption kernel)
    asm volatile("" : "+v"(p.x), "+v"(p.y));
    const bool hit = (p.x & 15u) == rd || ((p.x >> 16) & 15u) == rd || (p.y & 15u) == rd || ((p.y >> 16) & 15u) == rd;
    any = __builtin_amdgcn_ballot_w64(hit) != 0ull;
    if (any) { // wave-uniform: almost every round of an armed tile passes by
        auto one = [&](uint32_t rec) __attribute__((always_inline)) {
            const uint32_t mm = (rec & 15u) == rd ? 1u << ((rec >> 7) & 31u) : 0u, code = (rec >> 4) & 7u;
            d.s0 ^= code == 0u ? mm : 0u, d.s1 ^= code == 1u ? mm : 0u, d.s2 ^= code == 2u ? mm : 0u, d.s3 ^= code == 3u ? mm : 0u;
            d.k0 ^= code == 4u ? mm : 0u, d.k1 ^= code == 5u ? mm : 0u, d.k2 ^= code == 6u ? mm : 0u, d.k3 ^= code == 7u ? mm : 0u;
        };
        one(p.x & 0xffffu);
        one(p.x >> 16);
        one(p.y & 0xffffu);
        one(p.y >> 16);
    }
    return d;
}

__device__ __forceinline__ uint2 aes_tile_faults(const FaultTab &ft, uint64_t tile)
{
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range) {
        const uint2 rg = ft.range[tile];
        fr.x = __builtin_amdgcn_readfirstlane(rg.x);
        fr.y = __builtin_amdgcn_readfirstlane(rg.y);
    }
    return fr;
}

template <int NREP>
__global__ __launch_bounds__(256) void aes128_enc_fast_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                              uint64_t nblocksData, uint64_t ntiles, Counters ctr,
                                                              FaultTab ft, uint8_t *__restrict__ detected, size_t copyBytes = 0)
{
    __shared__ uint32_t sTe[4][256];
    __shared__ uint8_t sSb[256];
    __shared__ uint32_t sCnt[4];
    __shared__ uint2 sLf[256]; // the lanes' armed upsets (aes_gather_faults)
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        sTe[r][tid] = gAesTe[r][tid];
    sSb[tid] = gAesSbox[tid];
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    const uint64_t tile = (uint64_t)blockIdx.x * 4 + (tid >> 6);
    const bool skip = tile >= ntiles;
    const uint2 fr = skip ? make_uint2(0u, 0u) : aes_tile_faults(ft, tile); // this tile's armed upsets (wave-uniform)
    const uint64_t item = tile * IPW + (uint64_t)lm.q;
    const bool live = !skip && lm.live && item < nblocksData;
    const bool cnt = live && lm.r == 0;
    const uint64_t it = live ? item : 0;
    uint4 sv = reinterpret_cast<const uint4 *>(states + (size_t)lm.r * copyBytes)[it];
    uint4 kv = reinterpret_cast<const uint4 *>(keys + (size_t)lm.r * copyBytes)[it];
    uint32_t s0 = sv.x, s1 = sv.y, s2 = sv.z, s3 = sv.w; // column c = state[4c..4c+3], row r in bits 8r
    uint32_t k0 = kv.x, k1 = kv.y, k2 = kv.z, k3 = kv.w;

#define B0(x) ((x) & 0xffu)
#define B1(x) (((x) >> 8) & 0xffu)
#define B2(x) (((x) >> 16) & 0xffu)
#define B3(x) ((x) >> 24)
    // The ten rounds, instantiated twice: HOOKED = a tile that owns armed upsets applies the flips that are due at the start of
    // each round (and after the loop) to this lane's state / key registers; everything downstream -- the remaining rounds, the
    // sync points, the counters, the stores -- is the code every other tile runs.
    auto rounds = [&](auto hookTag) __attribute__((always_inline)) {
        constexpr bool HOOKED = decltype(hookTag)::value;
            AesLaneFaults lf; // this lane's armed upsets (read once)
            if constexpr (HOOKED)
                lf = aes_gather_faults((aes_lds_rec_p)(sLf + tid), ft, fr, lm.q, lm.r, lm.live);
        auto hook = [&](int rd) __attribute__((always_inline)) {
            if constexpr (HOOKED) {
                bool any;
                const AesDue d = aes_due_masks(lf, ft, fr, (uint32_t)rd, lm.q, lm.r, lm.live, any);
                if (any) {
                    s0 ^= d.s0, s1 ^= d.s1, s2 ^= d.s2, s3 ^= d.s3;
                    k0 ^= d.k0, k1 ^= d.k1, k2 ^= d.k2, k3 ^= d.k3;
                }
            }
        };
#pragma unroll
        for (int rd = 0; rd < 10; ++rd) {
            hook(rd);
            const uint32_t x0 = s0 ^ k0, x1 = s1 ^ k1, x2 = s2 ^ k2, x3 = s3 ^ k3; // state[i] ^ key[i]       (:144-146)
            if (rd < 9) { // SubBytes, ShiftRows (row r of column j comes from column j+r, :148-166), MixColumns (:168-185)
                s0 = aes_xor3(sTe[0][B0(x0)], sTe[1][B1(x1)], sTe[2][B2(x2)]) ^ sTe[3][B3(x3)];
                s1 = aes_xor3(sTe[0][B0(x1)], sTe[1][B1(x2)], sTe[2][B2(x3)]) ^ sTe[3][B3(x0)];
                s2 = aes_xor3(sTe[0][B0(x2)], sTe[1][B1(x3)], sTe[2][B2(x0)]) ^ sTe[3][B3(x1)];
                s3 = aes_xor3(sTe[0][B0(x3)], sTe[1][B1(x0)], sTe[2][B2(x1)]) ^ sTe[3][B3(x2)];
            } else { // last round: no MixColumns
                s0 = (uint32_t)sSb[B0(x0)] | ((uint32_t)sSb[B1(x1)] << 8) | ((uint32_t)sSb[B2(x2)] << 16) | ((uint32_t)sSb[B3(x3)] << 24);
                s1 = (uint32_t)sSb[B0(x1)] | ((uint32_t)sSb[B1(x2)] << 8) | ((uint32_t)sSb[B2(x3)] << 16) | ((uint32_t)sSb[B3(x0)] << 24);
                s2 = (uint32_t)sSb[B0(x2)] | ((uint32_t)sSb[B1(x3)] << 8) | ((uint32_t)sSb[B2(x0)] << 16) | ((uint32_t)sSb[B3(x1)] << 24);
                s3 = (uint32_t)sSb[B0(x3)] | ((uint32_t)sSb[B1(x0)] << 8) | ((uint32_t)sSb[B2(x1)] << 16) | ((uint32_t)sSb[B3(x2)] << 24);
            }
            // key schedule (:220-226): key[0..3] ^= sbox[key[13,14,15,12]] (^ Rcon on byte 0), then key[i] ^= key[i-4]
            const uint32_t sw = (uint32_t)sSb[B1(k3)] | ((uint32_t)sSb[B2(k3)] << 8) | ((uint32_t)sSb[B3(k3)] << 16) |
                                ((uint32_t)sSb[B0(k3)] << 24);
            k0 = aes_xor3(k0, sw, (uint32_t)kAesRcon[rd]);
            k1 ^= k0;
            k2 ^= k1;
            k3 ^= k2;
        }
        hook(10);
    };
    if (fr.y != 0u)
        rounds(std::true_type{});
    else
        rounds(std::false_type{});
#undef B0
#undef B1
#undef B2
#undef B3
    s0 ^= k0; // last AddRoundKey (:228-233)
    s1 ^= k1;
    s2 ^= k2;
    s3 ^= k3;

    Tally tl;
    s0 = xmr_sync<NREP>(s0, lm, cnt, tl); // in-place stores of state and key: store-data sync
    s1 = xmr_sync<NREP>(s1, lm, cnt, tl);
    s2 = xmr_sync<NREP>(s2, lm, cnt, tl);
    s3 = xmr_sync<NREP>(s3, lm, cnt, tl);
    k0 = xmr_sync<NREP>(k0, lm, cnt, tl);
    k1 = xmr_sync<NREP>(k1, lm, cnt, tl);
    k2 = xmr_sync<NREP>(k2, lm, cnt, tl);
    k3 = xmr_sync<NREP>(k3, lm, cnt, tl);
    uint32_t detItems = 0;
    if (cnt || (live && copyBytes != 0)) { // memory copies: every replica stores the voted state / key into its own copy
        reinterpret_cast<uint4 *>(states + (size_t)lm.r * copyBytes)[item] = make_uint4(s0, s1, s2, s3);
        reinterpret_cast<uint4 *>(keys + (size_t)lm.r * copyBytes)[item] = make_uint4(k0, k1, k2, k3);
    }
    if (cnt) {
        if (tl.det) { // unequal copies seen at a sync point of this block (DWC: detected, TMR: corrected)
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ fast decryption
// The reference's decryption round is InvKeySchedule, InvMixColumns (rounds 1..9), InvShiftRows, rsbox ^ key
// (TI_aes_128.c:132-141, 168-175, 188-212).  InvMixColumns is linear, so InvMix(rsbox[.] ^ key) = Td-lookups ^ InvMix(key):
// the state chain becomes 16 Td lookups per round, and InvMix(key) follows the (linear) inverse key schedule with one
// Tis lookup per S-box byte.  The running key itself is still walked back to the cipher key, as the contract requires.
__device__ __forceinline__ uint32_t aes_xtime4(uint32_t v)
{
    return ((v & 0x7f7f7f7fu) << 1) ^ (((v >> 7) & 0x01010101u) * 0x1bu);
}
__device__ __forceinline__ uint32_t aes_imc_col(uint32_t x) // InvMixColumns of one packed column (rows in bytes 0..3)
{
    const uint32_t t = x ^ __builtin_amdgcn_alignbit(x, x, 16);
    const uint32_t y = x ^ aes_xtime4(aes_xtime4(t));
    const uint32_t r8 = __builtin_amdgcn_alignbit(y, y, 8);
    return aes_xtime4(y ^ r8) ^ aes_xor3(r8, __builtin_amdgcn_alignbit(y, y, 16), __builtin_amdgcn_alignbit(y, y, 24));
}

// Injector hook of the decryption kernels.  At the start of reference round rd = 1..9 the lean kernels hold the state as
// x = InvMixColumns(state) (the first thing that round does to it, :168-175) and carry m = InvMixColumns(key) beside the key:
// a flipped state register becomes x ^= InvMix(mask) -- exact, the transformation is linear -- and a flipped key register
// keeps its image consistent, m ^= InvMix(mask).  Rounds 0 (nothing mixed yet) and 10 (after the loop) flip the registers as they are.
#define AES_DEC_HOOK(rd, mixedState, haveM)                                                                       \
    do {                                                                                                          \
        if constexpr (HOOKED) {                                                                                   \
            bool any_;                                                                                            \
            const AesDue d_ = aes_due_masks(lf, ft, fr, (uint32_t)(rd), lm.q, lm.r, lm.live, any_);               \
            if (any_) {                                                                                           \
                if (mixedState)                                                                                   \
                    x0 ^= aes_imc_col(d_.s0), x1 ^= aes_imc_col(d_.s1), x2 ^= aes_imc_col(d_.s2), x3 ^= aes_imc_col(d_.s3); \
                else                                                                                              \
                    x0 ^= d_.s0, x1 ^= d_.s1, x2 ^= d_.s2, x3 ^= d_.s3;                                           \
                k0 ^= d_.k0, k1 ^= d_.k1, k2 ^= d_.k2, k3 ^= d_.k3;                                               \
                if (haveM)                                                                                        \
                    m0 ^= aes_imc_col(d_.k0), m1 ^= aes_imc_col(d_.k1), m2 ^= aes_imc_col(d_.k2), m3 ^= aes_imc_col(d_.k3); \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

template <int NREP>
__global__ __launch_bounds__(256) void aes128_dec_fast_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                              uint64_t nblocksData, uint64_t ntiles, Counters ctr,
                                                              FaultTab ft, uint8_t *__restrict__ detected, size_t copyBytes = 0)
{
    __shared__ uint32_t sTd[4][256];
    __shared__ uint32_t sTis[4][256];
    __shared__ uint8_t sSb[256];
    __shared__ uint8_t sRsb[256];
    __shared__ uint32_t sCnt[4];
    __shared__ uint2 sLf[256]; // the lanes' armed upsets (aes_gather_faults)
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sTd[r][tid] = gAesTd[r][tid];
        sTis[r][tid] = gAesTis[r][tid];
    }
    sSb[tid] = gAesSbox[tid];
    sRsb[tid] = gAesRsbox[tid];
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    const uint64_t tile = (uint64_t)blockIdx.x * 4 + (tid >> 6);
    const bool skip = tile >= ntiles;
    const uint2 fr = skip ? make_uint2(0u, 0u) : aes_tile_faults(ft, tile); // this tile's armed upsets (wave-uniform)
    const uint64_t item = tile * IPW + (uint64_t)lm.q;
    const bool live = !skip && lm.live && item < nblocksData;
    const bool cnt = live && lm.r == 0;
    const uint64_t it = live ? item : 0;
    const uint4 sv = reinterpret_cast<const uint4 *>(states + (size_t)lm.r * copyBytes)[it];
    const uint4 kv = reinterpret_cast<const uint4 *>(keys + (size_t)lm.r * copyBytes)[it];
    uint32_t k0 = kv.x, k1 = kv.y, k2 = kv.z, k3 = kv.w;
    uint32_t x0, x1, x2, x3; // the state; after the loop: the plaintext columns

#define B0(x) ((x) & 0xffu)
#define B1(x) (((x) >> 8) & 0xffu)
#define B2(x) (((x) >> 16) & 0xffu)
#define B3(x) ((x) >> 24)
#define SUBROT(k) ((uint32_t)sSb[B1(k)] | ((uint32_t)sSb[B2(k)] << 8) | ((uint32_t)sSb[B3(k)] << 16) | ((uint32_t)sSb[B0(k)] << 24))
    // instantiated twice, as in aes128_enc_fast_kernel: HOOKED = a tile that owns armed upsets
    auto rounds = [&](auto hookTag) __attribute__((always_inline)) {
        constexpr bool HOOKED = decltype(hookTag)::value;
            AesLaneFaults lf; // this lane's armed upsets (read once)
            if constexpr (HOOKED)
                lf = aes_gather_faults((aes_lds_rec_p)(sLf + tid), ft, fr, lm.q, lm.r, lm.live);
        uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
        // the last encryption key first (:110-123)
#pragma unroll
        for (int rd = 0; rd < 10; ++rd) {
            k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[rd]);
            k1 ^= k0;
            k2 ^= k1;
            k3 ^= k2;
        }
        x0 = sv.x ^ k0, x1 = sv.y ^ k1, x2 = sv.z ^ k2, x3 = sv.w ^ k3; // first AddRoundKey (:126-128)
        AES_DEC_HOOK(0, false, false);

        // round 0: inverse key schedule to round key 9, no InvMixColumns on the state yet
        k3 ^= k2;
        k2 ^= k1;
        k1 ^= k0;
        k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[9]);
        m0 = aes_imc_col(k0), m1 = aes_imc_col(k1), m2 = aes_imc_col(k2), m3 = aes_imc_col(k3); // InvMix(key 9)
        {   // InvShiftRows: row r of column j comes from column j-r (:188-207); rsbox ^ key, then next round's InvMixColumns
            const uint32_t w0 = aes_xor3(sTd[0][B0(x0)], sTd[1][B1(x3)], sTd[2][B2(x2)]) ^ sTd[3][B3(x1)] ^ m0;
            const uint32_t w1 = aes_xor3(sTd[0][B0(x1)], sTd[1][B1(x0)], sTd[2][B2(x3)]) ^ sTd[3][B3(x2)] ^ m1;
            const uint32_t w2 = aes_xor3(sTd[0][B0(x2)], sTd[1][B1(x1)], sTd[2][B2(x0)]) ^ sTd[3][B3(x3)] ^ m2;
            const uint32_t w3 = aes_xor3(sTd[0][B0(x3)], sTd[1][B1(x2)], sTd[2][B2(x1)]) ^ sTd[3][B3(x0)] ^ m3;
            x0 = w0;
            x1 = w1;
            x2 = w2;
            x3 = w3;
        }
#pragma unroll
        for (int j = 8; j >= 1; --j) { // reference rounds 1..8: key j+1 -> key j, state through Td
            AES_DEC_HOOK(9 - j, true, true);
            k3 ^= k2;
            k2 ^= k1;
            k1 ^= k0;
            m3 ^= m2;
            m2 ^= m1;
            m1 ^= m0;
            k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[j]);
            m0 ^= aes_xor3(sTis[0][B1(k3)], sTis[1][B2(k3)], sTis[2][B3(k3)]) ^ sTis[3][B0(k3)] ^ gAesImcRcon[j];
            const uint32_t w0 = aes_xor3(sTd[0][B0(x0)], sTd[1][B1(x3)], sTd[2][B2(x2)]) ^ sTd[3][B3(x1)] ^ m0;
            const uint32_t w1 = aes_xor3(sTd[0][B0(x1)], sTd[1][B1(x0)], sTd[2][B2(x3)]) ^ sTd[3][B3(x2)] ^ m1;
            const uint32_t w2 = aes_xor3(sTd[0][B0(x2)], sTd[1][B1(x1)], sTd[2][B2(x0)]) ^ sTd[3][B3(x3)] ^ m2;
            const uint32_t w3 = aes_xor3(sTd[0][B0(x3)], sTd[1][B1(x2)], sTd[2][B2(x1)]) ^ sTd[3][B3(x0)] ^ m3;
            x0 = w0;
            x1 = w1;
            x2 = w2;
            x3 = w3;
        }
        // reference round 9: key 1 -> cipher key, InvShiftRows, rsbox ^ key (x already carries this round's InvMixColumns)
        AES_DEC_HOOK(9, true, false);
        k3 ^= k2;
        k2 ^= k1;
        k1 ^= k0;
        k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[0]);
        const uint32_t p0 = ((uint32_t)sRsb[B0(x0)] | ((uint32_t)sRsb[B1(x3)] << 8) | ((uint32_t)sRsb[B2(x2)] << 16) | ((uint32_t)sRsb[B3(x1)] << 24)) ^ k0;
        const uint32_t p1 = ((uint32_t)sRsb[B0(x1)] | ((uint32_t)sRsb[B1(x0)] << 8) | ((uint32_t)sRsb[B2(x3)] << 16) | ((uint32_t)sRsb[B3(x2)] << 24)) ^ k1;
        const uint32_t p2 = ((uint32_t)sRsb[B0(x2)] | ((uint32_t)sRsb[B1(x1)] << 8) | ((uint32_t)sRsb[B2(x0)] << 16) | ((uint32_t)sRsb[B3(x3)] << 24)) ^ k2;
        const uint32_t p3 = ((uint32_t)sRsb[B0(x3)] | ((uint32_t)sRsb[B1(x2)] << 8) | ((uint32_t)sRsb[B2(x1)] << 16) | ((uint32_t)sRsb[B3(x0)] << 24)) ^ k3;
        x0 = p0, x1 = p1, x2 = p2, x3 = p3;
        AES_DEC_HOOK(10, false, false);
    };
    if (fr.y != 0u)
        rounds(std::true_type{});
    else
        rounds(std::false_type{});
#undef SUBROT
#undef B0
#undef B1
#undef B2
#undef B3
    uint32_t s0 = x0, s1 = x1, s2 = x2, s3 = x3;

    Tally tl;
    s0 = xmr_sync<NREP>(s0, lm, cnt, tl); // in-place stores of state and key: store-data sync
    s1 = xmr_sync<NREP>(s1, lm, cnt, tl);
    s2 = xmr_sync<NREP>(s2, lm, cnt, tl);
    s3 = xmr_sync<NREP>(s3, lm, cnt, tl);
    k0 = xmr_sync<NREP>(k0, lm, cnt, tl);
    k1 = xmr_sync<NREP>(k1, lm, cnt, tl);
    k2 = xmr_sync<NREP>(k2, lm, cnt, tl);
    k3 = xmr_sync<NREP>(k3, lm, cnt, tl);
    uint32_t detItems = 0;
    if (cnt || (live && copyBytes != 0)) { // memory copies: every replica stores the voted state / key into its own copy
        reinterpret_cast<uint4 *>(states + (size_t)lm.r * copyBytes)[item] = make_uint4(s0, s1, s2, s3);
        reinterpret_cast<uint4 *>(keys + (size_t)lm.r * copyBytes)[item] = make_uint4(k0, k1, k2, k3);
    }
    if (cnt) {
        if (tl.det) {
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ bank-replicated tables
// The T-table kernels above are LDS-bound and half of their LDS cycles are bank conflicts: the 16 distinct blocks of a DWC
// lane group hit 32 banks at random (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 46 %, 4.1 cycles per lookup instead of 2;
// profiles/r02_aes_rocprofv3_summary.txt).  For large batches the tables are therefore REPLICATED: copy c of entry v sits at
// word v * 16 + c, i.e. in bank c + 16 (v & 1), and the lanes of block slot q read copy q % 16 -- the replicas of a block
// share a copy (same address: broadcast), different blocks of a 32-lane group can never meet on a bank, whatever their
// indices.  Still one shared read-only copy set per workgroup, outside the sphere of replication like the tables above.
// 64 KiB (encryption) / 112 KiB (decryption) of LDS, so the kernels are persistent: 1024-thread workgroups, waves
// grid-stride over tiles, tables filled once per workgroup.  The byte tables ride along instead of conflicting beside them:
//   encryption  S[v] is byte 1 of Te_0[v] (= (2S, S, S, 3S)): key schedule and last round look up Te_0 and pick the byte
//               with v_perm_b32;
//   decryption  Td_0 and Tis_0 are stored as 8-byte pairs {Td_0[v], rsbox[v] * 0x01010101} / {Tis_0[v], S[v] * 0x01010101}
//               and read with ds_read_b64 (same 2 LDS cycles as a dword read): the key-schedule word and its InvMixColumns
//               image come out of ONE lookup per byte (Tis_r = Tis_0 rotated by r bytes), the last round reads the .y halves.
// Round 3, the lookup address in ONE instruction: every table entry value v owns a 256-byte row of the table block -- four slots
// of 16 copies x 4 bytes -- so the LDS address of "slot r, entry x.byte[B], this lane's copy" is the byte string
// {copy * 4, x.byte[B], block, 0}: one v_perm_b32 of x and a per-lane constant, with r * 64 in the instruction's offset field
// (rounds 1-2 needed v_bfe_u32 + v_lshl_add_u32 per lookup; the kernels are VALU-bound).  Encryption: one 64 KiB
// block {Te_0, Te_1, Te_2, Te_3}.  Decryption: block 0 = {Td_0, Td_1, Td_2, Td_3}, block 1 = {Tis_0 | S pairs (two slots),
// rsbox, -}.  A copy still sits in bank 16 r + copy (dwords) / 2 copy + {0, 1} (pairs): the 16 distinct blocks of a 32-lane
// group never meet on a bank.
// 1 (shipped) = the exit votes of the persistent kernels in DPP form when only replica 0 stores; 0 = ds_bpermute everywhere (A/B: profiles/r06_aes_dpp_votes.txt)
#ifndef COAST_AES_DPP_VOTES
#define COAST_AES_DPP_VOTES 1
#endif
constexpr int kAesCopies = 16;
constexpr int kAesRepThreads = 1024;
constexpr int kAesRowBytes = 256;                      // one entry value: 4 slots x 16 copies x 4 bytes
constexpr int kAesBlockBytes = 256 * kAesRowBytes;     // 64 KiB
constexpr size_t kAesEncRepLds = (size_t)kAesBlockBytes + 16 + kAesRepThreads * 8; // tables, counters, the lanes' upset records
constexpr size_t kAesDecRepLds = (size_t)2 * kAesBlockBytes + 16 + kAesRepThreads * 8;
typedef const __attribute__((address_space(3))) uint32_t *aes_lds_u32p;
typedef uint32_t aes_u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) aes_u32x2 *aes_lds_u64p;
// LDS address of entry x.byte[B] in table block BLK for the lane whose selector word is laneSel = copy offset | 0x100
template <int B, int BLK> __device__ __forceinline__ uint32_t aes_rep_addr(uint32_t x, uint32_t laneSel)
{
    return __builtin_amdgcn_perm(x, laneSel, (BLK ? 0x0c010000u : 0x0c0c0000u) + ((4u + B) << 8));
}

__device__ __forceinline__ uint32_t aes_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32
// bytes 0..3 = byte 1 of t0..t3
__device__ __forceinline__ uint32_t aes_pick_b1(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3)
{
    return __builtin_amdgcn_perm(t1, t0, 0x0c0c0501u) | __builtin_amdgcn_perm(t3, t2, 0x05010c0cu);
}
// byte i of the result = byte i of yi (the yi carry one byte value in all four positions)
__device__ __forceinline__ uint32_t aes_pick_rep(uint32_t y0, uint32_t y1, uint32_t y2, uint32_t y3)
{
    return aes_bfi(0x000000ffu, y0, aes_bfi(0x0000ff00u, y1, aes_bfi(0x00ff0000u, y2, y3)));
}
__device__ __forceinline__ uint32_t aes_rotl8(uint32_t x, int bytes) { return __builtin_amdgcn_alignbit(x, x, 32 - 8 * bytes); }

template <int NREP>
__global__ __launch_bounds__(kAesRepThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void aes128_enc_rep_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                                        uint64_t nblocksData, uint64_t ntiles, Counters ctr,
                                                                        FaultTab ft, uint8_t *__restrict__ detected, size_t copyBytes = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemAes[];
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemAes + kAesBlockBytes);
    uint2 *sLf = reinterpret_cast<uint2 *>(smemAes + kAesBlockBytes + 16);
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smemAes != 0u)
        __builtin_trap(); // the permuted bytes are the LDS address: the dynamic segment starts at 0
    // the wave's first tile is requested before the tables are filled: its HBM latency runs under the fill and the barrier
    const uint64_t tile0 = (uint64_t)blockIdx.x * (kAesRepThreads / kWave) + (tid >> 6);
    auto loadTile = [&](uint64_t tile, uint4 &svO, uint4 &kvO) __attribute__((always_inline)) {
        // (the tile's lane map from a fresh lane id: the per-lane array bases of the memory-copies mode are loop invariants the
        // register allocator would otherwise carry -- spilled -- through every tile)
        const LaneMap<NREP> lmT(xmr_fresh_lane());
        const uint64_t item = tile * IPW + (uint64_t)lmT.q;
        const uint64_t it = (lmT.live && item < nblocksData) ? item : 0;
        const uint8_t *stBase = states, *kyBase = keys; // (kept in SGPRs up to here: their VGPR copies were hoisted and spilled)
        asm volatile("" : "+s"(stBase), "+s"(kyBase));
        svO = reinterpret_cast<const uint4 *>(stBase + (size_t)lmT.r * copyBytes)[it];
        kvO = reinterpret_cast<const uint4 *>(kyBase + (size_t)lmT.r * copyBytes)[it];
    };
    uint4 sv0 = make_uint4(0u, 0u, 0u, 0u), kv0 = sv0;
    uint2 fr0 = make_uint2(0u, 0u); // the tile's armed upsets (wave-uniform), requested with its blocks: a load per tile in front of a branch otherwise
    if (tile0 < ntiles) {
        loadTile(tile0, sv0, kv0);
        if (ft.range)
            fr0 = ft.range[tile0];
    }
    { // the table block: row v = entry v, slot r at 64 r, copy c at 4 c -- a straight copy of the image aes_images_kernel built (4 pieces per thread)
        const uint4 *src = reinterpret_cast<const uint4 *>(gAesEncImage);
        uint4 *dst = reinterpret_cast<uint4 *>(smemAes);
#pragma unroll
        for (int i = 0; i < kAesBlockBytes / 16 / kAesRepThreads; ++i)
            dst[i * kAesRepThreads + tid] = src[i * kAesRepThreads + tid];
    }
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();
    const uint32_t laneSel = (uint32_t)((NREP == 1 ? lm.lane : lm.q) & (kAesCopies - 1)) * 4u | 0x100u; // this lane's copy
#define TE(r, x, b) (*(aes_lds_u32p)(uintptr_t)(aes_rep_addr<b, 0>(x, laneSel) + (r) * 64))
    Tally tl;
    uint32_t detItems = 0;
    for (uint64_t tile = tile0; tile < ntiles; tile += (uint64_t)gridDim.x * (kAesRepThreads / kWave)) {
        const uint2 fr = make_uint2(__builtin_amdgcn_readfirstlane(fr0.x), __builtin_amdgcn_readfirstlane(fr0.y)); // this tile's armed upsets
        const LaneMap<NREP> lmT(xmr_fresh_lane());
        const uint4 sv = sv0, kv = kv0; // requested before the tables were filled / behind the previous tile's stores
        uint32_t s0 = sv.x, s1 = sv.y, s2 = sv.z, s3 = sv.w;
        uint32_t k0 = kv.x, k1 = kv.y, k2 = kv.z, k3 = kv.w;
        // One formulation for clean and armed tiles (round 4): AddRoundKey of the next round folded into the column sums, x = s ^ k carried.
        // An armed upset of round rd lands on the registers as they are at that point: a flipped state register is x ^= mask, a flipped key
        // register x ^= mask and k ^= mask (x holds their xor) -- exact, the same values the separate s / k form produced.  (Rounds 1-3 ran
        // armed tiles through an unfolded copy of the rounds: 45 % more instructions on one tile of a wave that owns four -- the wave, and with
        // it the persistent workgroup, finished 11 us late: 59 instead of 47 us per 1 Mi blocks with 1024 armed upsets.)
        auto rounds = [&](auto hookTag) __attribute__((always_inline)) {
            constexpr bool HOOKED = decltype(hookTag)::value;
            AesLaneFaults lf; // this lane's armed upsets (read once)
            if constexpr (HOOKED)
                lf = aes_gather_faults((aes_lds_rec_p)(sLf + tid), ft, fr, lmT.q, lmT.r, lmT.live);
            uint32_t x0 = 0u, x1 = 0u, x2 = 0u, x3 = 0u;
            auto hook = [&](int rd, bool folded) __attribute__((always_inline)) {
                if constexpr (HOOKED) {
                    bool any;
                    const AesDue d = aes_due_masks(lf, ft, fr, (uint32_t)rd, lmT.q, lmT.r, lmT.live, any);
                    if (any) {
                        if (folded)
                            x0 ^= d.s0 ^ d.k0, x1 ^= d.s1 ^ d.k1, x2 ^= d.s2 ^ d.k2, x3 ^= d.s3 ^ d.k3;
                        else
                            s0 ^= d.s0, s1 ^= d.s1, s2 ^= d.s2, s3 ^= d.s3;
                        k0 ^= d.k0, k1 ^= d.k1, k2 ^= d.k2, k3 ^= d.k3;
                    }
                }
            };
            hook(0, false);
            x0 = s0 ^ k0, x1 = s1 ^ k1, x2 = s2 ^ k2, x3 = s3 ^ k3;
#pragma unroll
            for (int rd = 0; rd < 10; ++rd) {
                if (rd > 0)
                    hook(rd, true);
                uint32_t a0, a1, a2, a3, d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;
                if (rd < 9) {
                    a0 = aes_xor3(TE(0, x0, 0), TE(1, x1, 1), TE(2, x2, 2)), d0 = TE(3, x3, 3);
                    a1 = aes_xor3(TE(0, x1, 0), TE(1, x2, 1), TE(2, x3, 2)), d1 = TE(3, x0, 3);
                    a2 = aes_xor3(TE(0, x2, 0), TE(1, x3, 1), TE(2, x0, 2)), d2 = TE(3, x1, 3);
                    a3 = aes_xor3(TE(0, x3, 0), TE(1, x0, 1), TE(2, x1, 2)), d3 = TE(3, x2, 3);
                } else {
                    a0 = aes_pick_b1(TE(0, x0, 0), TE(0, x1, 1), TE(0, x2, 2), TE(0, x3, 3));
                    a1 = aes_pick_b1(TE(0, x1, 0), TE(0, x2, 1), TE(0, x3, 2), TE(0, x0, 3));
                    a2 = aes_pick_b1(TE(0, x2, 0), TE(0, x3, 1), TE(0, x0, 2), TE(0, x1, 3));
                    a3 = aes_pick_b1(TE(0, x3, 0), TE(0, x0, 1), TE(0, x1, 2), TE(0, x2, 3));
                }
                const uint32_t sw = aes_pick_b1(TE(0, k3, 1), TE(0, k3, 2), TE(0, k3, 3), TE(0, k3, 0));
                k0 = aes_xor3(k0, sw, (uint32_t)kAesRcon[rd]);
                k1 ^= k0;
                k2 ^= k1;
                k3 ^= k2;
                if (rd < 9)
                    x0 = aes_xor3(a0, d0, k0), x1 = aes_xor3(a1, d1, k1), x2 = aes_xor3(a2, d2, k2), x3 = aes_xor3(a3, d3, k3);
                else
                    s0 = a0, s1 = a1, s2 = a2, s3 = a3; // the last AddRoundKey follows the loop
            }
            hook(10, false);
        };
        if (fr.y != 0u)
            rounds(std::true_type{});
        else
            rounds(std::false_type{});
        s0 ^= k0;
        s1 ^= k1;
        s2 ^= k2;
        s3 ^= k3;
        // the epilogue's lane map, item and gates from a fresh lane id: the kernel has 64 registers per lane (two workgroups per
        // CU), and what the votes and stores need must not stay alive across the ten rounds (it was being spilled to scratch)
        const LaneMap<NREP> lmE(xmr_fresh_lane());
        const uint64_t itemE = tile * IPW + (uint64_t)lmE.q;
        const bool liveE = lmE.live && itemE < nblocksData, cntE = liveE && lmE.r == 0;
        Tally te = tl;
        te.det = 0;
        if (COAST_AES_DPP_VOTES && copyBytes == 0) { // only replica 0 stores: the DPP form of the exit votes (its two neighbours; no trip through the LDS crossbar,
                              // which the lookups of the other waves are using -- a ds_bpermute costs it three ds_read_b32)
            s0 = xmr_final_vote_dpp<NREP>(s0, cntE, te), s1 = xmr_final_vote_dpp<NREP>(s1, cntE, te);
            s2 = xmr_final_vote_dpp<NREP>(s2, cntE, te), s3 = xmr_final_vote_dpp<NREP>(s3, cntE, te);
            k0 = xmr_final_vote_dpp<NREP>(k0, cntE, te), k1 = xmr_final_vote_dpp<NREP>(k1, cntE, te);
            k2 = xmr_final_vote_dpp<NREP>(k2, cntE, te), k3 = xmr_final_vote_dpp<NREP>(k3, cntE, te);
        } else { // memory copies: every replica stores the voted value into its own copy
            s0 = xmr_sync<NREP>(s0, lmE, cntE, te);
            s1 = xmr_sync<NREP>(s1, lmE, cntE, te);
            s2 = xmr_sync<NREP>(s2, lmE, cntE, te);
            s3 = xmr_sync<NREP>(s3, lmE, cntE, te);
            k0 = xmr_sync<NREP>(k0, lmE, cntE, te);
            k1 = xmr_sync<NREP>(k1, lmE, cntE, te);
            k2 = xmr_sync<NREP>(k2, lmE, cntE, te);
            k3 = xmr_sync<NREP>(k3, lmE, cntE, te);
        }
        tl.miss = te.miss;
        tl.syncs = te.syncs;
        if (cntE || (liveE && copyBytes != 0)) {
            reinterpret_cast<uint4 *>(states + (size_t)lmE.r * copyBytes)[itemE] = make_uint4(s0, s1, s2, s3);
            reinterpret_cast<uint4 *>(keys + (size_t)lmE.r * copyBytes)[itemE] = make_uint4(k0, k1, k2, k3);
        }
        if (cntE) {
            if (te.det) {
                if (NREP == 2)
                    detItems += 1;
                if (detected)
                    detected[itemE] = 1;
            }
        }
        // the next tile (other blocks: the in-place stores above cannot reach them).  Requested HERE, where nothing else is live -- the
        // kernel has 64 registers per lane (two workgroups per CU) and no room to carry a prefetch through the rounds
        if (tile + (uint64_t)gridDim.x * (kAesRepThreads / kWave) < ntiles) {
            loadTile(tile + (uint64_t)gridDim.x * (kAesRepThreads / kWave), sv0, kv0);
            if (ft.range)
                fr0 = ft.range[tile + (uint64_t)gridDim.x * (kAesRepThreads / kWave)];
        }
    }
#undef TE
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
    block_fold(ctr, blockIdx.x, sCnt + 3);
}

template <int NREP>
__global__ __launch_bounds__(kAesRepThreads) void aes128_dec_rep_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                                        uint64_t nblocksData, uint64_t ntiles, Counters ctr,
                                                                        FaultTab ft, uint8_t *__restrict__ detected, size_t copyBytes = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemAes[];
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemAes + 2 * kAesBlockBytes);
    uint2 *sLf = reinterpret_cast<uint2 *>(smemAes + 2 * kAesBlockBytes + 16);
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smemAes != 0u)
        __builtin_trap(); // the permuted bytes are the LDS address: the dynamic segment starts at 0
    // the wave's first tile is requested before the tables are filled, every later one a tile ahead (the waves of a workgroup run
    // their tiles in step: without the prefetch they all wait for HBM at the same time)
    const uint64_t tile0 = (uint64_t)blockIdx.x * (kAesRepThreads / kWave) + (tid >> 6);
    const uint64_t tstride = (uint64_t)gridDim.x * (kAesRepThreads / kWave);
    auto loadTile = [&](uint64_t tile, uint4 &svO, uint4 &kvO) __attribute__((always_inline)) {
        const uint64_t item = tile * IPW + (uint64_t)lm.q;
        const uint64_t it = (lm.live && item < nblocksData) ? item : 0;
        svO = reinterpret_cast<const uint4 *>(states + (size_t)lm.r * copyBytes)[it];
        kvO = reinterpret_cast<const uint4 *>(keys + (size_t)lm.r * copyBytes)[it];
    };
    uint4 svN = make_uint4(0u, 0u, 0u, 0u), kvN = svN;
    uint2 frN = make_uint2(0u, 0u); // the tile's armed upsets (wave-uniform), requested a tile ahead with its blocks
    if (tile0 < ntiles) {
        loadTile(tile0, svN, kvN);
        if (ft.range)
            frN = ft.range[tile0];
    }
    { // block 0: Td_0..3; block 1: {Tis_0[v], S[v] x 4} pairs in slots 0-1 (8 bytes per copy), rsbox[v] x 4 in slot 2 -- a straight copy of the
      // image aes_images_kernel built (8 pieces per thread)
        const uint4 *src = reinterpret_cast<const uint4 *>(gAesDecImage);
        uint4 *dst = reinterpret_cast<uint4 *>(smemAes);
#pragma unroll
        for (int i = 0; i < 2 * kAesBlockBytes / 16 / kAesRepThreads; ++i)
            dst[i * kAesRepThreads + tid] = src[i * kAesRepThreads + tid];
    }
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();
    const uint32_t copy = (uint32_t)((NREP == 1 ? lm.lane : lm.q) & (kAesCopies - 1));
    const uint32_t laneSel = copy * 4u | 0x100u, laneSelP = copy * 8u | 0x100u; // dword slots / 8-byte pair slots
#define TD(r, x, b) (*(aes_lds_u32p)(uintptr_t)(aes_rep_addr<b, 0>(x, laneSel) + (r) * 64))
#define TS(x, b) (*(aes_lds_u64p)(uintptr_t)(aes_rep_addr<b, 1>(x, laneSelP)))
#define TSY(x, b) (*(aes_lds_u32p)(uintptr_t)(aes_rep_addr<b, 1>(x, laneSelP) + 4))  /* the S-box half of a pair alone */
#define RSB(x, b) (*(aes_lds_u32p)(uintptr_t)(aes_rep_addr<b, 1>(x, laneSel) + 128))
#define SUBROT(k) aes_pick_rep(TSY(k, 1), TSY(k, 2), TSY(k, 3), TSY(k, 0))
#define TDCOL(a, b, c, d, m) aes_xor3(aes_xor3(TD(0, a, 0), TD(1, b, 1), TD(2, c, 2)), TD(3, d, 3), m) /* column ^ InvMix(key) */
    Tally tl;
    uint32_t detItems = 0;
    for (uint64_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint2 fr = make_uint2(__builtin_amdgcn_readfirstlane(frN.x), __builtin_amdgcn_readfirstlane(frN.y)); // this tile's armed upsets
        const uint64_t item = tile * IPW + (uint64_t)lm.q;
        const bool live = lm.live && item < nblocksData;
        const bool cnt = live && lm.r == 0;
        const uint4 sv = svN, kv = kvN;
        if (tile + tstride < ntiles) { // the next tile's blocks are other blocks: the in-place stores below cannot reach them
            loadTile(tile + tstride, svN, kvN);
            if (ft.range)
                frN = ft.range[tile + tstride];
        }
        uint32_t k0 = kv.x, k1 = kv.y, k2 = kv.z, k3 = kv.w;
        uint32_t x0, x1, x2, x3;
        auto rounds = [&](auto hookTag) __attribute__((always_inline)) { // as aes128_dec_fast_kernel, HOOKED included
            constexpr bool HOOKED = decltype(hookTag)::value;
            AesLaneFaults lf; // this lane's armed upsets (read once)
            if constexpr (HOOKED)
                lf = aes_gather_faults((aes_lds_rec_p)(sLf + tid), ft, fr, lm.q, lm.r, lm.live);
            uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
#pragma unroll
            for (int rd = 0; rd < 10; ++rd) { // the last encryption key first (:110-123)
                k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[rd]);
                k1 ^= k0;
                k2 ^= k1;
                k3 ^= k2;
            }
            x0 = sv.x ^ k0, x1 = sv.y ^ k1, x2 = sv.z ^ k2, x3 = sv.w ^ k3;
            AES_DEC_HOOK(0, false, false);
            k3 ^= k2; // round 0
            k2 ^= k1;
            k1 ^= k0;
            k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[9]);
            m0 = aes_imc_col(k0), m1 = aes_imc_col(k1), m2 = aes_imc_col(k2), m3 = aes_imc_col(k3);
            {
                const uint32_t w0 = TDCOL(x0, x3, x2, x1, m0), w1 = TDCOL(x1, x0, x3, x2, m1);
                const uint32_t w2 = TDCOL(x2, x1, x0, x3, m2), w3 = TDCOL(x3, x2, x1, x0, m3);
                x0 = w0;
                x1 = w1;
                x2 = w2;
                x3 = w3;
            }
#pragma unroll
            for (int j = 8; j >= 1; --j) {
                AES_DEC_HOOK(9 - j, true, true);
                k3 ^= k2;
                k2 ^= k1;
                k1 ^= k0;
                m3 ^= m2;
                m2 ^= m1;
                m1 ^= m0;
                // one 8-byte lookup per byte of k3: the S-box byte for the key word, Tis_0 for its InvMixColumns image
                const aes_u32x2 a1 = TS(k3, 1), a2 = TS(k3, 2), a3 = TS(k3, 3), a0 = TS(k3, 0);
                k0 = aes_xor3(k0, aes_pick_rep(a1.y, a2.y, a3.y, a0.y), (uint32_t)kAesRcon[j]);
                m0 ^= aes_xor3(a1.x, aes_rotl8(a2.x, 1), aes_rotl8(a3.x, 2)) ^ aes_rotl8(a0.x, 3) ^ gAesImcRcon[j];
                const uint32_t w0 = TDCOL(x0, x3, x2, x1, m0), w1 = TDCOL(x1, x0, x3, x2, m1);
                const uint32_t w2 = TDCOL(x2, x1, x0, x3, m2), w3 = TDCOL(x3, x2, x1, x0, m3);
                x0 = w0;
                x1 = w1;
                x2 = w2;
                x3 = w3;
            }
            AES_DEC_HOOK(9, true, false);
            k3 ^= k2; // reference round 9
            k2 ^= k1;
            k1 ^= k0;
            k0 = aes_xor3(k0, SUBROT(k3), (uint32_t)kAesRcon[0]);
            const uint32_t p0 = aes_pick_rep(RSB(x0, 0), RSB(x3, 1), RSB(x2, 2), RSB(x1, 3)) ^ k0;
            const uint32_t p1 = aes_pick_rep(RSB(x1, 0), RSB(x0, 1), RSB(x3, 2), RSB(x2, 3)) ^ k1;
            const uint32_t p2 = aes_pick_rep(RSB(x2, 0), RSB(x1, 1), RSB(x0, 2), RSB(x3, 3)) ^ k2;
            const uint32_t p3 = aes_pick_rep(RSB(x3, 0), RSB(x2, 1), RSB(x1, 2), RSB(x0, 3)) ^ k3;
            x0 = p0, x1 = p1, x2 = p2, x3 = p3;
            AES_DEC_HOOK(10, false, false);
        };
        if (fr.y != 0u)
            rounds(std::true_type{});
        else
            rounds(std::false_type{});
        uint32_t s0 = x0, s1 = x1, s2 = x2, s3 = x3;
        Tally te = tl;
        te.det = 0;
        if (COAST_AES_DPP_VOTES && copyBytes == 0) { // only replica 0 stores: the DPP form of the exit votes (see aes128_enc_rep_kernel)
            s0 = xmr_final_vote_dpp<NREP>(s0, cnt, te), s1 = xmr_final_vote_dpp<NREP>(s1, cnt, te);
            s2 = xmr_final_vote_dpp<NREP>(s2, cnt, te), s3 = xmr_final_vote_dpp<NREP>(s3, cnt, te);
            k0 = xmr_final_vote_dpp<NREP>(k0, cnt, te), k1 = xmr_final_vote_dpp<NREP>(k1, cnt, te);
            k2 = xmr_final_vote_dpp<NREP>(k2, cnt, te), k3 = xmr_final_vote_dpp<NREP>(k3, cnt, te);
        } else {
            s0 = xmr_sync<NREP>(s0, lm, cnt, te);
            s1 = xmr_sync<NREP>(s1, lm, cnt, te);
            s2 = xmr_sync<NREP>(s2, lm, cnt, te);
            s3 = xmr_sync<NREP>(s3, lm, cnt, te);
            k0 = xmr_sync<NREP>(k0, lm, cnt, te);
            k1 = xmr_sync<NREP>(k1, lm, cnt, te);
            k2 = xmr_sync<NREP>(k2, lm, cnt, te);
            k3 = xmr_sync<NREP>(k3, lm, cnt, te);
        }
        tl.miss = te.miss;
        tl.syncs = te.syncs;
        if (cnt || (live && copyBytes != 0)) {
            reinterpret_cast<uint4 *>(states + (size_t)lm.r * copyBytes)[item] = make_uint4(s0, s1, s2, s3);
            reinterpret_cast<uint4 *>(keys + (size_t)lm.r * copyBytes)[item] = make_uint4(k0, k1, k2, k3);
        }
        if (cnt) {
            if (te.det) {
                if (NREP == 2)
                    detItems += 1;
                if (detected)
                    detected[item] = 1;
            }
        }
    }
#undef RSB
#undef TSY
#undef TS
#undef TD
#undef SUBROT
#undef TDCOL
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
    block_fold(ctr, blockIdx.x, sCnt + 3);
}

#undef AES_DEC_HOOK

// ------------------------------------------------------------------------------------------------ general path
template <int NREP>
__global__ __launch_bounds__(256, 5) void aes128_xmr_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                        uint64_t nblocksData, int dirFlag, uint32_t syncEvery,
                                                        Counters ctr, FaultTab ft,
                                                        const uint32_t *__restrict__ tileList, uint32_t nslots,
                                                        uint8_t *__restrict__ detected)
{
    __shared__ __attribute__((aligned(16))) uint8_t sSb[256];
    __shared__ __attribute__((aligned(16))) uint8_t sRsb[256];
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    // blockDim.x / 64 tiles per workgroup: 1 when walking the faulted-tile list, 4 when covering a whole batch
    const uint32_t tslot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool slotOk = tslot < nslots; // the last workgroup of a whole-batch launch may hang over
    const uint32_t lb = slotOk ? (tileList ? tileList[tslot] : tslot) : 0u; // tile
    const int slot = lm.q;
    const uint64_t item = (uint64_t)lb * IPW + (uint64_t)slot;
    const bool live = slotOk && lm.live && item < nblocksData;
    const bool dir = dirFlag != 0;

    if (threadIdx.x < 64) {
        reinterpret_cast<uint32_t *>(sSb)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gAesSbox)[threadIdx.x];
        reinterpret_cast<uint32_t *>(sRsb)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gAesRsbox)[threadIdx.x];
    }
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range && slotOk)
        fr = ft.range[lb];
    const bool cnt = live && lm.r == 0;
    Tally tl;

    // one 16-byte load per array feeds the replica's 16 byte registers (buffers are 16-byte aligned by contract)
    const uint64_t it = live ? item : 0;
    const uint4 sv = reinterpret_cast<const uint4 *>(states)[it];
    const uint4 kv = reinterpret_cast<const uint4 *>(keys)[it];
    uint32_t s[16], k[16];
    unpack4(sv.x, s + 0);
    unpack4(sv.y, s + 4);
    unpack4(sv.z, s + 8);
    unpack4(sv.w, s + 12);
    unpack4(kv.x, k + 0);
    unpack4(kv.y, k + 4);
    unpack4(kv.z, k + 8);
    unpack4(kv.w, k + 12);

    if (dir) { // :110-128 last encryption key, then the first AddRoundKey
#pragma unroll 1
        for (int rd = 0; rd < 10; ++rd)
            aes_key_fwd(k, sSb, rd);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s[i] ^= k[i];
    }
#pragma unroll 1
    for (int rd = 0; rd <= 10; ++rd) {
        if (fr.y) { // injector hook at the start of round rd (rd == 10: after the loop)
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.step != (uint32_t)rd || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                    continue;
                const uint32_t m = 1u << (df.bit & 31u);
                const int byteIdx = 4 * (df.index & 3) + (int)((df.bit & 31u) >> 3);
                const uint32_t bm = (m >> (8 * ((df.bit & 31u) >> 3))) & 0xffu;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i == byteIdx) {
                        if (df.site == SITE_AES_STATE)
                            s[i] ^= bm;
                        else if (df.site == SITE_AES_KEY)
                            k[i] ^= bm;
                    }
            }
        }
        if (rd == 10)
            break;
        aes_round(s, k, sSb, sRsb, dir, rd);
        if (syncEvery && rd < 9)
            aes_sync<NREP>(s, k, lm, cnt, tl);
    }
    if (!dir) { // :228-233 last AddRoundKey
#pragma unroll
        for (int i = 0; i < 16; ++i)
            s[i] ^= k[i];
    }
    aes_sync<NREP>(s, k, lm, cnt, tl); // in-place stores of state and key: store-data sync

    uint32_t detItems = 0;
    if (cnt) {
        reinterpret_cast<uint4 *>(states)[item] = make_uint4(pack4(s), pack4(s + 4), pack4(s + 8), pack4(s + 12));
        reinterpret_cast<uint4 *>(keys)[item] = make_uint4(pack4(k), pack4(k + 4), pack4(k + 8), pack4(k + 12));
        if (tl.det) { // unequal copies seen at a sync point of this block (DWC: detected, TMR: corrected)
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
}

// aes_enc_dec with its loops as written (TI_aes_128.c:107-235), for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the two loop counters
// `round` and `i` (unsigned char) are replica-private lane registers -- one lane per (block, replica), one sequential walk.  Sync
// points added to the frozen schedule, the reference's rule set for -TMR -noMemReplication on the source as written:
//   every evaluated branch condition: the loop conditions `round < 10`, `i < 16`, `i > 3`, `i < 4`, the tests of `dir`, the operands of
//     `(round > 0 && dir) || (round < 9 && !dir)` in short-circuit order                                    synchronization.cpp:146-155
//   every GEP with a variable index: state[i], key[i], key[i-4], state[buf4 + c], Rcon[round], Rcon[9-round], and the table lookups
//     sbox[..] / rsbox[..], whose index is DATA (loads: off with -noLoadSync; the state[] / key[] stores: off with -noStoreAddrSync);
//     constant indices (key[13], the ShiftRows moves) have nothing to vote (syncGEP returns early, :428-431)
// state[] and key[] stay what they are in the frozen schedule: replica-private (the lane's own 16 + 16 bytes of LDS here, because they
// are indexed at run time) until the function's exit, where their 8 dwords are voted as stored data; a voted (or, unvoted, replica
// 0's) offset selects the element every copy accesses.  Fault sites: SITE_AES_ROUND / _I of a replica (8 bits live), `step` = how
// many LOOP conditions the call has evaluated; SITE_AES_STATE / _KEY keep their meaning (start of main-loop iteration `step`, 10: after
// the loop).  A wild index reads 0 / stores nothing; a walk that a corrupted counter keeps alive is cut after 4096 loop conditions
// (a clean call evaluates 373 / 514).  Oracle: aes_item_indexed.  The sync-point-parity form of the kernel, not the throughput form.
template <int NREP>
__global__ __launch_bounds__(64) void aes128_indexed_kernel(uint8_t *__restrict__ states, uint8_t *__restrict__ keys,
                                                            uint64_t nblocksData, int dirFlag, Counters ctr, FaultTab ft,
                                                            uint8_t *__restrict__ detected)
{
    __shared__ __attribute__((aligned(16))) uint8_t sSb[256];
    __shared__ __attribute__((aligned(16))) uint8_t sRsb[256];
    __shared__ __attribute__((aligned(16))) uint8_t sSt[64 * 16];
    __shared__ __attribute__((aligned(16))) uint8_t sKy[64 * 16];
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    // COAST_F_LOCAL_STORE_SYNC: the data of every store of the -O0 IR -- round++ / i++, buf1..buf4, and every byte stored into state[] /
    // key[] in place (600 + 779 votes per encryption, 904 + 981 per decryption: tools/ir_sync_counts.py)
    const bool lss = xmr_local_sync_on(ctr.flags);
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nblocksData;
    const bool cnt = live && lm.r == 0;
    const bool d = dirFlag != 0;
    reinterpret_cast<uint32_t *>(sSb)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gAesSbox)[threadIdx.x];
    reinterpret_cast<uint32_t *>(sRsb)[threadIdx.x] = reinterpret_cast<const uint32_t *>(gAesRsbox)[threadIdx.x];
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    uint8_t *S = sSt + lm.lane * 16, *K = sKy + lm.lane * 16; // this lane's copy of state[] and key[]
    {
        const uint64_t it = live ? item : 0;
        *reinterpret_cast<uint4 *>(S) = reinterpret_cast<const uint4 *>(states)[it];
        *reinterpret_cast<uint4 *>(K) = reinterpret_cast<const uint4 *>(keys)[it];
    }
    __syncthreads();
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    Tally tl;
    uint32_t round = 0u, i = 0u, tick = 0u;
    if (lm.live) { // (the idle lane of a TMR wave has no block of its own: its replica group would wrap to lanes 0, 1)
        auto loopc = [&](uint32_t &reg, uint32_t limit, bool gt) __attribute__((always_inline)) { // one evaluated loop condition
            for (uint32_t q = 0; q < fr.y; ++q) { // the counters' upsets land right before the condition reads them
                const DevFault df = ft.list[fr.x + q];
                if (df.step != tick || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const uint32_t m = (1u << (df.bit & 31u)) & 0xffu;
                if (df.site == SITE_AES_ROUND)
                    round ^= m;
                else if (df.site == SITE_AES_I)
                    i ^= m;
            }
            if (tick >= 4096u)
                return false;
            ++tick;
            return xmr_steer<NREP>((gt ? reg > limit : reg < limit) ? 1u : 0u, lm, bs, cnt, tl) != 0u;
        };
        auto ifc = [&](bool c) __attribute__((always_inline)) { return xmr_steer<NREP>(c ? 1u : 0u, lm, bs, cnt, tl) != 0u; };
        auto ld = [&](const uint8_t *arr, int32_t idx) __attribute__((always_inline)) -> uint32_t {
            const uint32_t o = xmr_steer<NREP>((uint32_t)idx, lm, ls, cnt, tl);
            return o < 16u ? (uint32_t)arr[o] : 0u;
        };
        auto lsy = [&](uint32_t v) __attribute__((always_inline)) { return xmr_local_sync<NREP>(v, lm, lss, cnt, tl); };
        auto st = [&](uint8_t *arr, int32_t idx, uint32_t v) __attribute__((always_inline)) {
            const uint32_t o = xmr_steer<NREP>((uint32_t)idx, lm, ss, cnt, tl);
            const uint32_t dv = lsy(v & 0xffu); // the data of the in-place store
            if (o < 16u)
                arr[o] = (uint8_t)dv;
        };
        auto mov = [&](uint8_t *arr, int dst, uint32_t v) __attribute__((always_inline)) { arr[dst] = (uint8_t)lsy(v & 0xffu); }; // constant index
        auto tab = [&](const uint8_t *t, uint32_t size, uint32_t x) __attribute__((always_inline)) -> uint32_t {
            const uint32_t o = xmr_steer<NREP>(x, lm, ls, cnt, tl);
            return o < size ? (uint32_t)t[o] : 0u;
        };
        auto rcon = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
            const uint32_t o = xmr_steer<NREP>(x, lm, ls, cnt, tl);
            return o < 10u ? (uint32_t)kAesRcon[o] : 0u;
        };
        auto keyCore = [&](uint32_t rc) __attribute__((always_inline)) { // key[0..3] ^= sbox[key[13, 14, 15, 12]] (^ Rcon[rc])
            const uint32_t s0 = tab(sSb, 256u, K[13]) ^ rcon(rc);
            mov(K, 0, (uint32_t)K[0] ^ (s0 & 0xffu));
            mov(K, 1, (uint32_t)K[1] ^ tab(sSb, 256u, K[14]));
            mov(K, 2, (uint32_t)K[2] ^ tab(sSb, 256u, K[15]));
            mov(K, 3, (uint32_t)K[3] ^ tab(sSb, 256u, K[12]));
        };
        auto keyXor = [&]() __attribute__((always_inline)) { // key[i] = key[i] ^ key[i-4]
            const uint32_t a = ld(K, (int32_t)i), b = ld(K, (int32_t)i - 4);
            st(K, (int32_t)i, a ^ b);
        };
        auto dataHook = [&](uint32_t step) __attribute__((always_inline)) { // SITE_AES_STATE / _KEY: dword `index`, as in aes128_xmr_kernel
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.step != step || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const int byteIdx = 4 * (df.index & 3) + (int)((df.bit & 31u) >> 3);
                const uint8_t bm = (uint8_t)(1u << (df.bit & 7u));
                if (df.site == SITE_AES_STATE)
                    S[byteIdx] ^= bm;
                else if (df.site == SITE_AES_KEY)
                    K[byteIdx] ^= bm;
            }
        };
        (void)lsy(d ? 1u : 0u);                                              // the parameter `dir` into its alloca (-O0)
        if (ifc(d)) {                                                        // if (dir)                                  :111
            for (round = 0u; loopc(round, 10u, false); round = lsy((round + 1u) & 0xffu)) { // for (round = 0; round < 10; round++) :113
                keyCore(round);
                for (i = 4u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu))     //   for (i = 4; i < 16; i++)                :119
                    keyXor();
            }
            for (i = 0u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu)) {       // first AddRoundKey                         :125
                const uint32_t a = ld(S, (int32_t)i), b = ld(K, (int32_t)i);
                st(S, (int32_t)i, a ^ b);
            }
        }
        uint32_t iter = 0u;
        for (round = 0u; loopc(round, 10u, false); round = lsy((round + 1u) & 0xffu)) { // main loop                           :131
            dataHook(iter < 10u ? iter : 0xffffffffu);
            ++iter;
            if (ifc(d)) {                                                    //   if (dir): inverse key schedule          :132-141
                for (i = 15u; loopc(i, 3u, true); i = lsy((i - 1u) & 0xffu))
                    keyXor();
                keyCore((uint32_t)(9 - (int32_t)round));
            } else {
                for (i = 0u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu)) {   //   state[i] = sbox[state[i] ^ key[i]]      :143-146
                    const uint32_t a = ld(S, (int32_t)i), b = ld(K, (int32_t)i);
                    const uint32_t v = tab(sSb, 256u, a ^ b);
                    st(S, (int32_t)i, v);
                }
                uint32_t b1, b2;                                             //   shift rows: constant indices            :148-166
                b1 = lsy(S[1]), mov(S, 1, S[5]), mov(S, 5, S[9]), mov(S, 9, S[13]), mov(S, 13, b1);
                b1 = lsy(S[2]), b2 = lsy(S[6]), mov(S, 2, S[10]), mov(S, 6, S[14]), mov(S, 10, b1), mov(S, 14, b2);
                b1 = lsy(S[15]), mov(S, 15, S[11]), mov(S, 11, S[7]), mov(S, 7, S[3]), mov(S, 3, b1);
            }
            bool mix = false;                                                //   if ((round > 0 && dir) || (round < 9 && !dir)) :168
            if (ifc(round > 0u))
                mix = ifc(d);
            if (!mix && ifc(round < 9u))
                mix = ifc(!d);
            if (mix) {
                for (i = 0u; loopc(i, 4u, false); i = lsy((i + 1u) & 0xffu)) {    //   for (i = 0; i < 4; i++)                 :169
                    const int32_t b4 = (int32_t)lsy((i << 2) & 0xffu);       //     buf4 = (i << 2), an unsigned char
                    uint32_t buf1, buf2, buf3;
                    if (ifc(d)) {                                            //     if (dir): precompute                  :171-175
                        const uint32_t a0 = ld(S, b4), a2 = ld(S, b4 + 2);
                        buf1 = lsy(xtime(xtime(a0 ^ a2)));
                        const uint32_t a1 = ld(S, b4 + 1), a3 = ld(S, b4 + 3);
                        buf2 = lsy(xtime(xtime(a1 ^ a3)));
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) { // state[buf4 + cc] ^= buf: ONE GEP serves the load and the store of a compound
                            // assignment, and its first user is the load (user_back(), synchronization.cpp:341-351): load class
                            const uint32_t o = xmr_steer<NREP>((uint32_t)(b4 + cc), lm, ls, cnt, tl);
                            const uint32_t xv = lsy((o < 16u ? (uint32_t)S[o] : 0u) ^ (((cc & 1) ? buf2 : buf1) & 0xffu)); // ... the stored byte: a data vote
                            if (o < 16u)
                                S[o] = (uint8_t)xv;
                        }
                    }
                    {
                        const uint32_t a = ld(S, b4), b = ld(S, b4 + 1), cv = ld(S, b4 + 2), dv = ld(S, b4 + 3);
                        buf1 = lsy(a ^ b ^ cv ^ dv);                         //     the column's xor                      :177
                    }
                    buf2 = lsy(ld(S, b4));                                   //     buf2 = state[buf4]                    :178
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {                         //     the four rows                         :179-182
                        const uint32_t a = ld(S, b4 + cc);
                        const uint32_t b = cc < 3 ? ld(S, b4 + cc + 1) : buf2;
                        buf3 = lsy((a ^ b) & 0xffu);                         //     buf3 = state[..] ^ state[..]
                        buf3 = lsy(xtime(buf3));                             //     buf3 = galois_mul2(buf3)
                        const uint32_t a2 = ld(S, b4 + cc);
                        st(S, b4 + cc, a2 ^ buf3 ^ buf1);
                    }
                }
            }
            if (ifc(d)) {                                                    //   if (dir): inverse shift rows, rsbox     :187-211
                uint32_t b1, b2;
                b1 = lsy(S[13]), mov(S, 13, S[9]), mov(S, 9, S[5]), mov(S, 5, S[1]), mov(S, 1, b1);           // Row 1
                b1 = lsy(S[10]), b2 = lsy(S[14]), mov(S, 10, S[2]), mov(S, 14, S[6]), mov(S, 2, b1), mov(S, 6, b2); // Row 2
                b1 = lsy(S[3]), mov(S, 3, S[7]), mov(S, 7, S[11]), mov(S, 11, S[15]), mov(S, 15, b1);         // Row 3
                for (i = 0u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu)) {   //   state[i] = rsbox[state[i]] ^ key[i]     :208-211
                    const uint32_t a = ld(S, (int32_t)i);
                    const uint32_t x = tab(sRsb, 256u, a);
                    const uint32_t b = ld(K, (int32_t)i);
                    st(S, (int32_t)i, x ^ b);
                }
            } else {                                                         //   key schedule                            :213-226
                keyCore(round);
                for (i = 4u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu))
                    keyXor();
            }
        }
        dataHook(10u);
        if (ifc(!d))                                                         // if (!dir): last AddRoundKey               :228-233
            for (i = 0u; loopc(i, 16u, false); i = lsy((i + 1u) & 0xffu)) {
                const uint32_t a = ld(S, (int32_t)i), b = ld(K, (int32_t)i);
                st(S, (int32_t)i, a ^ b);
            }
        // the function's exit: state[] and key[] as stored data, 8 dwords (the frozen schedule's sync points)
        uint4 sv = *reinterpret_cast<const uint4 *>(S), kv = *reinterpret_cast<const uint4 *>(K);
        sv.x = xmr_store_sync<NREP>(sv.x, lm, cnt, tl);
        sv.y = xmr_store_sync<NREP>(sv.y, lm, cnt, tl);
        sv.z = xmr_store_sync<NREP>(sv.z, lm, cnt, tl);
        sv.w = xmr_store_sync<NREP>(sv.w, lm, cnt, tl);
        kv.x = xmr_store_sync<NREP>(kv.x, lm, cnt, tl);
        kv.y = xmr_store_sync<NREP>(kv.y, lm, cnt, tl);
        kv.z = xmr_store_sync<NREP>(kv.z, lm, cnt, tl);
        kv.w = xmr_store_sync<NREP>(kv.w, lm, cnt, tl);
        if (cnt) {
            reinterpret_cast<uint4 *>(states)[item] = sv;
            reinterpret_cast<uint4 *>(keys)[item] = kv;
        }
    }
    uint32_t detItems = 0;
    if (cnt && tl.det) { // unequal copies seen at a sync point of this block (DWC: detected, TMR: corrected)
        if (NREP == 2)
            detItems = 1;
        if (detected)
            detected[item] = 1;
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
