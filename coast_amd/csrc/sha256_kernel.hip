// sha256_kernel.hip -- protected sha256_hash for gfx950.
//
// Replaces: the TMR/DWC-transformed sha256_hash / sha256_transform of tests/sha256_common/sha256_common_tmr.c:27-179.
// Logical work item = one message; lane NREP*q + r holds replica r of the wave's q-th message: the eight ctx_state
// words, the working variables a..h and a rolling 16-word window of the message schedule all live in that lane's
// VGPRs (the reference's m[64] array is only ever read 16 words back, :49-63, so the window is the same dataflow).
// A wave owns one TILE of IPW = 64/NREP consecutive messages.  The message bytes are the single memory copy
// (-noMemReplication): the replicas of a message load the same addresses, which the memory pipeline serves from one
// fetch.  Sync points (frozen in oracle/coast_oracle.c): the 8 ctx_state words after every compression (they are
// stores, :90-97) and the 8 digest words before the store (:169-178).  VALU bound: ~1400 integer instructions per
// compression per replica (rotates are v_alignbit_b32, 3-input xor / ch / maj are one v_bitop3_b32 each).
//
// Two kernels:
//   sha256_fast_kernel     every tile of an aligned batch: 64 rounds fully unrolled with rotating register names, whole
//                          64-byte blocks loaded as 4 x dwordx4 per lane.  A tile that owns an armed fault (wave-uniform test
//                          on the injector's range table) runs its compressions a round at a time with the injector hooks
//                          instead -- inside this kernel, so the flipped register meets this kernel's own sync points, lane
//                          map, counter gate and stores (round 2 handed such tiles to the kernel below and the voter here
//                          never saw unequal copies).
//   sha256_general_kernel  message arrays that are not 4-byte aligned, -noStoreDataSync, and the byte loop with its counters
//                          inside the sphere of replication (COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC).
#include "xmr.hpp"

namespace coast {

enum { SITE_SHA_M = 8, SITE_SHA_WV = 9, SITE_SHA_STATE = 10, SITE_SHA_DATALEN = 11, SITE_SHA_I = 12 };

__constant__ uint32_t kShaK[64] = { // FIPS 180-4 section 4.2.2; sha256_common_tmr.c:8-19
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

// the same constants as compile-time values for the unrolled kernel (they become SGPR/literal operands)
#define SHA_K_LIST                                                                                               \
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,      \
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,      \
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,      \
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,      \
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,      \
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,      \
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,      \
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); // a ^ b ^ c in one VALU op
}
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g)
{
    return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); // (e & f) ^ (~e & g)  == e ? f : g
}
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); // (a & b) ^ (a & c) ^ (b & c)
}

// word t (big-endian) of padded block c of a `len`-byte message (padding exactly as sha256_hash :129-164)
__device__ __forceinline__ uint32_t sha_word(const uint8_t *msg, uint32_t len, uint32_t c, int t, bool aligned,
                                             bool lastBlock)
{
    const uint32_t p = c * 64u + 4u * (uint32_t)t;
    uint32_t w;
    if (p + 4u <= len) {
        if (aligned) {
            w = bswap32(*reinterpret_cast<const uint32_t *>(msg + p));
        } else {
            w = ((uint32_t)msg[p] << 24) | ((uint32_t)msg[p + 1] << 16) | ((uint32_t)msg[p + 2] << 8) |
                (uint32_t)msg[p + 3];
        }
    } else {
        w = 0u;
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t pb = p + b;
            const uint32_t byte = (pb < len) ? (uint32_t)msg[pb] : ((pb == len) ? 0x80u : 0u);
            w = (w << 8) | byte;
        }
    }
    if (lastBlock) { // bit length appended as two u32 (DBL_INT_ADD :2-5, :152-160) == 64-bit big-endian len*8
        const uint64_t bits = (uint64_t)len * 8ull;
        if (t == 14)
            w = (uint32_t)(bits >> 32);
        if (t == 15)
            w = (uint32_t)bits;
    }
    return w;
}

// ------------------------------------------------------------------------------------------------ fast path
// sha256_transform (:27-98), 64 rounds unrolled; the working variables rotate by renaming, not by moves.
__device__ __forceinline__ void sha_compress_unrolled(uint32_t st[8], uint32_t m[16])
{
    constexpr uint32_t K[64] = {SHA_K_LIST};
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#define SHA_STEP(A, B, C, D, E, F, G, H, T)                                                                      \
    do {                                                                                                        \
        if ((T) >= 16) { /* schedule expansion :49-63 on the rolling window */                                  \
            const uint32_t x_ = m[((T) + 14) & 15], y_ = m[((T) + 1) & 15];                                     \
            const uint32_t s1_ = xor3(rotr32(x_, 17), rotr32(x_, 19), x_ >> 10);                                \
            const uint32_t s0_ = xor3(rotr32(y_, 7), rotr32(y_, 18), y_ >> 3);                                  \
            m[(T) & 15] = m[(T) & 15] + s0_ + (m[((T) + 9) & 15] + s1_);                                        \
        }                                                                                                       \
        const uint32_t ep1_ = xor3(rotr32(E, 6), rotr32(E, 11), rotr32(E, 25));                                 \
        const uint32_t t1_ = (H + ep1_ + sha_ch(E, F, G)) + (K[(T)] + m[(T) & 15]);                             \
        const uint32_t ep0_ = xor3(rotr32(A, 2), rotr32(A, 13), rotr32(A, 22));                                 \
        D += t1_;                                                                                               \
        H = t1_ + ep0_ + sha_maj(A, B, C);                                                                      \
    } while (0)
#define SHA_STEP8(T)                                                                                             \
    SHA_STEP(a, b, c, d, e, f, g, h, (T) + 0);                                                                  \
    SHA_STEP(h, a, b, c, d, e, f, g, (T) + 1);                                                                  \
    SHA_STEP(g, h, a, b, c, d, e, f, (T) + 2);                                                                  \
    SHA_STEP(f, g, h, a, b, c, d, e, (T) + 3);                                                                  \
    SHA_STEP(e, f, g, h, a, b, c, d, (T) + 4);                                                                  \
    SHA_STEP(d, e, f, g, h, a, b, c, (T) + 5);                                                                  \
    SHA_STEP(c, d, e, f, g, h, a, b, (T) + 6);                                                                  \
    SHA_STEP(b, c, d, e, f, g, h, a, (T) + 7)
    SHA_STEP8(0);
    SHA_STEP8(8);
    SHA_STEP8(16);
    SHA_STEP8(24);
    SHA_STEP8(32);
    SHA_STEP8(40);
    SHA_STEP8(48);
    SHA_STEP8(56);
#undef SHA_STEP8
#undef SHA_STEP
    st[0] += a;
    st[1] += b;
    st[2] += c;
    st[3] += d;
    st[4] += e;
    st[5] += f;
    st[6] += g;
    st[7] += h;
}

// ------------------------------------------------------------------------------------------------ injector hooks
#define SHA_ROUND(T, MW)                                                                                        \
    do {                                                                                                        \
        const uint32_t ep0_ = rotr32(v[0], 2) ^ rotr32(v[0], 13) ^ rotr32(v[0], 22);                            \
        const uint32_t ep1_ = rotr32(v[4], 6) ^ rotr32(v[4], 11) ^ rotr32(v[4], 25);                            \
        const uint32_t ch_ = (v[4] & v[5]) ^ (~v[4] & v[6]);                                                    \
        const uint32_t maj_ = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);                                    \
        const uint32_t t1_ = v[7] + ep1_ + ch_ + kShaK[(T)] + (MW);                                             \
        const uint32_t t2_ = ep0_ + maj_;                                                                       \
        v[7] = v[6];                                                                                            \
        v[6] = v[5];                                                                                            \
        v[5] = v[4];                                                                                            \
        v[4] = v[3] + t1_;                                                                                      \
        v[3] = v[2];                                                                                            \
        v[2] = v[1];                                                                                            \
        v[1] = v[0];                                                                                            \
        v[0] = t1_ + t2_;                                                                                       \
    } while (0)

// sha256_transform (:27-98), a round at a time with the injector hooks.  The reference produces all 64 schedule words
// before round 0; here word t is produced (and an M hook of step t applied) right before round t, and a working-variable
// hook is applied before round t: every value that depends on a flipped word is computed after the flip in both
// orders, so the dataflow is identical to the oracle's.
__device__ __forceinline__ void sha_compress_hooked(uint32_t st[8], uint32_t m[16], uint32_t cidx, const FaultTab &ft,
                                                    uint2 fr, int slot, int rep, bool laneLive)
{
    uint32_t v[8];
#pragma unroll
    for (int w = 0; w < 8; ++w)
        v[w] = st[w];
#pragma unroll 1
    for (int t16 = 0; t16 < 64; t16 += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = t16 + u;
            if (t16 > 0) {
                const uint32_t x = m[(u + 14) & 15], y = m[(u + 1) & 15];
                const uint32_t s1 = rotr32(x, 17) ^ rotr32(x, 19) ^ (x >> 10);
                const uint32_t s0 = rotr32(y, 7) ^ rotr32(y, 18) ^ (y >> 3);
                m[u] = s1 + m[(u + 9) & 15] + s0 + m[u];
            }
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if ((int)df.local != slot || (int)df.replica != rep || !laneLive ||
                    df.step != cidx * 64u + (uint32_t)t)
                    continue;
                const uint32_t mask = 1u << (df.bit & 31u);
                if (df.site == SITE_SHA_M) {
                    m[u] ^= mask;
                } else if (df.site == SITE_SHA_WV) {
#pragma unroll
                    for (int w = 0; w < 8; ++w)
                        if (w == (df.index & 7))
                            v[w] ^= mask;
                }
            }
            SHA_ROUND(t, m[u]);
        }
    }
#pragma unroll
    for (int w = 0; w < 8; ++w)
        st[w] += v[w];
}

__device__ __forceinline__ void sha_state_hook(uint32_t st[8], uint32_t step, const FaultTab &ft, uint2 fr, int slot,
                                               int rep, bool laneLive)
{
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault df = ft.list[fr.x + q];
        if (df.site != SITE_SHA_STATE || df.step != step || (int)df.local != slot || (int)df.replica != rep ||
            !laneLive)
            continue;
#pragma unroll
        for (int w = 0; w < 8; ++w)
            if (w == (df.index & 7))
                st[w] ^= 1u << (df.bit & 31u);
    }
}

// one wave per tile; requires 4-byte aligned message rows (stride % 4 == 0, base % 4 == 0); VEC16: rows 16-byte aligned
template <int NREP, bool VEC16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void sha256_fast_kernel(const uint8_t *__restrict__ msgs, size_t stride, uint32_t len,
                                                          uint64_t nmsgs, uint8_t *__restrict__ digests,
                                                          uint64_t ntiles, Counters ctr, FaultTab ft,
                                                          uint8_t *__restrict__ detected, size_t copyIn = 0, size_t copyOut = 0)
{
    // copyIn / copyOut != 0: COAST_F_MEMORY_COPIES -- the message array and the digest array are NREP copies back to back (that many
    // bytes apart); replica r loads from copy r and stores the voted digest into copy r (the reference's memory-replicated mode with
    // -storeDataSync: dataflowProtection.cpp:14-18, synchronization.cpp:197-224).  0: one memory copy, replica 0 stores.
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool skip = tile >= ntiles;
    uint2 fr = make_uint2(0u, 0u); // this tile's armed faults (the same for the whole wave): {first, count} in ft.list
    if (!skip && ft.range) {
        fr = ft.range[tile]; // left in vector registers: the hooks' loop state would otherwise cost the unrolled rounds SGPRs
    }
    const uint64_t item = tile * IPW + (uint64_t)lm.q;
    const bool live = !skip && lm.live && item < nmsgs;
    const bool cnt = live && lm.r == 0;
    const uint8_t *msg = msgs + (size_t)lm.r * copyIn + (live ? item : 0) * stride;

    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                      0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}; // :107-114
    Tally tl;
    const uint32_t nfull = len >> 6, rem = len & 63u;
    const uint32_t ncomp = nfull + (rem < 56u ? 1u : 2u);

    for (uint32_t c = 0; c < ncomp; ++c) {
        uint32_t m[16];
        if (c < nfull) { // a whole data block
            if (VEC16) {
                const uint4 *src = reinterpret_cast<const uint4 *>(msg + (size_t)c * 64);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const uint4 q = src[v];
                    m[4 * v + 0] = bswap32(q.x);
                    m[4 * v + 1] = bswap32(q.y);
                    m[4 * v + 2] = bswap32(q.z);
                    m[4 * v + 3] = bswap32(q.w);
                }
            } else {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(msg + (size_t)c * 64);
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    m[t] = bswap32(src[t]);
            }
        } else { // tail / padding blocks (:129-164).  A message of k * 64 bytes ends with a block that holds no data: its 16
                 // words and 48 schedule words depend on `len` alone, yet they are computed here, per replica lane, like
                 // every other block -- in the reference that expansion sits inside the triplicated sha256_transform (:49-63),
                 // so it belongs inside the sphere of replication (round 1 expanded it once on the host).
            const bool last = (c + 1u == ncomp);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                m[t] = sha_word(msg, len, c, t, true, last);
        }
        if (fr.y != 0u) { // a tile with armed upsets: the same compression a round at a time, flips applied where they are due
            sha_state_hook(st, c, ft, fr, lm.q, lm.r, lm.live);
            sha_compress_hooked(st, m, c, ft, fr, lm.q, lm.r, lm.live);
        } else {
            sha_compress_unrolled(st, m);
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) // ctx_state[w] += ... are stores: store-data sync
            st[w] = xmr_sync<NREP>(st[w], lm, cnt, tl);
    }
    if (fr.y != 0u)
        sha_state_hook(st, ncomp, ft, fr, lm.q, lm.r, lm.live); // step == ncompress: a ctx_state word before the digest
    uint32_t dg[8];
#pragma unroll
    for (int w = 0; w < 8; ++w)
        dg[w] = bswap32(xmr_sync<NREP>(st[w], lm, cnt, tl)); // digest words voted before the store (:169-178)

    uint32_t detItems = 0;
    if (cnt || (live && copyOut != 0)) { // one memory copy: the original store (replica 0); memory copies: every replica into its own
        uint8_t *out = digests + (size_t)lm.r * copyOut + item * 32u;
        if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0u) {
            reinterpret_cast<uint4 *>(out)[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]);
            reinterpret_cast<uint4 *>(out)[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    out[4 * w + b] = (uint8_t)(dg[w] >> (8 * b));
        }
    }
    if (cnt && tl.det) { // unequal copies seen at a sync point of this message (DWC: detected, TMR: corrected)
        if (NREP == 2)
            detItems = 1;
        if (detected)
            detected[item] = 1;
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ general path
// sha256_hash with its byte loop as written (sha256_common_tmr.c:119-127), for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the loop
// counter `i` and `ctx_datalen` are replica-private registers of the lane; ctx_data[64] is memory -- one copy per message, in
// LDS, written by the original store (replica 0's lane) at the voted or at its own offset; ctx_bitlen likewise.  Mirrors
// oracle/coast_oracle.c:sha_item_indexed statement by statement (bounded wild accesses, watchdog).  The replicas of a message
// always take the same (voted, or replica 0's) direction, so they stay convergent; different messages may diverge.
template <int NREP>
__device__ void sha_item_indexed(const uint8_t *msg, uint32_t len, uint32_t st[8], const LaneMap<NREP> &lm, bool laneLive,
                                 bool cnt, Tally &tl, const FaultTab &ft, uint2 fr, int slot, uint32_t flags, uint8_t *buf)
{
    const bool bs = (flags & kFlagBranchSync) != 0u, as = (flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(flags & kFlagNoLoadSync), ss = as && !(flags & kFlagNoStoreAddrSync);
    const bool writer = laneLive && lm.r == 0; // the single memory copy is written by the original instruction
    uint32_t ir = 0u, dl = 0u, cidx = 0u, bl0 = 0u, bl1 = 0u;
    const uint32_t cap = 4u * len + 256u;
    uint32_t *buf32 = reinterpret_cast<uint32_t *>(buf);
    if (writer) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            buf32[t] = 0u;
    }
    auto transform = [&]() __attribute__((always_inline)) {
        uint32_t m[16];
        wave_lds_sync(); // ctx_data was written by replica 0's lane: the other replicas' loads must not overtake its stores
#pragma unroll
        for (int t = 0; t < 16; ++t)
            m[t] = bswap32(buf32[t]);
        wave_lds_sync(); // ... and its next stores must not overtake these loads
        sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live);
        sha_compress_hooked(st, m, cidx, ft, fr, slot, lm.r, lm.live);
#pragma unroll
        for (int w = 0; w < 8; ++w)
            st[w] = xmr_store_sync<NREP>(st[w], lm, cnt, tl);
        ++cidx;
    };
    for (uint32_t it = 0;; ++it) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step != it || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            if (df.site == SITE_SHA_I)
                ir ^= 1u << (df.bit & 31u);
            else if (df.site == SITE_SHA_DATALEN)
                dl ^= 1u << (df.bit & 31u);
        }
        if (!xmr_steer<NREP>(ir < len ? 1u : 0u, lm, bs, cnt, tl) || it >= cap) // for (i = 0; i < len; ++i)        :119
            break;
        const uint32_t li = xmr_steer<NREP>(ir, lm, ls, cnt, tl);                // data[i]                          :120
        const uint32_t byte = li < len ? (uint32_t)msg[li] : 0u;
        const uint32_t si = xmr_steer<NREP>(dl, lm, ss, cnt, tl);                // ctx_data[ctx_datalen] = ...      :120
        if (writer && si < 64u)
            buf[si] = (uint8_t)byte;
        dl += 1u;                                                                // ctx_datalen++                    :121
        if (xmr_steer<NREP>(dl == 64u ? 1u : 0u, lm, bs, cnt, tl)) {             // if (ctx_datalen == 64)           :122
            transform();
            bl1 += (bl0 > 0xffffffffu - 512u) ? 1u : 0u;
            bl0 += 512u;
            dl = 0u;                                                             // ctx_datalen = 0                  :125
        }
        ir += 1u;
    }
    const bool shortPad = xmr_steer<NREP>(dl < 56u ? 1u : 0u, lm, bs, cnt, tl) != 0u; // if (ctx_datalen < 56)      :132
    const uint32_t pi = xmr_steer<NREP>(dl, lm, ss, cnt, tl);                    // ctx_data[i++] = 0x80, i = ctx_datalen
    if (writer) {
        if (pi < 64u)
            buf[pi] = 0x80u;
        for (uint32_t k = pi + 1u; k < (shortPad ? 56u : 64u); ++k)              // the zero fill: a memset after -O3
            buf[k] = 0u;
    }
    if (!shortPad) {
        transform();
        if (writer) {
#pragma unroll
            for (int t = 0; t < 14; ++t)
                buf32[t] = 0u;
        }
    }
    // DBL_INT_ADD(ctx_bitlen[0], ctx_bitlen[1], ctx_datalen * 8): a replicated value stored into the single ctx_bitlen   :150
    uint32_t add = xmr_store_sync<NREP>(dl * 8u, lm, cnt, tl);
    if (NREP != 3 || !lm.storeSync)
        add = xmr_rep0<NREP>(add, lm);
    bl1 += (bl0 > 0xffffffffu - add) ? 1u : 0u;
    bl0 += add;
    if (writer) {
        buf32[14] = bswap32(bl1);
        buf32[15] = bswap32(bl0);
    }
    transform();
    sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live); // step == ncompress: before the digest
}

// ------------------------------------------------------------------------------------------------ the -O0 shape
// sha256_hash + sha256_transform as the x86 / lli flow hands them to the pass (tests/sha256_common/Makefile: OPT_FLAGS empty -- the -O0 IR
// of sha256_common_tmr.c:27-178), for COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC | COAST_F_O0_SHAPE: every loop is a loop -- the byte loop,
// `while (i < 56 / 64)`, the long pad's `while (n--)`, the output loop, the three loops of sha256_transform -- with replica-private
// counters; every evaluated condition is a branch vote, every variable-index GEP an offset vote (3-byte message: 198 / 387 / 152,
// tools/ir_sync_counts.py).  COAST_F_LOCAL_STORE_SYNC adds the store-data votes of that IR (2009 into locals + 116 into memory at 3
// bytes); the digest then leaves as its 32 byte stores.  One lane per (message, replica), one sequential walk; ctx_data[64] is one LDS
// copy per message written by the original store (replica 0's lane), m[64] is the lane's own (64 words of LDS, indexed at run time).
// Mirrors oracle/coast_oracle.c:sha_item_o0 / sh0_transform statement by statement.  The sync-point-parity form, not the throughput form.
template <int NREP>
__global__ __launch_bounds__(64) void sha256_o0_kernel(const uint8_t *__restrict__ msgs, size_t stride, uint32_t lenArg, uint64_t nmsgs,
                                                       uint8_t *__restrict__ digests, Counters ctr, FaultTab ft,
                                                       uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    __shared__ __attribute__((aligned(16))) uint8_t sCtxData[IPW + 1][64];
    __shared__ uint32_t sW[64 * 64]; // m[t] of lane l at t * 64 + l
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    const bool lss = xmr_local_sync_on(ctr.flags);
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nmsgs;
    const bool cnt = live && lm.r == 0;
    const bool writer = cnt; // the single memory copies (ctx_data, hash) are written by the original instruction
    const uint8_t *msg = msgs + (live ? item : 0) * stride;
    const uint32_t len = live ? lenArg : 0u;
    uint8_t *buf = sCtxData[slot];
    uint32_t *W = sW + lm.lane;
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    if (writer)
        for (int t = 0; t < 16; ++t)
            reinterpret_cast<uint32_t *>(buf)[t] = 0u;
    __syncthreads();
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    Tally tl;
    // (the idle lane of a TMR wave has no message of its own -- its replica group would wrap to lanes 0, 1: it walks on its own values)
    auto br = [&](bool c) __attribute__((always_inline)) { return lm.live ? xmr_steer<NREP>(c ? 1u : 0u, lm, bs, cnt, tl) != 0u : c; };
    auto off = [&](uint32_t idx, bool store) __attribute__((always_inline)) {
        return lm.live ? xmr_steer<NREP>(idx, lm, store ? ss : ls, cnt, tl) : idx;
    };
    auto lsy = [&](uint32_t v) __attribute__((always_inline)) { return lm.live ? xmr_local_sync<NREP>(v, lm, lss, cnt, tl) : v; };
    auto ssy = [&](uint32_t v) __attribute__((always_inline)) { return lm.live ? xmr_store_sync<NREP>(v, lm, cnt, tl) : v; };
    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint32_t cidx = 0u;
    auto mFault = [&](uint32_t t) __attribute__((always_inline)) { // SITE_SHA_M: m[t] right after it is produced
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.site == SITE_SHA_M && df.step == cidx * 64u + t && (int)df.local == slot && (int)df.replica == lm.r && lm.live)
                W[t * 64u] ^= 1u << (df.bit & 31u);
        }
    };
    auto transform = [&]() __attribute__((noinline)) {
        wave_lds_sync(); // ctx_data was written by replica 0's lane: the other replicas' loads must not overtake its stores
        for (int t = 0; t < 64; ++t)
            W[t * 64] = 0u;
        uint32_t i = 0u, j = 0u, temp = 0u;
        for (;;) {                                                       // for (i = 0, j = 0; i < 16; ++i, j += 4)   :34
            if (!br(i < 16u))
                break;
#pragma unroll 1
            for (uint32_t b = 0; b < 4u; ++b) {                          //   temp = data[j] << 24; temp |= ..        :35-38
                const uint32_t o = off(j + b, false);
                const uint32_t byte = o < 64u ? (uint32_t)buf[o] : 0u;
                temp = lsy((b ? temp : 0u) | (byte << (24u - 8u * b)));
            }
            const uint32_t os = off(i, true);                            //   m[i] = temp                              :39
            temp = lsy(temp);
            if (os < 64u) {
                W[os * 64u] = temp;
                mFault(os);
            }
            i = lsy(i + 1u);
            j = lsy(j + 4u);
        }
        for (;;) {                                                       // for (; i < 64; ++i)                       :42
            if (!br(i < 64u))
                break;
            uint32_t o = off(i - 2u, false);
            uint32_t sv = lsy(o < 64u ? W[o * 64u] : 0u);                //   s = m[i - 2]                            :43
            uint32_t sig1 = lsy(rotr32(sv, 17));
            sig1 = lsy(sig1 ^ rotr32(sv, 19));
            sig1 = lsy(sig1 ^ (sv >> 10));
            o = off(i - 15u, false);
            sv = lsy(o < 64u ? W[o * 64u] : 0u);                         //   s = m[i - 15]                           :48
            uint32_t sig0 = lsy(rotr32(sv, 7));
            sig0 = lsy(sig0 ^ rotr32(sv, 18));
            sig0 = lsy(sig0 ^ (sv >> 3));
            temp = lsy(sig1);                                            //   temp = sig1; += m[i-7]; += sig0; += m[i-16]  :53-56
            o = off(i - 7u, false);
            temp = lsy(temp + (o < 64u ? W[o * 64u] : 0u));
            temp = lsy(temp + sig0);
            o = off(i - 16u, false);
            temp = lsy(temp + (o < 64u ? W[o * 64u] : 0u));
            const uint32_t os = off(i, true);                            //   m[i] = temp                              :57
            temp = lsy(temp);
            if (os < 64u) {
                W[os * 64u] = temp;
                mFault(os);
            }
            i = lsy(i + 1u);
        }
        uint32_t v[8];
#pragma unroll
        for (int w = 0; w < 8; ++w)                                      // a = ctx_state[0] ..                       :60-67
            v[w] = lsy(st[w]);
        i = 0u;
        for (uint32_t t = 0;; ++t) {                                     // for (i = 0; i < 64; ++i)                  :69
            if (!br(i < 64u) || t >= 64u)
                break;
            for (uint32_t q = 0; q < fr.y; ++q) {                        // SITE_SHA_WV: a..h before round t
                const DevFault df = ft.list[fr.x + q];
                if (df.site != SITE_SHA_WV || df.step != cidx * 64u + t || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                    continue;
#pragma unroll
                for (int w = 0; w < 8; ++w)
                    if (w == (df.index & 7))
                        v[w] ^= 1u << (df.bit & 31u);
            }
            uint32_t ep0 = lsy(rotr32(v[0], 2));
            ep0 = lsy(ep0 ^ rotr32(v[0], 13));
            ep0 = lsy(ep0 ^ rotr32(v[0], 22));
            uint32_t ep1 = lsy(rotr32(v[4], 6));
            ep1 = lsy(ep1 ^ rotr32(v[4], 11));
            ep1 = lsy(ep1 ^ rotr32(v[4], 25));
            const uint32_t ch = lsy((v[4] & v[5]) ^ (~v[4] & v[6]));
            const uint32_t maj = lsy((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            const uint32_t ok = off(i, false), om = off(i, false);       //   k[i], m[i]                               :78
            const uint32_t t1 = lsy(v[7] + ep1 + ch + (ok < 64u ? kShaK[ok] : 0u) + (om < 64u ? W[om * 64u] : 0u));
            const uint32_t t2 = lsy(ep0 + maj);
            v[7] = lsy(v[6]);                                            //   h = g .. a = t1 + t2                    :80-87
            v[6] = lsy(v[5]);
            v[5] = lsy(v[4]);
            v[4] = lsy(v[3] + t1);
            v[3] = lsy(v[2]);
            v[2] = lsy(v[1]);
            v[1] = lsy(v[0]);
            v[0] = lsy(t1 + t2);
            i = lsy(i + 1u);
        }
#pragma unroll
        for (int w = 0; w < 8; ++w)                                      // ctx_state[w] += ..: stored                :90-97
            st[w] = ssy(st[w] + v[w]);
        wave_lds_sync(); // ... and replica 0's next ctx_data stores must not overtake the loads above
        ++cidx;
    };
    uint32_t ir = 0u, dl = 0u, bl0 = 0u, bl1 = 0u;
    const uint32_t cap = 4u * len + 256u;
    (void)lsy(len);                                                      // the parameter `len` into its alloca
    for (uint32_t it = 0;; ++it) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step != it || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            if (df.site == SITE_SHA_I)
                ir ^= 1u << (df.bit & 31u);
            else if (df.site == SITE_SHA_DATALEN)
                dl ^= 1u << (df.bit & 31u);
        }
        if (!br(ir < len) || it >= cap)                                  // for (i = 0; i < len; ++i)                    :119
            break;
        const uint32_t li = off(ir, false);                              // data[i]                                       :120
        const uint32_t byte = li < len ? (uint32_t)msg[li] : 0u;
        const uint32_t si = off(dl, true);                               // ctx_data[ctx_datalen] = ...                   :120
        const uint32_t bv = lsy(byte);
        if (writer && si < 64u)
            buf[si] = (uint8_t)bv;
        dl = lsy(dl + 1u);                                               // ctx_datalen++                                 :121
        if (br(dl == 64u)) {                                             // if (ctx_datalen == 64)                        :122
            sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live);
            transform();
            if (br(bl0 > 0xffffffffu - 512u))                            // DBL_INT_ADD(ctx_bitlen[0], ctx_bitlen[1], 512)  :2-5
                bl1 = lsy(bl1 + 1u);
            bl0 = lsy(bl0 + 512u);
            dl = 0u;                                                     // ctx_datalen = 0                               :125
        }
        ir = lsy(ir + 1u);
    }
    ir = lsy(dl);                                                        // i = ctx_datalen                               :129
    const bool shortPad = br(dl < 56u);                                  // if (ctx_datalen < 56)                         :132
    {
        const uint32_t lim = shortPad ? 56u : 64u;
        uint32_t o = off(ir, true);                                      // ctx_data[i++] = 0x80                       :133,137
        if (writer && o < 64u)
            buf[o] = 0x80u;
        ir = lsy(ir + 1u);
        for (uint32_t guard = 0;; ++guard) {                             // while (i < 56 / 64) ctx_data[i++] = 0x00   :134,138
            if (!br(ir < lim) || guard >= 256u)
                break;
            o = off(ir, true);
            if (writer && o < 64u)
                buf[o] = 0u;
            ir = lsy(ir + 1u);
        }
    }
    if (!shortPad) {
        sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live);
        transform();
        uint32_t n = 56u;                                                // the inlined sha_memset(ctx_data, 0, 56)    :143-150
        (void)lsy(0u);                                                   // c = c & 0xFF
        for (uint32_t p = 0;; ++p) {
            const uint32_t old = n;
            n = lsy(n - 1u);                                             // while (n--): load, decrement, store, branch
            if (!br(old != 0u) || p >= 256u)
                break;
            (void)lsy(0u);                                               // *p++ = c: a constant-offset GEP, the stored c
            if (writer && p < 64u)
                buf[p] = 0u;
        }
    }
    {
        const uint32_t add = dl * 8u;                                    // DBL_INT_ADD(..., ctx_datalen * 8)             :150
        if (br(bl0 > 0xffffffffu - add))
            bl1 = lsy(bl1 + 1u);
        uint32_t sum = ssy(bl0 + add);                                   // a += c: a replicated value into the single ctx_bitlen[0]
        if (NREP != 3 || !lm.storeSync)
            sum = lm.live ? xmr_rep0<NREP>(sum, lm) : sum;
        bl0 = sum;
    }
#pragma unroll 1
    for (uint32_t b = 0; b < 4u; ++b) {                                  // ctx_data[63 - b] = ctx_bitlen[0] >> 8 b ..  :151-158
        const uint32_t x = lsy((bl0 >> (8u * b)) & 0xffu);
        if (writer)
            buf[63u - b] = (uint8_t)x;
    }
#pragma unroll 1
    for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t x = lsy((bl1 >> (8u * b)) & 0xffu);
        if (writer)
            buf[59u - b] = (uint8_t)x;
    }
    sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live);
    transform();
    sha_state_hook(st, cidx, ft, fr, slot, lm.r, lm.live); // step == ncompress: before the digest
    if (!lss) {
#pragma unroll
        for (int w = 0; w < 8; ++w)                                      // the digest: 8 words (the default schedule's exit votes)
            st[w] = ssy(st[w]);
    }
    uint8_t *out = digests + (live ? item : 0) * 32u;
    ir = 0u;
    for (uint32_t guard = 0;; ++guard) {                                 // for (i = 0; i < 4; ++i) hash[i + 4 w] = ..  :164-173
        if (!br(ir < 4u) || guard >= 4u)
            break;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const uint32_t o = off(ir + 4u * (uint32_t)w, true);
            const uint32_t x = lsy((st[w] >> ((24u - ir * 8u) & 31u)) & 0xffu);
            if (writer && o < 32u)
                out[o] = (uint8_t)x;
        }
        ir = lsy(ir + 1u);
    }
    uint32_t detItems = 0;
    if (cnt && tl.det) {
        if (NREP == 2)
            detItems = 1;
        if (detected)
            detected[item] = 1;
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

// one wave (64-thread workgroup) per tile
template <int NREP>
__global__ __launch_bounds__(64) void sha256_general_kernel(const uint8_t *__restrict__ msgs, size_t stride, uint32_t len,
                                                            uint64_t nmsgs, uint8_t *__restrict__ digests,
                                                            Counters ctr, FaultTab ft,
                                                            const uint32_t *__restrict__ tileList,
                                                            uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    __shared__ __attribute__((aligned(16))) uint8_t sCtxData[IPW + 1][64]; // indexed mode: ctx_data of every message of the tile
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const uint32_t tile = tileList ? tileList[blockIdx.x] : blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nmsgs;
    const uint8_t *msg = msgs + (live ? item : 0) * stride;
    const bool aligned = ((stride & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(msgs) & 3u) == 0u);

    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];

    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                      0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    Tally tl;
    const bool cnt = live && lm.r == 0;
    const uint32_t rem = len & 63u;
    const uint32_t ncomp = (len >> 6) + (rem < 56u ? 1u : 2u);

    if (ctr.flags & kFlagIndexed) {
        sha_item_indexed<NREP>(msg, live ? len : 0u, st, lm, live, cnt, tl, ft, fr, slot, ctr.flags, sCtxData[slot]);
    } else {
        for (uint32_t c = 0; c < ncomp; ++c) {
            uint32_t m[16];
            const bool last = (c + 1u == ncomp);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                m[t] = sha_word(msg, len, c, t, aligned, last);
            sha_state_hook(st, c, ft, fr, slot, lm.r, lm.live); // ctx_state hook before compression c
            sha_compress_hooked(st, m, c, ft, fr, slot, lm.r, lm.live);
#pragma unroll
            for (int w = 0; w < 8; ++w)
                st[w] = xmr_store_sync<NREP>(st[w], lm, cnt, tl);
        }
        sha_state_hook(st, ncomp, ft, fr, slot, lm.r, lm.live); // step == ncompress: before the digest
    }
    uint32_t dg[8];
#pragma unroll
    for (int w = 0; w < 8; ++w)
        dg[w] = bswap32(xmr_store_sync<NREP>(st[w], lm, cnt, tl));

    uint32_t detItems = 0;
    if (cnt) {
        uint8_t *out = digests + item * 32u;
        if ((reinterpret_cast<uintptr_t>(digests) & 15u) == 0u) {
            reinterpret_cast<uint4 *>(out)[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]);
            reinterpret_cast<uint4 *>(out)[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
        } else {
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    out[4 * w + b] = (uint8_t)(dg[w] >> (8 * b));
        }
        if (tl.det) { // unequal copies seen at a sync point of this message (DWC: detected, TMR: corrected)
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
