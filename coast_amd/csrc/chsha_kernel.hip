// chsha_kernel.hip -- protected CHStone sha (tests/chstone/sha/sha.c: sha_init / sha_update / sha_final; the benchmark of
// unittest/cfg/full.yml:5), a batch of messages.
//
// Not FIPS SHA-1: the message schedule has no rotate (W[i] = W[i-3]^W[i-8]^W[i-14]^W[i-16], sha.c:92-94) and the sixteen
// input words are assembled LITTLE-endian by the file's own memcpy (:61-80) -- on a little-endian GPU that is a plain
// dword load.  sha_final (:153-172) indexes the word array with a byte count; it pads as intended only when the length is a
// multiple of 64 (word 0 = 0x80, 14 = bit count high, 15 = bit count low), which is all the benchmark does (2 x 8192
// bytes) and all this entry point accepts.
//
// Work item = one message, walked by a lane group (NREP adjacent lanes); replicated registers: sha_info_digest (5), the
// working variables A..E, the 16-word schedule window.  Sync points: `sha_info_digest[i] += ...` at the end of every
// sha_transform are memory stores (:113-117) -> five store-data votes per transform (oracle chsha_item).
//
// Arithmetic: 80 rounds fully unrolled, rolling 16-register schedule window, v_bitop3 for f1 (0xCA), f3 (0xE8) and the
// three-way XORs (0x96), v_alignbit for the rotates: ~9 VALU per round per lane.  VALU-bound like sha256.
#include "xmr.hpp"

namespace coast {

enum { SITE_CHSHA_W = 40, SITE_CHSHA_WV = 41, SITE_CHSHA_DIGEST = 42, SITE_CHSHA_I = 43, SITE_CHSHA_COUNT = 44 };

__device__ __forceinline__ uint32_t chsha_rotl(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

// the armed faults of one tile, copied into registers once (a tile rarely owns more than a few)
struct TileFaults {
    static constexpr uint32_t kLocal = 4;
    DevFault lf[kLocal];
    uint32_t nLocal, n;
    const DevFault *list;
    __device__ __forceinline__ void load(const FaultTab &ft, uint2 fr)
    {
        n = fr.y;
        list = ft.list + fr.x;
        nLocal = fr.y <= kLocal ? fr.y : 0u;
#pragma unroll
        for (uint32_t q = 0; q < kLocal; ++q)
            if (q < nLocal)
                lf[q] = list[q];
    }
    template <class F> __device__ __forceinline__ void each(F &&f) const
    {
        if (nLocal) {
#pragma unroll
            for (uint32_t q = 0; q < kLocal; ++q)
                if (q < nLocal)
                    f(lf[q]);
        } else {
            for (uint32_t q = 0; q < n; ++q)
                f(list[q]);
        }
    }
};

// sha_transform (sha.c:82-118).  HOOKS: word t of the schedule is produced right before round t and a W hook of step t is
// applied there, a working-variable hook before round t; rounds never feed the schedule, so every value that depends on a
// flipped word is computed after the flip exactly as in the reference's "all of W first" order.
template <bool HOOKS>
__device__ __forceinline__ void chsha_transform(uint32_t dg[5], uint32_t W[16], uint32_t cidx, const TileFaults &tf, int slot,
                                                int rep, bool laneLive)
{
    uint32_t v[5] = {dg[0], dg[1], dg[2], dg[3], dg[4]};
#pragma unroll
    for (int t = 0; t < 80; ++t) {
        if (t >= 16)
            W[t & 15] = __builtin_amdgcn_bitop3_b32(W[(t - 3) & 15], W[(t - 8) & 15], W[(t - 14) & 15], 0x96) ^ W[t & 15];
        if constexpr (HOOKS) {
            tf.each([&](const DevFault &df) {
                if ((int)df.local != slot || (int)df.replica != rep || !laneLive || df.step != cidx * 80u + (uint32_t)t)
                    return;
                const uint32_t mask = 1u << (df.bit & 31u);
                if (df.site == SITE_CHSHA_W) {
                    W[t & 15] ^= mask;
                } else if (df.site == SITE_CHSHA_WV) {
#pragma unroll
                    for (int w = 0; w < 5; ++w)
                        if (w == (int)(df.index % 5u))
                            v[w] ^= mask;
                }
            });
        }
        uint32_t f, k;
        if (t < 20) {
            f = __builtin_amdgcn_bitop3_b32(v[1], v[2], v[3], 0xCA); // (B & C) | (~B & D)
            k = 0x5a827999u;
        } else if (t < 40) {
            f = __builtin_amdgcn_bitop3_b32(v[1], v[2], v[3], 0x96); // B ^ C ^ D
            k = 0x6ed9eba1u;
        } else if (t < 60) {
            f = __builtin_amdgcn_bitop3_b32(v[1], v[2], v[3], 0xE8); // majority
            k = 0x8f1bbcdcu;
        } else {
            f = __builtin_amdgcn_bitop3_b32(v[1], v[2], v[3], 0x96);
            k = 0xca62c1d6u;
        }
        const uint32_t temp = chsha_rotl(v[0], 5) + f + v[4] + W[t & 15] + k;
        v[4] = v[3];
        v[3] = v[2];
        v[2] = chsha_rotl(v[1], 30);
        v[1] = v[0];
        v[0] = temp;
    }
#pragma unroll
    for (int w = 0; w < 5; ++w)
        dg[w] += v[w];
}

// four waves per workgroup, one tile of IPW messages per wave, every tile in the one launch.  HOOKS = the launch has armed upsets: a
// transform that one of them points into takes the hooked (wave-uniform) branch, the digest hooks sit between the transforms; the
// clean launch is the HOOKS = false instance.  (tileList: a launch over a list of tiles -- rounds 1-3 ran the armed tiles that way,
// beside the main launch on a side stream.)
template <int NREP, bool HOOKS>
__global__ __launch_bounds__(256) void chsha_kernel(const uint8_t *__restrict__ msgs, size_t stride, uint32_t len,
                                                    uint64_t nmsgs, uint64_t ntiles, uint32_t *__restrict__ digests,
                                                    Counters ctr, FaultTab ft, const uint32_t *__restrict__ tileList,
                                                    uint32_t nListed, uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const uint64_t widx = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t tile = tileList ? (widx < nListed ? tileList[widx] : ntiles) : widx;
    bool tileOk = tile < ntiles;
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range && tileOk)
        fr = ft.range[tile];
    if (!HOOKS && fr.y != 0u) { // the side launch owns this tile
        tileOk = false;
        fr = make_uint2(0u, 0u);
    }
    TileFaults tf;
    tf.load(ft, HOOKS ? fr : make_uint2(0u, 0u));
    const int slot = lm.q;
    const uint64_t item = tile * IPW + (uint64_t)slot;
    const bool live = tileOk && lm.live && item < nmsgs;
    const bool cnt = live && lm.r == 0;
    const uint8_t *msg = msgs + (live ? item : 0) * stride;
    const bool aligned = ((stride & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(msgs) & 3u) == 0u);

    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    uint32_t dg[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u}; // sha_init, :120-128
    Tally tl;
    const uint32_t nblk = len >> 6;
    for (uint32_t c = 0; c <= nblk; ++c) {
        uint32_t W[16];
        if (c < nblk) {
            const uint8_t *p = msg + (size_t)c * 64;
            if (aligned) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    W[t] = reinterpret_cast<const uint32_t *>(p)[t];
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    W[t] = (uint32_t)p[4 * t] | ((uint32_t)p[4 * t + 1] << 8) | ((uint32_t)p[4 * t + 2] << 16) |
                           ((uint32_t)p[4 * t + 3] << 24);
            }
        } else { // sha_final with count == 0
#pragma unroll
            for (int t = 0; t < 16; ++t)
                W[t] = 0u;
            W[0] = 0x80u;
            W[14] = len >> 29;
            W[15] = len << 3;
        }
        if constexpr (HOOKS) {
            tf.each([&](const DevFault &df) {
                if (df.site != SITE_CHSHA_DIGEST || df.step != c || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                    return;
#pragma unroll
                for (int w = 0; w < 5; ++w)
                    if (w == (int)(df.index % 5u))
                        dg[w] ^= 1u << (df.bit & 31u);
            });
        }
        // only a transform that an armed fault points into pays for the hooks (wave-uniform: the table is the tile's)
        bool hooked = false;
        if constexpr (HOOKS) {
            tf.each([&](const DevFault &df) {
                if ((df.site == SITE_CHSHA_W || df.site == SITE_CHSHA_WV) && df.step / 80u == c)
                    hooked = true;
            });
        }
        if (HOOKS && hooked)
            chsha_transform<true>(dg, W, c, tf, slot, lm.r, lm.live);
        else
            chsha_transform<false>(dg, W, c, tf, slot, lm.r, lm.live);
#pragma unroll
        for (int w = 0; w < 5; ++w) // sha_info_digest[w] += ... are stores: store-data sync
            dg[w] = xmr_store_sync<NREP>(dg[w], lm, cnt, tl);
    }
    uint32_t detItems = 0;
    if (cnt) {
#pragma unroll
        for (int w = 0; w < 5; ++w)
            digests[item * 5 + w] = dg[w];
        if (tl.det) {
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// CHStone sha with its loops as written (sha.c:84-172), for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: sha_transform's loop counter `i`
// (an int) and sha_update's `count` are replica-private lane registers -- one lane per (message, replica), one sequential walk.
// memcpy / memset are library names (functions.config:12): calls outside the sphere of replication.  Sync points added to the frozen
// schedule, the reference's rule set for -TMR -noMemReplication on the source as written:
//   every evaluated branch condition: `count >= SHA_BLOCKSIZE` (:141), the carry test of sha_update (:136), `count > 56` of sha_final
//     (:162), the six loop conditions of sha_transform (17 + 65 + 4 x 21 = 166 per transform)                synchronization.cpp:146-155
//   every GEP with a variable index: sha_info_data[i] (load) and W[i] (store) of the copy loop; W[i-3], W[i-8], W[i-14], W[i-16]
//     (loads) and W[i] (store) of the expansion; W[i] (load) of the 80 rounds = 432 per transform (loads: off with -noLoadSync,
//     stores: off with -noStoreAddrSync)
// W[] stays replica-private as in the frozen schedule (the lane's own 320 bytes of LDS, since it is indexed at run time); a voted
// (or, unvoted, replica 0's) offset selects the element every copy accesses.  Fault sites: SITE_CHSHA_I / _COUNT of a replica, `step`
// = how many LOOP conditions the call has evaluated; SITE_CHSHA_DIGEST keeps its meaning.  A wild index reads 0 / stores nothing;
// blocks past the message read as 0; a walk that a corrupted counter keeps alive is cut after 4 x the clean count + 1024 loop
// conditions.  Oracle: chsha_item_indexed.  The sync-point-parity form of the kernel, not the throughput form.
template <int NREP>
__global__ __launch_bounds__(64) void chsha_indexed_kernel(const uint8_t *__restrict__ msgs, size_t stride, uint32_t len,
                                                           uint64_t nmsgs, uint32_t *__restrict__ digests, Counters ctr,
                                                           FaultTab ft, uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sW[80 * 64]; // W[t] of lane l at t * 64 + l
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    // COAST_F_LOCAL_STORE_SYNC: the data of every store of the -O0 IR -- ++i, W[i] = .., A..E = .., temp / E / D / C / B / A of FUNC,
    // count and its bit counts, sha_info_data[14 / 15]
    const bool lss = xmr_local_sync_on(ctr.flags);
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nmsgs;
    const bool cnt = live && lm.r == 0;
    const uint8_t *msg = msgs + (live ? item : 0) * stride;
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    uint32_t *W = sW + lm.lane;
    const uint32_t nblk = len / 64u;
    const uint32_t cap = 4u * ((nblk + 1u) * 167u + 1u) + 1024u;
    Tally tl;
    uint32_t dg[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u};
    uint32_t i = 0u, count = len, tick = 0u;
    if (lm.live) { // (the idle lane of a TMR wave has no message of its own: its replica group would wrap to lanes 0, 1)
        auto loopc = [&](int32_t limit, bool ge, bool isCount) __attribute__((always_inline)) { // one evaluated loop condition
            for (uint32_t q = 0; q < fr.y; ++q) { // the counters' upsets land right before the condition reads them
                const DevFault df = ft.list[fr.x + q];
                if (df.step != tick || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const uint32_t m = 1u << (df.bit & 31u);
                if (df.site == SITE_CHSHA_I)
                    i ^= m;
                else if (df.site == SITE_CHSHA_COUNT)
                    count ^= m;
            }
            if (tick >= cap)
                return false;
            ++tick;
            const int32_t v = (int32_t)(isCount ? count : i); // (read after the hook: it may just have flipped it)
            return xmr_steer<NREP>((ge ? v >= limit : v < limit) ? 1u : 0u, lm, bs, cnt, tl) != 0u;
        };
        auto off = [&](int32_t delta, bool store) __attribute__((always_inline)) {
            return xmr_steer<NREP>(i + (uint32_t)delta, lm, store ? ss : ls, cnt, tl);
        };
        auto digestHook = [&](uint32_t cidx) __attribute__((always_inline)) {
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.site != SITE_CHSHA_DIGEST || df.step != cidx || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const uint32_t m = 1u << (df.bit & 31u), w = df.index % 5u;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if ((uint32_t)k == w)
                        dg[k] ^= m;
            }
        };
        // sha_transform on block `blk` of the message (blocks past the message read as 0), or on sha_final's padding block
        auto transform = [&](uint32_t blk, bool pad) __attribute__((always_inline)) {
            auto inWord = [&](uint32_t o) __attribute__((always_inline)) -> uint32_t {
                if (o >= 16u)
                    return 0u;
                if (pad)
                    return o == 0u ? 0x80u : o == 14u ? (len >> 29) : o == 15u ? (len << 3) : 0u;
                if (blk < nblk) {
                    const uint8_t *p = msg + (size_t)blk * 64u + 4u * o;
                    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                }
                return 0u;
            };
            for (int t = 0; t < 80; ++t)
                W[t * 64] = 0u;
            auto lsy = [&](uint32_t v) __attribute__((always_inline)) { return xmr_local_sync<NREP>(v, lm, lss, cnt, tl); };
            for (i = 0u; loopc(16, false, false); i = lsy(i + 1u)) {           // W[i] = sha_info_data[i]          :88-90
                const uint32_t ol = off(0, false), os = off(0, true);
                const uint32_t x = lsy(inWord(ol));
                if (os < 80u)
                    W[os * 64u] = x;
            }
            for (i = 16u; loopc(80, false, false); i = lsy(i + 1u)) {          // the expansion                     :91-93
                const uint32_t o3 = off(-3, false), o8 = off(-8, false), o14 = off(-14, false), o16 = off(-16, false);
                const uint32_t os = off(0, true);
                const uint32_t x = lsy((o3 < 80u ? W[o3 * 64u] : 0u) ^ (o8 < 80u ? W[o8 * 64u] : 0u) ^ (o14 < 80u ? W[o14 * 64u] : 0u) ^
                                       (o16 < 80u ? W[o16 * 64u] : 0u));
                if (os < 80u)
                    W[os * 64u] = x;
            }
            uint32_t A = lsy(dg[0]), B = lsy(dg[1]), C = lsy(dg[2]), D = lsy(dg[3]), E = lsy(dg[4]);
#pragma unroll 1
            for (int seg = 0; seg < 4; ++seg)                                   // FUNC(1..4, i)                     :100-111
                for (i = 20u * (uint32_t)seg; loopc(20 * (seg + 1), false, false); i = lsy(i + 1u)) {
                    const uint32_t o = off(0, false);
                    const uint32_t f = seg == 0 ? ((B & C) | (~B & D)) : seg == 2 ? ((B & C) | (B & D) | (C & D)) : (B ^ C ^ D);
                    const uint32_t k = seg == 0 ? 0x5a827999u : seg == 1 ? 0x6ed9eba1u : seg == 2 ? 0x8f1bbcdcu : 0xca62c1d6u;
                    const uint32_t temp = lsy(chsha_rotl(A, 5) + f + E + (o < 80u ? W[o * 64u] : 0u) + k);
                    E = lsy(D);
                    D = lsy(C);
                    C = lsy(chsha_rotl(B, 30));
                    B = lsy(A);
                    A = lsy(temp);
                }
            dg[0] += A, dg[1] += B, dg[2] += C, dg[3] += D, dg[4] += E;
#pragma unroll
            for (int w = 0; w < 5; ++w)                                         // sha_info_digest[w] += ...: stored  :113-117
                dg[w] = xmr_store_sync<NREP>(dg[w], lm, cnt, tl);
        };
        count = xmr_local_sync<NREP>(count, lm, lss, cnt, tl);                  // the parameter `count` into its alloca
        (void)xmr_steer<NREP>(0u, lm, bs, cnt, tl);                             // the carry test of sha_update      :136
        (void)xmr_local_sync<NREP>(count << 3, lm, lss, cnt, tl);               // sha_info_count_lo += (LONG) count << 3   :139
        (void)xmr_local_sync<NREP>(count >> 29, lm, lss, cnt, tl);              // sha_info_count_hi += (LONG) count >> 29  :140
        uint32_t cidx = 0u;
        for (;; count = xmr_local_sync<NREP>(count - 64u, lm, lss, cnt, tl)) {  // while (count >= SHA_BLOCKSIZE)     :141
            if (!loopc(64, true, true))
                break;
            digestHook(cidx);
            transform(cidx, false);
            ++cidx;
        }
        (void)xmr_local_sync<NREP>(len << 3, lm, lss, cnt, tl);                 // sha_final: lo_bit_count = sha_info_count_lo  :157
        (void)xmr_local_sync<NREP>(len >> 29, lm, lss, cnt, tl);                //            hi_bit_count = sha_info_count_hi  :158
        (void)xmr_local_sync<NREP>(0u, lm, lss, cnt, tl);                       //            count = (lo_bit_count >> 3) & 0x3f :159
        (void)xmr_steer<NREP>(0u, lm, ss, cnt, tl);                             // sha_final: sha_info_data[count++] = 0x80 -- a store
                                                                                //   through a variable index (count = 0 here)  :161
        (void)xmr_local_sync<NREP>(1u, lm, lss, cnt, tl);                       //            count++
        (void)xmr_steer<NREP>(0u, lm, bs, cnt, tl);                             // sha_final: if (count > 56)        :162
        (void)xmr_local_sync<NREP>(len >> 29, lm, lss, cnt, tl);                // sha_info_data[14] = hi_bit_count  :168
        (void)xmr_local_sync<NREP>(len << 3, lm, lss, cnt, tl);                 // sha_info_data[15] = lo_bit_count  :169
        digestHook(cidx);
        transform(0u, true); // the padding block -- a derailed walk pads all the same (sha_final does)
    }
    uint32_t detItems = 0;
    if (cnt) {
#pragma unroll
        for (int w = 0; w < 5; ++w)
            digests[item * 5 + w] = dg[w];
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
