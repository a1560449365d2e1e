// crc16_kernel.hip -- protected crc16 for gfx950.
//
// Replaces: the TMR/DWC-transformed crc16() of tests/crc16/crc16.c:21-31 (CRC-16/CCITT, init 0xFFFF, byte serial):
//     x = crc >> 8 ^ *data_p++;  x ^= x >> 4;  crc = (crc << 8) ^ (x << 12) ^ (x << 5) ^ x;     (u8 / u16 truncations)
// The reference's `length` is an unsigned char, so a stream is a batch of independent blocks (<= 255 bytes in the
// reference; any block_len here).  Logical work item = one block; lane NREP*q + r holds replica r of the wave's q-th
// block (crc and x in VGPRs).  Sync points: the returned crc (ReturnInst sync, synchronization.cpp:741-949) and, with
// sync_every = V, crc after every V-th byte (the `while (length--)` loop-condition sync, crc16.c:25).
#include "xmr.hpp"

namespace coast {

enum { SITE_CRC_CRC = 24, SITE_CRC_X = 25 };

__device__ __forceinline__ uint32_t crc16_byte(uint32_t crc, uint32_t byte)
{
    uint32_t x = ((crc >> 8) ^ byte) & 0xffu;
    x ^= x >> 4;
    return ((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu;
}

template <int NREP>
__global__ __launch_bounds__(256) void crc16_xmr_kernel(const uint8_t *__restrict__ data, uint32_t blockLen,
                                                        uint64_t nblocksData, uint16_t *__restrict__ crcs,
                                                        uint32_t syncEvery, uint32_t nwg, Counters ctr, FaultTab ft,
                                                        int haveFaults, uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    constexpr int IPB = 4 * IPW;
    const LaneMap<NREP> lm;
    const int wave = threadIdx.x >> 6;
    const uint32_t lb = blockIdx.x;
    const int slot = wave * IPW + lm.q;
    const uint64_t item = (uint64_t)lb * IPB + (uint64_t)slot;
    const bool live = lm.live && item < nblocksData;
    const uint8_t *p = data + (live ? item : 0) * (uint64_t)blockLen;
    const bool aligned = ((blockLen & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(data) & 3u) == 0u);

    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    uint2 fr = make_uint2(0u, 0u);
    if (haveFaults)
        fr = ft.range[lb];
    const bool general = (fr.y != 0u) || (syncEvery != 0u);
    const bool cnt = live && lm.r == 0;
    Tally tl;
    uint32_t crc = 0xFFFFu;

    if (!general) {
        uint32_t t = 0;
        if (aligned) {
            for (; t + 4u <= blockLen; t += 4u) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(p + t);
                crc = crc16_byte(crc, w & 0xffu);
                crc = crc16_byte(crc, (w >> 8) & 0xffu);
                crc = crc16_byte(crc, (w >> 16) & 0xffu);
                crc = crc16_byte(crc, w >> 24);
            }
        }
        for (; t < blockLen; ++t)
            crc = crc16_byte(crc, p[t]);
    } else {
        for (uint32_t t = 0; t < blockLen; ++t) {
            uint32_t xm = 0u;
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.step != t || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                    continue;
                if (df.site == SITE_CRC_CRC)
                    crc = flip_bit(crc, df.bit, 0xffffu);
                else if (df.site == SITE_CRC_X)
                    xm ^= (1u << (df.bit & 31u)) & 0xffu;
            }
            uint32_t x = ((crc >> 8) ^ (uint32_t)p[t]) & 0xffu;
            x ^= x >> 4;
            x ^= xm;
            crc = ((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu;
            if (syncEvery && ((t + 1u) % syncEvery) == 0u && (t + 1u) < blockLen)
                crc = xmr_sync<NREP>(crc, lm, cnt, tl);
        }
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step == blockLen && df.site == SITE_CRC_CRC && (int)df.local == slot && (int)df.replica == lm.r &&
                lm.live)
                crc = flip_bit(crc, df.bit, 0xffffu);
        }
    }
    crc = xmr_sync<NREP>(crc, lm, cnt, tl); // return-value sync
    uint32_t detItems = 0;
    if (cnt) {
        crcs[item] = (uint16_t)crc;
        if (NREP == 2 && tl.det) {
            detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
    (void)nwg;
}

} // namespace coast
