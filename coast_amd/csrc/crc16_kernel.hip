// crc16_kernel.hip -- protected crc16 for gfx950.
//
// Replaces: the TMR/DWC-transformed crc16() of tests/crc16/crc16.c:21-31 (CRC-16/CCITT, init 0xFFFF, byte serial):
//     x = crc >> 8 ^ *data_p++;  x ^= x >> 4;  crc = (crc << 8) ^ (x << 12) ^ (x << 5) ^ x;     (u8 / u16 truncations)
// The reference's `length` is an unsigned char, so a stream is a batch of independent blocks (<= 255 bytes in the
// reference; any block_len here).  Logical work item = one block; lane NREP*q + r holds replica r of the wave's q-th
// block (crc and x in VGPRs); a wave owns one TILE of IPW = 64/NREP consecutive blocks.  Sync points: the returned
// crc (ReturnInst sync, synchronization.cpp:741-949) and, with sync_every = V, crc after every V-th byte (the
// `while (length--)` loop-condition sync, crc16.c:25).
//
// Two kernels:
//   crc16_stream_kernel   the HBM-streaming path (any block_len, mandatory sync point only).  A tile that owns an armed
//                         upset walks its blocks byte by byte with the injector hooks INSIDE this kernel (wave-uniform
//                         branch), so the flipped crc meets this kernel's own return-value sync, counter gate and store.
//                         The byte-serial update costs ~9 VALU ops per byte per replica -- 3 replicas would cap the
//                         chip near 2 TB/s -- so two update steps are folded into ONE lookup: for W = (b0<<8)|b1,
//                         crc' = T16[crc ^ W] (both byte steps depend on crc and the data only through crc ^ W; derived
//                         in crc16_table_kernel).  T16 is 64 Ki x u16 = 128 KiB: it fills the CU's LDS, so the kernel is
//                         persistent (one 1024-thread workgroup per CU, waves grid-stride over tiles).  The table is a
//                         single shared read-only copy (memory, outside the sphere of replication); crc stays
//                         replica-private, the three replica lanes look up the same address (LDS broadcast).
//                         Each lane streams its own block 64 bytes (4 x dwordx4) at a time, next batch in flight under
//                         the current one; the replicas of a block issue the same addresses (one fetch).  Blocks that
//                         are not 16-byte aligned (the reference's own maximum is 255 bytes) start anywhere: the lane
//                         loads the DWORD-aligned 16-byte chunks that cover its block (byte-misaligned wide loads take
//                         the memory pipeline's split path: 1.4 TB/s) and the byte order swap that every dword needs
//                         anyway becomes a funnel v_perm_b32 over two neighbouring dwords with a per-lane selector --
//                         alignment costs no instruction.  The tail (block_len % 4 bytes) comes out of the same
//                         registers: one more pair lookup and / or one byte-serial step.
//   crc16_general_kernel  byte-serial, exactly as written in crc16.c, with the injector hooks and the optional
//                         per-V-bytes votes; one wave per tile; every tile when the launch asks for sync_every != 0 or
//                         COAST_F_BRANCH_SYNC.
//
// Three more stream walks at the end of the file are round 6's measured-slower other formulation (no lookups: the recurrence on four
// blocks per register; crc16_hybrid_kernel, crc16_packed_kernel, crc16_mixed_kernel<PW>).  The library does not instantiate them;
// tools/crc_hyb_probe.hip does, and profiles/r06_crc16_hybrid.txt has their times.
#include "xmr.hpp"

namespace coast {

enum { SITE_CRC_CRC = 24, SITE_CRC_X = 25, SITE_CRC_LEN = 26 };

__device__ __forceinline__ uint32_t crc16_byte(uint32_t crc, uint32_t byte)
{
    uint32_t x = ((crc >> 8) ^ byte) & 0xffu;
    x ^= x >> 4;
    return ((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu;
}

// T16[(u << 8) | v] = state after two byte steps from a state/data pair with crc ^ ((b0<<8)|b1) == (u<<8)|v.
// Proof sketch: step 1 reads only u = (crc>>8)^b0 and produces s1 = ((crc&0xff)<<8) ^ T8[u]; step 2 reads
// (s1>>8)^b1 = v ^ (T8[u]>>8), and its (s1<<8) term keeps only T8[u]&0xff -- crc and the bytes enter only via u, v.
// Running the reference recurrence from crc = idx with both data bytes zero realises exactly that.
__global__ void crc16_table_kernel(uint16_t *__restrict__ t16)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; // 65536 threads
    t16[idx] = (uint16_t)crc16_byte(crc16_byte(idx, 0u), 0u);
}

// 1 (shipped) = the stream kernel's return-value vote in DPP form when only replica 0 stores; 0 = ds_bpermute (A/B)
#ifndef COAST_CRC_DPP_VOTE
#define COAST_CRC_DPP_VOTE 1
#endif
constexpr int kCrcStreamThreads = 1024;
constexpr int kCrcTableBytes = 65536 * 2;

// Slicing by four bytes (round 3).  The two-byte step M16 = T16 is linear over GF(2) (shifts and xors, T16[0] = 0), so one dword
// b0 b1 b2 b3 takes the state s to  M32(s ^ b0b1) ^ M16(b2b3)  and each term splits into its bytes:
//     s' = U3[s_hi ^ b0] ^ U2[s_lo ^ b1] ^ U1[b2] ^ U0[b3],   U3[e] = T16[T16[e << 8]], U2[e] = T16[T16[e]], U1[e] = T16[e << 8], U0[e] = T16[e].
// Only the first two lookups depend on the running crc: the dependent chain is ONE table level per dword instead of two, and the
// four 256-entry tables can be laid out conflict-free (entry e of lane l in bank l: dword e * 64 + l), where the 64 Ki-entry pair
// table loses half its LDS cycles to bank conflicts.  Two packed arrays of 64 KB: PA[e] = U3'[e] | U1'[e] << 16, PB[e] = U2'[e] |
// U0'[e] << 16 -- the halves are chosen so that the chain's two values meet in the low halves and the data's two in the high ones.
// The primes: every value is stored byte-swapped, because the walk keeps the state byte-swapped (sigma = s_lo << 8 | s_hi): then
// sigma ^ (little-endian data dword) carries s_hi ^ b0 and s_lo ^ b1 in its bytes 0 and 1 with no byte-order swap at all.
constexpr int kCrcSliceWords = 512; // PA[256], PB[256] behind the pair table in the context's table buffer
__global__ void crc16_slice_table_kernel(const uint16_t *__restrict__ t16, uint32_t *__restrict__ pab)
{
    const uint32_t e = threadIdx.x; // 256 threads
    auto sw = [](uint32_t v) { return ((v & 0xffu) << 8) | ((v >> 8) & 0xffu); };
    const uint32_t u1 = t16[e << 8], u0 = t16[e];
    const uint32_t u3 = t16[u1], u2 = t16[u0];
    pab[e] = sw(u3) | (sw(u1) << 16);
    pab[256 + e] = sw(u2) | (sw(u0) << 16);
}

// big-endian dword (b0<<24)|(b1<<16)|(b2<<8)|b3 of the four bytes that start `sh` bytes into the little-endian dword pair
// {hi, lo}: v_perm_b32 picks byte k of the result from byte sel[k] of the pair (0..3 = lo, 4..7 = hi).  sh = 0 is bswap(lo).
__device__ __forceinline__ uint32_t crc_perm_sel(uint32_t sh) { return 0x00010203u + 0x01010101u * sh; }
__device__ __forceinline__ uint32_t crc_be32(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// crc16() byte by byte as written (crc16.c:25-29) with the injector hooks: a flip of the lane's crc register before byte `step`
// (step == length: after the loop), of its temporary x right after `x ^= x >> 4`.  What the stream kernel runs for a tile that
// owns an armed upset (and for the last tiles of an unaligned stream).  The lane's upsets are gathered from the table once (it sits
// in HBM) and the row is fetched as the aligned dwords it covers, 8 at a time -- the walk costs about what the lookup walk of a
// clean tile costs, so a persistent workgroup is not held up by it.  More than four upsets on one lane: crc16_bytes_hooked_any.
__device__ __noinline__ uint32_t crc16_bytes_hooked_any(const uint8_t *p, uint32_t blockLen, const DevFault *list, uint2 fr, int slot,
                                                        int rep, bool laneLive)
{
    uint32_t crc = 0xFFFFu;
    for (uint32_t t = 0; t < blockLen; ++t) {
        uint32_t xm = 0u;
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = list[fr.x + q];
            if (df.step != t || (int)df.local != slot || (int)df.replica != rep || !laneLive)
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
    }
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault df = list[fr.x + q];
        if (df.step == blockLen && df.site == SITE_CRC_CRC && (int)df.local == slot && (int)df.replica == rep && laneLive)
            crc = flip_bit(crc, df.bit, 0xffffu);
    }
    return crc;
}

__device__ __forceinline__ uint32_t crc16_bytes_hooked(const uint8_t *p, uint32_t blockLen, const FaultTab &ft, uint2 fr, int slot,
                                                       int rep, bool laneLive)
{
    // this lane's upsets: step, and the XOR mask for crc (bits 0..15) or x (bits 16..23)
    uint32_t fs0 = 0xffffffffu, fs1 = 0xffffffffu, fs2 = 0xffffffffu, fs3 = 0xffffffffu, fm0 = 0u, fm1 = 0u, fm2 = 0u, fm3 = 0u;
    uint32_t nmine = 0u;
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault df = ft.list[fr.x + q];
        if ((int)df.local != slot || (int)df.replica != rep || !laneLive || (df.site != SITE_CRC_CRC && df.site != SITE_CRC_X))
            continue;
        if (df.site == SITE_CRC_X && df.step >= blockLen)
            continue; // x exists inside the loop only
        const uint32_t bit = 1u << (df.bit & 31u);
        const uint32_t m = df.site == SITE_CRC_CRC ? (bit & 0xffffu) : ((bit & 0xffu) << 16);
        fs0 = nmine == 0u ? df.step : fs0, fm0 = nmine == 0u ? m : fm0;
        fs1 = nmine == 1u ? df.step : fs1, fm1 = nmine == 1u ? m : fm1;
        fs2 = nmine == 2u ? df.step : fs2, fm2 = nmine == 2u ? m : fm2;
        fs3 = nmine == 3u ? df.step : fs3, fm3 = nmine == 3u ? m : fm3;
        nmine += 1u;
    }
    // (the ballot keeps the lanes of the wave together: all of them take the rare generic path, or none)
    if (__builtin_amdgcn_ballot_w64(nmine > 4u) != 0ull)
        return crc16_bytes_hooked_any(p, blockLen, ft.list, fr, slot, rep, laneLive);
    uint32_t crc = 0xFFFFu;
    auto byteStep = [&](uint32_t t, uint32_t byte) __attribute__((always_inline)) {
        const uint32_t m = (fs0 == t ? fm0 : 0u) ^ (fs1 == t ? fm1 : 0u) ^ (fs2 == t ? fm2 : 0u) ^ (fs3 == t ? fm3 : 0u);
        crc ^= m & 0xffffu;
        uint32_t x = ((crc >> 8) ^ byte) & 0xffu;
        x ^= x >> 4;
        x ^= m >> 16;
        crc = ((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu;
    };
    uint32_t t = 0u;
    const uint32_t head = min(blockLen, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u)) & 3u));
    for (; t < head; ++t)
        byteStep(t, (uint32_t)p[t]);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p + head);
    const uint32_t ndw = (blockLen - head) >> 2;
    uint32_t d = 0u;
#pragma unroll 1
    for (; d + 8u <= ndw; d += 8u) { // eight loads in flight, then their 32 byte steps
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = w[d + i];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                byteStep(t++, (v[i] >> (8 * b)) & 0xffu);
    }
    for (; d < ndw; ++d) {
        const uint32_t v = w[d];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            byteStep(t++, (v >> (8 * b)) & 0xffu);
    }
    for (; t < blockLen; ++t)
        byteStep(t, (uint32_t)p[t]);
    const uint32_t mEnd = (fs0 == blockLen ? fm0 : 0u) ^ (fs1 == blockLen ? fm1 : 0u) ^ (fs2 == blockLen ? fm2 : 0u) ^
                          (fs3 == blockLen ? fm3 : 0u);
    return crc ^ (mEnd & 0xffffu);
}

// THREADS: workgroup size (one persistent workgroup per CU: the table fills its LDS).  Lookup chains in flight per CU =
// THREADS / 64 x NT; registers per lane = 512 x 256 / THREADS.  1024 x NT 2 is the shipped shape for aligned rows; 768 x NT 4
// (48 chains, 170 registers) is the experiment of round 3 (COAST_CRC_SHAPE, profiles/r03_crc16_shapes.txt).
// T16[idx] without the table: the two byte steps of the reference recurrence from state idx with zero data (the map is what the
// table tabulates).  ~17 VALU instructions; the hybrid walk (HYB) runs every HYB-th pair step of a chain through it instead of
// through LDS, taking that share of lookups off the conflicted LDS array while the VALU has issue slots to spare.
__device__ __forceinline__ uint32_t crc16_pair_valu(uint32_t idx)
{
    uint32_t x = idx >> 8;
    x ^= x >> 4;
    const uint32_t c1 = ((idx << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu;
    uint32_t y = c1 >> 8;
    y ^= y >> 4;
    return ((c1 << 8) ^ (y << 12) ^ (y << 5) ^ y) & 0xffffu;
}

template <int NREP, int NT, bool ALIGNED, int THREADS = kCrcStreamThreads, int HYB = 0>
__global__ __launch_bounds__(THREADS) void crc16_stream_kernel(
    const uint8_t *__restrict__ data, uint32_t blockLen, uint64_t nblocksData, uint16_t *__restrict__ crcs,
    const uint16_t *__restrict__ t16g, uint64_t ntiles, uint64_t ntilesWalk, Counters ctr, FaultTab ft,
    uint8_t *__restrict__ detected, size_t copyIn = 0, size_t copyOut = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    uint16_t *T = reinterpret_cast<uint16_t *>(smemRaw);
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemRaw + kCrcTableBytes);
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;

    static_assert(HYB >= 0 || ALIGNED, "slicing walk: aligned rows");
    constexpr bool SLICE4 = HYB < 0; // the walk: slicing by four over bank-replicated byte tables instead of the pair table
    if constexpr (SLICE4) { // t16g points at PA[256], PB[256]: every entry is written 64 times, once per bank (= lane)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(t16g);
        uint32_t *dst = reinterpret_cast<uint32_t *>(smemRaw);
        for (int e = tid; e < kCrcTableBytes / 4; e += THREADS)
            dst[e] = src[e >> 6];
    } else { // 128 KiB table: 1024 threads x 8 x 16 B, L2-resident after the first workgroup
        const uint4 *src = reinterpret_cast<const uint4 *>(t16g);
        uint4 *dst = reinterpret_cast<uint4 *>(T);
        for (int e = tid; e < kCrcTableBytes / 16; e += THREADS)
            dst[e] = src[e];
    }
    if constexpr (SLICE4) {
        if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smemRaw != 0u)
            __builtin_trap(); // the slicing walk addresses the tables from LDS offset 0
    }
    // slicing walk: byte 0 = the lane's bank (4 * lane), byte 1 = 1 (the 64 KB step from PA to PB)
    const uint32_t laneSel = ((uint32_t)(tid & 63) << 2) | 0x100u;
    auto slice4 = [&](uint32_t sigma, uint32_t d) __attribute__((always_inline)) {
        const uint32_t y = sigma ^ d; // bytes 0, 1: s_hi ^ b0, s_lo ^ b1 (the upper half of sigma is never clean, nor read)
        const uint32_t a0 = __builtin_amdgcn_perm(y, laneSel, 0x0c0c0400u), a1 = __builtin_amdgcn_perm(y, laneSel, 0x0c010500u);
        const uint32_t a2 = __builtin_amdgcn_perm(d, laneSel, 0x0c0c0600u), a3 = __builtin_amdgcn_perm(d, laneSel, 0x0c010700u);
        // the permuted bytes ARE the LDS address (the dynamic segment starts at 0, checked above): no base add per lookup
        typedef const __attribute__((address_space(3))) uint32_t *lds_u32p;
        const uint32_t r0 = *(lds_u32p)(uintptr_t)(a0), r1 = *(lds_u32p)(uintptr_t)(a1), r2 = *(lds_u32p)(uintptr_t)(a2), r3 = *(lds_u32p)(uintptr_t)(a3);
        return r0 ^ r1 ^ ((r2 ^ r3) >> 16);
    };
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    const uint64_t wavesTotal = (uint64_t)gridDim.x * (THREADS / kWave);
    const uint64_t wave0 = (uint64_t)blockIdx.x * (THREADS / kWave) + (tid >> 6);
    Tally tl;
    uint32_t detItems = 0;
    // A block is walked as nd full dwords, then tb tail bytes.  Rows that do not start on a 16-byte boundary are read as the
    // DWORD-aligned 16-byte chunks that cover them and realigned by the byte swap every dword needs anyway: row dword i =
    // bytes s .. s+3 of stream dwords i, i + 1 (s = start & 3 per lane) = one funnel v_perm_b32 with a per-lane selector.
    // Measured on the 8 GiB stream of 255-byte blocks (profiles/r02_crc16_unaligned.txt): byte-aligned 16-byte loads + byte
    // tail 6.09 ms; this version 3.10 ms; 16-byte-aligned chunks + two more v_cndmask per dword for the lane's dword offset
    // 3.30 ms (a dword-aligned 16-byte load costs the memory pipeline about as much as the two selects cost the VALU).
    // Chunks are loaded only while they hold a needed dword, so a row over-reads < 20 bytes -- into the following blocks;
    // the host keeps the last tile(s) of an unaligned stream away from this kernel.
    const uint32_t nd = blockLen >> 2, tb = blockLen & 3u;
    const uint32_t nbFull = nd >> 4, R = nd & 15u;             // 64-byte batches of full dwords, dwords left after them
    const uint32_t nChunks = (nd + (tb ? 1u : 0u) + 1u + 3u) >> 2; // row dword i sits in stream dwords i, i + 1

    for (uint64_t tileBase = wave0 * NT; tileBase < ntiles; tileBase += wavesTotal * NT) {
        // NT independent tiles per wave: NT dependent lookup chains in flight per lane
        const uint8_t *p[NT];
        bool liveT[NT], cntT[NT], slowT[NT];
        uint64_t itemT[NT];
        uint32_t crc[NT], sel[NT];
        uint32_t cur[NT][16], nxt[NT][16]; // one 64-byte batch of the block's dword stream, and the batch after it
#define CRC_LOAD_CHUNK(dst, j, bt, v)                                                                        \
    do {                                                                                                     \
        const uint4 q__ = *reinterpret_cast<const uint4 *>(p[j] + (size_t)(bt) * 64 + 16 * (v));             \
        dst[j][4 * (v)] = q__.x, dst[j][4 * (v) + 1] = q__.y, dst[j][4 * (v) + 2] = q__.z, dst[j][4 * (v) + 3] = q__.w; \
    } while (0)
#define CRC_LOAD_BATCH_GUARDED(dst, bt) /* only the chunks that hold a needed dword (wave-uniform tests; clamping the */ \
    /* chunk index instead made the compiler narrow the vector loads to dwords: 2.5x slower)                           */ \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) if (4u * (bt) + (uint32_t)v < nChunks)                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) CRC_LOAD_CHUNK(dst, j, bt, v)
#define CRC_LOAD_BATCH(dst, bt)                                                                              \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int v = 0; v < 4; ++v) CRC_LOAD_CHUNK(dst, j, bt, v)
        // big-endian dword b0 b1 b2 b3 at row dword i of the current batch (the funnel's upper dword: stream dword i + 1)
#define CRC_NEXT(j, i, out) out = crc_be32((i) < 15 ? cur[j][((i) + 1) & 15] : nxt[j][0], cur[j][i], sel[j])
#define CRC_WINDOW_INIT(j) (void)0
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint64_t tile = tileBase + j;
            const bool skip = tile >= ntiles;
            // a tile that owns an armed upset, or one of the stream's last tiles (rows that are not 16-byte aligned are read as
            // the dword-aligned chunks that cover them: < 20 bytes past the row), is walked byte by byte below -- wave-uniform
            slowT[j] = !skip && (tile >= ntilesWalk ||
                                 (ft.range && __builtin_amdgcn_readfirstlane(ft.range[tile].y) != 0u));
            itemT[j] = tile * IPW + (uint64_t)lm.q;
            liveT[j] = !skip && lm.live && itemT[j] < nblocksData;
            cntT[j] = liveT[j] && lm.r == 0;
            // the lookup walk of a lane that has no block of its own (or whose tile is walked byte by byte) reads row 0
            // (copyIn != 0: COAST_F_MEMORY_COPIES -- NREP copies of the stream back to back, replica r walks copy r)
            const uint8_t *row = data + (size_t)lm.r * copyIn + ((liveT[j] && !slowT[j]) ? itemT[j] : 0) * (uint64_t)blockLen;
            const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(row) & 3u);
            p[j] = row - sh;
            sel[j] = crc_perm_sel(sh);
            crc[j] = 0xFFFFu;
        }
        bool anyWalk = false; // wave-uniform: some tile of this round takes the lookup walk
#pragma unroll
        for (int j = 0; j < NT; ++j)
            anyWalk = anyWalk || (tileBase + j < ntiles && !slowT[j]);
        if (anyWalk) {
        if constexpr (ALIGNED) { // rows are whole 16-byte chunks: exact loads, the byte swap is a plain v_perm
            const uint32_t nbatch = blockLen >> 6, rem = blockLen & 63u;
            uint4 c4[NT][4], n4[NT][4];
            if (nbatch) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        c4[j][v] = *reinterpret_cast<const uint4 *>(p[j] + 16 * v);
            }
            for (uint32_t b = 0; b < nbatch; ++b) {
                const bool more = (b + 1) < nbatch;
                if (more) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            n4[j][v] = *reinterpret_cast<const uint4 *>(p[j] + (size_t)(b + 1) * 64 + 16 * v);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (SLICE4) {
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const uint32_t d = (c == 0) ? c4[j][v].x : (c == 1) ? c4[j][v].y : (c == 2) ? c4[j][v].z : c4[j][v].w;
                                crc[j] = slice4(crc[j], d);
                            }
                            continue;
                        }
                        uint32_t e[NT];
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const uint32_t d = (c == 0) ? c4[j][v].x : (c == 1) ? c4[j][v].y : (c == 2) ? c4[j][v].z : c4[j][v].w;
                            e[j] = __builtin_bswap32(d); // (b0<<24)|(b1<<16)|(b2<<8)|b3
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int step = 8 * v + 2 * c + (HYB > 0 ? j * (HYB / 2) : 0); // the chains take their VALU turns apart
                            if (HYB > 0 && step % HYB == 0)
                                crc[j] = crc16_pair_valu(crc[j] ^ (e[j] >> 16));
                            else
                                crc[j] = T[crc[j] ^ (e[j] >> 16)];
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const int step = 8 * v + 2 * c + 1 + (HYB > 0 ? j * (HYB / 2) : 0);
                            if (HYB > 0 && step % HYB == 0)
                                crc[j] = crc16_pair_valu(crc[j] ^ (e[j] & 0xffffu));
                            else
                                crc[j] = T[crc[j] ^ (e[j] & 0xffffu)];
                        }
                    }
                }
                if (more) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            c4[j][v] = n4[j][v];
                }
            }
            for (uint32_t t = 0; t < rem; t += 4u) { // a block of 16, 32 or 48 bytes past the last batch
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (SLICE4) {
                        crc[j] = slice4(crc[j], *reinterpret_cast<const uint32_t *>(p[j] + (size_t)nbatch * 64 + t));
                        continue;
                    }
                    const uint32_t e = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(p[j] + (size_t)nbatch * 64 + t));
                    crc[j] = T[crc[j] ^ (e >> 16)];
                    crc[j] = T[crc[j] ^ (e & 0xffffu)];
                }
            }
        } else {
        if (nbFull) {
            CRC_LOAD_BATCH(cur, 0u);
        } else {
            CRC_LOAD_BATCH_GUARDED(cur, 0u);
        }
        for (uint32_t b = 0; b < nbFull; ++b) {
            if (b + 1u < nbFull) { // the batch after this one: full ...
                CRC_LOAD_BATCH(nxt, b + 1u);
            } else { // ... or partial, or just the dword a funnel reaches into
                CRC_LOAD_BATCH_GUARDED(nxt, b + 1u);
            }
            if (b == 0u) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    CRC_WINDOW_INIT(j);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint32_t e[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    CRC_NEXT(j, i, e[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    crc[j] = T[crc[j] ^ (e[j] >> 16)];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    crc[j] = T[crc[j] ^ (e[j] & 0xffffu)];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    cur[j][i] = nxt[j][i];
        }
        if (R | tb) { // what is left of the block: R < 16 full dwords, then tb < 4 bytes -- all of it already in `cur`
            if (4u * (nbFull + 1u) < nChunks) { // R == 15 with a tail: its funnel reaches one dword further
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    nxt[j][0] = *reinterpret_cast<const uint32_t *>(p[j] + (size_t)(nbFull + 1u) * 64);
            }
            if (nbFull == 0u) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    CRC_WINDOW_INIT(j);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((uint32_t)i < R) {
                    uint32_t e[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        CRC_NEXT(j, i, e[j]);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        crc[j] = T[crc[j] ^ (e[j] >> 16)];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        crc[j] = T[crc[j] ^ (e[j] & 0xffffu)];
                } else if ((uint32_t)i == R && tb) { // b0 [b1 [b2]]: a pair lookup for two bytes, a byte-serial step for the odd one
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        uint32_t e;
                        CRC_NEXT(j, i, e);
                        if (tb >= 2u)
                            crc[j] = T[crc[j] ^ (e >> 16)];
                        if (tb & 1u)
                            crc[j] = crc16_byte(crc[j], tb == 1u ? e >> 24 : (e >> 8) & 0xffu);
                    }
                }
            }
        }
        }
        if constexpr (SLICE4) { // sigma (byte-swapped, upper half dirty) -> crc
#pragma unroll
            for (int j = 0; j < NT; ++j)
                crc[j] = ((crc[j] & 0xffu) << 8) | ((crc[j] >> 8) & 0xffu);
        }
        } // anyWalk
#undef CRC_NEXT
#undef CRC_WINDOW_INIT
#undef CRC_LOAD_BATCH
#undef CRC_LOAD_BATCH_GUARDED
#undef CRC_LOAD_CHUNK
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (slowT[j]) { // the flips land in this lane's own crc / x registers; the sync point below is the one every tile meets
                uint2 fr = make_uint2(0u, 0u);
                if (ft.range) {
                    const uint2 rg = ft.range[tileBase + j];
                    fr.x = __builtin_amdgcn_readfirstlane(rg.x);
                    fr.y = __builtin_amdgcn_readfirstlane(rg.y);
                }
                crc[j] = crc16_bytes_hooked(data + (size_t)lm.r * copyIn + (liveT[j] ? itemT[j] : 0) * (uint64_t)blockLen,
                                            liveT[j] ? blockLen : 0u, ft, fr, lm.q, lm.r, lm.live);
            }
            Tally te = tl;
            te.det = 0;
            // return-value sync.  One memory copy: only replica 0 stores, so it votes on its two neighbours' values by DPP -- a ds_bpermute
            // is three ds_read_b32 worth of the LDS crossbar this kernel is bound by (COAST_CRC_DPP_VOTE; profiles/r06_aes_dpp_votes.txt)
            const uint32_t voted = (COAST_CRC_DPP_VOTE && copyOut == 0) ? xmr_final_vote_dpp<NREP>(crc[j], cntT[j], te)
                                                                         : xmr_sync<NREP>(crc[j], lm, cntT[j], te);
            tl.miss = te.miss;
            tl.syncs = te.syncs;
            if (cntT[j] || (liveT[j] && copyOut != 0)) // memory copies: every replica stores the voted crc into its own result copy
                crcs[(size_t)lm.r * copyOut + itemT[j]] = (uint16_t)(NREP == 3 ? voted : crc[j]);
            if (cntT[j]) {
                if (te.det) { // unequal copies at the sync point of this block (DWC: detected, TMR: corrected)
                    if (NREP == 2)
                        detItems += 1;
                    if (detected)
                        detected[itemT[j]] = 1;
                }
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ hybrid walk (round 6)
// crc16_hybrid_kernel: the lookup walk and a lookup-free walk side by side in one wave.  The pair-table walk above is bound by the
// latency of its dependent, bank-conflicted ds_read_u16 chain (the LDS array ~68 % busy, the VALU ~37 %, the wave in s_waitcnt
// 70 % of its cycles: profiles/r06_crc16_analysis.txt); round 3's hybrids put VALU steps INTO that chain and lengthened it.  Here
// a wave owns NTL tiles that walk the table as before and four more tiles whose blocks are walked by the reference recurrence
// itself (crc16.c:26-28), four blocks per instruction: lane NREP*q + r holds replica r of block q of EACH of the four tiles, their
// crc high bytes packed in one register (H), their low bytes in another (L), and a byte step is 9 instructions for the four
//     x = H ^ D;  y = x ^ ((x >> 4) & 0x0f0f0f0f);  H' = L ^ ((y << 4) & 0xf0f0f0f0) ^ ((y >> 3) & 0x1f1f1f1f);  L' = y ^ ((y << 5) & 0xe0e0e0e0)
// (the 16-bit recurrence split into its bytes: the high byte of crc << 8 is the old low byte, x << 12 reaches the high byte as
// x << 4, x << 5 as x >> 3; the masks stop a byte's bits at its neighbour's border).  D = the four blocks' data bytes, one
// v_perm_b32 transposition (8 per 16 bytes) behind the funnel every row dword takes anyway.  The packed walk has no lookup and no
// wait: it issues in the shadow of the table chains' LDS latency.  Same tiles, same lanes, same return-value sync, same armed /
// tail tiles (walked byte by byte with the hooks) as crc16_stream_kernel; bit-identical results (tests: every alignment / tail).
__device__ __forceinline__ void crc16_swar_step(uint32_t &H, uint32_t &L, uint32_t D)
{
    // a ^ (b & mask) is ONE v_bitop3_b32 (truth table 0x78); written out because the compiler splits two of the four into v_and + v_xor
    const uint32_t x = H ^ D;
    const uint32_t y = __builtin_amdgcn_bitop3_b32(x, x >> 4, 0x0f0f0f0fu, 0x78);
    const uint32_t u = __builtin_amdgcn_bitop3_b32(L, y << 4, 0xf0f0f0f0u, 0x78);
    H = __builtin_amdgcn_bitop3_b32(u, y >> 3, 0x1f1f1f1fu, 0x78);
    L = __builtin_amdgcn_bitop3_b32(y, y << 5, 0xe0e0e0e0u, 0x78);
}

constexpr int kCrcSwarTiles = 4;

template <int NREP, int NTL, bool ALIGNED, int BD = 8>
__global__ __launch_bounds__(kCrcStreamThreads) void crc16_hybrid_kernel(
    const uint8_t *__restrict__ data, uint32_t blockLen, uint64_t nblocksData, uint16_t *__restrict__ crcs,
    const uint16_t *__restrict__ t16g, uint64_t ntiles, uint64_t ntilesWalk, Counters ctr, FaultTab ft,
    uint8_t *__restrict__ detected, size_t copyIn = 0, size_t copyOut = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    uint16_t *T = reinterpret_cast<uint16_t *>(smemRaw);
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemRaw + kCrcTableBytes);
    constexpr int THREADS = kCrcStreamThreads;
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    constexpr int NT = NTL + kCrcSwarTiles, CB = BD / 4; // tiles per wave and round; 16-byte chunks per batch
    static_assert(BD == 4 || BD == 8 || BD == 16, "batch of 1, 2 or 4 chunks");
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    if constexpr (NTL > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(t16g);
        uint4 *dst = reinterpret_cast<uint4 *>(T);
        for (int e = tid; e < kCrcTableBytes / 16; e += THREADS)
            dst[e] = src[e];
    }
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    const uint64_t wavesTotal = (uint64_t)gridDim.x * (THREADS / kWave);
    const uint64_t wave0 = (uint64_t)blockIdx.x * (THREADS / kWave) + (tid >> 6);
    Tally tl;
    uint32_t detItems = 0;
    // the row geometry of crc16_stream_kernel: nd full dwords, tb tail bytes; unaligned rows as the dword-aligned chunks that cover them
    const uint32_t nd = blockLen >> 2, tb = blockLen & 3u;
    const uint32_t nbFull = nd / BD, R = nd % BD;
    const uint32_t nChunks = ALIGNED ? (blockLen >> 4) : ((nd + (tb ? 1u : 0u) + 1u + 3u) >> 2);

    for (uint64_t tileBase = wave0 * NT; tileBase < ntiles; tileBase += wavesTotal * NT) {
        const uint8_t *p[NT];
        bool liveT[NT], cntT[NT], slowT[NT];
        uint64_t itemT[NT];
        uint32_t crc[NT], sel[NT];
        uint32_t cur[NT][BD], nxt[NT][BD];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint64_t tile = tileBase + j;
            const bool skip = tile >= ntiles;
            slowT[j] = !skip && (tile >= ntilesWalk || (ft.range && __builtin_amdgcn_readfirstlane(ft.range[tile].y) != 0u));
            itemT[j] = tile * IPW + (uint64_t)lm.q;
            liveT[j] = !skip && lm.live && itemT[j] < nblocksData;
            cntT[j] = liveT[j] && lm.r == 0;
            const uint8_t *row = data + (size_t)lm.r * copyIn + ((liveT[j] && !slowT[j]) ? itemT[j] : 0) * (uint64_t)blockLen;
            const uint32_t sh = ALIGNED ? 0u : (uint32_t)(reinterpret_cast<uintptr_t>(row) & 3u);
            p[j] = row - sh;
            // table tiles take their row dwords big-endian (b0 b1 | b2 b3 are the two pair indices), the packed tiles as they lie
            sel[j] = j < NTL ? crc_perm_sel(sh) : 0x03020100u + 0x01010101u * sh;
            crc[j] = 0xFFFFu;
        }
        bool anyWalk = false;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            anyWalk = anyWalk || (tileBase + j < ntiles && !slowT[j]);
        if (anyWalk) {
            uint32_t H = 0xffffffffu, L = 0xffffffffu;
#define CRCH_LOAD_CHUNK(dst, j, bt, v)                                                                       \
    do {                                                                                                     \
        const uint4 q__ = *reinterpret_cast<const uint4 *>(p[j] + (size_t)(bt) * (4 * BD) + 16 * (v));       \
        dst[j][4 * (v)] = q__.x, dst[j][4 * (v) + 1] = q__.y, dst[j][4 * (v) + 2] = q__.z, dst[j][4 * (v) + 3] = q__.w; \
    } while (0)
#define CRCH_LOAD_GUARDED(dst, bt)                                                                           \
    _Pragma("unroll") for (int v = 0; v < CB; ++v) if ((uint32_t)CB * (bt) + (uint32_t)v < nChunks)         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) CRCH_LOAD_CHUNK(dst, j, bt, v)
#define CRCH_LOAD(dst, bt)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) _Pragma("unroll") for (int v = 0; v < CB; ++v) CRCH_LOAD_CHUNK(dst, j, bt, v)
            // row dword i of the current batch: the table tiles' big-endian, the packed tiles' little-endian
            auto rowDword = [&](int j, int i) __attribute__((always_inline)) {
                if constexpr (ALIGNED)
                    return j < NTL ? __builtin_bswap32(cur[j][i]) : cur[j][i];
                else
                    return __builtin_amdgcn_perm(i < BD - 1 ? cur[j][(i + 1) & (BD - 1)] : nxt[j][0], cur[j][i], sel[j]);
            };
            // one row dword of every tile: two pair lookups per table chain, four packed byte steps (`nbytes` of them at a row's tail)
            auto step = [&](int i, uint32_t nbytes) __attribute__((always_inline)) {
                uint32_t e[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    e[j] = rowDword(j, i);
                const uint32_t a01 = __builtin_amdgcn_perm(e[NTL + 1], e[NTL], 0x05010400u); // {e0.0, e1.0, e0.1, e1.1}
                const uint32_t a23 = __builtin_amdgcn_perm(e[NTL + 3], e[NTL + 2], 0x05010400u);
                const uint32_t b01 = __builtin_amdgcn_perm(e[NTL + 1], e[NTL], 0x07030602u); // {e0.2, e1.2, e0.3, e1.3}
                const uint32_t b23 = __builtin_amdgcn_perm(e[NTL + 3], e[NTL + 2], 0x07030602u);
                if (nbytes == 4u) {
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        crc[j] = T[crc[j] ^ (e[j] >> 16)];
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x05040100u));
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x07060302u));
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        crc[j] = T[crc[j] ^ (e[j] & 0xffffu)];
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x05040100u));
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x07060302u));
                } else { // b0 [b1 [b2]]: a pair lookup for two bytes and a byte-serial step for the odd one; nbytes packed steps
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        if (nbytes >= 2u)
                            crc[j] = T[crc[j] ^ (e[j] >> 16)];
                        if (nbytes & 1u)
                            crc[j] = crc16_byte(crc[j], nbytes == 1u ? e[j] >> 24 : (e[j] >> 8) & 0xffu);
                    }
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x05040100u));
                    if (nbytes >= 2u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x07060302u));
                    if (nbytes == 3u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x05040100u));
                }
            };
            if (nbFull) {
                CRCH_LOAD(cur, 0u);
            } else {
                CRCH_LOAD_GUARDED(cur, 0u);
            }
            for (uint32_t b = 0; b < nbFull; ++b) {
                if (b + 1u < nbFull) {
                    CRCH_LOAD(nxt, b + 1u);
                } else {
                    CRCH_LOAD_GUARDED(nxt, b + 1u);
                }
#pragma unroll
                for (int i = 0; i < BD; ++i)
                    step(i, 4u);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < BD; ++i)
                        cur[j][i] = nxt[j][i];
            }
            if (R | tb) { // R < BD full dwords, then tb < 4 bytes: all of it in `cur`, except the dword the last funnel reaches into
                if (!ALIGNED && (uint32_t)CB * (nbFull + 1u) < nChunks) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        nxt[j][0] = *reinterpret_cast<const uint32_t *>(p[j] + (size_t)(nbFull + 1u) * (4 * BD));
                }
#pragma unroll
                for (int i = 0; i < BD; ++i) {
                    if ((uint32_t)i < R)
                        step(i, 4u);
                    else if ((uint32_t)i == R && tb)
                        step(i, tb);
                }
            }
#undef CRCH_LOAD
#undef CRCH_LOAD_GUARDED
#undef CRCH_LOAD_CHUNK
#pragma unroll
            for (int k = 0; k < kCrcSwarTiles; ++k)
                crc[NTL + k] = (((H >> (8 * k)) & 0xffu) << 8) | ((L >> (8 * k)) & 0xffu);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (slowT[j]) { // armed / tail tiles: byte by byte with the hooks, as in crc16_stream_kernel
                uint2 fr = make_uint2(0u, 0u);
                if (ft.range) {
                    const uint2 rg = ft.range[tileBase + j];
                    fr.x = __builtin_amdgcn_readfirstlane(rg.x);
                    fr.y = __builtin_amdgcn_readfirstlane(rg.y);
                }
                crc[j] = crc16_bytes_hooked(data + (size_t)lm.r * copyIn + (liveT[j] ? itemT[j] : 0) * (uint64_t)blockLen,
                                            liveT[j] ? blockLen : 0u, ft, fr, lm.q, lm.r, lm.live);
            }
            Tally te = tl;
            te.det = 0;
            const uint32_t voted = xmr_sync<NREP>(crc[j], lm, cntT[j], te); // return-value sync
            tl.miss = te.miss;
            tl.syncs = te.syncs;
            if (cntT[j] || (liveT[j] && copyOut != 0))
                crcs[(size_t)lm.r * copyOut + itemT[j]] = (uint16_t)(NREP == 3 ? voted : crc[j]);
            if (cntT[j]) {
                if (te.det) {
                    if (NREP == 2)
                        detItems += 1;
                    if (detected)
                        detected[itemT[j]] = 1;
                }
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ packed walk from LDS (round 6)
// crc16_packed_kernel: the packed byte-step walk alone, its rows staged in LDS.  The probe of crc16_hybrid_kernel
// (profiles/r06_crc16_hybrid.txt) showed what holds the packed walk back: not its instructions (0.75 per byte and lane against
// the lookup walk's 1.4 + half a lookup) but its fetches -- four tiles per wave are four times the rows in flight per CU, every
// 16-byte piece of a row is its own request at a 255-byte stride, and the lines fall out of L1 / L2 between two pieces.  Without
// the 128 KiB table the LDS is free: a wave copies the 4 x IPW consecutive rows of its four tiles into its own LDS buffer with
// LDS-DMA (global_load_lds_dwordx4: 1 KiB per instruction, every HBM line requested once, whole, in address order; no register
// holds stream data outside the replicas -- the staged copy is memory, like the table it replaces) and every replica lane reads
// its four rows from there (its own ds_read_b32 per dword: a cloned load).  Layout: row n of the wave's region at LDS byte
// 272 n + (address of the row & 15): the 17 aligned 16-byte chunks that cover a row of <= 256 bytes, chunk j of row n fetched by
// lane (17 n + j) % 64 of DMA instruction (17 n + j) / 64 -- the source address is per lane, the destination lane-linear.  272 B
// = 68 dwords: the lanes of one ds_read_b32 (same dword of rows q, q + 1, ...) fall into banks 4 q + (0..3): conflict-free
// within each half of the wave; the replicas of a block read the same address (broadcast).  One workgroup of 7 waves per CU
// (7 x 84 x 272 B = 156 KiB); a wave's DMA phase runs under the other waves' walks.  Row lengths 160..256 bytes, TMR; armed /
// tail tiles byte by byte from HBM with the hooks, as in crc16_stream_kernel.  The DMA's address arithmetic is not replicated
// (neither is the persistent loop's tile counter); the walk, the crc registers and the return-value sync are.
constexpr int kCrcPackPitch = 272;
#ifndef CRC_PACK_KNOCK
#define CRC_PACK_KNOCK 0 // tools/crc_hyb_probe: 1 = no staging copy, 2 = no walk (timing knock-outs; results are wrong)
#endif
template <int NREP> struct CrcPack {
    static constexpr int kRows = kCrcSwarTiles * LaneMap<NREP>::kItemsPerWave;  // rows staged per wave and round
    static constexpr int kBuf = kRows * kCrcPackPitch;                          // LDS bytes per wave
    static constexpr int kWaves = (160 * 1024 - 512) / kBuf;                    // 7 (TMR), 4 (DWC), 2 (unprotected)
    static constexpr int kThreads = kWaves * kWave;
    static constexpr int kPieces = (kRows * 17 + 63) / 64;                      // DMA instructions per round
    static constexpr size_t kLds = (size_t)kWaves * kBuf + kCrcPackPitch + 16;  // + the reach of the last batch past the last row, counters
};

template <int NREP, bool ALIGNED, int BD = 8>
__global__ __launch_bounds__(CrcPack<NREP>::kThreads) void crc16_packed_kernel(
    const uint8_t *__restrict__ data, uint32_t blockLen, uint64_t nblocksData, uint16_t *__restrict__ crcs, uint64_t ntiles,
    uint64_t ntilesWalk, Counters ctr, FaultTab ft, uint8_t *__restrict__ detected)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    using CP = CrcPack<NREP>;
    constexpr int THREADS = CP::kThreads, IPW = LaneMap<NREP>::kItemsPerWave, NT = kCrcSwarTiles;
    typedef __attribute__((address_space(3))) uint8_t *lds_u8p;
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32p;
    typedef const __attribute__((address_space(1))) uint32_t *glb_u32p;
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemRaw + (size_t)CP::kWaves * CP::kBuf + kCrcPackPitch);
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const lds_u8p wbuf = (lds_u8p)smemRaw + wv * (uint32_t)CP::kBuf;
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    const uint64_t wavesTotal = (uint64_t)gridDim.x * CP::kWaves;
    const uint64_t wave0 = (uint64_t)blockIdx.x * CP::kWaves + wv;
    Tally tl;
    uint32_t detItems = 0;
    const uint32_t nd = blockLen >> 2, tb = blockLen & 3u;
    const uint32_t nbFull = nd / BD, R = nd % BD;
    const uint32_t qRow = lm.live ? (uint32_t)lm.q : 0u;

    for (uint64_t tileBase = wave0 * NT; tileBase < ntiles; tileBase += wavesTotal * NT) {
        bool liveT[NT], cntT[NT], slowT[NT];
        uint64_t itemT[NT];
        uint32_t crc[NT];
        bool anyWalk = false; // wave-uniform: some tile of this round takes the packed walk
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const uint64_t tile = tileBase + j;
            const bool skip = tile >= ntiles;
            slowT[j] = !skip && (tile >= ntilesWalk || (ft.range && __builtin_amdgcn_readfirstlane(ft.range[tile].y) != 0u));
            itemT[j] = tile * IPW + (uint64_t)lm.q;
            liveT[j] = !skip && lm.live && itemT[j] < nblocksData;
            cntT[j] = liveT[j] && lm.r == 0;
            crc[j] = 0xFFFFu;
            anyWalk = anyWalk || (!skip && !slowT[j]);
        }
        if (anyWalk) {
            // the rows to stage: the round's tiles below ntilesWalk (the stream's last tiles are never read ahead of), the data's end
            const uint64_t firstItem = tileBase * IPW;
            const uint64_t endTile = tileBase + NT < ntilesWalk ? tileBase + NT : ntilesWalk;
            const uint64_t endItem = endTile * IPW < nblocksData ? endTile * IPW : nblocksData;
            const uint32_t rowsStage = (uint32_t)(endItem - firstItem); // anyWalk: tileBase < ntilesWalk
            const uint8_t *region = data + firstItem * (uint64_t)blockLen;
            const uint32_t regLow = (uint32_t)reinterpret_cast<uintptr_t>(region) & 15u;
            if (CRC_PACK_KNOCK != 1) {
                uint32_t n = (uint32_t)lm.lane / 17u, jc = (uint32_t)lm.lane - 17u * n; // piece 0: chunk jc of row n
#pragma unroll
                for (int m = 0; m < CP::kPieces; ++m) {
                    const uint32_t rowOff = n * blockLen + regLow;  // the row's first byte, from the region's 16-byte floor
                    const uint32_t src = (rowOff & ~15u) + 16u * jc; // this lane's chunk, same origin
                    if (n < rowsStage && src < rowOff + blockLen)    // the chunk holds a byte of the row
                        __builtin_amdgcn_global_load_lds((glb_u32p)(uintptr_t)(region - regLow + src),
                                                         (__attribute__((address_space(3))) uint32_t *)(wbuf + m * 1024), 16, 0, 0);
                    n += 3u, jc += 13u; // 64 = 3 * 17 + 13
                    if (jc >= 17u)
                        jc -= 17u, n += 1u;
                }
            }
            lds_u32p lp[NT];
            uint32_t sel[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const uint32_t nrow = (uint32_t)(IPW * k) + qRow;
                const uint32_t b = nrow * (uint32_t)kCrcPackPitch + ((nrow * blockLen + regLow) & 15u);
                lp[k] = (lds_u32p)(wbuf + (b & ~3u));
                sel[k] = 0x03020100u + 0x01010101u * (b & 3u); // the funnel: row dword i = bytes s .. s + 3 of LDS dwords i, i + 1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the staged rows have landed (this wave's own buffer: no barrier)
            uint32_t H = 0xffffffffu, L = 0xffffffffu;
            uint32_t cur[NT][BD], nxt[NT][BD];
#define CRCP_LOAD(dst, bt)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < NT; ++k) _Pragma("unroll") for (int i = 0; i < BD; ++i) dst[k][i] = lp[k][(bt) * BD + i]
            auto step = [&](int i, uint32_t nbytes) __attribute__((always_inline)) {
                uint32_t e[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if constexpr (ALIGNED)
                        e[k] = cur[k][i];
                    else
                        e[k] = __builtin_amdgcn_perm(i < BD - 1 ? cur[k][(i + 1) & (BD - 1)] : nxt[k][0], cur[k][i], sel[k]);
                }
                const uint32_t a01 = __builtin_amdgcn_perm(e[1], e[0], 0x05010400u), a23 = __builtin_amdgcn_perm(e[3], e[2], 0x05010400u);
                crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x05040100u));
                if (nbytes >= 2u)
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x07060302u));
                if (nbytes >= 3u) {
                    const uint32_t b01 = __builtin_amdgcn_perm(e[1], e[0], 0x07030602u), b23 = __builtin_amdgcn_perm(e[3], e[2], 0x07030602u);
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x05040100u));
                    if (nbytes == 4u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x07060302u));
                }
            };
            CRCP_LOAD(cur, 0u);
            for (uint32_t b = 0; b < (CRC_PACK_KNOCK == 2 ? 1u : nbFull); ++b) {
                CRCP_LOAD(nxt, b + 1u); // (LDS reads past a row's end stay inside the workgroup's allocation)
#pragma unroll
                for (int i = 0; i < BD; ++i)
                    step(i, 4u);
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int i = 0; i < BD; ++i)
                        cur[k][i] = nxt[k][i];
            }
            if (R | tb) {
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    nxt[k][0] = lp[k][(nbFull + 1u) * BD];
#pragma unroll
                for (int i = 0; i < BD; ++i) {
                    if ((uint32_t)i < R)
                        step(i, 4u);
                    else if ((uint32_t)i == R && tb)
                        step(i, tb);
                }
            }
#undef CRCP_LOAD
#pragma unroll
            for (int k = 0; k < NT; ++k)
                crc[k] = (((H >> (8 * k)) & 0xffu) << 8) | ((L >> (8 * k)) & 0xffu);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (slowT[j]) { // armed / tail tiles: byte by byte from HBM with the hooks, as in crc16_stream_kernel
                uint2 fr = make_uint2(0u, 0u);
                if (ft.range) {
                    const uint2 rg = ft.range[tileBase + j];
                    fr.x = __builtin_amdgcn_readfirstlane(rg.x);
                    fr.y = __builtin_amdgcn_readfirstlane(rg.y);
                }
                crc[j] = crc16_bytes_hooked(data + (liveT[j] ? itemT[j] : 0) * (uint64_t)blockLen, liveT[j] ? blockLen : 0u, ft, fr, lm.q,
                                            lm.r, lm.live);
            }
            Tally te = tl;
            te.det = 0;
            const uint32_t voted = xmr_sync<NREP>(crc[j], lm, cntT[j], te); // return-value sync
            tl.miss = te.miss;
            tl.syncs = te.syncs;
            if (cntT[j])
                crcs[itemT[j]] = (uint16_t)(NREP == 3 ? voted : crc[j]);
            if (cntT[j]) {
                if (te.det) {
                    if (NREP == 2)
                        detItems += 1;
                    if (detected)
                        detected[itemT[j]] = 1;
                }
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ mixed walk (round 6)
// crc16_mixed_kernel: both walks on one CU, each on the resource the other leaves idle.  The lookup walk is bound by the latency of
// its conflicted ds_read_u16 chain and uses a third of the VALU; the packed walk uses the VALU alone, and what it needs from LDS is
// a staging buffer.  Next to the 128 KiB table there are 32 KiB: one buffer of 84 rows x 96 bytes per packed wave -- a 64-byte
// SEGMENT of each row (the six aligned 16-byte chunks that cover it and the dword its last funnel reaches into), four packed waves
// (wave w of the workgroup runs on SIMD w % 4: one per SIMD), twelve lookup waves.  A packed wave reads a staged segment into
// registers (17 dwords x 4 rows), has the NEXT segment (or the next round's first) copied into the same buffer by LDS-DMA, and walks
// the 64 bytes from registers while the copy is in flight.  The stream is split statically: tiles [0, ntilesPacked) to the packed
// waves (four consecutive tiles per wave and round), the rest to the lookup waves; the host picks the share (COAST_CRC_PACK_SHARE).
constexpr int kCrcMixPackWaves = 4;
constexpr int kCrcMixSegDwords = 16;
constexpr int kCrcMixPitch = 96;                                         // LDS bytes per row segment
constexpr int kCrcMixRows = kCrcSwarTiles * 21;                          // TMR
constexpr int kCrcMixBuf = kCrcMixRows * kCrcMixPitch;                   // 8064 bytes per packed wave
constexpr int kCrcMixPieces = (kCrcMixRows * 6 + 63) / 64;               // DMA instructions per segment
// PW packed waves of the workgroup's 16: 4 = beside the table and 12 lookup waves; 16 = no table, no lookup waves
template <int PW> constexpr size_t crc_mix_lds() { return 16 + (PW < 16 ? (size_t)kCrcTableBytes : 0) + (size_t)PW * kCrcMixBuf; }
static_assert(crc_mix_lds<4>() <= 160 * 1024 && crc_mix_lds<16>() <= 160 * 1024, "table + staging buffers fit the CU's LDS");

template <int PW>
__global__ __launch_bounds__(kCrcStreamThreads) void crc16_mixed_kernel(
    const uint8_t *__restrict__ data, uint32_t blockLen, uint64_t nblocksData, uint16_t *__restrict__ crcs,
    const uint16_t *__restrict__ t16g, uint64_t ntiles, uint64_t ntilesWalk, uint64_t ntilesPacked, Counters ctr, FaultTab ft,
    uint8_t *__restrict__ detected)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemRaw[];
    constexpr int NREP = 3, THREADS = kCrcStreamThreads, IPW = LaneMap<NREP>::kItemsPerWave, NP = kCrcSwarTiles;
    constexpr int TW = THREADS / kWave - PW, SD = kCrcMixSegDwords;
    constexpr uint32_t kBufBase = 16u + (TW ? (uint32_t)kCrcTableBytes : 0u);
    typedef __attribute__((address_space(3))) uint8_t *lds_u8p;
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32p;
    typedef const __attribute__((address_space(1))) uint32_t *glb_u32p;
    uint16_t *T = reinterpret_cast<uint16_t *>(smemRaw + 16);
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemRaw);
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    const uint32_t wv = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    if constexpr (TW > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(t16g);
        uint4 *dst = reinterpret_cast<uint4 *>(T);
        for (int e = tid; e < kCrcTableBytes / 16; e += THREADS)
            dst[e] = src[e];
    }
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();

    Tally tl;
    uint32_t detItems = 0;
    const uint32_t nd = blockLen >> 2, tb = blockLen & 3u;

    // what every tile ends with: armed / tail tiles byte by byte with the hooks, the return-value sync, the store
    auto finishTile = [&](uint64_t tile, uint32_t crcWalk) __attribute__((always_inline)) {
        const bool skip = tile >= ntiles;
        const bool slow = !skip && (tile >= ntilesWalk || (ft.range && __builtin_amdgcn_readfirstlane(ft.range[tile].y) != 0u));
        const uint64_t item = tile * IPW + (uint64_t)lm.q;
        const bool live = !skip && lm.live && item < nblocksData, cnt = live && lm.r == 0;
        uint32_t crc = crcWalk;
        if (slow) {
            const uint2 rg = ft.range ? ft.range[tile] : make_uint2(0u, 0u);
            const uint2 fr = make_uint2(__builtin_amdgcn_readfirstlane(rg.x), __builtin_amdgcn_readfirstlane(rg.y));
            crc = crc16_bytes_hooked(data + (live ? item : 0) * (uint64_t)blockLen, live ? blockLen : 0u, ft, fr, lm.q, lm.r, lm.live);
        }
        Tally te = tl;
        te.det = 0;
        const uint32_t voted = xmr_sync<NREP>(crc, lm, cnt, te);
        tl.miss = te.miss;
        tl.syncs = te.syncs;
        if (cnt) {
            crcs[item] = (uint16_t)voted;
            if (te.det && detected)
                detected[item] = 1;
        }
    };

    if (wv < (uint32_t)PW) {
        // ---- packed waves: tiles [0, ntilesPacked), four per round
        const lds_u8p wbuf = (lds_u8p)smemRaw + kBufBase + wv * (uint32_t)kCrcMixBuf;
        const uint64_t stride = (uint64_t)gridDim.x * PW * NP;
        const uint32_t nseg = (nd + (tb ? 1u : 0u) + SD - 1u) / SD;
        const uint32_t qRow = lm.live ? (uint32_t)lm.q : 0u;
        // segment `seg` of the 84 rows that start at tile `tb0`: chunk jc (0..5) of row n -> LDS 96 n + 16 jc
        auto stage = [&](uint64_t tb0, uint32_t seg) __attribute__((always_inline)) {
            const uint8_t *region = data + tb0 * IPW * (uint64_t)blockLen;
            const uint32_t regLow = (uint32_t)reinterpret_cast<uintptr_t>(region) & 15u;
            uint32_t n = (uint32_t)lm.lane / 6u, jc = (uint32_t)lm.lane - 6u * n;
#pragma unroll
            for (int m = 0; m < kCrcMixPieces; ++m) {
                const uint32_t rowOff = n * blockLen + regLow;       // the row's first byte, from the region's 16-byte floor
                const uint32_t o = rowOff + 4u * SD * seg;           // the segment's first byte
                const uint32_t src = (o & ~15u) + 16u * jc;
                const uint32_t end = min(o + 4u * SD + 4u, rowOff + blockLen);
                if (n < (uint32_t)kCrcMixRows && src < end)
                    __builtin_amdgcn_global_load_lds((glb_u32p)(uintptr_t)(region - regLow + src),
                                                     (__attribute__((address_space(3))) uint32_t *)(wbuf + m * 1024), 16, 0, 0);
                n += 10u, jc += 4u; // 64 = 10 * 6 + 4
                if (jc >= 6u)
                    jc -= 6u, n += 1u;
            }
        };
        uint64_t tileBase = ((uint64_t)blockIdx.x * PW + wv) * NP;
        if (tileBase < ntilesPacked)
            stage(tileBase, 0u);
        for (; tileBase < ntilesPacked; tileBase += stride) {
            const uint8_t *region = data + tileBase * IPW * (uint64_t)blockLen;
            const uint32_t regLow = (uint32_t)reinterpret_cast<uintptr_t>(region) & 15u;
            lds_u32p lp[NP];
            uint32_t sel[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const uint32_t nrow = (uint32_t)(IPW * k) + qRow;
                const uint32_t b = nrow * (uint32_t)kCrcMixPitch + ((nrow * blockLen + regLow) & 15u);
                lp[k] = (lds_u32p)(wbuf + (b & ~3u));
                sel[k] = 0x03020100u + 0x01010101u * (b & 3u);
            }
            uint32_t H = 0xffffffffu, L = 0xffffffffu;
            for (uint32_t seg = 0; seg < nseg; ++seg) {
                uint32_t cur[NP][SD + 1];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the segment has landed
#pragma unroll
                for (int k = 0; k < NP; ++k)
#pragma unroll
                    for (int i = 0; i <= SD; ++i)
                        cur[k][i] = lp[k][i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... and is in registers: the buffer is free for the next one
                if (seg + 1u < nseg)
                    stage(tileBase, seg + 1u);
                else if (tileBase + stride < ntilesPacked)
                    stage(tileBase + stride, 0u);
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    const uint32_t g = seg * SD + (uint32_t)i; // row dword
                    const uint32_t nbytes = g < nd ? 4u : (g == nd ? tb : 0u);
                    if (nbytes == 0u)
                        break;
                    uint32_t e[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        e[k] = __builtin_amdgcn_perm(cur[k][i + 1], cur[k][i], sel[k]);
                    const uint32_t a01 = __builtin_amdgcn_perm(e[1], e[0], 0x05010400u), a23 = __builtin_amdgcn_perm(e[3], e[2], 0x05010400u);
                    const uint32_t b01 = __builtin_amdgcn_perm(e[1], e[0], 0x07030602u), b23 = __builtin_amdgcn_perm(e[3], e[2], 0x07030602u);
                    crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x05040100u));
                    if (nbytes >= 2u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(a23, a01, 0x07060302u));
                    if (nbytes >= 3u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x05040100u));
                    if (nbytes == 4u)
                        crc16_swar_step(H, L, __builtin_amdgcn_perm(b23, b01, 0x07060302u));
                }
            }
#pragma unroll
            for (int k = 0; k < NP; ++k)
                finishTile(tileBase + k, (((H >> (8 * k)) & 0xffu) << 8) | ((L >> (8 * k)) & 0xffu));
        }
    } else if constexpr (TW > 0) {
        // ---- lookup waves: tiles [ntilesPacked, ntiles), one per round (the walk of crc16_stream_kernel<3, 1, false>)
        const uint32_t nbFull = nd >> 4, R = nd & 15u;
        const uint32_t nChunks = (nd + (tb ? 1u : 0u) + 1u + 3u) >> 2;
        const uint64_t stride = (uint64_t)gridDim.x * TW;
#ifdef CRC_MIX_PRIO
        __builtin_amdgcn_s_setprio(3); // the lookup chain is latency-bound: its few VALU instructions go first
#endif
        for (uint64_t tile = ntilesPacked + (uint64_t)blockIdx.x * TW + (wv - PW); tile < ntiles; tile += stride) {
            const bool slow = tile >= ntilesWalk || (ft.range && __builtin_amdgcn_readfirstlane(ft.range[tile].y) != 0u);
            const uint64_t item = tile * IPW + (uint64_t)lm.q;
            const bool live = lm.live && item < nblocksData;
            uint32_t crc = 0xFFFFu;
            if (!slow) {
                const uint8_t *row = data + (live ? item : 0) * (uint64_t)blockLen;
                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(row) & 3u);
                const uint8_t *p = row - sh;
                const uint32_t sel = crc_perm_sel(sh);
                uint32_t cur[16], nxt[16];
#define CRCM_CHUNK(dst, bt, v)                                                                               \
    do {                                                                                                     \
        const uint4 q__ = *reinterpret_cast<const uint4 *>(p + (size_t)(bt) * 64 + 16 * (v));                \
        dst[4 * (v)] = q__.x, dst[4 * (v) + 1] = q__.y, dst[4 * (v) + 2] = q__.z, dst[4 * (v) + 3] = q__.w;  \
    } while (0)
#define CRCM_LOAD_GUARDED(dst, bt) _Pragma("unroll") for (int v = 0; v < 4; ++v) if (4u * (bt) + (uint32_t)v < nChunks) CRCM_CHUNK(dst, bt, v)
#define CRCM_LOAD(dst, bt) _Pragma("unroll") for (int v = 0; v < 4; ++v) CRCM_CHUNK(dst, bt, v)
                if (nbFull) {
                    CRCM_LOAD(cur, 0u);
                } else {
                    CRCM_LOAD_GUARDED(cur, 0u);
                }
                for (uint32_t b = 0; b < nbFull; ++b) {
                    if (b + 1u < nbFull) {
                        CRCM_LOAD(nxt, b + 1u);
                    } else {
                        CRCM_LOAD_GUARDED(nxt, b + 1u);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t e = crc_be32(i < 15 ? cur[(i + 1) & 15] : nxt[0], cur[i], sel);
                        crc = T[crc ^ (e >> 16)];
                        crc = T[crc ^ (e & 0xffffu)];
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        cur[i] = nxt[i];
                }
                if (R | tb) {
                    if (4u * (nbFull + 1u) < nChunks)
                        nxt[0] = *reinterpret_cast<const uint32_t *>(p + (size_t)(nbFull + 1u) * 64);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if ((uint32_t)i < R) {
                            const uint32_t e = crc_be32(i < 15 ? cur[(i + 1) & 15] : nxt[0], cur[i], sel);
                            crc = T[crc ^ (e >> 16)];
                            crc = T[crc ^ (e & 0xffffu)];
                        } else if ((uint32_t)i == R && tb) {
                            const uint32_t e = crc_be32(i < 15 ? cur[(i + 1) & 15] : nxt[0], cur[i], sel);
                            if (tb >= 2u)
                                crc = T[crc ^ (e >> 16)];
                            if (tb & 1u)
                                crc = crc16_byte(crc, tb == 1u ? e >> 24 : (e >> 8) & 0xffu);
                        }
                    }
                }
#undef CRCM_LOAD
#undef CRCM_LOAD_GUARDED
#undef CRCM_CHUNK
            }
            finishTile(tile, crc);
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// one wave (64-thread workgroup) per tile
template <int NREP>
__global__ __launch_bounds__(64) void crc16_general_kernel(const uint8_t *__restrict__ data, uint32_t blockLen,
                                                           uint64_t nblocksData, uint16_t *__restrict__ crcs,
                                                           uint32_t syncEvery, Counters ctr, FaultTab ft,
                                                           const uint32_t *__restrict__ tileList,
                                                           uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    const LaneMap<NREP> lm;
    const uint32_t tile = tileList ? tileList[blockIdx.x] : blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nblocksData;
    const uint8_t *p = data + (live ? item : 0) * (uint64_t)blockLen;
    const bool aligned = ((blockLen & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(data) & 3u) == 0u);

    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    const bool stepwise = (fr.y != 0u) || (syncEvery != 0u);
    const bool cnt = live && lm.r == 0;
    Tally tl;
    uint32_t crc = 0xFFFFu;

    if (ctr.flags & kFlagBranchSync) {
        // `while (length--)` as written (crc16.c:25): `length` is a replica-private unsigned char and its loop condition is
        // voted at every evaluation (terminator sync on an i1, synchronization.cpp:146-155).  `*data_p++` has a constant GEP
        // offset: no address vote in this function.  Mirrors oracle/coast_oracle.c:crc_item_branch (watchdog, bounded reads).
        uint32_t ln = blockLen & 0xffu;
        const uint32_t cap = 4u * blockLen + 256u;
        const bool lss = xmr_local_sync_on(ctr.flags); // COAST_F_LOCAL_STORE_SYNC: length, x (twice) and crc are stored into allocas at -O0
        ln = xmr_local_sync<NREP>(ln, lm, lss, cnt, tl); // the parameter into its alloca
        for (uint32_t it = 0;; ++it) {
            uint32_t xm = 0u, cm = 0u;
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.step != it || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                    continue;
                if (df.site == SITE_CRC_LEN)
                    ln ^= (1u << (df.bit & 31u)) & 0xffu;
                else if (df.site == SITE_CRC_CRC && it < blockLen)
                    cm ^= (1u << (df.bit & 31u)) & 0xffffu;
                else if (df.site == SITE_CRC_X && it < blockLen)
                    xm ^= (1u << (df.bit & 31u)) & 0xffu;
            }
            // while (length--): load, decrement, STORE (the data vote), then the branch on the loaded value
            const uint32_t old = ln;
            ln = xmr_local_sync<NREP>((ln - 1u) & 0xffu, lm, lss, cnt, tl); // length--: on both exits
            const uint32_t go = xmr_steer<NREP>(old != 0u ? 1u : 0u, lm, true, cnt, tl);
            if (!go || it >= cap)
                break;
            const uint32_t byte = (live && it < blockLen) ? (uint32_t)p[it] : 0u;
            crc ^= cm;
            uint32_t x = xmr_local_sync<NREP>(((crc >> 8) ^ byte) & 0xffu, lm, lss, cnt, tl);
            x = xmr_local_sync<NREP>(x ^ (x >> 4), lm, lss, cnt, tl);
            x ^= xm;
            crc = xmr_local_sync<NREP>(((crc << 8) ^ (x << 12) ^ (x << 5) ^ x) & 0xffffu, lm, lss, cnt, tl);
            if (syncEvery && ((it + 1u) % syncEvery) == 0u && (it + 1u) < blockLen)
                crc = xmr_sync<NREP>(crc, lm, cnt, tl);
        }
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step == blockLen && df.site == SITE_CRC_CRC && (int)df.local == slot && (int)df.replica == lm.r &&
                lm.live)
                crc = flip_bit(crc, df.bit, 0xffffu);
        }
    } else if (!stepwise) {
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
