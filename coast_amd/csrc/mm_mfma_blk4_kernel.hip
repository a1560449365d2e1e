// mm_mfma_blk4_kernel.hip -- round 6: the register-block TMR matrix_multiply kernel on a 128-ROW PANEL.
//
// mm_mfma_blk3_kernel (read its header and mm_mfma_blk2_kernel's first) gives a workgroup 64 rows of a matrix: four column-tile lanes x two
// row halves, so the four panel workgroups of a matrix each convert all of s into byte planes -- 0.9 ms of the 7 ms launch by the knock-out
// table (profiles/r04_mm_knockouts.txt).  Here a workgroup owns 128 rows: TWO column-tile lanes x FOUR row quarters (a wave still holds a
// 32-row x 16-column tile: 96 accumulator registers, two waves per SIMD), and s is converted twice per matrix instead of four times.
//
//   * A slab (64 k x 16 columns) of a lane is converted by the lane's four waves TOGETHER, a quarter each (16 k x 16 columns: one group of
//     four k of one column per thread), in the first half of the step in front of the one that multiplies it.  Every wave runs the same
//     code: the row quarter and the lane are wave-uniform run-time values (one instruction stream for the eight waves; mm_mfma_blk3_kernel
//     has two).  Per wave and step: 5 conversion stages instead of 10, four one-word raw loads (one register set of four words, requested a
//     step ahead) instead of sixteen words in flight.
//   * The f panel is 128 rows x 256 k x 4 byte planes = 128 KB: single-buffered (160 KB of LDS hold it and the lanes' two slab buffers).
//     The next matrix's panel replaces it REGION BY REGION -- region s = k-slab s of all 128 rows, 32 KB, four 16-byte pieces per thread --
//     each region behind the barrier that follows its last read (step s of the matrix's last column tile) and in front of the barrier that
//     precedes its first read for the next matrix: two pieces per thread and half step from the second half of the last tile's step 1 to
//     the first half of the next tile's step 1.  Those five steps have bodies of their own (BG = 1: the last tile of an item, BG = 2: the first
//     tile of the next); the other 27 of an item's 32 steps carry no f work at all.
//   * Everything else is mm_mfma_blk3_kernel's: six sets of ten MFMAs per step (row block x replica) with their own A and B fragment reads,
//     the tile end as a chain of 24 stages behind the MFMAs of the tile's last and the next tile's first step, one workgroup barrier per
//     step, injector deltas on the running limb-0 sums, select(a == b, a, c) in-lane (synchronization.cpp:934-938), counters by a
//     wave-uniform `real` predicate.
//   * CLONE (COAST_F_CLONE_STAGING; cloning.cpp:2187-2209, 2247-2255): every raw word of s and f is loaded a second time and compared in
//     front of the first instruction that consumes it; a mismatch loads it a third time and keeps select(a == b, a, c).  The clone of an s
//     word and of an f piece is requested right behind the original (the line the original just fetched).
//
// TMR only: DWC and the unprotected mode stay on mm_mfma_blk3_kernel<2 / 1>; so does COAST_SITE_MM_VGPR, which names that kernel's registers.
// PHYS == 2: COAST_SITE_MM_PREG as in mm_mfma_blk3_kernel -- a real exclusive-or on ANY physical register of a wave (v0..v255 through the VGPR
// index mode, s0..s101 through s_movrels / s_movreld) in front of any MFMA slot of any of an item's 32 steps (coast_fault.step: slot | step % 16
// << 6 | lane << 10 | wave << 16 | file << 19 | register << 20 | step / 16 << 29), the compiler knowing nothing of it: an instantiation of its
// own (60 hook points per step body), the vehicle of tools/campaign.py --reg-model uniform --kernel panel128.
#include <type_traits>
#include <utility>

#include "xmr.hpp"

// 0 = clone loads of s half a step ahead of the compare; 1 (shipped) = right behind the original: the clone hits the line the original just
// fetched (profiles/r06_mm_blk4_ab.txt: 6.87 -> 6.84 ms)
#ifndef COAST_MM4_DUP_ADJ
#define COAST_MM4_DUP_ADJ 1
#endif
// register sets of raw s words per wave (2: requested two steps ahead of their conversion; 1 (shipped): one step ahead, four registers fewer --
// the clone form 7.03 -> 6.87 ms, profiles/r06_mm_blk4_ab.txt)
#ifndef COAST_MM4_SETS
#define COAST_MM4_SETS 1
#endif
// A/B builds (profiles/r06_mm_blk4_ab.txt, fourth pass): bit 0 = under CLONE the original f piece is loaded with the default cache policy and only
// its clone non-temporal (no effect: 6.78 ms either way); bit 1 = the conversion's store-base compare rides in the staging compare's branch at
// the step's first slot (- 1.0 %: one ballot and branch per step instead of two -- NOT shipped: the base register is then single for the 24 slots
// between the compare and the stores, and three 5000-run campaigns of that build read 98.5 / 98.5 / 98.7 % against 98.9 % with the compare in
// front of the stores); bit 2 (shipped under CLONE) = the votes' agreement tally through a scalar population count instead of a per-lane add
// (CLONE: - 0.6 % and no register left in scratch; single staging: + 0.8 %)
#ifndef COAST_MM4_VAR
#define COAST_MM4_VAR 4
#endif
// development: conversion stage stride in the steps without f work (6: spread over the half step; 2: the first ten slots)
#ifndef COAST_MM4_CONV_STRIDE
#define COAST_MM4_CONV_STRIDE 6
#endif

namespace coast {

struct MmBlk4 {
    static constexpr int N = 256, KS = 64, NSLAB = N / KS;
    static constexpr int CT = 16;
    static constexpr int NLANE = 2, NQ = 4, NW = NLANE * NQ, NTHR = 64 * NW; // two column-tile lanes x four row quarters
    static constexpr int BM = 128, NPANEL = N / BM;
    static constexpr int NCT = N / CT, TPW = NCT / NLANE, SPP = TPW * NSLAB; // 8 tiles, 32 steps per panel
    // f panel: [row][byte plane][256 B] -- a row's four planes side by side, so that every fragment address of a step is one register + a 16-bit
    // instruction offset (plane-major planes of 32 KB put planes 2 and 3 beyond it: two v_add_u32 per set of ten MFMAs)
    static constexpr int PLANE_A = N, ROW_A = 4 * N, A_PANEL = BM * ROW_A;   // 128 KB, single-buffered
    static constexpr int PLANE_B = CT * KS, B_BUF = 4 * PLANE_B;
    static constexpr int LANE_LDS = 2 * B_BUF;
    static constexpr size_t LDS_BYTES = (size_t)A_PANEL + NLANE * LANE_LDS; // 144 KB
    static constexpr int A_PER_THR = (BM * (N / 4)) / NTHR;                 // 16 pieces per thread and panel
};

template <bool FLAGS, bool CLONE, int PHYS = 0>
__global__ __launch_bounds__(MmBlk4::NTHR, 1) void mm_mfma_blk4_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                                      uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
                                                                      FaultTab ft, uint8_t *__restrict__ detected)
{
    using G = MmBlk4;
    constexpr int NREP = 3, NS = 20 * NREP, HALF = NS / 2, NSET = 2 * NREP;
    constexpr bool DUP = CLONE;
    constexpr bool DUPADJ = DUP && COAST_MM4_DUP_ADJ != 0;
    constexpr int NSETS = COAST_MM4_SETS;
    static_assert(NSETS == 1 || NSETS == 2, "one or two register sets of raw s words");
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = wv & 1;   // column-tile lane
    const int HQ = wv >> 1; // row quarter: panel rows 32 HQ .. 32 HQ + 31; waves wv and wv + 4 (quarters HQ and HQ + 2 of one lane) share a SIMD
    const int l16 = lane & 15, kg = lane >> 4;
    constexpr int kSlabBase = G::A_PANEL;
    const int wbufOff = kSlabBase + L * G::LANE_LDS; // the lane's slab double buffer

    constexpr size_t nn = (size_t)G::N * G::N;
    const uint32_t stride = gridDim.x / G::NPANEL;
    const bool xcdMap = (gridDim.x % 16u) == 0u; // the two panel workgroups of a matrix on one XCD (hardware block b runs on XCD b % 8)
    const uint32_t slotX = blockIdx.x >> 3;
    const uint32_t mat0 = xcdMap ? (blockIdx.x & 7u) * (gridDim.x >> 4) + (slotX >> 1) : blockIdx.x >> 1;
    const int pnl = (int)(xcdMap ? slotX & 1u : blockIdx.x & 1u);
    auto matOf = [&](int item) __attribute__((always_inline)) { return mat0 + (uint32_t)item * stride; };
    auto rsrcOf = [&](const void *base, bool live, int bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(live ? base : (const void *)F), 0, live ? bytes : 0, 0x00020000);
    };
    auto rsFof = [&](int item) __attribute__((always_inline)) { return rsrcOf(F + matOf(item) * nn, matOf(item) < nblocks, (int)(nn * 4)); };
    auto rsSof = [&](int item) __attribute__((always_inline)) { return rsrcOf(S + matOf(item) * nn, matOf(item) < nblocks, (int)(nn * 4)); };
    const uint32_t *f = F + mat0 * nn, *s = S + mat0 * nn;
    __amdgpu_buffer_rsrc_t rsR = rsrcOf(R + mat0 * nn, true, (int)(nn * 4));
    auto freshLane = []() __attribute__((always_inline)) { return xmr_fresh_lane(); };
    auto voffRof = [&]() __attribute__((always_inline)) {
        const int l = freshLane();
        return ((4 * (l >> 4)) * G::N + (l & 15)) * 4;
    };
    auto flagsOf = [&](uint32_t m) __attribute__((always_inline)) {
        const bool on = FLAGS && detected != nullptr;
        return rsrcOf(on ? detected + m * nn : (const uint8_t *)F, on, (int)nn);
    };
    __amdgpu_buffer_rsrc_t rsD = flagsOf(mat0);
    auto launder = [](int v) __attribute__((always_inline)) {
        asm volatile("" : "+v"(v)); // a clone's address register is its own: not to be merged with the original's
        return v;
    };
    auto flagElem = [&](uint32_t mat, int row, int col) __attribute__((always_inline)) {
        if constexpr (FLAGS)
            if (detected != nullptr && mat < nblocks)
                detected[mat * nn + (size_t)row * G::N + col] = 1;
    };
    uint32_t stageMiss = 0; // this lane's words whose two staged copies differed

    // ---- f panel.  Thread (wave wv, lane l): row32 = 16 (wv / 4) + wv % 4 + 4 (l / 16) (0..31: a wave's four rows differ in bits 2-3, so that
    // their conversion stores fall into four different bank groups -- rows 4 wv .. + 3 met on the same sixteen banks, SQ_LDS_BANK_CONFLICT
    // 1.0e8 per launch, profiles/r06_mm_rocprofv3_summary.txt), k-quad in a slab kqi = l % 16.  Piece pc = 4 s + jj of the thread: k-slab s
    // (region s), panel row 32 jj + row32, words k = 64 s + 4 kqi .. + 3.  LDS: row * 1024 + p * 256 + (slot ^ (row & 15)) * 16 + (kqi & 3) * 4
    // with slot = 4 s + kqi / 4 (a fragment read of 16 rows and a conversion store of two rows x 16 k-quads both cover every bank group)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int wvRow = 16 * (wv >> 2) + (wv & 3); // the wave's first row of a 32-row piece group
    auto voffFof = [&]() __attribute__((always_inline)) {
        const int l = freshLane();
        return 4 * (l >> 4) * (G::N * 4) + (l & 15) * 16;
    };
    auto soffF = [&](int pc) __attribute__((always_inline)) { return ((pnl * G::BM + 32 * (pc & 3) + wvRow) * G::N + 64 * (pc >> 2)) * 4; };
    auto panelDst = [&](int pc) __attribute__((always_inline)) {
        const int l = freshLane();
        const int row32 = wvRow + 4 * (l >> 4), kqi = l & 15;
        const int d0 = row32 * G::ROW_A + (((kqi >> 2) ^ (row32 & 15)) * 16) + (kqi & 3) * 4;
        return (d0 ^ ((pc >> 2) * 64)) + (pc & 3) * 32 * G::ROW_A;
    };
    auto storePiece = [&](const u32x4_t raw, int pc) __attribute__((always_inline)) {
        const uint32_t y[4] = {mm_digits(raw[0]), mm_digits(raw[1]), mm_digits(raw[2]), mm_digits(raw[3])};
        uint32_t w[4];
        mm_transpose4(y, w);
        const int dst = panelDst(pc);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<uint32_t *>(smemP + p * G::PLANE_A + dst) = w[p];
    };
    // The first panel: pieces 0..9 (regions 0, 1 and half of region 2) are staged here; the first item's tile 0 runs the same bodies as every
    // later item's (BG = 2: half steps 5..7 of the panel replacement), which convert pieces 10..15 -- so pieces 10 and 11 are requested here.
    auto stageNow = [&](auto cntTag, int pc0) __attribute__((always_inline)) {
        constexpr int CNT = decltype(cntTag)::value;
        const __amdgpu_buffer_rsrc_t rsF = rsFof(0);
        u32x4_t pa[CNT];
#pragma unroll
        for (int u = 0; u < CNT; ++u)
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, voffFof(), soffF(pc0 + u), COAST_MM_AUX_F);
        if constexpr (CLONE) {
            u32x4_t pd[CNT];
#pragma unroll
            for (int u = 0; u < CNT; ++u)
                pd[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, launder(voffFof()), soffF(pc0 + u), COAST_MM_AUX_F);
            bool mis = false;
#pragma unroll
            for (int u = 0; u < CNT; ++u)
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    mis = mis || pa[u][d] != pd[u][d];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    const u32x4_t pc = __builtin_amdgcn_raw_buffer_load_b128(rsF, launder(voffFof()), soffF(pc0 + u), COAST_MM_AUX_F);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bool e = pa[u][d] == pd[u][d];
                        stageMiss += e ? 0u : 1u;
                        pa[u][d] = e ? pa[u][d] : pc[d];
                        if (!e) // (row of the word, column 0: the first element it reaches)
                            flagElem(mat0, pnl * G::BM + 32 * ((pc0 + u) & 3) + wvRow + 4 * (freshLane() >> 4), 0);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CNT; ++u)
            storePiece(pa[u], pc0 + u);
    };
    stageNow(std::integral_constant<int, 8>{}, 0);
    stageNow(std::integral_constant<int, 2>{}, 8);

    auto tileCol0 = [&](int g) __attribute__((always_inline)) { return (L + G::NLANE * ((g >> 2) & 7)) * G::CT; };

    // ---- s staging.  A wave converts quarter HQ of its lane's slab: k = 16 HQ + 4 (l / 16) + kk (kk = 0..3: one word per load), column l % 16.
    // Slab layout as in mm_mfma_blk2_kernel.hip: plane q, row colRow(c) of 64 bytes, its four 16-byte k-slots XORed with colSwz(c); this
    // thread's four bytes: slot HQ, dword l / 16 (32 lanes = 16 columns x 2 dwords: 32 different banks)
    auto colRow = [](int c) { return ((c & 7) << 1) | (c >> 3); };
    auto colSwz = [](int c) { return (c >> 1) & 3; };
    // ADDRESS REGISTERS (round 6, after the first uniform campaign of this kernel: profiles/r06_campaign_uniform_*): every LDS address of
    // the steps is ONE per-lane base register + a compile-time instruction offset (the slab buffer's parity is the step's position in its
    // tile).  Left to itself the compiler hoists `base + wave-uniform term` for every (plane, buffer) pair into registers that live for the
    // whole kernel -- fifteen of them, each a single point of failure every replica depends on (every flip of any bit of any lane a
    // silently wrong product: 91.2 % coverage against mm_mfma_blk3_kernel's 96.4 %).  Now: the fragment reads' bases are REPLICA-PRIVATE
    // registers (aOffR[r], offBR[r]: a flip there corrupts one replica's operands and is out-voted at the tile's votes); the conversion's
    // store base and the staging loads' offset exist twice under CLONE and are compared where they are used (a mismatch recomputes both
    // from a fresh lane id and counts one corrected error -- select(a == b, a, c) with the third copy evaluated lazily).
    const int voffS = ((4 * kg) * G::N + l16) * 4;
    const int dstS = wbufOff + colRow(l16) * G::KS + ((HQ ^ colSwz(l16)) * 16) + kg * 4; // (the lane's slab buffers included)
    auto slabOff = [&](int g) __attribute__((always_inline)) { return (((g & 3) * G::KS + 16 * HQ) * G::N + tileCol0(g)) * 4; };

    const int aOff = (32 * HQ + l16) * G::ROW_A + ((kg ^ l16) * 16);
    const int bOff = wbufOff + colRow(l16) * G::KS + ((kg ^ colSwz(l16)) * 16); // (the lane's slab buffers included)

    uint32_t agree = 0;            // votes of this lane whose three copies were equal (phantom votes included)
    uint32_t agreeS = 0;           // (COAST_MM4_VAR & 4: the same, summed over the wave's lanes in a scalar register)
    uint32_t nExec = 0, nReal = 0; // wave-uniform: votes executed / votes of tiles that exist (= the lane's __SYNC_COUNT)
    v4i_t acc[2][NREP][4];         // row blocks 2 HQ and 2 HQ + 1
#pragma unroll
    for (int rbz = 0; rbz < 2; ++rbz)
#pragma unroll
        for (int rz = 0; rz < NREP; ++rz)
#pragma unroll
            for (int pz = 0; pz < 4; ++pz)
                acc[rbz][rz][pz] = v4i_t{0, 0, 0, 0};

    // raw s words: set j % NSETS holds this wave's four words of slab j, requested in step j - 1 - NSETS behind the conversion of slab j - NSETS
    uint32_t pbs[NSETS][4];
    uint32_t dupS[DUPADJ ? NSETS : 1][4] = {}; // their clones (CLONE)
    u32x4_t bgRaw[2], dupF[2];             // the two f pieces of a half step (BG steps) and their clones
    dupF[0] = dupF[1] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) { // pieces 10 and 11 of the first panel: converted in the first half of step 0
        bgRaw[sel] = __builtin_amdgcn_raw_buffer_load_b128(rsFof(0), voffFof(), soffF(10 + sel), COAST_MM_AUX_F);
        if constexpr (DUP)
            dupF[sel] = __builtin_amdgcn_raw_buffer_load_b128(rsFof(0), launder(voffFof()), soffF(10 + sel), COAST_MM_AUX_F);
    }
    auto loadS = [&](int gs, auto setTag, auto kkTag) __attribute__((always_inline)) {
        constexpr int set = decltype(setTag)::value, kk = decltype(kkTag)::value;
        pbs[set][kk] = __builtin_amdgcn_raw_buffer_load_b32(rsSof(gs >> 5), voffS + kk * G::N * 4, slabOff(gs), 0);
    };
    int voffS2 = launder(voffS), dstS2 = launder(dstS); // the clones' own address registers (CLONE)
    auto loadDupS = [&](int gs, auto setTag, auto kkTag) __attribute__((always_inline)) {
        constexpr int set = decltype(setTag)::value, kk = decltype(kkTag)::value;
        dupS[DUPADJ ? set : 0][kk] = __builtin_amdgcn_raw_buffer_load_b32(rsSof(gs >> 5), voffS2 + kk * G::N * 4, slabOff(gs), 0);
    };
    // the conversion's store base against its clone, in front of the stores (cold path: both from a fresh lane id)
    int dstSv = dstS;
    auto verifyDstCold = [&]() __attribute__((always_inline)) {
        const int l = freshLane(), c = l & 15;
        const int fresh = wbufOff + (((c & 7) << 1) | (c >> 3)) * G::KS + ((HQ ^ ((c >> 1) & 3)) * 16) + (l >> 4) * 4;
        stageMiss += (dstSv != dstS2) ? 1u : 0u;
        dstSv = fresh;
        dstS2 = launder(fresh);
    };
    auto verifyDst = [&]() __attribute__((always_inline)) {
        if constexpr (DUP && (COAST_MM4_VAR & 2) == 0) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(dstSv != dstS2) != 0, 0))
                verifyDstCold();
        }
    };
    // in front of the first instruction that consumes a word of the set; gs = the slab it belongs to
    auto verifyS = [&](auto setTag, int gs) __attribute__((always_inline)) {
        constexpr int set = decltype(setTag)::value;
        constexpr int ds = DUPADJ ? set : 0;
        bool mis = false;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            mis = mis || pbs[set][kk] != dupS[ds][kk];
        if constexpr ((COAST_MM4_VAR & 2) != 0)
            mis = mis || dstSv != dstS2;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
            if constexpr ((COAST_MM4_VAR & 2) != 0)
                verifyDstCold();
            const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 5);
            const int so = slabOff(gs), l = freshLane();
            const int vo = ((4 * (l >> 4)) * G::N + (l & 15)) * 4;
            const bool exists = matOf(gs >> 5) < nblocks; // (a slab staged ahead for a matrix behind the batch's last: zeros, and no vote of anybody's)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t c = __builtin_amdgcn_raw_buffer_load_b32(rs, vo + kk * G::N * 4, so, 0);
                const bool e = pbs[set][kk] == dupS[ds][kk] || !exists;
                stageMiss += e ? 0u : 1u;
                pbs[set][kk] = e ? pbs[set][kk] : c;
                if (!e) // (first row of the panel, the word's column: the first element it reaches)
                    flagElem(matOf(gs >> 5), pnl * G::BM, tileCol0(gs) + (l & 15));
            }
        }
    };
    auto convNow = [&](auto setTag, int bufOff) __attribute__((always_inline)) {
        constexpr int set = decltype(setTag)::value;
        const uint32_t y[4] = {mm_digits(pbs[set][0]), mm_digits(pbs[set][1]), mm_digits(pbs[set][2]), mm_digits(pbs[set][3])};
        uint32_t w[4];
        mm_transpose4(y, w);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint32_t *>(smemP + dstSv + bufOff + q * G::PLANE_B) = w[q];
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using SEQ4 = std::make_integer_sequence<int, 4>;
    // prologue: slab 0 converted here (set 0); two sets: slab 1 in set 1 for step 0, slab 2 in set 0 for step 1; one set: slab 1 in set 0
    using SLAST = std::integral_constant<int, NSETS - 1>;
    for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadS(0, U0{}, kkTag); });
    if constexpr (NSETS == 2)
        for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadS(1, SLAST{}, kkTag); });
    if constexpr (DUP) {
        for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadDupS(0, U0{}, kkTag); });
        verifyS(U0{}, 0);
    }
    convNow(U0{}, 0);
    for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadS(NSETS, U0{}, kkTag); });
    if constexpr (DUP) {
        for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadDupS(1, SLAST{}, kkTag); }); // compared in step 0's first slot
        if constexpr (DUPADJ && NSETS == 2)
            for_each_index(SEQ4{}, [&](auto kkTag) __attribute__((always_inline)) { loadDupS(2, U0{}, kkTag); });
    }
    __syncthreads(); // panel 0 and the lanes' slab 0 are complete

    // ---- tile end, set by set: as in mm_mfma_blk3_kernel (sets 0-3 at slot 13 + 3 n of the tile's last step, sets 4, 5 at slot 1 + 6 n' of
    // the next step)
    constexpr int NNEXT = NREP - 1, NLAST = NSET - NNEXT;
    uint32_t teV[NREP - 1][4] = {};
    __amdgpu_buffer_rsrc_t rsRp = rsrcOf(R, false, 0), rsDp = rsRp;
    auto recombine = [&](auto rbTag, auto rrTag, auto iTag) __attribute__((always_inline)) {
        constexpr int rb = decltype(rbTag)::value, rr = decltype(rrTag)::value, i = decltype(iTag)::value;
        uint32_t t; // Horner: three v_lshl_add_u32
        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(acc[rb][rr][3][i]), "v"(acc[rb][rr][2][i]));
        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(t), "v"(acc[rb][rr][1][i]));
        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(t), "v"(acc[rb][rr][0][i]));
        return t;
    };
    auto voteStore = [&](int g, int voffR, uint32_t real, auto rbTag, auto iTag, uint32_t vLast) __attribute__((always_inline)) {
        constexpr int rb = decltype(rbTag)::value, i = decltype(iTag)::value;
        const uint32_t v0 = teV[0][i], v1 = teV[1][i];
        const bool e01 = v0 == v1, e02 = v0 == vLast;
        const uint32_t voted = e01 ? v0 : vLast; // select(a == b, a, c), synchronization.cpp:934-938
        const bool same = e01 && e02;
        if constexpr ((COAST_MM4_VAR & 4) != 0 && DUP) // (the lanes' agreements counted once per wave: 64 lanes x this count = the per-lane tallies' sum)
            agreeS += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(same));
        else
            agree += same ? 1u : 0u;
        nExec += 1u;
        nReal += real; // __SYNC_COUNT is counted where the vote happens
        const int erow = pnl * G::BM + (2 * HQ + rb) * 16 + i;
        __builtin_amdgcn_raw_buffer_store_b32(voted, rb == 0 ? rsR : rsRp, voffR, (erow * G::N + tileCol0(g)) * 4, COAST_MM_AUX_R);
        if constexpr (FLAGS)
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)1, rb == 0 ? rsD : rsDp, same ? 0x40000000 : (voffR >> 2), erow * G::N + tileCol0(g), 0);
    };
    auto teStage = [&](int g, int voffR, uint32_t real, auto setTag, auto iTag) __attribute__((always_inline)) {
        constexpr int set = decltype(setTag)::value, rb = set / NREP, rr = set % NREP;
        using RB = std::integral_constant<int, rb>;
        const uint32_t v = recombine(RB{}, std::integral_constant<int, rr>{}, iTag);
        if constexpr (rr == NREP - 1)
            voteStore(g, voffR, real, RB{}, iTag, v);
        else
            teV[rr][decltype(iTag)::value] = v;
    };
    auto teLast = [&](int g, int voffR, auto mTag) __attribute__((always_inline)) {
        constexpr int m = decltype(mTag)::value;
        if constexpr (m >= 13 && m % 3 == 1)
            teStage(g, voffR, 1u, std::integral_constant<int, (m - 13) / 12>{}, std::integral_constant<int, ((m - 13) / 3) % 4>{});
    };
    auto teNext = [&](int g, int voffR, uint32_t real, auto nTag) __attribute__((always_inline)) { // stage n' = 0 .. 4 NNEXT - 1
        constexpr int n = decltype(nTag)::value;
        teStage(g, voffR, real, std::integral_constant<int, NLAST + n / 4>{}, std::integral_constant<int, n % 4>{});
    };
    constexpr int kNextStride = 6;

    uint32_t fFirst = 0, fCount = 0;
    // word w of entry q of the upset table.  PHYS: through a descriptor that ends with the panel's entries, no vector register in the address --
    // an upset of any VGPR cannot send this read anywhere
    auto ftWord = [&](uint32_t q, int w) __attribute__((always_inline)) {
        if constexpr (PHYS == 2) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<DevFault *>(ft.list), 0, (int)((fFirst + fCount) * 16u), 0x00020000);
            return __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rs, 0, (int)(q * 16u) + 4 * w, 0));
        } else {
            const DevFault *fp = ft.list + q;
            return __builtin_amdgcn_readfirstlane(w == 1 ? fp->local : w == 2 ? fp->step : *reinterpret_cast<const uint32_t *>(&fp->replica));
        }
    };
    // COAST_SITE_MM_PREG (PHYS): the upset of this wave in the current item, if any: key = step of the item << 6 | slot; sel = register file << 9 |
    // register number; lane | bit << 8
    uint32_t pregKey = 0xffffffffu, pregSel = 0u, pregLaneBit = 0u;
    auto pregScan = [&]() __attribute__((always_inline)) {
        pregKey = 0xffffffffu;
#pragma unroll 1
        for (uint32_t q = fFirst, nq = 0; q < fFirst + fCount && nq < 64u; ++q, ++nq) { // (nq: a flipped count must not walk the table for ever)
            const uint32_t sw = ftWord(q, 2), packed = ftWord(q, 3);
            if (((packed >> 8) & 0xffu) == 7u /* COAST_SITE_MM_PREG */ && ((sw >> 16) & 7u) == (uint32_t)wv) {
                pregKey = (sw & 0x3ffu) | (((sw >> 29) & 1u) << 10);
                pregSel = (((sw >> 19) & 1u) << 9) | ((sw >> 20) & 511u);
                pregLaneBit = ((sw >> 10) & 63u) | (((packed >> 16) & 31u) << 8);
            }
        }
    };
    auto pregFlip = [&]() __attribute__((always_inline)) {
        const uint32_t idx = pregSel & 511u, bitMask = 1u << (pregLaneBit >> 8);
        if ((pregSel >> 9) == 0u) {
            const uint32_t vm = freshLane() == (int)(pregLaneBit & 63u) ? bitMask : 0u;
            uint32_t m0save;
            asm volatile("s_mov_b32 %0, m0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_set_gpr_idx_on %1, 0x9\n\ts_nop 1\n\tv_xor_b32 v0, v0, %2\n\ts_nop 1\n\t"
                         "s_set_gpr_idx_off\n\ts_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 1"
                         : "=&s"(m0save)
                         : "s"(idx), "v"(vm)
                         : "memory");
        } else {
            uint32_t tmp, m0save;
            asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\ts_movrels_b32 %0, s0\n\ts_xor_b32 %0, %0, %3\n\ts_nop 0\n\ts_movreld_b32 s0, %0\n\t"
                         "s_nop 2\n\ts_mov_b32 m0, %1\n\ts_nop 1"
                         : "=&s"(tmp), "=&s"(m0save)
                         : "s"(idx), "s"(bitMask)
                         : "scc", "memory");
        }
    };
    // ---- injector hook: the consequence of an armed upset on the replica's word is an additive constant (everything downstream is linear
    // mod 2^32), written on the replica's limb-0 sums before the tile's last step -- mm_mfma_kernel.hip, file header; .local = panel row << 8 | column
    auto tileHook = [&](int g) __attribute__((always_inline)) {
        const int col0 = tileCol0(g), prow0 = pnl * G::BM;
        bool hooked = false;
#pragma unroll 1
        for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) { // (PHYS: an upset of the count must not walk the table for minutes)
            const int fcol = (int)(ftWord(q, 1) & 255u);
            hooked = hooked || (fcol >= col0 && fcol < col0 + G::CT);
        }
        if (!hooked)
            return;
        uint32_t curKey = 0xffffffffu, curStep = 0xffffffffu;
        uint32_t dsum[3] = {0u, 0u, 0u}, am[3] = {0u, 0u, 0u}, bm[3] = {0u, 0u, 0u};
#pragma unroll 1
        for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) {
            const uint32_t local = ftWord(q, 1);
            const int frow = (int)(local >> 8), fcol = (int)(local & 255u);
            if (fcol < col0 || fcol >= col0 + G::CT)
                continue;
            const uint32_t fstep = ftWord(q, 2), packed = ftWord(q, 3);
            const uint32_t frep = packed & 0xffu, fsite = (packed >> 8) & 0xffu, m = 1u << ((packed >> 16) & 31u);
            if constexpr (PHYS != 0)
                if (fsite > (uint32_t)SITE_MM_OPB)
                    continue; // (a physical register upset: applied where the register lives)
            if (local != curKey) { // a new element: its replicas start from clean running deltas
                curKey = local;
                curStep = 0xffffffffu;
                dsum[0] = dsum[1] = dsum[2] = 0u;
            }
            if (fstep != curStep) { // operand masks belong to one MAC
                curStep = fstep;
                am[0] = am[1] = am[2] = bm[0] = bm[1] = bm[2] = 0u;
            }
            const uint32_t *fr = f + (prow0 + frow) * G::N, *sc = s + fcol;
            const uint32_t dprev = frep == 0u ? dsum[0] : frep == 1u ? dsum[1] : dsum[2];
            uint32_t delta = 0u;
            if (fsite == (uint32_t)SITE_MM_ACC) {
                const uint32_t kEnd = fstep < (uint32_t)G::N ? fstep : (uint32_t)G::N;
                uint32_t part = 0u; // this replica's accumulator before the MAC of k == step (step >= n: after the loop)
                for (uint32_t k = (uint32_t)lane; k < kEnd; k += 64u)
                    part += fr[k] * sc[k * G::N];
                const uint32_t pfx = __builtin_amdgcn_readfirstlane(wave_sum(part)) + dprev;
                delta = (pfx ^ m) - pfx;
            } else if (fstep < (uint32_t)G::N) {
                const uint32_t a = __builtin_amdgcn_readfirstlane(fr[fstep]), bq = __builtin_amdgcn_readfirstlane(sc[fstep * G::N]);
                const uint32_t ma = frep == 0u ? am[0] : frep == 1u ? am[1] : am[2];
                const uint32_t mb = frep == 0u ? bm[0] : frep == 1u ? bm[1] : bm[2];
                const uint32_t ma2 = fsite == (uint32_t)SITE_MM_OPA ? ma ^ m : ma, mb2 = fsite == (uint32_t)SITE_MM_OPB ? mb ^ m : mb;
                delta = (a ^ ma2) * (bq ^ mb2) - (a ^ ma) * (bq ^ mb);
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
                    if (frep == (uint32_t)rr) {
                        am[rr] = ma2;
                        bm[rr] = mb2;
                    }
            } else {
                continue; // an operand of a MAC that never runs
            }
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
                if (frep == (uint32_t)rr)
                    dsum[rr] += delta;
            // the replica's register: panel row -> (quarter, rb, lane group, i), column -> lane, replica -> block
            const int r16 = frow & 15;
            const bool mineLane = lane == (r16 >> 2) * 16 + (fcol - col0);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int rr = 0; rr < NREP; ++rr)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[rb][rr][0][i] += (int)((mineLane && frep == (uint32_t)rr && (frow >> 4) == 2 * HQ + rb && (r16 & 3) == i) ? delta : 0u);
        }
    };

    // ---- one pipeline step of this wave: the 60 MFMAs of slab `g` as six sets of ten (mm_mfma_blk3_kernel's step), the conversion of this
    // wave's quarter of slab g + 1 in the first half, and in the BG steps two f pieces of the next panel per half step
    // A fragments: two register sets of four in the steps without f work -- a set's fragments are requested a whole set (ten MFMAs) ahead, into
    // the other set (mm_mfma_blk3_kernel's COAST_MM3_ABUF: 1.4 % faster) --, ONE set in the BG steps (re-read behind each fragment's last
    // use): the sixteen registers that frees are where the two f pieces in flight and their clones live.  Both forms enter and leave a
    // step with a[0..3] = the first set's fragments.
    v4i_t a[8], b[NREP][4];
    int aOffR[NREP], offBR[NREP]; // replica-private fragment bases
#pragma unroll
    for (int r = 0; r < NREP; ++r) {
        aOffR[r] = launder(aOff);
        offBR[r] = launder(bOff);
    }
    int offAset = aOffR[0]; // the A base of the set whose fragments are being read: aOffR[replica] ^ 64 * (g % 4), a temporary of a few slots
    auto loadAi = [&](auto idxTag, auto pTag, int rbl, int off) __attribute__((always_inline)) { // plane p of row block rbl into register idx
        constexpr int p = decltype(pTag)::value, idx = decltype(idxTag)::value;
        a[idx] = *reinterpret_cast<const v4i_t *>(smemP + off + p * G::PLANE_A + rbl * 16 * G::ROW_A);
    };
    auto loadB = [&](auto rrTag, auto qTag, auto parTag) __attribute__((always_inline)) { // parTag: which of the lane's two slab buffers
        constexpr int rr = decltype(rrTag)::value, q = decltype(qTag)::value, par = decltype(parTag)::value;
        b[rr][q] = *reinterpret_cast<const v4i_t *>(smemP + offBR[rr] + par * G::B_BUF + q * G::PLANE_B);
    };
    for_each_index(SEQ4{}, [&](auto pTag) __attribute__((always_inline)) { loadAi(pTag, pTag, 0, offAset); });
    for_each_index(std::make_integer_sequence<int, NREP>{}, [&](auto rrTag) __attribute__((always_inline)) {
        for_each_index(SEQ4{}, [&](auto qTag) __attribute__((always_inline)) { loadB(rrTag, qTag, U0{}); });
    });

    auto step = [&](int g, auto firstTag, auto posTag, auto bgTag) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(firstTag)::value; // first slab of a tile: the sums start from zero, the previous tile's last stages run
        constexpr int POS = decltype(posTag)::value;     // g % 4
        constexpr int BG = decltype(bgTag)::value;       // 0: no f work; 1: the last tile of an item; 2: the first tile of the next item
        // the half steps of the panel replacement, K = 0..7: pieces 2 K and 2 K + 1 of the next panel are converted in half step K (region K / 2)
        constexpr int K1 = BG == 1 ? (POS == 2 ? 1 : POS == 3 ? 3 : -1) : BG == 2 ? (POS == 0 ? 5 : POS == 1 ? 7 : -1) : -1; // first half
        constexpr int K2 = BG == 1 ? (POS == 1 ? 0 : POS == 2 ? 2 : POS == 3 ? 4 : -1) : BG == 2 ? (POS == 0 ? 6 : -1) : -1; // second half
        constexpr bool PRELOAD = BG == 1 && POS == 1; // pieces 0 and 1 are requested in the half step in front of half step 0
        constexpr int CSET = NSETS == 2 ? (POS + 1) & 1 : 0; // register set of slab g + 1
        constexpr int CS = (K1 >= 0 || PRELOAD) ? 2 : COAST_MM4_CONV_STRIDE; // conversion stage stride (BG steps: the first ten slots, the f stages behind)
        constexpr int PARN = (POS + 1) & 1; // slab g + 1's buffer (an item has an even number of steps: the parity of g is that of POS)
        using PARNT = std::integral_constant<int, PARN>;
        // (the base of the next set's fragments: that set's own replica's register, XORed with the k-slab's swizzle where the reads start)
        auto setBase = [&](auto setTag) __attribute__((always_inline)) {
            constexpr int st = decltype(setTag)::value; // NSET = the next step's first set
            if constexpr (st == NSET)
                offAset = aOffR[0] ^ (((POS + 1) & 3) * 64);
            else
                offAset = aOffR[st % NREP] ^ (POS * 64);
        };
        const int bgItem = BG == 1 ? (g >> 5) + 1 : (g >> 5); // the item whose panel is being staged
        const uint32_t realPrev = g != 0 ? 1u : 0u;
        int voffR = 0;
        if constexpr (FIRST != 0 || POS == 3)
            voffR = voffRof();
        __builtin_amdgcn_sched_barrier(0);

        uint32_t y[4], t[4];
        uint32_t (&w)[4] = y;
        auto digits2 = [&](uint32_t x0, uint32_t x1, auto halfTag) __attribute__((always_inline)) {
            constexpr int hf = decltype(halfTag)::value;
            y[2 * hf] = mm_digits(x0);
            y[2 * hf + 1] = mm_digits(x1);
        };
        auto perm1 = [&]() __attribute__((always_inline)) {
            t[0] = __builtin_amdgcn_perm(y[1], y[0], 0x05010400u);
            t[1] = __builtin_amdgcn_perm(y[1], y[0], 0x07030602u);
            t[2] = __builtin_amdgcn_perm(y[3], y[2], 0x05010400u);
            t[3] = __builtin_amdgcn_perm(y[3], y[2], 0x07030602u);
        };
        auto perm2 = [&]() __attribute__((always_inline)) {
            w[0] = __builtin_amdgcn_perm(t[2], t[0], 0x05040100u);
            w[1] = __builtin_amdgcn_perm(t[2], t[0], 0x07060302u);
            w[2] = __builtin_amdgcn_perm(t[3], t[1], 0x05040100u);
            w[3] = __builtin_amdgcn_perm(t[3], t[1], 0x07060302u);
        };
        using CSETT = std::integral_constant<int, CSET>;
        auto convStage = [&](auto subTag) __attribute__((always_inline)) { // this wave's quarter of slab g + 1
            constexpr int sub = decltype(subTag)::value;
            if constexpr (sub == 0 && DUP)
                verifyS(CSETT{}, g + 1);
            if constexpr (sub == 0)
                digits2(pbs[CSET][0], pbs[CSET][1], U0{});
            else if constexpr (sub == 1)
                digits2(pbs[CSET][2], pbs[CSET][3], U1{});
            else if constexpr (sub == 2)
                perm1();
            else if constexpr (sub == 3)
                perm2();
            else {
                verifyDst();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(smemP + dstSv + PARN * G::B_BUF + q * G::PLANE_B) = w[q];
            }
        };
        // the f pieces' per-lane address registers, once per BG step (two registers of the sixteen the single A fragment set frees; recomputed
        // per piece they cost as many instructions as the piece's conversion)
        // (with the clones and two sets of raw s words the BG steps have no register left for them: recomputed per piece there)
        constexpr bool HOISTF = BG != 0 && (!DUP || NSETS == 1);
        int voffFh = 0, dstFh = 0;
        if constexpr (HOISTF) {
            voffFh = voffFof();
            dstFh = panelDst(0);
        }
        auto voffFget = [&]() __attribute__((always_inline)) { return HOISTF ? voffFh : voffFof(); };
        auto dstFget = [&]() __attribute__((always_inline)) { return HOISTF ? dstFh : panelDst(0); };
        auto bgLoad = [&](int pc, auto selTag) __attribute__((always_inline)) {
            constexpr int sel = decltype(selTag)::value;
            bgRaw[sel] = __builtin_amdgcn_raw_buffer_load_b128(rsFof(bgItem), voffFget(), soffF(pc), (DUP && (COAST_MM4_VAR & 1)) ? 0 : COAST_MM_AUX_F);
            if constexpr (DUP) // the clone, right behind the original: its own address register
                dupF[sel] = __builtin_amdgcn_raw_buffer_load_b128(rsFof(bgItem), launder(voffFget()), soffF(pc), COAST_MM_AUX_F);
        };
        auto verifyF = [&](int pc, auto selTag) __attribute__((always_inline)) {
            constexpr int sel = decltype(selTag)::value;
            bool mis = false;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                mis = mis || bgRaw[sel][d] != dupF[sel][d];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
                const u32x4_t c = __builtin_amdgcn_raw_buffer_load_b128(rsFof(bgItem), launder(voffFof()), soffF(pc), COAST_MM_AUX_F);
                const bool exists = matOf(bgItem) < nblocks;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bool e = bgRaw[sel][d] == dupF[sel][d] || !exists;
                    stageMiss += e ? 0u : 1u;
                    bgRaw[sel][d] = e ? bgRaw[sel][d] : c[d];
                    if (!e)
                        flagElem(matOf(bgItem), pnl * G::BM + 32 * (pc & 3) + wvRow + 4 * (freshLane() >> 4), 0);
                }
            }
        };
        auto bgStage = [&](auto kTag, auto selTag, auto subTag) __attribute__((always_inline)) { // piece 2 K + sel
            constexpr int K = decltype(kTag)::value, sel = decltype(selTag)::value, sub = decltype(subTag)::value, pc = 2 * K + sel;
            if constexpr (sub == 0 && DUP)
                verifyF(pc, selTag);
            if constexpr (sub == 0)
                digits2(bgRaw[sel][0], bgRaw[sel][1], U0{});
            else if constexpr (sub == 1) {
                digits2(bgRaw[sel][2], bgRaw[sel][3], U1{});
                if constexpr (K < 7)
                    bgLoad(pc + 2, selTag); // the same slot of the next half step
            } else if constexpr (sub == 2)
                perm1();
            else if constexpr (sub == 3)
                perm2();
            else {
                const int dst = (dstFget() ^ ((pc >> 2) * 64)) + (pc & 3) * 32 * G::ROW_A; // = panelDst(pc)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    *reinterpret_cast<uint32_t *>(smemP + dst + p * G::PLANE_A) = w[p];
            }
        };
        const v4i_t zero = {0, 0, 0, 0};
        const uint32_t pregSlot = PHYS == 2 ? pregKey - ((uint32_t)(g & 31) << 6) : 0xffffffffu; // the slot of THIS step the upset sits in front of, if any
        auto slot = [&](auto mTag) __attribute__((always_inline)) {
            constexpr int m = decltype(mTag)::value;
            constexpr int set = m / 10, j = m % 10, rb = set / NREP, rr = set % NREP;
            constexpr int p = j < 4 ? 0 : j < 7 ? 1 : j < 9 ? 2 : 3;
            constexpr int jj = j - (p == 0 ? 0 : p == 1 ? 4 : p == 2 ? 7 : 9);
            constexpr int q = 3 - p - jj;
            constexpr bool fromZero = FIRST != 0 && p == 0;
            constexpr int half = m / HALF, mh = m % HALF;
            if constexpr (PHYS == 2)
                if (pregSlot == (uint32_t)m)
                    pregFlip();
            constexpr bool ABUF = BG == 0;
            constexpr auto aIdx = [](int st, int pp) { return ABUF ? 4 * (st & 1) + pp : pp; }; // register of fragment pp of set st
            acc[rb][rr][p + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[aIdx(set, p)], b[rr][q], fromZero ? zero : acc[rb][rr][p + q], 0, 0, 0);
            if constexpr (ABUF && j < 4) { // the NEXT set's fragment j, into the other register set: a whole set ahead of its first use
                if constexpr (j == 0)
                    setBase(std::integral_constant<int, set + 1>{});
                loadAi(std::integral_constant<int, aIdx(set + 1, j)>{}, std::integral_constant<int, j>{}, set == NSET - 1 ? 0 : (set + 1) / NREP, offAset);
            }
            if constexpr (!ABUF && jj == 3 - p) { // last use of a[p] in this set: the next set's (the next step's first set behind the last)
                if constexpr (p == 0)
                    setBase(std::integral_constant<int, set + 1>{});
                loadAi(std::integral_constant<int, p>{}, std::integral_constant<int, p>{}, set == NSET - 1 ? 0 : (set + 1) / NREP, offAset);
            }
            if constexpr (rb == 1 && jj == 0) { // last use of b[rr][3 - p] in this step
                loadB(std::integral_constant<int, rr>{}, std::integral_constant<int, 3 - p>{}, PARNT{});
            }
            // conversion of slab g + 1: five stages in the first half
            if constexpr (half == 0 && mh % CS == 0 && mh / CS < 5)
                convStage(std::integral_constant<int, mh / CS>{});
            // the set just read is free behind stage 1: this wave's words of slab g + 1 + NSETS, one load per slot
            {
                constexpr int s1 = CS + 1; // the slot behind stage 1
                if constexpr (half == 0 && mh >= s1 && mh < s1 + 4)
                    loadS(g + 1 + NSETS, CSETT{}, std::integral_constant<int, mh - s1>{});
                if constexpr (DUPADJ && half == 0 && mh >= s1 + 4 && mh < s1 + 8)
                    loadDupS(g + 1 + NSETS, CSETT{}, std::integral_constant<int, mh - s1 - 4>{});
            }
            if constexpr (DUP && !DUPADJ && half == 1 && mh >= 1 && mh < 5) // the clones of slab g + 2 (set g % 2), compared in the next step's first slot
                loadDupS(g + 2, std::integral_constant<int, NSETS == 2 ? POS & 1 : 0>{}, std::integral_constant<int, mh - 1>{});
            // f pieces of the panel replacement: stages at the even slots 10..28 of a half step (piece A: 10..18, piece B: 20..28)
            if constexpr (PRELOAD && m == 11)
                bgLoad(0, U0{});
            if constexpr (PRELOAD && m == 21)
                bgLoad(1, U1{});
            {
                constexpr int K = half == 0 ? K1 : K2;
                if constexpr (K >= 0 && mh >= 10 && mh % 2 == 0)
                    bgStage(std::integral_constant<int, K>{}, std::integral_constant<int, (mh - 10) / 10>{}, std::integral_constant<int, ((mh - 10) / 2) % 5>{});
            }
            if constexpr (POS == 3)
                teLast(g, voffR, mTag);
            if constexpr (FIRST != 0 && m % kNextStride == 1 && m / kNextStride < 4 * NNEXT)
                teNext(g - 1, voffR, realPrev, std::integral_constant<int, m / kNextStride>{});
            // the workgroup's one barrier per step, behind the last slot of the first half: slab g + 1 is complete, slab g's buffer is free,
            // the f region written in this half step is complete; the B fragments of slab g + 1 are read from the next slot on
            if constexpr (m == HALF - 1)
                __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_index(std::make_integer_sequence<int, NS>{}, slot);
    };

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    int gLast = 3;
    uint32_t anyTile = 0u;
    // An item's 32 steps: [tile 0: BG2 POS 0, BG2 POS 1] [tiles 0..5: POS 2, POS 3, the next tile's POS 0, POS 1] [tile 6: POS 2, POS 3, tile 7's POS 0]
    // [tile 7: BG1 POS 1, 2, 3].  Straight-line code and ONE clean inner loop: two alternative bodies that join (a diamond per step) made the
    // register allocator spill ~400 registers, an inner loop with a mid-loop exit ~150 (accumulator tuples reloaded in front of MFMAs) --
    // at the price of twelve inlined step bodies instead of nine.
    // (PHYS: a second, independent bound -- an upset of the loop's registers must not turn a campaign run into a walk through memory)
    const uint32_t itemCap = PHYS == 2 ? (nblocks + stride - 1u) / stride : 0xffffffffu;
#pragma unroll 1
    for (int item = 0; matOf(item) < nblocks && (uint32_t)item < itemCap; ++item) {
        const uint32_t mat = matOf(item);
        if (ft.range) {
            const uint2 rg = ft.range[mat * (uint32_t)G::NPANEL + (uint32_t)pnl];
            fFirst = __builtin_amdgcn_readfirstlane(rg.x);
            fCount = __builtin_amdgcn_readfirstlane(rg.y);
        }
        if constexpr (PHYS == 2)
            pregScan();
        if (item > 0) { // hand-over: regions 0 and 1 of the panel (and half of 2) are this item's already, the rest follows in its first two steps
            f = F + mat * nn;
            s = S + mat * nn;
            rsR = rsrcOf(R + mat * nn, true, (int)(nn * 4));
            rsD = flagsOf(mat);
        }
        const int gI = item * G::SPP;
        step(gI, T1{}, T0{}, T2{}); // + the previous tile's last eight stages (the previous item's resources at a hand-over)
        rsRp = rsR;
        rsDp = rsD;
        step(gI + 1, T0{}, T1{}, T2{});
        [[maybe_unused]] uint32_t tileGuard = 0u; // (PHYS: the same for the tile counter)
#pragma unroll 1
        for (int tile = 0; tile < G::TPW - 2; ++tile) {
            if constexpr (PHYS == 2)
                if (tileGuard++ >= (uint32_t)G::TPW)
                    break;
            const int g0 = gI + tile * G::NSLAB;
            step(g0 + 2, T0{}, T2{}, T0{});
            if (fCount != 0u)
                tileHook(g0);
            step(g0 + 3, T0{}, T3{}, T0{});
            step(g0 + 4, T1{}, T0{}, T0{});
            rsRp = rsR;
            rsDp = rsD;
            step(g0 + 5, T0{}, T1{}, T0{});
        }
        {
            const int g0 = gI + (G::TPW - 2) * G::NSLAB;
            step(g0 + 2, T0{}, T2{}, T0{});
            if (fCount != 0u)
                tileHook(g0);
            step(g0 + 3, T0{}, T3{}, T0{});
            step(g0 + 4, T1{}, T0{}, T0{});
            rsRp = rsR;
            rsDp = rsD;
        }
        const int g7 = gI + (G::TPW - 1) * G::NSLAB; // the item's last tile: the next panel's regions 0, 1 and half of 2
        step(g7 + 1, T0{}, T1{}, T1{});
        step(g7 + 2, T0{}, T2{}, T1{});
        if (fCount != 0u)
            tileHook(g7);
        step(g7 + 3, T0{}, T3{}, T1{});
        gLast = g7 + 3;
        anyTile = 1u;
    }
    { // the last tile's last eight stages: nothing left to hide them behind
        const int voffR = voffRof();
        for_each_index(std::make_integer_sequence<int, 4 * NNEXT>{}, [&](auto nTag) __attribute__((always_inline)) { teNext(gLast, voffR, anyTile, nTag); });
    }

    __syncthreads();
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemP + kSlabBase);
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();
    // TMR_ERROR_CNT = the votes whose copies were not all equal + the staging compares that failed (a corrected word each); __SYNC_COUNT =
    // the votes of tiles that exist
    if constexpr ((COAST_MM4_VAR & 4) != 0 && DUP) { // lane 0 carries the wave's mismatches (64 nExec - agreeS), every lane its own staging misses
        const uint32_t waveMiss = 64u * nExec - agreeS;
        block_tally((lane == 0 ? waveMiss : 0u) + stageMiss, nReal, 0u, sCnt, ctr, blockIdx.x);
    } else
        block_tally(nExec - agree + stageMiss, nReal, 0u, sCnt, ctr, blockIdx.x);
}

} // namespace coast
