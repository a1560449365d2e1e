// mm_mfma_blk3_kernel.hip -- the register-block TMR matrix_multiply kernel of round 4: mm_mfma_blk2_kernel's geometry (read that
// file's header first) with two changes.
//
// 1. EVERY OPERAND LOAD FROM LDS IS REPLICATED.  (Not the global -> LDS staging loads in front of them: by default those fill ONE register
//    set whose upsets are common-mode; COAST_F_CLONE_STAGING -- the CLONE instantiation below -- clones them too.  ADVICE r4.)  The pass
//    clones every load of the protected region and its users (cloning.cpp:2187-2209; under -noMemReplication the three loads keep one
//    address, :2247-2255): an upset in a loaded f[i][k] or s[k][j] operand register is out-voted (tests/mm_common/mm_common_tmr.c:13).  mm_mfma_blk2_kernel had replica-private B fragments but ONE A fragment set
//    for the three replicas.  Here a step's 60 MFMAs run as six sets of ten -- set = (row block, replica) -- and a set reads ITS
//    OWN four A fragments from the LDS panel (the load is replicated, the memory is not): an A register is live for one replica's
//    1-4 MFMAs, 24 instead of 8 ds_read_b128 per step and wave.
//
// 2. THE NON-MFMA WORK IS SPREAD OVER THE STEP.  tools/mfma_probe3 (profiles/r04_mfma_probes.txt): beside v_mfma_i32_16x16x64_i8 on
//    random bytes every VALU filler and every ds_read_b128 costs matrix-pipe time at any number of resident waves (two waves per
//    SIMD: 3.76 POP/s bare, 3.36 / 2.97 with one / two fillers per MFMA, 3.29 with a fragment read every other MFMA).
//    mm_mfma_blk2_kernel packed the s conversion and a tile end into the first 30 slots of a step and left the last 30 bare.  Here
//      * a slab's conversion takes a full step: the wave that owns slab g + 2 converts its first staging round behind the barrier
//        of step g (slots 30..57, into the buffer slab g just left) and its second round in slots 0..27 of step g + 1; its partner
//        on the SIMD does the same one step later -- at any time exactly one wave of a pair converts, at ~1 VALU per slot;
//      * the background f piece goes into the half steps in which the wave does not convert, and only in the two middle steps of
//        a tile;
//      * a set's sums are final ten slots after it started, so the tile end is a chain of 24 small stages (recombine one element
//        row of one set: 3 VALU; every third set: + vote + store) from slot 13 of a tile's last step to slot 43 of the next
//        tile's first step, each set recombined before its accumulators restart.
//    Measured (A/B on one box, profiles/r04_mm_ab.txt): the replicated A loads cost 5 % (7.10 ms against 6.76 with COAST_MM3_KNOCK=1,
//    one A fragment set per row block) -- 288 instead of 160 fragment reads per workgroup and step, the LDS array 43 % busy
//    (SQ_LDS_IDX_ACTIVE, profiles/r04_mm_rocprofv3_summary.txt); the even spread itself does not pay: with shared A this schedule runs
//    3 % behind mm_mfma_blk2_kernel's (6.76 against 6.57), and mm_mfma_blk2_kernel's schedule with this kernel's sets of ten and
//    private A fragments runs 7.10 against this kernel's 6.99.  Removing the per-step barrier makes it slower (7.26).
//    One workgroup barrier per step (slot 29), as before; the item hand-over needs none of its own (the next panel is complete
//    behind the barrier of the item's last step, and its A fragments are read after it).
//
// DWC and the unprotected mode run the same kernel (NREP = 2 / 1): 2 NREP sets of ten MFMAs per step, one conversion / background stage per
// NREP slots, the barrier behind the first half's last slot, the tile end set by set (DWC: replica 0's value is stored, a failed compare flags
// the element; unprotected: recombine and store).
//
// __SYNC_COUNT / TMR_ERROR_CNT: the votes of tiles that do not exist (the first step's look-back, a workgroup without items) are
// executed on zeroed accumulators and are not counted: `real` is a wave-uniform predicate, not a compensation constant.
#include <type_traits>
#include <utility>

#include "xmr.hpp"

// development: timing experiments (1: the replicas of a row block share one A fragment set, as in mm_mfma_blk2_kernel -- results
// still right; 2: no per-step barrier; 4: no r stores; 8: no tile end at all; 16: no s conversion arithmetic / stores; 32: no MFMAs;
// 64: no background f piece; 128: no s loads; 256: no fragment reads behind the first; 512: fragment reads of fixed data; 1024: no B
// reloads; 2048: no A reloads -- results wrong from 2 on.  CAREFUL: without the tile end (8) the accumulators are dead and the MFMAs go with
// them; without A reloads (256, 2048) both row blocks multiply the same registers and the compiler keeps one of them -- those builds time
// less work than they name (profiles/r04_mm_knockouts.txt))
#ifndef COAST_MM3_KNOCK
#define COAST_MM3_KNOCK 0
#endif
// two A fragment sets: a set's four A fragments are requested a whole set (ten MFMAs) ahead, into the other buffer (0: one set, re-read behind
// each fragment's last use).  1.4 % faster (7.10 -> 7.00 ms, profiles/r04_mm_ab.txt) and the allocator fits it: 256 VGPRs, 1 spill outside the steps.
// (Round 5 measured a ring of six fragment registers with the same leads -- eight registers fewer -- at + 1.7 %: profiles/r05_mm_clone_ab.txt.)
#ifndef COAST_MM3_ABUF
#define COAST_MM3_ABUF 1
#endif

// CLONE (round 5, COAST_F_CLONE_STAGING): THE STAGING PATH AS A CLONED LOAD.  The kernel loads every raw word of s and f once, converts it once
// and writes the byte planes into the LDS image every replica reads: an upset of a staging register (pbs, bgRaw: 20 of 256 VGPRs, alive for
// more than a pipeline step) is common-mode -- 233 of 501 / 76 of 79 real flips there stored wrong matrices with TMR_ERROR_CNT unchanged
// (profiles/r05_campaign_physical_real_all_seed0_5000.txt).  The pass clones the load (cloning.cpp:2187-2209; one address under
// -noMemReplication, :2247-2255).  With CLONE every raw word is loaded a SECOND time (an L2 hit) half a step before its conversion and compared
// with the staged copy in front of the first instruction that consumes it.  TMR: a mismatch loads the word a third time through a freshly
// computed address and keeps select(a == b, a, c) (synchronization.cpp:934-938, the third copy evaluated lazily), TMR_ERROR_CNT + 1 per word;
// DWC: the compare that fails counts a detected item and flags the first element the word reaches.  What stays single: the five conversion
// stages' temporaries (digits -> byte planes, alive for three to six slots each).
// PRICE: the register file is full (256 VGPRs, two waves per SIMD) -- the twelve clone registers push address registers into scratch, whose
// reloads drain the whole VMEM queue: first forms ran + 30 %; the shipped form (two-word clones, one compare per round, the f piece requested one
// step ahead) + 10-12 % (profiles/r05_mm_clone_ab.txt, profiles/r06_mm_blk4_ab.txt).  Coverage with and without: docs/design/campaign.md (the
// one table; round 5's own figures came from a broken register-selector decode).  Round 6: the clones are the library's default (ABI 8,
// COAST_F_SINGLE_STAGING opts out) and the TMR default kernel is mm_mfma_blk4_kernel, where they cost + 6 %.

// round 5, measured and NOT the default: WIDE staging loads of s.  A buffer load costs this kernel about what seven LDS reads cost
// (profiles/r05_mm_clone_ab.txt), and the s staging issues eight two-word loads per slab and wave (a lane: two adjacent columns x eight k).
// 1: four four-word loads -- a lane owns four adjacent columns x four k (lane -> column quad l % 4, k-quad l / 4), one load instruction
// fetches sixteen full tile rows -- into the same LDS image through the same conflict-free stores (tests/test_lds_layouts_cpu.py), requested
// in the duty step behind the last read of the registers they replace (slots 21 - 48).  - 0.5 % kernel time on two boxes -- and 1.7 points
// of coverage: all sixteen raw words of a lane now sit in registers from one request to the last group's conversion, and the uniform
// register-file campaign (same 5000 draws) went from 223 wrong products to 308 (95.5 -> 93.8 %; the sixteen registers: 60 - 75 % of their
// hits silent against 35 - 50 % for the two-word form).  Half a percent of time is not worth a third more silent corruptions.
#ifndef COAST_MM3_WIDE
#define COAST_MM3_WIDE 0
#endif

namespace coast {

// PHYS (round 4): the instantiation that runs when a COAST_SITE_MM_VGPR upset is armed -- a REAL exclusive-or on one bit of one lane of a
// named vector register of this kernel while it computes: an A-operand fragment of one replica's set of ten MFMAs (flipped before the
// set's first MFMA; the next set re-reads its own fragment), a B-operand fragment of one replica (flipped at the start of the k-slab's
// step, used by the replica's sets of both row blocks, re-read for the next step), or a limb-sum accumulator (flipped at the start of a
// step that is not the tile's first; it stays until the tile's vote).  The hooks sit in front of the MFMAs they precede; the clean
// instantiations do not contain them.
// PHYS == 2, SITE_MM_PREG (round 5; an instantiation of its own: its 60 hook points per step body cost the named-register instantiation a
// factor in speed): the same kind of upset, but of ANY register of the wave, named by its PHYSICAL number -- v0 .. v255 through the
// VGPR index mode (s_set_gpr_idx_on), s0 .. s101 through s_movrels / s_movreld -- in front of any of a step's MFMA slots, the compiler knowing
// nothing of it: what the reference's injector does when it draws a register of the core (simulation/platform/resources/injector.py:70-72,
// 237-260).  Address registers, lane constants, loop counters, descriptors, the staging clones, whatever the allocator put there.
enum { SITE_MM_VGPR = 6, SITE_MM_PREG = 7 };
template <int NREP, bool FLAGS, int PHYS = 0, bool CLONE = false>
__global__ __launch_bounds__(MmBlk2<NREP>::NTHR, 1) void mm_mfma_blk3_kernel(const uint32_t *__restrict__ F,
                                                                            const uint32_t *__restrict__ S,
                                                                            uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
                                                                            FaultTab ft, uint8_t *__restrict__ detected)
{
    using G = MmBlk2<NREP>;
    static_assert(NREP >= 1 && NREP <= 3, "a step = 2 NREP sets of ten MFMAs per wave");
    // DWC and the unprotected mode (round 4) run the same kernel with four / two sets per step: NS = 20 NREP slots, the conversion and
    // background stages at the same relative places (one per NREP slots), the barrier in the middle, the tile end set by set.
    constexpr int NS = 20 * NREP, HALF = NS / 2, NSET = 2 * NREP;
    constexpr bool DUP = CLONE && NREP > 1; // (the unprotected mode has nothing to compare with)
    constexpr bool WIDE = COAST_MM3_WIDE != 0 && !DUP;
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wv & 3; // column-tile lane; waves wv and wv + 4 land on the same SIMD
    const int l16 = lane & 15, kg = lane >> 4;
    constexpr int kSlabBase = 2 * G::A_PANEL;
    const int wbufOff = kSlabBase + wave * G::PAIR_LDS; // the pair's slab double buffer

    constexpr size_t nn = (size_t)G::N * G::N;
    const uint32_t stride = gridDim.x / G::NPANEL;
    const bool xcdMap = (gridDim.x % 32u) == 0u;
    const uint32_t slotX = blockIdx.x >> 3;
    const uint32_t mat0 = xcdMap ? (blockIdx.x & 7u) * (gridDim.x >> 5) + (slotX >> 2) : blockIdx.x >> 2;
    const int pnl = (int)(xcdMap ? slotX & 3u : blockIdx.x & 3u);
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

    // ---- f panels: piece j of a panel for thread t (512 threads): row 8 j + t / 64, k-quad t % 64 (as in mm_mfma_blk2_kernel)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int soffFw = wv * G::N * 4;
    auto voffFof = [&]() __attribute__((always_inline)) { return freshLane() * 16; };
    auto panelDst = [&](int j) __attribute__((always_inline)) {
        const int l = freshLane();
        const int d0 = wv * G::N + (((l >> 2) ^ wv) * 16) + (l & 3) * 4;
        return (d0 ^ ((j & 1) * 128)) + j * 8 * G::N;
    };
    // a staging compare that failed (cold path).  TMR: third copy, select(a == b, a, c), one corrected error per word; DWC: replica 0's word
    // stays, one detected item per word
    uint32_t stageMiss = 0; // this lane's words whose two staged copies differed
    auto launder = [](int v) __attribute__((always_inline)) {
        asm volatile("" : "+v"(v)); // the clone's address register is its own: not to be merged with the original's
        return v;
    };
    {
        const __amdgpu_buffer_rsrc_t rsF = rsFof(0);
        u32x4_t pa[G::A_PER_THR];
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u)
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, voffFof(), soffFw + (pnl * G::BM + u * 8) * G::N * 4, COAST_MM_AUX_F);
        if constexpr (DUP) { // the first panel is staged in one go: its clone too
            u32x4_t pd[G::A_PER_THR];
#pragma unroll
            for (int u = 0; u < G::A_PER_THR; ++u)
                pd[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, launder(voffFof()), soffFw + (pnl * G::BM + u * 8) * G::N * 4, COAST_MM_AUX_F);
            bool mis = false;
#pragma unroll
            for (int u = 0; u < G::A_PER_THR; ++u)
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    mis = mis || pa[u][d] != pd[u][d];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
#pragma unroll
                for (int u = 0; u < G::A_PER_THR; ++u) {
                    const u32x4_t pc = __builtin_amdgcn_raw_buffer_load_b128(rsF, launder(voffFof()), soffFw + (pnl * G::BM + u * 8) * G::N * 4, COAST_MM_AUX_F);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const bool e = pa[u][d] == pd[u][d];
                        stageMiss += e ? 0u : 1u;
                        if constexpr (NREP == 3)
                            pa[u][d] = e ? pa[u][d] : pc[d];
                        if constexpr (FLAGS)
                            if (!e && detected != nullptr && mat0 < nblocks) // (row of the word, column 0: the first element it reaches)
                                detected[mat0 * nn + (size_t)(pnl * G::BM + u * 8 + wv) * G::N] = 1;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u) {
            const uint32_t y[4] = {mm_digits(pa[u][0]), mm_digits(pa[u][1]), mm_digits(pa[u][2]), mm_digits(pa[u][3])};
            uint32_t w[4];
            mm_transpose4(y, w);
            const int dst = panelDst(u);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint32_t *>(smemP + p * G::PLANE_A + dst) = w[p];
        }
    }
    // background piece of a bg step (the two middle steps of a tile): piece 2 * tile + (g % 4 - 1) of the NEXT item's panel
    auto bgPiece = [](int g) { return 2 * ((g >> 2) & 3) + ((g & 3) - 1); };
    auto bgLoad = [&](int g) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsFof((g >> 4) + 1), voffFof(), soffFw + (pnl * G::BM + 8 * bgPiece(g)) * G::N * 4, COAST_MM_AUX_F);
    };
    auto bgLoadDup = [&](int g) __attribute__((always_inline)) { // the clone's load: its own address register
        return __builtin_amdgcn_raw_buffer_load_b128(rsFof((g >> 4) + 1), launder(voffFof()), soffFw + (pnl * G::BM + 8 * bgPiece(g)) * G::N * 4, COAST_MM_AUX_F);
    };

    auto tileCol0 = [&](int g) __attribute__((always_inline)) { return (wave + G::NLANE * ((g >> 2) & 3)) * G::CT; };

    // s staging and the slab layout: as in mm_mfma_blk2_kernel.hip (conflict-free fragment reads and conversion stores)
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    auto colRow = [](int c) { return ((c & 7) << 1) | (c >> 3); };
    auto colSwz = [](int c) { return (c >> 1) & 3; };
    // ADDRESS REGISTERS (round 6; mm_mfma_blk4_kernel.hip has the story): every LDS address of the slab buffers is ONE per-lane base register
    // (the pair's buffer offset folded in) + a compile-time instruction offset -- the buffer's parity is the step's position in its tile.
    // With `base + wave-uniform term` computed per use, the DWC and unprotected instantiations (which have registers to spare) kept some twenty
    // hoisted sums alive for the whole kernel, each a single point of failure both replicas depend on: uniform-campaign coverage of DWC 88.4 %
    // (profiles/r06_campaign_uniform_DWC_single_5000.txt).  PRIV (DWC): the fragment reads' bases are replica-private registers too, and
    // under CLONE the conversion's store base and the staging loads' offset exist twice (compared / used by the clone loads).  The TMR
    // instantiation keeps its single base registers: its register file is full (and mm_mfma_blk4_kernel is the TMR default).
    constexpr bool PRIV = NREP == 2;
    constexpr bool FOLD = NREP < 3; // (TMR: byte-identical to round 5's kernel)
    const int wbufFold = FOLD ? wbufOff : 0, wbufRest = FOLD ? 0 : wbufOff;
    const int voffB = ((4 * (lane >> 3)) * G::N + 2 * (lane & 7)) * 4;
    const int dstB0c = wbufFold + colRow(2 * (lane & 7)) * G::KS + ((((lane >> 3) >> 2) ^ colSwz(2 * (lane & 7))) * 16) + ((lane >> 3) & 3) * 4;
    int dstB0v = dstB0c; // (PRIV + CLONE: replaced by a fresh value when its clone disagrees.  The low 13 bits of the pair's buffer offset are
                         // zero: the XOR / adds below stay inside the lane's own bits)
    auto dstB = [&](int u, int h) __attribute__((always_inline)) { return ((PRIV ? dstB0v : dstB0c) ^ (u * 32)) + h * 2 * G::KS; };
    constexpr int kRoundOff = 8 * 4 * G::N * 4;
    // WIDE: lane -> (column quad cq = l % 4: columns 4 cq .. + 3; k-quad l / 4: rows 4 (l / 4) .. + 3 of the slab).  Column c = 4 cq + h sits in
    // row colRow(c) = 8 (cq % 2) + 2 h + cq / 2 of a plane, its slots XORed with colSwz(c) = 2 (cq % 2) ^ (h / 2)
    const int voffW = ((4 * (lane >> 2)) * G::N + 4 * (lane & 3)) * 4;
    const int dstW0 = wbufFold + (8 * (lane & 1) + ((lane & 3) >> 1)) * G::KS + (((lane >> 4) ^ (2 * (lane & 1))) * 16) + ((lane >> 2) & 3) * 4;
    auto dstW = [&](int h) __attribute__((always_inline)) { return (dstW0 ^ ((h >> 1) * 16)) + h * 2 * G::KS; };
    auto slabOff = [&](int g) __attribute__((always_inline)) { return ((g & 3) * G::KS * G::N + tileCol0(g)) * 4; };

    const int aOff = l16 * G::N + ((kg ^ l16) * 16);
    const int bOff = wbufFold + colRow(l16) * G::KS + ((kg ^ colSwz(l16)) * 16);
    auto panelOff = [&](int g) __attribute__((always_inline)) { return ((g >> 4) & 1) * G::A_PANEL + (aOff ^ ((g & 3) * 64)); };

    auto run = [&](auto hTag) __attribute__((always_inline)) {
        constexpr int H = decltype(hTag)::value;
        uint32_t agree = 0;                 // votes of this lane whose three copies were equal (phantom votes included)
        uint32_t nExec = 0, nReal = 0;      // wave-uniform: votes executed / votes of tiles that exist (= the lane's __SYNC_COUNT)
        uint32_t detItems = 0;
        v4i_t acc[2][NREP][4]; // row blocks 2 H and 2 H + 1
#pragma unroll
        for (int rbz = 0; rbz < 2; ++rbz)
#pragma unroll
            for (int rz = 0; rz < NREP; ++rz)
#pragma unroll
                for (int pz = 0; pz < 4; ++pz)
                    acc[rbz][rz][pz] = v4i_t{0, 0, 0, 0};
        // raw s words of this wave's slabs (the slabs of one parity): round 0 of slab g + 2 is converted in the second half of a
        // step g in which the wave is off duty, round 1 in the first half of step g + 1 (duty); each register set is reloaded in
        // the duty step with the slab two further on
        u32x2_t pbs[G::B_ROUNDS][4];
        u32x4_t pw[4]; // WIDE: row 4 (l / 4) + kk of the slab, columns 4 (l % 4) .. + 3
        u32x4_t bgRaw;
        // conversion group grp = 0..3 of a slab: four consecutive k of one column.  Two-word loads: staging round grp / 2, column grp % 2 of the
        // lane's pair; WIDE: column grp of the lane's quad
        auto rawWord = [&](auto grpTag, auto kkTag) __attribute__((always_inline)) {
            constexpr int grp = decltype(grpTag)::value, kk = decltype(kkTag)::value;
            if constexpr (WIDE)
                return pw[kk][grp];
            else
                return pbs[grp / 2][kk][grp % 2];
        };
        auto dstGrp = [&](int grp) __attribute__((always_inline)) { return WIDE ? dstW(grp) : dstB(grp / 2, grp % 2); };
        auto loadSlabW = [&](int gs) __attribute__((always_inline)) {
            const int so = slabOff(gs);
            const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pw[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, voffW + kk * G::N * 4, so, 0);
        };
        auto convGroupNowW = [&](auto grpTag, int bufOff) __attribute__((always_inline)) {
            constexpr int grp = decltype(grpTag)::value;
            const uint32_t y[4] = {mm_digits(pw[0][grp]), mm_digits(pw[1][grp]), mm_digits(pw[2][grp]), mm_digits(pw[3][grp])};
            uint32_t w[4];
            mm_transpose4(y, w);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint32_t *>(smemP + bufOff + q * G::PLANE_B + dstW(grp)) = w[q];
        };

        auto loadRound = [&](int gs, auto uTag) __attribute__((always_inline)) {
            constexpr int u = decltype(uTag)::value;
            const int so = slabOff(gs);
            const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pbs[u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rs, voffB + kk * G::N * 4, so + u * kRoundOff, 0);
        };
        auto convRound = [&](auto uTag, int bufOff) __attribute__((always_inline)) {
            constexpr int u = decltype(uTag)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t y[4] = {mm_digits(pbs[u][0][h]), mm_digits(pbs[u][1][h]), mm_digits(pbs[u][2][h]), mm_digits(pbs[u][3][h])};
                uint32_t w[4];
                mm_transpose4(y, w);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(smemP + bufOff + q * G::PLANE_B + dstB(u, h)) = w[q];
            }
        };
        using U0 = std::integral_constant<int, 0>;
        using U1 = std::integral_constant<int, 1>;
        // ---- the clones of the staged words (CLONE).  dupS[kk]: the second copy of pbs[u][kk] (two adjacent words of one row of s) of the
        // round u that is converted next -- requested behind the compare of the round before, 26-29 slots ahead of its own.  dupF: the second
        // copy of bgRaw.
        u32x2_t dupS[4] = {};
        u32x4_t dupF = {0u, 0u, 0u, 0u};
        int voffB2 = voffB, dstB02 = dstB0c; // PRIV: the clones' own offset register, the store base's clone
        if constexpr (PRIV && DUP) {
            voffB2 = launder(voffB);
            dstB02 = launder(dstB0c);
        }
        auto dupLoadS = [&](auto uTag, auto kkTag, const __amdgpu_buffer_rsrc_t rs, int so) __attribute__((always_inline)) {
            constexpr int u = decltype(uTag)::value, kk = decltype(kkTag)::value;
            dupS[kk] = __builtin_amdgcn_raw_buffer_load_b64(rs, (PRIV ? voffB2 : voffB) + kk * G::N * 4, so + u * kRoundOff, 0);
        };
        // PRIV + CLONE: the conversion's store base against its clone, in front of a group's stores (cold path: both from a fresh lane id; one
        // detected / corrected word)
        auto verifyDst = [&]() __attribute__((always_inline)) {
            if constexpr (PRIV && DUP) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(dstB0v != dstB02) != 0, 0)) {
                    const int l = freshLane();
                    const int fresh = wbufFold + colRow(2 * (l & 7)) * G::KS + ((((l >> 3) >> 2) ^ colSwz(2 * (l & 7))) * 16) + ((l >> 3) & 3) * 4;
                    stageMiss += (dstB0v != dstB02) ? 1u : 0u;
                    dstB0v = fresh;
                    dstB02 = launder(fresh);
                }
            }
        };
        auto flagElem = [&](uint32_t mat, int row, int col) __attribute__((always_inline)) {
            if constexpr (FLAGS)
                if (detected != nullptr && mat < nblocks)
                    detected[mat * nn + (size_t)row * G::N + col] = 1;
        };
        // in front of the first instruction that consumes a word of the round; gs = the slab it belongs to.  (The round's second column is
        // converted fifteen slots later: for those slots its words are single again -- a sixth of the time they spend in a register.)
        auto verifyS = [&](auto uTag, int gs) __attribute__((always_inline)) {
            constexpr int u = decltype(uTag)::value;
            bool mis = false;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                mis = mis || pbs[u][kk][0] != dupS[kk][0] || pbs[u][kk][1] != dupS[kk][1];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
                const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 4);
                const int so = slabOff(gs) + u * kRoundOff, l = freshLane();
                const int vo = ((4 * (l >> 3)) * G::N + 2 * (l & 7)) * 4;
                const bool exists = matOf(gs >> 4) < nblocks; // (a slab staged ahead for a matrix behind the batch's last: zeros, and no vote of anybody's)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const u32x2_t c = __builtin_amdgcn_raw_buffer_load_b64(rs, vo + kk * G::N * 4, so, 0);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bool e = pbs[u][kk][h] == dupS[kk][h] || !exists;
                        stageMiss += e ? 0u : 1u;
                        if constexpr (NREP == 3)
                            pbs[u][kk][h] = e ? pbs[u][kk][h] : c[h];
                        if (!e) // (first row of the panel, the word's column: the first element it reaches)
                            flagElem(matOf(gs >> 4), pnl * G::BM, tileCol0(gs) + 2 * (l & 7) + h);
                    }
                }
            }
        };
        auto verifyF = [&](int gc) __attribute__((always_inline)) { // gc = the bg step that consumes bgRaw
            bool mis = false;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                mis = mis || bgRaw[d] != dupF[d];
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mis) != 0, 0)) {
                const u32x4_t c = bgLoad(gc);
                const bool exists = matOf((gc >> 4) + 1) < nblocks;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const bool e = bgRaw[d] == dupF[d] || !exists;
                    stageMiss += e ? 0u : 1u;
                    if constexpr (NREP == 3)
                        bgRaw[d] = e ? bgRaw[d] : c[d];
                    if (!e)
                        flagElem(matOf((gc >> 4) + 1), pnl * G::BM + 8 * bgPiece(gc) + wv, 0);
                }
            }
        };
        auto verifyRoundNow = [&](int gs, auto uTag) __attribute__((always_inline)) { // prologue: a round that is converted at once
            if constexpr (DUP) {
                const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 4);
                const int so = slabOff(gs);
                for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto kkTag) __attribute__((always_inline)) { dupLoadS(uTag, kkTag, rs, so); });
                verifyS(uTag, gs);
            }
        };
        // prologue.  Wave 1 of the pair owns the even slabs: slab 0 whole, slab 2 in its registers.  Wave 0 owns the odd ones: slab
        // 1's first round converted here, its second round in the registers for step 0 (its duty step).
        if constexpr (WIDE) {
            using T2 = std::integral_constant<int, 2>;
            using T3 = std::integral_constant<int, 3>;
            if (H == 1) {
                loadSlabW(0);
                convGroupNowW(U0{}, wbufRest);
                convGroupNowW(U1{}, wbufRest);
                convGroupNowW(T2{}, wbufRest);
                convGroupNowW(T3{}, wbufRest);
                loadSlabW(2);
            } else {
                loadSlabW(1);
                convGroupNowW(U0{}, wbufRest + G::B_BUF);
                convGroupNowW(U1{}, wbufRest + G::B_BUF);
            }
        } else if (H == 1) {
            loadRound(0, U0{});
            loadRound(0, U1{});
            verifyRoundNow(0, U0{});
            convRound(U0{}, wbufRest);
            verifyRoundNow(0, U1{});
            convRound(U1{}, wbufRest);
            loadRound(2, U0{});
            loadRound(2, U1{});
        } else {
            loadRound(1, U0{});
            loadRound(1, U1{});
            verifyRoundNow(1, U0{});
            convRound(U0{}, wbufRest + G::B_BUF);
            if constexpr (DUP) { // step 0 is this wave's duty step: the clones of slab 1's second round
                const __amdgpu_buffer_rsrc_t rs = rsSof(0);
                const int so = slabOff(1);
                for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto kkTag) __attribute__((always_inline)) { dupLoadS(U1{}, kkTag, rs, so); });
            }
        }
        if constexpr (DUP)
            bgRaw = u32x4_t{0u, 0u, 0u, 0u}; // (the first bg step's piece is requested in step 0)
        else
            bgRaw = bgLoad(1); // the first bg step
        __syncthreads(); // panel 0 and the pairs' slab 0 are complete

        // ---- tile end, set by set.  A set's sums are final ten slots after it started in the tile's LAST step; the sets that are through
        // in time leave behind that step's remaining MFMAs, the others behind the first MFMAs of the NEXT step (through the previous tile's
        // buffer resources), each before its accumulators restart.  Stage = (set, element row i): recombine; the last replica's set of a
        // row block also votes and stores.  NREP = 3: sets 0-3 at slot 13 + 3 n of the last step, sets 4, 5 at slot 1 + 6 n' of the next;
        // NREP = 2: sets 0-2 at slot 10 set + 11 + 2 i, set 3 at 1 + 6 i; NREP = 1: set 0 at 11 + 2 i, set 1 at 1 + 2 i.
        constexpr int NNEXT = NREP == 1 ? 1 : NREP - 1, NLAST = NSET - NNEXT;
        uint32_t teV[NREP > 1 ? NREP - 1 : 1][4] = {};
        __amdgpu_buffer_rsrc_t rsRp = rsrcOf(R, false, 0), rsDp = rsRp;
        auto recombine = [&](auto rbTag, auto rrTag, auto iTag) __attribute__((always_inline)) {
            constexpr int rb = decltype(rbTag)::value, rr = decltype(rrTag)::value, i = decltype(iTag)::value;
            uint32_t t; // Horner: three v_lshl_add_u32 (the compiler reassociates the C form into four instructions)
            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(acc[rb][rr][3][i]), "v"(acc[rb][rr][2][i]));
            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(t), "v"(acc[rb][rr][1][i]));
            asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(t), "v"(acc[rb][rr][0][i]));
            return t;
        };
        auto voteStore = [&](int g, int voffR, uint32_t real, auto rbTag, auto iTag, uint32_t vLast) __attribute__((always_inline)) {
            constexpr int rb = decltype(rbTag)::value, i = decltype(iTag)::value;
            uint32_t voted = vLast;
            bool same = true;
            if constexpr (NREP == 3) {
                const uint32_t v0 = teV[0][i], v1 = teV[1][i];
                const bool e01 = v0 == v1, e02 = v0 == vLast;
                voted = e01 ? v0 : vLast; // select(a == b, a, c), synchronization.cpp:934-938
                same = e01 && e02;
            } else if constexpr (NREP == 2) {
                voted = teV[0][i]; // DWC: a compare, replica 0's value is what the original store writes (:1117-1192)
                same = voted == vLast;
            }
            if constexpr (NREP > 1) {
                agree += same ? 1u : 0u;
                nExec += 1u;
                nReal += real; // __SYNC_COUNT is counted where the vote happens
            }
            const int erow = pnl * G::BM + (2 * H + rb) * 16 + i;
            if constexpr (!(COAST_MM3_KNOCK & 4))
                __builtin_amdgcn_raw_buffer_store_b32(voted, rb == 0 ? rsR : rsRp, voffR, (erow * G::N + tileCol0(g)) * 4, COAST_MM_AUX_R);
            if constexpr (FLAGS && NREP > 1)
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)1, rb == 0 ? rsD : rsDp, same ? 0x40000000 : (voffR >> 2),
                                                     erow * G::N + tileCol0(g), 0);
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
        // the stage of slot m of the tile's last step / of the step behind it (or of the final flush), if any
        auto teLast = [&](int g, int voffR, auto mTag) __attribute__((always_inline)) {
            constexpr int m = decltype(mTag)::value;
            if constexpr (NREP == 3) {
                if constexpr (m >= 13 && m % 3 == 1)
                    teStage(g, voffR, 1u, std::integral_constant<int, (m - 13) / 12>{}, std::integral_constant<int, ((m - 13) / 3) % 4>{});
            } else {
                constexpr int set = (m - 11) / 10, off = m - 11 - 10 * set;
                if constexpr (m >= 11 && set < NLAST && off < 8 && off % 2 == 0)
                    teStage(g, voffR, 1u, std::integral_constant<int, set>{}, std::integral_constant<int, off / 2>{});
            }
        };
        auto teNext = [&](int g, int voffR, uint32_t real, auto nTag) __attribute__((always_inline)) { // stage n' = 0 .. 4 NNEXT - 1
            constexpr int n = decltype(nTag)::value;
            teStage(g, voffR, real, std::integral_constant<int, NLAST + n / 4>{}, std::integral_constant<int, n % 4>{});
        };
        constexpr int kNextStride = NREP == 1 ? 2 : 6; // slots between the stages of the next step

        uint32_t fFirst = 0, fCount = 0;
        // COAST_SITE_MM_PREG (PHYS): the upset of this wave in the current item, if any: key = step of the item << 6 | slot; sel = register file
        // << 9 | register number; lane | bit << 8
        uint32_t pregKey = 0xffffffffu, pregSel = 0u, pregLaneBit = 0u;
        // word w of entry q of the upset table (1: .local, 2: .step, 3: .replica | .site << 8 | .bit << 16 | .index << 24).  PHYS: through a
        // descriptor that ends with the panel's entries, no vector register in the address -- an upset of any VGPR cannot send this read anywhere
        auto ftWord = [&](uint32_t q, int w) __attribute__((always_inline)) {
            if constexpr (PHYS == 2) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<DevFault *>(ft.list), 0, (int)((fFirst + fCount) * 16u), 0x00020000);
                return __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rs, 0, (int)(q * 16u) + 4 * w, 0));
            } else {
                const DevFault *fp = ft.list + q;
                return __builtin_amdgcn_readfirstlane(w == 1 ? fp->local : w == 2 ? fp->step : *reinterpret_cast<const uint32_t *>(&fp->replica));
            }
        };
        auto pregScan = [&]() __attribute__((always_inline)) {
            pregKey = 0xffffffffu;
#pragma unroll 1
            for (uint32_t q = fFirst, nq = 0; q < fFirst + fCount && nq < 64u; ++q, ++nq) { // (nq: a flipped count must not walk the table for ever)
                const uint32_t sw = ftWord(q, 2), packed = ftWord(q, 3);
                if (((packed >> 8) & 0xffu) == (uint32_t)SITE_MM_PREG && ((sw >> 16) & 7u) == (uint32_t)wv) {
                    pregKey = sw & 0x3ffu;
                    pregSel = (((sw >> 19) & 1u) << 9) | ((sw >> 20) & 511u); // file << 9 | register (the step word packs file << 19 | register << 20)
                    pregLaneBit = ((sw >> 10) & 63u) | (((packed >> 16) & 31u) << 8);
                }
            }
        };
        auto pregFlip = [&]() __attribute__((always_inline)) {
            const uint32_t idx = pregSel & 511u, bitMask = 1u << (pregLaneBit >> 8);
            if ((pregSel >> 9) == 0u) {
                const uint32_t vm = freshLane() == (int)(pregLaneBit & 63u) ? bitMask : 0u;
                // (the wait states a matrix-core result needs before a VALU may touch its register are the compiler's business everywhere
                // else; m0 carries the index and is put back)
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
        // ---- injector hook: the consequence of an armed upset on the replica's word is an additive constant (everything downstream
        // is linear mod 2^32), written on the replica's limb-0 sums before the tile's last step -- see mm_mfma_kernel.hip, file header.  OPA
        // names replica r's loaded f[i][k]: here that register exists per replica (the A fragment of the replica's set).
        auto tileHook = [&](int g) __attribute__((always_inline)) {
            const int col0 = tileCol0(g), prow0 = pnl * G::BM;
            bool hooked = false;
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) { // (PHYS: a bound of its own -- an upset of the count must not walk the table for minutes)
                const int fcol = (int)(ftWord(q, 1) & 255u);
                hooked = hooked || (fcol >= col0 && fcol < col0 + G::CT);
            }
            if (!hooked)
                return;
            uint32_t curKey = 0xffffffffu, curStep = 0xffffffffu;
            uint32_t dsum[3] = {0u, 0u, 0u}, am[3] = {0u, 0u, 0u}, bm[3] = {0u, 0u, 0u};
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) { // (PHYS: a bound of its own -- an upset of the count must not walk the table for minutes)
                const uint32_t local = ftWord(q, 1);
                const int frow = (int)(local >> 8), fcol = (int)(local & 255u);
                if (fcol < col0 || fcol >= col0 + G::CT)
                    continue;
                const uint32_t fstep = ftWord(q, 2), packed = ftWord(q, 3);
                const uint32_t frep = packed & 0xffu, fsite = (packed >> 8) & 0xffu, m = 1u << ((packed >> 16) & 31u);
                if constexpr (PHYS)
                    if (fsite > (uint32_t)SITE_MM_OPB)
                        continue; // (a physical register upset: applied where the register lives, below)
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
                // the replica's register: panel row -> (rb, lane group, i), column -> lane, replica -> block
                const int r16 = frow & 15;
                const bool mineLane = lane == (r16 >> 2) * 16 + (fcol - col0);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int rr = 0; rr < NREP; ++rr)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[rb][rr][0][i] += (int)((mineLane && frep == (uint32_t)rr && (frow >> 4) == 2 * H + rb && (r16 & 3) == i) ? delta : 0u);
            }
            return;
        };

        // ---- one pipeline step of this wave: the 60 MFMAs of slab `g` as six sets (row block rb = set / 3, replica rr = set % 3) of
        // ten -- A plane p = 0..3 against B planes q = 3 - p .. 0.  a[p] is re-read behind its last use in the set, for the NEXT set
        // (same row block and address for the next replica: the load is the replicated instruction); b[rr][q] is re-read behind its
        // last use in the second row block, from the other slab buffer.
        constexpr bool ABUF = COAST_MM3_ABUF != 0;
        constexpr int NA = ABUF ? 8 : 4;
        constexpr auto aIdx = [](int set, int p) { return ABUF ? 4 * (set & 1) + p : p; }; // register of fragment p of a set
        v4i_t a[NA], b[NREP][4];
        int offA = panelOff(0), offB = bOff;
        // PRIV: replica-private fragment bases (a flip there corrupts one replica's operands: the compare at the tile's end sees it)
        int aOffR[NREP], offBR[NREP];
#pragma unroll
        for (int r = 0; r < NREP; ++r) {
            aOffR[r] = PRIV ? launder(aOff) : aOff;
            offBR[r] = PRIV ? launder(bOff) : bOff;
        }
        auto panelOffR = [&](int g, int r) __attribute__((always_inline)) { return ((g >> 4) & 1) * G::A_PANEL + (aOffR[r] ^ ((g & 3) * 64)); };
        auto loadA = [&](auto pTag, int rbl, int off) __attribute__((always_inline)) { // (one set: register p)
            constexpr int p = decltype(pTag)::value;
            if constexpr (COAST_MM3_KNOCK & 512) // timing: the read is issued and awaited, but always of the same 16 bytes per lane
                a[p] = *reinterpret_cast<const v4i_t *>(smemP + (off & 0) + aOff + p * G::PLANE_A);
            else
                a[p] = *reinterpret_cast<const v4i_t *>(smemP + off + p * G::PLANE_A + (2 * H + rbl) * 16 * G::N);
        };
        auto loadAi = [&](auto idxTag, auto pTag, int rbl, int off) __attribute__((always_inline)) { // plane p of row block rbl into register idx
            constexpr int p = decltype(pTag)::value, idx = decltype(idxTag)::value;
            a[idx] = *reinterpret_cast<const v4i_t *>(smemP + off + p * G::PLANE_A + (2 * H + rbl) * 16 * G::N);
        };
        auto loadB = [&](auto rrTag, auto qTag, int bufOff) __attribute__((always_inline)) {
            constexpr int rr = decltype(rrTag)::value, q = decltype(qTag)::value;
            if constexpr (COAST_MM3_KNOCK & 512)
                b[rr][q] = *reinterpret_cast<const v4i_t *>(smemP + (bufOff & 0) + (FOLD ? 0 : kSlabBase) + offB + q * G::PLANE_B);
            else
                b[rr][q] = *reinterpret_cast<const v4i_t *>(smemP + bufOff + (PRIV ? offBR[rr] : offB) + q * G::PLANE_B);
        };
        for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto pTag) __attribute__((always_inline)) { loadA(pTag, 0, offA); });
        for_each_index(std::make_integer_sequence<int, NREP>{}, [&](auto rrTag) __attribute__((always_inline)) {
            asm volatile("" : "+v"(offB)); // one load per replica: not to be merged
            for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto qTag) __attribute__((always_inline)) { loadB(rrTag, qTag, wbufRest); });
        });
        // ---- COAST_SITE_MM_VGPR (PHYS): coast_fault.item names the panel, the row half (row / 32), the row block (row / 16) and the column
        // tile (column / 16); .replica the replica; .step = k-slab of the tile (bits 1:0) | lane << 8 | dword of the 4-dword fragment << 16 |
        // register << 24 (0-3: A fragment of byte plane p; 4-7: B fragment of plane q; 8-11: limb-sum accumulator t); .bit the bit.
        auto physFields = [&](uint32_t q, int g, uint32_t &frep, uint32_t &reg, uint32_t &dword, uint32_t &frb, uint32_t &mask) __attribute__((always_inline)) {
            const uint32_t local = ftWord(q, 1), sw = ftWord(q, 2), packed = ftWord(q, 3);
            const uint32_t frow = local >> 8, fcol = local & 255u;
            frep = packed & 0xffu;
            reg = (sw >> 24) & 31u;
            dword = (sw >> 16) & 3u;
            frb = (frow >> 4) & 1u;
            mask = lane == (int)((sw >> 8) & 63u) ? 1u << ((packed >> 16) & 31u) : 0u;
            return ((packed >> 8) & 0xffu) == (uint32_t)SITE_MM_VGPR && (frow >> 5) == (uint32_t)H && (sw & 3u) == (uint32_t)(g & 3) &&
                   (int)(fcol / (uint32_t)G::CT) * G::CT == tileCol0(g);
        };
        auto physA = [&](int g, auto setTag) __attribute__((always_inline)) { // in front of the first MFMA of set `set`
            constexpr int set = decltype(setTag)::value;
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) { // (PHYS: a bound of its own -- an upset of the count must not walk the table for minutes)
                uint32_t frep, reg, dword, frb, mask;
                if (!physFields(q, g, frep, reg, dword, frb, mask) || reg > 3u || frb != (uint32_t)(set / NREP) || frep != (uint32_t)(set % NREP))
                    continue;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        a[aIdx(set, pp)][d] ^= (int)((reg == (uint32_t)pp && dword == (uint32_t)d) ? mask : 0u);
            }
        };
        auto physStart = [&](int g, bool accLive) __attribute__((always_inline)) { // at the start of a step: B fragments, accumulators
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + (PHYS == 2 && fCount > 256u ? 256u : fCount); ++q) { // (PHYS: a bound of its own -- an upset of the count must not walk the table for minutes)
                uint32_t frep, reg, dword, frb, mask;
                if (!physFields(q, g, frep, reg, dword, frb, mask) || reg < 4u)
                    continue;
                if (reg >= 12u) { // the staging registers every replica's data passes through: raw words of s (12-15: round 0, 16-19:
                                  // round 1; two dwords each) and of the next panel of f (20; four dwords) on their way into LDS
#pragma unroll
                    for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                if constexpr (WIDE) // (the same sixteen words: register 12 + 4 u + kk, dword h = row kk's word 2 u + h)
                                    pw[kk][2 * u + h] ^= (reg == (uint32_t)(12 + 4 * u + kk) && dword == (uint32_t)h) ? mask : 0u;
                                else
                                    pbs[u][kk][h] ^= (reg == (uint32_t)(12 + 4 * u + kk) && dword == (uint32_t)h) ? mask : 0u;
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        bgRaw[d] ^= (reg == 20u && dword == (uint32_t)d) ? mask : 0u;
                    continue;
                }
                if (reg > 11u || frep >= (uint32_t)NREP)
                    continue;
#pragma unroll
                for (int rr = 0; rr < NREP; ++rr)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const bool sel = frep == (uint32_t)rr && dword == (uint32_t)d;
                            b[rr][qq][d] ^= (int)((sel && reg == (uint32_t)(4 + qq)) ? mask : 0u);
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb)
                                acc[rb][rr][qq][d] ^= (int)((sel && accLive && reg == (uint32_t)(8 + qq) && frb == (uint32_t)rb) ? mask : 0u);
                        }
            }
        };
        auto step = [&](int g, auto firstTag, auto posTag) __attribute__((always_inline)) {
            constexpr int FIRST = decltype(firstTag)::value; // first slab of a tile: the sums start from zero, the previous tile's last stages run
            constexpr int POS = decltype(posTag)::value;     // g % 4
            constexpr bool DUTY = (POS & 1) == H;            // first half: second staging round of slab g + 1; otherwise second half: first round of slab g + 2
            constexpr bool BG = POS == 1 || POS == 2;        // a background f piece, in the half without conversion
            const int gLoad = (DUP && !DUTY) ? g + 2 : g + 3; // duty step: the staged copies of slab g + 3; off duty: the clones of slab g + 2
            const int soffLoad = slabOff(gLoad);
            const __amdgpu_buffer_rsrc_t rsLoad = rsSof(gLoad >> 4);
            // (an item has an even number of steps: the parity of g is that of POS -- the slab buffers' offsets are instruction offsets)
            const int bufNext = FOLD ? ((POS + 1) & 1) * G::B_BUF : wbufOff + ((g + 1) & 1) * G::B_BUF;
            const int bufConv = DUTY ? bufNext : (FOLD ? (POS & 1) * G::B_BUF : wbufOff + (g & 1) * G::B_BUF);
            int offAnext = PRIV ? panelOffR(g + 1, 0) : panelOff(g + 1);
            [[maybe_unused]] int offAR[NREP]; // PRIV: this step's A bases per replica
            if constexpr (PRIV) {
#pragma unroll
                for (int r = 0; r < NREP; ++r)
                    offAR[r] = panelOffR(g, r);
            }
            offA = panelOff(g);
            const int bgDst = BG ? (((g >> 4) + 1) & 1) * G::A_PANEL + panelDst(bgPiece(g)) : 0;
            const uint32_t realPrev = g != 0 ? 1u : 0u;
            int voffR = 0;
            if constexpr (FIRST != 0 || POS == 3)
                voffR = voffRof();
            __builtin_amdgcn_sched_barrier(0);

            uint32_t y[4], t[4];
            uint32_t (&w)[4] = y;
            auto digits4 = [&](uint32_t x0, uint32_t x1, auto halfTag) __attribute__((always_inline)) {
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
            auto convStage = [&](auto kTag) __attribute__((always_inline)) {
                constexpr int k = decltype(kTag)::value, u = k / 10, h = (k / 5) % 2, sub = k % 5;
                if constexpr (sub == 0 && h == 0 && DUP)
                    verifyS(std::integral_constant<int, u>{}, DUTY ? g + 1 : g + 2);
                using GRP = std::integral_constant<int, k / 5>; // (two-word loads: round u, column h of the pair; WIDE: column k / 5 of the quad)
                using K0 = std::integral_constant<int, 0>;
                using K1 = std::integral_constant<int, 1>;
                using K2 = std::integral_constant<int, 2>;
                using K3 = std::integral_constant<int, 3>;
                if constexpr (sub == 0)
                    digits4(rawWord(GRP{}, K0{}), rawWord(GRP{}, K1{}), std::integral_constant<int, 0>{});
                else if constexpr (sub == 1)
                    digits4(rawWord(GRP{}, K2{}), rawWord(GRP{}, K3{}), std::integral_constant<int, 1>{});
                else if constexpr (sub == 2)
                    perm1();
                else if constexpr (sub == 3)
                    perm2();
                else {
                    verifyDst();
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint32_t *>(smemP + bufConv + q * G::PLANE_B + dstGrp(k / 5)) = w[q];
                }
            };
            auto bgStage = [&](auto subTag) __attribute__((always_inline)) {
                constexpr int sub = decltype(subTag)::value;
                if constexpr (sub == 0 && DUP)
                    verifyF(g);
                if constexpr (sub == 0)
                    digits4(bgRaw[0], bgRaw[1], std::integral_constant<int, 0>{});
                else if constexpr (sub == 1) {
                    digits4(bgRaw[2], bgRaw[3], std::integral_constant<int, 1>{});
                    if constexpr (POS == 1 || !DUP)
                        bgRaw = bgLoad(POS == 1 ? g + 1 : g + 3); // the next bg step: this tile's third step, or the next tile's second
                } else if constexpr (sub == 2)
                    perm1();
                else if constexpr (sub == 3)
                    perm2();
                else {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        *reinterpret_cast<uint32_t *>(smemP + bgDst + p * G::PLANE_A) = w[p];
                }
            };
            const v4i_t zero = {0, 0, 0, 0};
            if constexpr (PHYS)
                if (fCount != 0u)
                    physStart(g, FIRST == 0);
            const uint32_t pregSlot = PHYS == 2 ? pregKey - ((uint32_t)(g & 15) << 6) : 0xffffffffu; // the slot of THIS step the upset sits in front of, if any
            auto slot = [&](auto mTag) __attribute__((always_inline)) {
                constexpr int m = decltype(mTag)::value;
                constexpr int set = m / 10, j = m % 10, rb = set / NREP, rr = set % NREP;
                constexpr int p = j < 4 ? 0 : j < 7 ? 1 : j < 9 ? 2 : 3;
                constexpr int jj = j - (p == 0 ? 0 : p == 1 ? 4 : p == 2 ? 7 : 9);
                constexpr int q = 3 - p - jj;
                constexpr bool fromZero = FIRST != 0 && p == 0;
                if constexpr (j == 0 && set != 0)
                    asm volatile("" : "+v"(offA)); // this set's A fragments are its own loads
                if constexpr (PHYS == 2)
                    if (pregSlot == (uint32_t)m)
                        pregFlip();
                if constexpr (PHYS && j == 0)
                    if (fCount != 0u)
                        physA(g, std::integral_constant<int, set>{});
                if constexpr (!(COAST_MM3_KNOCK & 32))
                    acc[rb][rr][p + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[aIdx(set, p)], b[rr][q], fromZero ? zero : acc[rb][rr][p + q], 0, 0, 0);
                if constexpr (ABUF && j < 4) { // the NEXT set's fragment j, into the other buffer: a whole set ahead of its first use
                    if constexpr (set == NSET - 1) {
                        if constexpr (j == 0)
                            asm volatile("" : "+v"(offAnext));
                        loadAi(std::integral_constant<int, aIdx(set + 1, j)>{}, std::integral_constant<int, j>{}, 0, offAnext);
                    } else {
                        if constexpr (j == 0)
                            asm volatile("" : "+v"(offA));
                        loadAi(std::integral_constant<int, aIdx(set + 1, j)>{}, std::integral_constant<int, j>{}, (set + 1) / NREP, PRIV ? offAR[(set + 1) % NREP] : offA);
                    }
                }
                if constexpr (!ABUF && jj == 3 - p && !((COAST_MM3_KNOCK & 1) && set % NREP != NREP - 1) && !(COAST_MM3_KNOCK & (256 | 2048))) { // last use of a[p] in this set: the next set's (the next step's first set behind the last)
                    if constexpr (set == NSET - 1) {
                        if constexpr (p == 0)
                            asm volatile("" : "+v"(offAnext));
                        loadA(std::integral_constant<int, p>{}, 0, offAnext);
                    } else {
                        if constexpr (p == 0)
                            asm volatile("" : "+v"(offA));
                        loadA(std::integral_constant<int, p>{}, (set + 1) / NREP, PRIV ? offAR[(set + 1) % NREP] : offA);
                    }
                }
                if constexpr (rb == 1 && jj == 0 && !(COAST_MM3_KNOCK & (256 | 1024))) { // last use of b[rr][3 - p] in this step
                    if constexpr (p == 0)
                        asm volatile("" : "+v"(offB));
                    loadB(std::integral_constant<int, rr>{}, std::integral_constant<int, 3 - p>{}, bufNext);
                }
                // (slot numbers of the TMR step, NREP = 3: one stage per three slots; DWC / unprotected: per two / one)
                if constexpr (DUTY && m < HALF && m % NREP == 0 && !(COAST_MM3_KNOCK & 16)) // second staging round (stages 10..19) of slab g + 1
                    convStage(std::integral_constant<int, 10 + m / NREP>{});
                if constexpr (!DUTY && m >= HALF && (m - HALF) % NREP == 0 && !(COAST_MM3_KNOCK & 16)) // first staging round (stages 0..9) of slab g + 2
                    convStage(std::integral_constant<int, (m - HALF) / NREP>{});
                if constexpr (WIDE) {
                    // the slab two further on: four four-word loads, behind the last read of the registers they replace (the fourth group's
                    // second digits stage: slot 6 NREP of this duty step)
                    constexpr int W0 = 7 * NREP, WS = 3 * NREP; // (requested right behind the registers' last read instead, slots 19 - 28: the same time)
                    if constexpr (DUTY && m >= W0 && (m - W0) % WS == 0 && (m - W0) / WS < 4 && !(COAST_MM3_KNOCK & 128))
                        pw[(m - W0) / WS] = __builtin_amdgcn_raw_buffer_load_b128(rsLoad, voffW + ((m - W0) / WS) * G::N * 4, soffLoad, 0);
                }
                if constexpr (!WIDE && DUTY && m % (2 * NREP) == 2 * NREP - 1 && m < 8 * NREP && !(COAST_MM3_KNOCK & 128)) // round 0's registers: free since the previous step's second half
                    pbs[0][m / (2 * NREP)] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + (m / (2 * NREP)) * G::N * 4, soffLoad, 0);
                if constexpr (!WIDE && DUTY && m % (2 * NREP) == 2 * NREP - 1 && m >= HALF && m < HALF + 8 * NREP && !(COAST_MM3_KNOCK & 128)) // round 1's: free after stage 16
                    pbs[1][(m - HALF) / (2 * NREP)] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + ((m - HALF) / (2 * NREP)) * G::N * 4, soffLoad + kRoundOff, 0);
                if constexpr (BG && (m / HALF == (DUTY ? 1 : 0)) && (m % HALF) % (2 * NREP) == (NREP == 1 ? 0 : 2) && !(COAST_MM3_KNOCK & 64))
                    bgStage(std::integral_constant<int, (m % HALF) / (2 * NREP)>{});
                // CLONE: the f piece of the tile's second step is requested one step ahead, not three (in the previous tile's third step): the
                // staged word and its clone sit in registers for a step instead of two on average -- and with four registers free for most of
                // the tile the clone registers fit without a reload inside the MFMA blocks.  (Measured for the default too: - 1.1 % for
                // 94.4 -> 94.7 % coverage, not taken: profiles/r05_mm_clone_ab.txt.)
                if constexpr (DUP && POS == 0 && m == (H == 0 ? 1 : HALF + 1) && !(COAST_MM3_KNOCK & 64))
                    bgRaw = bgLoad(g + 1);
                if constexpr (DUP && !DUTY) {
                    // the clones of slab g + 2, off duty: those of its first round in the first slots of the step (compared in the second half), those
                    // of its second round behind that compare (compared in the next step's first slot)
                    constexpr int u = m / HALF, mh = m % HALF;
                    if constexpr (mh >= 1 && mh < 5)
                        dupLoadS(std::integral_constant<int, u>{}, std::integral_constant<int, mh - 1>{}, rsLoad, soffLoad);
                }
                if constexpr (DUP) {
                    // bgRaw's clone: a step whose bg stages run in its second half requests it in slot 1; for a next step whose stages run in its
                    // first half it is requested here, behind this step's own compare
                    if constexpr (BG && DUTY && m == 1)
                        dupF = bgLoadDup(g);
                    if constexpr ((POS == 0 || POS == 1) && (((POS + 1) & 1) != H) && m == HALF + 4)
                        dupF = bgLoadDup(g + 1);
                }
                if constexpr (POS == 3 && !(COAST_MM3_KNOCK & 8))
                    teLast(g, voffR, mTag);
                if constexpr (FIRST != 0 && m % kNextStride == 1 && m / kNextStride < 4 * NNEXT && !(COAST_MM3_KNOCK & 8))
                    teNext(g - 1, voffR, realPrev, std::integral_constant<int, m / kNextStride>{});
                // the workgroup's one barrier per step, behind the last slot of the first half (and behind the conversion stage that slot may
                // carry: NREP = 1 has one in every slot): slab g + 1 is complete, slab g's buffer is free; the B fragments of slab g + 1 are
                // read from the next slot on
                if constexpr (m == HALF - 1 && !(COAST_MM3_KNOCK & 2))
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
            if (item > 0) { // hand-over: the panel is complete behind the barrier of the previous item's last step
                f = F + mat * nn;
                s = S + mat * nn;
                rsR = rsrcOf(R + mat * nn, true, (int)(nn * 4));
                rsD = flagsOf(mat);
            }
            [[maybe_unused]] uint32_t tileGuard = 0u; // (PHYS: the same for the tile counter)
#pragma unroll 1
            for (int tile = 0; tile < G::TPW; ++tile) {
                if constexpr (PHYS == 2)
                    if (tileGuard++ >= (uint32_t)G::TPW)
                        break;
                const int g0 = item * G::SPP + tile * G::NSLAB;
                step(g0, T1{}, T0{}); // + the previous tile's last eight stages (the previous item's resources at a hand-over)
                rsRp = rsR;
                rsDp = rsD;
                step(g0 + 1, T0{}, T1{});
                step(g0 + 2, T0{}, T2{});
                if (fCount != 0u) // armed upsets in this panel (wave-uniform, rare): their deltas go on top of the running limb-0 sums
                    tileHook(g0);
                step(g0 + 3, T0{}, T3{}); // + this tile's first sixteen stages
                gLast = g0 + 3;
                anyTile = 1u;
            }
        }
        { // the last tile's last eight stages: nothing left to hide them behind
            const int voffR = voffRof();
            for_each_index(std::make_integer_sequence<int, 4 * NNEXT>{}, [&](auto nTag) __attribute__((always_inline)) { teNext(gLast, voffR, anyTile, nTag); });
        }

        __syncthreads();
        uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemP + 2 * G::A_PANEL);
        if (tid < 4)
            sCnt[tid] = 0;
        __syncthreads();
        // TMR: TMR_ERROR_CNT = the votes whose copies were not all equal; DWC: every element is an item with one compare -- a failed one is
        // a detected item; __SYNC_COUNT = the votes of tiles that exist (none in the unprotected mode)
        // (+ the staging compares that failed: a corrected word under TMR, a detected one under DWC)
        block_tally(NREP == 3 ? nExec - agree + stageMiss : 0u, nReal, NREP == 2 ? detItems + (nExec - agree) + stageMiss : 0u, sCnt, ctr, blockIdx.x);
    };
    if (wv >= G::NLANE)
        run(std::integral_constant<int, 1>{});
    else
        run(std::integral_constant<int, 0>{});
}

} // namespace coast
