// mm_mfma_blk2_kernel.hip -- the register-block TMR matrix_multiply kernel with TWO waves per SIMD (its one-wave-per-SIMD predecessor,
// mm_mfma_blk_kernel -- the replicas of an output element in three accumulator blocks of one lane, 192 accumulator registers in AGPRs --
// was retired in round 5: docs/design/mm.md section 4.1a has its design, git history its source).
//
// Replicas in REGISTER BLOCKS: replica r of r[i][j] is accumulator block r of ONE lane, fed by its own B-operand registers (read from the
// single LDS copy once per replica: the load is part of the replicated computation, the memory is not) and accumulated by its own MFMAs -- no
// idle lane-columns and no ragged last tile (16 tiles of 16 logical columns per 64-row panel), the converted s slab (16 columns x 64 k) keeps
// all 64 conversion lanes busy, the voter compares registers of one lane and the voted tile is stored straight from the registers.  A
// workgroup is persistent (one per CU) and takes one 64-row panel position of a sequence of matrices: the next matrix's rows of f are fetched
// and converted into a second LDS panel buffer in the background of the current panel's steps, and the s pipeline runs straight across the
// matrix boundary.  Injector hooks: everything downstream of an upset is linear mod 2^32, so its exact consequence on the replica's
// recombined word is an additive constant (mm_mfma_kernel.hip, file header), placed on the replica's limb-0 accumulator.
//
// With one wave per SIMD every non-MFMA instruction of the step costs matrix-core issue time (~6.6 cycles each, ~200 of them
// per 120 MFMAs: profiles/microbench_r02.txt).  Two waves per SIMD hide one wave's VALU / LDS / VMEM instructions behind the other
// wave's MFMAs.  The register file allows it once a wave carries half the accumulators: the 64 x 16 tile of a column-tile lane is
// split between a PAIR of waves (w, 0) and (w, 1) -- same SIMD, row blocks {0, 1} and {2, 3} -- 96 accumulator registers each,
// ~230 registers per wave.  The pair shares one s slab double buffer in LDS (both need the same slab), and takes turns
// converting: wave H converts slab g + 1 during the steps with g % 2 == H (and the background f piece in the others), so each
// slab is still converted once per pair and the conversion work per wave halves.  One workgroup barrier per step, in its middle:
// the converter is through by slot 28, the fragments of the next slab are read behind it.
//
// Tile end inside the steps: a wave's first row block is final after slot 29 of the tile's last step and is voted and stored behind
// that step's remaining 30 MFMAs; its second row block leaves behind the first 30 MFMAs of the NEXT tile's first step (across an
// item hand-over through the previous item's buffer resources), before slot 30 restarts its sums.  Only the very last tile's second
// row block is flushed in the open.  The first step of all therefore votes a tile that does not exist: zeroed accumulators (equal
// replicas, nothing counted) and a zero-sized buffer resource (stores dropped).
//
// Register budget: 256 VGPRs per wave at two waves per SIMD, no AGPRs to spill into.  Any per-lane constant that stayed live across
// the steps and got spilled came back through scratch -- and its s_waitcnt vmcnt(0) drained the two-slab-deep s prefetch (a variant
// with ~40 spilled registers ran 8-13 ms instead of 6.6).  The rarely used ones (f piece offsets, panel destinations, store offsets)
// are recomputed from a fresh lane id at their point of use instead: 0 VGPR spills in the step bodies.
#include <type_traits>
#include <utility>

#include "xmr.hpp"

// Cache policy of the streams that are touched once (profiles/r03_mm_nt.txt): the aux field of the buffer instructions, bit 1 = nt
// (non-temporal).  r is written once and f is read once per panel workgroup; s is the operand the four panel workgroups of a matrix
// share through their XCD's L2 -- 4 MB for the eight matrices an XCD has in flight, which f + s + r (6 MB) did not fit: about half of s
// was filled twice (15.3 GB of HBM traffic per launch against 12.9 algorithmic).  With r and f non-temporal: 13.6 GB = 1.05 x.
#ifndef COAST_MM_AUX_R
#define COAST_MM_AUX_R 2
#endif
#ifndef COAST_MM_AUX_F
#define COAST_MM_AUX_F 2
#endif

namespace coast {

// f(integral_constant<int, 0>), f(<1>), ...: a loop whose index is a constant in every iteration's own instantiation (the step's
// slots are too big for `#pragma unroll` to take -- its cost estimate runs before the slot tests fold away)
template <class Fn, int... Is> __device__ __forceinline__ void for_each_index(std::integer_sequence<int, Is...>, Fn &&fn)
{
    (fn(std::integral_constant<int, Is>{}), ...);
}

template <int NREP> struct MmBlk2 {
    static constexpr int N = 256, KS = 64, NSLAB = N / KS;
    static constexpr int CT = 16;
    static constexpr int NLANE = 4, NW = 2 * NLANE, NTHR = 64 * NW; // four column-tile lanes x two row halves: two waves per SIMD
    static constexpr int BM = 64, NPANEL = N / BM;
    static constexpr int NCT = N / CT, TPW = NCT / NLANE, SPP = TPW * NSLAB; // 16 steps per panel
    static constexpr int PLANE_A = BM * N, A_PANEL = 4 * PLANE_A;
    static constexpr int PLANE_B = CT * KS, B_BUF = 4 * PLANE_B;
    static constexpr int PAIR_LDS = 2 * B_BUF;
    static constexpr size_t LDS_BYTES = (size_t)2 * A_PANEL + NLANE * PAIR_LDS; // 160 KB
    static constexpr int A_PER_THR = (BM * (N / 4)) / NTHR; // 8 background pieces per thread and panel: one every other step
    static constexpr int B_ROUNDS = 2;
    static_assert(2 * A_PER_THR == SPP, "one background piece of the next panel per thread every other step");
};

template <int NREP, bool FLAGS>
__global__ __launch_bounds__(MmBlk2<NREP>::NTHR, 1) void mm_mfma_blk2_kernel(const uint32_t *__restrict__ F,
                                                                            const uint32_t *__restrict__ S,
                                                                            uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
                                                                            FaultTab ft, uint8_t *__restrict__ detected)
{
    using G = MmBlk2<NREP>;
    static_assert(NREP == 3, "the step's slots are laid out for 60 MFMAs per wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wv & 3; // column-tile lane; waves wv and wv + 4 land on the same SIMD
    const int l16 = lane & 15, kg = lane >> 4;
    uint8_t *const wbuf = smemP + 2 * G::A_PANEL + wave * G::PAIR_LDS; // the pair's slab double buffer

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
    const __amdgpu_buffer_rsrc_t rsF = rsFof(0), rsS = rsSof(0);
    __amdgpu_buffer_rsrc_t rsR = rsrcOf(R + mat0 * nn, true, (int)(nn * 4));
    // per-lane constants that are needed once or twice per step are recomputed from a fresh lane id where they are used: every
    // register that stays live across the steps is one the allocator has to take from the accumulators' neighbourhood (a spilled
    // constant comes back through scratch, and its s_waitcnt vmcnt(0) drains the s prefetches)
    auto freshLane = []() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto voffRof = [&]() __attribute__((always_inline)) {
        const int l = freshLane();
        return ((4 * (l >> 4)) * G::N + (l & 15)) * 4;
    };
    auto flagsOf = [&](uint32_t m) __attribute__((always_inline)) {
        const bool on = FLAGS && detected != nullptr;
        return rsrcOf(on ? detected + m * nn : (const uint8_t *)F, on, (int)nn);
    };
    __amdgpu_buffer_rsrc_t rsD = flagsOf(mat0);

    // ---- f panels: piece j of a panel for thread t (512 threads): row 8 j + t / 64, k-quad t % 64
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int soffFw = wv * G::N * 4; // the wave's row of a piece goes into the scalar offset
    auto voffFof = [&]() __attribute__((always_inline)) { return freshLane() * 16; };
    auto panelDst = [&](int j) __attribute__((always_inline)) { // row & 15 = 8 (j & 1) | t / 64
        const int l = freshLane();
        const int d0 = wv * G::N + (((l >> 2) ^ wv) * 16) + (l & 3) * 4;
        return (d0 ^ ((j & 1) * 128)) + j * 8 * G::N;
    };
    {
        u32x4_t pa[G::A_PER_THR];
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u)
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, voffFof(), soffFw + (pnl * G::BM + u * 8) * G::N * 4, COAST_MM_AUX_F);
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
    // background piece of step g (a step in which this wave is not converting s): piece (g % 16) / 2 of the next item's panel
    auto bgLoad = [&](int g) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsFof((g >> 4) + 1), voffFof(), soffFw + (pnl * G::BM + 8 * ((g & 15) >> 1)) * G::N * 4, COAST_MM_AUX_F);
    };

    auto tileCol0 = [&](int g) __attribute__((always_inline)) { return (wave + G::NLANE * ((g >> 2) & 3)) * G::CT; };

    // s staging: one conversion item = four consecutive k of one column -> one word in each of the four planes; a lane owns the two
    // columns of a pair, lane -> (pair l % 8, k-quad l / 8 + 8 * round): a dwordx2 load instruction fetches eight full tile rows
    // (64 B each).  A plane holds one 64-byte row per column, column c in row (c % 8) * 2 + c / 8, its four 16-byte slots (slot =
    // k / 16) at slot ^ ((c / 2) % 4): the fragment reads (ds_read_b128: lane = column, k group; 64 banks, lane groups of 16) and
    // the conversion stores (ds_write_b32: the eight columns of one parity x four k-quads per 32 lanes; 32 banks) are both
    // conflict-free -- checked by enumeration (tests/test_lds_layouts_cpu.py), SQ_LDS_BANK_CONFLICT agrees.
    // Fragment addresses: A row block rb, plane p, slab sl: (aOff ^ (sl * 64)) + rb * 16 * N + p * PLANE_A (slot 4 sl + kg, swizzled).
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    auto colRow = [](int c) { return ((c & 7) << 1) | (c >> 3); };
    auto colSwz = [](int c) { return (c >> 1) & 3; };
    // staging round u of a lane: columns 2 (lane & 7) + {0, 1}, k-quad 8 u + lane / 8.  The two rounds' and the two columns'
    // destinations differ by constants (round: bit 1 of the 16-byte chunk; column: two slab rows) -- one register.
    const int voffB = ((4 * (lane >> 3)) * G::N + 2 * (lane & 7)) * 4;
    const int dstB0 = colRow(2 * (lane & 7)) * G::KS + ((((lane >> 3) >> 2) ^ colSwz(2 * (lane & 7))) * 16) + ((lane >> 3) & 3) * 4;
    auto dstB = [&](int u, int h) __attribute__((always_inline)) { return (dstB0 ^ (u * 32)) + h * 2 * G::KS; };
    constexpr int kRoundOff = 8 * 4 * G::N * 4;
    auto slabOff = [&](int g) __attribute__((always_inline)) { return ((g & 3) * G::KS * G::N + tileCol0(g)) * 4; };

    const int aOff = l16 * G::N + ((kg ^ l16) * 16);
    const int bOff = colRow(l16) * G::KS + ((kg ^ colSwz(l16)) * 16);
    auto panelA = [&](int g) __attribute__((always_inline)) { return smemP + ((g >> 4) & 1) * G::A_PANEL + (aOff ^ ((g & 3) * 64)); };

    // Everything below is instantiated twice, for the wave's row half H (a compile-time constant: which steps convert s, which
    // row blocks the accumulators stand for); a wave takes its branch once.
    auto run = [&](auto hTag) __attribute__((always_inline)) {
        constexpr int H = decltype(hTag)::value;
        Tally tl; // votes of a tile that does not exist (the first step's look-back at zeroed accumulators; in a workgroup without an
                  // item, the final flush) are executed but not counted: `real` is a wave-uniform predicate (no compensation constant)
        uint32_t detItems = 0;
        v4i_t acc[2][NREP][4]; // row blocks 2 H and 2 H + 1
#pragma unroll
        for (int rbz = 0; rbz < 2; ++rbz)
#pragma unroll
            for (int rz = 0; rz < NREP; ++rz)
#pragma unroll
                for (int pz = 0; pz < 4; ++pz)
                    acc[rbz][rz][pz] = v4i_t{0, 0, 0, 0}; // the first step votes a tile that does not exist: equal replicas, stores out of range
        // raw s words of this wave's next conversion: wave H converts slab g + 1 in the steps with g % 2 == H, i.e. the slabs of
        // one parity; behind each staging round it reloads the registers with the slab two further on -- two steps to arrive
        u32x2_t pbs[G::B_ROUNDS][4];
        u32x4_t bgRaw;

        auto loadSlab = [&](int gs) __attribute__((always_inline)) {
            const int so = slabOff(gs);
            const __amdgpu_buffer_rsrc_t rs = rsSof(gs >> 4);
#pragma unroll
            for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    pbs[u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rs, voffB + kk * G::N * 4, so + u * kRoundOff, 0);
        };
        // prologue: wave 1 of the pair converts slab 0 (the even slabs are its), wave 0 has slab 1 in flight for step 0
        if (H == 1) {
            loadSlab(0);
#pragma unroll
            for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t y[4] = {mm_digits(pbs[u][0][h]), mm_digits(pbs[u][1][h]), mm_digits(pbs[u][2][h]), mm_digits(pbs[u][3][h])};
                    uint32_t w[4];
                    mm_transpose4(y, w);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint32_t *>(wbuf + q * G::PLANE_B + dstB(u, h)) = w[q];
                }
            loadSlab(2);
        } else {
            loadSlab(1);
        }
        bgRaw = bgLoad(H == 1 ? 0 : 1); // this wave's first background step
        __syncthreads(); // panel 0 and the pairs' slab 0 are complete

        // ---- tile end (branch-free): this wave's two row blocks
        uint32_t teV[3], teVoted = 0u, teMiss = 0u;
        __amdgpu_buffer_rsrc_t rsRp = rsrcOf(R, false, 0), rsDp = rsRp; // where the previous tile's second row block goes (nowhere before the first tile)
        auto teStage = [&](int g, int voffR, uint32_t real, auto rbTag, auto kTag) __attribute__((always_inline)) {
            constexpr int rb = decltype(rbTag)::value, k = decltype(kTag)::value;
            constexpr int i = k / 5, sub = k % 5;
            if constexpr (sub < 3) {
                constexpr int rs = sub < NREP ? sub : NREP - 1;
                teV[sub] = ((((((uint32_t)acc[rb][rs][3][i] << 8) + (uint32_t)acc[rb][rs][2][i]) << 8) + (uint32_t)acc[rb][rs][1][i]) << 8) +
                           (uint32_t)acc[rb][rs][0][i];
            } else if constexpr (sub == 3) {
                const bool e01 = teV[0] == teV[1], e02 = teV[0] == teV[2];
                teVoted = (NREP == 3 && !e01) ? teV[2] : teV[0];
                teMiss = (e01 && e02) ? 0u : 1u;
                tl.syncs += real; // __SYNC_COUNT is counted where the vote of a tile that exists happens
                if (NREP == 3)
                    tl.miss += teMiss;
                else
                    detItems += teMiss;
            } else {
                const int erow = pnl * G::BM + (2 * H + rb) * 16 + i;
                __builtin_amdgcn_raw_buffer_store_b32(teVoted, rb == 0 ? rsR : rsRp, voffR, (erow * G::N + tileCol0(g)) * 4, COAST_MM_AUX_R);
                if constexpr (FLAGS)
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)1, rb == 0 ? rsD : rsDp, teMiss ? (voffR >> 2) : 0x40000000, erow * G::N + tileCol0(g), 0);
            }
        };
        auto flushTile = [&](int g, uint32_t real) __attribute__((always_inline)) {
            const int voffR = voffRof();
            for_each_index(std::make_integer_sequence<int, 20>{}, [&](auto kTag) __attribute__((always_inline)) {
                teStage(g, voffR, real, std::integral_constant<int, 1>{}, kTag); // the first row block's went out inside the last step
            });
        };

        uint32_t fFirst = 0, fCount = 0;
        auto tileHook = [&](int g) __attribute__((always_inline)) {
            const int col0 = tileCol0(g), prow0 = pnl * G::BM;
            bool hooked = false;
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + fCount; ++q) {
                const int fcol = (int)(__builtin_amdgcn_readfirstlane(ft.list[q].local) & 255u);
                hooked = hooked || (fcol >= col0 && fcol < col0 + G::CT);
            }
            if (!hooked)
                return;
            uint32_t curKey = 0xffffffffu, curStep = 0xffffffffu;
            uint32_t dsum[3] = {0u, 0u, 0u}, am[3] = {0u, 0u, 0u}, bm[3] = {0u, 0u, 0u};
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + fCount; ++q) {
                const DevFault *fp = ft.list + q;
                const uint32_t local = __builtin_amdgcn_readfirstlane(fp->local);
                const int frow = (int)(local >> 8), fcol = (int)(local & 255u);
                if (fcol < col0 || fcol >= col0 + G::CT)
                    continue;
                const uint32_t fstep = __builtin_amdgcn_readfirstlane(fp->step);
                const uint32_t packed = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t *>(&fp->replica));
                const uint32_t frep = packed & 0xffu, fsite = (packed >> 8) & 0xffu, m = 1u << ((packed >> 16) & 31u);
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


        // ---- one pipeline step of this wave = the 60 MFMAs of slab `g` on its two row blocks (buffer g & 1 of the pair).  Order and
        // operand refresh (a[p] behind its block; b[rr][q] in the second row block, A plane 3 - q, from the
        // other slab buffer).  After slot 29 every wave of the workgroup meets at a barrier: the converters are through (stage 19
        // stores at slot 28), the fragments of slab g + 1 may be read.
        v4i_t a[4], b[NREP][4];
        int bOffR[NREP];
#pragma unroll
        for (int rr = 0; rr < NREP; ++rr) {
            bOffR[rr] = bOff;
            asm volatile("" : "+v"(bOffR[rr]));
        }
        auto loadA = [&](auto pTag, int rbl, const uint8_t *pA) __attribute__((always_inline)) {
            constexpr int p = decltype(pTag)::value;
            a[p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A + (2 * H + rbl) * 16 * G::N);
        };
        auto loadB = [&](auto rrTag, auto qTag, const uint8_t *buf) __attribute__((always_inline)) {
            constexpr int rr = decltype(rrTag)::value, q = decltype(qTag)::value;
            b[rr][q] = *reinterpret_cast<const v4i_t *>(buf + bOffR[rr] + q * G::PLANE_B);
        };
        auto loadAllA = [&](const uint8_t *pA) __attribute__((always_inline)) {
            for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto pTag) __attribute__((always_inline)) { loadA(pTag, 0, pA); });
        };
        loadAllA(panelA(0));
        for_each_index(std::make_integer_sequence<int, NREP * 4>{}, [&](auto kTag) __attribute__((always_inline)) {
            constexpr int k = decltype(kTag)::value;
            loadB(std::integral_constant<int, k / 4>{}, std::integral_constant<int, k % 4>{}, wbuf);
        });
        auto step = [&](int g, auto firstTag, auto posTag) __attribute__((always_inline)) {
            constexpr int FIRST = decltype(firstTag)::value; // 1: first slab of a tile: the previous tile's second row block goes out, then the sums start from zero
            constexpr int POS = decltype(posTag)::value;     // g % 4
            constexpr bool DUTY = (POS & 1) == H;            // this wave converts slab g + 1 now; otherwise its background f piece
            const int soffLoad = slabOff(g + 3);
            const __amdgpu_buffer_rsrc_t rsLoad = rsSof((g + 3) >> 4);
            const uint8_t *pA = panelA(g);
            const uint8_t *pAnext = panelA(g + 1);
            uint8_t *bufNext = wbuf + ((g + 1) & 1) * G::B_BUF;
            uint8_t *bgDst = smemP + (((g >> 4) + 1) & 1) * G::A_PANEL + panelDst((g & 15) >> 1);
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
                if constexpr (sub == 0)
                    digits4(pbs[u][0][h], pbs[u][1][h], std::integral_constant<int, 0>{});
                else if constexpr (sub == 1)
                    digits4(pbs[u][2][h], pbs[u][3][h], std::integral_constant<int, 1>{});
                else if constexpr (sub == 2)
                    perm1();
                else if constexpr (sub == 3)
                    perm2();
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint32_t *>(bufNext + q * G::PLANE_B + dstB(u, h)) = w[q];
                }
            };
            auto bgStage = [&](auto subTag) __attribute__((always_inline)) {
                constexpr int sub = decltype(subTag)::value;
                if constexpr (sub == 0)
                    digits4(bgRaw[0], bgRaw[1], std::integral_constant<int, 0>{});
                else if constexpr (sub == 1) {
                    digits4(bgRaw[2], bgRaw[3], std::integral_constant<int, 1>{});
                    bgRaw = bgLoad(g + 2); // this wave's next background step
                } else if constexpr (sub == 2)
                    perm1();
                else if constexpr (sub == 3)
                    perm2();
                else {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        *reinterpret_cast<uint32_t *>(bgDst + p * G::PLANE_A) = w[p];
                }
            };
            const v4i_t zero = {0, 0, 0, 0};
            auto slot = [&](auto mTag) __attribute__((always_inline)) {
                constexpr int m = decltype(mTag)::value;
                constexpr int rb = m / 30, j = m % 30;
                constexpr int p = j < 12 ? 0 : j < 21 ? 1 : j < 27 ? 2 : 3;
                constexpr int jj = j - (p == 0 ? 0 : p == 1 ? 12 : p == 2 ? 21 : 27);
                constexpr int q = 3 - p - jj / 3, rr = jj % 3;
                constexpr bool fromZero = FIRST != 0 && p == 0;
                acc[rb][rr][p + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p], b[rr][q], fromZero ? zero : acc[rb][rr][p + q], 0, 0, 0);
                if constexpr (jj == 3 * (4 - p) - 1) // the plane's block is through: this wave's other row block, or the next slab's first
                    loadA(std::integral_constant<int, p>{}, (rb + 1) % 2, rb == 1 ? pAnext : pA);
                if constexpr (m == 29)
                    __syncthreads(); // the workgroup's one barrier per step: slab g + 1 is complete in the pairs' other buffers
                if constexpr (rb == 1 && jj < 3) // last use of b[rr][3 - p] in this step
                    loadB(std::integral_constant<int, rr>{}, std::integral_constant<int, 3 - p>{}, bufNext);
                if constexpr (DUTY && m < 29 && m % 3 != 2) // the 20 conversion stages in slots 0 .. 28
                    convStage(std::integral_constant<int, m - m / 3>{});
                if constexpr (DUTY && (m & 1) == 1 && m >= 15 && m < 23) // staging round 0's registers are free after stage 9 (slot 13)
                    pbs[0][(m - 15) / 2] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + ((m - 15) / 2) * G::N * 4, soffLoad, 0);
                if constexpr (DUTY && (m & 1) == 1 && m >= 31 && m < 39) // staging round 1: free after stage 19 (slot 28)
                    pbs[1][(m - 31) / 2] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + ((m - 31) / 2) * G::N * 4, soffLoad + kRoundOff, 0);
                if constexpr (!DUTY && (m & 3) == 0 && m < 20)
                    bgStage(std::integral_constant<int, m / 4>{});
                if constexpr (POS == 3 && m >= 30 && (m - 30) % 3 != 2) // last slab: the first row block's sums are final after slot 29
                    teStage(g, voffR, 1u, std::integral_constant<int, 0>{}, std::integral_constant<int, (m - 30) - (m - 30) / 3>{});
                if constexpr (FIRST != 0 && m < 30 && m % 3 != 1) // the previous tile's second row block, before slot 30 restarts its sums
                    teStage(g - 1, voffR, g != 0 ? 1u : 0u, std::integral_constant<int, 1>{}, std::integral_constant<int, m - (m + 2) / 3>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            for_each_index(std::make_integer_sequence<int, 60>{}, slot);
        };

        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        using T2 = std::integral_constant<int, 2>;
        using T3 = std::integral_constant<int, 3>;
        int gLast = 3;
        uint32_t anyTile = 0u;
#pragma unroll 1
        for (int item = 0; matOf(item) < nblocks; ++item) {
            const uint32_t mat = matOf(item);
            if (ft.range) {
                const uint2 rg = ft.range[mat * (uint32_t)G::NPANEL + (uint32_t)pnl];
                fFirst = __builtin_amdgcn_readfirstlane(rg.x);
                fCount = __builtin_amdgcn_readfirstlane(rg.y);
            }
            if (item > 0) { // hand-over: every wave has stored its pieces of this item's panel
                __syncthreads();
                loadAllA(panelA(item * G::SPP));
                f = F + mat * nn;
                s = S + mat * nn;
                rsR = rsrcOf(R + mat * nn, true, (int)(nn * 4));
                rsD = flagsOf(mat);
            }
#pragma unroll 1
            for (int tile = 0; tile < G::TPW; ++tile) {
                const int g0 = item * G::SPP + tile * G::NSLAB;
                step(g0, T1{}, T0{}); // + the previous tile's second row block (the previous item's resources at a hand-over)
                rsRp = rsR;
                rsDp = rsD;
                step(g0 + 1, T0{}, T1{});
                step(g0 + 2, T0{}, T2{});
                if (fCount != 0u) // armed upsets in this panel (wave-uniform, rare): their deltas go on top of the running limb-0 sums
                    tileHook(g0);
                step(g0 + 3, T0{}, T3{}); // + this tile's first row block
                gLast = g0 + 3;
                anyTile = 1u;
            }
        }
        flushTile(gLast, anyTile); // the last tile's second row block: nothing left to hide it behind

        __syncthreads();
        uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemP + 2 * G::A_PANEL);
        if (tid < 4)
            sCnt[tid] = 0;
        __syncthreads();
        block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
    };
    if (wv >= G::NLANE)
        run(std::integral_constant<int, 1>{});
    else
        run(std::integral_constant<int, 0>{});
}

} // namespace coast
