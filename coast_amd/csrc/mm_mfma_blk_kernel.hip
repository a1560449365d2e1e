// mm_mfma_blk_kernel.hip -- side-256 protected matrix_multiply on the matrix cores, replicas in REGISTER BLOCKS (TMR).
//
// Same arithmetic as mm_mfma_kernel.hip (signed-byte limb decomposition: ten int8 MFMAs per limb pair set, accumulated exactly
// in int32).  What changes is where the replicas of an output element live.  There: NREP adjacent LANES of one accumulator
// register -- a wave tile is 32 lane-columns = 10 logical columns for TMR, two lane-columns idle, 26 tiles for 256 columns (the
// last one 60 % empty), dealt 7 / 7 / 6 / 6 to the four waves.  Here: the SAME lane of NREP accumulator blocks -- replica r of
// r[i][j] is accumulator block r, fed by its own B-operand registers (read from the single LDS copy once per replica: the load
// is part of the replicated computation, the memory is not) and accumulated by its own MFMAs:
//     * no idle lane-columns and no ragged last tile: 16 tiles of 16 logical columns per 64-row panel, four per wave
//     * the s slab a wave converts (16 columns x 64 k) keeps all 64 conversion lanes busy (40 there) and feeds 120 MFMAs
//     * the voter compares registers of one lane: no cross-lane traffic; the voted tile is stored straight from the registers
// Three replicas x four limb sums of a 64-row tile only fit the register file with v_mfma_i32_16x16x64_i8 (4 accumulator
// registers per 16 x 16 block: 4 row blocks x 3 x 4 x 4 = 192) and one wave per SIMD.  With nobody else on the SIMD the wave has
// to hide every latency by its own instruction order:
//     * every operand register is re-read right after its last use in the step, tens of MFMAs before its next
//     * a workgroup is persistent (one per CU) and takes one 64-row panel position of a sequence of matrices: the next
//       matrix's rows of f are fetched and converted into a second LDS panel buffer in the background of the current panel's
//       steps (one 16-byte piece per thread and step), and the s pipeline runs straight across the matrix boundary, so the
//       load -> convert -> barrier prologue is paid once per workgroup, not once per panel
//     * the vote and the stores of a tile's row blocks 0..2 run behind the MFMAs of the tile's last step (a row block's sums are
//       final 30 MFMAs before the next one's); only row block 3 is left for after the step
// tools/mfma_probe2.hip (stepZ) measured the step shape before the kernel was written: 2.7-2.9 POP/s executed, all of it
// useful, against 2.12 executed / 1.97 useful for the lane-replica kernel.
//
// Injector hooks: everything downstream of an upset is linear mod 2^32, so its exact consequence on the replica's recombined
// word is an additive constant (mm_mfma_kernel.hip, file header); r = C0 + (C1 << 8) + ..., so the constant is placed in the
// replica's limb-0 accumulator BEFORE the tile's first MFMA (the only point where control flow does not fight 192 live
// accumulators for registers), the matrix core adds the products on top, and the voter sees the upset word.
#include <type_traits>
#include <utility>

#include "xmr.hpp"

// cache policy of the streams touched once (r stores, f panel loads): the aux field of the buffer instructions, bit 1 = nt -- see
// mm_mfma_blk2_kernel.hip and profiles/r03_mm_nt.txt
#ifndef COAST_MM_AUX_R
#define COAST_MM_AUX_R 2
#endif
#ifndef COAST_MM_AUX_F
#define COAST_MM_AUX_F 2
#endif


#ifndef COAST_BLK_KNOCK
#define COAST_BLK_KNOCK 0 // development timing knock-outs (results wrong): 1 no tile end, 2 no background panel, 4 no s conversion
#endif
namespace coast {

// f(integral_constant<int, 0>), f(<1>), ...: a loop whose index is a constant in every iteration's own instantiation (the step's
// 120 slots are too big for `#pragma unroll` to take -- its cost estimate runs before the slot tests fold away)
template <class Fn, int... Is> __device__ __forceinline__ void for_each_index(std::integer_sequence<int, Is...>, Fn &&fn)
{
    (fn(std::integral_constant<int, Is>{}), ...);
}

template <int NREP> struct MmBlk {
    static constexpr int N = 256, KS = 64, NSLAB = N / KS; // 64-deep k slabs: four steps per tile
    static constexpr int CT = 16;                 // logical columns per wave tile
    static constexpr int NW = 4, NTHR = 64 * NW;  // one wave per SIMD
    static constexpr int BM = 64, NPANEL = N / BM, NRB = BM / 16;
    static constexpr int NCT = N / CT;            // 16 column tiles per panel, tile t -> wave t % NW
    static constexpr int TPW = NCT / NW;          // tiles per wave and panel
    static constexpr int SPP = TPW * NSLAB;       // steps per panel (16)
    static constexpr int PLANE_A = BM * N, A_PANEL = 4 * PLANE_A; // 64 KB, two of them
    static constexpr int PLANE_B = CT * KS, B_BUF = 4 * PLANE_B;  // plane[q][column][64 B]: 4 KB per slab
    static constexpr int WAVE_LDS = 2 * B_BUF;    // double buffer
    static constexpr size_t LDS_BYTES = (size_t)2 * A_PANEL + NW * WAVE_LDS; // 160 KB: the whole CU
    static constexpr int A_PER_THR = (BM * (N / 4)) / NTHR; // uint4 (four k of one row) of a panel per thread = SPP
    static constexpr int B_ROUNDS = 2;            // staging: lane -> (column pair l % 8, k-quad l / 8 + 8 * round)
    static_assert(A_PER_THR == SPP, "one background piece of the next panel per step");
};

template <int NREP, bool FLAGS>
__global__ __launch_bounds__(MmBlk<NREP>::NTHR, 1) void mm_mfma_blk_kernel(const uint32_t *__restrict__ F,
                                                                           const uint32_t *__restrict__ S,
                                                                           uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
                                                                           FaultTab ft, uint8_t *__restrict__ detected)
{
    using G = MmBlk<NREP>;
    static_assert(NREP == 3, "the step's load / conversion slots are laid out for 120 MFMAs");
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kg = lane >> 4; // operand row / column inside a 16-block, 16-byte k group of the slab
    uint8_t *const wbuf = smemP + 2 * G::A_PANEL + wave * G::WAVE_LDS;

    // A workgroup is persistent and owns ONE 64-row panel position: panel `pnl` of matrices mat0, mat0 + stride, mat0 + 2 stride, ...
    // ("items"), with the load / convert pipelines running straight across the item boundaries.  The four panels of a matrix
    // are taken by four workgroups of the same XCD at about the same time (workgroups go round-robin over the 8 XCDs), so the
    // matrix's s -- which every one of them streams in full -- comes out of that XCD's L2 three times out of four.
    constexpr size_t nn = (size_t)G::N * G::N;
    const uint32_t stride = gridDim.x / G::NPANEL; // matrices in flight
    const bool xcdMap = (gridDim.x % 32u) == 0u;
    const uint32_t slotX = blockIdx.x >> 3; // position inside the XCD
    const uint32_t mat0 = xcdMap ? (blockIdx.x & 7u) * (gridDim.x >> 5) + (slotX >> 2) : blockIdx.x >> 2;
    const int pnl = (int)(xcdMap ? slotX & 3u : blockIdx.x & 3u);
    auto matOf = [&](int item) __attribute__((always_inline)) { return mat0 + (uint32_t)item * stride; };
    auto rsrcOf = [&](const void *base, bool live, int bytes) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(live ? base : (const void *)F), 0, live ? bytes : 0, 0x00020000);
    };
    // the item's f / s as buffers; past the batch an empty one: reads give 0 (the pipelines run two steps ahead of the last item)
    auto rsFof = [&](int item) __attribute__((always_inline)) { return rsrcOf(F + matOf(item) * nn, matOf(item) < nblocks, (int)(nn * 4)); };
    auto rsSof = [&](int item) __attribute__((always_inline)) { return rsrcOf(S + matOf(item) * nn, matOf(item) < nblocks, (int)(nn * 4)); };
    const uint32_t *f = F + mat0 * nn, *s = S + mat0 * nn; // the current item's, for the injector hook
    const __amdgpu_buffer_rsrc_t rsF = rsFof(0), rsS = rsSof(0);
    __amdgpu_buffer_rsrc_t rsR = rsrcOf(R + mat0 * nn, true, (int)(nn * 4));
    const int voffR = ((4 * ((tid & 63) >> 4)) * G::N + (tid & 15)) * 4; // C/D layout: lane -> column lane % 16, rows 4 (lane / 16) + i
    // per-item flags (FLAGS): one byte per element; the range check drops the stores of the lanes whose element agreed (their
    // offset is pushed out of range): no branch
    auto flagsOf = [&](uint32_t m) __attribute__((always_inline)) { // per-item flags of matrix m; none: an empty buffer
        const bool on = FLAGS && detected != nullptr;
        return rsrcOf(on ? detected + m * nn : (const uint8_t *)F, on, (int)nn);
    };
    __amdgpu_buffer_rsrc_t rsD = flagsOf(mat0);

    // ---- f panels: plane[p][row][256 B], the row's sixteen 16-byte slots (slot = k / 16) at slot ^ (row & 15) -- the layout of
    // mm_mfma_panel_kernel; a 16x16x64 fragment read (lane = row & 15, k group) finds its 16 lanes' slots in 16 different banks.
    // Piece j of a panel for thread t: row 4 j + t / 64, k-quad t % 64.  Panel 0 is converted here, panels 1..3 in the background.
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int voffF = ((tid >> 6) * G::N + 4 * (tid & 63)) * 4;
    // piece j: row 4 j + w (w = wave < 4), so row & 15 = 4 (j & 3) | w and the swizzle splits into a per-thread and a per-piece part
    const int panelDst0 = (tid >> 6) * G::N + ((((tid & 63) >> 2) ^ (tid >> 6)) * 16) + (tid & 3) * 4;
    auto panelDst = [&](int j) __attribute__((always_inline)) { return (panelDst0 ^ ((j & 3) * 64)) + j * 4 * G::N; };
    {
        u32x4_t pa[G::A_PER_THR];
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u)
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rsF, voffF, (pnl * G::BM + u * 4) * G::N * 4, COAST_MM_AUX_F);
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
    // background pieces in flight: piece g % 16 of the NEXT item's panel is loaded during step g - 2 and converted during step g
    // (g = step number in the workgroup's life: item g / 16, tile (g / 4) % 4, slab g % 4)
    auto bgLoad = [&](int g) __attribute__((always_inline)) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsFof((g >> 4) + 1), voffF, (pnl * G::BM + 4 * (g & 15)) * G::N * 4, COAST_MM_AUX_F);
    };
    u32x4_t bgRaw[2] = {bgLoad(0), bgLoad(1)}; // two pieces in flight: piece g lives in set g & 1, reloaded with piece g + 2

    // ---- this wave's work: per item the column tiles wave, wave + NW, ...; one pipeline step = one 64-deep k slab of one tile
    auto tileCol0 = [&](int g) __attribute__((always_inline)) { return (wave + G::NW * ((g >> 2) & 3)) * G::CT; };

    // s staging: one conversion item = four consecutive k of one column -> one word in each of the four planes; a lane owns the two
    // columns of a pair, lane -> (pair l % 8, k-quad l / 8 + 8 * round): a dwordx2 load instruction fetches eight full tile rows
    // (64 B each).  A plane holds one 64-byte row per column, column c in row (c % 8) * 2 + c / 8, its four 16-byte slots (slot =
    // k / 16) at slot ^ ((c / 2) % 4): the fragment reads (ds_read_b128: lane = column, k group; 64 banks, lane groups of 16) and
    // the conversion stores (ds_write_b32: the eight columns of one parity x four k-quads per 32 lanes; 32 banks) are both
    // conflict-free -- checked by enumeration, SQ_LDS_BANK_CONFLICT agrees.
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    auto colRow = [](int c) { return ((c & 7) << 1) | (c >> 3); }; // column c's 64-byte row in a plane
    auto colSwz = [](int c) { return (c >> 1) & 3; };               // XORed onto its 16-byte slot number
    int voffB, dstB[G::B_ROUNDS][2]; // staging round u loads 32 u rows further down: that goes into the scalar offset
    constexpr int kRoundOff = 8 * 4 * G::N * 4;
#pragma unroll
    for (int u = 0; u < G::B_ROUNDS; ++u) {
        const int c = 2 * (lane & 7), kq = u * 8 + (lane >> 3);
        if (u == 0)
            voffB = ((4 * kq) * G::N + c) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = c + h;
            dstB[u][h] = colRow(ch) * G::KS + (((kq >> 2) ^ colSwz(ch)) * 16) + (kq & 3) * 4;
        }
    }
    auto slabOff = [&](int g) __attribute__((always_inline)) { return ((g & 3) * G::KS * G::N + tileCol0(g)) * 4; };
    auto rawWord = [&](const u32x2_t &v, int h) __attribute__((always_inline)) { return v[h]; };

    // fragment addresses: A row block rb, plane p, slab sl: (aOff ^ (sl * 64)) + rb * 16 * N + p * PLANE_A   (slot 4 sl + kg, swizzled)
    const int aOff = l16 * G::N + ((kg ^ l16) * 16);
    const int bOff = colRow(l16) * G::KS + ((kg ^ colSwz(l16)) * 16);
    auto panelA = [&](int g) __attribute__((always_inline)) { return smemP + ((g >> 4) & 1) * G::A_PANEL + (aOff ^ ((g & 3) * 64)); };

    Tally tl;
    uint32_t detItems = 0;
    v4i_t acc[G::NRB][NREP][4];
    // raw s words in flight, two slabs deep: slab G lives in set G & 1; a step converts slab g + 1 and reloads each staging round's
    // registers with slab g + 3 as soon as the round is converted -- almost two steps for the data to arrive (one step is not enough:
    // every step then opened with a wait for memory, 15 % of the kernel).  The set index is a compile-time constant: the slab
    // position inside a tile fixes the parity.
    u32x2_t pbs[2][G::B_ROUNDS][4];

    // prologue: slab 0 converted into buffer 0, slabs 1 and 2 in flight
    {
        const int so = slabOff(0);
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pbs[0][u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rsS, voffB + kk * G::N * 4, so + u * kRoundOff, 0);
        const int so1 = slabOff(1);
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pbs[1][u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rsS, voffB + kk * G::N * 4, so1 + u * kRoundOff, 0);
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t y[4] = {mm_digits(rawWord(pbs[0][u][0], h)), mm_digits(rawWord(pbs[0][u][1], h)),
                                       mm_digits(rawWord(pbs[0][u][2], h)), mm_digits(rawWord(pbs[0][u][3], h))};
                uint32_t w[4];
                mm_transpose4(y, w);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(wbuf + q * G::PLANE_B + dstB[u][h]) = w[q];
            }
        const int so2 = slabOff(2);
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pbs[0][u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rsS, voffB + kk * G::N * 4, so2 + u * kRoundOff, 0);
    }
    __syncthreads(); // panel 0 (and this wave's slab 0) is complete

    // ---- tile end, BRANCH-FREE while accumulators are alive (control flow makes the register allocator spill accumulator tuples
    // inside the main loop): the compare-and-select voter runs on every element, the out-voted elements are collected in a lane
    // mask, the per-item flags are written from that mask once the accumulators are dead.  One element = five small stages so
    // that they can sit behind MFMAs: recombine replica 0 / 1 / 2, vote, store.  C/D layout of 16x16 blocks: lane -> column
    // lane % 16, rows 4 (lane / 16) + i.
    uint32_t teV[3], teVoted = 0u, teMiss = 0u;
    auto teStage = [&](int g, auto rbTag, auto kTag) __attribute__((always_inline)) {
        constexpr int rb = decltype(rbTag)::value, k = decltype(kTag)::value;
        constexpr int i = k / 5, sub = k % 5;
        if constexpr (sub < 3) {
            constexpr int rs = sub < NREP ? sub : NREP - 1;
            teV[sub] = ((((((uint32_t)acc[rb][rs][3][i] << 8) + (uint32_t)acc[rb][rs][2][i]) << 8) + (uint32_t)acc[rb][rs][1][i]) << 8) +
                       (uint32_t)acc[rb][rs][0][i]; // three v_lshl_add_u32
        } else if constexpr (sub == 3) {
            const bool e01 = teV[0] == teV[1], e02 = teV[0] == teV[2];
            teVoted = (NREP == 3 && !e01) ? teV[2] : teV[0]; // select(a == b, a, c); DWC keeps replica 0's
            teMiss = (e01 && e02) ? 0u : 1u;
            if (NREP == 3)
                tl.miss += teMiss; // TMR_ERROR_CNT
            else
                detItems += teMiss;
        } else {
            // one per-lane offset for the whole kernel; the element's row / the tile's column are a scalar offset
            const int erow = pnl * G::BM + rb * 16 + i;
            __builtin_amdgcn_raw_buffer_store_b32(teVoted, rsR, voffR, (erow * G::N + tileCol0(g)) * 4, COAST_MM_AUX_R);
            if constexpr (FLAGS)
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)1, rsD, teMiss ? (voffR >> 2) : 0x40000000, erow * G::N + tileCol0(g), 0);
        }
    };
    // row block 3's sums are final with the step's last MFMA: its tile end runs after the step.  (Moving it behind the first MFMAs of
    // the next tile's first step was measured: slower -- a slot only has room for about five VALU instructions.)
    auto flushRb3 = [&](int g) __attribute__((always_inline)) {
        for_each_index(std::make_integer_sequence<int, 20>{},
                       [&](auto kTag) __attribute__((always_inline)) { teStage(g, std::integral_constant<int, G::NRB - 1>{}, kTag); });
    };

    // ---- injector hook, at the START of a tile (the accumulators of the previous tile are dead: room for control flow):
    //     ACC, step k <= n : (p ^ m) - p,  p = sum_{k' < k} f[i][k'] s[k'][j] + the replica's earlier deltas   (k = n: p is the
    //                        finished sum -- the flip of the register after the loop)
    //     OPA / OPB, step k: (a ^ ma')(b ^ mb') - (a ^ ma)(b ^ mb)       (masks of this MAC before / after this upset)
    // summed into the replica's limb-0 accumulator.  Returns whether this tile has any armed upset (wave-uniform).
    uint32_t fFirst = 0, fCount = 0; // armed upsets of the current panel: {first, count} in the injector's table
    auto tileHook = [&](int g) __attribute__((always_inline)) {
        const int col0 = tileCol0(g), prow0 = pnl * G::BM;
        bool hooked = false;
#pragma unroll 1
        for (uint32_t q = fFirst; q < fFirst + fCount; ++q) {
            const int fcol = (int)(__builtin_amdgcn_readfirstlane(ft.list[q].local) & 255u);
            hooked = hooked || (fcol >= col0 && fcol < col0 + G::CT);
        }
        if (!hooked)
            return false;
        const v4i_t zero = {0, 0, 0, 0};
#pragma unroll
        for (int rb = 0; rb < G::NRB; ++rb)
#pragma unroll
            for (int rr = 0; rr < NREP; ++rr)
                acc[rb][rr][0] = zero;
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
            for (int rb = 0; rb < G::NRB; ++rb)
#pragma unroll
                for (int rr = 0; rr < NREP; ++rr)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[rb][rr][0][i] += (int)((mineLane && frep == (uint32_t)rr && (frow >> 4) == rb && (r16 & 3) == i) ? delta : 0u);
        }
        return true;
    };

    // ---- one pipeline step = the 120 MFMAs of slab `g` (buffer g & 1), hand-scheduled as one basic block.  Order: row block, A
    // plane p, B plane q (3 - p down to 0), replica -- three consecutive MFMAs share both operands' planes and never an accumulator.
    // Operand registers are refreshed in place, each right behind its last use:
    //   a[p]      plane p of the current row block (12 / 9 / 6 / 3 MFMAs); behind its block it is re-read with plane p of the next row
    //             block -- of this slab, or of row block 0 of the next slab (not across a panel hand-over: after the barrier)
    //   b[rr][q]  used in every row block; its last use in the step is in row block 3, A plane 3 - q: re-read there from the other slab
    //             buffer (filled by slot 76); the next step needs b[.][3] first and b[.][0] last, the order in which they come free
    // Behind the MFMAs: the 20 conversion stages of slab g + 1 (every fourth slot up to 76), the loads of slab g + 3 as soon as a
    // staging round's registers are free (after stage 9 / stage 19), the background piece of the next f panel (five stages from
    // slot 82), and in a tile's last step the tile end of row blocks 0 .. 2.
    v4i_t a[4], b[NREP][4];
    int bOffR[NREP]; // opaque copies of the B fragment offset: each replica block issues its own operand loads
#pragma unroll
    for (int rr = 0; rr < NREP; ++rr) {
        bOffR[rr] = bOff;
        asm volatile("" : "+v"(bOffR[rr]));
    }
    auto loadA = [&](auto pTag, int rb, const uint8_t *pA) __attribute__((always_inline)) {
        constexpr int p = decltype(pTag)::value;
        a[p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A + rb * 16 * G::N);
    };
    auto loadB = [&](auto rrTag, auto qTag, const uint8_t *buf) __attribute__((always_inline)) {
        constexpr int rr = decltype(rrTag)::value, q = decltype(qTag)::value;
        b[rr][q] = *reinterpret_cast<const v4i_t *>(buf + bOffR[rr] + q * G::PLANE_B);
    };
    auto loadAllA = [&](const uint8_t *pA) __attribute__((always_inline)) { // row block 0 of a panel's first slab
        for_each_index(std::make_integer_sequence<int, 4>{}, [&](auto pTag) __attribute__((always_inline)) { loadA(pTag, 0, pA); });
    };
    loadAllA(panelA(0));
    for_each_index(std::make_integer_sequence<int, NREP * 4>{}, [&](auto kTag) __attribute__((always_inline)) {
        constexpr int k = decltype(kTag)::value;
        loadB(std::integral_constant<int, k / 4>{}, std::integral_constant<int, k % 4>{}, wbuf);
    });
    auto step = [&](int g, auto firstTag, auto lastTag, auto parTag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parTag)::value, CS = PAR ^ 1; // g & 1; the set that holds slab g + 1 (converted now, reloaded with slab g + 3)
        constexpr int FIRST = decltype(firstTag)::value; // 0 running slab; 1 first slab of a tile; 2 first slab, limb-0 sums hold the hook's deltas
        constexpr int LAST = decltype(lastTag)::value;   // 1: last slab of a tile
        const int soffLoad = slabOff(g + 3);
        const __amdgpu_buffer_rsrc_t rsLoad = rsSof((g + 3) >> 4); // the last three steps of an item load the next item's s
        const uint8_t *pA = panelA(g);
        const uint8_t *pAnext = panelA(g + 1);
        uint8_t *bufNext = wbuf + ((g + 1) & 1) * G::B_BUF;
        uint8_t *bgDst = smemP + (((g >> 4) + 1) & 1) * G::A_PANEL + panelDst(g & 15);
        __builtin_amdgcn_sched_barrier(0);

        uint32_t y[4], t[4];
        uint32_t (&w)[4] = y; // the transposed words replace the digit words
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
        auto convStage = [&](auto kTag) __attribute__((always_inline)) { // s slab g + 1: (round u, the column handled h-th) in five stages
            constexpr int k = decltype(kTag)::value, u = k / 10, h = (k / 5) % 2, sub = k % 5;
            if constexpr (sub == 0)
                digits4(rawWord(pbs[CS][u][0], h), rawWord(pbs[CS][u][1], h), std::integral_constant<int, 0>{});
            else if constexpr (sub == 1)
                digits4(rawWord(pbs[CS][u][2], h), rawWord(pbs[CS][u][3], h), std::integral_constant<int, 1>{});
            else if constexpr (sub == 2)
                perm1();
            else if constexpr (sub == 3)
                perm2();
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(bufNext + q * G::PLANE_B + dstB[u][h]) = w[q];
            }
        };
        auto bgStage = [&](auto subTag) __attribute__((always_inline)) { // the piece of the next panel loaded during the previous step
            constexpr int sub = decltype(subTag)::value;
            if constexpr (sub == 0)
                digits4(bgRaw[PAR][0], bgRaw[PAR][1], std::integral_constant<int, 0>{});
            else if constexpr (sub == 1) {
                digits4(bgRaw[PAR][2], bgRaw[PAR][3], std::integral_constant<int, 1>{});
                bgRaw[PAR] = bgLoad(g + 2); // the registers are free: the piece two steps ahead
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
        // slot m of the step: row block m / 30; inside it A plane p = 0 (12 slots), 1 (9), 2 (6), 3 (3); inside that B plane q from
        // 3 - p down to 0, three replicas each
        auto slot = [&](auto mTag) __attribute__((always_inline)) {
            constexpr int m = decltype(mTag)::value;
            constexpr int rb = m / 30, j = m % 30;
            constexpr int p = j < 12 ? 0 : j < 21 ? 1 : j < 27 ? 2 : 3;
            constexpr int jj = j - (p == 0 ? 0 : p == 1 ? 12 : p == 2 ? 21 : 27);
            constexpr int q = 3 - p - jj / 3, rr = jj % 3;
            constexpr bool fromZero = FIRST != 0 && p == 0 && !(FIRST == 2 && q == 0);
            acc[rb][rr][p + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p], b[rr][q], fromZero ? zero : acc[rb][rr][p + q], 0, 0, 0);
            if constexpr (jj == 3 * (4 - p) - 1) // the plane's block is through: the next row block's (across a panel hand-over the
                                                 // read is early and is repeated behind the barrier)
                loadA(std::integral_constant<int, p>{}, (rb + 1) % G::NRB, rb == G::NRB - 1 ? pAnext : pA);
            if constexpr (rb == G::NRB - 1 && jj < 3) { // last use of b[rr][3 - p] in this step
                if constexpr (m == 90)
                    wave_lds_sync(); // the other buffer is complete (stage 19 stored at slot 76)
                loadB(std::integral_constant<int, rr>{}, std::integral_constant<int, 3 - p>{}, bufNext);
            }
            if constexpr (!(COAST_BLK_KNOCK & 4) && (m & 3) == 0 && m < 80)
                convStage(std::integral_constant<int, m / 4>{});
            if constexpr (!(COAST_BLK_KNOCK & 2) && (m & 7) == 2 && m >= 82)
                bgStage(std::integral_constant<int, (m - 82) / 8>{});
            if constexpr ((m & 3) == 1 && m >= 41 && m < 57) // staging round 0's registers are free after stage 9 (slot 36)
                pbs[CS][0][(m - 41) / 4] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + ((m - 41) / 4) * G::N * 4, soffLoad, 0);
            if constexpr ((m & 3) == 1 && m >= 81 && m < 97) // staging round 1: free after stage 19 (slot 76)
                pbs[CS][1][(m - 81) / 4] = __builtin_amdgcn_raw_buffer_load_b64(rsLoad, voffB + ((m - 81) / 4) * G::N * 4, soffLoad + kRoundOff, 0);
            // tile end of the row block that finished at slot 30 (rb + 1) - 1: its 20 stages in two of every three slots (a slot has
            // room for about five VALU instructions behind its MFMA before the next MFMA is held up)
            if constexpr (!(COAST_BLK_KNOCK & 1) && LAST != 0 && m >= 31 && (m - 31) % 30 < 29 && ((m - 31) % 30) % 3 != 2)
                teStage(g, std::integral_constant<int, (m - 31) / 30>{},
                        std::integral_constant<int, (m - 31) % 30 - ((m - 31) % 30) / 3>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_index(std::make_integer_sequence<int, 120>{}, slot);
    };

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int item = 0; matOf(item) < nblocks; ++item) {
        const uint32_t mat = matOf(item);
        if (ft.range) {
            const uint2 rg = ft.range[mat * (uint32_t)G::NPANEL + (uint32_t)pnl];
            fFirst = __builtin_amdgcn_readfirstlane(rg.x);
            fCount = __builtin_amdgcn_readfirstlane(rg.y);
        }
        if (item > 0) { // hand-over: every wave has stored its pieces of this item's panel (the last one during the previous step)
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
            tl.syncs += 16u;
            if (fCount != 0u && tileHook(g0)) // armed upsets in this tile (wave-uniform, rare)
                step(g0, T2{}, T0{}, T0{});
            else
                step(g0, T1{}, T0{}, T0{});
            step(g0 + 1, T0{}, T0{}, T1{});
            step(g0 + 2, T0{}, T0{}, T0{});
            step(g0 + 3, T0{}, T1{}, T1{});
            if (!(COAST_BLK_KNOCK & 1))
                flushRb3(g0 + 3);
            else { // keep every MFMA alive: one element of every accumulator, one store per tile
                uint32_t x = 0;
                for_each_index(std::make_integer_sequence<int, G::NRB * NREP * 4>{}, [&](auto kTag) __attribute__((always_inline)) {
                    constexpr int k = decltype(kTag)::value;
                    x ^= (uint32_t)acc[k / (NREP * 4)][(k / 4) % NREP][k % 4][0];
                });
                __builtin_amdgcn_raw_buffer_store_b32(x, rsR, voffR, (pnl * G::BM * G::N + tileCol0(g0)) * 4, COAST_MM_AUX_R);
            }
        }
    }

    __syncthreads();
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(wbuf - wave * G::WAVE_LDS);
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

} // namespace coast
