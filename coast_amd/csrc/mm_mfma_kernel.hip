// mm_mfma_kernel.hip -- side-256 protected matrix_multiply on the gfx950 matrix cores.
//
// Same contract as mm_fast256_kernel (mm_kernel.hip): r[i][j] = (uint32) sum_k f[i][k]*s[k][j], replicas of a logical
// output element in NREP ADJACENT LANES, store-data vote across them, replica 0 writes the single copy.  What changes is
// the arithmetic unit.  A 32-bit wrapping product has no MFMA form directly, but it has an exact one in signed bytes:
//     x = sum_p d_p(x) * 256^p  (mod 2^32),  d_p(x) = signed byte p of ((x + 0x80808080) ^ 0x80808080)   in [-128, 127]
// (adding 0x80808080 lets the 32-bit carry chain do the digit carries; the final carry drops out mod 2^32), hence
//     f*s = sum_{p+q<=3} d_p(f) d_q(s) 256^(p+q)  (mod 2^32)
// and r = C0 + (C1<<8) + (C2<<16) + (C3<<24) with C_t = sum_k sum_{p+q=t} d_p(f[i][k]) d_q(s[k][j]) -- ten int8 GEMMs
// accumulated exactly in int32 by v_mfma_i32_32x32x32_i8 (|C_t| <= 4*256*128^2 < 2^31).
//
// Replica geometry.  The MFMA C/D layout puts output column (lane & 31) in lane `lane`; the kernel feeds the SAME logical
// column of s to NREP adjacent lane-columns (B operand lanes 3q, 3q+1, 3q+2 read the same LDS bytes: broadcast, "one load
// feeds all replicas"), so the NREP copies of r[i][j] come out of the matrix core in NREP adjacent lanes, each from its
// own B registers and its own accumulators, and the voter is the same cross-lane exchange as everywhere else.  The A
// operand (rows of f) is shared by all output columns of the instruction -- it is common-mode, like the LDS copy it was
// read from (memory is outside the sphere of replication in this mode).
//
// Injector hooks (mm_patch_faults below).  The reference's fault model is a single-bit flip of a 32-bit register of ONE
// replica at a step of the k loop: the `unsigned long sum` accumulator (mm_common_tmr.c:13) or one of the two loaded
// operands.  The matrix core holds that accumulator as four int32 limb sums of 32-deep slabs, so there is no register to
// flip at "k = 37"; but everything after the flip is linear mod 2^32, so the upset has an exact consequence on the
// replica's recombined word and THAT is what the hook applies, in the replica's own lane, before the voter looks at it:
//     ACC, step k < n :  v += (p ^ m) - p,  p = sum_{k'<k} f[i][k'] s[k'][j]  (+ the replica's earlier deltas)
//     ACC, step n     :  v ^= m                                  (the register itself: the loop is over)
//     OPA / OPB, step k: v += (a ^ ma)(b ^ mb) - (a ^ ma')(b ^ mb')   (masks of this step before / after this fault)
// The prefix p is recomputed by the wave on the VALU from the single memory copy (<= 4 MACs per lane).  From there on the
// kernel is on its own: `bad` sees the disagreeing lanes, the select voter out-votes (or, for a double hit, does not), the
// counters and per-item flags follow -- the VALU engine is not involved (coast_last_launch_info reports general_blocks 0).
//
// Tiling (MmPanel below).  A workgroup owns 64 rows of ONE matrix for ALL its columns: the rows' byte planes
// (64 x 256 x 4 planes = 64 KB) are converted once -- 2 VALU ops per element + a 4x4 byte transpose with v_perm_b32 -- and stay
// in LDS.  After that single barrier every wave is autonomous: it walks its own column tiles (32/NREP logical columns each,
// as two 32x32 MFMA tiles -> 2 x 4 accumulators of 16 registers), converts the s slabs it needs into a wave-private double
// buffer, runs its 20 MFMAs per 32-deep k slab against the shared panel, votes and stores its 64 x 32/NREP tile through the
// same private buffer.  No barrier and no cross-wave dependency inside the main loop.  (A first version re-converted 128 rows
// of f per column block -- 13x for TMR -- and met at a barrier every slab: the matrix core was busy 30 % of the time.)
// Two workgroups of four waves per CU for TMR / DWC (2 x 75.5 / 80 KB of LDS), one of eight for the unprotected mode.
// LDS layouts, both conflict-free for reads and writes without padding (MI355X_MICROARCH.md section LDS: a ds_read_b128 is
// served in four groups of 16 lanes {0-3,12-15,20-27} ..., a ds_write_b32 in two groups of 32):
//   f panel   plane[p][row][256 B]: the row's sixteen 16-byte slots (slot = 2 * slab + half) sit at slot ^ (row & 15).  A
//             fragment read (lane = row, fixed slot) hits 16 different slots in every lane group; a conversion store (one row, 32
//             consecutive k-quads per group) writes 32 consecutive words of a permuted row.
//   s slabs   plane[q][column][32 B], the two halves swapped on columns whose bit 3 is set; a conversion lane owns a column
//             pair and every other pair stores its columns in the opposite order, which spreads a store group over all banks.
//
// Voter.  The common case costs 2 VALU per element: bad |= v ^ shl1(v) (DPP: the next lane's copy), and once per tile
// bad |= shl1(bad) (OR distributes over the lane shift).  For replica 0, bad == 0 <=> all copies of all its elements agree,
// and then the vote IS v.  Only when some lane of the wave saw a difference is the full compare-and-select voter
// (xmr_final_vote_vals, with its counters and per-item flags) run over the tile: same result, same counts.
#include <type_traits>

#include "xmr.hpp"

#ifndef COAST_MM_KNOCK
// development: timing knock-outs, results wrong (1 no s loads, 2 no s conversion, 16 no vote / staging / stores, 32 no f
// panel).  What they showed (profiles/r02_mm_knockouts.txt): the parts ADD UP -- MFMAs + fragment reads 5.68 ms, s staging
// +1.28, epilogue +1.07, f panel +0.24..0.47 = the 8.27 ms of the whole kernel -- i.e. the chip runs at its power limit and
// every instruction is paid in clock, overlapped or not.
#define COAST_MM_KNOCK 0
#endif

namespace coast {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// signed byte digits of x (mod 2^32), still interleaved: byte p of the result is d_p(x)
__device__ __forceinline__ uint32_t mm_digits(uint32_t x) { return (x + 0x80808080u) ^ 0x80808080u; }

// 4x4 byte transpose: in y[i] byte p  ->  out w[p] byte i   (k-contiguous plane words)
__device__ __forceinline__ void mm_transpose4(const uint32_t y[4], uint32_t w[4])
{
    const uint32_t t0 = __builtin_amdgcn_perm(y[1], y[0], 0x05010400u); // y0.b0 y1.b0 y0.b1 y1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(y[1], y[0], 0x07030602u); // y0.b2 y1.b2 y0.b3 y1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(y[3], y[2], 0x05010400u);
    const uint32_t t3 = __builtin_amdgcn_perm(y[3], y[2], 0x07030602u);
    w[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    w[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    w[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    w[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

template <int NREP> struct MmPanel {
    static constexpr int N = 256, KS = 32, NSLAB = N / KS;
    static constexpr int CPW = 32 / NREP;             // logical columns per wave tile (10 / 16 / 32)
    static constexpr int LPW = CPW * NREP;            // lane-columns in use per 32
    static constexpr int NW = NREP == 1 ? 8 : 4;      // waves per workgroup
    static constexpr int WG_PER_CU = NREP == 1 ? 1 : 2;
    static constexpr int NTHR = 64 * NW;
    static constexpr int BM = 64;                     // rows per workgroup
    static constexpr int BPM = N / BM;                // workgroups per matrix
    static constexpr int NCT = (N + CPW - 1) / CPW;   // column tiles per matrix (26 / 16 / 8), tile t -> wave t % NW
    static constexpr int PLANE_A = BM * N;            // bytes per f byte-plane: [row][256], 16-byte slots swizzled by the row
    static constexpr int A_PANEL = 4 * PLANE_A;       // 64 KB
    static constexpr int A_PER_THR = (BM * 8 * NSLAB) / NTHR; // uint4 of the panel per thread (16 / 8)
    static constexpr int PLANE_B = CPW * 32;          // bytes per s byte-plane of one slab of one wave: [column][32]
    static constexpr int B_BUF = 4 * PLANE_B;
    static constexpr int WAVE_LDS = 2 * B_BUF;        // double buffer == the wave's 64 x CPW output tile
    // TMR leaves two of the 32 lane-columns idle: their B operand is read from an all-zero block (one 16-byte slot per plane,
    // at the planes' stride), so the idle columns of the multiplier array see constant zeros instead of a copy of column 0 --
    // the kernel runs at the chip's power limit and every toggling MAC is time
    static constexpr int ZERO_PAD = LPW < 32 ? 3 * PLANE_B + 16 : 0;
    static constexpr size_t LDS_BYTES = (size_t)A_PANEL + NW * WAVE_LDS + ZERO_PAD;
    static constexpr int CPAIR = CPW / 2;             // column pairs per tile: a staging lane owns 2 columns x 4 k
    static constexpr int KQPR = CPAIR > 8 ? 4 : 8;    // k-quads (tile rows x 4) one staging round of a wave covers
    static constexpr int B_ROUNDS = 8 / KQPR;         // staging rounds per slab (CPAIR * KQPR <= 64 lanes each)
    static constexpr int NSETS = 2;                   // register sets of raw s words = prefetch distance in steps (4: no gain)
    static constexpr int VW = (CPW % 4 == 0) ? 4 : 2; // words per output store
    static constexpr int SEGW = CPW / VW;             // stores per tile row
    static_assert(BM * CPW * 4 == WAVE_LDS, "output tile == slab double buffer");
};

// PHYS == 2 (round 6): COAST_SITE_MM_PREG as in the register-block kernels -- a real exclusive-or on ANY physical register of a
// wave in front of any of the 20 MFMA slots of any of the wave's pipeline steps (coast_fault.step: slot | step % 16 << 6 | lane << 10 | wave << 16 |
// file << 19 | register << 20 | step / 16 << 29) -- so that tools/campaign.py --reg-model uniform --kernel lanes puts a coverage figure on the
// replica layout north_star names: three adjacent lanes, every vector register of a work item replica-private by construction.
template <int NREP, int PHYS = 0>
__global__ __launch_bounds__(MmPanel<NREP>::NTHR, MmPanel<NREP>::WG_PER_CU) void mm_mfma_panel_kernel(
    const uint32_t *__restrict__ F, const uint32_t *__restrict__ S, uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
    FaultTab ft, uint8_t *__restrict__ detected)
{
    using G = MmPanel<NREP>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: tile columns and slab offsets stay in SGPRs (no waterfall loops)
    const int lc = lane & 31, kh = lane >> 5;
    uint8_t *const wbuf = smemP + G::A_PANEL + wave * G::WAVE_LDS;

    const uint32_t lb = xcd_logical_block(blockIdx.x, nblocks);
    const uint32_t mat = lb / (uint32_t)G::BPM;
    const int row0 = (int)(lb - mat * (uint32_t)G::BPM) * G::BM;
    constexpr size_t nn = (size_t)G::N * G::N;
    const uint32_t *f = F + mat * nn + (size_t)row0 * G::N;
    const uint32_t *s = S + mat * nn;
    uint32_t *r = R + mat * nn + (size_t)row0 * G::N;

    // ---- the f panel: 64 rows x 64 k-quads, converted to byte planes once (all loads in flight before the first convert)
    if (!(COAST_MM_KNOCK & 32)) {
        uint4 pa[G::A_PER_THR];
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u) {
            const int i = tid + G::NTHR * u; // row = i / 64, k-quad of the row = i % 64
            pa[u] = *reinterpret_cast<const uint4 *>(f + (uint32_t)((i >> 6) * G::N + 4 * (i & 63)));
        }
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u) {
            const int i = tid + G::NTHR * u;
            const int row = i >> 6, kq = i & 63; // 16-byte slot kq >> 2 = 2 * slab + half, word kq & 3 inside it
            const uint32_t y[4] = {mm_digits(pa[u].x), mm_digits(pa[u].y), mm_digits(pa[u].z), mm_digits(pa[u].w)};
            uint32_t w[4];
            mm_transpose4(y, w);
            // a wave stores one row's 64 k-quads: 32 lanes -> 32 consecutive words of (slot ^ row) order = 32 distinct banks
            const int dst = row * G::N + (((kq >> 2) ^ (row & 15)) * 16) + (kq & 3) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint32_t *>(smemP + p * G::PLANE_A + dst) = w[p];
        }
    }

    // Armed faults of this workgroup's 64 x 256 elements: {first, count} in the table the injector sorted by
    // (workgroup, element, step, site).  Wave-uniform; almost always count == 0.
    uint32_t fFirst = 0, fCount = 0;
    if (ft.range) {
        const uint2 rg = ft.range[lb];
        fFirst = __builtin_amdgcn_readfirstlane(rg.x);
        fCount = __builtin_amdgcn_readfirstlane(rg.y);
    }

    // COAST_SITE_MM_PREG (PHYS): this wave's upset, if any: key = step << 6 | slot; sel = register file << 9 | register; lane | bit << 8
    uint32_t pregKey = 0xffffffffu, pregSel = 0u, pregLaneBit = 0u;
    if constexpr (PHYS == 2) {
#pragma unroll 1
        for (uint32_t q = fFirst, nq = 0; q < fFirst + fCount && nq < 64u; ++q, ++nq) {
            const DevFault *fp = ft.list + q;
            const uint32_t sw = __builtin_amdgcn_readfirstlane(fp->step);
            const uint32_t packed = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t *>(&fp->replica));
            if (((packed >> 8) & 0xffu) == 7u /* COAST_SITE_MM_PREG */ && ((sw >> 16) & 7u) == (uint32_t)wave) {
                pregKey = (sw & 63u) | ((((sw >> 6) & 15u) | (((sw >> 29) & 3u) << 4)) << 6);
                pregSel = (((sw >> 19) & 1u) << 9) | ((sw >> 20) & 511u);
                pregLaneBit = ((sw >> 10) & 63u) | (((packed >> 16) & 31u) << 8);
            }
        }
    }
    auto pregFlip = [&]() __attribute__((always_inline)) {
        const uint32_t idx = pregSel & 511u, bitMask = 1u << (pregLaneBit >> 8);
        if ((pregSel >> 9) == 0u) {
            const uint32_t vm = xmr_fresh_lane() == (int)(pregLaneBit & 63u) ? bitMask : 0u;
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

    // ---- this wave's work: column tiles wave, wave + NW, ... ; one pipeline step = one k slab of one tile
    const int tw = wave; // (dealing the tiles in reverse order in odd workgroups to even out the SIMDs was measured: slower --
                         //  two workgroups of one matrix on a CU stop sharing their s lines in L1)
    const int nTiles = (G::NCT - tw + G::NW - 1) / G::NW;
    const int nIt = nTiles * G::NSLAB;
    auto tileCol0 = [&](int it) __attribute__((always_inline)) { return (tw + G::NW * (it >> 3)) * G::CPW; };

    // s: one conversion item = four consecutive k of one column -> one word in each of the four planes; a staging lane owns
    // the two columns of a pair.  Lane -> (pair l % CPAIR, k-quad l / CPAIR + KQPR*round): one dwordx2 load instruction
    // fetches KQPR full tile rows (CPW contiguous words each) -- single-dword loads of the same bytes cost 12 % of the whole
    // kernel (measured: the vector-memory path is what the step waits for).  Buffer loads: per-lane voffset fixed for the whole
    // kernel, the step's slab / tile column in the scalar offset, the four k rows in the immediate; reads past the matrix
    // return 0 (columns past the edge and steps past the end are never consumed).  Lanes beyond CPAIR*KQPR mirror a live lane's
    // item: same data, same destination, no branch.
    // Store banks: a column's plane row is 32 bytes, so columns c and c + 4 share banks and the pairs of a store instruction
    // (columns 0, 2, 4, 6, 8 ...) would pile three lanes on each bank.  Pairs whose index has bit 1 set therefore handle their
    // two columns in the opposite order: one instruction then writes columns 0, 2, 5, 7, 8 -- all four bank classes -- and
    // the only lanes that still share a bank are two-way, which a ds_write_b32 absorbs (MI355X_MICROARCH.md section LDS).
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsS =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(s), 0, (int)(nn * 4), 0x00020000);
    int voffB[G::B_ROUNDS], dstB[G::B_ROUNDS][2];
    bool swapB;
    {
        // pair fastest: consecutive lanes load consecutive 8-byte pieces of one tile row.  (k-quad fastest would make every store
        // group 4 pairs x 8 k-quads = 32 distinct banks, SQ_LDS_BANK_CONFLICT 0 -- and cost 10.6 % of the kernel, 9.14 vs 8.26
        // ms, because a load instruction's neighbouring lanes then sit in eight different rows.  Pair fastest leaves pairs 0 and
        // 4 of a group on one bank class: two-way, absorbed by the store's 4-cycle issue; 18 % of the LDS-active cycles count as
        // conflicts, down from 35 %, at no cost in time.)
        const int l = lane < G::CPAIR * G::KQPR ? lane : lane - G::CPAIR * G::KQPR;
        swapB = (((l % G::CPAIR) >> 1) & 1) != 0;
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u) {
            const int c = 2 * (l % G::CPAIR), kq = u * G::KQPR + l / G::CPAIR;
            voffB[u] = ((4 * kq) * G::N + c) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) { // the column this lane converts h-th
                const int ch = c + (h ^ (swapB ? 1 : 0));
                dstB[u][h] = ch * 32 + (((kq >> 2) ^ ((ch >> 3) & 1)) * 16) + (kq & 3) * 4;
            }
        }
    }
    auto gloadB = [&](int it, u32x2_t (&pb)[G::B_ROUNDS][4]) __attribute__((always_inline)) {
        const int soff = ((it & 7) * G::KS * G::N + tileCol0(it)) * 4;
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pb[u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rsS, voffB[u] + kk * G::N * 4, soff, 0);
    };
    auto rawWord = [&](const u32x2_t &v, int h) __attribute__((always_inline)) { // word of the column handled h-th
        return h == 0 ? (swapB ? v[1] : v[0]) : (swapB ? v[0] : v[1]);
    };
    auto lstoreB = [&](int it, const u32x2_t (&pb)[G::B_ROUNDS][4]) __attribute__((always_inline)) {
        uint8_t *dstBuf = wbuf + (it & 1) * G::B_BUF;
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t y[4] = {mm_digits(rawWord(pb[u][0], h)), mm_digits(rawWord(pb[u][1], h)),
                                       mm_digits(rawWord(pb[u][2], h)), mm_digits(rawWord(pb[u][3], h))};
                uint32_t w[4];
                mm_transpose4(y, w);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(dstBuf + q * G::PLANE_B + dstB[u][h]) = w[q];
            }
        }
    };

    // this lane's operand row / column inside the wave tile
    const int tq = lc < G::LPW ? lc / NREP : 0; // idle lane-columns re-read the tile's first column
    const int aOff = lc * G::N + ((kh ^ (lc & 15)) * 16); // slab's slot: XOR with 32 * slab (bits 5..7 of the same field)
    const int bOff = tq * 32 + ((kh ^ ((tq >> 3) & 1)) * 16);
    const uint8_t *pBpar[2]; // this lane's B fragment address in slab buffer 0 / 1; idle lane-columns: the zero block
#pragma unroll
    for (int par = 0; par < 2; ++par)
        pBpar[par] = (lc < G::LPW) ? wbuf + par * G::B_BUF + bOff : smemP + G::A_PANEL + G::NW * G::WAVE_LDS;
    if (G::ZERO_PAD && tid < G::ZERO_PAD / 4)
        reinterpret_cast<uint32_t *>(smemP + G::A_PANEL + G::NW * G::WAVE_LDS)[tid] = 0u;
    const int rrep = lc % NREP;
    const bool live = lc < G::LPW;

    Tally tl;
    uint32_t detItems = 0;
    v16i_t acc[2][4];
    u32x2_t pbs[G::NSETS][G::B_ROUNDS][4]; // raw s words in flight: set x % NSETS belongs to pipeline step x

#pragma unroll
    for (int j = 0; j < G::NSETS; ++j)
        gloadB(j, pbs[j]);
    lstoreB(0, pbs[0]);
    __syncthreads(); // the panel is complete; from here on the waves do not meet again until the counters

    // recombine the digit products, vote across the replica lanes, single-copy store of the wave's 64 x CPW tile
    auto tileEnd = [&](int it) __attribute__((always_inline)) {
        const int col0 = tileCol0(it), col = col0 + tq;
        const bool writer = live && col < G::N && rrep == 0;
        auto elemRow = [&](int rb, int e) __attribute__((always_inline)) { return rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh; };
        uint32_t bad = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t v = (uint32_t)acc[rb][0][e] + ((uint32_t)acc[rb][1][e] << 8) + ((uint32_t)acc[rb][2][e] << 16) +
                                   ((uint32_t)acc[rb][3][e] << 24);
                acc[rb][0][e] = (int)v;
                if (NREP > 1)
                    bad |= v ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
            }
        }
        if (fCount != 0u) { // rare (wave-uniform): armed upsets somewhere in this workgroup's rows -- any in this tile?
            bool hit = false;
            uint32_t curKey = 0xffffffffu, curStep = 0xffffffffu;
            uint32_t dsum[3] = {0u, 0u, 0u}, am[3] = {0u, 0u, 0u}, bm[3] = {0u, 0u, 0u};
#pragma unroll 1
            for (uint32_t q = fFirst; q < fFirst + fCount; ++q) {
                const DevFault *fp = ft.list + q;
                const uint32_t local = __builtin_amdgcn_readfirstlane(fp->local);
                const int frow = (int)(local >> 8), fcol = (int)(local & 255u);
                if (fcol < col0 || fcol >= col0 + G::CPW)
                    continue;
                const uint32_t fstep = __builtin_amdgcn_readfirstlane(fp->step);
                const uint32_t packed = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t *>(&fp->replica));
                const uint32_t frep = packed & 0xffu, fsite = (packed >> 8) & 0xffu, m = 1u << ((packed >> 16) & 31u);
                if constexpr (PHYS != 0)
                    if (fsite > 2u)
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
                const uint32_t *fr = f + frow * G::N, *sc = s + fcol;
                const uint32_t dprev = frep == 0u ? dsum[0] : frep == 1u ? dsum[1] : dsum[2];
                uint32_t delta = 0u;
                bool flipFinal = false;
                if (fsite == (uint32_t)SITE_MM_ACC) {
                    if (fstep >= (uint32_t)G::N) {
                        flipFinal = true; // the loop is over: the register is the recombined word itself
                    } else {
                        uint32_t part = 0u; // this replica's accumulator before the MAC of k == step
                        for (uint32_t k = (uint32_t)lane; k < fstep; k += 64u)
                            part += fr[k] * sc[k * G::N];
                        const uint32_t pfx = __builtin_amdgcn_readfirstlane(wave_sum(part)) + dprev;
                        delta = (pfx ^ m) - pfx;
                    }
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
                // the replica's lane and register: tile row -> (rb, e, kh) as in elemRow
                const int tl32 = frow & 31;
                const int tIdx = (frow >> 5) * 16 + (tl32 & 3) + 4 * (tl32 >> 3);
                const bool mineLane = lane == ((tl32 >> 2) & 1) * 32 + (fcol - col0) * NREP + (int)frep;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t v = (uint32_t)acc[rb][0][e];
                        const uint32_t nv = flipFinal ? v ^ m : v + delta;
                        acc[rb][0][e] = (int)((mineLane && tIdx == rb * 16 + e) ? nv : v);
                    }
                hit = true;
            }
            if (hit && NREP > 1) { // the voter's quick look, again, on the upset registers
                bad = 0;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        bad |= (uint32_t)acc[rb][0][e] ^ (uint32_t)__builtin_amdgcn_update_dpp(0, acc[rb][0][e], 0x130, 0xf, 0xf, false);
            }
        }
        if (NREP == 3) // OR distributes over the lane shift: one more exchange covers (replica 1 ^ replica 2) of every element
            bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0x130, 0xf, 0xf, false);
        const bool slow = __ballot(writer && bad != 0u) != 0ull; // wave-uniform
        uint32_t *stage = reinterpret_cast<uint32_t *>(wbuf); // both slab buffers are idle: slab 7 was just consumed
        if (!slow) {
            wave_lds_sync();
            if (writer) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stage[elemRow(rb, e) * G::CPW + tq] = (uint32_t)acc[rb][0][e];
                tl.syncs += (NREP > 1) ? 32u : 0u;
            }
            wave_lds_sync();
            // row segments: lane -> (row inside a group of RPG rows, VW-word segment of the row)
            constexpr int RPG = 64 / G::SEGW;
            const int srow = lane / G::SEGW, cs = (lane - srow * G::SEGW) * G::VW;
            const bool son = srow < RPG && col0 + cs < G::N;
            const uint32_t *sp = stage + srow * G::CPW + cs;
            uint32_t *gp = r + (uint32_t)(srow * G::N + col0 + cs);
#pragma unroll
            for (int i = 0; i < (G::BM + RPG - 1) / RPG; ++i) {
                if (son && (G::BM % RPG == 0 || srow + i * RPG < G::BM)) {
                    if (G::VW == 4)
                        *reinterpret_cast<uint4 *>(gp + i * RPG * G::N) = *reinterpret_cast<const uint4 *>(sp + i * RPG * G::CPW);
                    else
                        *reinterpret_cast<uint2 *>(gp + i * RPG * G::N) = *reinterpret_cast<const uint2 *>(sp + i * RPG * G::CPW);
                }
            }
            wave_lds_sync();
        } else {
            // The copies of some element of this tile disagree: an upset was caught.  Rare, so compact rather than fast: eight
            // elements of every lane at a time go through the wave's LDS, the full compare-and-select voter with its counters
            // reads its two neighbours from there, element-wise stores.
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int rb = ch >> 1, eb = (ch & 1) * 8;
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    stage[j * 64 + lane] = (uint32_t)acc[rb][0][eb + j];
                wave_lds_sync();
#pragma unroll 1
                for (int j = 0; j < 8; ++j) {
                    const int e = eb + j, orow = elemRow(rb, e);
                    const uint32_t v = stage[j * 64 + lane], b = stage[j * 64 + ((lane + 1) & 63)],
                                   c = stage[j * 64 + ((lane + 2) & 63)];
                    const bool mine = writer;
                    Tally te = tl;
                    te.det = 0;
                    const uint32_t voted = xmr_final_vote_vals<NREP>(v, b, c, mine, te);
                    tl.miss = te.miss;
                    tl.syncs = te.syncs;
                    if (mine) {
                        r[(uint32_t)(orow * G::N + col)] = voted;
                        if (te.det) {
                            if (NREP == 2)
                                detItems += 1;
                            if (detected)
                                detected[mat * nn + (size_t)(row0 + orow) * G::N + col] = 1;
                        }
                    }
                }
            }
            wave_lds_sync();
        }
    };

    // (Also tried: reading the next step's B and row-block-0 A fragments behind the last MFMAs of this step -- no gain.)
    // One pipeline step, hand-scheduled as ONE basic block: the loads for step it+2 go out, the 12 fragment reads of step
    // it, then its 20 MFMAs with the conversion of step it+1 slotted between them in ten small stages (sched_barrier pins the
    // order) -- the matrix core and the VALU run side by side instead of taking turns; two waves per SIMD drift into the same
    // phase otherwise.  FIRST (k slab 0 of a tile): the first MFMA into each accumulator starts from zero instead of
    // clearing 128 registers.
    auto step = [&](int it, u32x2_t (&pbLoad)[G::B_ROUNDS][4], const u32x2_t (&pbConv)[G::B_ROUNDS][4], auto firstTag)
                    __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(firstTag)::value;
        const int soffLoad = (((it + G::NSETS) & 7) * G::KS * G::N + tileCol0(it + G::NSETS)) * 4;
        const int slab = it & 7;
        const uint8_t *pA = smemP + (aOff ^ (slab * 32));
        const uint8_t *pB = (it & 1) ? pBpar[1] : pBpar[0]; // `it` is wave-uniform: a scalar select
        v4i_t a[2][4], b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
            b[p] = *reinterpret_cast<const v4i_t *>(pB + p * G::PLANE_B);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            a[0][p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            a[1][p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A + 32 * G::N);
        __builtin_amdgcn_sched_barrier(0);

        uint8_t *dstBuf = wbuf + ((it + 1) & 1) * G::B_BUF;
        uint32_t y[4], t[4], w[4];
        // conversion of (round u, the column the lane handles h-th) in five stages
        auto convStage = [&](int k) __attribute__((always_inline)) {
            const int u = k / 10, h = (k / 5) % 2, sub = k % 5;
            if (u >= G::B_ROUNDS)
                return;
            if (sub == 0) {
                y[0] = mm_digits(rawWord(pbConv[u][0], h));
                y[1] = mm_digits(rawWord(pbConv[u][1], h));
            } else if (sub == 1) {
                y[2] = mm_digits(rawWord(pbConv[u][2], h));
                y[3] = mm_digits(rawWord(pbConv[u][3], h));
            } else if (sub == 2) {
                t[0] = __builtin_amdgcn_perm(y[1], y[0], 0x05010400u);
                t[1] = __builtin_amdgcn_perm(y[1], y[0], 0x07030602u);
                t[2] = __builtin_amdgcn_perm(y[3], y[2], 0x05010400u);
                t[3] = __builtin_amdgcn_perm(y[3], y[2], 0x07030602u);
            } else if (sub == 3) {
                w[0] = __builtin_amdgcn_perm(t[2], t[0], 0x05040100u);
                w[1] = __builtin_amdgcn_perm(t[2], t[0], 0x07060302u);
                w[2] = __builtin_amdgcn_perm(t[3], t[1], 0x05040100u);
                w[3] = __builtin_amdgcn_perm(t[3], t[1], 0x07060302u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(dstBuf + q * G::PLANE_B + dstB[u][h]) = w[q];
            }
        };
        constexpr int NSTAGE = 10 * G::B_ROUNDS;
        const v16i_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int m = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int qq = 0; qq + p < 4; ++qq) {
                    // limb sums in the order t = 3 2 1 0 | 3 2 1 | 3 2 | 3: consecutive MFMAs never share an accumulator
                    // (measured: no difference to 0 1 2 3 | 1 2 3 | ..., the partner wave fills the dependency gap anyway)
                    const int q = 3 - p - qq;
                    if constexpr (PHYS == 2)
                        if (pregKey == (((uint32_t)it << 6) | (uint32_t)m))
                            pregFlip();
                    acc[rb][p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rb][p], b[q],
                                                                           (FIRST && p == 0) ? zero : acc[rb][p + q], 0, 0, 0);
                    // behind the MFMAs: the 4 * B_ROUNDS loads for step it + NSETS (spread, so that the address unit never holds
                    // the wave up) and the NSTAGE (10 / 20) conversion stages of step it + 1
                    constexpr int LSTRIDE = 4 * G::B_ROUNDS <= 10 ? 2 : 1;
                    if (!(COAST_MM_KNOCK & 1) && m % LSTRIDE == 0 && m / LSTRIDE < 4 * G::B_ROUNDS) {
                        const int u = (m / LSTRIDE) / 4, kk = (m / LSTRIDE) % 4;
                        pbLoad[u][kk] = __builtin_amdgcn_raw_buffer_load_b64(rsS, voffB[u] + kk * G::N * 4, soffLoad, 0);
                    }
                    if (COAST_MM_KNOCK & 2) {
                    } else if (NSTAGE <= 10) {
                        if (m & 1)
                            convStage(m >> 1);
                    } else {
                        convStage(m);
                    }
                    ++m;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // step x: loads for step x + NSETS go into set x % NSETS (converted during step x - 1), set (x + 1) % NSETS is converted
#pragma unroll 1
    for (int it = 0; it < nIt; it += G::NSETS) {
#pragma unroll
        for (int j = 0; j < G::NSETS; ++j) {
            if (j == 0 && (it & 7) == 0) // NSETS divides 8: only the first sub-step can open a tile ...
                step(it, pbs[0], pbs[1 % G::NSETS], std::true_type{});
            else
                step(it + j, pbs[j], pbs[(j + 1) % G::NSETS], std::false_type{});
            wave_lds_sync();
        }
        if (((it + G::NSETS - 1) & 7) == 7) { // ... and only the last one can close it
            // the tile's epilogue stages through both slab buffers: the next step was converted into one of them a moment
            // ago, so redo that (cheap, the registers still hold it) once the tile is out
            if (COAST_MM_KNOCK & 16) { // keep the MFMAs alive, drop the vote / staging / stores
                int sum = 0;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            sum += acc[rb][t][e];
                if (sum == 0x12345678)
                    r[lane] = (uint32_t)sum;
            } else
                tileEnd(it + G::NSETS - 1);
            lstoreB(it + G::NSETS, pbs[0]);
            wave_lds_sync();
        }
    }

    __syncthreads();
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemP + G::A_PANEL); // the wave buffers are idle now
    if (tid < 4)
        sCnt[tid] = 0;
    __syncthreads();
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
}

} // namespace coast
