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
// read from (memory is outside the sphere of replication in this mode).  Armed faults keep their exact semantics: the
// VALU mm_general_kernel recomputes every workgroup-sized tile range that owns a fault, and this kernel neither stores nor
// counts those elements.
//
// Tiling.  256-thread workgroup = 2 (rows) x 2 (columns) waves, two workgroups per CU so that one's staging stalls are
// covered by the other's MFMAs; a wave owns 64 rows x (32/NREP) logical columns as two
// 32x32 MFMA tiles -> 2 x 4 accumulators of 16 registers.  Per 32-deep k slab the workgroup converts its 128 rows of f and
// its columns of s to byte planes (2 VALU ops per element + a 4x4 byte transpose with v_perm_b32) while staging them into
// LDS (k-contiguous per row / per column, 48-byte stride: conflict-free ds_read_b128), double buffered with the next slab's
// global loads in flight; each wave then issues 20 MFMAs per slab.
#include <type_traits>

#include "xmr.hpp"

namespace coast {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

template <int NREP> struct MmMfma {
    static constexpr int N = 256, KS = 32, NSLAB = N / KS;
    static constexpr int CPW = 32 / NREP;          // logical columns per wave (10 / 16 / 32)
    static constexpr int LPW = CPW * NREP;         // lane-columns in use per 32 (30 / 32 / 32)
    static constexpr int WM = 2, WN = 2;           // waves per workgroup along rows / columns
    static constexpr int NTHR = 64 * WM * WN;      // threads per workgroup
    static constexpr int A_SLOTS = (128 * 8) / NTHR; // uint4 of the f panel per thread per slab
    static constexpr int BM = WM * 64;             // 128 rows per workgroup
    static constexpr int BNC = WN * CPW;           // logical columns per workgroup (40 / 64 / 128)
    static constexpr int NBN = (N + BNC - 1) / BNC; // column blocks per matrix (7 / 4 / 2)
    static constexpr int BPM = (N / BM) * NBN;     // workgroups per matrix (14 / 8 / 4)
    static constexpr int RSTR = 48;                // bytes per LDS row: 32 of data + 16 pad
    static constexpr int PLANE_A = BM * RSTR;      // bytes per f byte-plane
    static constexpr int PLANE_B = BNC * RSTR;     // bytes per s byte-plane
    static constexpr int BUF = 4 * (PLANE_A + PLANE_B);
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF + 16; // >= BM*OSTR*4 (checked below)
    static constexpr int OSTR = ((BNC + 7) / 16) * 16 + 8; // words per row of the output staging tile: 4 rows apart = 32 banks apart
    static constexpr int B_ITEMS = (BNC / 4) * (KS / 4); // (4 columns x 4 k) work items of the s panel per slab
    // geometry of the VALU kernels, for the faulted-workgroup test (mm_kernel.hip)
    static constexpr int V_TPB = 4 * (kWave / NREP);
    static constexpr int V_BPM = (64 * 64 + V_TPB - 1) / V_TPB;
};

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

static_assert(MmMfma<1>::BM * MmMfma<1>::OSTR * 4 <= 2 * MmMfma<1>::BUF && MmMfma<3>::BM * MmMfma<3>::OSTR * 4 <= 2 * MmMfma<3>::BUF, "staging tile");

template <int NREP>
__global__ __launch_bounds__(MmMfma<NREP>::NTHR, 2) void mm_mfma256_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                         uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
                                                         const uint2 *__restrict__ faultRange,
                                                         uint8_t *__restrict__ detected)
{
    using G = MmMfma<NREP>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smemB[];
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemB + 2 * G::BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int lc = lane & 31, kh = lane >> 5;

    const uint32_t lb = xcd_logical_block(blockIdx.x, nblocks);
    const uint32_t mat = lb / (uint32_t)G::BPM;
    const int bim = (int)(lb - mat * (uint32_t)G::BPM);
    const int bm = bim / G::NBN, bn = bim - bm * G::NBN;
    const int row0 = bm * G::BM, col0 = bn * G::BNC;
    constexpr size_t nn = (size_t)G::N * G::N;
    const uint32_t *f = F + mat * nn;
    const uint32_t *s = S + mat * nn;
    if (tid < 4)
        sCnt[tid] = 0;

    // ---- staging slots (fixed for the whole k loop)
    // f: 128 rows x 8 k-quads = 1024 uint4 per slab, A_SLOTS per thread: row = tid/8 + (NTHR/8)u, k-quad = tid%8
    const int aRow = tid >> 3, aKq = tid & 7;
    // s: (BNC/4 column quads) x (8 k-quads) items of 4 k-rows x 4 columns, one per thread while they last
    const bool bOn = tid < G::B_ITEMS;
    const int bCq = tid / 8, bKq = tid & 7; // column quad, k-quad
    const bool bColOk = bOn && (col0 + 4 * bCq) < G::N;

    constexpr int AS = G::A_SLOTS, ARS = G::NTHR / 8; // slots, row stride between a thread's slots
    uint4 pa[AS], pb[4];
    auto gload = [&](int slab) {
        const int k0 = slab * G::KS;
#pragma unroll
        for (int u = 0; u < AS; ++u)
            pa[u] = *reinterpret_cast<const uint4 *>(f + (size_t)(row0 + aRow + ARS * u) * G::N + k0 + 4 * aKq);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            pb[kk] = make_uint4(0u, 0u, 0u, 0u);
            if (bColOk)
                pb[kk] = *reinterpret_cast<const uint4 *>(s + (size_t)(k0 + 4 * bKq + kk) * G::N + col0 + 4 * bCq);
        }
    };
    auto lstore = [&](int buf) {
        uint8_t *base = smemB + buf * G::BUF;
#pragma unroll
        for (int u = 0; u < AS; ++u) {
            const uint32_t y[4] = {mm_digits(pa[u].x), mm_digits(pa[u].y), mm_digits(pa[u].z), mm_digits(pa[u].w)};
            uint32_t w[4];
            mm_transpose4(y, w);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint32_t *>(base + p * G::PLANE_A + (aRow + ARS * u) * G::RSTR + 4 * aKq) = w[p];
        }
        if (bOn) {
            uint8_t *bb = base + 4 * G::PLANE_A;
            const uint32_t col[4][4] = {{pb[0].x, pb[1].x, pb[2].x, pb[3].x},  // column c: its 4 consecutive k
                                        {pb[0].y, pb[1].y, pb[2].y, pb[3].y},
                                        {pb[0].z, pb[1].z, pb[2].z, pb[3].z},
                                        {pb[0].w, pb[1].w, pb[2].w, pb[3].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t y[4] = {mm_digits(col[c][0]), mm_digits(col[c][1]), mm_digits(col[c][2]), mm_digits(col[c][3])};
                uint32_t w[4];
                mm_transpose4(y, w);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(bb + q * G::PLANE_B + (4 * bCq + c) * G::RSTR + 4 * bKq) = w[q];
            }
        }
    };

    v16i_t acc[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[rb][t][e] = 0;

    // this lane's operand rows / column
    const int myCol = wn * G::CPW + (lc < G::LPW ? lc / NREP : 0); // idle lane-columns re-read column 0 (ignored)
    const int aOff = (wm * 64 + lc) * G::RSTR + 16 * kh;
    const int bOff = 4 * G::PLANE_A + myCol * G::RSTR + 16 * kh;

    LaneMap<NREP> lm; // reuse the voter with this kernel's lane geometry: replicas = adjacent lane-columns of a half-wave
    lm.lane = lane;
    lm.r = lc % NREP;
    lm.q = lc / NREP;
    lm.live = lc < G::LPW;
    lm.base4 = (lane - lm.r) * 4;
    const int col = col0 + myCol;
    const bool colOk = lm.live && col < G::N;
    // which of this lane's 8 row groups (4 rows each) lie in a VALU workgroup range that owns an armed fault
    uint32_t skip = 0;
    if (faultRange && colOk) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int row = row0 + wm * 64 + (g >> 2) * 32 + 8 * (g & 3) + 4 * kh;
            const uint32_t vb = mat * (uint32_t)G::V_BPM + (uint32_t)(((row >> 2) * 64 + (col >> 2)) / G::V_TPB);
            skip |= (faultRange[vb].y != 0u ? 1u : 0u) << g;
        }
    }

    gload(0);
    lstore(0);
    const bool anySkip = __syncthreads_or(skip != 0u);
#pragma unroll 1
    for (int slab = 0; slab < G::NSLAB; ++slab) {
        const bool more = (slab + 1) < G::NSLAB;
        if (more)
            gload(slab + 1); // in flight under the MFMAs
        const uint8_t *base = smemB + (slab & 1) * G::BUF;
        v4i_t a[2][4], b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            a[0][p] = *reinterpret_cast<const v4i_t *>(base + p * G::PLANE_A + aOff);
            a[1][p] = *reinterpret_cast<const v4i_t *>(base + p * G::PLANE_A + aOff + 32 * G::RSTR);
            b[p] = *reinterpret_cast<const v4i_t *>(base + p * G::PLANE_B + bOff);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q + p < 4; ++q)
                    acc[rb][p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rb][p], b[q], acc[rb][p + q], 0, 0, 0);
        if (more)
            lstore((slab + 1) & 1); // the other buffer: every wave left it at the previous barrier
        __syncthreads();
    }

    // ---- recombine the digit products, vote across the replica lanes, single-copy store
    uint32_t *r = R + mat * nn;
    Tally tl;
    uint32_t detItems = 0;
    const bool writer = colOk && lm.r == 0;
    auto elemRow = [&](int rb, int e) { return wm * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh; }; // row inside the workgroup tile
    auto combine = [&](int rb, int e) {
        return (uint32_t)acc[rb][0][e] + ((uint32_t)acc[rb][1][e] << 8) + ((uint32_t)acc[rb][2][e] << 16) +
               ((uint32_t)acc[rb][3][e] << 24);
    };
    if (!anySkip) {
        // common case: the voted tile goes through LDS (the k-loop buffers are free after its last barrier) and leaves as
        // full-width row segments
        uint32_t *stage = reinterpret_cast<uint32_t *>(smemB);
        uint32_t missAcc = 0, detMask = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                Tally te;
                te.miss = 0;
                te.syncs = 0;
                te.det = 0;
                const uint32_t voted = xmr_final_vote_dpp<NREP>(combine(rb, e), true, te);
                missAcc += te.miss;
                detMask |= te.det << (rb * 16 + e);
                acc[rb][0][e] = (int)voted;
            }
        }
        if (writer) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    stage[elemRow(rb, e) * G::OSTR + myCol] = (uint32_t)acc[rb][0][e];
            tl.miss = missAcc;
            tl.syncs = 32;
            if (detMask) { // a real upset was out-voted / detected in this lane's elements
                detItems = (NREP == 2) ? (uint32_t)__builtin_popcount(detMask) : 0u;
                if (detected)
                    for (int i = 0; i < 32; ++i)
                        if ((detMask >> i) & 1u)
                            detected[mat * nn + (size_t)(row0 + elemRow(i >> 4, i & 15)) * G::N + col] = 1;
            }
        }
        __syncthreads();
        constexpr int SEG = G::BNC / 4; // uint4 per tile row
#pragma unroll
        for (int i = 0; i < (G::BM * SEG + G::NTHR - 1) / G::NTHR; ++i) {
            const int idx = tid + G::NTHR * i;
            const int orow = idx / SEG, c4 = idx - orow * SEG;
            if (idx < G::BM * SEG && col0 + 4 * c4 < G::N)
                *reinterpret_cast<uint4 *>(r + (size_t)(row0 + orow) * G::N + col0 + 4 * c4) =
                    *reinterpret_cast<const uint4 *>(stage + orow * G::OSTR + 4 * c4);
        }
    } else {
        // a VALU workgroup that owns an armed fault overlaps this tile: its elements belong to mm_general_kernel
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + elemRow(rb, e);
                const bool mine = colOk && !((skip >> (rb * 4 + (e >> 2))) & 1u);
                Tally te = tl;
                te.det = 0;
                const uint32_t voted = xmr_final_vote_dpp<NREP>(combine(rb, e), mine && lm.r == 0, te);
                tl.miss = te.miss;
                tl.syncs = te.syncs;
                if (mine && lm.r == 0) {
                    r[(size_t)row * G::N + col] = voted;
                    if (te.det) {
                        if (NREP == 2)
                            detItems += 1;
                        if (detected)
                            detected[mat * nn + (size_t)row * G::N + col] = 1;
                    }
                }
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
}


// ---------------------------------------------------------------------------------------------------------------------
// Panel-resident variant.  The kernel above converts the same 128 rows of f once per column block -- 13x for TMR -- and
// synchronises its waves at every k slab; profiling showed the matrix core busy a third of the time.  Here a workgroup owns
// 64 rows of ONE matrix for ALL its columns: the rows' byte planes (64 x 256 x 4 planes = 64 KB) are converted once and
// stay in LDS.  After that single barrier every wave is autonomous: it walks its own column tiles (32/NREP logical columns
// each), converts the s slabs it needs into a wave-private double buffer (loads issued two slabs ahead), runs its 20 MFMAs
// per slab against the shared panel, votes and stores its 64 x 32/NREP tile through the same private buffer.  No barrier,
// no cross-wave dependency, so one wave's load/convert/vote phases overlap its SIMD neighbour's MFMAs.
// LDS rows are 32 bytes (one k slab of one plane) with the two 16-byte halves swapped on rows whose bit 3 is set:
// 16 consecutive lanes of a ds_read_b128 then cover all 64 banks, without padding.
//
// Voter.  The common case costs 3 VALU per element: x = v ^ shl1(v), y = x | shl1(x) (DPP), bad |= y.  For replica 0,
// y == 0 <=> all copies agree, and then the vote IS v.  Only when some lane of the wave saw y != 0 is the full
// compare-and-select voter (xmr_final_vote_dpp, with its counters) run over the tile -- same result, same counts.
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
    static constexpr int PLANE_A = NSLAB * BM * 32;   // bytes per f byte-plane: [slab][row][32]
    static constexpr int A_PANEL = 4 * PLANE_A;       // 64 KB
    static constexpr int A_PER_THR = (BM * 8 * NSLAB) / NTHR; // uint4 of the panel per thread (16 / 8)
    static constexpr int PLANE_B = CPW * 32;          // bytes per s byte-plane of one slab of one wave: [column][32]
    static constexpr int B_BUF = 4 * PLANE_B;
    static constexpr int WAVE_LDS = 2 * B_BUF;        // double buffer == the wave's 64 x CPW output tile
    static constexpr size_t LDS_BYTES = (size_t)A_PANEL + NW * WAVE_LDS;
    static constexpr int KQPR = CPW > 16 ? 2 : 4;     // k-quads (tile rows x 4) one staging round of a wave covers
    static constexpr int B_ROUNDS = 8 / KQPR;         // staging rounds per slab (CPW * KQPR <= 64 lanes each)
    static constexpr int VW = (CPW % 4 == 0) ? 4 : 2; // words per output store
    static constexpr int SEGW = CPW / VW;             // stores per tile row
    static constexpr int V_TPB = 4 * (kWave / NREP);  // geometry of the VALU kernels, for the faulted-workgroup test
    static constexpr int V_BPM = (64 * 64 + V_TPB - 1) / V_TPB;
    static_assert(BM * CPW * 4 == WAVE_LDS, "output tile == slab double buffer");
};

__device__ __forceinline__ void wave_lds_sync()
{ // LDS operations of one wave execute in issue order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef MM_EXP
#define MM_EXP 0 // development knobs (tools/mm_exp.sh): 1 no s staging, 2 no epilogue, 4 no MFMA, 8 no interleave hints
#endif

template <int NREP>
__global__ __launch_bounds__(MmPanel<NREP>::NTHR, MmPanel<NREP>::WG_PER_CU) void mm_mfma_panel_kernel(
    const uint32_t *__restrict__ F, const uint32_t *__restrict__ S, uint32_t *__restrict__ R, uint32_t nblocks, Counters ctr,
    const uint2 *__restrict__ faultRange, uint8_t *__restrict__ detected)
{
    using G = MmPanel<NREP>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smemP[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    {
        uint4 pa[G::A_PER_THR];
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u) {
            const int i = tid + G::NTHR * u; // row = i / 64, k-quad of the row = i % 64
            pa[u] = *reinterpret_cast<const uint4 *>(f + (uint32_t)((i >> 6) * G::N + 4 * (i & 63)));
        }
#pragma unroll
        for (int u = 0; u < G::A_PER_THR; ++u) {
            const int i = tid + G::NTHR * u;
            const int row = i >> 6, kq = i & 63, slab = kq >> 3, k8 = kq & 7;
            const uint32_t y[4] = {mm_digits(pa[u].x), mm_digits(pa[u].y), mm_digits(pa[u].z), mm_digits(pa[u].w)};
            uint32_t w[4];
            mm_transpose4(y, w);
            const int dst = slab * (G::BM * 32) + row * 32 + (((k8 >> 2) ^ ((row >> 3) & 1)) * 16) + (k8 & 3) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint32_t *>(smemP + p * G::PLANE_A + dst) = w[p];
        }
    }

    // Armed faults: does any VALU workgroup range that overlaps this workgroup's 64 rows own one?  (wave-uniform; the
    // fine per-tile test below runs only then)
    bool blockArmed = false;
    if (faultRange) {
        const uint32_t rb64 = (uint32_t)row0 / 64u;
        const uint32_t vbLo = (rb64 * 1024u) / (uint32_t)G::V_TPB, vbHi = (rb64 * 1024u + 1023u) / (uint32_t)G::V_TPB;
        uint32_t any = 0;
        for (uint32_t vb = vbLo + (uint32_t)lane; vb <= vbHi; vb += 64u)
            any |= faultRange[mat * (uint32_t)G::V_BPM + vb].y;
        blockArmed = __ballot(any != 0u) != 0ull;
    }

    // ---- this wave's work: column tiles wave, wave + NW, ... ; one pipeline step = one k slab of one tile
    const int nTiles = (G::NCT - wave + G::NW - 1) / G::NW;
    const int nIt = nTiles * G::NSLAB;
    auto tileCol0 = [&](int it) __attribute__((always_inline)) { return (wave + G::NW * (it >> 3)) * G::CPW; };

    // s: one conversion item = four consecutive k of one column -> one word in each of the four planes.  Lane -> (column
    // c = l % CPW, k-quad l / CPW + KQPR*round): one load instruction fetches KQPR full tile rows (CPW contiguous words each).
    // Buffer loads: per-lane voffset fixed for the whole kernel, the step's slab / tile column in the scalar offset, the four
    // k rows in the immediate; reads past the matrix return 0 (columns past the edge and steps past the end are never consumed).
    // Lanes beyond CPW*KQPR mirror a live lane's item: same data, same destination, no branch.
    const __amdgpu_buffer_rsrc_t rsS =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(s), 0, (int)(nn * 4), 0x00020000);
    int voffB[G::B_ROUNDS], dstB[G::B_ROUNDS];
#pragma unroll
    for (int u = 0; u < G::B_ROUNDS; ++u) {
        const int l = lane < G::CPW * G::KQPR ? lane : lane - G::CPW * G::KQPR;
        const int c = l % G::CPW, kq = u * G::KQPR + l / G::CPW;
        voffB[u] = ((4 * kq) * G::N + c) * 4;
        dstB[u] = c * 32 + (((kq >> 2) ^ ((c >> 3) & 1)) * 16) + (kq & 3) * 4;
    }
    auto gloadB = [&](int it, uint32_t (&pb)[G::B_ROUNDS][4]) __attribute__((always_inline)) {
        const int soff = ((it & 7) * G::KS * G::N + tileCol0(it)) * 4;
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                pb[u][kk] = __builtin_amdgcn_raw_buffer_load_b32(rsS, voffB[u] + kk * G::N * 4, soff, 0);
    };
    auto lstoreB = [&](int it, const uint32_t (&pb)[G::B_ROUNDS][4]) __attribute__((always_inline)) {
        uint8_t *dstBuf = wbuf + (it & 1) * G::B_BUF;
#pragma unroll
        for (int u = 0; u < G::B_ROUNDS; ++u) {
            const uint32_t y[4] = {mm_digits(pb[u][0]), mm_digits(pb[u][1]), mm_digits(pb[u][2]), mm_digits(pb[u][3])};
            uint32_t w[4];
            mm_transpose4(y, w);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<uint32_t *>(dstBuf + q * G::PLANE_B + dstB[u]) = w[q];
        }
    };

    // this lane's operand row / column inside the wave tile
    const int tq = lc < G::LPW ? lc / NREP : 0; // idle lane-columns re-read the tile's first column
    const int aOff = lc * 32 + ((kh ^ ((lc >> 3) & 1)) * 16);
    const int bOff = tq * 32 + ((kh ^ ((tq >> 3) & 1)) * 16);
    const int rrep = lc % NREP;
    const bool live = lc < G::LPW;

    Tally tl;
    uint32_t detItems = 0;
    v16i_t acc[2][4];
    uint32_t pb0[G::B_ROUNDS][4], pb1[G::B_ROUNDS][4];

    gloadB(0, pb0);
    gloadB(1, pb1);
    lstoreB(0, pb0);
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
        if (NREP == 3) // OR distributes over the lane shift: one more exchange covers (replica 1 ^ replica 2) of every element
            bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0x130, 0xf, 0xf, false);
        uint32_t skip = 0;
        if (blockArmed) { // rare: which of this lane's 8 row groups (4 rows each) lie in a VALU workgroup range with a fault
            const int colc = col < G::N ? col : G::N - 1;
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const int row = row0 + (q4 >> 2) * 32 + 8 * (q4 & 3) + 4 * kh;
                const uint32_t vb = mat * (uint32_t)G::V_BPM + (uint32_t)(((row >> 2) * 64 + (colc >> 2)) / G::V_TPB);
                skip |= (faultRange[vb].y != 0u ? 1u : 0u) << q4;
            }
        }
        const bool slow = __ballot(writer && (skip != 0u || bad != 0u)) != 0ull; // wave-uniform
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
            // An upset was caught in this tile, or a VALU workgroup that owns an armed fault overlaps it (its elements belong
            // to mm_general_kernel).  Rare, so compact rather than fast: eight elements of every lane at a time go through
            // the wave's LDS, the full voter reads its two neighbours from there, element-wise stores.
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
                    const bool mine = writer && !((skip >> (rb * 4 + (e >> 2))) & 1u);
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

    // One pipeline step, hand-scheduled as ONE basic block: the loads for step it+2 go out, the 12 fragment reads of step
    // it, then its 20 MFMAs with the conversion of step it+1 slotted between them in ten small stages (sched_barrier pins the
    // order) -- the matrix core and the VALU run side by side instead of taking turns; two waves per SIMD drift into the same
    // phase otherwise.  FIRST (k slab 0 of a tile): the first MFMA into each accumulator starts from zero instead of
    // clearing 128 registers.
    auto step = [&](int it, uint32_t (&pbLoad)[G::B_ROUNDS][4], const uint32_t (&pbConv)[G::B_ROUNDS][4], auto firstTag)
                    __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(firstTag)::value;
        if (!(MM_EXP & 1))
            gloadB(it + 2, pbLoad);
        const int slab = it & 7;
        const uint8_t *pA = smemP + slab * (G::BM * 32) + aOff;
        const uint8_t *pB = wbuf + (it & 1) * G::B_BUF + bOff;
        v4i_t a[2][4], b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
            b[p] = *reinterpret_cast<const v4i_t *>(pB + p * G::PLANE_B);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            a[0][p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            a[1][p] = *reinterpret_cast<const v4i_t *>(pA + p * G::PLANE_A + 32 * 32);
        __builtin_amdgcn_sched_barrier(0);

        uint8_t *dstBuf = wbuf + ((it + 1) & 1) * G::B_BUF;
        uint32_t y[G::B_ROUNDS][4], t[4], w[4];
        // conversion of round u in five stages
        auto convStage = [&](int k) __attribute__((always_inline)) {
            const int u = k / 5, sub = k % 5;
            if (u >= G::B_ROUNDS || (MM_EXP & 1))
                return;
            if (sub == 0) {
                y[u][0] = mm_digits(pbConv[u][0]);
                y[u][1] = mm_digits(pbConv[u][1]);
            } else if (sub == 1) {
                y[u][2] = mm_digits(pbConv[u][2]);
                y[u][3] = mm_digits(pbConv[u][3]);
            } else if (sub == 2) {
                t[0] = __builtin_amdgcn_perm(y[u][1], y[u][0], 0x05010400u);
                t[1] = __builtin_amdgcn_perm(y[u][1], y[u][0], 0x07030602u);
                t[2] = __builtin_amdgcn_perm(y[u][3], y[u][2], 0x05010400u);
                t[3] = __builtin_amdgcn_perm(y[u][3], y[u][2], 0x07030602u);
            } else if (sub == 3) {
                w[0] = __builtin_amdgcn_perm(t[2], t[0], 0x05040100u);
                w[1] = __builtin_amdgcn_perm(t[2], t[0], 0x07060302u);
                w[2] = __builtin_amdgcn_perm(t[3], t[1], 0x05040100u);
                w[3] = __builtin_amdgcn_perm(t[3], t[1], 0x07060302u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint32_t *>(dstBuf + q * G::PLANE_B + dstB[u]) = w[q];
            }
        };
        constexpr int NSTAGE = 5 * G::B_ROUNDS;
        const v16i_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int m = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int q = 0; q + p < 4; ++q) {
                    if (!(MM_EXP & 4))
                        acc[rb][p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rb][p], b[q],
                                                                               (FIRST && p == 0) ? zero : acc[rb][p + q], 0, 0, 0);
                    // 20 MFMAs, NSTAGE (10 / 20) conversion stages: one or two stages behind every second / every MFMA
                    if (NSTAGE <= 10) {
                        if (m & 1)
                            convStage(m >> 1);
                    } else {
                        convStage(m);
                    }
                    ++m;
                    if (!(MM_EXP & 8))
                        __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

#pragma unroll 1
    for (int it = 0; it < nIt; it += 2) {
        // even step (slabs 0, 2, 4, 6): pb1 holds step it+1, pb0 is free for step it+2
        if ((it & 7) == 0)
            step(it, pb0, pb1, std::true_type{});
        else
            step(it, pb0, pb1, std::false_type{});
        wave_lds_sync();
        // odd step (slabs 1, 3, 5, 7)
        step(it + 1, pb1, pb0, std::false_type{});
        wave_lds_sync();
        if (((it + 1) & 7) == 7) {
            // the tile's epilogue stages through both slab buffers: step it+2 was converted into one of them a moment ago, so
            // redo that (cheap, the registers still hold it) once the tile is out
            if (!(MM_EXP & 2))
                tileEnd(it + 1);
            else if (acc[0][0][0] + acc[1][3][5] + acc[0][1][2] + acc[1][2][7] + acc[0][2][1] + acc[0][3][3] + acc[1][0][0] + acc[1][1][1] == 0x12345)
                r[lane] = 1;
            if (!(MM_EXP & 1))
                lstoreB(it + 2, pb0);
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
