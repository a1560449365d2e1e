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
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF + 16;
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

    gload(0);
    lstore(0);
    __syncthreads();
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
    LaneMap<NREP> lm; // reuse the voter with this kernel's lane geometry: replicas = adjacent lane-columns of a half-wave
    lm.lane = lane;
    lm.r = lc % NREP;
    lm.q = lc / NREP;
    lm.live = lc < G::LPW;
    lm.base4 = (lane - lm.r) * 4;
    const int col = col0 + myCol;
    const bool colOk = lm.live && col < G::N;
    uint32_t *r = R + mat * nn;
    Tally tl;
    uint32_t detItems = 0;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + wm * 64 + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const uint32_t v = (uint32_t)acc[rb][0][e] + ((uint32_t)acc[rb][1][e] << 8) + ((uint32_t)acc[rb][2][e] << 16) +
                               ((uint32_t)acc[rb][3][e] << 24);
            bool mine = colOk;
            if (faultRange && mine) { // elements of a faulted VALU workgroup belong to mm_general_kernel
                const uint32_t vb = mat * (uint32_t)G::V_BPM + (uint32_t)(((row >> 2) * 64 + (col >> 2)) / G::V_TPB);
                mine = faultRange[vb].y == 0u;
            }
            Tally te = tl;
            te.det = 0;
            const uint32_t voted = xmr_final_vote_dpp<NREP>(v, mine && lm.r == 0, te);
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
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
}

} // namespace coast
