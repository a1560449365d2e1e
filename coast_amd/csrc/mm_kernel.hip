// mm_kernel.hip -- protected matrix_multiply for gfx950.
//
// Replaces: the TMR/DWC-transformed matrix_multiply of tests/mm_common/mm_common_tmr.c:3-20
//   r[i][j] = (mm_t) sum_k f[i][k] * s[k][j]     (32-bit wrapping product, sum truncated on the store)
//
// Mapping.  Logical work item = one output element r[i][j]; a lane owns a 4x4 register tile of items and the NREP
// replicas of that tile sit in NREP adjacent lanes (21 tiles/wave for TMR, 32 for DWC).  A workgroup (4 waves) owns
// 4*IPW consecutive tiles of ONE matrix in row-major tile order, so it needs a few rows of f and a full-width panel
// of s per k-chunk: both are loaded from HBM/L2 once, coalesced, into LDS, and every replica lane reads the same LDS
// word (broadcast) -- the reference's -noMemReplication rule: one memory copy, loads repeated from the same address
// (cloning.cpp:2247-2255).  Before the store every element is voted across its replicas (store-data sync,
// synchronization.cpp:476-561); only replica 0 writes the single output copy.
//
// Two kernels share the mapping:
//   mm_fast_kernel     every workgroup no armed fault points at: double-buffered LDS panels, next chunk prefetched
//                      into registers under the MACs, one v_mad_u64_u32 per MAC (the 64-bit accumulator is the
//                      reference's `unsigned long sum`; only its low word is ever stored).  VALU bound: 3x the integer
//                      MACs of the unprotected kernel, 2 ds_read_b128 per 16 MACs.
//   mm_general_kernel  workgroups that own an armed fault, or every workgroup when sync_every != 0: one k step at a
//                      time with the injector hooks (flip = old XOR 1<<bit on the named replica's register) and the
//                      optional loop-condition sync points.
#include "xmr.hpp"
#include <type_traits>

namespace coast {

struct MmGeom {
    int n;        // matrix side
    int tc;       // tiles per tile-row = ceil(n/4)
    int tiles;    // tc*tc tiles per matrix
    int bpm;      // workgroups per matrix
    int kt;       // k-chunk staged per barrier (power of two)
    int ktLog2;
    int rs;       // LDS row stride of the f panel (rows, multiple of 4)
    int npad;     // 4*tc
    uint32_t nblocks;
};

enum { SITE_MM_ACC = 0, SITE_MM_OPA = 1, SITE_MM_OPB = 2, SITE_MM_I = 3, SITE_MM_J = 4, SITE_MM_K = 5 };

constexpr int kMmMaxB = 4; // uint4 of the s panel a thread stages per chunk  (kt * npad/4 <= 256*kMmMaxB)
constexpr int kMmMaxA = 4; // dwords of the f panel a thread stages per chunk (kt * rs     <= 256*kMmMaxA)

// where this lane's tile sits
template <int NREP> struct MmLane {
    LaneMap<NREP> lm;
    uint32_t lb, mat;
    int t0, tb, row0, rows, aRow, i0, j0;
    bool live;
    __device__ __forceinline__ MmLane(const MmGeom &g, uint32_t logicalBlock)
    {
        constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
        constexpr int TPB = 4 * IPW;
        lb = logicalBlock;
        mat = lb / (uint32_t)g.bpm;
        const int bim = (int)(lb - mat * (uint32_t)g.bpm);
        t0 = bim * TPB;
        tb = (int)(threadIdx.x >> 6) * IPW + lm.q;
        live = lm.live && (t0 + tb) < g.tiles;
        const int t = live ? (t0 + tb) : t0;
        const int tr = t / g.tc, tcI = t - tr * g.tc;
        row0 = (t0 / g.tc) * 4;
        const int tLast = min(t0 + TPB, g.tiles) - 1;
        rows = min((tLast / g.tc) * 4 + 4, g.n) - row0; // rows of f staged (<= rs)
        aRow = tr * 4 - row0;
        i0 = tr * 4;
        j0 = tcI * 4;
    }
};

// store-data sync + single-copy store + counters (shared epilogue)
template <int NREP>
__device__ __forceinline__ void mm_epilogue(const uint32_t acc[16], const MmLane<NREP> &L, const MmGeom &g,
                                            uint32_t *__restrict__ r, Tally &tl, uint8_t *__restrict__ detected,
                                            size_t matOff, uint32_t *sCnt, const Counters &ctr)
{
    const int n = g.n;
    const bool vec = (n & 3) == 0;
    uint32_t out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const bool valid = L.live && (L.i0 + (e >> 2)) < n && (L.j0 + (e & 3)) < n;
        Tally te = tl;
        te.det = 0;
        out[e] = xmr_store_sync<NREP>(acc[e], L.lm, valid && L.lm.r == 0, te);
        tl.miss = te.miss;
        tl.syncs = te.syncs;
        tl.det |= te.det << e; // per-element DWC flags
    }
    uint32_t detItems = 0;
    if (L.live && L.lm.r == 0) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = L.i0 + ii;
            if (i >= n)
                continue;
            uint32_t *dst = r + (size_t)i * n + L.j0;
            if (vec) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(out[ii * 4], out[ii * 4 + 1], out[ii * 4 + 2], out[ii * 4 + 3]);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (L.j0 + jj < n)
                        dst[jj] = out[ii * 4 + jj];
            }
        }
        if (tl.det) { // per-element flags: unequal copies at a sync point (DWC: detected, TMR: corrected)
            if (NREP == 2)
                detItems = (uint32_t)__builtin_popcount(tl.det);
            if (detected) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((tl.det >> e) & 1u)
                        detected[matOff + (size_t)(L.i0 + (e >> 2)) * n + (L.j0 + (e & 3))] = 1;
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, L.lb);
}

// One k step of a lane's 4x4 tile as a single asm block: acc[4*i+j] += a[i] * b[j], one v_mad_u64_u32 per MAC
// (measured 5.1 cycles per wave-instruction vs 4.5 + 2.1 for v_mul_lo_u32 + half a v_add3_u32, tools/valu_microbench).
// Written as asm because hipcc narrows a 64-bit accumulation whose high word is dead back to v_mul_lo_u32 + add, and as
// ONE block so that no compiler boundary nops sit between the 16 MACs.  Pure register VALU: no memory operand, no
// hazard with its neighbours beyond the vcc clobber.
__device__ __forceinline__ void mac16_u64(unsigned long long (&acc)[16], const uint4 &a, const uint4 &b)
{
    asm("v_mad_u64_u32 %0, vcc, %16, %20, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %21, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %16, %22, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %16, %23, %3\n\t"
        "v_mad_u64_u32 %4, vcc, %17, %20, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %17, %21, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %17, %22, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %17, %23, %7\n\t"
        "v_mad_u64_u32 %8, vcc, %18, %20, %8\n\t"
        "v_mad_u64_u32 %9, vcc, %18, %21, %9\n\t"
        "v_mad_u64_u32 %10, vcc, %18, %22, %10\n\t"
        "v_mad_u64_u32 %11, vcc, %18, %23, %11\n\t"
        "v_mad_u64_u32 %12, vcc, %19, %20, %12\n\t"
        "v_mad_u64_u32 %13, vcc, %19, %21, %13\n\t"
        "v_mad_u64_u32 %14, vcc, %19, %22, %14\n\t"
        "v_mad_u64_u32 %15, vcc, %19, %23, %15"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
          "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]),
          "+v"(acc[14]), "+v"(acc[15])
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)
        : "vcc");
}

// ------------------------------------------------------------------------------------------------ fast path
// VEC: n % 4 == 0, every panel row is 16-byte aligned (the bench shape); !VEC handles ragged sides (9, 19, 30 ...).
// KT: the k-chunk as a compile-time constant so the MAC loop is fully unrolled and the LDS reads of step kk+1 are
// issued under the 16 MACs of step kk.
// one workgroup, a k step at a time, with the injector hooks and the optional accumulator votes (defined below)
template <int NREP>
__device__ __forceinline__ void mm_stepwise_body(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                 uint32_t *__restrict__ R, const MmGeom &g, uint32_t syncEvery, const Counters &ctr,
                                                 const FaultTab &ft, uint32_t lb, uint8_t *__restrict__ detected);

template <int NREP, bool VEC, int KT>
__global__ __launch_bounds__(256) void mm_fast_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                      uint32_t *__restrict__ R, MmGeom g, Counters ctr, FaultTab ft,
                                                      uint8_t *__restrict__ detected)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int panel = g.kt * (g.rs + g.npad); // dwords per buffer: As[kt][rs] then Bs[kt][npad]
    uint32_t *sCnt = smem + 2 * panel;

    const uint32_t lb = xcd_logical_block(blockIdx.x, g.nblocks);
    if (ft.range && ft.range[lb].y != 0u) {
        // round 3: a workgroup that owns an armed upset walks its k steps with the injector hooks HERE (same LDS budget, same
        // geometry, and the same mm_epilogue: voter, stores, counters) instead of leaving them to a twin launch on the side stream
        mm_stepwise_body<NREP>(F, S, R, g, 0u, ctr, ft, lb, detected);
        return;
    }
    const MmLane<NREP> L(g, lb);
    const int tid = threadIdx.x;
    const int n = g.n;
    const size_t nn = (size_t)n * n;
    const uint32_t *f = F + L.mat * nn;
    const uint32_t *s = S + L.mat * nn;

    if (tid < 4)
        sCnt[tid] = 0;

    // per-thread staging slots (fixed for the whole k loop)
    constexpr bool vec = VEC;
    const int npad4 = g.npad >> 2;
    int bKk[kMmMaxB], bCol[kMmMaxB];
#pragma unroll
    for (int u = 0; u < kMmMaxB; ++u) {
        const int idx = tid + 256 * u;
        bKk[u] = (idx < g.kt * npad4) ? idx / npad4 : -1;
        bCol[u] = 4 * (idx - max(bKk[u], 0) * npad4);
    }
    int aKk[kMmMaxA], aRl[kMmMaxA];
#pragma unroll
    for (int u = 0; u < kMmMaxA; ++u) {
        const int idx = tid + 256 * u;
        aRl[u] = idx >> g.ktLog2;
        aKk[u] = (aRl[u] < g.rs) ? (idx & (g.kt - 1)) : -1;
    }
    uint4 pb[kMmMaxB];
    uint32_t pa[kMmMaxA];

    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < kMmMaxB; ++u) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int k = k0 + bKk[u];
            if (bKk[u] >= 0 && k < n) {
                const uint32_t *src = s + (size_t)k * n + bCol[u];
                if (vec) {
                    v = *reinterpret_cast<const uint4 *>(src);
                } else {
                    v.x = (bCol[u] + 0 < n) ? src[0] : 0u;
                    v.y = (bCol[u] + 1 < n) ? src[1] : 0u;
                    v.z = (bCol[u] + 2 < n) ? src[2] : 0u;
                    v.w = (bCol[u] + 3 < n) ? src[3] : 0u;
                }
            }
            pb[u] = v;
        }
#pragma unroll
        for (int u = 0; u < kMmMaxA; ++u) {
            uint32_t v = 0u;
            const int k = k0 + aKk[u];
            if (aKk[u] >= 0 && aRl[u] < L.rows && k < n)
                v = f[(size_t)(L.row0 + aRl[u]) * n + k];
            pa[u] = v;
        }
    };
    auto lstore = [&](int buf) {
        uint32_t *As = smem + buf * panel;
        uint32_t *Bs = As + g.kt * g.rs;
#pragma unroll
        for (int u = 0; u < kMmMaxB; ++u)
            if (bKk[u] >= 0)
                *reinterpret_cast<uint4 *>(Bs + bKk[u] * g.npad + bCol[u]) = pb[u];
#pragma unroll
        for (int u = 0; u < kMmMaxA; ++u)
            if (aKk[u] >= 0)
                As[aKk[u] * g.rs + aRl[u]] = pa[u];
    };

    unsigned long long acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        acc[e] = 0ull;

    gload(0);
    lstore(0);
    __syncthreads();
    const int nchunks = (n + g.kt - 1) >> g.ktLog2;
    for (int c = 0; c < nchunks; ++c) {
        const bool more = (c + 1) < nchunks;
        if (more)
            gload((c + 1) << g.ktLog2); // in flight under the MACs below
        const uint32_t *As = smem + (c & 1) * panel + L.aRow;
        const uint32_t *Bs = smem + (c & 1) * panel + KT * g.rs + L.j0;
        uint4 a = *reinterpret_cast<const uint4 *>(As);
        uint4 b = *reinterpret_cast<const uint4 *>(Bs);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            uint4 an = a, bn = b;
            if (kk + 1 < KT) {
                an = *reinterpret_cast<const uint4 *>(As + (kk + 1) * g.rs);
                bn = *reinterpret_cast<const uint4 *>(Bs + (kk + 1) * g.npad);
            }
            mac16_u64(acc, a, b);
            a = an;
            b = bn;
        }
        if (more)
            lstore((c + 1) & 1); // the other buffer: every wave left it at the previous barrier
        __syncthreads();
    }

    uint32_t lo[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        lo[e] = (uint32_t)acc[e]; // r_matrix[i][j] = sum truncates (mm_common_tmr.c:16)
    Tally tl;
    mm_epilogue<NREP>(lo, L, g, R + L.mat * nn, tl, detected, L.mat * nn, sCnt, ctr);
}

// ------------------------------------------------------------------------------------------------ fast path, side 256
// The headline shape (BASELINE.json configs[1]) with its whole geometry as compile-time constants: panel loads are
// buffer_load with one precomputed per-thread voffset and a scalar per-chunk soffset (no per-chunk address VALU, rows
// past the matrix come back as 0 from the descriptor's bounds check), LDS reads carry immediate offsets, and the chunk
// loop is unrolled by two so that both LDS buffers are addressed statically.  Same mapping and results as
// mm_fast_kernel<NREP, true, 16>.
// KT = 8 (17 KiB of LDS per workgroup, 78 VGPRs) measured 1.5 % (TMR) to 5 % (unprotected) faster than 16 or 4.
template <int NREP, int KT_ = 8> struct Mm256 {
    static constexpr int N = 256, TC = 64, TILES = TC * TC, NPAD = 256, KT = KT_;
    static constexpr int IPW = kWave / NREP, TPB = 4 * IPW;
    static constexpr int BPM = (TILES + TPB - 1) / TPB;
    static constexpr int max_tile_rows()
    {
        int m = 1;
        for (int b = 0; b < BPM; ++b) {
            const int t0 = b * TPB, t1 = (t0 + TPB < TILES ? t0 + TPB : TILES) - 1;
            const int r = t1 / TC - t0 / TC + 1;
            m = r > m ? r : m;
        }
        return m;
    }
    static constexpr int RS = 4 * max_tile_rows();
    static constexpr int PANEL = KT * (RS + NPAD); // dwords per LDS buffer
    static constexpr size_t LDS_BYTES = (size_t)2 * PANEL * 4 + 16;
    static_assert(RS * KT <= 256, "one f-panel dword per thread");
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int NREP, int KT_ = 8>
__global__ __launch_bounds__(256) void mm_fast256_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                         uint32_t *__restrict__ R, MmGeom g, Counters ctr,
                                                         const uint2 *__restrict__ faultRange,
                                                         uint8_t *__restrict__ detected)
{
    using G = Mm256<NREP, KT_>;
    constexpr int NB = G::KT / 4; // uint4 of the s panel per thread per chunk
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *sCnt = smem + 2 * G::PANEL;

    const uint32_t lb = xcd_logical_block(blockIdx.x, g.nblocks);
    if (faultRange && faultRange[lb].y != 0u)
        return; // an armed fault points into this workgroup: mm_general_kernel owns it
    const MmLane<NREP> L(g, lb);
    const int tid = threadIdx.x;
    constexpr size_t nn = (size_t)G::N * G::N;
    if (tid < 4)
        sCnt[tid] = 0;

    // descriptors from blockIdx-derived scalars only (provably wave-uniform: no waterfall loops)
    const __amdgpu_buffer_rsrc_t rsS =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(S + L.mat * nn), 0, (int)(nn * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsF =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(F + L.mat * nn), 0, (int)(nn * 4), 0x00020000);

    // s panel: KT rows x 64 uint4 -> thread owns (kk = tid/64 + 4u, c4 = tid%64); byte offset inside the chunk is the
    // same in HBM and in LDS.  f panel: RS rows x KT k -> thread tid < KT*RS owns (rl = tid/KT, kk = tid%KT).
    int voffB[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u)
        voffB[u] = (((tid >> 6) + 4 * u) * G::NPAD + 4 * (tid & 63)) * 4;
    const bool aOn = tid < G::RS * G::KT;
    const int aRl = tid / G::KT, aKk = tid % G::KT;
    const int voffA = aOn ? ((L.row0 + aRl) * G::N + aKk) * 4 : -4; // -4 = 0xfffffffc: out of range -> 0
    const int ldsA = (aKk * G::RS + aRl) * 4;

    u32x4_t pb[NB];
    uint32_t pa;
    auto gload = [&](int c) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
            pb[u] = __builtin_amdgcn_raw_buffer_load_b128(rsS, voffB[u], c * (G::KT * G::N * 4), 0);
        pa = __builtin_amdgcn_raw_buffer_load_b32(rsF, voffA, c * (G::KT * 4), 0);
    };
    auto lstore = [&](auto bufTag) {
        constexpr int BUF = decltype(bufTag)::value;
        char *base = reinterpret_cast<char *>(smem) + BUF * G::PANEL * 4;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            *reinterpret_cast<u32x4_t *>(base + G::KT * G::RS * 4 + voffB[u]) = pb[u];
        if (aOn)
            *reinterpret_cast<uint32_t *>(base + ldsA) = pa;
    };

    unsigned long long acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        acc[e] = 0ull;
    const char *rdA = reinterpret_cast<const char *>(smem) + L.aRow * 4;
    const char *rdB = reinterpret_cast<const char *>(smem) + (G::KT * G::RS + L.j0) * 4;
    auto compute = [&](auto bufTag) {
        constexpr int BUF = decltype(bufTag)::value;
        // LDS reads run two k steps ahead of the MACs that consume them
        uint4 a0 = *reinterpret_cast<const uint4 *>(rdA + BUF * G::PANEL * 4);
        uint4 b0 = *reinterpret_cast<const uint4 *>(rdB + BUF * G::PANEL * 4);
        uint4 a1 = *reinterpret_cast<const uint4 *>(rdA + (BUF * G::PANEL + G::RS) * 4);
        uint4 b1 = *reinterpret_cast<const uint4 *>(rdB + (BUF * G::PANEL + G::NPAD) * 4);
#pragma unroll
        for (int kk = 0; kk < G::KT; ++kk) {
            uint4 an = a1, bn = b1;
            if (kk + 2 < G::KT) {
                an = *reinterpret_cast<const uint4 *>(rdA + (BUF * G::PANEL + (kk + 2) * G::RS) * 4);
                bn = *reinterpret_cast<const uint4 *>(rdB + (BUF * G::PANEL + (kk + 2) * G::NPAD) * 4);
            }
            mac16_u64(acc, a0, b0);
            a0 = a1;
            b0 = b1;
            a1 = an;
            b1 = bn;
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    constexpr int NCHUNK = G::N / G::KT; // even
    gload(0);
    lstore(B0{});
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NCHUNK; c += 2) {
        gload(c + 1); // in flight under the MACs
        compute(B0{});
        lstore(B1{});
        __syncthreads();
        const bool more = (c + 2) < NCHUNK;
        if (more)
            gload(c + 2);
        compute(B1{});
        if (more)
            lstore(B0{});
        __syncthreads();
    }

    uint32_t lo[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        lo[e] = (uint32_t)acc[e];
    Tally tl;
    mm_epilogue<NREP>(lo, L, g, R + L.mat * nn, tl, detected, L.mat * nn, sCnt, ctr);
}

// ------------------------------------------------------------------------------------------------ general path
template <int NREP>
__device__ __forceinline__ void mm_stepwise_body(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                 uint32_t *__restrict__ R, const MmGeom &g, uint32_t syncEvery, const Counters &ctr,
                                                 const FaultTab &ft, uint32_t lb, uint8_t *__restrict__ detected)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *As = smem;                 // [kt][rs]   f panel, k-major
    uint32_t *Bs = smem + g.kt * g.rs;   // [kt][npad] s panel
    uint32_t *sCnt = smem + 2 * g.kt * (g.rs + g.npad);

    MmLane<NREP> L(g, lb);
    L.lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const LaneMap<NREP> &lm = L.lm;
    const int tid = threadIdx.x;
    const int n = g.n;
    const size_t nn = (size_t)n * n;
    const uint32_t *f = F + L.mat * nn;
    const uint32_t *s = S + L.mat * nn;

    if (tid < 4)
        sCnt[tid] = 0;
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[lb];

    uint32_t acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        acc[e] = 0u;
    Tally tl;
    const bool vec = (n & 3) == 0;
    const int npad4 = g.npad >> 2;

    for (int k0 = 0; k0 < n; k0 += g.kt) {
        __syncthreads();
        for (int idx = tid; idx < g.kt * npad4; idx += 256) {
            const int kk = idx / npad4, c4 = idx - kk * npad4;
            const int k = k0 + kk;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (k < n) {
                const uint32_t *src = s + (size_t)k * n + 4 * c4;
                if (vec) {
                    v = *reinterpret_cast<const uint4 *>(src);
                } else {
                    const int j = 4 * c4;
                    v.x = (j + 0 < n) ? src[0] : 0u;
                    v.y = (j + 1 < n) ? src[1] : 0u;
                    v.z = (j + 2 < n) ? src[2] : 0u;
                    v.w = (j + 3 < n) ? src[3] : 0u;
                }
            }
            *reinterpret_cast<uint4 *>(Bs + kk * g.npad + 4 * c4) = v;
        }
        for (int idx = tid; idx < (g.rs << g.ktLog2); idx += 256) {
            const int rl = idx >> g.ktLog2, kk = idx & (g.kt - 1);
            const int k = k0 + kk;
            uint32_t v = 0u;
            if (rl < L.rows && k < n)
                v = f[(size_t)(L.row0 + rl) * n + k];
            As[kk * g.rs + rl] = v;
        }
        __syncthreads();

        const int kEnd = min(g.kt, n - k0);
        for (int kk = 0; kk < kEnd; ++kk) {
            const int k = k0 + kk;
            const uint4 a = *reinterpret_cast<const uint4 *>(As + kk * g.rs + L.aRow);
            const uint4 b = *reinterpret_cast<const uint4 *>(Bs + kk * g.npad + L.j0);
            uint32_t ae[16], be[16]; // per-element operand copies: an operand upset hits ONE item's MAC
            {
                const uint32_t av[4] = {a.x, a.y, a.z, a.w};
                const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    ae[e] = av[e >> 2];
                    be[e] = bv[e & 3];
                }
            }
            for (uint32_t q = 0; q < fr.y; ++q) {
                const DevFault df = ft.list[fr.x + q];
                if (df.step != (uint32_t)k || (int)(df.local >> 4) != L.tb || (int)df.replica != lm.r || !lm.live)
                    continue;
                const uint32_t m = 1u << (df.bit & 31u);
                const int fe = (int)(df.local & 15u);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e == fe) {
                        if (df.site == SITE_MM_ACC)
                            acc[e] ^= m;
                        else if (df.site == SITE_MM_OPA)
                            ae[e] ^= m;
                        else if (df.site == SITE_MM_OPB)
                            be[e] ^= m;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[e] += ae[e] * be[e];
            if (syncEvery && ((uint32_t)(k + 1) % syncEvery) == 0u && (k + 1) < n) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool valid = L.live && (L.i0 + (e >> 2)) < n && (L.j0 + (e & 3)) < n;
                    Tally te = tl;
                    te.det = 0;
                    acc[e] = xmr_sync<NREP>(acc[e], lm, valid && lm.r == 0, te);
                    tl.miss = te.miss;
                    tl.syncs = te.syncs;
                    tl.det |= te.det << e;
                }
            }
        }
    }
    // injector hook after the loop (step == n hits the finished accumulator)
    for (uint32_t q = 0; q < fr.y; ++q) {
        const DevFault df = ft.list[fr.x + q];
        if (df.step != (uint32_t)n || df.site != SITE_MM_ACC || (int)(df.local >> 4) != L.tb ||
            (int)df.replica != lm.r || !lm.live)
            continue;
        const int fe = (int)(df.local & 15u);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (e == fe)
                acc[e] ^= 1u << (df.bit & 31u);
    }
    mm_epilogue<NREP>(acc, L, g, R + L.mat * nn, tl, detected, L.mat * nn, sCnt, ctr);
}


template <int NREP>
__global__ __launch_bounds__(256) void mm_general_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                         uint32_t *__restrict__ R, MmGeom g, uint32_t syncEvery,
                                                         Counters ctr, FaultTab ft,
                                                         const uint32_t *__restrict__ blockList,
                                                         uint8_t *__restrict__ detected)
{
    mm_stepwise_body<NREP>(F, S, R, g, syncEvery, ctr, ft, blockList ? blockList[blockIdx.x] : blockIdx.x, detected);
}

// ------------------------------------------------------------------------------------------------ counters inside the sphere of replication
// matrix_multiply with its three loops as written (mm_common_tmr.c:3-20), for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the work
// item is the CALL -- one matrix product per lane group -- and i, j, k and `sum` are replica-private registers of one sequential
// walk, the registers the pass triplicates.  Sync points (frozen in oracle/coast_oracle.c:mm_call_indexed, statement by
// statement): every evaluation of the three loop conditions -- (N+1)(N^2+N+1) of them, SURVEY.md section 3.2 --, the GEP
// offsets of f[i][k], s[k][j] (loads) and r[i][j] (store), the data of that store.  f, s, r are memory: one copy; a load uses
// the original instruction's address in every copy (cloning.cpp:2247-2255), i.e. the voted offset or replica 0's.  Opt-in and
// slow by construction (one lane walks N^3 MACs): the sync-point-parity form of the kernel, not the throughput form.
template <int NREP>
__global__ __launch_bounds__(64) void mm_indexed_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                        uint32_t *__restrict__ R, uint32_t n, uint64_t nmats, Counters ctr,
                                                        FaultTab ft, uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    const bool lss = xmr_local_sync_on(ctr.flags); // COAST_F_LOCAL_STORE_SYNC: sum += .., k++, j++, i++ are stores into allocas at -O0
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t mat = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && mat < nmats;
    const bool cnt = live && lm.r == 0;
    const bool writer = cnt; // the single memory copy is written by the original store (replica 0's lane)
    const uint64_t nn = (uint64_t)n * n;
    const uint32_t *f = F + (live ? mat : 0) * nn, *s = S + (live ? mat : 0) * nn;
    uint32_t *r = R + (live ? mat : 0) * nn;
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();
    if (writer) // elements a derailed walk never stores stay 0
        for (uint64_t e = 0; e < nn; ++e)
            r[e] = 0u;
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    const uint32_t N = live ? n : 0u;
    const uint64_t cap = 4ull * ((uint64_t)n + 1ull) * (nn + n + 1ull) + 1024ull;
    Tally tl;
    uint32_t i = 0u, j = 0u, k = 0u, sum = 0u;
    uint64_t tick = 0;
    auto hook = [&]() __attribute__((always_inline)) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if ((uint64_t)df.step != tick || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            const uint32_t m = 1u << (df.bit & 31u);
            if (df.site == SITE_MM_I)
                i ^= m;
            else if (df.site == SITE_MM_J)
                j ^= m;
            else if (df.site == SITE_MM_K)
                k ^= m;
            else if (df.site == SITE_MM_ACC)
                sum ^= m;
        }
    };
    auto cond = [&](uint32_t reg) __attribute__((always_inline)) { // one evaluated loop condition
        ++tick;
        if (!lm.live) // the idle lane of a TMR wave has no call of its own (its replica group would wrap to lanes 0, 1): N = 0, it leaves
            return reg < N;
        return xmr_steer<NREP>(reg < N ? 1u : 0u, lm, bs, cnt, tl) != 0u;
    };
    // the lanes of one matrix always take the same (voted, or replica 0's) direction; different matrices have the same trip
    // counts unless an upset derails one, so the wave stays together except around upsets
    for (;;) {                                                           // for (i = 0; i < side; i++)              :10
        hook();
        if (tick >= cap || !cond(i))
            break;
        j = 0u;
        for (;;) {                                                       // for (j = 0; j < side; j++)              :11
            hook();
            if (tick >= cap || !cond(j))
                break;
            sum = 0u;
            k = 0u;
            for (;;) {                                                   // for (k = 0; k < side; k++)              :13
                hook();
                if (tick >= cap || !cond(k))
                    break;
                const uint32_t fi = xmr_steer<NREP>(i, lm, ls, cnt, tl), fk = xmr_steer<NREP>(k, lm, ls, cnt, tl); // f[i][k]
                const uint32_t sk = xmr_steer<NREP>(k, lm, ls, cnt, tl), sj = xmr_steer<NREP>(j, lm, ls, cnt, tl); // s[k][j]
                const uint32_t a = (fi < N && fk < N) ? f[(uint64_t)fi * n + fk] : 0u;
                const uint32_t b = (sk < N && sj < N) ? s[(uint64_t)sk * n + sj] : 0u;
                sum = xmr_local_sync<NREP>(sum + a * b, lm, lss, cnt, tl);
                k = xmr_local_sync<NREP>(k + 1u, lm, lss, cnt, tl);
            }
            const uint32_t ri = xmr_steer<NREP>(i, lm, ss, cnt, tl), rj = xmr_steer<NREP>(j, lm, ss, cnt, tl);     // r[i][j] = sum
            uint32_t v = xmr_store_sync<NREP>(sum, lm, cnt, tl);
            if (NREP != 3 || !lm.storeSync)
                v = xmr_rep0<NREP>(v, lm);
            if (writer && ri < N && rj < N)
                r[(uint64_t)ri * n + rj] = v;
            j = xmr_local_sync<NREP>(j + 1u, lm, lss, cnt, tl);
        }
        i = xmr_local_sync<NREP>(i + 1u, lm, lss, cnt, tl);
    }
    uint32_t detItems = 0;
    if (cnt && tl.det) { // unequal copies at a sync point of this call (DWC: detected, TMR: corrected)
        if (NREP == 2)
            detItems = 1;
        if (detected)
            detected[mat * nn] = 1; // the item is the call: its flag is the matrix's first byte
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
