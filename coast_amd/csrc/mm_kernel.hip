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
// (cloning.cpp:2247-2255).  The k-loop is VALU bound (3x the integer MACs of the unprotected kernel); LDS traffic is
// 2 ds_read_b128 per 16 MACs.  Before the store every element is voted across its replicas (store-data sync,
// synchronization.cpp:476-561); only replica 0 writes the single output copy.
#include "xmr.hpp"

namespace coast {

struct MmGeom {
    int n;        // matrix side
    int tc;       // tiles per tile-row = ceil(n/4)
    int tiles;    // tc*tc tiles per matrix
    int bpm;      // workgroups per matrix
    int kt;       // k-chunk staged per barrier pair (power of two)
    int ktLog2;
    int rs;       // LDS row stride of the f panel (rows, multiple of 4)
    int npad;     // 4*tc
    uint32_t nblocks;
};

enum { SITE_MM_ACC = 0, SITE_MM_OPA = 1, SITE_MM_OPB = 2 };

template <int NREP>
__global__ __launch_bounds__(256) void mm_xmr_kernel(const uint32_t *__restrict__ F, const uint32_t *__restrict__ S,
                                                     uint32_t *__restrict__ R, MmGeom g, uint32_t syncEvery,
                                                     Counters ctr, FaultTab ft, int haveFaults,
                                                     uint8_t *__restrict__ detected)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *As = smem;                  // [kt][rs]   f panel, k-major
    uint32_t *Bs = smem + g.kt * g.rs;    // [kt][npad] s panel
    uint32_t *sCnt = Bs + g.kt * g.npad;  // 4 counters

    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    constexpr int TPB = 4 * IPW; // tiles per workgroup
    const LaneMap<NREP> lm;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int n = g.n;

    const uint32_t lb = xcd_logical_block(blockIdx.x, g.nblocks);
    const uint32_t mat = lb / (uint32_t)g.bpm;
    const int bim = (int)(lb - mat * (uint32_t)g.bpm);
    const int t0 = bim * TPB;
    const int tb = wave * IPW + lm.q;
    const bool live = lm.live && (t0 + tb) < g.tiles;
    const int t = live ? (t0 + tb) : t0;
    const int tr = t / g.tc, tcI = t - tr * g.tc;
    const int row0 = (t0 / g.tc) * 4;
    const int tLast = min(t0 + TPB, g.tiles) - 1;
    const int rows = min((tLast / g.tc) * 4 + 4, n) - row0; // rows of f staged (<= rs)
    const int aRow = tr * 4 - row0;
    const int i0 = tr * 4, j0 = tcI * 4;

    const size_t nn = (size_t)n * n;
    const uint32_t *f = F + mat * nn;
    const uint32_t *s = S + mat * nn;
    uint32_t *r = R + mat * nn;

    if (tid < 4)
        sCnt[tid] = 0;

    uint2 fr = make_uint2(0u, 0u);
    if (haveFaults)
        fr = ft.range[lb];
    const bool general = (fr.y != 0u) || (syncEvery != 0u);

    uint32_t acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        acc[e] = 0u;
    Tally tl;
    uint32_t detItems = 0;

    const bool vec = (n & 3) == 0;
    const int npad4 = g.npad >> 2;

    for (int k0 = 0; k0 < n; k0 += g.kt) {
        __syncthreads(); // previous chunk fully consumed
        // ---- stage the s panel: rows k0..k0+kt-1, all columns (coalesced, 16 B/lane when n % 4 == 0)
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
        // ---- stage the f panel transposed: As[kk][row]
        for (int idx = tid; idx < (g.rs << g.ktLog2); idx += 256) {
            const int rl = idx >> g.ktLog2, kk = idx & (g.kt - 1);
            const int k = k0 + kk, row = row0 + rl;
            uint32_t v = 0u;
            if (rl < rows && k < n)
                v = f[(size_t)row * n + k];
            As[kk * g.rs + rl] = v;
        }
        __syncthreads();

        if (!general) {
            // ---- fast path: no fault targets this workgroup, mandatory sync points only
#pragma unroll 4
            for (int kk = 0; kk < g.kt; ++kk) {
                const uint4 a = *reinterpret_cast<const uint4 *>(As + kk * g.rs + aRow);
                const uint4 b = *reinterpret_cast<const uint4 *>(Bs + kk * g.npad + j0);
                const uint32_t av[4] = {a.x, a.y, a.z, a.w};
                const uint32_t bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        acc[ii * 4 + jj] += av[ii] * bv[jj];
            }
        } else {
            // ---- general path: per-step injector hooks and optional loop-condition sync points
            const int kEnd = min(g.kt, n - k0);
            for (int kk = 0; kk < kEnd; ++kk) {
                const int k = k0 + kk;
                const uint4 a = *reinterpret_cast<const uint4 *>(As + kk * g.rs + aRow);
                const uint4 b = *reinterpret_cast<const uint4 *>(Bs + kk * g.npad + j0);
                uint32_t ae[16], be[16];
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
                    if (df.step != (uint32_t)k || (int)(df.local >> 4) != tb || (int)df.replica != lm.r || !lm.live)
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
                        const bool valid = live && (i0 + (e >> 2)) < n && (j0 + (e & 3)) < n;
                        Tally te = tl;
                        te.det = 0;
                        acc[e] = xmr_sync<NREP>(acc[e], lm, valid && lm.r == 0, te);
                        tl.miss = te.miss;
                        tl.syncs = te.syncs;
                        tl.det |= te.det << e; // per-element DWC flags
                    }
                }
            }
        }
    }

    // ---- injector hook after the loop (step == n hits the finished accumulator)
    if (general) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step != (uint32_t)n || df.site != SITE_MM_ACC || (int)(df.local >> 4) != tb ||
                (int)df.replica != lm.r || !lm.live)
                continue;
            const int fe = (int)(df.local & 15u);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e == fe)
                    acc[e] ^= 1u << (df.bit & 31u);
        }
    }

    // ---- store-data sync: vote every element across its replicas, replica 0 writes the single copy
    uint32_t out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const bool valid = live && (i0 + (e >> 2)) < n && (j0 + (e & 3)) < n;
        Tally te = tl;
        te.det = 0;
        out[e] = xmr_sync<NREP>(acc[e], lm, valid && lm.r == 0, te);
        tl.miss = te.miss;
        tl.syncs = te.syncs;
        tl.det |= te.det << e;
    }
    if (live && lm.r == 0) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = i0 + ii;
            if (i >= n)
                continue;
            uint32_t *dst = r + (size_t)i * n + j0;
            if (vec) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4(out[ii * 4], out[ii * 4 + 1], out[ii * 4 + 2], out[ii * 4 + 3]);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (j0 + jj < n)
                        dst[jj] = out[ii * 4 + jj];
            }
        }
        if (NREP == 2 && tl.det) {
            detItems = (uint32_t)__builtin_popcount(tl.det);
            if (detected) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((tl.det >> e) & 1u)
                        detected[mat * nn + (size_t)(i0 + (e >> 2)) * n + (j0 + (e & 3))] = 1;
            }
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, lb);
}

} // namespace coast
