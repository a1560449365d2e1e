// vote_kernel.hip -- default-mode (memory-replicated) TMR / DWC: the sync point where three (two) memory copies
// re-converge.
//
// COAST's DEFAULT mode replicates memory as well as registers and does not vote stores (docs/source/passes.rst:329,337;
// cloning.cpp:2417-2462, synchronization.cpp:211-215): the clones of a protected region run on their own copies of the
// data, and values are voted where they leave the sphere of replication -- return values, arguments of unprotected calls,
// stores to unprotected globals (synchronization.cpp:741-949, 563-738; verification.cpp:625-682).  On the GPU the three
// clones are three launches of the unprotected kernel on three HBM copies (there is nothing to exchange inside the
// region, so lane adjacency buys nothing in this mode), and this kernel is the exit vote over the result arrays:
//     vote = (a == b) ? a : c  per 32-bit word,  TMR_ERROR_CNT += !((a==b)&&(a==c)),  __SYNC_COUNT += 1 per word;
// with `scrub` the voted word is written back into every copy that differs (the copies re-converge, as all three IR
// values continue from the voted one, synchronization.cpp:527-529).  DWC: a != b flags the word, nothing is repaired.
// Pure HBM streaming: 12 (8) bytes read + 4 written per word; 16-byte vector accesses, grid-stride.
#include "xmr.hpp"

namespace coast {

template <int NC>
__global__ __launch_bounds__(256) void sync_copies_kernel(uint32_t *__restrict__ c0, uint32_t *__restrict__ c1,
                                                          uint32_t *__restrict__ c2, uint64_t nwords,
                                                          uint32_t *__restrict__ voted, int scrub, Counters ctr,
                                                          uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t miss = 0, syncs = 0, det = 0;
    const uint64_t nvec = nwords >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const uint4 a = reinterpret_cast<const uint4 *>(c0)[v];
        const uint4 b = reinterpret_cast<const uint4 *>(c1)[v];
        uint4 c = a;
        if (NC == 3)
            c = reinterpret_cast<const uint4 *>(c2)[v];
        const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
        uint32_t o[4];
        uint32_t bad = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool e01 = av[e] == bv[e], e02 = av[e] == cv[e];
            o[e] = (NC == 3) ? (e01 ? av[e] : cv[e]) : av[e];
            const uint32_t m = (NC == 3) ? ((e01 && e02) ? 0u : 1u) : (e01 ? 0u : 1u);
            bad |= m << e;
        }
        syncs += 4;
        if (bad) {
            const uint32_t nb = (uint32_t)__builtin_popcount(bad);
            if (NC == 3)
                miss += nb;
            else
                det += nb;
            if (detected) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((bad >> e) & 1u)
                        detected[4 * v + e] = 1;
            }
            if (scrub && NC == 3) { // copies re-converge on the voted value
                const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
                reinterpret_cast<uint4 *>(c0)[v] = ov;
                reinterpret_cast<uint4 *>(c1)[v] = ov;
                reinterpret_cast<uint4 *>(c2)[v] = ov;
            }
        }
        if (voted)
            reinterpret_cast<uint4 *>(voted)[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    // tail words (nwords % 4), one per thread of the first workgroup
    if (blockIdx.x == 0 && threadIdx.x < (nwords & 3u)) {
        const uint64_t w = (nvec << 2) + threadIdx.x;
        const uint32_t a = c0[w], b = c1[w], c = (NC == 3) ? c2[w] : a;
        const bool e01 = a == b, e02 = a == c;
        const uint32_t o = (NC == 3) ? (e01 ? a : c) : a;
        const uint32_t m = (NC == 3) ? ((e01 && e02) ? 0u : 1u) : (e01 ? 0u : 1u);
        syncs += 1;
        if (m) {
            if (NC == 3)
                miss += 1;
            else
                det += 1;
            if (detected)
                detected[w] = 1;
            if (scrub && NC == 3) {
                c0[w] = o;
                c1[w] = o;
                c2[w] = o;
            }
        }
        if (voted)
            voted[w] = o;
    }
    block_tally(miss, syncs, det, sCnt, ctr, blockIdx.x);
}

// injectFaultMem (simulation/platform/resources/injector.py:209-235): flip one bit of one byte of device memory
__global__ void flip_memory_kernel(uint8_t *p, unsigned bit)
{
    *p ^= (uint8_t)(1u << (bit & 7u));
}

} // namespace coast
