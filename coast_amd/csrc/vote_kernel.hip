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
//
// Operand types (round 3; synchronization.cpp:57-62, 70-88, 1380-1443, 1469-1530).  The pass compares integers with `icmp eq` and
// floating-point values with `fcmp oeq` -- ordered: a NaN equals nothing, itself included; -0.0 equals +0.0 -- and a VECTOR operand
// takes a different counter path: the select is lane-wise, TMR_ERROR_CNT += the add-reduction over the lanes of
// (a ne b) | (a ne c) with `icmp ne` / `fcmp one` (ordered-and-not-equal: a NaN lane is NOT counted), and the function returns
// before the -countSyncs increment (:1394-1396), so a vector sync point does not move __SYNC_COUNT.  FP32 / VECTOR select those rules.
#include "xmr.hpp"

namespace coast {

template <bool FP32> __device__ __forceinline__ bool vote_eq(uint32_t a, uint32_t b)
{
    if constexpr (FP32)
        return __uint_as_float(a) == __uint_as_float(b); // fcmp oeq
    else
        return a == b;
}
template <bool FP32> __device__ __forceinline__ bool vote_ne(uint32_t a, uint32_t b)
{
    if constexpr (FP32) {
        const float x = __uint_as_float(a), y = __uint_as_float(b);
        return x < y || x > y; // fcmp one: false when either is a NaN
    } else {
        return a != b;
    }
}

// one voted word: returns the vote; m = counted (TMR) / flagged (DWC); differs = some copy is not bitwise the voted value
template <int NC, bool FP32, bool VECTOR>
__device__ __forceinline__ uint32_t vote_word(uint32_t a, uint32_t b, uint32_t c, uint32_t &m, bool &differs)
{
    const bool e01 = vote_eq<FP32>(a, b), e02 = vote_eq<FP32>(a, c);
    const uint32_t o = (NC == 3) ? (e01 ? a : c) : a;
    if (NC == 3)
        m = VECTOR ? ((vote_ne<FP32>(a, b) || vote_ne<FP32>(a, c)) ? 1u : 0u) : ((e01 && e02) ? 0u : 1u);
    else
        m = e01 ? 0u : 1u;
    differs = NC == 3 && (a != o || b != o || c != o);
    return o;
}

template <int NC, bool FP32 = false, bool VECTOR = false>
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
        bool anyDiff = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t m;
            bool d;
            o[e] = vote_word<NC, FP32, VECTOR>(av[e], bv[e], cv[e], m, d);
            bad |= m << e;
            anyDiff = anyDiff || d;
        }
        if (!VECTOR) // a vector sync point returns before the -countSyncs increment (synchronization.cpp:1394-1396)
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
        }
        if (scrub && anyDiff) { // the copies re-converge on the voted value (for integers: exactly the counted words)
            const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<uint4 *>(c0)[v] = ov;
            reinterpret_cast<uint4 *>(c1)[v] = ov;
            reinterpret_cast<uint4 *>(c2)[v] = ov;
        }
        if (voted)
            reinterpret_cast<uint4 *>(voted)[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    // tail words (nwords % 4), one per thread of the first workgroup
    if (blockIdx.x == 0 && threadIdx.x < (nwords & 3u)) {
        const uint64_t w = (nvec << 2) + threadIdx.x;
        const uint32_t a = c0[w], b = c1[w], c = (NC == 3) ? c2[w] : a;
        uint32_t m;
        bool d;
        const uint32_t o = vote_word<NC, FP32, VECTOR>(a, b, c, m, d);
        if (!VECTOR)
            syncs += 1;
        if (m) {
            if (NC == 3)
                miss += 1;
            else
                det += 1;
            if (detected)
                detected[w] = 1;
        }
        if (scrub && d) {
            c0[w] = o;
            c1[w] = o;
            c2[w] = o;
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
