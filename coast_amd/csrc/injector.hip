// injector.hip -- device side of the fault injector.
//
// Replaces: the QEMU/GDB campaign engine (simulation/platform/supervisor.py, resources/injector.py:125-260).  The host
// decodes each coast_fault into the geometry of the launch that will consume it (which workgroup owns the item, which
// slot inside it) and sorts by workgroup; the table is copied and indexed ON A SIDE STREAM while the main stream keeps
// running, then event-ordered before the consuming kernel.  The flip itself -- old XOR (1 << bit), flipOneBit,
// injector.py:202-207 -- is applied inside the protected kernel, at the named step, to the named replica's register.
#include "xmr.hpp"

namespace coast {

// range[b] = {first index, count} for every workgroup b that owns >= 1 fault; `range` is zeroed beforehand.
__global__ void fault_range_kernel(const DevFault *__restrict__ list, uint32_t k, uint2 *__restrict__ range)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k)
        return;
    const uint32_t b = list[i].block;
    if (i != 0 && list[i - 1].block == b)
        return; // not the head of its run
    uint32_t c = 1;
    while (i + c < k && list[i + c].block == b)
        ++c;
    range[b] = make_uint2(i, c);
}

// Fold the per-workgroup counter slots into the totals {errors, syncs, dwc_items, launches} and clear the slots.
__global__ void reduce_counters_kernel(unsigned long long *__restrict__ slots, unsigned long long *__restrict__ totals,
                                       unsigned long long launches)
{
    __shared__ unsigned long long acc[3];
    const int t = threadIdx.x; // one thread per slot, kCounterSlots threads
    if (t < 3)
        acc[t] = 0ull;
    __syncthreads();
    unsigned long long *slot = slots + (size_t)t * kSlotStride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long v = slot[c];
        if (v) {
            atomicAdd(&acc[c], v);
            slot[c] = 0ull;
        }
    }
    __syncthreads();
    if (t < 3 && acc[t])
        atomicAdd(&totals[t], acc[t]);
    if (t == 3 && launches)
        atomicAdd(&totals[3], launches);
}

} // namespace coast
