// tools/aes_fixed_probe.hip -- round 6: where the fixed cost of a launch of the persistent aes kernels goes (profiles/r06_aes_fixed_cost.txt).
// The launch shape of aes128_dec_rep_kernel / aes128_enc_rep_kernel (256 x 1024 threads x 128 KiB of LDS; 512 x 1024 x 64 KiB) with the kernel's
// fixed parts added one at a time: nothing, the table fill, the barrier, block_tally, block_fold (xmr.hpp's own).  Two clocks per variant: HIP events
// around each launch (what coast_set_profiling reports as kernel_ms) and 200 launches back to back divided by 200.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/aes_fixed_probe tools/aes_fixed_probe.hip
#include "../coast_amd/csrc/xmr.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace coast;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e__ = (x);                                                                  \
        if (e__ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

// PARTS: 0 = nothing, 1 = + fill, 2 = + barrier and an LDS read, 3 = + block_tally, 4 = + block_fold, 5 = 2 + one atomic per workgroup and counter on the totals
template <int PARTS, int LDSKB>
__global__ __launch_bounds__(1024) void fixed_kernel(const uint4 *__restrict__ image, uint32_t *__restrict__ sink, Counters ctr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smem + LDSKB * 1024);
    const int tid = threadIdx.x;
    if constexpr (PARTS >= 1) {
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
#pragma unroll
        for (int i = 0; i < LDSKB * 1024 / 16 / 1024; ++i)
            dst[i * 1024 + tid] = image[i * 1024 + tid];
    }
    if (tid < 4)
        sCnt[tid] = 0;
    if constexpr (PARTS >= 2) {
        __syncthreads();
        const uint32_t v = reinterpret_cast<uint32_t *>(smem)[(tid * 37) & 8191];
        if (v == 0x12345678u)
            sink[0] = v; // (never: keeps the fill alive)
    }
    if constexpr (PARTS == 3 || PARTS == 4)
        block_tally(0u, (tid & 63) == 0 ? 1u : 0u, 0u, sCnt, ctr, blockIdx.x);
    if constexpr (PARTS == 4)
        block_fold(ctr, blockIdx.x, sCnt + 3);
    if constexpr (PARTS == 5) { // the workgroup's sums straight into the totals: no slots, no ticket, no last workgroup
        const uint32_t ws = wave_sum((tid & 63) == 0 ? 1u : 0u);
        if ((tid & 63) == 0 && ws)
            atomicAdd(&sCnt[1], ws);
        __syncthreads();
        if (tid == 0) {
            if (sCnt[0])
                atomicAdd(&ctr.totals[0], (unsigned long long)sCnt[0]);
            if (sCnt[1])
                atomicAdd(&ctr.totals[1], (unsigned long long)sCnt[1]);
            if (sCnt[2])
                atomicAdd(&ctr.totals[2], (unsigned long long)sCnt[2]);
            if (blockIdx.x == 0)
                atomicAdd(&ctr.totals[3], (unsigned long long)ctr.foldLaunches);
        }
    }
}

template <typename K> static void run(const char *name, K kern, int grid, size_t lds, const uint4 *image, uint32_t *sink, Counters ctr)
{
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    std::vector<float> ms;
    for (int r = 0; r < 220; ++r) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, image, sink, ctr);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float m;
        CK(hipEventElapsedTime(&m, t0, t1));
        if (r >= 20)
            ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    CK(hipEventRecord(t0));
    for (int r = 0; r < 200; ++r)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, image, sink, ctr);
    CK(hipEventRecord(t1));
    CK(hipEventSynchronize(t1));
    float tot;
    CK(hipEventElapsedTime(&tot, t0, t1));
    printf("%-58s bracketed: min %.2f med %.2f us    back to back: %.2f us per launch\n", name, ms[0] * 1e3, ms[ms.size() / 2] * 1e3, tot / 200 * 1e3);
    fflush(stdout);
}

int main()
{
    uint4 *image;
    uint32_t *sink, *ticket;
    unsigned long long *slots, *totals;
    CK(hipMalloc((void **)&image, 128 * 1024));
    CK(hipMemset(image, 1, 128 * 1024));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMalloc((void **)&ticket, kTicketWords * 4));
    CK(hipMemset(ticket, 0, kTicketWords * 4));
    CK(hipMalloc((void **)&slots, kCounterSlots * kSlotStride * 8));
    CK(hipMemset(slots, 0, kCounterSlots * kSlotStride * 8));
    CK(hipMalloc((void **)&totals, 64));
    CK(hipMemset(totals, 0, 64));
    Counters plain{slots, 0u, 0u, nullptr, nullptr};
    Counters fold{slots, 0u, 1u, totals, ticket};
    const size_t l128 = 128 * 1024 + 16 + 1024 * 8, l64 = 64 * 1024 + 16 + 1024 * 8;
    printf("decrypt shape: 256 workgroups x 1024 threads, 128 KiB of LDS\n");
    run("  nothing", fixed_kernel<0, 128>, 256, l128, image, sink, plain);
    run("  + fill (8 x uint4 per thread)", fixed_kernel<1, 128>, 256, l128, image, sink, plain);
    run("  + barrier, one LDS read", fixed_kernel<2, 128>, 256, l128, image, sink, plain);
    run("  + block_tally", fixed_kernel<3, 128>, 256, l128, image, sink, plain);
    run("  + block_fold", fixed_kernel<4, 128>, 256, l128, image, sink, fold);
    run("  instead: the workgroup's sums straight into the totals", fixed_kernel<5, 128>, 256, l128, image, sink, fold);
    printf("encrypt shape: 512 workgroups x 1024 threads, 64 KiB of LDS\n");
    run("  nothing", fixed_kernel<0, 64>, 512, l64, image, sink, plain);
    run("  + fill (4 x uint4 per thread)", fixed_kernel<1, 64>, 512, l64, image, sink, plain);
    run("  + barrier, one LDS read", fixed_kernel<2, 64>, 512, l64, image, sink, plain);
    run("  + block_tally", fixed_kernel<3, 64>, 512, l64, image, sink, plain);
    run("  + block_fold", fixed_kernel<4, 64>, 512, l64, image, sink, fold);
    run("  instead: the workgroup's sums straight into the totals", fixed_kernel<5, 64>, 512, l64, image, sink, fold);
    printf("for scale: 256 workgroups x 1024 threads, 16 bytes of LDS\n");
    run("  nothing", fixed_kernel<0, 0>, 256, 16 + 0, image, sink, plain);
    unsigned long long h[4];
    CK(hipMemcpy(h, totals, 32, hipMemcpyDeviceToHost));
    printf("(fold totals: syncs %llu launches %llu)\n", h[1], h[3]);
    return 0;
}
