// tools/crc_hyb_probe.hip -- round 6: the lookup-free (packed byte-step) crc16 walks against the shipped crc16_stream_kernel on the
// BASELINE stream (2^25 blocks per GPU, TMR): crc16_hybrid_kernel (table chains + packed walk in one wave, rows from HBM),
// crc16_packed_kernel (packed walk alone, whole rows staged in LDS by LDS-DMA), crc16_mixed_kernel<PW> (PW packed waves beside the
// table and 16 - PW lookup waves on one CU).  Kernel-only times with HIP events; every result compared word for word with the shipped
// kernel's and a sample with the reference recurrence (crc16.c:21-31) on the host; exit status 1 on any difference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/crc_hyb_probe tools/crc_hyb_probe.hip
//   tools/crc_hyb_probe [block_len = 255 (160..256)] [log2 blocks = 25] [reps = 5] [1 = the hybrid sweep too]
//   -DCRC_PACK_KNOCK=1 / 2: crc16_packed_kernel without its staging copy / without its walk (timing only);  -DCRC_MIX_PRIO: s_setprio 3
//   in the lookup waves of crc16_mixed_kernel.   Results: profiles/r06_crc16_hybrid.txt; tests/test_gpu_parity.py runs it small.
#include "../coast_amd/csrc/crc16_kernel.hip"
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

__global__ void fill_kernel(uint32_t *p, size_t nwords)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = (uint32_t)(z ^ (z >> 31));
    }
}

__global__ void diff_kernel(const uint16_t *a, const uint16_t *b, size_t n, unsigned long long *ndiff)
{
    unsigned long long d = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d += a[i] != b[i];
    if (d)
        atomicAdd(ndiff, d);
}

static uint16_t crc16_host(const uint8_t *p, unsigned len)
{
    uint16_t crc = 0xFFFF;
    while (len--) {
        uint8_t x = crc >> 8 ^ *p++;
        x ^= x >> 4;
        crc = (uint16_t)((crc << 8) ^ ((uint16_t)(x << 12)) ^ ((uint16_t)(x << 5)) ^ ((uint16_t)x));
    }
    return crc;
}

static int gMismatch = 0;
struct Env {
    uint8_t *data;
    uint32_t blockLen;
    uint64_t nblocks;
    uint16_t *out, *ref, *table;
    unsigned long long *slots, *ndiff;
    int numCUs, reps;
    uint64_t ntiles, ntWalk;
};

template <typename K> static void run(const Env &e, const char *name, K kern, int nt, bool isRef)
{
    const size_t lds = (size_t)kCrcTableBytes + 16;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t wavesNeeded = (e.ntiles + nt - 1) / nt;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)e.numCUs, (wavesNeeded + 15) / 16);
    Counters ctr{e.slots, 0u, 0u, nullptr, nullptr};
    FaultTab ft{nullptr, nullptr};
    uint16_t *dst = isRef ? e.ref : e.out;
    CK(hipMemset(dst, 0xA5, e.nblocks * 2));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    std::vector<float> ms;
    for (int r = 0; r < e.reps + 1; ++r) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kCrcStreamThreads), lds, 0, (const uint8_t *)e.data, e.blockLen, e.nblocks, dst,
                           (const uint16_t *)e.table, e.ntiles, e.ntWalk, ctr, ft, (uint8_t *)nullptr, (size_t)0, (size_t)0);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float m;
        CK(hipEventElapsedTime(&m, t0, t1));
        if (r)
            ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    unsigned long long nd = 0;
    if (!isRef) {
        CK(hipMemset(e.ndiff, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, (const uint16_t *)e.out, (const uint16_t *)e.ref, (size_t)e.nblocks, e.ndiff);
        CK(hipMemcpy(&nd, e.ndiff, 8, hipMemcpyDeviceToHost));
    }
    const double bytes = (double)e.nblocks * (e.blockLen + 2.0);
    const double med = ms[ms.size() / 2];
    printf("%-44s min %.4f med %.4f ms   %.3f TB/s = %.3f of 8 TB/s   %s\n", name, ms[0], med, bytes / med * 1e-9, bytes / med * 1e-9 / 8.0,
           isRef ? "(reference for the comparison)" : nd ? "MISMATCH" : "identical");
    if (nd)
        gMismatch = 1, printf("    %llu of %llu words differ\n", nd, (unsigned long long)e.nblocks);
    fflush(stdout);
}

template <typename K> static void run_packed(const Env &e, const char *name, K kern)
{
    using CP = CrcPack<3>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CP::kLds));
    const uint64_t wavesNeeded = (e.ntiles + kCrcSwarTiles - 1) / kCrcSwarTiles;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)e.numCUs, (wavesNeeded + CP::kWaves - 1) / CP::kWaves);
    Counters ctr{e.slots, 0u, 0u, nullptr, nullptr};
    FaultTab ft{nullptr, nullptr};
    CK(hipMemset(e.out, 0xA5, e.nblocks * 2));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    std::vector<float> ms;
    for (int r = 0; r < e.reps + 1; ++r) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(CP::kThreads), CP::kLds, 0, (const uint8_t *)e.data, e.blockLen, e.nblocks, e.out, e.ntiles,
                           e.ntWalk, ctr, ft, (uint8_t *)nullptr);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float m;
        CK(hipEventElapsedTime(&m, t0, t1));
        if (r)
            ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    unsigned long long nd = 0;
    CK(hipMemset(e.ndiff, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, (const uint16_t *)e.out, (const uint16_t *)e.ref, (size_t)e.nblocks, e.ndiff);
    CK(hipMemcpy(&nd, e.ndiff, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)e.nblocks * (e.blockLen + 2.0);
    const double med = ms[ms.size() / 2];
    printf("%-44s min %.4f med %.4f ms   %.3f TB/s = %.3f of 8 TB/s   %s\n", name, ms[0], med, bytes / med * 1e-9, bytes / med * 1e-9 / 8.0,
           nd ? "MISMATCH" : "identical");
    if (nd)
        gMismatch = 1, printf("    %llu of %llu words differ\n", nd, (unsigned long long)e.nblocks);
    fflush(stdout);
}

template <int PW> static void run_mixed(const Env &e, double share)
{
    const auto crc16_mixed_kernel = coast::crc16_mixed_kernel<PW>;
    const size_t kCrcMixLds = crc_mix_lds<PW>();
    CK(hipFuncSetAttribute((const void *)crc16_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCrcMixLds));
    const uint64_t fullTiles = std::min<uint64_t>(e.ntWalk, e.nblocks / 21);
    const uint64_t ntPacked = (uint64_t)(share * (double)fullTiles) / kCrcSwarTiles * kCrcSwarTiles;
    Counters ctr{e.slots, 0u, 0u, nullptr, nullptr};
    FaultTab ft{nullptr, nullptr};
    CK(hipMemset(e.out, 0xA5, e.nblocks * 2));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    std::vector<float> ms;
    for (int r = 0; r < e.reps + 1; ++r) {
        CK(hipEventRecord(t0));
        hipLaunchKernelGGL(crc16_mixed_kernel, dim3(e.numCUs), dim3(kCrcStreamThreads), kCrcMixLds, 0, (const uint8_t *)e.data, e.blockLen, e.nblocks,
                           e.out, (const uint16_t *)e.table, e.ntiles, e.ntWalk, ntPacked, ctr, ft, (uint8_t *)nullptr);
        CK(hipEventRecord(t1));
        CK(hipEventSynchronize(t1));
        float m;
        CK(hipEventElapsedTime(&m, t0, t1));
        if (r)
            ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    unsigned long long nd = 0;
    CK(hipMemset(e.ndiff, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, (const uint16_t *)e.out, (const uint16_t *)e.ref, (size_t)e.nblocks, e.ndiff);
    CK(hipMemcpy(&nd, e.ndiff, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)e.nblocks * (e.blockLen + 2.0);
    const double med = ms[ms.size() / 2];
    printf("crc16_mixed_kernel<%d>, packed share %.2f        min %.4f med %.4f ms   %.3f TB/s = %.3f of 8 TB/s   %s\n", PW, share, ms[0], med,
           bytes / med * 1e-9, bytes / med * 1e-9 / 8.0, nd ? "MISMATCH" : "identical");
    if (nd)
        gMismatch = 1, printf("    %llu of %llu words differ\n", nd, (unsigned long long)e.nblocks);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    Env e{};
    e.blockLen = argc > 1 ? (uint32_t)atoi(argv[1]) : 255u;
    const int lg = argc > 2 ? atoi(argv[2]) : 25;
    e.reps = argc > 3 ? atoi(argv[3]) : 5;
    e.nblocks = 1ull << lg;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    e.numCUs = prop.multiProcessorCount;
    const size_t bytes = (size_t)e.nblocks * e.blockLen;
    CK(hipMalloc((void **)&e.data, (bytes + 3) / 4 * 4 + 64));
    CK(hipMalloc((void **)&e.out, e.nblocks * 2));
    CK(hipMalloc((void **)&e.ref, e.nblocks * 2));
    CK(hipMalloc((void **)&e.table, kCrcTableBytes));
    CK(hipMalloc((void **)&e.slots, kCounterSlots * kSlotStride * 8));
    CK(hipMalloc((void **)&e.ndiff, 8));
    CK(hipMemset(e.slots, 0, kCounterSlots * kSlotStride * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t *)e.data, (bytes + 3) / 4);
    hipLaunchKernelGGL(crc16_table_kernel, dim3(256), dim3(256), 0, 0, e.table);
    CK(hipDeviceSynchronize());
    const bool aligned16 = (e.blockLen & 15u) == 0u;
    constexpr int IPW = 21;
    e.ntiles = (e.nblocks + IPW - 1) / IPW;
    const uint64_t tailRows = (24u + e.blockLen - 1u) / e.blockLen;
    const uint64_t tailTiles = std::min<uint64_t>(e.ntiles, (tailRows + IPW - 1) / IPW + 1);
    e.ntWalk = aligned16 ? e.ntiles : e.ntiles - tailTiles;
    printf("crc16 TMR stream: %llu blocks x %u bytes, %d CUs, %d timed launches each\n", (unsigned long long)e.nblocks, e.blockLen, e.numCUs, e.reps);
    const bool hyb = argc > 4 && atoi(argv[4]) != 0; // the hybrid sweep of profiles/r06_crc16_hybrid.txt
    if (aligned16) {
        run(e, "crc16_stream_kernel<3,2,true>  (shipped)", crc16_stream_kernel<3, 2, true>, 2, true);
        if (hyb) {
            run(e, "crc16_hybrid_kernel<3,0,true,8>  (packed only)", crc16_hybrid_kernel<3, 0, true, 8>, 4, false);
            run(e, "crc16_hybrid_kernel<3,1,true,4>", crc16_hybrid_kernel<3, 1, true, 4>, 5, false);
            run(e, "crc16_hybrid_kernel<3,1,true,8>", crc16_hybrid_kernel<3, 1, true, 8>, 5, false);
            run(e, "crc16_hybrid_kernel<3,2,true,4>", crc16_hybrid_kernel<3, 2, true, 4>, 6, false);
        }
        run_packed(e, "crc16_packed_kernel<3,true,8>  (LDS-staged)", crc16_packed_kernel<3, true, 8>);
        run_packed(e, "crc16_packed_kernel<3,true,16> (LDS-staged)", crc16_packed_kernel<3, true, 16>);
        run_packed(e, "crc16_packed_kernel<3,false,8> on aligned rows", crc16_packed_kernel<3, false, 8>);
    } else {
        run(e, "crc16_stream_kernel<3,1,false> (shipped)", crc16_stream_kernel<3, 1, false>, 1, true);
        if (hyb) {
            run(e, "crc16_hybrid_kernel<3,0,false,8> (packed only)", crc16_hybrid_kernel<3, 0, false, 8>, 4, false);
            run(e, "crc16_hybrid_kernel<3,1,false,4>", crc16_hybrid_kernel<3, 1, false, 4>, 5, false);
            run(e, "crc16_hybrid_kernel<3,1,false,8>", crc16_hybrid_kernel<3, 1, false, 8>, 5, false);
            run(e, "crc16_hybrid_kernel<3,2,false,4>", crc16_hybrid_kernel<3, 2, false, 4>, 6, false);
        }
        run_packed(e, "crc16_packed_kernel<3,false,8>  (LDS-staged)", crc16_packed_kernel<3, false, 8>);
        run_packed(e, "crc16_packed_kernel<3,false,16> (LDS-staged)", crc16_packed_kernel<3, false, 16>);
    }
    for (double share : {0.0, 0.25, 0.35, 0.45, 1.0})
        run_mixed<4>(e, share);
    for (double share : {0.1, 0.2})
        run_mixed<2>(e, share);
    // a sample of the reference kernel's words against the recurrence as written (crc16.c:21-31)
    const size_t ns = std::min<size_t>(e.nblocks, 4096);
    std::vector<uint8_t> hd(ns * e.blockLen);
    std::vector<uint16_t> hr(ns), ho(ns);
    CK(hipMemcpy(hd.data(), e.data, hd.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), e.ref, ns * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ho.data(), e.out, ns * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, bad2 = 0;
    for (size_t i = 0; i < ns; ++i) {
        const uint16_t w = crc16_host(hd.data() + i * e.blockLen, e.blockLen);
        bad += hr[i] != w;
        bad2 += ho[i] != w;
    }
    // ... and the stream's last blocks (the tail tiles)
    std::vector<uint8_t> td(64 * (size_t)e.blockLen);
    std::vector<uint16_t> to(64);
    CK(hipMemcpy(td.data(), e.data + (e.nblocks - 64) * e.blockLen, td.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(to.data(), e.out + (e.nblocks - 64), 128, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < 64; ++i)
        bad2 += to[i] != crc16_host(td.data() + i * e.blockLen, e.blockLen);
    printf("host recurrence on %zu blocks: shipped kernel %zu wrong, last variant (first blocks + the stream's last 64) %zu wrong\n", ns, bad, bad2);
    return (bad || bad2 || (gMismatch && CRC_PACK_KNOCK == 0)) ? 1 : 0;
}
