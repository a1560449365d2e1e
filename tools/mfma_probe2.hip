// mfma_probe2.hip -- what can the int8 matrix core sustain on RANDOM data (DVFS: power-limited clock), and what does a
// fully software-pipelined step (one wave per SIMD, 64 x 64 lane-column wave tile, fragments double-buffered in registers)
// reach?  Development tool for mm_mfma_kernel.hip; prints one line per experiment.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t rnd(uint32_t &s) { s = s * 1664525u + 1013904223u; return s ^ (s >> 13); }

// (1) pure MFMA, NACC independent accumulators, operands = RND ? per-lane random bytes : constants
template <int NACC, bool RND, int WPS> __global__ __launch_bounds__(256, WPS) void rate(int *out, int iters)
{
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    v4i a[4], b[4];
    for (int p = 0; p < 4; ++p) {
        a[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){1, 2, 3, 4};
        b[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){5, 6, 7, 8};
    }
    v16i c[NACC];
    for (int t = 0; t < NACC; ++t) c[t] = (v16i){0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            c[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 3], b[(t >> 2) & 3], c[t], 0, 0, 0);
    }
    int r = 0;
    for (int t = 0; t < NACC; ++t) for (int e = 0; e < 16; ++e) r += c[t][e];
    if (r == 0x12345678) out[threadIdx.x] = r;
}
template <int NACC, bool RND, int WPS> static void run_rate(int *dD, int cus)
{
    const int iters = 20000 / NACC * 4, blocks = cus * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<NACC, RND, WPS>), dim3(blocks), dim3(256), 0, 0, dD, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((rate<NACC, RND, WPS>), dim3(blocks), dim3(256), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * iters * NACC;
    printf("rate  acc=%2d %s waves/SIMD=%d: %.3f ms  %.0f TOPS  (= %.2f GHz if every SIMD issues one MFMA per 32 cycles)\n", NACC,
           RND ? "random" : "const ", WPS, ms, mf * 65536 / (ms * 1e-3) * 1e-12, mf / (cus * 4) * 32 / (ms * 1e-3) * 1e-9);
}

// (2) Design-Y step: one wave per SIMD.  Wave tile 64 rows x 64 lane-columns: 2 x 2 blocks x 4 limb sums = 16 accumulators.
// Per 32-deep k slab: 8 A + 8 B fragment reads (ds_read_b128) for the NEXT slab while the 40 MFMAs of this slab run on the
// fragments read one slab earlier; FLAGS&2: conversion VALU (80) ; &4: 16 ds_write_b32 ; &8: 8 global dwordx2 loads.
template <int FLAGS> __global__ __launch_bounds__(256, 1) void stepY(int *out, const uint2 *g, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    for (int i = threadIdx.x; i < 36864; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = rnd(s);
    __syncthreads();
    v4i a[2][2][4], b[2][2][4]; // [buffer][block][plane]
    for (int u = 0; u < 2; ++u) for (int k = 0; k < 2; ++k) for (int p = 0; p < 4; ++p) {
        a[u][k][p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
        b[u][k][p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    }
    v16i c[16];
    for (int t = 0; t < 16; ++t) c[t] = (v16i){0};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t) x[t] = rnd(s);
    uint2 gl[8] = {};
    const unsigned char *pa = lds + (lane & 31) * 32 + ((lane >> 5) ^ ((lane >> 3) & 1)) * 16;
    unsigned char *pw = lds + 131072 + wave * 4096 + (lane % 40) * 32 + (lane / 40) * 4;
    auto body = [&](int i, auto curTag) __attribute__((always_inline)) {
        constexpr int cur = decltype(curTag)::value, nxt = cur ^ 1;
        int m = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q + p < 4; ++q) {
                        c[(rb * 2 + cb) * 4 + p + q] =
                            __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][rb][p], b[cur][cb][q], c[(rb * 2 + cb) * 4 + p + q], 0, 0, 0);
                        if ((FLAGS & 1) && m < 16) { // next slab's fragments: one read behind each of the first 16 MFMAs
                            const int k = m >> 3, blk = (m >> 2) & 1, p2 = m & 3;
                            if (k == 0)
                                a[nxt][blk][p2] = *reinterpret_cast<const v4i *>(pa + p2 * 16384 + ((i + 1) & 7) * 2048 + blk * 1024);
                            else
                                b[nxt][blk][p2] = *reinterpret_cast<const v4i *>(pa + 65536 + p2 * 16384 + ((i + 1) & 7) * 2048 + blk * 1024);
                        }
                        if (FLAGS & 2) {
                            x[m % 8] = __builtin_amdgcn_perm(x[m % 8], x[(m + 3) % 8] + gl[m % 8].x, 0x05010400u);
                            x[(m + 1) % 8] = (x[(m + 1) % 8] + 0x80808080u) ^ gl[(m + 1) % 8].y;
                        }
                        if ((FLAGS & 4) && m % 2 == 1 && m < 32)
                            *reinterpret_cast<uint32_t *>(pw + (m / 2) * 128 + ((i + 1) & 1) * 2048) = x[m % 8];
                        if ((FLAGS & 8) && m >= 30 && m < 38)
                            gl[m - 30] = g[(size_t)((i * 8 + (m - 30)) & 1023) * 4096 + blockIdx.x * 64 + lane];
                        ++m;
                        __builtin_amdgcn_sched_barrier(0);
                    }
    };
    for (int i = 0; i < iters; i += 2) {
        body(i, std::integral_constant<int, 0>{});
        body(i + 1, std::integral_constant<int, 1>{});
    }
    int r = 0;
    for (int t = 0; t < 16; ++t) for (int e = 0; e < 16; ++e) r += c[t][e];
    for (int t = 0; t < 8; ++t) r += (int)x[t];
    if (r == 0x12345678) out[threadIdx.x] = r;
}
template <int FLAGS> static void run_stepY(int *dD, const uint2 *g, int cus)
{
    const int iters = 4000, blocks = cus;
    hipFuncSetAttribute((const void *)stepY<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stepY<FLAGS>, dim3(blocks), dim3(256), 160 * 1024, 0, dD, g, 10);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    hipEventRecord(e0); hipLaunchKernelGGL(stepY<FLAGS>, dim3(blocks), dim3(256), 160 * 1024, 0, dD, g, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * iters * 40;
    printf("stepY flags=%2d (1 frag reads, 2 VALU, 4 ds_write, 8 global loads) 1 wave/SIMD: %.1f ns per 40-MFMA step, %.0f TOPS  [%s]\n", FLAGS,
           ms * 1e6 / iters, mf * 65536 / (ms * 1e-3) * 1e-12, hipGetErrorString(err));
}

// (3) the same questions for v_mfma_i32_16x16x64_i8 (half the MACs per instruction, a quarter of the accumulator registers per
// output block): pure rate, and a mock of a "replicas in register blocks" step -- wave tile 64 rows x 16 logical columns x 3
// replicas = 4 x 3 x 4 limb sums of 4 registers (192), 64-deep k slab: 120 MFMAs, 16 A + 12 B fragment reads, the conversion of
// a 16-column x 64-k slab (FLAGS as above).
typedef int v4acc __attribute__((ext_vector_type(4)));
template <int NACC, bool RND, int WPS = 1> __global__ __launch_bounds__(256 * WPS, 1) void rate16(int *out, int iters)
{
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    v4i a[4], b[4];
    for (int p = 0; p < 4; ++p) {
        a[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){1, 2, 3, 4};
        b[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){5, 6, 7, 8};
    }
    v4acc c[NACC];
    for (int t = 0; t < NACC; ++t) c[t] = (v4acc){0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 3], b[(t >> 2) & 3], c[t], 0, 0, 0);
    }
    int r = 0;
    for (int t = 0; t < NACC; ++t) for (int e = 0; e < 4; ++e) r += c[t][e];
    if (r == 0x12345678) out[threadIdx.x] = r;
}
template <int NACC, bool RND, int WPS = 1> static void run_rate16(int *dD, int cus)
{
    const int iters = 40000 / NACC * 4, blocks = cus;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate16<NACC, RND, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, dD, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((rate16<NACC, RND, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * WPS * iters * NACC;
    printf("rate16x16x64 acc=%2d %s waves/SIMD=%d: %.3f ms  %.0f TOPS\n", NACC, RND ? "random" : "const ", WPS, ms, mf * 32768 / (ms * 1e-3) * 1e-12);
}
template <int FLAGS> __global__ __launch_bounds__(256, 1) void stepZ(int *out, const uint2 *g, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    for (int i = threadIdx.x; i < 36864; i += 256) reinterpret_cast<uint32_t *>(lds)[i] = rnd(s);
    __syncthreads();
    v4i a[4][4], b[3][4]; // [row block][plane], [replica][plane]
    for (int k = 0; k < 4; ++k) for (int p = 0; p < 4; ++p)
        a[k][p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    for (int k = 0; k < 3; ++k) for (int p = 0; p < 4; ++p)
        b[k][p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    v4acc c[48];
    for (int t = 0; t < 48; ++t) c[t] = (v4acc){0};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t) x[t] = rnd(s);
    uint2 gl[8] = {};
    const unsigned char *pa = lds + (lane & 15) * 64 + (lane >> 4) * 16;
    unsigned char *pw = lds + 131072 + wave * 4096 + lane * 4;
    auto body = [&](int i) __attribute__((always_inline)) {
        int m = 0;
        // order: row block, A plane, replica, B plane; operands of a (row block, replica) pair are re-read right after its last use
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int q = 0; q + p < 4; ++q) {
                        c[(rb * 3 + rr) * 4 + p + q] =
                            __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rb][p], b[rr][q], c[(rb * 3 + rr) * 4 + p + q], 0, 0, 0);
                        if ((FLAGS & 1) && m % 4 == 3 && m / 4 < 28) { // 16 A + 12 B fragment reads per step, spread
                            const int k = m / 4;
                            if (k < 16)
                                a[(k / 4 + 3) & 3][k & 3] = *reinterpret_cast<const v4i *>(pa + (k & 3) * 16384 + ((i + 1) & 3) * 4096 + (k / 4) * 1024);
                            else
                                b[(k - 16) / 4][(k - 16) & 3] = *reinterpret_cast<const v4i *>(pa + 65536 + ((k - 16) & 3) * 1024 + ((i + 1) & 1) * 4096);
                        }
                        if ((FLAGS & 2) && !(FLAGS & 16) && m < 80) { // 160 VALU spread: 2 per slot
                            x[m % 8] = __builtin_amdgcn_perm(x[m % 8], x[(m + 3) % 8] + gl[m % 8].x, 0x05010400u);
                        }
                        if ((FLAGS & 16) && m < 80 && m % 4 == 0) { // the same 160 VALU in bursts: 8 every fourth slot
#pragma unroll
                            for (int z = 0; z < 4; ++z)
                                x[(m + z) % 8] = __builtin_amdgcn_perm(x[(m + z) % 8], x[(m + z + 3) % 8] + gl[(m + z) % 8].x, 0x05010400u);
                        }
                        if ((FLAGS & 32) && m < 80 && m % 4 == 0) { // bursts of 6 (120 VALU): what a conversion stage issues
#pragma unroll
                            for (int z = 0; z < 3; ++z)
                                x[(m + z) % 8] = __builtin_amdgcn_perm(x[(m + z) % 8], x[(m + z + 3) % 8] + gl[(m + z) % 8].x, 0x05010400u);
                        }
                        if ((FLAGS & 4) && m % 4 == 1 && m < 64)
                            *reinterpret_cast<uint32_t *>(pw + (m / 4) * 256 + ((i + 1) & 1) * 8192) = x[m % 8];
                        if ((FLAGS & 8) && m >= 90 && m < 98)
                            gl[m - 90] = g[(size_t)((i * 8 + (m - 90)) & 1023) * 4096 + blockIdx.x * 64 + lane];
                        ++m;
                        __builtin_amdgcn_sched_barrier(0);
                    }
    };
    for (int i = 0; i < iters; ++i)
        body(i);
    int r = 0;
    for (int t = 0; t < 48; ++t) for (int e = 0; e < 4; ++e) r += c[t][e];
    for (int t = 0; t < 8; ++t) r += (int)x[t];
    if (r == 0x12345678) out[threadIdx.x] = r;
}
template <int FLAGS> static void run_stepZ(int *dD, const uint2 *g, int cus)
{
    const int iters = 2000, blocks = cus;
    hipFuncSetAttribute((const void *)stepZ<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stepZ<FLAGS>, dim3(blocks), dim3(256), 160 * 1024, 0, dD, g, 10);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    hipEventRecord(e0); hipLaunchKernelGGL(stepZ<FLAGS>, dim3(blocks), dim3(256), 160 * 1024, 0, dD, g, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * iters * 120;
    printf("stepZ flags=%2d (16x16x64, 120 MFMAs = 60 big ones per step) 1 wave/SIMD: %.1f ns per step, %.0f TOPS  [%s]\n", FLAGS,
           ms * 1e6 / iters, mf * 32768 / (ms * 1e-3) * 1e-12, hipGetErrorString(err));
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int *dD; hipMalloc(&dD, 4096);
    const int cus = p.multiProcessorCount;
    if (getenv("PROBE_ALL")) {
    run_rate<4, false, 1>(dD, cus); run_rate<4, true, 1>(dD, cus); run_rate<8, true, 1>(dD, cus); run_rate<16, true, 1>(dD, cus);
    run_rate<4, false, 2>(dD, cus); run_rate<4, true, 2>(dD, cus); run_rate<8, true, 2>(dD, cus);
    }
    uint2 *g; hipMalloc(&g, (size_t)1024 * 4096 * 8 + 65536 * 8); hipMemset(g, 0x5a, (size_t)1024 * 4096 * 8);
    run_rate16<16, false>(dD, cus); run_rate16<16, true>(dD, cus); run_rate16<48, true>(dD, cus);
    run_rate16<24, true, 1>(dD, cus); run_rate16<24, false, 2>(dD, cus); run_rate16<24, true, 2>(dD, cus); // two waves per SIMD: mm_mfma_blk2_kernel's regime
    if (getenv("PROBE_RATE16_ONLY"))
        return 0;
    run_stepZ<1>(dD, g, cus); run_stepZ<3>(dD, g, cus); run_stepZ<17>(dD, g, cus); run_stepZ<33>(dD, g, cus); run_stepZ<15>(dD, g, cus); run_stepZ<29>(dD, g, cus);
    run_stepY<0>(dD, g, cus); run_stepY<1>(dD, g, cus); run_stepY<3>(dD, g, cus); run_stepY<7>(dD, g, cus); run_stepY<15>(dD, g, cus);
    return 0;
}
