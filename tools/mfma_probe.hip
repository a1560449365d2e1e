// mfma_probe.hip -- layout and rate probe for v_mfma_i32_32x32x32_i8 on gfx950 (development tool).
//  (1) checks D[i][j] = sum_k A[i][k]*B[k][j] with the operand mapping  lane l: A row / B column = l & 31, k = 16*(l>>5)+byte
//      and the C/D mapping col = l & 31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5);
//  (2) measures the issue rate with 4 independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void one_tile(const int8_t *A, const int8_t *B, int *D) // A[32][32] row-major (i,k), B[32][32] (k,j)
{
    const int l = threadIdx.x;
    int8_t ab[16], bb[16];
    for (int b = 0; b < 16; ++b) {
        const int k = 16 * (l >> 5) + b;
        ab[b] = A[(l & 31) * 32 + k];
        bb[b] = B[k * 32 + (l & 31)];
    }
    v4i av, bv;
    __builtin_memcpy(&av, ab, 16);
    __builtin_memcpy(&bv, bb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[r];
    }
}

__global__ __launch_bounds__(256) void rate(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 0x12345678) out[threadIdx.x] = s;
}

// (3) do independent VALU instructions issue under a running MFMA?  20 MFMAs over 8 accumulators (the mm kernel's
//     pattern) + K v_perm_b32 per loop trip, 2 waves per SIMD.
template <int K> __global__ __launch_bounds__(256, 2) void mix(int *out, int iters)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    v16i c[8];
    for (int t = 0; t < 8; ++t) c[t] = (v16i){0};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t) x[t] = threadIdx.x * 2654435761u + t;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            c[m % 8] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[m % 8], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < K / 20; ++v) {
                const int t = (m * (K / 20) + v) % 8;
                x[t] = __builtin_amdgcn_perm(x[t], x[(t + 3) % 8], 0x05010400u + v);
            }
        }
    }
    int s = 0;
    for (int t = 0; t < 8; ++t) { for (int r = 0; r < 16; ++r) s += c[t][r]; s += (int)x[t]; }
    if (s == 0x12345678) out[threadIdx.x] = s;
}
template <int K> static void run_mix(int *dD, int cus)
{
    const int iters = 4000, blocks = cus * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix<K>, dim3(blocks), dim3(256), 0, 0, dD, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(mix<K>, dim3(blocks), dim3(256), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mix K=%3d VALU per 20 MFMA, 2 waves/SIMD: %.3f ms, %.1f ns per trip per wave-pair (MFMA-only floor = 2*20*32 cycles)\n", K, ms, ms * 1e6 / iters);
}

// (4) the mm step in isolation: per trip 20 MFMAs over 8 accumulators whose A/B operands come from 12 ds_read_b128
//     (FLAGS&1), 40 conversion VALU (FLAGS&2), 8 ds_write_b32 (FLAGS&4), 4 global dwordx2 loads (FLAGS&8); 2 waves/SIMD.
template <int FLAGS> __global__ __launch_bounds__(256, 2) void stepmix(int *out, const uint2 *g, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<int *>(lds)[i] = i * 7;
    __syncthreads();
    v4i a[2][4], b[4];
    for (int p = 0; p < 4; ++p) { a[0][p] = (v4i){lane, p, 3, 4}; a[1][p] = (v4i){p, lane, 3, 4}; b[p] = (v4i){5, 6, lane, p}; }
    v16i c[8];
    for (int t = 0; t < 8; ++t) c[t] = (v16i){0};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t) x[t] = threadIdx.x * 2654435761u + t;
    uint2 gl[4] = {};
    const unsigned char *pa = lds + (lane & 31) * 32 + ((lane >> 5) ^ ((lane >> 3) & 1)) * 16;
    unsigned char *pw = lds + 49152 + wave * 2560 + (lane % 40) * 32 + (lane / 40) * 4;
    for (int i = 0; i < iters; ++i) {
        if (FLAGS & 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                b[p] = *reinterpret_cast<const v4i *>(pa + 40960 + p * 320 + (i & 1) * 1280);
                a[0][p] = *reinterpret_cast<const v4i *>(pa + p * 8192 + (i & 7) * 1024);
                a[1][p] = *reinterpret_cast<const v4i *>(pa + p * 8192 + (i & 7) * 1024 + 512);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q + p < 4; ++q) {
                    c[rb * 4 + p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[rb][p], b[q], c[rb * 4 + p + q], 0, 0, 0);
                    if (FLAGS & 2) {
                        x[m % 8] = __builtin_amdgcn_perm(x[m % 8], x[(m + 3) % 8] + gl[m % 4].x, 0x05010400u);
                        x[(m + 1) % 8] = (x[(m + 1) % 8] + 0x80808080u) ^ gl[(m + 1) % 4].y;
                    }
                    if ((FLAGS & 4) && m % 2 == 1 && m < 16)
                        *reinterpret_cast<uint32_t *>(pw + (m / 2) * 320 + ((i + 1) & 1) * 1280) = x[m % 8];
                    if ((FLAGS & 8) && m >= 12 && m < 16)
                        gl[m - 12] = g[(size_t)((i * 4 + (m - 12)) & 1023) * 4096 + blockIdx.x * 64 + lane];
                    ++m;
                    __builtin_amdgcn_sched_barrier(0);
                }
    }
    int s = 0;
    for (int t = 0; t < 8; ++t) { for (int r = 0; r < 16; ++r) s += c[t][r]; s += (int)x[t]; }
    if (s == 0x12345678) out[threadIdx.x] = s;
}
template <int FLAGS> static void run_stepmix(int *dD, const uint2 *g, int cus)
{
    const int iters = 4000, blocks = cus * 2;
    hipFuncSetAttribute((const void *)stepmix<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stepmix<FLAGS>, dim3(blocks), dim3(256), 65536, 0, dD, g, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(stepmix<FLAGS>, dim3(blocks), dim3(256), 65536, 0, dD, g, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stepmix flags=%2d (1 frag reads, 2 VALU, 4 ds_write, 8 global loads): %.1f ns per trip per wave-pair\n", FLAGS, ms * 1e6 / iters);
}

int main()
{
    std::vector<int8_t> A(1024), B(1024);
    std::vector<int> D(1024), R(1024, 0);
    srand(1);
    for (auto &x : A) x = (int8_t)(rand() % 256 - 128);
    for (auto &x : B) x = (int8_t)(rand() % 256 - 128);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_tile, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int e = 0; e < 1024; ++e) bad += D[e] != R[e];
    printf("layout check: %d mismatches of 1024\n", bad);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    for (int wpb = 1; wpb <= 2; ++wpb) {
        const int iters = 20000, blocks = p.multiProcessorCount * wpb;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dD, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)blocks * 4 * iters * 4;
        printf("waves/SIMD %d: %.3f ms, %.1f cycles/MFMA/SIMD @2.4GHz, %.0f TOPS\n", wpb, ms, 2.4e9 * ms * 1e-3 / (mf / (p.multiProcessorCount * 4)), mf * 32768 * 2 / (ms * 1e-3) * 1e-12);
    }
    run_mix<0>(dD, p.multiProcessorCount); run_mix<40>(dD, p.multiProcessorCount); run_mix<80>(dD, p.multiProcessorCount);
    run_mix<120>(dD, p.multiProcessorCount); run_mix<160>(dD, p.multiProcessorCount); run_mix<240>(dD, p.multiProcessorCount);
    uint2 *g; hipMalloc(&g, (size_t)1024 * 4096 * 8 + 65536 * 8); hipMemset(g, 1, (size_t)1024 * 4096 * 8);
    run_stepmix<0>(dD, g, p.multiProcessorCount); run_stepmix<1>(dD, g, p.multiProcessorCount); run_stepmix<2>(dD, g, p.multiProcessorCount);
    run_stepmix<3>(dD, g, p.multiProcessorCount); run_stepmix<7>(dD, g, p.multiProcessorCount); run_stepmix<11>(dD, g, p.multiProcessorCount);
    run_stepmix<15>(dD, g, p.multiProcessorCount); run_stepmix<5>(dD, g, p.multiProcessorCount);
    return bad != 0;
}
