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
    return bad != 0;
}
