// mfma_probe3.hip -- how many co-resident waves does a SIMD need before v_mfma_i32_16x16x64_i8 is paced by the matrix pipe (16
// cycles per instruction) rather than by the issuing wave, and how much other work fits beside it?  Round-4 development tool for
// the mm headline kernel: rate per (waves per SIMD, VALU fillers per MFMA, LDS fragment reads per MFMA) on random operand bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_probe3 mfma_probe3.hip && ./mfma_probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rnd(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s ^ (s >> 13);
}

// One wave: NACC independent accumulators; per MFMA NF v_perm_b32 on a rotating register set; LD: one ds_read_b128 into an operand
// register every LD-th MFMA (0: none).  WPS waves per SIMD = 256 * WPS threads in the one workgroup of a CU.
template <int WPS, int NF, int LD, bool RND> __global__ __launch_bounds__(256 * WPS, 1) void probe(int *out, int iters)
{
    constexpr int NACC = 12;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    for (int i = threadIdx.x; i < 16384; i += 256 * WPS)
        reinterpret_cast<uint32_t *>(lds)[i] = RND ? rnd(s) : 0x01020304u;
    __syncthreads();
    v4i a[4], b[4];
    for (int p = 0; p < 4; ++p) {
        a[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){1, 2, 3, 4};
        b[p] = RND ? (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)} : (v4i){5, 6, 7, 8};
    }
    v4i c[NACC];
    for (int t = 0; t < NACC; ++t)
        c[t] = (v4i){0, 0, 0, 0};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t)
        x[t] = rnd(s);
    const int lane = threadIdx.x & 63;
    const unsigned char *pa = lds + (lane & 15) * 64 + ((lane >> 4) ^ ((lane >> 1) & 3)) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t & 3], b[(t >> 2) & 3], c[t], 0, 0, 0);
            if constexpr (LD != 0) {
                if (t % LD == LD - 1) {
                    // refresh the operand that was used longest ago
                    if ((t / LD) & 1)
                        a[(t + 2) & 3] = *reinterpret_cast<const v4i *>(pa + ((i * 3 + t) & 15) * 4096);
                    else
                        b[((t >> 2) + 2) & 3] = *reinterpret_cast<const v4i *>(pa + 1024 + ((i * 5 + t) & 15) * 4096);
                }
            }
#pragma unroll
            for (int z = 0; z < NF; ++z)
                x[(t * NF + z) % 8] = __builtin_amdgcn_perm(x[(t * NF + z) % 8], x[(t * NF + z + 3) % 8], 0x05010400u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int r = 0;
    for (int t = 0; t < NACC; ++t)
        for (int e = 0; e < 4; ++e)
            r += c[t][e];
    for (int t = 0; t < 8; ++t)
        r += (int)x[t];
    if (r == 0x12345678)
        out[threadIdx.x] = r;
}

template <int WPS, int NF, int LD, bool RND> static void run(int *dD, int cus)
{
    const int iters = 24000 / WPS, blocks = cus;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<WPS, NF, LD, RND>), dim3(blocks), dim3(256 * WPS), 65536, 0, dD, 10);
    hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<WPS, NF, LD, RND>), dim3(blocks), dim3(256 * WPS), 65536, 0, dD, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * WPS * iters * 12;
    printf("waves/SIMD=%d fillers/MFMA=%d lds-read every %d %s: %7.3f ms  %5.0f TOPS  [%s]\n", WPS, NF, LD, RND ? "random" : "const ", ms,
           mf * 32768 / (ms * 1e-3) * 1e-12, hipGetErrorString(err));
    fflush(stdout);
}

template <int WPS> static void sweep(int *dD, int cus)
{
    run<WPS, 0, 0, false>(dD, cus);
    run<WPS, 0, 0, true>(dD, cus);
    run<WPS, 1, 0, true>(dD, cus);
    run<WPS, 2, 0, true>(dD, cus);
    run<WPS, 3, 0, true>(dD, cus);
    run<WPS, 4, 0, true>(dD, cus);
    run<WPS, 0, 2, true>(dD, cus);
    run<WPS, 2, 2, true>(dD, cus);
    run<WPS, 3, 2, true>(dD, cus);
    run<WPS, 2, 3, true>(dD, cus);
    run<WPS, 2, 2, false>(dD, cus);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int *dD;
    hipMalloc(&dD, 4096);
    const int cus = p.multiProcessorCount;
    sweep<1>(dD, cus);
    sweep<2>(dD, cus);
    sweep<3>(dD, cus);
    sweep<4>(dD, cus);
    return 0;
}
