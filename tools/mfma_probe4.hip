// mfma_probe4.hip -- v_mfma_i32_32x32x32_i8 with ONE wave per SIMD against v_mfma_i32_16x16x64_i8 with one / two: the limb step of the
// TMR matrix multiply (three replicas x ten plane products into four accumulators per replica), random operand bytes, with NF VALU
// fillers per 16x16x64-equivalent of matrix work and NLD ds_read_b128 operand refreshes per ten MFMAs.  Round-4 development tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_probe4 mfma_probe4.hip && ./mfma_probe4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t rnd(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s ^ (s >> 13);
}

// BIG: 32x32x32 (NF is per 16x16x64-equivalent: 2 NF fillers per instruction); NT: independent tiles per replica (accumulators = NT x 3 x 4)
template <bool BIG, int NT, int WPS, int NF, int NLD> __global__ __launch_bounds__(256 * WPS, 1) void probe(int *out, int iters)
{
    using acc_t = std::conditional_t<BIG, v16i, v4i>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    for (int i = threadIdx.x; i < 16384; i += 256 * WPS)
        reinterpret_cast<uint32_t *>(lds)[i] = rnd(s);
    __syncthreads();
    v4i a[4], b[4];
    for (int p = 0; p < 4; ++p) {
        a[p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
        b[p] = (v4i){(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)};
    }
    acc_t c[NT][3][4];
    for (int t = 0; t < NT * 12; ++t)
        (&c[0][0][0])[t] = acc_t{};
    uint32_t x[8];
    for (int t = 0; t < 8; ++t)
        x[t] = rnd(s);
    const int lane = threadIdx.x & 63;
    const unsigned char *pa = lds + (lane & 15) * 64 + ((lane >> 4) ^ ((lane >> 1) & 3)) * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                int m = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 3 - p; q >= 0; --q, ++m) {
                        if constexpr (BIG)
                            c[tl][rr][p + q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[p], b[q], c[tl][rr][p + q], 0, 0, 0);
                        else
                            c[tl][rr][p + q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p], b[q], c[tl][rr][p + q], 0, 0, 0);
                        if (m < NLD) { // operand refresh: a[p] behind its last use (q == 0), b[] in the first products of a later plane
                            if (m < 4)
                                b[(m + 1) & 3] = *reinterpret_cast<const v4i *>(pa + 1024 + ((i * 5 + m + rr) & 15) * 4096);
                            else
                                a[(m - 4) & 3] = *reinterpret_cast<const v4i *>(pa + ((i * 3 + m + rr) & 15) * 4096);
                        }
#pragma unroll
                        for (int z = 0; z < NF * (BIG ? 2 : 1); ++z)
                            x[(m * 3 + z) % 8] = __builtin_amdgcn_perm(x[(m * 3 + z) % 8], x[(m * 3 + z + 3) % 8], 0x05010400u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
    }
    int r = 0;
    for (int t = 0; t < NT * 12; ++t)
        for (int e = 0; e < (BIG ? 16 : 4); ++e)
            r += (&c[0][0][0])[t][e];
    for (int t = 0; t < 8; ++t)
        r += (int)x[t];
    if (r == 0x12345678)
        out[threadIdx.x] = r;
}

template <bool BIG, int NT, int WPS, int NF, int NLD> static void run(int *dD, int cus)
{
    const int iters = 12000 / WPS / NT / (BIG ? 2 : 1);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<BIG, NT, WPS, NF, NLD>), dim3(cus), dim3(256 * WPS), 65536, 0, dD, 10);
    (void)hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<BIG, NT, WPS, NF, NLD>), dim3(cus), dim3(256 * WPS), 65536, 0, dD, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)cus * 4 * WPS * iters * NT * 30 * (BIG ? 32768.0 : 16384.0);
    printf("%s tiles=%d waves/SIMD=%d fillers per 16x16x64-equivalent=%d lds reads per ten MFMAs=%d: %7.3f ms %5.0f TOPS [%s]\n",
           BIG ? "32x32x32" : "16x16x64", NT, WPS, NF, NLD, ms, 2.0 * macs / (ms * 1e-3) * 1e-12, hipGetErrorString(err));
    fflush(stdout);
}

template <bool BIG, int NT, int WPS> static void sweep(int *dD, int cus)
{
    run<BIG, NT, WPS, 0, 0>(dD, cus);
    run<BIG, NT, WPS, 1, 0>(dD, cus);
    run<BIG, NT, WPS, 2, 0>(dD, cus);
    run<BIG, NT, WPS, 0, 4>(dD, cus);
    run<BIG, NT, WPS, 0, 8>(dD, cus);
    run<BIG, NT, WPS, 1, 8>(dD, cus);
    run<BIG, NT, WPS, 2, 8>(dD, cus);
    run<BIG, NT, WPS, 2, 4>(dD, cus);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    int *dD;
    (void)hipMalloc(&dD, 4096);
    const int cus = p.multiProcessorCount;
    sweep<true, 1, 1>(dD, cus);  // 192 accumulator registers: blk4's wave
    sweep<false, 2, 2>(dD, cus); // 96 accumulator registers, two waves per SIMD: blk2 / blk3's wave
    sweep<false, 4, 1>(dD, cus); // 192, one wave per SIMD: blk's wave
    sweep<true, 2, 1>(dD, cus);  // 384: does it fit at all
    return 0;
}
