// valu_microbench.hip -- issue-rate probe for the integer VALU instructions the protected kernels lean on (gfx950).
// Prints wave-instructions/s per CU-SIMD and the implied cycles per wave-instruction at the measured clock.
// Used to set the VALU roofline for the mm kernel (32-bit wrapping multiply has no MFMA form).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define KERNEL(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                 \
    {                                                                                                         \
        uint32_t a[CHAINS], b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;                     \
        for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;                                 \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                                     \
        uint32_t r = 0;                                                                                       \
        for (int i = 0; i < CHAINS; ++i) r ^= a[i];                                                           \
        if (r == 0x12345678u) out[threadIdx.x] = r;                                                           \
    }

KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]")
KERNEL(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(k_xor_lit, "v_xor_b32_e32 %0, 0x12345678, %0")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(k_lshr, "v_lshrrev_b32_e32 %0, 3, %0")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")

__global__ __launch_bounds__(256) void k_mad_u64_u32(uint32_t *out, uint32_t seed)
{
    unsigned long long a[CHAINS];
    uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9e3779b9u;
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long r = 0;
    for (int i = 0; i < CHAINS; ++i) r ^= a[i];
    if (r == 0x12345678ull) out[threadIdx.x] = (uint32_t)r;
}

__global__ __launch_bounds__(256) void k_bpermute(uint32_t *out, uint32_t seed)
{
    int a[CHAINS];
    const int addr = ((threadIdx.x & 63) / 3 * 3) * 4;
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = __builtin_amdgcn_ds_bpermute(addr, a[i]);
    }
    int r = 0;
    for (int i = 0; i < CHAINS; ++i) r ^= a[i];
    if (r == 0x12345678) out[threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_dpp_wave_shl(uint32_t *out, uint32_t seed)
{
    uint32_t a[CHAINS];
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + i * 77u + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    }
    uint32_t r = 0;
    for (int i = 0; i < CHAINS; ++i) r ^= a[i];
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

static int gBlocksPerCu = 8;
template <typename K> int run(const char *name, K kern, uint32_t *d, double extraPerIter = 1.0)
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * gBlocksPerCu; // N workgroups of 4 waves per CU = N waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waveInstr = (double)blocks * 4 * ITERS * CHAINS * extraPerIter;
    const double perSimdPerSec = waveInstr / (cus * 4) / (best * 1e-3);
    const double clk = p.clockRate * 1e3; // Hz
    printf("%-16s %8.3f ms  %7.2f Gwave-instr/s/SIMD  %6.2f cycles/wave-instr @%.0f MHz  lane-ops/s chip %.2f T\n", name, best,
           perSimdPerSec * 1e-9, clk / perSimdPerSec, clk * 1e-6, waveInstr * 64 / (best * 1e-3) * 1e-12);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1) gBlocksPerCu = atoi(argv[1]);
    printf("waves per SIMD: %d\n", gBlocksPerCu);
    uint32_t *d;
    CHECK(hipMalloc(&d, 4096));
    run("v_add_u32", k_add, d);
    run("v_xor_b32", k_xor, d);
    run("v_mul_lo_u32", k_mul_lo, d);
    run("v_mad_u64_u32", k_mad_u64_u32, d);
    run("v_mad_u32_u24", k_mad_u32_u24, d);
    run("v_mad_u32_u16", k_mad_u32_u16, d);
    run("v_pk_mad_u16", k_pk_mad_u16, d);
    run("v_dot4_u32_u8", k_dot4_u32_u8, d);
    run("v_alignbit_b32", k_alignbit, d);
    run("v_bfi_b32", k_bfi, d);
    run("v_add3_u32", k_add3, d);
    run("v_perm_b32", k_perm, d);
    run("v_lshl_add_u32", k_lshl_add, d);
    run("v_add_u32_e64", k_add_e64, d);
    run("v_xor_b32 literal", k_xor_lit, d);
    run("v_bitop3_b32", k_bitop3, d);
    run("v_lshrrev_b32", k_lshr, d);
    run("v_and_or_b32", k_and_or, d);
    run("ds_bpermute_b32", k_bpermute, d);
    run("dpp wave_shl:1", k_dpp_wave_shl, d);
    return 0;
}
