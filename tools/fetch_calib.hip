// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of the mm kernels?  MI355X_MICROARCH.md
// calibrates FETCH_SIZE (x 2 on gfx950) on wide coalesced reads only and asks for a calibration on a known byte count in one's own
// pattern.  Each kernel below moves exactly 1 GiB once (no reuse; 4x the Infinity Cache would be needed to defeat it -- the guide says
// its hits are counted):
//   read_b128_rows   the f panel loads: 16 B per lane, a wave covers one 1 KiB row (mm_mfma_blk3_kernel's bgLoad)
//   read_b64_slab    the s slab loads: 8 B per lane, lanes 0-7 one 64-byte run of a row, the wave 8 rows 1 KiB apart (loadRound)
//   write_b32_tile   the r stores: 4 B per lane, 16 lanes one 64-byte run, the wave 4 rows 1 KiB apart (voteStore)
//   write_b128_rows  reference: 16 B per lane, coalesced
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out_f -o c -- ./fetch_calib ; rocprofv3 --pmc WRITE_SIZE -d out_w -o c -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr size_t kBytes = 1ull << 30;
constexpr int kRow = 1024; // bytes per matrix row (256 x uint32)

__global__ void read_b128_rows(const uint8_t *p, uint32_t *sink)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
    uint32_t acc = 0;
    for (size_t row = wave; row < kBytes / kRow; row += nwaves) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(p + row * kRow + lane * 16);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}
// 8 rows x 64 bytes per wave-instruction; a "slab" = 64 rows x 64 bytes: 8 instructions; consecutive column groups of 64 bytes
__global__ void read_b64_slab(const uint8_t *p, uint32_t *sink)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
    uint32_t acc = 0;
    // the buffer as 256 KiB matrices (256 rows x 1 KiB); unit = (matrix, 64-row k-slab, 64-byte column group): 4 KiB
    for (size_t unit = wave; unit < kBytes / 4096; unit += nwaves) {
        const size_t mat = unit / 64, sl = (unit / 16) % 4, cg = unit % 16;
        const uint8_t *base = p + mat * 262144 + sl * 64 * kRow + cg * 64;
        for (int r8 = 0; r8 < 8; ++r8) {
            const u32x2 v = *reinterpret_cast<const u32x2 *>(base + (size_t)(r8 * 8 + (lane >> 3)) * kRow + (lane & 7) * 8);
            acc ^= v[0] ^ v[1];
        }
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}
__global__ void write_b32_tile(uint8_t *p)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
    // unit = (matrix, 16-row block, 64-byte column group): 16 rows x 64 bytes = 1 KiB: 4 instructions of 4 rows
    for (size_t unit = wave; unit < kBytes / 1024; unit += nwaves) {
        const size_t mat = unit / 256, rb = (unit / 16) % 16, cg = unit % 16;
        uint8_t *base = p + mat * 262144 + rb * 16 * kRow + cg * 64;
        for (int i = 0; i < 4; ++i)
            __builtin_nontemporal_store((uint32_t)(unit + i), reinterpret_cast<uint32_t *>(base + (size_t)(4 * (lane >> 4) + i) * kRow + (lane & 15) * 4));
    }
}
__global__ void write_b128_rows(uint8_t *p)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
    for (size_t row = wave; row < kBytes / kRow; row += nwaves)
        *reinterpret_cast<u32x4 *>(p + row * kRow + lane * 16) = u32x4{(uint32_t)row, 1u, 2u, 3u};
}

int main()
{
    uint8_t *a, *b;
    uint32_t *sink;
    if (hipMalloc(&a, kBytes) != hipSuccess || hipMalloc(&b, kBytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess)
        return 1;
    (void)hipMemset(a, 1, kBytes);
    (void)hipMemset(b, 0, kBytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_b128_rows, dim3(2048), dim3(256), 0, 0, a, sink);
        hipLaunchKernelGGL(read_b64_slab, dim3(2048), dim3(256), 0, 0, a, sink);
        hipLaunchKernelGGL(write_b32_tile, dim3(2048), dim3(256), 0, 0, b);
        hipLaunchKernelGGL(write_b128_rows, dim3(2048), dim3(256), 0, 0, b);
    }
    const hipError_t e = hipDeviceSynchronize();
    printf("fetch_calib: every kernel moves %zu bytes once [%s]\n", kBytes, hipGetErrorString(e));
    return e != hipSuccess;
}
