/*
 * multi_gpu_c_demo.c -- the multi-GPU leg of the hot path from a plain C host: independent protected blocks sharded across
 * the GPUs of one node, and the ONE collective of the path -- the all-reduce of the fault counters over RCCL/xGMI, which
 * replaces the reference's single global `TMR_ERROR_CNT += 1` (projects/dataflowProtection/synchronization.cpp:1428-1431).
 *
 * One coast_ctx per GPU (here: one process driving all visible GPUs through ncclCommInitAll; a one-process-per-GPU host does
 * the same with ncclCommInitRank).  Every GPU runs crc16 (tests/crc16/crc16.c:21-31) in TMR over its own shard with its own
 * seeded single-bit upsets; after coast_allreduce_counters every context reports the job-wide corrected-fault count.
 *
 * Build: gcc -std=gnu11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../include multi_gpu_c_demo.c \
 *            -L../coast_amd/lib -lcoast_hip -L/opt/rocm/lib -lamdhip64 -lrccl -Wl,-rpath,'$ORIGIN/../coast_amd/lib' \
 *            -Wl,-rpath,/opt/rocm/lib -o multi_gpu_c_demo
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "coast_hip.h"

#define MAXDEV 8
#define BLOCK_LEN 255u        /* the reference's maximum: `unsigned char length` */
#define BLOCKS_PER_GPU 65536u
#define FAULTS_PER_GPU 100u

#define CK(call)                                                                         \
    do {                                                                                 \
        int rc__ = (int)(call);                                                          \
        if (rc__ != 0) {                                                                 \
            fprintf(stderr, "%s failed with %d (%s:%d)\n", #call, rc__, __FILE__, __LINE__); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static unsigned short crc16_ref(const unsigned char *p, unsigned len) /* crc16.c:21-31, for the self-check */
{
    unsigned char x;
    unsigned short crc = 0xFFFF;
    while (len--) {
        x = crc >> 8 ^ *p++;
        x ^= x >> 4;
        crc = (crc << 8) ^ ((unsigned short)(x << 12)) ^ ((unsigned short)(x << 5)) ^ ((unsigned short)x);
    }
    return crc;
}

int main(int argc, char **argv)
{
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    if (argc > 1 && atoi(argv[1]) > 0 && atoi(argv[1]) < ndev)
        ndev = atoi(argv[1]);
    if (ndev > MAXDEV)
        ndev = MAXDEV;
    if (ndev < 1) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    int devs[MAXDEV];
    ncclComm_t comm[MAXDEV];
    for (int d = 0; d < ndev; ++d)
        devs[d] = d;
    CK(ncclCommInitAll(comm, ndev, devs));

    coast_ctx *ctx[MAXDEV];
    unsigned char *d_data[MAXDEV];
    uint16_t *d_crc[MAXDEV];
    const size_t bytes = (size_t)BLOCKS_PER_GPU * BLOCK_LEN;
    unsigned char *h = (unsigned char *)malloc(bytes);
    const coast_cfg tmr = {3u, 0u, 0u};
    int errors = 0;

    for (int d = 0; d < ndev; ++d) { /* shard d: its own bytes, its own upsets; no data ever moves between GPUs */
        CK(hipSetDevice(d));
        CK(coast_create(&ctx[d], d));
        CK(hipMalloc((void **)&d_data[d], bytes));
        CK(hipMalloc((void **)&d_crc[d], BLOCKS_PER_GPU * sizeof(uint16_t)));
        uint32_t x = 0x9e3779b9u * (uint32_t)(d + 1);
        for (size_t i = 0; i < bytes; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = (unsigned char)(x >> 24);
        }
        CK(hipMemcpy(d_data[d], h, bytes, hipMemcpyHostToDevice));
        coast_fault fl[FAULTS_PER_GPU];
        for (unsigned q = 0; q < FAULTS_PER_GPU; ++q) {
            x = x * 1664525u + 1013904223u;
            fl[q].item = (uint64_t)q * (BLOCKS_PER_GPU / FAULTS_PER_GPU); /* distinct blocks */
            fl[q].step = (x >> 8) % (BLOCK_LEN + 1u);
            fl[q].replica = (uint8_t)((x >> 4) % 3u);
            fl[q].site = COAST_SITE_CRC_CRC;
            fl[q].bit = (uint8_t)(x & 15u); /* the crc register is 16 bits wide: every flip is live */
            fl[q].index = 0;
        }
        CK(coast_inject_faults(ctx[d], fl, FAULTS_PER_GPU));
        CK(coast_crc16_batch(ctx[d], d_data[d], BLOCK_LEN, BLOCKS_PER_GPU, d_crc[d], &tmr, NULL));
        /* spot-check this shard against the reference recurrence: every upset was out-voted */
        uint16_t got[4];
        CK(hipMemcpy(got, d_crc[d], sizeof got, hipMemcpyDeviceToHost));
        for (int b = 0; b < 4; ++b)
            errors += got[b] != crc16_ref(h + (size_t)b * BLOCK_LEN, BLOCK_LEN);
    }

    CK(ncclGroupStart()); /* one process, several communicators: the per-device calls form one group */
    for (int d = 0; d < ndev; ++d) {
        CK(hipSetDevice(d));
        CK(coast_allreduce_counters(ctx[d], comm[d]));
    }
    CK(ncclGroupEnd());

    for (int d = 0; d < ndev; ++d) {
        coast_stats st;
        CK(hipSetDevice(d));
        CK(coast_read_stats(ctx[d], &st));
        printf("gpu %d of %d: TMR_ERROR_CNT(global) = %llu  __SYNC_COUNT(global) = %llu\n", d, ndev,
               (unsigned long long)st.errors_corrected, (unsigned long long)st.sync_count);
        errors += st.errors_corrected != (uint64_t)FAULTS_PER_GPU * (uint64_t)ndev;
        errors += st.sync_count != (uint64_t)BLOCKS_PER_GPU * (uint64_t)ndev;
    }
    printf("C:0 E:%d F:%u T:0us\n", errors, FAULTS_PER_GPU * (unsigned)ndev);
    for (int d = 0; d < ndev; ++d) {
        CK(hipSetDevice(d));
        coast_destroy(ctx[d]);
        ncclCommDestroy(comm[d]);
        (void)hipFree(d_data[d]);
        (void)hipFree(d_crc[d]);
    }
    free(h);
    return errors;
}
