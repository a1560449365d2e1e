/*
 * dropin.c -- libcoast_dropin.so: the reference's own symbol names on top of libcoast_hip.so, so that the unmodified
 * tests/ drivers of the reference link against this backend instead of being pushed through `opt -TMR|-DWC`.
 * Host code stays C.  What it provides:
 *
 *   unsigned short crc16(const unsigned char*, unsigned char)                     tests/crc16/crc16.c:21
 *   void aes_enc_dec(unsigned char *state, unsigned char *key, unsigned char dir) tests/aes/TI_aes_128.h:42
 *   void sha256_hash(ctx_data, ctx_bitlen, ctx_state, data, len, hash)            tests/sha256_common/sha256_common_tmr.c:101
 *   int  coast_dropin_calc_sum(array, n)               target of cache_glue.c (calc_sum, tests/cache_test/cacheTest.c:101;
 *                                                      `data_array_elements` is a macro, :78)
 *   void coast_dropin_sha_stream(indata, in_i, vsize, block_size, digest)   target of chsha_glue.c (CHStone sha_stream,
 *                                                      tests/chstone/sha/sha.c:174-186; VSIZE / BLOCK_SIZE are macros)
 *   int  coast_dropin_chstone_aes(statemt, key, type, dir)   target of chaes_glue.c (CHStone encrypt / decrypt,
 *                                                      tests/chstone/aes/aes_enc.c:67, aes_dec.c:66)
 *   void coast_dropin_matrix_multiply(f, s, r, side)   target of the per-benchmark glue TU (matrix_multiply's `side`
 *                                                      is a macro, tests/mm_common/mm_tmr.c:10, so it is not in its ABI)
 *   TMR_ERROR_CNT, __SYNC_COUNT      the globals the pass emits (synchronization.cpp:269-294, :103-121); weak, because
 *                                    a program may define its own (tests/hifive1/sha256.tmr/sha256_tmr.c:20)
 *   FAULT_DETECTED_DWC()             default handler = abort() (synchronization.cpp:1251-1266); weak, because a program
 *                                    may define its own (tests/TMRregression/unitTests/stackProtect.c:67)
 *
 * The protection mode replaces the Makefile's OPT_PASSES (tests/crc16/Makefile:3, unittest/cfg/full.yml:18-36):
 *   COAST_OPT_PASSES = the reference's own flag string, e.g. "-TMR -countErrors", "-DWC -noMemReplication -noLoadSync":
 *       -TMR / -DWC / neither          3 / 2 / 1 replicas
 *       -noMemReplication              the lane-replicated engine; WITHOUT it (the reference's default) every clone runs on its
 *                                      own memory copy and the copies are voted at the region exit (COAST_F_HOST_MEMORY_REPLICATED)
 *       -noStoreDataSync               COAST_F_NO_STORE_DATA_SYNC (only meaningful next to -noMemReplication, as in the reference)
 *       -countErrors -countSyncs -storeDataSync -i -s   accepted; always on / no effect here (include/coast_hip.h says why; the
 *                                      batch ABI has the memory-replicated -storeDataSync form as COAST_F_MEMORY_COPIES)
 *       -noLoadSync -noStoreAddrSync   accepted; real knobs for matrix_multiply / crc16 / sha256_hash / calc_sum / aes_enc_dec / CHStone sha / aes once COAST_COUNTERS_IN_SOR=1 puts their loop
 *                                      counters inside the sphere of replication (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC, the
 *                                      reference's -noMemReplication rule set for them); otherwise no replicated address exists
 *   or, shorter, COAST_MODE = TMR (default) | DWC | NONE  (lane-replicated engine);
 *   COAST_SYNC_EVERY = V adds the optional loop-condition sync points.
 *
 * Fault injection into the unmodified program (what supervisor.py does through GDB, simulation/platform/
 * threadFunctions.py:588-600): COAST_INJECT="item:replica:site:step:bit[:index][,...]" arms those single-bit flips for the
 * COAST_INJECT_CALL-th protected call of the process (default 0, the first one).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "coast_hip.h"

__attribute__((weak)) uint32_t TMR_ERROR_CNT = 0;
__attribute__((weak)) uint64_t __SYNC_COUNT = 0;

__attribute__((weak, noinline)) void FAULT_DETECTED_DWC(void)
{
    abort();
}

static int has_flag(const char *passes, const char *flag)
{
    const size_t n = strlen(flag);
    for (const char *p = strstr(passes, flag); p; p = strstr(p + 1, flag))
        if ((p == passes || p[-1] == ' ') && (p[n] == '\0' || p[n] == ' '))
            return 1;
    return 0;
}

static coast_cfg dropin_cfg(void)
{
    coast_cfg c = {3u, 0u, 0u};
    const char *passes = getenv("COAST_OPT_PASSES");
    const char *m = getenv("COAST_MODE");
    if (passes) {
        c.replicas = has_flag(passes, "-TMR") ? 3u : has_flag(passes, "-DWC") ? 2u : 1u;
        if (!has_flag(passes, "-noMemReplication"))
            c.flags |= COAST_F_HOST_MEMORY_REPLICATED;
        else if (has_flag(passes, "-noStoreDataSync"))
            c.flags |= COAST_F_NO_STORE_DATA_SYNC;
    } else if (m) {
        if (!strcmp(m, "DWC") || !strcmp(m, "-DWC"))
            c.replicas = 2u;
        else if (!strcmp(m, "NONE") || !strcmp(m, ""))
            c.replicas = 1u;
    }
    const char *v = getenv("COAST_SYNC_EVERY");
    if (v)
        c.sync_every = (uint32_t)strtoul(v, NULL, 10);
    return c;
}

/* matrix_multiply / crc16 / sha256_hash / calc_sum / aes_enc_dec / CHStone sha and aes: the kernels whose loop counters can be put inside the sphere of replication */
static coast_cfg dropin_cfg_counters(void)
{
    coast_cfg c = dropin_cfg();
    const char *in = getenv("COAST_COUNTERS_IN_SOR");
    if (in && *in && *in != '0' && !(c.flags & COAST_F_HOST_MEMORY_REPLICATED) && c.replicas > 1u) {
        const char *passes = getenv("COAST_OPT_PASSES");
        c.flags |= COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC;
        c.sync_every = 0u; /* every loop condition is a sync point already: COAST_SYNC_EVERY has nothing to add (the batch entry points reject the pair) */
        if (passes && has_flag(passes, "-noLoadSync"))
            c.flags |= COAST_F_NO_LOAD_SYNC;
        if (passes && has_flag(passes, "-noStoreAddrSync"))
            c.flags |= COAST_F_NO_STORE_ADDR_SYNC;
        if (in[0] == '2') /* COAST_COUNTERS_IN_SOR=2: + the -O0 IR's stores into locals / in-place arrays as data votes (__SYNC_COUNT = the IR's branches + GEPs + stores) */
            c.flags |= COAST_F_LOCAL_STORE_SYNC;
    }
    return c;
}

/* arm the environment-described faults right before the chosen protected call */
static void dropin_maybe_inject(void)
{
    static long calls = 0;
    const char *spec = getenv("COAST_INJECT");
    const long mine = calls++;
    if (!spec || !*spec)
        return;
    const char *which = getenv("COAST_INJECT_CALL");
    if (mine != (which ? strtol(which, NULL, 10) : 0))
        return;
    coast_fault fl[64];
    size_t k = 0;
    const char *p = spec;
    while (*p && k < 64) {
        unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
        int nf = 0;
        char *end;
        for (;;) {
            v[nf++] = strtoull(p, &end, 0);
            p = end;
            if (*p != ':' || nf == 6)
                break;
            ++p;
        }
        if (nf >= 5) {
            fl[k].item = v[0];
            fl[k].replica = (uint8_t)v[1];
            fl[k].site = (uint8_t)v[2];
            fl[k].step = (uint32_t)v[3];
            fl[k].bit = (uint8_t)v[4];
            fl[k].index = (uint8_t)v[5];
            ++k;
        }
        while (*p && *p != ',')
            ++p;
        if (*p == ',')
            ++p;
    }
    if (k)
        (void)coast_host_inject_faults(fl, k);
}

static void dropin_fail(const char *what, int rc)
{
    if (rc == COAST_ETIMEOUT) /* the reference program would hang here and the supervisor would file a timeout */
        fprintf(stderr, "libcoast_dropin: %s did not terminate (watchdog / recursion stack limit)\n", what);
    else
        fprintf(stderr, "libcoast_dropin: %s failed with code %d (no GPU / no CPU fallback)\n", what, rc);
    abort();
}

/* fold the launch's counters into the globals the protected program exposes; DWC mismatch -> handler, no return */
static void dropin_account(void)
{
    coast_stats st;
    if (coast_host_stats(&st, 1) != 0)
        return;
    TMR_ERROR_CNT += (uint32_t)st.errors_corrected;
    __SYNC_COUNT += st.sync_count;
    if (st.dwc_detected)
        FAULT_DETECTED_DWC();
}

unsigned short crc16(const unsigned char *data_p, unsigned char length)
{
    const coast_cfg cfg = dropin_cfg_counters();
    dropin_maybe_inject();
    uint16_t crc = 0;
    const int rc = coast_crc16_host(data_p, length, &crc, &cfg);
    if (rc)
        dropin_fail("crc16", rc);
    dropin_account();
    return crc;
}

/* quick_sort(int *A, int len), tests/quicksort/quicksort.c:109: the whole recursive sort is one protected region */
void quick_sort(int *A, int len)
{
    const coast_cfg cfg = dropin_cfg_counters();
    coast_cfg q = cfg; /* the sort's indices are data: branch / address votes are its default, the -no...Sync flags its knobs */
    q.flags &= ~(uint32_t)(COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC | COAST_F_LOCAL_STORE_SYNC);
    const char *passes = getenv("COAST_OPT_PASSES");
    if (passes && !(q.flags & COAST_F_HOST_MEMORY_REPLICATED)) {
        if (has_flag(passes, "-noLoadSync"))
            q.flags |= COAST_F_NO_LOAD_SYNC;
        if (has_flag(passes, "-noStoreAddrSync"))
            q.flags |= COAST_F_NO_STORE_ADDR_SYNC;
    }
    dropin_maybe_inject();
    const int rc = coast_quicksort_host((int32_t *)A, len < 0 ? 0u : (uint32_t)len, &q);
    if (rc)
        dropin_fail("quick_sort", rc);
    dropin_account();
}

void aes_enc_dec(unsigned char *state, unsigned char *key, unsigned char dir)
{
    const coast_cfg cfg = dropin_cfg_counters(); /* COAST_COUNTERS_IN_SOR=1: `round` and `i` replica-private, their conditions and GEP offsets voted */
    dropin_maybe_inject();
    const int rc = coast_aes_enc_dec_host(state, key, dir, &cfg);
    if (rc)
        dropin_fail("aes_enc_dec", rc);
    dropin_account();
}

void sha256_hash(unsigned char ctx_data[], uint32_t ctx_bitlen[], uint32_t ctx_state[], unsigned char data[],
                 uint32_t len, unsigned char hash[])
{
    coast_cfg cfg = dropin_cfg_counters();
    {   /* COAST_SHA256_O0=1 (with COAST_COUNTERS_IN_SOR): the walk in the -O0 IR's shape, what tests/sha256_common/Makefile (no OPT_FLAGS)
         * hands the pass; default: the post--O3 shape of the hifive1 flow, which has no -O0 store census */
        const char *o0 = getenv("COAST_SHA256_O0");
        if (o0 && *o0 && *o0 != '0' && (cfg.flags & COAST_F_BRANCH_SYNC))
            cfg.flags |= COAST_F_O0_SHAPE;
        else
            cfg.flags &= ~(uint32_t)COAST_F_LOCAL_STORE_SYNC;
    }
    dropin_maybe_inject();
    uint32_t st[8];
    const int rc = coast_sha256_host(data, len, hash, st, &cfg);
    if (rc)
        dropin_fail("sha256_hash", rc);
    /* caller-visible scratch exactly as the reference leaves it: final state, the bit length as two u32
     * (DBL_INT_ADD, sha256_common_tmr.c:2-5) and the last padded block (:129-164) */
    if (ctx_state)
        memcpy(ctx_state, st, sizeof st);
    const uint64_t bits = (uint64_t)len * 8u;
    if (ctx_bitlen) {
        ctx_bitlen[0] = (uint32_t)bits;
        ctx_bitlen[1] = (uint32_t)(bits >> 32);
    }
    if (ctx_data) {
        const uint32_t rem = len & 63u;
        memset(ctx_data, 0, 64);
        if (rem < 56u) {
            memcpy(ctx_data, data + (len - rem), rem);
            ctx_data[rem] = 0x80;
        }
        for (int b = 0; b < 8; ++b)
            ctx_data[63 - b] = (unsigned char)(bits >> (8 * b));
    }
    dropin_account();
}

void coast_dropin_matrix_multiply(const void *f, const void *s, void *r, int side)
{
    coast_cfg cfg = dropin_cfg_counters(); /* COAST_COUNTERS_IN_SOR=1: i, j, k, sum replica-private, loop conditions + GEP offsets voted */
    if (cfg.flags & (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC))
        cfg.sync_every = 0u; /* every loop condition is a sync point already */
    {
        /* side 256 on the matrix cores: the staging loads are cloned and compared -- what the pass does to every load of the region
         * (cloning.cpp:2187-2209) and the library's default since ABI 8; COAST_CLONE_STAGING=0 opts out (COAST_F_SINGLE_STAGING) */
        const char *cl = getenv("COAST_CLONE_STAGING");
        if (cl && *cl == '0')
            cfg.flags |= COAST_F_SINGLE_STAGING;
    }
    dropin_maybe_inject();
    const int rc = coast_matrix_multiply_host((const uint32_t *)f, (const uint32_t *)s, (uint32_t *)r, side, &cfg);
    if (rc)
        dropin_fail("matrix_multiply", rc);
    dropin_account();
}

/* calc_sum (tests/cache_test/cacheTest.c:101-177).  The sum, the compare of every element with its index and the rewrite
 * run protected on the GPU; what calc_sum does around them with the program's own globals (error bookkeeping and the
 * YAML-ish report, robust_printing = 1) is replayed here from the array as it was found, so the program's output stays
 * byte-identical.  The globals are the benchmark's: weak references, absent in every other program this object links into. */
extern unsigned long int ind __attribute__((weak));
extern int local_errors __attribute__((weak));
extern int sum_errors __attribute__((weak));
extern int in_block __attribute__((weak));
extern int golden __attribute__((weak));

int coast_dropin_calc_sum(int *array, int n)
{
    const coast_cfg cfg = dropin_cfg_counters(); /* COAST_COUNTERS_IN_SOR=1: the loop counter i replica-private, its condition and GEP offsets voted */
    dropin_maybe_inject();
    int *found = (int *)malloc((size_t)n * sizeof(int));
    if (!found)
        dropin_fail("calc_sum", COAST_ENOMEM);
    memcpy(found, array, (size_t)n * sizeof(int));
    int32_t sum = 0;
    uint32_t nerr = 0;
    const int rc = coast_cache_test_host((int32_t *)array, (uint32_t)n, &sum, &nerr, &cfg);
    if (rc)
        dropin_fail("calc_sum", rc);
    dropin_account();
    int first_error = 0;
    if (nerr && &local_errors && &in_block && &ind) {
        for (int i = 0; i < n; ++i) {
            if (found[i] == i)
                continue;
            if (!first_error) {
                if (!in_block)
                    printf(" - i: %lu\r\n", ind);
                printf("   E: {%i: %i,", i, found[i]);
                first_error = 1;
                in_block = 1;
            } else {
                printf("%i: %i,", i, found[i]);
            }
            local_errors++;
        }
        if (first_error)
            printf("}\r\n");
    }
    if (&golden && &local_errors && &sum_errors && &in_block && &ind && sum != golden && local_errors == 0) {
        sum_errors++;
        local_errors++;
        if (!in_block)
            printf(" - i: %lu\r\n", ind);
        printf("   S: {%i: %i}\r\n", golden, sum);
        in_block = 1;
    }
    free(found);
    return sum;
}

/* sha_stream (tests/chstone/sha/sha.c:174-186): sha_init, one sha_update per input vector, sha_final -- a running hash over
 * the concatenation of the vectors' first in_i[j] bytes.  Every in_i[j] must be a multiple of 64 (the benchmark's are 8192):
 * the reference's sha_update keeps no partial block across calls and its sha_final pads only block-aligned totals. */
void coast_dropin_sha_stream(const unsigned char *indata, const int *in_i, int vsize, int block_size, unsigned int *digest)
{
    const coast_cfg cfg = dropin_cfg_counters(); /* COAST_COUNTERS_IN_SOR=1: sha_transform's / sha_update's loop counters inside the sphere of replication */
    size_t total = 0;
    for (int j = 0; j < vsize; ++j) {
        if (in_i[j] < 0 || in_i[j] > block_size || (in_i[j] & 63))
            dropin_fail("sha_stream (vector length not a multiple of 64)", COAST_EINVAL);
        total += (size_t)in_i[j];
    }
    unsigned char *buf = (unsigned char *)malloc(total ? total : 1);
    if (!buf)
        dropin_fail("sha_stream", COAST_ENOMEM);
    size_t off = 0;
    for (int j = 0; j < vsize; ++j) {
        memcpy(buf + off, indata + (size_t)j * (size_t)block_size, (size_t)in_i[j]);
        off += (size_t)in_i[j];
    }
    dropin_maybe_inject();
    uint32_t dg[5];
    const int rc = coast_chsha_host(buf, (uint32_t)total, dg, &cfg);
    free(buf);
    if (rc)
        dropin_fail("sha_stream", rc);
    for (int w = 0; w < 5; ++w)
        digest[w] = dg[w];
    dropin_account();
}

/* CHStone aes: encrypt / decrypt (tests/chstone/aes/aes_enc.c:67-134, aes_dec.c:66-140) keep the block and the key as one
 * byte per int and take the Rijndael size as `type` (key bits * 1000 + block bits; the benchmark's main() runs 128128,
 * aes.c:93-94).  All nine sizes go to the protected Rijndael kernel (coast_chaes_batch); the key array is left alone, as in the
 * reference (KeySchedule expands it into `word`).  Returns the state in the caller's int array; the glue prints and checks it
 * the way the two functions do. */
int coast_dropin_chstone_aes(int *statemt, const int *key, int type, int dir)
{
    const int kb = type / 1000, bb = type % 1000;
    if ((kb != 128 && kb != 192 && kb != 256) || (bb != 128 && bb != 192 && bb != 256))
        return -1; /* KeySchedule's default case (aes_key.c:132-133) */
    coast_cfg cfg = dropin_cfg_counters(); /* COAST_COUNTERS_IN_SOR=1: the round counter, the callees' j / i, switches, returns and GEP offsets voted; =2: the stored data too */
    unsigned char st[32], k[32];
    for (int i = 0; i < bb / 8; ++i)
        st[i] = (unsigned char)statemt[i];
    for (int i = 0; i < kb / 8; ++i)
        k[i] = (unsigned char)key[i];
    dropin_maybe_inject();
    const int rc = coast_chaes_host(st, k, type, dir, &cfg);
    if (rc)
        dropin_fail("CHStone aes", rc);
    for (int i = 0; i < bb / 8; ++i)
        statemt[i] = st[i];
    dropin_account();
    return 0;
}
