/*
 * cpu_tmr_baseline.c -- CPU-TMR timing baseline (test/bench infrastructure only, see coast_oracle.h).
 *
 * The real baseline -- tests/mm_common/mm_tmr.c pushed through `opt-7 -TMR -countErrors` -- cannot be built
 * here (LLVM 7.0 exact, projects/CMakeLists.txt:11).  This file restates what that pass emits for
 * matrix_multiply in COAST's DEFAULT mode (docs/source/passes.rst:329,337):
 *   - memory replicated x3: every protected global gets _DWC/_TMR copies (cloning.cpp:2442-2449), every load,
 *     multiply, add and store exists three times on the three copies (cloning.cpp:2187-2209);
 *   - stores are not voted (synchronization.cpp:211-215);
 *   - every conditional branch is a sync point: the i1 loop condition of the three copies is voted with
 *     select(a==b, a, c) (synchronization.cpp:146-155, 934-940) -> (n+1)(n^2+n+1) votes per call;
 *   - -countErrors: second compare, and TMR_ERROR_CNT += 1 when the copies disagree (synchronization.cpp:1391-1443);
 *   - checkGolden's return value is voted (ReturnInst sync).
 * The three copies are kept apart from the optimiser with empty asm barriers, the way three separately
 * allocated IR values stay apart after the pass has run behind -O3 (tests/pynq/matrixMultiply.tmr/Makefile:3).
 */
#include "coast_oracle.h"

#include <stdlib.h>
#include <string.h>

#define OPAQUE(x) __asm__ volatile("" : "+r"(x))

typedef struct {
    uint32_t cnt;
    uint64_t syncs;
} tmr_counters;

static inline int vote_cond(int a, int b, int c, tmr_counters *t)
{
    OPAQUE(a);
    OPAQUE(b);
    OPAQUE(c);
    t->syncs += 1;
    if (!((a == b) & (a == c)))
        t->cnt += 1;
    return (a == b) ? a : c;
}

int orc_cpu_tmr_mm(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, uint32_t xor_golden, uint32_t *cnt,
                   uint64_t *syncs)
{
    const size_t nn = (size_t)n * n;
    uint32_t *mem = (uint32_t *)malloc(9 * nn * sizeof(uint32_t));
    uint32_t *f0 = mem, *f1 = mem + nn, *f2 = mem + 2 * nn;
    uint32_t *s0 = mem + 3 * nn, *s1 = mem + 4 * nn, *s2 = mem + 5 * nn;
    uint32_t *r0 = mem + 6 * nn, *r1 = mem + 7 * nn, *r2 = mem + 8 * nn;
    tmr_counters t = {0, 0};
    /* addGlobalRuntimeInit-style copy of the initialisers into the clones (cloning.cpp:2543) */
    memcpy(f0, f, nn * 4);
    memcpy(f1, f, nn * 4);
    memcpy(f2, f, nn * 4);
    memcpy(s0, s, nn * 4);
    memcpy(s1, s, nn * 4);
    memcpy(s2, s, nn * 4);

    int i0 = 0, i1 = 0, i2 = 0;
    while (vote_cond(i0 < n, i1 < n, i2 < n, &t)) {
        int j0 = 0, j1 = 0, j2 = 0;
        while (vote_cond(j0 < n, j1 < n, j2 < n, &t)) {
            unsigned long sum0 = 0, sum1 = 0, sum2 = 0;
            int k0 = 0, k1 = 0, k2 = 0;
            while (vote_cond(k0 < n, k1 < n, k2 < n, &t)) {
                sum0 += (uint32_t)(f0[(size_t)i0 * n + k0] * s0[(size_t)k0 * n + j0]);
                sum1 += (uint32_t)(f1[(size_t)i1 * n + k1] * s1[(size_t)k1 * n + j1]);
                sum2 += (uint32_t)(f2[(size_t)i2 * n + k2] * s2[(size_t)k2 * n + j2]);
                ++k0;
                ++k1;
                ++k2;
            }
            r0[(size_t)i0 * n + j0] = (uint32_t)sum0;
            r1[(size_t)i1 * n + j1] = (uint32_t)sum1;
            r2[(size_t)i2 * n + j2] = (uint32_t)sum2;
            ++j0;
            ++j1;
            ++j2;
        }
        ++i0;
        ++i1;
        ++i2;
    }

    /* checkGolden (mm_common_tmr.c:22-32) on the three result copies; its loop conditions and return value vote */
    uint32_t x0 = 0, x1 = 0, x2 = 0;
    size_t e0 = 0, e1 = 0, e2 = 0;
    while (vote_cond(e0 < nn, e1 < nn, e2 < nn, &t)) {
        x0 ^= r0[e0++];
        x1 ^= r1[e1++];
        x2 ^= r2[e2++];
    }
    const int ret = vote_cond(x0 != xor_golden, x1 != xor_golden, x2 != xor_golden, &t);
    memcpy(r, r0, nn * 4);
    free(mem);
    if (cnt)
        *cnt = t.cnt;
    if (syncs)
        *syncs = t.syncs;
    return ret;
}

/* ---- all-host-cores variant: independent matrices, one per thread ---- */
#include <pthread.h>
#include <time.h>

typedef struct {
    const uint32_t *f, *s;
    int n, reps;
    uint32_t golden;
    int bad;
} mt_arg;

static void *mt_worker(void *p)
{
    mt_arg *a = (mt_arg *)p;
    uint32_t *r = (uint32_t *)malloc((size_t)a->n * a->n * sizeof(uint32_t));
    for (int i = 0; i < a->reps; ++i) {
        uint32_t cnt = 0;
        uint64_t syncs = 0;
        a->bad |= orc_cpu_tmr_mm(a->f, a->s, r, a->n, a->golden, &cnt, &syncs) | (cnt != 0);
    }
    free(r);
    return NULL;
}

/* runs nthreads x reps protected multiplications; returns wall seconds (< 0 on error) */
double orc_cpu_tmr_mm_threads(const uint32_t *f, const uint32_t *s, int n, uint32_t golden, int nthreads, int reps)
{
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    mt_arg *args = (mt_arg *)malloc(sizeof(mt_arg) * (size_t)nthreads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; ++t) {
        args[t] = (mt_arg){f, s, n, reps, golden, 0};
        pthread_create(&th[t], NULL, mt_worker, &args[t]);
    }
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        bad |= args[t].bad;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(args);
    if (bad)
        return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
