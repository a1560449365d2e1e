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

/* One protected run.  with_faults is a compile-time constant at both call sites (the timed path carries no fault checks).
 * A register upset (item = output element e, replica, site ACC / OPA / OPB, step k, bit) hits ONE clone's register -- its
 * `sum`, or the operand it just loaded -- exactly as in coast_oracle.c:mm_item; in this mode the clone's wrong sum is stored
 * into ITS copy of `results_matrix` unvoted (synchronization.cpp:211-215) and surfaces in checkGolden's return vote. */
static inline __attribute__((always_inline)) int tmr_mm_run(const uint32_t *f, const uint32_t *s, uint32_t *r, int n,
                                                            uint32_t xor_golden, uint32_t *cnt, uint64_t *syncs,
                                                            const int with_faults, const orc_fault *fl, size_t nf)
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
                uint32_t a0 = f0[(size_t)i0 * n + k0], b0 = s0[(size_t)k0 * n + j0];
                uint32_t a1 = f1[(size_t)i1 * n + k1], b1 = s1[(size_t)k1 * n + j1];
                uint32_t a2 = f2[(size_t)i2 * n + k2], b2 = s2[(size_t)k2 * n + j2];
                if (with_faults) {
                    const uint64_t e = (uint64_t)i0 * n + j0;
                    for (size_t q = 0; q < nf; ++q) {
                        if (fl[q].item != e || fl[q].step != (uint32_t)k0)
                            continue;
                        const uint32_t m = 1u << (fl[q].bit & 31);
                        unsigned long *sm = fl[q].replica == 0 ? &sum0 : fl[q].replica == 1 ? &sum1 : &sum2;
                        uint32_t *pa = fl[q].replica == 0 ? &a0 : fl[q].replica == 1 ? &a1 : &a2;
                        uint32_t *pb = fl[q].replica == 0 ? &b0 : fl[q].replica == 1 ? &b1 : &b2;
                        if (fl[q].site == ORC_SITE_MM_ACC)
                            *sm ^= m;
                        else if (fl[q].site == ORC_SITE_MM_OPA)
                            *pa ^= m;
                        else if (fl[q].site == ORC_SITE_MM_OPB)
                            *pb ^= m;
                    }
                }
                sum0 += (uint32_t)(a0 * b0);
                sum1 += (uint32_t)(a1 * b1);
                sum2 += (uint32_t)(a2 * b2);
                ++k0;
                ++k1;
                ++k2;
            }
            if (with_faults) {
                const uint64_t e = (uint64_t)i0 * n + j0;
                for (size_t q = 0; q < nf; ++q)
                    if (fl[q].item == e && fl[q].step == (uint32_t)n && fl[q].site == ORC_SITE_MM_ACC) {
                        unsigned long *sm = fl[q].replica == 0 ? &sum0 : fl[q].replica == 1 ? &sum1 : &sum2;
                        *sm ^= 1u << (fl[q].bit & 31);
                    }
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

int orc_cpu_tmr_mm(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, uint32_t xor_golden, uint32_t *cnt,
                   uint64_t *syncs)
{
    return tmr_mm_run(f, s, r, n, xor_golden, cnt, syncs, 0, NULL, 0);
}

/* Campaign on the default-mode restatement: `nruns` runs of the same n x n product, run b carrying the upsets whose item lies
 * in [b*n*n, (b+1)*n*n) (one per run = the reference campaign's regime, threadFunctions.py:588-600).  Outcome classes as
 * jsonParser.py:162-186: E (checkGolden's voted return) != 0 -> error; else TMR_ERROR_CNT > 0 -> fault corrected; else
 * success.  Returns the summed TMR_ERROR_CNT. */
uint64_t orc_cpu_tmr_mm_campaign(const uint32_t *f, const uint32_t *s, int n, uint32_t xor_golden, const orc_fault *faults,
                                 size_t nfaults, size_t nruns, uint64_t *n_error, uint64_t *n_corrected, uint64_t *n_success)
{
    const uint64_t nn = (uint64_t)n * n;
    uint32_t *r = (uint32_t *)malloc(nn * sizeof(uint32_t));
    orc_fault *mine = (orc_fault *)malloc((nfaults ? nfaults : 1) * sizeof(orc_fault));
    uint64_t total = 0, ne = 0, nc = 0, ns = 0;
    for (size_t b = 0; b < nruns; ++b) {
        size_t k = 0;
        for (size_t q = 0; q < nfaults; ++q)
            if (faults[q].item / nn == b && faults[q].replica < 3) {
                mine[k] = faults[q];
                mine[k].item = faults[q].item % nn;
                ++k;
            }
        uint32_t cnt = 0;
        uint64_t syncs = 0;
        const int err = tmr_mm_run(f, s, r, n, xor_golden, &cnt, &syncs, 1, mine, k);
        total += cnt;
        if (err)
            ++ne;
        else if (cnt)
            ++nc;
        else
            ++ns;
    }
    free(mine);
    free(r);
    if (n_error)
        *n_error = ne;
    if (n_corrected)
        *n_corrected = nc;
    if (n_success)
        *n_success = ns;
    return total;
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
