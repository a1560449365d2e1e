/*
 * coast_oracle.c -- CPU ORACLE (test infrastructure only, see coast_oracle.h).
 *
 * Restates, in plain C, (1) the arithmetic of the four COAST benchmark kernels and
 * (2) the replicate / vote / count semantics that the dataflowProtection pass gives them,
 * in the reference mode the GPU engine instantiates: `-TMR -noMemReplication -countErrors
 * -countSyncs` (single memory copy, every register value replicated, store data voted:
 * synchronization.cpp:197-224, cloning.cpp:90-95; docs/source/passes.rst:331) and `-DWC`.
 *
 * Sync-point schedules frozen here (SURVEY.md section 8 a'):
 *   mm     : r[i][j] voted once before the store (mm_common_tmr.c:16); with sync_every=V also the
 *            accumulator after every V-th k step (loop-condition sync, mm_common_tmr.c:12).
 *   sha256 : the 8 ctx_state words after every compression (sha256_common_tmr.c:90-97 are stores),
 *            the 8 digest words before they are written out (:169-178).
 *   aes    : the 4 state dwords and 4 round-key dwords at the end (in-place stores, TI_aes_128.c:145,
 *            211,228); with sync_every=1 also after every main-loop round.
 *   crc16  : the returned crc (crc16.c:30); with sync_every=V also crc after every V-th byte.
 * Wave-uniform control values (loop counters, lengths) are single-copy, i.e. __NO_xMR
 * (tests/COAST.h:11) -- they are outside the sphere of replication on the GPU and here.
 */
#include "coast_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* voter / comparator                                                                         */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    unsigned nrep;
    unsigned sync_every;
    orc_stats *st;
    int detected; /* unequal copies seen at a sync point of the current item (DWC: detected, TMR: corrected) */
    unsigned flags;
} sync_ctx;

/* One sync point on a 32-bit value.  TMR: synchronization.cpp:934-938 (cmp orig,clone1 ; select) and
 * :1391-1443 (second compare, AND, conditional TMR_ERROR_CNT+1); afterwards all three replicas continue
 * from the voted value (:527-529).  DWC: :1117-1192 -- a mismatch branches to the error block; here it is
 * recorded per item and the replicas keep their own values. */
static void sync32(sync_ctx *c, uint32_t v[3])
{
    if (c->nrep == 3) {
        const int e01 = (v[0] == v[1]);
        const int e02 = (v[0] == v[2]);
        c->st->sync_count += 1;
        if (!(e01 && e02)) {
            c->st->errors_corrected += 1;
            c->detected = 1; /* per-item flag: this item had a value corrected */
        }
        const uint32_t voted = e01 ? v[0] : v[2];
        v[0] = v[1] = v[2] = voted;
    } else if (c->nrep == 2) {
        c->st->sync_count += 1;
        if (v[0] != v[1])
            c->detected = 1;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* fault list handling                                                                        */
/* ------------------------------------------------------------------------------------------ */

static int fault_cmp(const void *a, const void *b)
{
    const orc_fault *x = (const orc_fault *)a, *y = (const orc_fault *)b;
    if (x->item != y->item)
        return x->item < y->item ? -1 : 1;
    if (x->step != y->step)
        return x->step < y->step ? -1 : 1;
    return 0;
}

static orc_fault *sorted_faults(const orc_fault *f, size_t n)
{
    orc_fault *s = (orc_fault *)malloc((n ? n : 1) * sizeof(orc_fault));
    if (n)
        memcpy(s, f, n * sizeof(orc_fault));
    qsort(s, n, sizeof(orc_fault), fault_cmp);
    return s;
}

/* first index with item >= key */
static size_t fault_lower(const orc_fault *s, size_t n, uint64_t key)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (s[mid].item < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

/* flipOneBit, injector.py:202-207: new = old XOR (1 << bit), on the 32-bit register; `mask` is the live width */
static inline uint32_t flip(uint32_t v, unsigned bit, uint32_t mask)
{
    return v ^ ((1u << (bit & 31)) & mask);
}

/* A sync point on the DATA of a store (synchronization.cpp:476-561): dropped by -noStoreDataSync (:197-224, :324) --
 * every copy keeps its own value and replica 0's is what reaches the single memory copy. */
static void store_sync32(sync_ctx *c, uint32_t v[3])
{
    if (!(c->flags & ORC_F_NO_STORE_DATA_SYNC))
        sync32(c, v);
}

/* ORC_F_LOCAL_STORE_SYNC: the data vote of a store into a local's alloca / into an array in place, on the -O0 IR (:197-224, 476-561).
 * TMR: the voted value is what reaches the single memory copy, every copy reloads it; DWC: compared, the copies keep their values. */
static void local_sync32(sync_ctx *c, uint32_t v[3])
{
    if ((c->flags & ORC_F_LOCAL_STORE_SYNC) && !(c->flags & ORC_F_NO_STORE_DATA_SYNC))
        sync32(c, v);
}

/* ------------------------------------------------------------------------------------------ */
/* matrix multiply                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* mm_common_tmr.c:3-20: r[i][j] = (mm_t) sum_k f[i][k]*s[k][j]; product is a 32-bit wrapping multiply, the
 * sum is truncated on the store, so only the low 32 bits ever matter. */
void orc_mm_plain(const uint32_t *f, const uint32_t *s, uint32_t *r, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            unsigned long sum = 0;
            for (int k = 0; k < n; ++k)
                sum += (uint32_t)(f[(size_t)i * n + k] * s[(size_t)k * n + j]);
            r[(size_t)i * n + j] = (uint32_t)sum;
        }
}

uint32_t orc_mm_xor(const uint32_t *r, int n)
{
    uint32_t x = 0;
    for (size_t e = 0; e < (size_t)n * n; ++e)
        x ^= r[e];
    return x;
}

/* one protected output element; fl[0..nf) are the faults of this item */
static uint32_t mm_item(const uint32_t *f, const uint32_t *s, int n, int i, int j, sync_ctx *c, const orc_fault *fl,
                        size_t nf)
{
    uint32_t acc[3] = {0, 0, 0};
    const unsigned R = c->nrep;
    for (int k = 0; k < n; ++k) {
        for (unsigned r = 0; r < R; ++r) {
            uint32_t a = f[(size_t)i * n + k]; /* loads repeated from the same address: cloning.cpp:2247-2255 */
            uint32_t b = s[(size_t)k * n + j];
            for (size_t q = 0; q < nf; ++q) {
                if (fl[q].step != (uint32_t)k || fl[q].replica != r)
                    continue;
                if (fl[q].site == ORC_SITE_MM_ACC)
                    acc[r] = flip(acc[r], fl[q].bit, 0xffffffffu);
                else if (fl[q].site == ORC_SITE_MM_OPA)
                    a = flip(a, fl[q].bit, 0xffffffffu);
                else if (fl[q].site == ORC_SITE_MM_OPB)
                    b = flip(b, fl[q].bit, 0xffffffffu);
            }
            acc[r] += a * b;
        }
        if (c->sync_every && ((k + 1) % (int)c->sync_every) == 0 && (k + 1) < n)
            sync32(c, acc);
    }
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].step == (uint32_t)n && fl[q].site == ORC_SITE_MM_ACC && fl[q].replica < R)
            acc[fl[q].replica] = flip(acc[fl[q].replica], fl[q].bit, 0xffffffffu);
    store_sync32(c, acc); /* store-data sync, synchronization.cpp:476-561 */
    return acc[0];
}

static uint32_t gep_offset(sync_ctx *c, const uint32_t idx[3], int synced);
static int branch_cond(sync_ctx *c, uint32_t c0, uint32_t c1, uint32_t c2, int synced);

/* matrix_multiply with its three loops as written (tests/mm_common/mm_common_tmr.c:3-20), for ORC_F_BRANCH_SYNC /
 * ORC_F_ADDR_SYNC: the work item is the CALL -- one matrix product -- and i, j, k and `sum` are replica-private registers of
 * one sequential walk, exactly the registers the pass triplicates (cloning.cpp:2187-2209).  Sync points, the reference's rule
 * set for -TMR -noMemReplication on the -O0 IR shape (the shape SURVEY.md section 3.2 derives the count from):
 *   the three loop conditions, at every evaluation: (N+1) + N (N+1) + N^2 (N+1) = (N+1)(N^2+N+1)    synchronization.cpp:146-155
 *   the GEP offsets: f[i][k] and s[k][j] are two GEPs each -- the row, then the element; syncGEP votes the LAST operand of a
 *     GEP (:413-420) -- i, k, k, j per MAC (loads: off with -noLoadSync); r[i][j]: i, j per element (off with -noStoreAddrSync)
 *   the data of the store r[i][j] = sum                                                             :197-224, 476-561
 * A load keeps the ORIGINAL instruction's address in every copy under -noMemReplication (cloning.cpp:2247-2255): unvoted
 * offsets are replica 0's.  Fault sites: ORC_SITE_MM_I / _J / _K / _ACC (= sum) of a replica, `step` = how many loop
 * conditions the call has evaluated (the flip lands right before the next one).  A wild index reads 0 / stores nothing; a
 * walk that a corrupted counter keeps alive is cut after 4 (N+1)(N^2+N+1) + 1024 conditions (the supervisor's timeout). */
static int mm_call_indexed(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    const unsigned R = c->nrep;
    const uint32_t N = (uint32_t)n;
    const int bs = (c->flags & ORC_F_BRANCH_SYNC) != 0, as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    const int ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC), ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    const uint64_t cap = 4ull * (N + 1ull) * ((uint64_t)N * N + N + 1ull) + 1024ull;
    uint32_t i[3] = {0, 0, 0}, j[3] = {0, 0, 0}, k[3] = {0, 0, 0}, sum[3] = {0, 0, 0};
    uint64_t tick = 0;
#define MM_HOOK()                                                                                              \
    do {                                                                                                       \
        for (size_t q_ = 0; q_ < nf; ++q_)                                                                     \
            if ((uint64_t)fl[q_].step == tick && fl[q_].replica < R) {                                         \
                uint32_t *t_ = fl[q_].site == ORC_SITE_MM_I ? i : fl[q_].site == ORC_SITE_MM_J ? j :           \
                               fl[q_].site == ORC_SITE_MM_K ? k : fl[q_].site == ORC_SITE_MM_ACC ? sum : NULL; \
                if (t_)                                                                                        \
                    t_[fl[q_].replica] = flip(t_[fl[q_].replica], fl[q_].bit, 0xffffffffu);                    \
            }                                                                                                  \
    } while (0)
#define MM_COND(reg) (tick++, branch_cond(c, reg[0] < N, reg[R > 1 ? 1 : 0] < N, reg[R > 2 ? 2 : 0] < N, bs))
    for (;;) {                                                   /* for (i = 0; i < side; i++)                 :10 */
        MM_HOOK();
        if (tick >= cap || !MM_COND(i))
            break;
        j[0] = j[1] = j[2] = 0;
        for (;;) {                                               /* for (j = 0; j < side; j++)                 :11 */
            MM_HOOK();
            if (tick >= cap || !MM_COND(j))
                break;
            sum[0] = sum[1] = sum[2] = 0;                        /* sum = 0                                    :12 */
            k[0] = k[1] = k[2] = 0;
            for (;;) {                                           /* for (k = 0; k < side; k++)                 :13 */
                MM_HOOK();
                if (tick >= cap || !MM_COND(k))
                    break;
                const uint32_t fi = gep_offset(c, i, ls), fk = gep_offset(c, k, ls);  /* f[i][k]                :14 */
                const uint32_t sk = gep_offset(c, k, ls), sj = gep_offset(c, j, ls);  /* s[k][j]                    */
                const uint32_t a = (fi < N && fk < N) ? f[(size_t)fi * N + fk] : 0u;
                const uint32_t b = (sk < N && sj < N) ? s[(size_t)sk * N + sj] : 0u;
                for (unsigned q = 0; q < 3; ++q)
                    sum[q] += a * b;
                local_sync32(c, sum);                            /* store sum (its alloca)                          */
                for (unsigned q = 0; q < 3; ++q)
                    k[q] += 1;
                local_sync32(c, k);                              /* k++                                             */
            }
            const uint32_t ri = gep_offset(c, i, ss), rj = gep_offset(c, j, ss);      /* r[i][j] = sum          :16 */
            uint32_t v[3] = {sum[0], sum[1], sum[2]};
            store_sync32(c, v);
            if (ri < N && rj < N)
                r[(size_t)ri * N + rj] = v[0];
            for (unsigned q = 0; q < 3; ++q)
                j[q] += 1;
            local_sync32(c, j);                                  /* j++                                             */
        }
        for (unsigned q = 0; q < 3; ++q)
            i[q] += 1;
        local_sync32(c, i);                                      /* i++                                             */
    }
#undef MM_HOOK
#undef MM_COND
    return tick >= cap;
}

void orc_mm_xmr(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, size_t batch, const orc_cfg *cfg,
                const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    const size_t nn = (size_t)n * n;
    size_t fp = 0;
    if (cfg->flags & ORC_F_INDEXED) { /* the loop counters inside the sphere of replication: one work item per matrix */
        for (size_t b = 0; b < batch; ++b) {
            while (fp < nfaults && fs[fp].item < (uint64_t)b * nn)
                ++fp;
            size_t fe = fp;
            while (fe < nfaults && fs[fe].item < (uint64_t)(b + 1) * nn)
                ++fe;
            memset(r + b * nn, 0, nn * sizeof(uint32_t)); /* elements a derailed walk never stores stay 0 */
            c.detected = 0;
            mm_call_indexed(f + b * nn, s + b * nn, r + b * nn, n, &c, fs + fp, fe - fp);
            if (c.detected) {
                st->dwc_detected += (cfg->replicas == 2);
                if (detected)
                    detected[b * nn] = 1; /* the item is the call: its flag is the matrix's first byte */
            }
            fp = fe;
        }
        free(fs);
        return;
    }
    for (size_t b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const uint64_t item = (uint64_t)b * nn + (uint64_t)i * n + j;
                while (fp < nfaults && fs[fp].item < item)
                    ++fp;
                size_t fe = fp;
                while (fe < nfaults && fs[fe].item == item)
                    ++fe;
                c.detected = 0;
                r[item] = mm_item(f + b * nn, s + b * nn, n, i, j, &c, fs + fp, fe - fp);
                if (c.detected) {
                    st->dwc_detected += (cfg->replicas == 2);
                    if (detected)
                        detected[item] = 1;
                }
                fp = fe;
            }
    free(fs);
}

void orc_mm_xmr_items(const uint32_t *f, const uint32_t *s, int n, const uint64_t *items, size_t nitems,
                      uint32_t *out, const orc_cfg *cfg, const orc_fault *faults, size_t nfaults, orc_stats *st,
                      uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    const size_t nn = (size_t)n * n;
    for (size_t q = 0; q < nitems; ++q) {
        const uint64_t item = items[q];
        const size_t b = item / nn;
        const int i = (int)((item % nn) / n), j = (int)(item % n);
        size_t fp = fault_lower(fs, nfaults, item), fe = fp;
        while (fe < nfaults && fs[fe].item == item)
            ++fe;
        c.detected = 0;
        out[q] = mm_item(f + b * nn, s + b * nn, n, i, j, &c, fs + fp, fe - fp);
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[q] = 1;
        }
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* sha256                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static const uint32_t SHA_K[64] = { /* FIPS 180-4 round constants; sha256_common_tmr.c:8-19 */
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

static const uint32_t SHA_IV[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                                   0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};

static inline uint32_t rotr(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }

/* sha256_transform (sha256_common_tmr.c:27-98) on nrep register copies; `blk` is the single memory copy of the
 * 64-byte block.  `cidx` numbers the compressions of this message for the fault steps. */
static void sha_compress3(uint32_t st[3][8], const uint8_t *const blk3[3], unsigned nrep, uint32_t cidx, const orc_fault *fl,
                          size_t nf)
{
    for (unsigned r = 0; r < nrep; ++r) {
        const uint8_t *blk = blk3[r]; /* memory-copies mode: replica r loads from its own copy */
        uint32_t m[64], v[8];
        for (unsigned t = 0; t < 64; ++t) {
            if (t < 16) {
                m[t] = ((uint32_t)blk[4 * t] << 24) | ((uint32_t)blk[4 * t + 1] << 16) |
                       ((uint32_t)blk[4 * t + 2] << 8) | (uint32_t)blk[4 * t + 3];
            } else {
                const uint32_t x = m[t - 2], y = m[t - 15];
                const uint32_t s1 = rotr(x, 17) ^ rotr(x, 19) ^ (x >> 10);
                const uint32_t s0 = rotr(y, 7) ^ rotr(y, 18) ^ (y >> 3);
                m[t] = s1 + m[t - 7] + s0 + m[t - 16];
            }
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_SHA_M && fl[q].replica == r && fl[q].step == cidx * 64 + t)
                    m[t] = flip(m[t], fl[q].bit, 0xffffffffu);
        }
        for (unsigned w = 0; w < 8; ++w)
            v[w] = st[r][w];
        for (unsigned t = 0; t < 64; ++t) {
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_SHA_WV && fl[q].replica == r && fl[q].step == cidx * 64 + t)
                    v[fl[q].index & 7] = flip(v[fl[q].index & 7], fl[q].bit, 0xffffffffu);
            const uint32_t a = v[0], b = v[1], cc = v[2], e = v[4], ff = v[5], g = v[6];
            const uint32_t ep0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            const uint32_t ep1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            const uint32_t ch = (e & ff) ^ (~e & g);
            const uint32_t maj = (a & b) ^ (a & cc) ^ (b & cc);
            const uint32_t t1 = v[7] + ep1 + ch + SHA_K[t] + m[t];
            const uint32_t t2 = ep0 + maj;
            v[7] = g;
            v[6] = ff;
            v[5] = e;
            v[4] = v[3] + t1;
            v[3] = cc;
            v[2] = b;
            v[1] = a;
            v[0] = t1 + t2;
        }
        for (unsigned w = 0; w < 8; ++w)
            st[r][w] += v[w];
    }
}

static void sha_compress(uint32_t st[3][8], const uint8_t *blk, unsigned nrep, uint32_t cidx, const orc_fault *fl, size_t nf)
{
    const uint8_t *const b3[3] = {blk, blk, blk};
    sha_compress3(st, b3, nrep, cidx, fl, nf);
}

static void sha_state_faults(uint32_t st[3][8], unsigned nrep, uint32_t cidx, const orc_fault *fl, size_t nf)
{
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].site == ORC_SITE_SHA_STATE && fl[q].step == cidx && fl[q].replica < nrep)
            st[fl[q].replica][fl[q].index & 7] = flip(st[fl[q].replica][fl[q].index & 7], fl[q].bit, 0xffffffffu);
}

static void sha_sync_state(sync_ctx *c, uint32_t st[3][8])
{
    for (unsigned w = 0; w < 8; ++w) {
        uint32_t v[3] = {st[0][w], st[1][w], st[2][w]};
        store_sync32(c, v); /* ctx_state[w] / hash[] are memory stores */
        st[0][w] = v[0];
        st[1][w] = v[1];
        st[2][w] = v[2];
    }
}

/* sha256_hash (sha256_common_tmr.c:100-179): one-shot hash of `len` bytes incl. padding; the bit length is
 * kept as two u32 exactly like DBL_INT_ADD (:2-5). */
/* data3[r] / hash3[r]: replica r's memory copy of the message / of the digest (all three the same pointer in the single-copy
 * mode).  COAST_F_MEMORY_COPIES -- the reference's memory-replicated mode with -storeDataSync (dataflowProtection.cpp:14-18,
 * synchronization.cpp:197-224): every clone loads from its own copy, the data of every store is voted, and every clone stores the
 * voted value into its own copy. */
static void sha_item3(const uint8_t *const data3[3], uint32_t len, uint8_t *const hash3[3], sync_ctx *c, const orc_fault *fl,
                      size_t nf)
{
    uint32_t st[3][8];
    uint8_t buf[3][64];
    const uint8_t *const b3[3] = {buf[0], buf[1], buf[2]};
    uint32_t bitlen[2] = {0, 0};
    uint32_t datalen = 0, cidx = 0;
    const unsigned R = c->nrep;
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned w = 0; w < 8; ++w)
            st[r][w] = SHA_IV[w];

    for (uint32_t i = 0; i < len; ++i) {
        for (unsigned r = 0; r < 3; ++r)
            buf[r][datalen] = data3[r][i];
        ++datalen;
        if (datalen == 64) {
            sha_state_faults(st, R, cidx, fl, nf);
            sha_compress3(st, b3, R, cidx, fl, nf);
            sha_sync_state(c, st);
            ++cidx;
            if (bitlen[0] > 0xffffffffu - 512u)
                ++bitlen[1];
            bitlen[0] += 512u;
            datalen = 0;
        }
    }
    uint32_t i = datalen;
    for (unsigned r = 0; r < 3; ++r)
        buf[r][i] = 0x80;
    ++i;
    if (datalen < 56) {
        for (unsigned r = 0; r < 3; ++r)
            memset(buf[r] + i, 0, 56 - i);
    } else {
        for (unsigned r = 0; r < 3; ++r)
            memset(buf[r] + i, 0, 64 - i);
        sha_state_faults(st, R, cidx, fl, nf);
        sha_compress3(st, b3, R, cidx, fl, nf);
        sha_sync_state(c, st);
        ++cidx;
        for (unsigned r = 0; r < 3; ++r)
            memset(buf[r], 0, 56);
    }
    if (bitlen[0] > 0xffffffffu - datalen * 8u)
        ++bitlen[1];
    bitlen[0] += datalen * 8u;
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned b = 0; b < 4; ++b) {
            buf[r][63 - b] = (uint8_t)(bitlen[0] >> (8 * b));
            buf[r][59 - b] = (uint8_t)(bitlen[1] >> (8 * b));
        }
    sha_state_faults(st, R, cidx, fl, nf);
    sha_compress3(st, b3, R, cidx, fl, nf);
    sha_sync_state(c, st);
    ++cidx;

    sha_state_faults(st, R, cidx, fl, nf); /* step == ncompress: between the last compression and the digest */
    sha_sync_state(c, st);                 /* digest words voted before the store, :169-178 */
    for (unsigned r = 0; r < 3; ++r)
        if (r == 0 || (r < R && hash3[r] != hash3[0]))
            for (unsigned w = 0; w < 8; ++w)
                for (unsigned b = 0; b < 4; ++b)
                    hash3[r][4 * w + b] = (uint8_t)(st[r][w] >> (24 - 8 * b));
}

static void sha_item(const uint8_t *data, uint32_t len, uint8_t hash[32], sync_ctx *c, const orc_fault *fl, size_t nf)
{
    const uint8_t *const d3[3] = {data, data, data};
    uint8_t *const h3[3] = {hash, hash, hash};
    sha_item3(d3, len, h3, c, fl, nf);
}

/* A sync point on a GEP offset (syncGEP, synchronization.cpp:413-474): the voted value replaces the offset operand of the
 * GEP in all copies (:456-458) -- the index REGISTERS keep their own values (the next `++` works on them).  Returns the
 * offset the single memory access uses: the voted one, or replica 0's (the original instruction's operand) when this class
 * of address is not synchronised. */
static uint32_t gep_offset(sync_ctx *c, const uint32_t idx[3], int synced)
{
    uint32_t v[3] = {idx[0], idx[1], idx[2]};
    if (synced)
        sync32(c, v);
    return v[0];
}

/* A sync point on a conditional branch (syncTerminator, :741-949): the i1 condition of every copy is voted and all copies
 * take the voted direction; not synchronised, the branch is the original instruction's. */
static int branch_cond(sync_ctx *c, uint32_t c0, uint32_t c1, uint32_t c2, int synced)
{
    uint32_t v[3] = {c0, c1, c2};
    if (synced)
        sync32(c, v);
    return (int)v[0];
}

/* sha256_hash with its byte loop as written (:119-127), for ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC: the loop counter `i` and
 * `ctx_datalen` are replica-private registers; ctx_data[64] and ctx_bitlen[] are memory (single copy, -noMemReplication).
 * Shape after -O3 (tests/hifive1/sha256.tmr/Makefile:4 runs it before -TMR): the byte loop and its two branches survive,
 * the padding loops are memsets (library calls, not replicated: functions.config:12) and the output loop is unrolled.
 * Accesses outside the message / outside ctx_data[64] are bounded here (a wild index reads 0 / stores nothing) and a loop that
 * a corrupted counter keeps alive is cut by a watchdog after 4 * len + 256 iterations -- on the reference those are wild
 * accesses and a supervisor timeout (jsonParser.py:162-186). */
static void sha_item_indexed(const uint8_t *data, uint32_t len, uint8_t hash[32], sync_ctx *c, const orc_fault *fl, size_t nf)
{
    uint32_t st[3][8], ir[3] = {0, 0, 0}, dl[3] = {0, 0, 0};
    uint8_t buf[64];
    uint32_t bitlen[2] = {0, 0};
    uint32_t cidx = 0, it = 0;
    const unsigned R = c->nrep;
    const int bs = (c->flags & ORC_F_BRANCH_SYNC) != 0, as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    const int ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC), ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    const uint64_t cap = 4ull * len + 256ull;
    memset(buf, 0, sizeof buf);
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned w = 0; w < 8; ++w)
            st[r][w] = SHA_IV[w];
    for (;; ++it) {
        for (size_t q = 0; q < nf; ++q)
            if (fl[q].step == it && fl[q].replica < R) {
                if (fl[q].site == ORC_SITE_SHA_I)
                    ir[fl[q].replica] = flip(ir[fl[q].replica], fl[q].bit, 0xffffffffu);
                else if (fl[q].site == ORC_SITE_SHA_DATALEN)
                    dl[fl[q].replica] = flip(dl[fl[q].replica], fl[q].bit, 0xffffffffu);
            }
        if (!branch_cond(c, ir[0] < len, ir[R > 1 ? 1 : 0] < len, ir[R > 2 ? 2 : 0] < len, bs) || it >= cap)
            break;                                            /* for (i = 0; i < len; ++i)                    :119 */
        const uint32_t li = gep_offset(c, ir, ls);            /* data[i]                                       :120 */
        const uint8_t byte = li < len ? data[li] : 0;
        const uint32_t si = gep_offset(c, dl, ss);            /* ctx_data[ctx_datalen] = ...                   :120 */
        if (si < 64)
            buf[si] = byte;
        for (unsigned r = 0; r < 3; ++r)
            dl[r] += 1;                                       /* ctx_datalen++                                 :121 */
        if (branch_cond(c, dl[0] == 64, dl[R > 1 ? 1 : 0] == 64, dl[R > 2 ? 2 : 0] == 64, bs)) { /*           :122 */
            sha_state_faults(st, R, cidx, fl, nf);
            sha_compress(st, buf, R, cidx, fl, nf);
            sha_sync_state(c, st);
            ++cidx;
            if (bitlen[0] > 0xffffffffu - 512u)
                ++bitlen[1];
            bitlen[0] += 512u;
            dl[0] = dl[1] = dl[2] = 0;                        /* ctx_datalen = 0                               :125 */
        }
        for (unsigned r = 0; r < 3; ++r)
            ir[r] += 1;
    }
    const int shortPad = branch_cond(c, dl[0] < 56, dl[R > 1 ? 1 : 0] < 56, dl[R > 2 ? 2 : 0] < 56, bs); /*     :132 */
    const uint32_t pi = gep_offset(c, dl, ss);                /* ctx_data[i++] = 0x80 with i = ctx_datalen  :133,137 */
    if (pi < 64)
        buf[pi] = 0x80;
    for (uint32_t k = pi + 1; k < (shortPad ? 56u : 64u); ++k) /* while (i < 56 / 64) ctx_data[i++] = 0: a memset */
        buf[k] = 0;
    if (!shortPad) {
        sha_state_faults(st, R, cidx, fl, nf);
        sha_compress(st, buf, R, cidx, fl, nf);
        sha_sync_state(c, st);
        ++cidx;
        memset(buf, 0, 56);
    }
    uint32_t add[3] = {dl[0] * 8u, dl[1] * 8u, dl[2] * 8u};   /* DBL_INT_ADD(..., ctx_datalen * 8): a store of a    */
    store_sync32(c, add);                                     /* replicated value into ctx_bitlen[0]            :150 */
    if (bitlen[0] > 0xffffffffu - add[0])
        ++bitlen[1];
    bitlen[0] += add[0];
    for (unsigned b = 0; b < 4; ++b) {
        buf[63 - b] = (uint8_t)(bitlen[0] >> (8 * b));
        buf[59 - b] = (uint8_t)(bitlen[1] >> (8 * b));
    }
    sha_state_faults(st, R, cidx, fl, nf);
    sha_compress(st, buf, R, cidx, fl, nf);
    sha_sync_state(c, st);
    ++cidx;
    sha_state_faults(st, R, cidx, fl, nf);
    sha_sync_state(c, st);
    for (unsigned w = 0; w < 8; ++w)
        for (unsigned b = 0; b < 4; ++b)
            hash[4 * w + b] = (uint8_t)(st[0][w] >> (24 - 8 * b));
}

/* sha256_hash + sha256_transform in the shape the x86 / lli flow gives the pass (tests/sha256_common/Makefile: OPT_FLAGS empty, i.e. the
 * -O0 IR of sha256_common_tmr.c:27-178), for ORC_F_BRANCH_SYNC | ORC_F_ADDR_SYNC | ORC_F_O0_SHAPE: every loop is a loop -- the byte
 * loop, `while (i < 56 / 64)`, the long pad's `while (n--)`, the output loop, the three loops of sha256_transform -- with replica-private
 * counters (i, ctx_datalen, n; the transform's i, j), every evaluated condition a branch vote, every variable-index GEP an offset vote by
 * the class of its first user (data[j..j+3], m[i-2], m[i-15], m[i-7], m[i-16], k[i], m[i], data[i]: loads; m[i], ctx_data[ctx_datalen],
 * ctx_data[i++], hash[i + 4 w]: stores).  The data votes of the default schedule stay: ctx_state[w] += (8 per transform), the bit count's
 * `a += ctx_datalen * 8`, the digest (8 words).  ORC_F_LOCAL_STORE_SYNC adds every other store of a computed value of the -O0 IR (1944
 * into locals + 64 m[i] per transform; the byte stores of ctx_data[] / hash[], the counters) -- the digest then leaves as its 32 byte
 * stores.  Fault sites as in the default schedule (SHA_STATE / _M / _WV) and the byte loop's SHA_I / SHA_DATALEN; the transform's own
 * counters have no site: their votes always agree.  A wild index reads 0 / stores nothing; the byte loop has the watchdog of
 * sha_item_indexed, the padding loops are cut after 256 iterations. */
#define O0_ALL(dst, expr)                                                                                      \
    do {                                                                                                       \
        for (unsigned r = 0; r < 3; ++r)                                                                       \
            (dst)[r] = (expr);                                                                                 \
    } while (0)
typedef struct {
    sync_ctx *c;
    int bs, ls, ss;
    unsigned R;
    const orc_fault *fl;
    size_t nf;
} sh0;
static int sh0_br(sh0 *m, const uint32_t cnd[3]) { return branch_cond(m->c, cnd[0], cnd[m->R > 1 ? 1 : 0], cnd[m->R > 2 ? 2 : 0], m->bs); }
static uint32_t sh0_off(sh0 *m, const uint32_t idx[3], int store) { return gep_offset(m->c, idx, store ? m->ss : m->ls); }

static void sh0_transform(sh0 *m, uint32_t st[3][8], const uint8_t buf[64], uint32_t cidx)
{
    sync_ctx *c = m->c;
    uint32_t W[3][64], i[3] = {0, 0, 0}, j[3] = {0, 0, 0}, temp[3], s[3], sig0[3], sig1[3], idx[3], cnd[3];
    memset(W, 0, sizeof W);
#define O0_M_FAULT(t)                                                                                          \
    for (size_t q_ = 0; q_ < m->nf; ++q_)                                                                      \
        if (m->fl[q_].site == ORC_SITE_SHA_M && m->fl[q_].replica < m->R && m->fl[q_].step == cidx * 64 + (t)) \
            W[m->fl[q_].replica][t] = flip(W[m->fl[q_].replica][t], m->fl[q_].bit, 0xffffffffu)
    for (;;) {                                                       /* for (i = 0, j = 0; i < 16; ++i, j += 4)   :34 */
        O0_ALL(cnd, i[r] < 16u);
        if (!sh0_br(m, cnd))
            break;
        for (unsigned b = 0; b < 4; ++b) {                           /*   temp = data[j] << 24; temp |= ..        :35-38 */
            O0_ALL(idx, j[r] + b);
            const uint32_t o = sh0_off(m, idx, 0);
            const uint32_t byte = o < 64u ? buf[o] : 0u;
            O0_ALL(temp, (b ? temp[r] : 0u) | (byte << (24 - 8 * b)));
            local_sync32(c, temp);
        }
        const uint32_t os = sh0_off(m, i, 1);                        /*   m[i] = temp                              :39 */
        local_sync32(c, temp);
        if (os < 64u) {
            for (unsigned r = 0; r < 3; ++r)
                W[r][os] = temp[r];
            O0_M_FAULT(os);
        }
        O0_ALL(i, i[r] + 1u);
        local_sync32(c, i);
        O0_ALL(j, j[r] + 4u);
        local_sync32(c, j);
    }
    for (;;) {                                                       /* for (; i < 64; ++i)                       :42 */
        O0_ALL(cnd, i[r] < 64u);
        if (!sh0_br(m, cnd))
            break;
        uint32_t o;
        O0_ALL(idx, i[r] - 2u);
        o = sh0_off(m, idx, 0);
        O0_ALL(s, o < 64u ? W[r][o] : 0u);                           /*   s = m[i - 2]                            :43 */
        local_sync32(c, s);
        O0_ALL(sig1, rotr(s[r], 17));
        local_sync32(c, sig1);
        O0_ALL(sig1, sig1[r] ^ rotr(s[r], 19));
        local_sync32(c, sig1);
        O0_ALL(sig1, sig1[r] ^ (s[r] >> 10));
        local_sync32(c, sig1);
        O0_ALL(idx, i[r] - 15u);
        o = sh0_off(m, idx, 0);
        O0_ALL(s, o < 64u ? W[r][o] : 0u);                           /*   s = m[i - 15]                           :48 */
        local_sync32(c, s);
        O0_ALL(sig0, rotr(s[r], 7));
        local_sync32(c, sig0);
        O0_ALL(sig0, sig0[r] ^ rotr(s[r], 18));
        local_sync32(c, sig0);
        O0_ALL(sig0, sig0[r] ^ (s[r] >> 3));
        local_sync32(c, sig0);
        O0_ALL(temp, sig1[r]);                                       /*   temp = sig1; += m[i-7]; += sig0; += m[i-16]  :53-56 */
        local_sync32(c, temp);
        O0_ALL(idx, i[r] - 7u);
        o = sh0_off(m, idx, 0);
        O0_ALL(temp, temp[r] + (o < 64u ? W[r][o] : 0u));
        local_sync32(c, temp);
        O0_ALL(temp, temp[r] + sig0[r]);
        local_sync32(c, temp);
        O0_ALL(idx, i[r] - 16u);
        o = sh0_off(m, idx, 0);
        O0_ALL(temp, temp[r] + (o < 64u ? W[r][o] : 0u));
        local_sync32(c, temp);
        const uint32_t os = sh0_off(m, i, 1);                        /*   m[i] = temp                              :57 */
        local_sync32(c, temp);
        if (os < 64u) {
            for (unsigned r = 0; r < 3; ++r)
                W[r][os] = temp[r];
            O0_M_FAULT(os);
        }
        O0_ALL(i, i[r] + 1u);
        local_sync32(c, i);
    }
#undef O0_M_FAULT
    uint32_t v[8][3];
    for (unsigned w = 0; w < 8; ++w) {                               /* a = ctx_state[0] ..                       :60-67 */
        O0_ALL(v[w], st[r][w]);
        local_sync32(c, v[w]);
    }
    O0_ALL(i, 0u);
    for (uint32_t t = 0;; ++t) {                                     /* for (i = 0; i < 64; ++i)                  :69 */
        O0_ALL(cnd, i[r] < 64u);
        if (!sh0_br(m, cnd) || t >= 64u)
            break;
        for (size_t q = 0; q < m->nf; ++q)
            if (m->fl[q].site == ORC_SITE_SHA_WV && m->fl[q].replica < m->R && m->fl[q].step == cidx * 64 + t)
                v[m->fl[q].index & 7][m->fl[q].replica] = flip(v[m->fl[q].index & 7][m->fl[q].replica], m->fl[q].bit, 0xffffffffu);
        uint32_t ep0[3], ep1[3], ch[3], maj[3], t1[3], t2[3], x[3];
        O0_ALL(ep0, rotr(v[0][r], 2));
        local_sync32(c, ep0);
        O0_ALL(ep0, ep0[r] ^ rotr(v[0][r], 13));
        local_sync32(c, ep0);
        O0_ALL(ep0, ep0[r] ^ rotr(v[0][r], 22));
        local_sync32(c, ep0);
        O0_ALL(ep1, rotr(v[4][r], 6));
        local_sync32(c, ep1);
        O0_ALL(ep1, ep1[r] ^ rotr(v[4][r], 11));
        local_sync32(c, ep1);
        O0_ALL(ep1, ep1[r] ^ rotr(v[4][r], 25));
        local_sync32(c, ep1);
        O0_ALL(ch, (v[4][r] & v[5][r]) ^ (~v[4][r] & v[6][r]));
        local_sync32(c, ch);
        O0_ALL(maj, (v[0][r] & v[1][r]) ^ (v[0][r] & v[2][r]) ^ (v[1][r] & v[2][r]));
        local_sync32(c, maj);
        const uint32_t ok = sh0_off(m, i, 0), om = sh0_off(m, i, 0); /*   k[i], m[i]                               :78 */
        O0_ALL(t1, v[7][r] + ep1[r] + ch[r] + (ok < 64u ? SHA_K[ok] : 0u) + (om < 64u ? W[r][om] : 0u));
        local_sync32(c, t1);
        O0_ALL(t2, ep0[r] + maj[r]);
        local_sync32(c, t2);
#define O0_MOVE(dst, expr)                                                                                     \
    do {                                                                                                       \
        O0_ALL(x, (expr));                                                                                     \
        local_sync32(c, x);                                                                                    \
        O0_ALL(v[dst], x[r]);                                                                                  \
    } while (0)
        O0_MOVE(7, v[6][r]);                                         /*   h = g .. a = t1 + t2                    :80-87 */
        O0_MOVE(6, v[5][r]);
        O0_MOVE(5, v[4][r]);
        O0_MOVE(4, v[3][r] + t1[r]);
        O0_MOVE(3, v[2][r]);
        O0_MOVE(2, v[1][r]);
        O0_MOVE(1, v[0][r]);
        O0_MOVE(0, t1[r] + t2[r]);
#undef O0_MOVE
        O0_ALL(i, i[r] + 1u);
        local_sync32(c, i);
    }
    for (unsigned w = 0; w < 8; ++w) {                               /* ctx_state[w] += ..: stored                :90-97 */
        uint32_t x[3];
        O0_ALL(x, st[r][w] + v[w][r]);
        store_sync32(c, x);
        for (unsigned r = 0; r < 3; ++r)
            st[r][w] = x[r];
    }
}

static void sha_item_o0(const uint8_t *data, uint32_t len, uint8_t hash[32], sync_ctx *c, const orc_fault *fl, size_t nf)
{
    sh0 mm = {c, (c->flags & ORC_F_BRANCH_SYNC) != 0, 0, 0, c->nrep, fl, nf}, *m = &mm;
    const int as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    m->ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC);
    m->ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    const int lss = (c->flags & ORC_F_LOCAL_STORE_SYNC) && !(c->flags & ORC_F_NO_STORE_DATA_SYNC);
    uint32_t st[3][8], ir[3] = {0, 0, 0}, dl[3] = {0, 0, 0}, cnd[3], x[3];
    uint8_t buf[64];
    uint32_t bitlen[2] = {0, 0};
    uint32_t cidx = 0, it = 0;
    const unsigned R = c->nrep;
    const uint64_t cap = 4ull * len + 256ull;
    memset(buf, 0, sizeof buf);
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned w = 0; w < 8; ++w)
            st[r][w] = SHA_IV[w];
    O0_ALL(x, len);
    local_sync32(c, x);                                        /* the parameter `len` into its alloca */
#define O0_BITLEN_ADD(cv)                                      /* DBL_INT_ADD(ctx_bitlen[0], ctx_bitlen[1], cv) :2-5 */ \
    do {                                                                                                       \
        const uint32_t carry_ = bitlen[0] > 0xffffffffu - (cv);                                                \
        O0_ALL(cnd, carry_);                                                                                   \
        if (sh0_br(m, cnd)) {                                                                                  \
            ++bitlen[1];                                                                                       \
            O0_ALL(x, bitlen[1]);                                                                              \
            local_sync32(c, x);                                                                                \
        }                                                                                                      \
        bitlen[0] += (cv);                                                                                     \
    } while (0)
    for (;; ++it) {
        for (size_t q = 0; q < nf; ++q)
            if (fl[q].step == it && fl[q].replica < R) {
                if (fl[q].site == ORC_SITE_SHA_I)
                    ir[fl[q].replica] = flip(ir[fl[q].replica], fl[q].bit, 0xffffffffu);
                else if (fl[q].site == ORC_SITE_SHA_DATALEN)
                    dl[fl[q].replica] = flip(dl[fl[q].replica], fl[q].bit, 0xffffffffu);
            }
        O0_ALL(cnd, ir[r] < len);
        if (!sh0_br(m, cnd) || it >= cap)
            break;                                             /* for (i = 0; i < len; ++i)                    :119 */
        const uint32_t li = sh0_off(m, ir, 0);                 /* data[i]                                       :120 */
        const uint8_t byte = li < len ? data[li] : 0;
        const uint32_t si = sh0_off(m, dl, 1);                 /* ctx_data[ctx_datalen] = ...                   :120 */
        O0_ALL(x, byte);
        local_sync32(c, x);
        if (si < 64)
            buf[si] = (uint8_t)x[0];
        O0_ALL(dl, dl[r] + 1u);                                /* ctx_datalen++                                 :121 */
        local_sync32(c, dl);
        O0_ALL(cnd, dl[r] == 64u);
        if (sh0_br(m, cnd)) {                                  /* if (ctx_datalen == 64)                        :122 */
            sha_state_faults(st, R, cidx, fl, nf);
            sh0_transform(m, st, buf, cidx);
            ++cidx;
            O0_BITLEN_ADD(512u);
            O0_ALL(x, bitlen[0]);
            local_sync32(c, x);                                /* the store of a += c                                */
            O0_ALL(dl, 0u);                                    /* ctx_datalen = 0                               :125 */
        }
        O0_ALL(ir, ir[r] + 1u);
        local_sync32(c, ir);
    }
    O0_ALL(ir, dl[r]);                                         /* i = ctx_datalen                               :129 */
    local_sync32(c, ir);
    O0_ALL(cnd, dl[r] < 56u);
    const int shortPad = sh0_br(m, cnd);                       /* if (ctx_datalen < 56)                         :132 */
    {
        const uint32_t lim = shortPad ? 56u : 64u;
        uint32_t o = sh0_off(m, ir, 1);                        /* ctx_data[i++] = 0x80                       :133,137 */
        if (o < 64)
            buf[o] = 0x80;
        O0_ALL(ir, ir[r] + 1u);
        local_sync32(c, ir);
        for (uint32_t guard = 0;; ++guard) {                   /* while (i < 56 / 64) ctx_data[i++] = 0x00   :134,138 */
            O0_ALL(cnd, ir[r] < lim);
            if (!sh0_br(m, cnd) || guard >= 256u)
                break;
            o = sh0_off(m, ir, 1);
            if (o < 64)
                buf[o] = 0;
            O0_ALL(ir, ir[r] + 1u);
            local_sync32(c, ir);
        }
    }
    if (!shortPad) {
        sha_state_faults(st, R, cidx, fl, nf);
        sh0_transform(m, st, buf, cidx);
        ++cidx;
        uint32_t n[3] = {56u, 56u, 56u};                       /* the inlined sha_memset(ctx_data, 0, 56)    :143-150 */
        O0_ALL(x, 0u);
        local_sync32(c, x);                                    /* c = c & 0xFF                                       */
        for (uint32_t p = 0;; ++p) {
            const uint32_t old[3] = {n[0], n[1], n[2]};
            O0_ALL(n, n[r] - 1u);                              /* while (n--): load, decrement, store, branch        */
            local_sync32(c, n);
            O0_ALL(cnd, old[r] != 0u);
            if (!sh0_br(m, cnd) || p >= 256u)
                break;
            O0_ALL(x, 0u);
            local_sync32(c, x);                                /* *p++ = c: a constant-offset GEP, the stored c       */
            if (p < 64)
                buf[p] = 0;
        }
    }
    uint32_t add[3] = {dl[0] * 8u, dl[1] * 8u, dl[2] * 8u};    /* DBL_INT_ADD(..., ctx_datalen * 8)             :150 */
    {
        const uint32_t carry = bitlen[0] > 0xffffffffu - add[0];
        O0_ALL(cnd, bitlen[0] > 0xffffffffu - add[r]);
        (void)carry;
        if (sh0_br(m, cnd)) {
            ++bitlen[1];
            O0_ALL(x, bitlen[1]);
            local_sync32(c, x);
        }
    }
    O0_ALL(add, bitlen[0] + add[r]);                           /* a += c: the store of a replicated value (the default schedule's vote) */
    store_sync32(c, add);
    bitlen[0] = add[0];
    for (unsigned b = 0; b < 4; ++b) {                         /* ctx_data[63 - b] = ctx_bitlen[0] >> 8 b ..  :151-158 */
        O0_ALL(x, (bitlen[0] >> (8 * b)) & 0xffu);
        local_sync32(c, x);
        buf[63 - b] = (uint8_t)x[0];
    }
    for (unsigned b = 0; b < 4; ++b) {
        O0_ALL(x, (bitlen[1] >> (8 * b)) & 0xffu);
        local_sync32(c, x);
        buf[59 - b] = (uint8_t)x[0];
    }
    sha_state_faults(st, R, cidx, fl, nf);
    sh0_transform(m, st, buf, cidx);
    ++cidx;
    sha_state_faults(st, R, cidx, fl, nf);
    if (!lss)
        sha_sync_state(c, st);                                 /* the digest: 8 words (the default schedule's exit votes) */
    memset(hash, 0, 32);
    O0_ALL(ir, 0u);
    for (uint32_t guard = 0;; ++guard) {                       /* for (i = 0; i < 4; ++i) hash[i + 4 w] = ..  :164-173 */
        O0_ALL(cnd, ir[r] < 4u);
        if (!sh0_br(m, cnd) || guard >= 4u)
            break;
        for (unsigned w = 0; w < 8; ++w) {
            uint32_t idx[3];
            O0_ALL(idx, ir[r] + 4u * w);
            const uint32_t o = sh0_off(m, idx, 1);
            O0_ALL(x, (st[r][w] >> ((24u - ir[r] * 8u) & 31u)) & 0xffu);
            local_sync32(c, x);                                /* (ORC_F_LOCAL_STORE_SYNC: the digest's 32 byte stores) */
            if (o < 32)
                hash[o] = (uint8_t)x[0];
        }
        O0_ALL(ir, ir[r] + 1u);
        local_sync32(c, ir);
    }
#undef O0_BITLEN_ADD
}
#undef O0_ALL

void orc_sha256_plain(const uint8_t *data, uint32_t len, uint8_t hash[32])
{
    orc_stats st = {0, 0, 0, 0};
    sync_ctx c = {1, 0, &st, 0, 0};
    sha_item(data, len, hash, &c, NULL, 0);
}

void orc_sha256_xmr(const uint8_t *msgs, size_t stride, uint32_t len, size_t nmsgs, uint8_t *digests,
                    const orc_cfg *cfg, const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t m = 0; m < nmsgs; ++m) {
        while (fp < nfaults && fs[fp].item < m)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == m)
            ++fe;
        c.detected = 0;
        if ((cfg->flags & ORC_F_INDEXED) && (cfg->flags & ORC_F_O0_SHAPE)) {
            sha_item_o0(msgs + m * stride, len, digests + 32 * m, &c, fs + fp, fe - fp);
        } else if (cfg->flags & ORC_F_INDEXED) {
            sha_item_indexed(msgs + m * stride, len, digests + 32 * m, &c, fs + fp, fe - fp);
        } else if (cfg->flags & ORC_F_MEMORY_COPIES) { /* the arrays are `replicas` copies back to back */
            const size_t cin = nmsgs * stride, cout = nmsgs * 32;
            const unsigned R = cfg->replicas;
            const uint8_t *const d3[3] = {msgs + m * stride, msgs + (R > 1 ? cin : 0) + m * stride, msgs + (R > 2 ? 2 * cin : 0) + m * stride};
            uint8_t *const h3[3] = {digests + 32 * m, digests + (R > 1 ? cout : 0) + 32 * m, digests + (R > 2 ? 2 * cout : 0) + 32 * m};
            sha_item3(d3, len, h3, &c, fs + fp, fe - fp);
        } else {
            sha_item(msgs + m * stride, len, digests + 32 * m, &c, fs + fp, fe - fp);
        }
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[m] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* aes-128                                                                                    */
/* ------------------------------------------------------------------------------------------ */

static uint8_t AES_S[256], AES_RS[256];
static int aes_tables_ready = 0;

/* FIPS-197 S-box computed from its definition (GF(2^8) inverse + affine map) instead of a pasted table;
 * tests/ check it against the reference's sbox/rsbox (TI_aes_128.c:44,64) through oracle/_ref. */
static uint8_t gf_mul(uint8_t a, uint8_t b)
{
    uint8_t p = 0;
    for (int i = 0; i < 8; ++i) {
        if (b & 1)
            p ^= a;
        const uint8_t hi = a & 0x80;
        a = (uint8_t)(a << 1);
        if (hi)
            a ^= 0x1b;
        b >>= 1;
    }
    return p;
}

static void aes_tables(void)
{
    if (aes_tables_ready)
        return;
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x)
            for (int y = 1; y < 256; ++y)
                if (gf_mul((uint8_t)x, (uint8_t)y) == 1) {
                    inv = (uint8_t)y;
                    break;
                }
        uint8_t s = inv;
        for (int k = 1; k <= 4; ++k)
            s ^= (uint8_t)((inv << k) | (inv >> (8 - k)));
        s ^= 0x63;
        AES_S[x] = s;
        AES_RS[s] = (uint8_t)x;
    }
    aes_tables_ready = 1;
}

const uint8_t *orc_aes_sbox(void)
{
    aes_tables();
    return AES_S;
}
const uint8_t *orc_aes_rsbox(void)
{
    aes_tables();
    return AES_RS;
}

static const uint8_t AES_RCON[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36};

static inline uint8_t xtime(uint8_t v) { return (uint8_t)((v << 1) ^ ((v & 0x80) ? 0x1b : 0)); } /* galois_mul2 :88-99 */

static void aes_key_fwd(uint8_t k[16], int rd) /* TI_aes_128.c:115-121, 220-226 */
{
    k[0] ^= AES_S[k[13]] ^ AES_RCON[rd];
    k[1] ^= AES_S[k[14]];
    k[2] ^= AES_S[k[15]];
    k[3] ^= AES_S[k[12]];
    for (int i = 4; i < 16; ++i)
        k[i] ^= k[i - 4];
}

static void aes_key_inv(uint8_t k[16], int rd) /* :134-141 */
{
    for (int i = 15; i > 3; --i)
        k[i] ^= k[i - 4];
    k[0] ^= AES_S[k[13]] ^ AES_RCON[rd];
    k[1] ^= AES_S[k[14]];
    k[2] ^= AES_S[k[15]];
    k[3] ^= AES_S[k[12]];
}

static void aes_mix_col(uint8_t *c, int inverse) /* :168-185 */
{
    if (inverse) {
        const uint8_t u = xtime(xtime(c[0] ^ c[2])), v = xtime(xtime(c[1] ^ c[3]));
        c[0] ^= u;
        c[1] ^= v;
        c[2] ^= u;
        c[3] ^= v;
    }
    const uint8_t t = c[0] ^ c[1] ^ c[2] ^ c[3], c0 = c[0];
    c[0] ^= xtime(c[0] ^ c[1]) ^ t;
    c[1] ^= xtime(c[1] ^ c[2]) ^ t;
    c[2] ^= xtime(c[2] ^ c[3]) ^ t;
    c[3] ^= xtime(c[3] ^ c0) ^ t;
}

/* one main-loop iteration of aes_enc_dec (TI_aes_128.c:131-227) on one replica */
static void aes_round(uint8_t s[16], uint8_t k[16], int dir, int rd)
{
    uint8_t t[16];
    if (dir) {
        aes_key_inv(k, 9 - rd);
        if (rd > 0)
            for (int c = 0; c < 4; ++c)
                aes_mix_col(s + 4 * c, 1);
        for (int c = 0; c < 4; ++c) /* inverse shift rows: row r rotates right by r columns */
            for (int r = 0; r < 4; ++r)
                t[4 * ((c + r) & 3) + r] = s[4 * c + r];
        for (int i = 0; i < 16; ++i)
            s[i] = AES_RS[t[i]] ^ k[i];
    } else {
        for (int i = 0; i < 16; ++i)
            t[i] = AES_S[s[i] ^ k[i]];
        for (int c = 0; c < 4; ++c) /* shift rows: row r rotates left by r columns */
            for (int r = 0; r < 4; ++r)
                s[4 * c + r] = t[4 * ((c + r) & 3) + r];
        if (rd < 9)
            for (int c = 0; c < 4; ++c)
                aes_mix_col(s + 4 * c, 0);
        aes_key_fwd(k, rd);
    }
}

static inline uint32_t ld32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void st32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

static void aes_faults(uint8_t s[3][16], uint8_t k[3][16], unsigned nrep, uint32_t step, const orc_fault *fl,
                       size_t nf)
{
    for (size_t q = 0; q < nf; ++q) {
        if (fl[q].step != step || fl[q].replica >= nrep)
            continue;
        uint8_t *p = NULL;
        if (fl[q].site == ORC_SITE_AES_STATE)
            p = s[fl[q].replica] + 4 * (fl[q].index & 3);
        else if (fl[q].site == ORC_SITE_AES_KEY)
            p = k[fl[q].replica] + 4 * (fl[q].index & 3);
        if (p)
            st32(p, flip(ld32(p), fl[q].bit, 0xffffffffu));
    }
}

static void aes_sync(sync_ctx *c, uint8_t s[3][16], uint8_t k[3][16])
{
    for (int half = 0; half < 2; ++half)
        for (int w = 0; w < 4; ++w) {
            uint8_t(*a)[16] = half ? k : s;
            uint32_t v[3] = {ld32(a[0] + 4 * w), ld32(a[1] + 4 * w), ld32(a[2] + 4 * w)};
            store_sync32(c, v); /* state[i] = ..., key[i] = ... are memory stores */
            for (int r = 0; r < 3; ++r)
                st32(a[r] + 4 * w, v[r]);
        }
}

/* state3[r] / key3[r]: replica r's memory copy (all three the same pointer in the single-copy mode; ORC_F_MEMORY_COPIES: see
 * sha_item3) */
static void aes_item3(uint8_t *const state3[3], uint8_t *const key3[3], int dir, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    uint8_t s[3][16], k[3][16];
    const unsigned R = c->nrep;
    aes_tables();
    for (unsigned r = 0; r < 3; ++r) {
        memcpy(s[r], state3[r], 16);
        memcpy(k[r], key3[r], 16);
    }
    if (dir)
        for (unsigned r = 0; r < R; ++r) { /* :110-128 last encryption key, first AddRoundKey */
            for (int rd = 0; rd < 10; ++rd)
                aes_key_fwd(k[r], rd);
            for (int i = 0; i < 16; ++i)
                s[r][i] ^= k[r][i];
        }
    for (int rd = 0; rd < 10; ++rd) {
        aes_faults(s, k, R, (uint32_t)rd, fl, nf);
        for (unsigned r = 0; r < R; ++r)
            aes_round(s[r], k[r], dir, rd);
        if (c->sync_every && rd < 9)
            aes_sync(c, s, k);
    }
    aes_faults(s, k, R, 10u, fl, nf);
    if (!dir)
        for (unsigned r = 0; r < R; ++r) /* :228-233 last AddRoundKey */
            for (int i = 0; i < 16; ++i)
                s[r][i] ^= k[r][i];
    aes_sync(c, s, k);
    for (unsigned r = 0; r < 3; ++r)
        if (r == 0 || (r < R && state3[r] != state3[0])) {
            memcpy(state3[r], s[r], 16);
            memcpy(key3[r], k[r], 16);
        }
}

static void aes_item(uint8_t state[16], uint8_t key[16], int dir, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    uint8_t *const s3[3] = {state, state, state}, *const k3[3] = {key, key, key};
    aes_item3(s3, k3, dir, c, fl, nf);
}

/* aes_enc_dec with its loops as written (TI_aes_128.c:107-235), for ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC: the two loop counters
 * `round` and `i` (unsigned char) are replica-private registers.  Sync points added to the frozen schedule, the reference's rule set
 * for -TMR -noMemReplication on the source as written:
 *   every evaluated branch condition: the loop conditions `round < 10`, `i < 16`, `i > 3`, `i < 4`, the tests of `dir`, and the
 *     operands of `(round > 0 && dir) || (round < 9 && !dir)` in short-circuit order                 synchronization.cpp:146-155
 *   every GEP with a variable index: state[i], key[i], key[i-4], state[buf4 + c] (buf4 = i << 2), Rcon[round], Rcon[9-round], and the
 *     table lookups sbox[..] / rsbox[..], whose index is DATA (loads: off with -noLoadSync; the state[] / key[] stores: off with
 *     -noStoreAddrSync); constant indices (key[13], the ShiftRows moves) have nothing to vote (syncGEP returns early, :428-431)
 * state[] and key[] stay what they are in the frozen schedule: replica-private until the function's exit, where their 8 dwords are
 * voted as stored data; a voted (or, unvoted, replica 0's) offset selects the element every copy accesses.  Fault sites:
 * ORC_SITE_AES_ROUND / _I of a replica (8 bits live), `step` = how many LOOP conditions the call has evaluated (the flip lands right
 * before the next one); ORC_SITE_AES_STATE / _KEY keep their meaning (start of main-loop iteration `step`, 10: after the loop).  A wild
 * index reads 0 / stores nothing; a walk that a corrupted counter keeps alive is cut after 4096 loop conditions (a clean call
 * evaluates at most 551). */
typedef struct {
    sync_ctx *c;
    const orc_fault *fl;
    size_t nf;
    unsigned R;
    int bs, ls, ss, wd;
    uint8_t s[3][16], k[3][16];
    uint32_t round[3], i[3];
    uint64_t tick;
} aesx;

static int aesx_loop(aesx *m, int c0, int c1, int c2, const uint32_t cur[3], uint32_t limit, int gt)
{
    /* the counter's upsets land right before the condition reads it: re-evaluate after the hook */
    (void)c0, (void)c1, (void)c2;
    for (size_t q = 0; q < m->nf; ++q)
        if ((uint64_t)m->fl[q].step == m->tick && m->fl[q].replica < m->R) {
            uint32_t *t = m->fl[q].site == ORC_SITE_AES_ROUND ? m->round : m->fl[q].site == ORC_SITE_AES_I ? m->i : NULL;
            if (t)
                t[m->fl[q].replica] = flip(t[m->fl[q].replica], m->fl[q].bit, 0xffu);
        }
    if (m->tick >= 4096u) {
        m->wd = 1;
        return 0;
    }
    m->tick++;
    return gt ? branch_cond(m->c, cur[0] > limit, cur[1] > limit, cur[2] > limit, m->bs)
              : branch_cond(m->c, cur[0] < limit, cur[1] < limit, cur[2] < limit, m->bs);
}
static void aesx_set(uint32_t reg[3], uint32_t v) { reg[0] = reg[1] = reg[2] = v; }
static int aesx_if(aesx *m, int c0, int c1, int c2) { return branch_cond(m->c, (uint32_t)c0, (uint32_t)c1, (uint32_t)c2, m->bs); }
static uint32_t aesx_off(aesx *m, const int32_t idx[3], int store)
{
    const uint32_t v[3] = {(uint32_t)idx[0], (uint32_t)idx[1], (uint32_t)idx[2]};
    return gep_offset(m->c, v, store ? m->ss : m->ls);
}
static void aesx_ld(aesx *m, uint8_t (*arr)[16], const int32_t idx[3], uint32_t out[3])
{
    const uint32_t o = aesx_off(m, idx, 0);
    for (int r = 0; r < 3; ++r)
        out[r] = o < 16u ? arr[r][o] : 0u;
}
static void aesx_st(aesx *m, uint8_t (*arr)[16], const int32_t idx[3], const uint32_t v[3])
{
    const uint32_t o = aesx_off(m, idx, 1);
    uint32_t d[3] = {v[0] & 0xffu, v[1] & 0xffu, v[2] & 0xffu};
    local_sync32(m->c, d); /* ORC_F_LOCAL_STORE_SYNC: the data of the in-place store */
    if (o < 16u)
        for (int r = 0; r < 3; ++r)
            arr[r][o] = (uint8_t)d[r];
}
/* a counter update as the -O0 IR has it: load, add, STORE -- the store's data vote under ORC_F_LOCAL_STORE_SYNC */
static void aesx_upd(aesx *m, uint32_t reg[3], uint32_t d)
{
    for (int r = 0; r < 3; ++r)
        reg[r] = (reg[r] + d) & 0xffu;
    local_sync32(m->c, reg);
}
/* arr[dst] = arr[src] with constant indices (the shift rows): an in-place store, its data voted under ORC_F_LOCAL_STORE_SYNC */
static void aesx_mov(aesx *m, uint8_t (*arr)[16], int dst, const uint32_t v[3])
{
    uint32_t d[3] = {v[0] & 0xffu, v[1] & 0xffu, v[2] & 0xffu};
    local_sync32(m->c, d);
    for (int r = 0; r < 3; ++r)
        arr[r][dst] = (uint8_t)d[r];
}
/* buf = arr[src]: a store into a local's alloca */
static void aesx_buf(aesx *m, uint32_t buf[3], uint8_t (*arr)[16], int src)
{
    for (int r = 0; r < 3; ++r)
        buf[r] = arr[r][src];
    local_sync32(m->c, buf);
}
static void aesx_tab(aesx *m, const uint8_t *tab, uint32_t size, const uint32_t x[3], uint32_t out[3])
{
    const int32_t idx[3] = {(int32_t)x[0], (int32_t)x[1], (int32_t)x[2]};
    const uint32_t o = aesx_off(m, idx, 0);
    for (int r = 0; r < 3; ++r)
        out[r] = o < size ? tab[o] : 0u;
}
#define AX3(dst, expr)                                                                                         \
    do {                                                                                                       \
        for (int r_ = 0; r_ < 3; ++r_) {                                                                       \
            const int r = r_;                                                                                  \
            (dst)[r] = (expr);                                                                                 \
        }                                                                                                      \
    } while (0)
/* key[0..3] ^= sbox[key[13, 14, 15, 12]] (^ Rcon[rc] on byte 0): the key[] indices are constants, the table indices are data */
static void aesx_key_core(aesx *m, const uint32_t rc[3])
{
    static const int src[4] = {13, 14, 15, 12};
    for (int b = 0; b < 4; ++b) {
        uint32_t x[3], sb[3];
        AX3(x, m->k[r][src[b]]);
        aesx_tab(m, AES_S, 256u, x, sb);
        if (b == 0) {
            uint32_t rcv[3];
            aesx_tab(m, AES_RCON, 10u, rc, rcv);
            AX3(sb, sb[r] ^ rcv[r]);
        }
        uint32_t kv[3];
        for (int r = 0; r < 3; ++r)
            kv[r] = (uint32_t)m->k[r][b] ^ (sb[r] & 0xffu);
        aesx_mov(m, m->k, b, kv);
    }
}
static void aesx_key_xor(aesx *m) /* key[i] = key[i] ^ key[i-4] with the replicas' own i */
{
    int32_t ii[3], im4[3];
    uint32_t a[3], b[3], v[3];
    AX3(ii, (int32_t)m->i[r]);
    AX3(im4, (int32_t)m->i[r] - 4);
    aesx_ld(m, m->k, ii, a);
    aesx_ld(m, m->k, im4, b);
    AX3(v, a[r] ^ b[r]);
    aesx_st(m, m->k, ii, v);
}

static int aes_item_indexed(uint8_t *state, uint8_t *key, int dir, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    aesx mm, *m = &mm;
    memset(m, 0, sizeof *m);
    m->c = c, m->fl = fl, m->nf = nf, m->R = c->nrep;
    m->bs = (c->flags & ORC_F_BRANCH_SYNC) != 0;
    const int as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    m->ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC);
    m->ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    aes_tables();
    for (int r = 0; r < 3; ++r) {
        memcpy(m->s[r], state, 16);
        memcpy(m->k[r], key, 16);
    }
    const int d = dir ? 1 : 0;
    {
        uint32_t dv[3] = {(uint32_t)d, (uint32_t)d, (uint32_t)d};
        local_sync32(c, dv); /* the parameter `dir` into its alloca (-O0) */
    }
#define LOOP_LT(reg, lim) aesx_loop(m, 0, 0, 0, m->reg, (lim), 0)
#define LOOP_GT(reg, lim) aesx_loop(m, 0, 0, 0, m->reg, (lim), 1)
#define INC(reg) aesx_upd(m, m->reg, 1u)
#define SET(reg, v) aesx_set(m->reg, (uint32_t)(v))
    if (aesx_if(m, d, d, d)) {                                              /* if (dir)                                  :111 */
        for (SET(round, 0); LOOP_LT(round, 10u); INC(round)) {              /*   for (round = 0; round < 10; round++)    :113 */
            aesx_key_core(m, m->round);                                     /*     key[0..3] ^= sbox[..] (^ Rcon[round]) :115-118 */
            for (SET(i, 4); LOOP_LT(i, 16u); INC(i))                        /*     for (i = 4; i < 16; i++)              :119 */
                aesx_key_xor(m);
        }
        for (SET(i, 0); LOOP_LT(i, 16u); INC(i)) {                          /*   first AddRoundKey                       :125 */
            int32_t ii[3];
            uint32_t a[3], b[3], v[3];
            AX3(ii, (int32_t)m->i[r]);
            aesx_ld(m, m->s, ii, a);
            aesx_ld(m, m->k, ii, b);
            AX3(v, a[r] ^ b[r]);
            aesx_st(m, m->s, ii, v);
        }
    }
    uint32_t iter = 0;
    for (SET(round, 0); LOOP_LT(round, 10u); INC(round)) {                  /* main loop                                 :131 */
        aes_faults(m->s, m->k, m->R, iter < 10u ? iter : 0xffffffffu, fl, nf);
        ++iter;
        if (aesx_if(m, d, d, d)) {                                          /*   if (dir): inverse key schedule          :132-141 */
            for (SET(i, 15); LOOP_GT(i, 3u); aesx_upd(m, m->i, 0xffu))
                aesx_key_xor(m);
            uint32_t rc[3];
            AX3(rc, (uint32_t)(9 - (int32_t)m->round[r]));
            aesx_key_core(m, rc);
        } else {
            for (SET(i, 0); LOOP_LT(i, 16u); INC(i)) {                      /*   state[i] = sbox[state[i] ^ key[i]]      :143-146 */
                int32_t ii[3];
                uint32_t a[3], b[3], x[3], v[3];
                AX3(ii, (int32_t)m->i[r]);
                aesx_ld(m, m->s, ii, a);
                aesx_ld(m, m->k, ii, b);
                AX3(x, a[r] ^ b[r]);
                aesx_tab(m, AES_S, 256u, x, v);
                aesx_st(m, m->s, ii, v);
            }
            {                                                               /*   shift rows: constant indices            :148-166 */
                uint32_t b1[3], b2[3], t[3];
#define ST_MOV(dst, src) do { AX3(t, m->s[r][src]); aesx_mov(m, m->s, dst, t); } while (0)
                aesx_buf(m, b1, m->s, 1);
                ST_MOV(1, 5);
                ST_MOV(5, 9);
                ST_MOV(9, 13);
                aesx_mov(m, m->s, 13, b1);
                aesx_buf(m, b1, m->s, 2);
                aesx_buf(m, b2, m->s, 6);
                ST_MOV(2, 10);
                ST_MOV(6, 14);
                aesx_mov(m, m->s, 10, b1);
                aesx_mov(m, m->s, 14, b2);
                aesx_buf(m, b1, m->s, 15);
                ST_MOV(15, 11);
                ST_MOV(11, 7);
                ST_MOV(7, 3);
                aesx_mov(m, m->s, 3, b1);
            }
        }
        /* if ((round > 0 && dir) || (round < 9 && !dir))                                                           :168 */
        int mix = 0;
        if (aesx_if(m, m->round[0] > 0u, m->round[1] > 0u, m->round[2] > 0u))
            mix = aesx_if(m, d, d, d);
        if (!mix && aesx_if(m, m->round[0] < 9u, m->round[1] < 9u, m->round[2] < 9u))
            mix = aesx_if(m, !d, !d, !d);
        if (mix) {
            for (SET(i, 0); LOOP_LT(i, 4u); INC(i)) {                       /*   for (i = 0; i < 4; i++)                 :169 */
                int32_t b4[3][4];
                for (int r = 0; r < 3; ++r)
                    for (int cc = 0; cc < 4; ++cc)
                        b4[r][cc] = (int32_t)(((m->i[r] << 2) & 0xffu) + (uint32_t)cc); /* buf4 = (i << 2), an unsigned char */
#define IDX(cc) ((const int32_t[3]){b4[0][cc], b4[1][cc], b4[2][cc]})
                uint32_t a[3], b[3], cv[3], dv[3], buf1[3], buf2[3], buf3[3], v[3];
                {
                    uint32_t b4v[3] = {(uint32_t)b4[0][0], (uint32_t)b4[1][0], (uint32_t)b4[2][0]};
                    local_sync32(m->c, b4v);                                /*     buf4 = (i << 2): a local's store      :170 */
                    for (int r = 0; r < 3; ++r)
                        for (int cc = 0; cc < 4; ++cc)
                            b4[r][cc] = (int32_t)(b4v[r] + (uint32_t)cc);
                }
                if (aesx_if(m, d, d, d)) {                                  /*     if (dir): precompute                  :171-175 */
                    aesx_ld(m, m->s, IDX(0), a);
                    aesx_ld(m, m->s, IDX(2), b);
                    AX3(buf1, xtime(xtime((uint8_t)(a[r] ^ b[r]))));
                    local_sync32(m->c, buf1);
                    aesx_ld(m, m->s, IDX(1), a);
                    aesx_ld(m, m->s, IDX(3), b);
                    AX3(buf2, xtime(xtime((uint8_t)(a[r] ^ b[r]))));
                    local_sync32(m->c, buf2);
                    for (int cc = 0; cc < 4; ++cc) {                        /*     state[buf4 + cc] ^= buf1 / buf2: ONE GEP serves the   */
                        const uint32_t o = aesx_off(m, IDX(cc), 0);          /*     load and the store of a compound assignment; its first */
                        uint32_t x[3];                                      /*     user is the load (user_back(), :341-351): load class    */
                        for (int r = 0; r < 3; ++r)
                            x[r] = (o < 16u ? (uint32_t)m->s[r][o] : 0u) ^ (((cc & 1) ? buf2[r] : buf1[r]) & 0xffu);
                        local_sync32(m->c, x);                              /*     ... and the stored byte is a data vote                  */
                        if (o < 16u)
                            for (int r = 0; r < 3; ++r)
                                m->s[r][o] = (uint8_t)x[r];
                    }
                }
                aesx_ld(m, m->s, IDX(0), a);                                /*     buf1 = the column's xor               :177 */
                aesx_ld(m, m->s, IDX(1), b);
                aesx_ld(m, m->s, IDX(2), cv);
                aesx_ld(m, m->s, IDX(3), dv);
                AX3(buf1, a[r] ^ b[r] ^ cv[r] ^ dv[r]);
                local_sync32(m->c, buf1);
                aesx_ld(m, m->s, IDX(0), buf2);                             /*     buf2 = state[buf4]                    :178 */
                local_sync32(m->c, buf2);
                for (int cc = 0; cc < 4; ++cc) {                            /*     the four rows                         :179-182 */
                    aesx_ld(m, m->s, IDX(cc), a);
                    if (cc < 3)
                        aesx_ld(m, m->s, IDX(cc + 1), b);
                    else
                        AX3(b, buf2[r]);
                    AX3(buf3, (uint32_t)(uint8_t)(a[r] ^ b[r]));
                    local_sync32(m->c, buf3);                               /*     buf3 = state[..] ^ state[..]                           */
                    AX3(buf3, xtime((uint8_t)buf3[r]));
                    local_sync32(m->c, buf3);                               /*     buf3 = galois_mul2(buf3)                               */
                    aesx_ld(m, m->s, IDX(cc), a);
                    AX3(v, a[r] ^ buf3[r] ^ buf1[r]);
                    aesx_st(m, m->s, IDX(cc), v);
                }
#undef IDX
            }
        }
        if (aesx_if(m, d, d, d)) {                                          /*   if (dir): inverse shift rows, rsbox     :187-211 */
            {
                uint32_t b1[3], b2[3], t[3];
                aesx_buf(m, b1, m->s, 13);                                  /*   Row 1                                                    */
                ST_MOV(13, 9);
                ST_MOV(9, 5);
                ST_MOV(5, 1);
                aesx_mov(m, m->s, 1, b1);
                aesx_buf(m, b1, m->s, 10);                                  /*   Row 2                                                    */
                aesx_buf(m, b2, m->s, 14);
                ST_MOV(10, 2);
                ST_MOV(14, 6);
                aesx_mov(m, m->s, 2, b1);
                aesx_mov(m, m->s, 6, b2);
                aesx_buf(m, b1, m->s, 3);                                   /*   Row 3                                                    */
                ST_MOV(3, 7);
                ST_MOV(7, 11);
                ST_MOV(11, 15);
                aesx_mov(m, m->s, 15, b1);
            }
            for (SET(i, 0); LOOP_LT(i, 16u); INC(i)) {                      /*   state[i] = rsbox[state[i]] ^ key[i]     :208-211 */
                int32_t ii[3];
                uint32_t a[3], b[3], x[3], v[3];
                AX3(ii, (int32_t)m->i[r]);
                aesx_ld(m, m->s, ii, a);
                aesx_tab(m, AES_RS, 256u, a, x);
                aesx_ld(m, m->k, ii, b);
                AX3(v, x[r] ^ b[r]);
                aesx_st(m, m->s, ii, v);
            }
        } else {                                                            /*   key schedule                            :213-226 */
            aesx_key_core(m, m->round);
            for (SET(i, 4); LOOP_LT(i, 16u); INC(i))
                aesx_key_xor(m);
        }
    }
    aes_faults(m->s, m->k, m->R, 10u, fl, nf);
    if (aesx_if(m, !d, !d, !d))                                             /* if (!dir): last AddRoundKey               :228-233 */
        for (SET(i, 0); LOOP_LT(i, 16u); INC(i)) {
            int32_t ii[3];
            uint32_t a[3], b[3], v[3];
            AX3(ii, (int32_t)m->i[r]);
            aesx_ld(m, m->s, ii, a);
            aesx_ld(m, m->k, ii, b);
            AX3(v, a[r] ^ b[r]);
            aesx_st(m, m->s, ii, v);
        }
#undef LOOP_LT
#undef LOOP_GT
#undef INC
#undef SET
#undef ST_MOV
    aes_sync(c, m->s, m->k);
    memcpy(state, m->s[0], 16);
    memcpy(key, m->k[0], 16);
    return m->wd;
}
#undef AX3

void orc_aes128_plain(uint8_t state[16], uint8_t key[16], uint8_t dir)
{
    orc_stats st = {0, 0, 0, 0};
    sync_ctx c = {1, 0, &st, 0, 0};
    aes_item(state, key, dir ? 1 : 0, &c, NULL, 0);
}

void orc_aes128_xmr(uint8_t *states, uint8_t *keys, size_t nblocks, int dir, const orc_cfg *cfg,
                    const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        while (fp < nfaults && fs[fp].item < b)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == b)
            ++fe;
        c.detected = 0;
        if (cfg->flags & ORC_F_MEMORY_COPIES) { /* `replicas` copies of both arrays back to back */
            const size_t cs = nblocks * 16;
            const unsigned R = cfg->replicas;
            uint8_t *const s3[3] = {states + 16 * b, states + (R > 1 ? cs : 0) + 16 * b, states + (R > 2 ? 2 * cs : 0) + 16 * b};
            uint8_t *const k3[3] = {keys + 16 * b, keys + (R > 1 ? cs : 0) + 16 * b, keys + (R > 2 ? 2 * cs : 0) + 16 * b};
            aes_item3(s3, k3, dir ? 1 : 0, &c, fs + fp, fe - fp);
        } else if (cfg->flags & ORC_F_INDEXED) { /* the loop counters inside the sphere of replication */
            aes_item_indexed(states + 16 * b, keys + 16 * b, dir ? 1 : 0, &c, fs + fp, fe - fp);
        } else {
            aes_item(states + 16 * b, keys + 16 * b, dir ? 1 : 0, &c, fs + fp, fe - fp);
        }
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[b] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* crc16                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* crc16.c:21-31, all truncations as written (x is unsigned char, crc unsigned short) */
static inline uint16_t crc_step(uint16_t crc, uint8_t byte, uint8_t *xout)
{
    uint8_t x = (uint8_t)((crc >> 8) ^ byte);
    x ^= (uint8_t)(x >> 4);
    if (xout)
        *xout = x;
    return (uint16_t)((uint16_t)(crc << 8) ^ (uint16_t)((uint16_t)x << 12) ^ (uint16_t)((uint16_t)x << 5) ^
                      (uint16_t)x);
}

uint16_t orc_crc16_plain(const uint8_t *data, uint32_t length)
{
    uint16_t crc = 0xFFFF;
    for (uint32_t t = 0; t < length; ++t)
        crc = crc_step(crc, data[t], NULL);
    return crc;
}

/* crc16 with `while (length--)` as written (crc16.c:25), for ORC_F_BRANCH_SYNC: `length` is a replica-private unsigned char,
 * its loop condition is voted at every evaluation (a terminator sync on an i1, synchronization.cpp:146-155); `*data_p++` has a
 * constant GEP offset, so there is no address vote in this function (syncGEP returns early, :428-431) and ORC_F_ADDR_SYNC
 * changes nothing.  A corrupted counter that keeps the loop alive is cut after 4 * len + 256 iterations; bytes past the
 * block read as 0. */
static uint16_t crc_item_branch(const uint8_t *data, uint32_t len, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    uint32_t crc[3] = {0xFFFF, 0xFFFF, 0xFFFF}, ln[3] = {len & 0xffu, len & 0xffu, len & 0xffu};
    const unsigned R = c->nrep;
    const int bs = (c->flags & ORC_F_BRANCH_SYNC) != 0;
    const uint64_t cap = 4ull * len + 256ull;
    local_sync32(c, ln); /* the parameter `length` into its alloca (-O0) */
    for (uint32_t it = 0;; ++it) {
        for (size_t q = 0; q < nf; ++q)
            if (fl[q].site == ORC_SITE_CRC_LEN && fl[q].step == it && fl[q].replica < R)
                ln[fl[q].replica] = flip(ln[fl[q].replica], fl[q].bit, 0xffu);
        /* while (length--): load, decrement, STORE (the data vote of ORC_F_LOCAL_STORE_SYNC), then the branch on the loaded value */
        const uint32_t old[3] = {ln[0], ln[1], ln[2]};
        for (unsigned r = 0; r < 3; ++r)
            ln[r] = (ln[r] - 1u) & 0xffu; /* length-- : the decrement happens on both exits */
        local_sync32(c, ln);
        const int go = branch_cond(c, old[0] != 0, old[R > 1 ? 1 : 0] != 0, old[R > 2 ? 2 : 0] != 0, bs);
        if (!go || it >= cap)
            break;
        const uint8_t byte = it < len ? data[it] : 0;
        uint32_t x[3] = {0, 0, 0};
        for (unsigned r = 0; r < R; ++r) {
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CRC_CRC && fl[q].step == it && it < len && fl[q].replica == r)
                    crc[r] = flip(crc[r], fl[q].bit, 0xffffu);
            x[r] = (uint8_t)((crc[r] >> 8) ^ byte);
        }
        local_sync32(c, x);                                   /* x = crc >> 8 ^ *data_p++      :26 */
        for (unsigned r = 0; r < R; ++r)
            x[r] = (uint8_t)(x[r] ^ (x[r] >> 4));
        local_sync32(c, x);                                   /* x ^= x >> 4                   :27 */
        for (unsigned r = 0; r < R; ++r) {
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CRC_X && fl[q].step == it && it < len && fl[q].replica == r)
                    x[r] = (uint8_t)flip(x[r], fl[q].bit, 0xffu);
            crc[r] = (uint16_t)((uint16_t)(crc[r] << 8) ^ (uint16_t)((uint16_t)x[r] << 12) ^
                                (uint16_t)((uint16_t)x[r] << 5) ^ (uint16_t)x[r]);
        }
        local_sync32(c, crc);                                 /* crc = ...                     :28 */
        if (c->sync_every && ((it + 1) % c->sync_every) == 0 && (it + 1) < len)
            sync32(c, crc);
    }
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].site == ORC_SITE_CRC_CRC && fl[q].step == len && fl[q].replica < R)
            crc[fl[q].replica] = flip(crc[fl[q].replica], fl[q].bit, 0xffffu);
    sync32(c, crc);
    return (uint16_t)crc[0];
}

/* data3[r]: replica r's memory copy of the block (all three the same pointer in the single-copy mode; ORC_F_MEMORY_COPIES: see
 * sha_item3).  The return value is the function's one sync point either way. */
static uint16_t crc_item3(const uint8_t *const data3[3], uint32_t len, sync_ctx *c, const orc_fault *fl, size_t nf, uint32_t out[3])
{
    uint32_t crc[3] = {0xFFFF, 0xFFFF, 0xFFFF};
    const unsigned R = c->nrep;
    if (c->flags & ORC_F_BRANCH_SYNC) {
        out[0] = out[1] = out[2] = crc_item_branch(data3[0], len, c, fl, nf);
        return (uint16_t)out[0];
    }
    for (uint32_t t = 0; t < len; ++t) {
        for (unsigned r = 0; r < R; ++r) {
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CRC_CRC && fl[q].step == t && fl[q].replica == r)
                    crc[r] = flip(crc[r], fl[q].bit, 0xffffu);
            uint8_t x = (uint8_t)((crc[r] >> 8) ^ data3[r][t]);
            x ^= (uint8_t)(x >> 4);
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CRC_X && fl[q].step == t && fl[q].replica == r)
                    x = (uint8_t)flip(x, fl[q].bit, 0xffu);
            crc[r] = (uint16_t)((uint16_t)(crc[r] << 8) ^ (uint16_t)((uint16_t)x << 12) ^
                                (uint16_t)((uint16_t)x << 5) ^ (uint16_t)x);
        }
        if (c->sync_every && ((t + 1) % c->sync_every) == 0 && (t + 1) < len)
            sync32(c, crc);
    }
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].site == ORC_SITE_CRC_CRC && fl[q].step == len && fl[q].replica < R)
            crc[fl[q].replica] = flip(crc[fl[q].replica], fl[q].bit, 0xffffu);
    sync32(c, crc); /* return-value sync, synchronization.cpp:741-949 (ReturnInst) */
    out[0] = crc[0], out[1] = crc[1], out[2] = crc[2];
    return (uint16_t)crc[0];
}

static uint16_t crc_item(const uint8_t *data, uint32_t len, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    const uint8_t *const d3[3] = {data, data, data};
    uint32_t out[3];
    return crc_item3(d3, len, c, fl, nf, out);
}

void orc_crc16_xmr(const uint8_t *data, uint32_t block_len, size_t nblocks, uint16_t *crcs, const orc_cfg *cfg,
                   const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        while (fp < nfaults && fs[fp].item < b)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == b)
            ++fe;
        c.detected = 0;
        if (cfg->flags & ORC_F_MEMORY_COPIES) { /* `replicas` copies of the stream (and of the result array) back to back */
            const size_t cs = nblocks * (size_t)block_len;
            const unsigned R = cfg->replicas;
            const uint8_t *const d3[3] = {data + (size_t)b * block_len, data + (R > 1 ? cs : 0) + (size_t)b * block_len,
                                          data + (R > 2 ? 2 * cs : 0) + (size_t)b * block_len};
            uint32_t out[3];
            crc_item3(d3, block_len, &c, fs + fp, fe - fp, out);
            for (unsigned r = 0; r < R; ++r)
                crcs[r * nblocks + b] = (uint16_t)out[r];
        } else {
            crcs[b] = crc_item(data + (size_t)b * block_len, block_len, &c, fs + fp, fe - fp);
        }
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[b] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* CHStone sha                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* tests/chstone/sha/sha.c.  Not FIPS SHA-1: the schedule has no rotate (W[i] = W[i-3]^W[i-8]^W[i-14]^W[i-16], :92-94 --
 * the original "SHA"), and the 16 input words are assembled LITTLE-endian from the byte stream by the file's own memcpy
 * (:61-80).  sha_final (:153-172) writes the 0x80 marker with `sha_info_data[count++] = 0x80` where count is a BYTE count
 * used as a WORD index: only count == 0, i.e. a total length that is a multiple of 64, pads the way the routine means to
 * (word 0 = 0x80, words 1..13 = 0, word 14 = bit count high, word 15 = bit count low), and that is the only case the
 * benchmark exercises (2 x 8192 bytes, sha.h:59-60).  Sync points of the protected version: sha_info_digest[i] += ... at
 * the end of every sha_transform are memory stores (:113-117) -> five store-data votes per transform. */
static inline uint32_t rotl(uint32_t x, unsigned n) { return (x << n) | (x >> (32 - n)); }

static void chsha_transform(uint32_t dg[3][5], const uint32_t in[16], unsigned nrep, uint32_t cidx, const orc_fault *fl,
                            size_t nf)
{
    for (unsigned r = 0; r < nrep; ++r) {
        uint32_t W[80], v[5];
        for (unsigned t = 0; t < 80; ++t) {
            W[t] = t < 16 ? in[t] : (W[t - 3] ^ W[t - 8] ^ W[t - 14] ^ W[t - 16]);
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CHSHA_W && fl[q].replica == r && fl[q].step == cidx * 80 + t)
                    W[t] = flip(W[t], fl[q].bit, 0xffffffffu);
        }
        for (unsigned w = 0; w < 5; ++w)
            v[w] = dg[r][w];
        for (unsigned t = 0; t < 80; ++t) {
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CHSHA_WV && fl[q].replica == r && fl[q].step == cidx * 80 + t)
                    v[fl[q].index % 5] = flip(v[fl[q].index % 5], fl[q].bit, 0xffffffffu);
            const uint32_t A = v[0], B = v[1], C = v[2], D = v[3], E = v[4];
            uint32_t f, k;
            if (t < 20) {
                f = (B & C) | (~B & D);
                k = 0x5a827999u;
            } else if (t < 40) {
                f = B ^ C ^ D;
                k = 0x6ed9eba1u;
            } else if (t < 60) {
                f = (B & C) | (B & D) | (C & D);
                k = 0x8f1bbcdcu;
            } else {
                f = B ^ C ^ D;
                k = 0xca62c1d6u;
            }
            const uint32_t temp = rotl(A, 5) + f + E + W[t] + k;
            v[4] = D;
            v[3] = C;
            v[2] = rotl(B, 30);
            v[1] = A;
            v[0] = temp;
        }
        for (unsigned w = 0; w < 5; ++w)
            dg[r][w] += v[w];
    }
}

static void chsha_item(const uint8_t *data, uint32_t len, uint32_t out[5], sync_ctx *c, const orc_fault *fl, size_t nf)
{
    static const uint32_t IV[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u};
    uint32_t dg[3][5];
    const unsigned R = c->nrep;
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned w = 0; w < 5; ++w)
            dg[r][w] = IV[w];
    const uint32_t nblk = len / 64;
    for (uint32_t cidx = 0; cidx <= nblk; ++cidx) {
        uint32_t in[16];
        if (cidx < nblk) {
            const uint8_t *p = data + (size_t)cidx * 64;
            for (unsigned t = 0; t < 16; ++t)
                in[t] = (uint32_t)p[4 * t] | ((uint32_t)p[4 * t + 1] << 8) | ((uint32_t)p[4 * t + 2] << 16) |
                        ((uint32_t)p[4 * t + 3] << 24);
        } else { /* sha_final with count == 0 */
            memset(in, 0, sizeof in);
            in[0] = 0x80u;
            in[14] = len >> 29;
            in[15] = len << 3;
        }
        for (size_t q = 0; q < nf; ++q)
            if (fl[q].site == ORC_SITE_CHSHA_DIGEST && fl[q].step == cidx && fl[q].replica < R)
                dg[fl[q].replica][fl[q].index % 5] = flip(dg[fl[q].replica][fl[q].index % 5], fl[q].bit, 0xffffffffu);
        chsha_transform(dg, in, R, cidx, fl, nf);
        for (unsigned w = 0; w < 5; ++w) {
            uint32_t v[3] = {dg[0][w], dg[1][w], dg[2][w]};
            store_sync32(c, v);
            dg[0][w] = v[0];
            dg[1][w] = v[1];
            dg[2][w] = v[2];
        }
    }
    for (unsigned w = 0; w < 5; ++w)
        out[w] = dg[0][w];
}

/* CHStone sha with its loops as written (sha.c:84-172), for ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC: sha_transform's loop counter `i`
 * (an int) and sha_update's `count` are replica-private registers.  memcpy / memset are library names (functions.config:12): calls
 * outside the sphere of replication, as in sha256.  Sync points added to the frozen schedule, the reference's rule set for -TMR
 * -noMemReplication on the source as written:
 *   every evaluated branch condition: `count >= SHA_BLOCKSIZE` (:141), the carry test of sha_update (:136), `count > 56` of sha_final
 *     (:162), and the six loop conditions of sha_transform (17 + 65 + 4 x 21 = 166 per transform)            synchronization.cpp:146-155
 *   every GEP with a variable index: sha_info_data[i] (load) and W[i] (store) of the copy loop (:88-90); W[i-3], W[i-8], W[i-14],
 *     W[i-16] (loads) and W[i] (store) of the expansion (:91-93); W[i] (load) of the 80 rounds (:100-111) = 432 per transform
 *     (loads: off with -noLoadSync; stores: off with -noStoreAddrSync); the digest words have constant indices
 * W[] stays replica-private as in the frozen schedule; a voted (or, unvoted, replica 0's) offset selects the element every copy
 * accesses.  Fault sites: ORC_SITE_CHSHA_I / _COUNT of a replica, `step` = how many LOOP conditions the call has evaluated;
 * ORC_SITE_CHSHA_DIGEST keeps its meaning (before transform `step`).  A wild index reads 0 / stores nothing; blocks past the message
 * read as 0; a walk that a corrupted counter keeps alive is cut after 4 x the clean count + 1024 loop conditions. */
typedef struct {
    sync_ctx *c;
    const orc_fault *fl;
    size_t nf;
    unsigned R;
    int bs, ls, ss;
    uint32_t i[3], count[3];
    uint64_t tick, cap;
} chx;

static int chx_loop(chx *m, const uint32_t reg[3], int32_t limit, int ge)
{
    for (size_t q = 0; q < m->nf; ++q)
        if ((uint64_t)m->fl[q].step == m->tick && m->fl[q].replica < m->R) {
            uint32_t *t = m->fl[q].site == ORC_SITE_CHSHA_I ? m->i : m->fl[q].site == ORC_SITE_CHSHA_COUNT ? m->count : NULL;
            if (t)
                t[m->fl[q].replica] = flip(t[m->fl[q].replica], m->fl[q].bit, 0xffffffffu);
        }
    if (m->tick >= m->cap)
        return 0;
    m->tick++;
    const int32_t a = (int32_t)reg[0], b = (int32_t)reg[1], d = (int32_t)reg[2]; /* the counters are ints */
    return ge ? branch_cond(m->c, a >= limit, b >= limit, d >= limit, m->bs) : branch_cond(m->c, a < limit, b < limit, d < limit, m->bs);
}
static uint32_t chx_off(chx *m, int32_t delta, int store)
{
    const uint32_t v[3] = {m->i[0] + (uint32_t)delta, m->i[1] + (uint32_t)delta, m->i[2] + (uint32_t)delta};
    return gep_offset(m->c, v, store ? m->ss : m->ls);
}
static void chx_set(uint32_t reg[3], uint32_t v) { reg[0] = reg[1] = reg[2] = v; }
static void chx_add(uint32_t reg[3], uint32_t d)
{
    for (int r = 0; r < 3; ++r)
        reg[r] += d;
}
/* a counter update as the -O0 IR has it: load, add, STORE -- the store's data vote under ORC_F_LOCAL_STORE_SYNC */
static void chx_upd(chx *m, uint32_t reg[3], uint32_t d)
{
    chx_add(reg, d);
    local_sync32(m->c, reg);
}

static void chx_transform(chx *m, uint32_t dg[3][5], const uint32_t in[16])
{
    uint32_t W[3][80], v[3][5];
    memset(W, 0, sizeof W);
    /* ORC_F_LOCAL_STORE_SYNC: the data of every store of the -O0 IR -- ++i, W[i] = .., A..E = .., temp / E / D / C / B / A of FUNC */
    for (chx_set(m->i, 0); chx_loop(m, m->i, 16, 0); chx_upd(m, m->i, 1)) {   /* W[i] = sha_info_data[i]          :88-90 */
        const uint32_t ol = chx_off(m, 0, 0), os = chx_off(m, 0, 1);
        uint32_t x[3];
        for (int r = 0; r < 3; ++r)
            x[r] = ol < 16u ? in[ol] : 0u;
        local_sync32(m->c, x);
        if (os < 80u)
            for (int r = 0; r < 3; ++r)
                W[r][os] = x[r];
    }
    for (chx_set(m->i, 16); chx_loop(m, m->i, 80, 0); chx_upd(m, m->i, 1)) {  /* the expansion                     :91-93 */
        const uint32_t o3 = chx_off(m, -3, 0), o8 = chx_off(m, -8, 0), o14 = chx_off(m, -14, 0), o16 = chx_off(m, -16, 0);
        const uint32_t os = chx_off(m, 0, 1);
        uint32_t x[3];
        for (int r = 0; r < 3; ++r)
            x[r] = (o3 < 80u ? W[r][o3] : 0u) ^ (o8 < 80u ? W[r][o8] : 0u) ^ (o14 < 80u ? W[r][o14] : 0u) ^
                   (o16 < 80u ? W[r][o16] : 0u);
        local_sync32(m->c, x);
        if (os < 80u)
            for (int r = 0; r < 3; ++r)
                W[r][os] = x[r];
    }
    for (int w = 0; w < 5; ++w) {                                             /* A = sha_info_digest[0] ..         :94-98 */
        uint32_t x[3] = {dg[0][w], dg[1][w], dg[2][w]};
        local_sync32(m->c, x);
        for (int r = 0; r < 3; ++r)
            v[r][w] = x[r];
    }
    for (int seg = 0; seg < 4; ++seg)                                         /* FUNC(1..4, i)                     :100-111 */
        for (chx_set(m->i, 20u * (uint32_t)seg); chx_loop(m, m->i, 20 * (seg + 1), 0); chx_upd(m, m->i, 1)) {
            const uint32_t o = chx_off(m, 0, 0);
            uint32_t temp[3], x[3];
            for (int r = 0; r < 3; ++r) {
                const uint32_t A = v[r][0], B = v[r][1], C = v[r][2], D = v[r][3], E = v[r][4];
                const uint32_t f = seg == 0 ? ((B & C) | (~B & D)) : seg == 2 ? ((B & C) | (B & D) | (C & D)) : (B ^ C ^ D);
                const uint32_t k = seg == 0 ? 0x5a827999u : seg == 1 ? 0x6ed9eba1u : seg == 2 ? 0x8f1bbcdcu : 0xca62c1d6u;
                temp[r] = rotl(A, 5) + f + E + (o < 80u ? W[r][o] : 0u) + k;
            }
            local_sync32(m->c, temp);                                         /* temp = ..                                  */
#define CHX_MOVE(dst, expr)                                                                                    \
    do {                                                                                                       \
        for (int r = 0; r < 3; ++r)                                                                            \
            x[r] = (expr);                                                                                     \
        local_sync32(m->c, x);                                                                                 \
        for (int r = 0; r < 3; ++r)                                                                            \
            v[r][dst] = x[r];                                                                                  \
    } while (0)
            CHX_MOVE(4, v[r][3]);                                             /* E = D                                      */
            CHX_MOVE(3, v[r][2]);                                             /* D = C                                      */
            CHX_MOVE(2, rotl(v[r][1], 30));                                   /* C = ROT32(B, 30)                           */
            CHX_MOVE(1, v[r][0]);                                             /* B = A                                      */
            CHX_MOVE(0, temp[r]);                                             /* A = temp                                   */
#undef CHX_MOVE
        }
    for (int r = 0; r < 3; ++r)
        for (int w = 0; w < 5; ++w)
            dg[r][w] += v[r][w];
    for (unsigned w = 0; w < 5; ++w) {                                        /* sha_info_digest[w] += ...: stored  :113-117 */
        uint32_t x[3] = {dg[0][w], dg[1][w], dg[2][w]};
        store_sync32(m->c, x);
        dg[0][w] = x[0], dg[1][w] = x[1], dg[2][w] = x[2];
    }
}

static void chsha_item_indexed(const uint8_t *data, uint32_t len, uint32_t out[5], sync_ctx *c, const orc_fault *fl, size_t nf)
{
    static const uint32_t IV[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u};
    chx mm, *m = &mm;
    memset(m, 0, sizeof *m);
    m->c = c, m->fl = fl, m->nf = nf, m->R = c->nrep;
    m->bs = (c->flags & ORC_F_BRANCH_SYNC) != 0;
    const int as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    m->ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC);
    m->ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    const uint32_t nblk = len / 64;
    m->cap = 4ull * ((uint64_t)(nblk + 1u) * 167ull + 1ull) + 1024ull;
    uint32_t dg[3][5];
    for (unsigned r = 0; r < 3; ++r)
        for (unsigned w = 0; w < 5; ++w)
            dg[r][w] = IV[w];
    chx_set(m->count, len);
    local_sync32(c, m->count); /* the parameter `count` into its alloca (-O0) */
    /* if ((sha_info_count_lo + ((LONG) count << 3)) < sha_info_count_lo): count_lo is 0 on entry                  :136 */
    (void)branch_cond(c, 0u, 0u, 0u, m->bs); /* (0 + x < 0 is false whatever a replica's count holds) */
    {
        uint32_t lo[3] = {m->count[0] << 3, m->count[1] << 3, m->count[2] << 3}, hi[3] = {m->count[0] >> 29, m->count[1] >> 29, m->count[2] >> 29};
        local_sync32(c, lo); /* sha_info_count_lo += (LONG) count << 3                                              :139 */
        local_sync32(c, hi); /* sha_info_count_hi += (LONG) count >> 29                                             :140 */
    }
    uint32_t cidx = 0;
    for (;; chx_upd(m, m->count, (uint32_t)-64)) {                            /* while (count >= SHA_BLOCKSIZE)     :141 */
        if (!chx_loop(m, m->count, 64, 1))
            break;
        uint32_t in[16];
        for (unsigned t = 0; t < 16; ++t) {
            const uint8_t *p = data + (size_t)cidx * 64 + 4 * t;
            in[t] = cidx < nblk ? ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) : 0u;
        }
        for (size_t q = 0; q < nf; ++q)
            if (fl[q].site == ORC_SITE_CHSHA_DIGEST && fl[q].step == cidx && fl[q].replica < m->R)
                dg[fl[q].replica][fl[q].index % 5] = flip(dg[fl[q].replica][fl[q].index % 5], fl[q].bit, 0xffffffffu);
        chx_transform(m, dg, in);
        ++cidx;
    }
    /* sha_final: count = (lo_bit_count >> 3) & 0x3f = 0; sha_info_data[count++] = 0x80; if (count > 56)           :159-162 */
    {
        const uint32_t zero[3] = {0u, 0u, 0u};
        uint32_t lo[3] = {len << 3, len << 3, len << 3}, hi[3] = {len >> 29, len >> 29, len >> 29}, cn[3] = {0u, 0u, 0u}, c1[3] = {1u, 1u, 1u};
        local_sync32(c, lo);              /* lo_bit_count = sha_info_count_lo (memory: one copy, equal in every replica)          :157 */
        local_sync32(c, hi);              /* hi_bit_count = sha_info_count_hi                                                     :158 */
        local_sync32(c, cn);              /* count = (int) ((lo_bit_count >> 3) & 0x3f)                                           :159 */
        (void)gep_offset(c, zero, m->ss); /* sha_info_data[count++] = 0x80: a store through a variable index (count = 0 here)   :161 */
        local_sync32(c, c1);              /* count++                                                                              */
    }
    (void)branch_cond(c, 0u, 0u, 0u, m->bs);
    uint32_t in[16];
    memset(in, 0, sizeof in);
    in[0] = 0x80u;
    in[14] = len >> 29;
    in[15] = len << 3;
    {
        uint32_t hi[3] = {in[14], in[14], in[14]}, lo[3] = {in[15], in[15], in[15]};
        local_sync32(c, hi);              /* sha_info_data[14] = hi_bit_count                                                     :168 */
        local_sync32(c, lo);              /* sha_info_data[15] = lo_bit_count                                                     :169 */
    }
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].site == ORC_SITE_CHSHA_DIGEST && fl[q].step == cidx && fl[q].replica < m->R)
            dg[fl[q].replica][fl[q].index % 5] = flip(dg[fl[q].replica][fl[q].index % 5], fl[q].bit, 0xffffffffu);
    chx_transform(m, dg, in);
    for (unsigned w = 0; w < 5; ++w)
        out[w] = dg[0][w];
}

void orc_chsha_plain(const uint8_t *data, uint32_t len, uint32_t digest[5])
{
    orc_stats st = {0, 0, 0, 0};
    sync_ctx c = {1, 0, &st, 0, 0};
    chsha_item(data, len, digest, &c, NULL, 0);
}

void orc_chsha_xmr(const uint8_t *msgs, size_t stride, uint32_t len, size_t nmsgs, uint32_t *digests, const orc_cfg *cfg,
                   const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t m = 0; m < nmsgs; ++m) {
        while (fp < nfaults && fs[fp].item < m)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == m)
            ++fe;
        c.detected = 0;
        if (cfg->flags & ORC_F_INDEXED)
            chsha_item_indexed(msgs + m * stride, len, digests + 5 * m, &c, fs + fp, fe - fp);
        else
            chsha_item(msgs + m * stride, len, digests + 5 * m, &c, fs + fp, fe - fp);
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[m] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* cache_test                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* calc_sum, tests/cache_test/cacheTest.c:101-177 without its printing: sum += array[i] (:108); an element that is not its
 * index is counted and rewritten (:110-111, :134-135). */
void orc_cache_test_plain(int32_t *array, uint32_t n, int32_t *sum, uint32_t *nerr)
{
    uint32_t s = 0, e = 0;
    for (uint32_t i = 0; i < n; ++i) {
        s += (uint32_t)array[i];
        if (array[i] != (int32_t)i) {
            ++e;
            array[i] = (int32_t)i;
        }
    }
    *sum = (int32_t)s;
    *nerr = e;
}

/* Protected calc_sum, -noMemReplication rule set.  Replicated registers: sum, the loaded element, numberOfErrors.
 * Sync points: the data-dependent branch condition `array[i] != i` of every element (terminator sync on the i1,
 * synchronization.cpp:146-155, 741-949 -- all copies continue on the voted outcome, so the clones cannot take different
 * paths); the returned sum (ReturnInst sync); the error count where it is stored (store-data sync).  `array[i] = i` stores
 * the loop index, which is a wave-uniform scalar outside the sphere of replication: nothing to vote.  DWC: a mismatch is
 * flagged (the reference would not return from the handler) and the region continues on replica 0's condition. */
static void ct_item(int32_t *a, uint32_t n, int32_t *sum_out, uint32_t *nerr_out, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    uint32_t sum[3] = {0, 0, 0}, nerr[3] = {0, 0, 0};
    const unsigned R = c->nrep;
    for (uint32_t i = 0; i <= n; ++i) {
        for (size_t q = 0; q < nf; ++q) {
            if (fl[q].step != i || fl[q].replica >= R)
                continue;
            if (fl[q].site == ORC_SITE_CT_SUM)
                sum[fl[q].replica] = flip(sum[fl[q].replica], fl[q].bit, 0xffffffffu);
            else if (fl[q].site == ORC_SITE_CT_NERR)
                nerr[fl[q].replica] = flip(nerr[fl[q].replica], fl[q].bit, 0xffffffffu);
        }
        if (i == n)
            break;
        uint32_t cond[3] = {0, 0, 0};
        for (unsigned r = 0; r < R; ++r) {
            uint32_t v = (uint32_t)a[i]; /* loads repeated from the same address: cloning.cpp:2247-2255 */
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CT_VAL && fl[q].step == i && fl[q].replica == r)
                    v = flip(v, fl[q].bit, 0xffffffffu);
            sum[r] += v;
            cond[r] = (v != i) ? 1u : 0u;
        }
        sync32(c, cond); /* the branch condition */
        if (cond[0]) {
            for (unsigned r = 0; r < R; ++r)
                nerr[r] += 1;
            a[i] = (int32_t)i;
        }
    }
    sync32(c, sum);        /* return value */
    store_sync32(c, nerr); /* stored to the caller's error count */
    *sum_out = (int32_t)sum[0];
    *nerr_out = nerr[0];
}

/* calc_sum with its loop as written (cacheTest.c:107-131), for ORC_F_BRANCH_SYNC / ORC_F_ADDR_SYNC: the loop counter i is a
 * replica-private register beside sum and numberOfErrors.  Sync points, the reference's rule set for -TMR -noMemReplication on
 * the source as written:
 *   `i < data_array_elements` at every evaluation                                                   synchronization.cpp:146-155
 *   the GEP offsets: array[i] of `sum += array[i]` and of `array[i] != i` (loads: off with -noLoadSync), array[i] of the scrub
 *     `array[i] = i` (a store: off with -noStoreAddrSync)                                           :333-372, 413-474
 *   `array[i] != i` (voted in every schedule), the data of `array[i] = i` -- the counter itself (off with -noStoreDataSync),
 *   the returned sum, the stored error count
 * The printf block of the error branch is I/O outside the batch model, as in the default schedule.  A load keeps the ORIGINAL
 * instruction's address in every copy (cloning.cpp:2247-2255): unvoted offsets are replica 0's.  Fault sites: ORC_SITE_CT_I /
 * _SUM / _NERR of a replica, `step` = how many loop conditions the call has evaluated (the flip lands right before the next
 * one); ORC_SITE_CT_VAL = the element loaded in the iteration that condition `step` entered.  A wild index reads 0 / stores
 * nothing; a walk that a corrupted counter keeps alive is cut after 4 (n + 1) + 1024 conditions. */
static void ct_item_indexed(int32_t *a, uint32_t n, int32_t *sum_out, uint32_t *nerr_out, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    const unsigned R = c->nrep;
    const int bs = (c->flags & ORC_F_BRANCH_SYNC) != 0, as = (c->flags & ORC_F_ADDR_SYNC) != 0;
    const int ls = as && !(c->flags & ORC_F_NO_LOAD_SYNC), ss = as && !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    const uint64_t cap = 4ull * ((uint64_t)n + 1ull) + 1024ull;
    uint32_t i[3] = {0, 0, 0}, sum[3] = {0, 0, 0}, nerr[3] = {0, 0, 0};
    uint32_t first_error = 0, in_block = 0, local_errors = 0; /* the report block's state: equal in every copy (no fault site) */
    uint64_t tick = 0;
    for (;;) {
        for (size_t q = 0; q < nf; ++q)
            if ((uint64_t)fl[q].step == tick && fl[q].replica < R) {
                uint32_t *t = fl[q].site == ORC_SITE_CT_I ? i : fl[q].site == ORC_SITE_CT_SUM ? sum : fl[q].site == ORC_SITE_CT_NERR ? nerr : NULL;
                if (t)
                    t[fl[q].replica] = flip(t[fl[q].replica], fl[q].bit, 0xffffffffu);
            }
        if (tick >= cap)
            break;
        const uint64_t t0 = tick++;
        if (!branch_cond(c, i[0] < n, i[R > 1 ? 1 : 0] < n, i[R > 2 ? 2 : 0] < n, bs))
            break;
        const uint32_t o1 = gep_offset(c, i, ls);                 /* sum += array[i]        :108 */
        uint32_t v[3], cond[3] = {0, 0, 0};
        for (unsigned r = 0; r < 3; ++r) {
            v[r] = o1 < n ? (uint32_t)a[o1] : 0u;
            for (size_t q = 0; q < nf; ++q)
                if (fl[q].site == ORC_SITE_CT_VAL && (uint64_t)fl[q].step == t0 && fl[q].replica == r && r < R)
                    v[r] = flip(v[r], fl[q].bit, 0xffffffffu);
            sum[r] += v[r];
        }
        local_sync32(c, sum);                                     /* store sum (its alloca) */
        (void)gep_offset(c, i, ls);                               /* if (array[i] != i)     :110 */
        for (unsigned r = 0; r < 3; ++r)
            cond[r] = (v[r < R ? r : 0] != i[r < R ? r : 0]) ? 1u : 0u;
        if (branch_cond(c, cond[0], cond[1], cond[2], 1)) {
            for (unsigned r = 0; r < 3; ++r)
                nerr[r] += 1;                                     /* numberOfErrors++       :111 */
            local_sync32(c, nerr);
            /* the report block (:114-131), its printing aside: `if (!first_error)`, for the first bad element `if (!in_block && ..)`
             * (in_block and local_errors are the program's globals: 0 when the call starts, in this batch model), and the
             * `array[i]` argument of the printf -- one more load offset */
            if (branch_cond(c, !first_error, !first_error, !first_error, bs)) {
                (void)branch_cond(c, !in_block, !in_block, !in_block, bs);
                first_error = 1, in_block = 1;
            }
            (void)gep_offset(c, i, ls);
            local_errors += 1;
            const uint32_t os = gep_offset(c, i, ss);             /* array[i] = i           :127 */
            uint32_t d[3] = {i[0], i[R > 1 ? 1 : 0], i[R > 2 ? 2 : 0]};
            store_sync32(c, d);
            if (os < n)
                a[os] = (int32_t)d[0];
            {
                uint32_t le[3] = {local_errors, local_errors, local_errors}; /* local_errors++ (a global: equal in every copy)  :128 */
                local_sync32(c, le);
            }
        }
        for (unsigned r = 0; r < 3; ++r)
            i[r] += 1;
        local_sync32(c, i);                                       /* i++ */
    }
    /* after the loop: `if (first_error && robust_printing)` (:139) and `if (sum != golden)` (:157) with golden = n (n - 1) / 2; a
     * wrong sum looks at `local_errors == 0` (:161) and, with no element error behind it, at `!in_block` (:165) */
    (void)branch_cond(c, first_error, first_error, first_error, bs);
    {
        const uint32_t golden = (uint32_t)(((uint64_t)n * (n - 1u)) / 2u);
        if (branch_cond(c, sum[0] != golden, sum[R > 1 ? 1 : 0] != golden, sum[R > 2 ? 2 : 0] != golden, bs))
            if (branch_cond(c, local_errors == 0u, local_errors == 0u, local_errors == 0u, bs)) {
                uint32_t se[3] = {1u, 1u, 1u};                    /* sum_errors++; local_errors++ (globals)      :162-163 */
                local_sync32(c, se);
                local_sync32(c, se);
                (void)branch_cond(c, !in_block, !in_block, !in_block, bs);
            }
    }
    uint32_t vs[3] = {sum[0], sum[R > 1 ? 1 : 0], sum[R > 2 ? 2 : 0]}, vn[3] = {nerr[0], nerr[R > 1 ? 1 : 0], nerr[R > 2 ? 2 : 0]};
    sync32(c, vs);        /* return value */
    store_sync32(c, vn);  /* stored to the caller's error count */
    *sum_out = (int32_t)vs[0];
    *nerr_out = vn[0];
}

void orc_cache_test_xmr(int32_t *arrays, uint32_t n, size_t narrays, int32_t *sums, uint32_t *nerrs, const orc_cfg *cfg,
                        const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t b = 0; b < narrays; ++b) {
        while (fp < nfaults && fs[fp].item < b)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == b)
            ++fe;
        c.detected = 0;
        if (cfg->flags & ORC_F_INDEXED)
            ct_item_indexed(arrays + (size_t)b * n, n, &sums[b], &nerrs[b], &c, fs + fp, fe - fp);
        else
            ct_item(arrays + (size_t)b * n, n, &sums[b], &nerrs[b], &c, fs + fp, fe - fp);
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[b] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* ------------------------------------------------------------------------------------------ */
/* default mode: vote where three (two) memory copies re-converge                             */
/* ------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------ */
/* quicksort                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* quick_sort (tests/quicksort/quicksort.c:109-129, the Rosetta-code Hoare partition), the way the reference writes it. */
static void qs_plain(int32_t *A, int len)
{
    if (len < 2)
        return;
    const int32_t pivot = A[len / 2];
    int i, j;
    for (i = 0, j = len - 1;; i++, j--) {
        while (A[i] < pivot)
            i++;
        while (A[j] > pivot)
            j--;
        if (i >= j)
            break;
        const int32_t temp = A[i];
        A[i] = A[j];
        A[j] = temp;
    }
    qs_plain(A, i);
    qs_plain(A + i, len - i);
}

void orc_quicksort_plain(int32_t *array, uint32_t n) { qs_plain(array, (int)n); }

/* The protected sort of one array.  Everything the function computes with is replica-private: i, j, the pivot, the loaded
 * values, the (base, len) of every pending call -- the recursion is an explicit per-replica stack walked in the reference's
 * order (left part first, then the right part).  The array is memory: one copy (-noMemReplication).  Sync points, the
 * reference's rule set for that mode applied to the source as written:
 *   every evaluated branch condition (`len < 2`, `A[i] < pivot`, `A[j] > pivot`, `i >= j`)   synchronization.cpp:146-155
 *   every GEP offset: A[len/2], A[i], A[j] loads (off with -noLoadSync), A[i], A[j] stores (off with -noStoreAddrSync)  :333-372
 *   the data of both stores of a swap (off with -noStoreDataSync)                                                     :197-224
 * `temp = A[i]` and the A[j] of `A[i] = A[j]` reuse the values the two scans loaded last (what -O3 leaves of them).
 * The replicas of an array always take the same direction (voted, or replica 0's under DWC), so control flow -- and with it the
 * condition counter that addresses the fault steps -- is one per array.  A corrupted index can leave the array: such loads
 * return the replica's pivot (both scans stop), such stores are dropped; a sort that does not end within 64 n + 1024
 * conditions, or nests deeper than ORC_QS_MAXDEPTH pending right parts, is cut (status WATCHDOG / STACK -- the reference's
 * supervisor files those runs under timeout / stack overflow, jsonParser.py:162-186). */
static int qs_item(int32_t *A, uint32_t n, sync_ctx *c, const orc_fault *fl, size_t nf)
{
    const unsigned R = c->nrep;
    const int ls = !(c->flags & ORC_F_NO_LOAD_SYNC), ss = !(c->flags & ORC_F_NO_STORE_ADDR_SYNC);
    uint32_t base[3] = {0, 0, 0}, len[3] = {n, n, n};
    uint32_t stk[3][ORC_QS_MAXDEPTH][2];
    uint32_t sp = 0, tick = 0;
    const uint32_t cap = 64u * n + 1024u;
    uint32_t i[3] = {0, 0, 0}, j[3] = {0, 0, 0}, pv[3] = {0, 0, 0}, vi[3] = {0, 0, 0}, vj[3] = {0, 0, 0};
#define QS_HOOK()                                                                                              \
    do {                                                                                                       \
        for (size_t q_ = 0; q_ < nf; ++q_)                                                                     \
            if (fl[q_].step == tick && fl[q_].replica < R) {                                                   \
                uint32_t *t_ = fl[q_].site == ORC_SITE_QS_I ? i : fl[q_].site == ORC_SITE_QS_J ? j :           \
                               fl[q_].site == ORC_SITE_QS_PIVOT ? pv : fl[q_].site == ORC_SITE_QS_VI ? vi :    \
                               fl[q_].site == ORC_SITE_QS_VJ ? vj : NULL;                                      \
                if (t_)                                                                                        \
                    t_[fl[q_].replica] = flip(t_[fl[q_].replica], fl[q_].bit, 0xffffffffu);                    \
            }                                                                                                  \
    } while (0)
#define QS_COND(expr0, expr1, expr2) (tick++, branch_cond(c, (expr0), R > 1 ? (expr1) : (expr0), R > 2 ? (expr2) : (expr0), 1))
#define QS_LOAD(dst, off)                                                                                      \
    do {                                                                                                       \
        const uint32_t o_ = (off);                                                                             \
        for (unsigned r_ = 0; r_ < 3; ++r_)                                                                    \
            dst[r_] = o_ < n ? (uint32_t)A[o_] : pv[r_];                                                       \
    } while (0)
    for (;;) {
        if (tick >= cap)
            return ORC_QS_WATCHDOG;
        QS_HOOK();
        if (QS_COND(len[0] < 2, len[1] < 2, len[2] < 2)) {     /* if (len < 2) return;                          :110 */
            if (sp == 0)
                return ORC_QS_OK;
            --sp;
            for (unsigned r = 0; r < 3; ++r) {
                base[r] = stk[r][sp][0];
                len[r] = stk[r][sp][1];
            }
            continue;
        }
        {
            const uint32_t mid[3] = {base[0] + len[0] / 2, base[1] + len[1] / 2, base[2] + len[2] / 2};
            const uint32_t po = gep_offset(c, mid, ls);         /* pivot = A[len / 2]                            :112 */
            for (unsigned r = 0; r < 3; ++r)
                pv[r] = po < n ? (uint32_t)A[po] : 0u;
        }
        for (unsigned r = 0; r < 3; ++r) {
            i[r] = base[r];
            j[r] = base[r] + len[r] - 1;
        }
        for (;;) {                                              /* for (i = 0, j = len - 1; ; i++, j--)          :115 */
            for (;;) {                                          /* while (A[i] < pivot) i++;                     :116 */
                QS_LOAD(vi, gep_offset(c, i, ls));
                QS_HOOK();
                if (!QS_COND((int32_t)vi[0] < (int32_t)pv[0], (int32_t)vi[1] < (int32_t)pv[1], (int32_t)vi[2] < (int32_t)pv[2]) ||
                    tick >= cap)
                    break;
                for (unsigned r = 0; r < 3; ++r)
                    i[r] += 1;
            }
            for (;;) {                                          /* while (A[j] > pivot) j--;                     :117 */
                QS_LOAD(vj, gep_offset(c, j, ls));
                QS_HOOK();
                if (!QS_COND((int32_t)vj[0] > (int32_t)pv[0], (int32_t)vj[1] > (int32_t)pv[1], (int32_t)vj[2] > (int32_t)pv[2]) ||
                    tick >= cap)
                    break;
                for (unsigned r = 0; r < 3; ++r)
                    j[r] -= 1;
            }
            QS_HOOK();
            if (QS_COND((int32_t)i[0] >= (int32_t)j[0], (int32_t)i[1] >= (int32_t)j[1], (int32_t)i[2] >= (int32_t)j[2]) ||
                tick >= cap)                                    /* if (i >= j) break;                            :119 */
                break;
            {                                                   /* temp = A[i]; A[i] = A[j]; A[j] = temp;   :121-123 */
                const uint32_t oi = gep_offset(c, i, ss);
                uint32_t d[3] = {vj[0], vj[1], vj[2]};
                store_sync32(c, d);
                if (oi < n)
                    A[oi] = (int32_t)d[0];
                const uint32_t oj = gep_offset(c, j, ss);
                uint32_t e[3] = {vi[0], vi[1], vi[2]};
                store_sync32(c, e);
                if (oj < n)
                    A[oj] = (int32_t)e[0];
            }
            for (unsigned r = 0; r < 3; ++r) {
                i[r] += 1;
                j[r] -= 1;
            }
        }
        if (tick >= cap)
            return ORC_QS_WATCHDOG;
        if (sp == ORC_QS_MAXDEPTH)
            return ORC_QS_STACK;
        for (unsigned r = 0; r < 3; ++r) {                      /* quick_sort(A, i); quick_sort(A + i, len - i); :126-127 */
            stk[r][sp][0] = i[r];
            stk[r][sp][1] = base[r] + len[r] - i[r];
            len[r] = i[r] - base[r];
        }
        ++sp;
    }
#undef QS_LOAD
#undef QS_COND
#undef QS_HOOK
}

void orc_quicksort_xmr(int32_t *arrays, uint32_t n, size_t narrays, const orc_cfg *cfg, const orc_fault *faults, size_t nfaults,
                       orc_stats *st, uint8_t *detected, uint8_t *status)
{
    orc_fault *fs = sorted_faults(faults, nfaults);
    sync_ctx c = {cfg->replicas, cfg->sync_every, st, 0, cfg->flags};
    size_t fp = 0;
    for (size_t a = 0; a < narrays; ++a) {
        while (fp < nfaults && fs[fp].item < a)
            ++fp;
        size_t fe = fp;
        while (fe < nfaults && fs[fe].item == a)
            ++fe;
        c.detected = 0;
        const int rc = qs_item(arrays + a * (size_t)n, n, &c, fs + fp, fe - fp);
        if (status)
            status[a] = (uint8_t)rc;
        if (c.detected) {
            st->dwc_detected += (cfg->replicas == 2);
            if (detected)
                detected[a] = 1;
        }
        fp = fe;
    }
    free(fs);
}

/* The exit vote of COAST's default (memory-replicated) mode over result arrays: docs/source/passes.rst:329,337 --
 * stores are not voted, values are where they leave the sphere of replication (synchronization.cpp:741-949,
 * verification.cpp:625-682).  Word-wise (32 bit): TMR vote + count (+ scrub of the copies), DWC compare. */
static int orc_eq(uint32_t a, uint32_t b, int fp)
{
    if (!fp)
        return a == b;
    float x, y; /* fcmp oeq: ordered and equal */
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    return x == y;
}
static int orc_ne(uint32_t a, uint32_t b, int fp)
{
    if (!fp)
        return a != b;
    float x, y; /* fcmp one: ordered and not equal -- false when either operand is a NaN */
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    return x < y || x > y;
}

/* The exit vote with the pass's operand-type rules.  Scalars (synchronization.cpp:1380-1443): cmp = eq(a, b), cmp2 = eq(a, c) with
 * `icmp eq` / `fcmp oeq` (:57-62, 70-88), vote = select(cmp, a, c) (:934-938), __SYNC_COUNT += 1, TMR_ERROR_CNT += !(cmp & cmp2).
 * Vectors of `vw` lanes (:1394-1396, 1469-1530): lane-wise select; TMR_ERROR_CNT += add-reduce over the lanes of
 * (a ne b) | (a ne c) with `icmp ne` / `fcmp one`; the function returns before the -countSyncs increment, so __SYNC_COUNT does not
 * move.  DWC: a word is flagged when !eq(a, b).  scrub: every copy that is not bitwise the voted value is rewritten (:527-529). */
void orc_sync_copies_typed(uint32_t *c0, uint32_t *c1, uint32_t *c2, int ncopies, size_t nwords, uint32_t *voted, int scrub,
                           orc_stats *st, uint8_t *detected, int fp, uint32_t vw)
{
    for (size_t w = 0; w < nwords; ++w) {
        if (vw <= 1 || ncopies == 2) /* (the pass counts syncs in its TMR path only, :1415; the engine's DWC count of compared */
            st->sync_count += 1;      /*  words is its own and does not depend on the operand kind)                          */
        if (ncopies == 3) {
            const int e01 = orc_eq(c0[w], c1[w], fp), e02 = orc_eq(c0[w], c2[w], fp);
            const uint32_t v = e01 ? c0[w] : c2[w];
            const int counted = vw > 1 ? (orc_ne(c0[w], c1[w], fp) || orc_ne(c0[w], c2[w], fp)) : !(e01 && e02);
            if (counted) {
                st->errors_corrected += 1;
                if (detected)
                    detected[w] = 1;
            }
            if (scrub && (c0[w] != v || c1[w] != v || c2[w] != v))
                c0[w] = c1[w] = c2[w] = v;
            if (voted)
                voted[w] = v;
        } else {
            if (!orc_eq(c0[w], c1[w], fp)) {
                st->dwc_detected += 1;
                if (detected)
                    detected[w] = 1;
            }
            if (voted)
                voted[w] = c0[w];
        }
    }
}

void orc_sync_copies(uint32_t *c0, uint32_t *c1, uint32_t *c2, int ncopies, size_t nwords, uint32_t *voted, int scrub,
                     orc_stats *st, uint8_t *detected)
{
    orc_sync_copies_typed(c0, c1, c2, ncopies, nwords, voted, scrub, st, detected, 0, 1);
}

#include "chaes_oracle.inc"
#include "crazycf_xmr.inc"
