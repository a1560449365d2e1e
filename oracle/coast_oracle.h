/*
 * coast_oracle.h -- CPU ORACLE for the COAST dataflowProtection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libcoast_hip.so) never
 * links, imports or falls back to anything in this directory.
 *
 * What it restates (all paths relative to the reference checkout):
 *   - the four benchmark kernels' arithmetic
 *       matrix_multiply   tests/mm_common/mm_common_tmr.c:3-20
 *       sha256_hash       tests/sha256_common/sha256_common_tmr.c:27-179
 *       aes_enc_dec       tests/aes/TI_aes_128.c:107-231
 *       crc16             tests/crc16/crc16.c:21-31
 *   - the TMR voter / corrected-fault counter / DWC comparator that the
 *     dataflowProtection pass inserts
 *       voter   vote(a,b,c) = (a==b) ? a : c      projects/dataflowProtection/synchronization.cpp:934-938, 512-522
 *       counter cnt += !((a==b)&&(a==c))          synchronization.cpp:1391-1443
 *       syncs   __SYNC_COUNT += 1 per counted vote synchronization.cpp:1415-1425
 *       DWC     a != b -> FAULT_DETECTED_DWC()    synchronization.cpp:1117-1192, 1299-1302
 *   - the fault model: one single-bit flip of a 32-bit datum
 *       flipOneBit  simulation/platform/resources/injector.py:202-207
 *
 * Parity pins: kernel arithmetic is pinned by the reference's own golden vectors
 * (mm xor_golden x3 sizes, sha256 x2 lengths, 568 AES KATs) and by oracle/_ref (the
 * reference C compiled unmodified).  Vote outcomes under faults and TMR_ERROR_CNT
 * values are "parity unpinned" by the reference (no reference test injects a fault and
 * asserts a count); they are pinned by the voter/counter rules above.
 */
#ifndef COAST_ORACLE_H
#define COAST_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fault sites (kernel specific).  Same numeric values as include/coast_hip.h. */
enum {
    ORC_SITE_MM_ACC = 0,   /* accumulator, before the MAC of k == step (step == n: after the loop) */
    ORC_SITE_MM_OPA = 1,   /* loaded f[i][k] of k == step */
    ORC_SITE_MM_OPB = 2,   /* loaded s[k][j] of k == step */
    ORC_SITE_MM_I = 3,     /* ORC_F_BRANCH_SYNC / ADDR_SYNC (the call is the item): loop counter i before loop condition number `step` */
    ORC_SITE_MM_J = 4,     /* ... j */
    ORC_SITE_MM_K = 5,     /* ... k (ORC_SITE_MM_ACC is `sum`, same timing, in that mode) */

    ORC_SITE_SHA_M = 8,    /* schedule word m[step%64] of compression step/64, right after it is produced */
    ORC_SITE_SHA_WV = 9,   /* working variable a..h (index 0..7) before round step%64 of compression step/64 */
    ORC_SITE_SHA_STATE = 10, /* ctx_state[index] before compression `step` (step == ncompress: before the digest) */
    ORC_SITE_SHA_DATALEN = 11, /* indexed mode: ctx_datalen before the loop condition of iteration `step` is evaluated */
    ORC_SITE_SHA_I = 12,       /* indexed mode: the byte loop's counter i, same timing */

    ORC_SITE_AES_STATE = 16, /* dword `index` (0..3) of the state at the start of main-loop round `step` (10: after loop) */
    ORC_SITE_AES_KEY = 17,   /* dword `index` (0..3) of the running round key, same timing */
    ORC_SITE_AES_ROUND = 18, /* ORC_F_BRANCH_SYNC / ADDR_SYNC: the loop counter `round` (8 bits) before loop condition `step` of the call */
    ORC_SITE_AES_I = 19,     /* the loop counter `i`, same timing */

    ORC_SITE_CRC_CRC = 24, /* crc register before byte `step` (step == length: after the loop) */
    ORC_SITE_CRC_X = 25,   /* temporary x of byte `step`, after x ^= x>>4 */
    ORC_SITE_CRC_LEN = 26, /* ORC_F_BRANCH_SYNC mode: the `length` register (8 bits live) before the loop condition of iteration `step` */

    ORC_SITE_CT_SUM = 32,  /* cache_test: running sum before element `step` is added (step == n: after the loop) */
    ORC_SITE_CT_VAL = 33,  /* the loaded array[step], right after the load */
    ORC_SITE_CT_NERR = 34, /* numberOfErrors before element `step` (step == n: after the loop) */
    ORC_SITE_CT_I = 35,    /* ORC_F_BRANCH_SYNC / ADDR_SYNC: the loop counter; `step` then counts evaluated loop conditions */

    ORC_SITE_CHSHA_W = 40,     /* CHStone sha: schedule word W[step%80] of transform step/80, right after it is produced */
    ORC_SITE_CHSHA_WV = 41,    /* working variable index 0..4 (A..E) before round step%80 of transform step/80 */
    ORC_SITE_CHSHA_DIGEST = 42, /* sha_info_digest[index] before transform `step` */
    ORC_SITE_CHSHA_I = 43,     /* ORC_F_BRANCH_SYNC / ADDR_SYNC: sha_transform's loop counter i before loop condition `step` of the call */
    ORC_SITE_CHSHA_COUNT = 44, /* sha_update's `count`, same timing */
    /* quicksort: `step` counts the branch conditions the sort has evaluated so far; the flip lands right before condition
     * number `step` is evaluated (after the load that feeds it) */
    ORC_SITE_QS_I = 48,     /* the left scan index i */
    ORC_SITE_QS_J = 49,     /* the right scan index j */
    ORC_SITE_QS_PIVOT = 50, /* the pivot value */
    ORC_SITE_QS_VI = 51,    /* the value last loaded from A[i] */
    ORC_SITE_QS_VJ = 52,    /* the value last loaded from A[j] */
    /* CHStone aes (chaes_oracle.inc) */
    ORC_SITE_CHAES_STATE = 64, /* packed state column `index` at round boundary `step` (0 entry; r: before the r-th ShiftRow/ByteSub; Nr+1: exit) */
    ORC_SITE_CHAES_WORD = 65,  /* expanded-key column `step`, right after KeySchedule produced it */
    /* ORC_F_BRANCH_SYNC / ADDR_SYNC (chaes_indexed.inc): a loop counter (32 bits live) of one replica before loop condition `step` of the call */
    ORC_SITE_CHAES_RND = 66,   /* encrypt's / decrypt's round counter `i` */
    ORC_SITE_CHAES_J = 67,     /* the running callee's `j` (KeySchedule, AddRoundKey, the two MixColumn functions) */
    ORC_SITE_CHAES_I = 68,     /* the running callee's `i` (KeySchedule, AddRoundKey_InversMixColumn) */
    /* crazyCF under -TMR / -DWC (crazycf_xmr.inc): a register of one replica before branch condition `step` of the run */
    ORC_SITE_CCF_I = 72,     /* main's i */
    ORC_SITE_CCF_TOTAL = 73, /* total */
    ORC_SITE_CCF_TIMES = 74, /* timesThroughWhile */
    ORC_SITE_CCF_FI = 75,    /* fillArray's i */
    /* control-flow signatures (cfcss_oracle.c): `step` = block transitions made so far */
    ORC_SITE_CFC_PC = 56,   /* the branch target of transition `step` */
    ORC_SITE_CFC_RTS = 57,  /* BasicBlockSignatureTracker between the store and the next check */
    ORC_SITE_CFC_RTSA = 58  /* RunTimeSignatureAdjuster, same timing */
};
enum { ORC_QS_OK = 0, ORC_QS_WATCHDOG = 1, ORC_QS_STACK = 2, ORC_QS_MAXDEPTH = 48 };

/* One single-bit flip.  16 bytes; identical layout to coast_fault in include/coast_hip.h. */
typedef struct {
    uint64_t item;    /* logical work item: mm b*n*n+i*n+j ; sha message ; aes block ; crc block */
    uint32_t step;    /* see site */
    uint8_t replica;  /* 0..nrep-1 */
    uint8_t site;     /* ORC_SITE_* */
    uint8_t bit;      /* 0..31, bit of the 32-bit register holding the value */
    uint8_t index;    /* word / variable index where the site has several */
} orc_fault;

typedef struct {
    uint64_t errors_corrected; /* TMR_ERROR_CNT analogue */
    uint64_t sync_count;       /* __SYNC_COUNT analogue */
    uint64_t dwc_detected;     /* items on which a DWC compare failed */
    uint64_t reserved;
} orc_stats;

typedef struct {
    uint32_t replicas;   /* 1 = unprotected, 2 = DWC, 3 = TMR */
    uint32_t sync_every; /* 0 = only at the mandatory sync points; V>0 = also every V steps (kernel specific) */
    uint32_t flags;      /* ORC_F_* */
} orc_cfg;
/* -noStoreDataSync (dataflowProtection.cpp:16; synchronization.cpp:197-224,324): the data of stores is not synchronised */
enum {
    ORC_F_NO_STORE_DATA_SYNC = 1u,
    /* loop / byte counters INSIDE the sphere of replication (default: outside, SURVEY 8a' last table row): */
    ORC_F_BRANCH_SYNC = 2u,        /* their branch conditions are voted (synchronization.cpp:146-155, 741-949) */
    ORC_F_ADDR_SYNC = 4u,          /* GEP offsets built from them are voted (:226-235, 333-372, 413-474) ... */
    ORC_F_NO_LOAD_SYNC = 8u,       /* ... except load addresses (-noLoadSync, :341-352) */
    ORC_F_NO_STORE_ADDR_SYNC = 16u, /* ... except store addresses (-noStoreAddrSync, :354-367) */
    /* the reference's memory-replicated mode with -storeDataSync (sha256 / aes / crc16): every array is `replicas` copies back
     * to back, replica r loads from copy r, the data of every store is voted, every replica stores the voted value to its copy */
    ORC_F_MEMORY_COPIES = 32u,
    /* with ORC_F_BRANCH_SYNC | ORC_F_ADDR_SYNC: the store-data votes the pass emits on the -O0 IR under -noMemReplication for the stores
     * the default schedules do not have -- every store of a computed value into one of the function's own locals (i++, sum += ..:
     * their allocas stay single-copy) and into state[] / key[] / ctx_data[] in place (synchronization.cpp:197-224, 476-561).  With it
     * sync_count = branches + GEP offsets + stores of tools/ir_sync_counts.py.  Off under -noStoreDataSync, like every data vote. */
    ORC_F_LOCAL_STORE_SYNC = 64u,
    /* sha256 with ORC_F_BRANCH_SYNC | ORC_F_ADDR_SYNC: the walk in the shape the x86 / lli flow hands the pass -- tests/sha256_common/
     * Makefile has no OPT_FLAGS, so sha256_hash and sha256_transform arrive as -O0 IR: the padding loops, the output loop and the three
     * loops of sha256_transform are loops with replica-private counters, every evaluated condition and every variable GEP offset a vote
     * (3-byte message: 198 branches, 387 load and 152 store offsets; tools/ir_sync_counts.py).  Without it the walk is the post--O3 shape
     * of the hifive1 flow (byte loop only).  ORC_F_LOCAL_STORE_SYNC adds this shape's 2009 + 116 stores. */
    ORC_F_O0_SHAPE = 128u
};
#define ORC_F_INDEXED (ORC_F_BRANCH_SYNC | ORC_F_ADDR_SYNC)

/* ---- plain (unprotected) restatements of the reference kernels ---- */
void orc_mm_plain(const uint32_t *f, const uint32_t *s, uint32_t *r, int n);
uint32_t orc_mm_xor(const uint32_t *r, int n); /* checkGolden's XOR reduce, mm_common_tmr.c:22-32 */
void orc_sha256_plain(const uint8_t *data, uint32_t len, uint8_t hash[32]);
void orc_aes128_plain(uint8_t state[16], uint8_t key[16], uint8_t dir);
uint16_t orc_crc16_plain(const uint8_t *data, uint32_t length);
const uint8_t *orc_aes_sbox(void);
const uint8_t *orc_aes_rsbox(void);

/* ---- replicated (TMR / DWC) semantic model with fault list ---- */
/* faults need not be sorted; `detected` (may be NULL) gets one byte per item: 1 = a sync point of that item saw
 * unequal copies (DWC: the item would have aborted; TMR: the item had a value corrected). */
void orc_mm_xmr(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, size_t batch, const orc_cfg *cfg,
                const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);
void orc_sha256_xmr(const uint8_t *msgs, size_t stride, uint32_t len, size_t nmsgs, uint8_t *digests,
                    const orc_cfg *cfg, const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);
void orc_aes128_xmr(uint8_t *states, uint8_t *keys, size_t nblocks, int dir, const orc_cfg *cfg,
                    const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);
void orc_crc16_xmr(const uint8_t *data, uint32_t block_len, size_t nblocks, uint16_t *crcs, const orc_cfg *cfg,
                   const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);

/* calc_sum (tests/cache_test/cacheTest.c:101-177): n_arrays arrays of n ints, scrubbed in place; per array the sum (taken
 * BEFORE the fixes, :108) and the number of elements that were != their index (:110-111) */
void orc_cache_test_plain(int32_t *array, uint32_t n, int32_t *sum, uint32_t *nerr);
/* quick_sort of tests/quicksort/quicksort.c:109-129, in place; status[a] = ORC_QS_* (NULL allowed) */
void orc_quicksort_plain(int32_t *array, uint32_t n);
void orc_quicksort_xmr(int32_t *arrays, uint32_t n, size_t narrays, const orc_cfg *cfg, const orc_fault *faults, size_t nfaults,
                       orc_stats *st, uint8_t *detected, uint8_t *status);
void orc_cache_test_xmr(int32_t *arrays, uint32_t n, size_t narrays, int32_t *sums, uint32_t *nerrs, const orc_cfg *cfg,
                        const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);

/* CHStone sha (tests/chstone/sha/sha.c; unittest/cfg/full.yml:5): sha_init + sha_update over `len` bytes + sha_final, the
 * five sha_info_digest words.  len must be a multiple of 64 (what sha_final supports, see the .c file). */
void orc_chsha_plain(const uint8_t *data, uint32_t len, uint32_t digest[5]);
void orc_chsha_xmr(const uint8_t *msgs, size_t stride, uint32_t len, size_t nmsgs, uint32_t *digests, const orc_cfg *cfg,
                   const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);

/* CHStone aes (tests/chstone/aes): Rijndael, type = key bits * 1000 + block bits; block b = 4 Nb bytes at states + 4 Nb b (in
 * place), its key = 4 Nk bytes at keys + 4 Nk b (left alone).  -1 = unknown type. */
int orc_chaes_plain(uint8_t *state, const uint8_t *key, int type, int dir);
int orc_chaes_xmr(uint8_t *states, const uint8_t *keys, size_t nblocks, int type, int dir, const orc_cfg *cfg,
                  const orc_fault *faults, size_t nfaults, orc_stats *st, uint8_t *detected);

/* ---- CFCSS (projects/CFCSS/CFCSS.cpp) and its test program tests/crazyCF/crazyCF.c: cfcss_oracle.c ---- */
enum { ORC_CFC_MAX_NODES = 256, ORC_CFC_MAX_SUCC = 1024, ORC_CFC_MAX_CALLS = 64 };
enum { ORC_CFC_FAN_IN = 1, ORC_CFC_CHECKED = 2, ORC_CFC_BUFFER = 4, ORC_CFC_SKIP = 8, ORC_CFC_RET = 16 };
enum { ORC_CFC_OK = 0, ORC_CFC_DETECTED = 1, ORC_CFC_WATCHDOG = 2, ORC_CFC_WILD = 3 };
typedef struct { /* same layout as coast_cfc_graph */
    uint32_t n_nodes;
    const uint8_t *flags;
    const uint16_t *func;
    const uint32_t *succ_begin;
    const uint16_t *succ;
    uint32_t n_calls;
    const uint16_t *call_node;
    const uint16_t *call_entry;
    uint32_t main_func;
} orc_cfc_graph;
typedef struct { /* same layout as coast_cfc_tables */
    uint32_t n_nodes, n_buffers;
    uint16_t sig[ORC_CFC_MAX_NODES], sig_diff[ORC_CFC_MAX_NODES], sig_adj[ORC_CFC_MAX_NODES];
    uint8_t flags[ORC_CFC_MAX_NODES];
    uint32_t succ_begin[ORC_CFC_MAX_NODES + 1];
    uint16_t succ[ORC_CFC_MAX_SUCC];
    uint16_t call_pre_adj[ORC_CFC_MAX_CALLS], call_post_adj[ORC_CFC_MAX_CALLS];
} orc_cfc_tables;
typedef struct {
    int32_t total, printed;
    uint32_t n_prints, blocks;
} orc_crazycf_result;
int orc_cfcss_assign(const orc_cfc_graph *in, orc_cfc_tables *out); /* calls srand(1): the unseeded libc state */
void orc_crazycf_graph(orc_cfc_graph *g);
void orc_crazycf_plain(int32_t seed, int32_t size, int32_t timesThroughWhile, orc_crazycf_result *res);
void orc_crazycf_run(const orc_cfc_tables *T, int cfcss, int32_t seed, int32_t size, int32_t times, uint64_t item,
                     const orc_fault *fl, size_t nf, orc_crazycf_result *res, uint8_t *status);
void orc_crazycf_batch(const orc_cfc_tables *T, int cfcss, const int32_t *params, size_t n, const orc_fault *fl, size_t nf,
                       orc_crazycf_result *res, uint8_t *status);
void orc_glibc_rand_seq(uint32_t seed, uint32_t *out, size_t k);
/* crazyCF under -TMR / -DWC (unittest/cfg/full_tmr.yml:8): params = n x (seed, size, timesThroughWhile); status 0 ok, 2 watchdog */
void orc_crazycf_xmr(const int32_t *params, size_t n, const orc_cfg *cfg, const orc_fault *faults, size_t nfaults, orc_stats *st,
                     orc_crazycf_result *res, uint8_t *status, uint8_t *detected);

/* sparse variants: evaluate only the listed items (used to check huge batches) */
void orc_mm_xmr_items(const uint32_t *f, const uint32_t *s, int n, const uint64_t *items, size_t nitems,
                      uint32_t *out, const orc_cfg *cfg, const orc_fault *faults, size_t nfaults, orc_stats *st,
                      uint8_t *detected);

/* ---- default (memory-replicated) mode: the exit vote over three (two) result copies, 32-bit words ---- */
void orc_sync_copies(uint32_t *c0, uint32_t *c1, uint32_t *c2, int ncopies, size_t nwords, uint32_t *voted, int scrub,
                     orc_stats *st, uint8_t *detected);
/* ... with the operand-type rules: fp = words are floats (fcmp oeq / one), vw > 1 = IR vectors of vw lanes (per-lane count, no
 * __SYNC_COUNT increment) */
void orc_sync_copies_typed(uint32_t *c0, uint32_t *c1, uint32_t *c2, int ncopies, size_t nwords, uint32_t *voted, int scrub,
                           orc_stats *st, uint8_t *detected, int fp, uint32_t vw);

/* ---- CPU-TMR baseline: default COAST mode (memory x3, loop-condition votes, -countErrors) ---- */
/* returns XOR-golden mismatch flag like checkGolden; *cnt gets TMR_ERROR_CNT, *syncs the dynamic vote count */
int orc_cpu_tmr_mm(const uint32_t *f, const uint32_t *s, uint32_t *r, int n, uint32_t xor_golden, uint32_t *cnt,
                   uint64_t *syncs);
/* the same on nthreads host threads, reps independent matrices each; returns wall seconds (< 0 on a wrong result) */
uint64_t orc_cpu_tmr_mm_campaign(const uint32_t *f, const uint32_t *s, int n, uint32_t xor_golden, const orc_fault *faults,
                                 size_t nfaults, size_t nruns, uint64_t *n_error, uint64_t *n_corrected, uint64_t *n_success);
double orc_cpu_tmr_mm_threads(const uint32_t *f, const uint32_t *s, int n, uint32_t golden, int nthreads, int reps);

#ifdef __cplusplus
}
#endif
#endif
