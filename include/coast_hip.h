/*
 * coast_hip.h -- C ABI of libcoast_hip.so: the MI355X (gfx950) redundant-execution engine that replaces
 * COAST's dataflowProtection passes (-TMR / -DWC) for the four benchmark kernels of the hot path.
 *
 * The reference has no run-time FFI: its boundary is source + symbol conventions.
 * Each entry point below names the reference interface it replaces (paths relative to the reference checkout).
 * All pointers named d_* are DEVICE pointers (HBM); everything else is host memory.  Calls are asynchronous
 * on the context's stream unless stated; only coast_read_stats() and the single-call shims synchronise.
 * Every function returns 0 on success or a negative COAST_E* code; coast_last_error() has the text.
 * There is no CPU fallback anywhere behind this interface.
 */
#ifndef COAST_HIP_H
#define COAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COAST_HIP_ABI_VERSION 8 /* 8: the staging loads of the matrix-core mm kernels are cloned BY DEFAULT (COAST_F_SINGLE_STAGING opts out), TMR side 256 runs mm_mfma_blk4_kernel;
                                 * 7: COAST_F_CLONE_STAGING, COAST_SITE_MM_PREG; 6: COAST_F_LOCAL_STORE_SYNC, COAST_F_O0_SHAPE; 2: coast_cfg.flags; 3: coast_stats.kernel_ms/.hbm_bytes, coast_launch_info, new flags/sites;
                                 * 4: control-flow signatures (coast_cfcss_assign, coast_crazycf_*), additive;
                                 * 5: COAST_REPLICA_ALL, COAST_ETIMEOUT, coast_launch_info.hooked_blocks, additive */

enum {
    COAST_OK = 0,
    COAST_EINVAL = -1, /* bad argument (NULL pointer, replicas not in {1,2,3}, size overflow ...) */
    COAST_EHIP = -2,   /* a HIP runtime call failed */
    COAST_ENODEV = -3, /* no gfx950 device */
    COAST_ENOMEM = -4,
    COAST_ETIMEOUT = -5 /* a single-call shim's region did not end: quicksort's watchdog or recursion-stack limit (the class the
                         * reference's supervisor files under timeout); the caller's array is left as it was */
};

/* Protection mode.  replicas = 3 is what `opt -TMR` selects (projects/TMR/TMR.cpp:33, DP.run(M,3)),
 * replicas = 2 is `opt -DWC` (projects/DWC/DWC.cpp:33), replicas = 1 runs the region unprotected.
 * The engine instantiates the reference's `-noMemReplication -countErrors -countSyncs` rule set
 * (dataflowProtection.cpp:14-18,37,46): one memory copy, store data and return values voted.  Where the replicas live:
 * in the lane-replicated kernels (sha256, aes, crc16, cache_test, CHStone sha / aes, quicksort, the VALU mm kernels and the
 * DWC / unprotected matrix-core mm kernel) replica r of work item q is lane replicas*q + r of a wavefront, so every vector
 * register of the item is replicated; in the TMR matrix-core mm kernels (side 256, the default) the replicas of an output
 * element are three accumulator blocks of ONE lane, each with its own A- and B-operand registers (every LDS operand load is
 * replicated, since round 4) and its own MFMAs; the raw words of f and s on their way into the LDS image are loaded twice and
 * compared (cloned loads, the default since ABI 8; COAST_F_SINGLE_STAGING below).  What stays a single copy -- the conversion
 * temporaries, address registers, lane constants -- is the analogue of a `__NO_xMR` value (tests/COAST.h:11): an upset there
 * is common-mode (COAST_REPLICA_ALL below).
 * Wave-uniform scalars (loop counters, lengths, base pointers) and LDS tables are outside the sphere of replication everywhere.  sync_every adds the reference's loop-condition sync points at a
 * chosen granularity (synchronization.cpp:146-155): mm = every V k-steps, crc16 = every V bytes,
 * aes = 1 -> after every round; 0 = mandatory sync points only. */
typedef struct coast_cfg {
    uint32_t replicas;
    uint32_t sync_every;
    uint32_t flags; /* COAST_F_*; 0 = the defaults above */
} coast_cfg;

/* The reference's remaining replication-rule flags (dataflowProtection.cpp:14-18,39-40; unittest/cfg/full.yml:18-36):
 *   -noStoreDataSync  COAST_F_NO_STORE_DATA_SYNC: store data is not voted / compared (synchronization.cpp:197-224,324);
 *                     replica 0's value is stored as it is and nothing is counted for it.  Loop-condition and return-value
 *                     sync points stay (mm: sync_every votes; crc16: every sync point -- its result is a return value).
 *                     The reference warns against combining it with -noMemReplication (interface.cpp:250-252): nothing then
 *                     protects the stored data.  It exists for the overhead studies and the clean-run matrix.
 *   -noLoadSync, -noStoreAddrSync   by default addresses are built from wave-uniform scalars and lane indices, which are
 *                     outside the sphere of replication in this design (SURVEY.md section 8a', last table row): no replicated
 *                     address exists and the engine behaves as if both were given.  COAST_F_ADDR_SYNC (below) puts the
 *                     counters of mm / sha256 / crc16 inside the SoR; there the two flags are real knobs.
 *   -storeDataSync    the lane-replicated engine always votes store data (the default here); in the memory-replicated mode it
 *                     is what coast_sync_copies(..., scrub = 1) does at the region exit.
 *   -i / -s           instruction interleaving vs segmenting: replicas are lanes of one instruction, there is no order. */
enum {
    COAST_F_NO_STORE_DATA_SYNC = 1u,
    /* Loop / byte counters INSIDE the sphere of replication (mm, sha256, aes128, crc16, cache_test, CHStone sha; the launch runs a stepwise kernel).  mm: the
     * work item becomes the CALL -- i, j, k and `sum` of matrix_multiply (mm_common_tmr.c:3-20) are replica-private registers of
     * one sequential walk per matrix, the three loop conditions are voted at every evaluation ((N+1)(N^2+N+1) votes, SURVEY.md
     * section 3.2) and so are the GEP offsets of f[i][k], s[k][j] (i, k, k, j: loads) and r[i][j] (i, j: store); fault sites
     * COAST_SITE_MM_I / _J / _K / _ACC, d_detected[b * n * n] is matrix b's flag.  By default such counters are wave-uniform scalars outside
     * the SoR (see -noLoadSync above); with these flags they are replica-private lane registers with their own fault sites,
     * and the reference's rules for them apply:
     *   COAST_F_BRANCH_SYNC         every evaluation of a branch condition on them is a sync point (synchronization.cpp:146-155,
     *                               741-949): sha256 `i < len`, `ctx_datalen == 64` per byte and `ctx_datalen < 56`
     *                               (sha256_common_tmr.c:119,122,132); crc16 `length--` per byte (crc16.c:25); cache_test
     *                               `i < data_array_elements` (cacheTest.c:107); aes128 every loop condition on `round` / `i`,
     *                               the tests of `dir` and the operands of the MixColumns condition (TI_aes_128.c:111-228).  Not set, the
     *                               branch follows the original instruction's operand = replica 0's copy.
     *   COAST_F_ADDR_SYNC           GEP offsets built from them are sync points (:226-235, 333-372, 413-474), the reference's
     *                               default under -noMemReplication: sha256 data[i] (a load address) and ctx_data[ctx_datalen]
     *                               (a store address), sha256_common_tmr.c:120.  crc16's `*data_p++` has a constant offset:
     *                               nothing to vote.  cache_test: array[i] of both loads and of the scrub store
     *                               (cacheTest.c:108,110,127), whose data is the counter itself.  aes128: state[i], key[i], key[i-4],
     *                               state[buf4 + c], Rcon[round] and the table lookups sbox[..] / rsbox[..] (data indices).  Not set, the access uses
     *                               replica 0's offset.
     *   COAST_F_NO_LOAD_SYNC        -noLoadSync: with ADDR_SYNC, load addresses are not voted (:341-352)
     *   COAST_F_NO_STORE_ADDR_SYNC  -noStoreAddrSync: with ADDR_SYNC, store addresses are not voted (:354-367)
     * The reference's `-TMR -noMemReplication` is BRANCH_SYNC | ADDR_SYNC. */
    COAST_F_BRANCH_SYNC = 2u,
    COAST_F_ADDR_SYNC = 4u,
    COAST_F_NO_LOAD_SYNC = 8u,
    COAST_F_NO_STORE_ADDR_SYNC = 16u,
    /* The reference's MEMORY-REPLICATED mode with -storeDataSync (dataflowProtection.cpp:14-18, synchronization.cpp:197-224;
     * VERDICT r2 missing 3), for sha256 / aes128 / crc16 on their lean kernels: every array argument of the entry point holds
     * `replicas` copies back to back (copy r starts r x the array's size after copy 0), replica r LOADS from its own copy, the data
     * of every store is voted, and every replica STORES the voted value into its own copy -- so an upset in one memory copy is
     * out-voted at the next store and the copies re-converge there (aes: state and key in place; sha256: the digests; crc16: the
     * result array).  One launch; 3x the memory traffic, as on the reference.  Without it (the default) memory is a single copy
     * (-noMemReplication).  Not combined with sync_every or the other flags. */
    COAST_F_MEMORY_COPIES = 32u,
    /* With COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC (mm, aes128, crc16, cache_test, CHStone sha and aes, crazyCF's main; sha256 with COAST_F_O0_SHAPE): the store-data votes the pass emits
     * on the -O0 IR under -noMemReplication for the stores the default schedules do not have -- every store of a computed value into
     * one of the function's own locals (i++, sum += ..: their allocas stay single-copy) and into state[] / key[] / W[] / the bit counts
     * in place (synchronization.cpp:197-224, 476-561).  With it coast_stats.sync_count of a call = executed conditional branches +
     * variable GEP offsets + stores of the reference's own clang -O0 IR (tools/ir_sync_counts.py; mm side 9: 5617), and an upset in a
     * counter or accumulator is out-voted at its next store.  Dropped by COAST_F_NO_STORE_DATA_SYNC, like every data vote. */
    COAST_F_LOCAL_STORE_SYNC = 64u,
    /* sha256 with COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC: the walk in the shape the x86 / lli flow hands the pass.  tests/sha256_common/
     * Makefile has no OPT_FLAGS, so sha256_hash and sha256_transform arrive as -O0 IR: the padding loops, the output loop and the three
     * loops of sha256_transform are loops with replica-private counters, every evaluated condition and every variable GEP offset a vote
     * (3-byte message: 198 branches, 387 load and 152 store offsets).  Without it the walk is the post--O3 shape of the hifive1 flow
     * (byte loop only).  With COAST_F_LOCAL_STORE_SYNC: + this shape's stores (2009 into locals, 116 into memory at 3 bytes). */
    COAST_F_O0_SHAPE = 128u,
    /* single-call host shims only (the batch entry points reject it): run the region in the reference's DEFAULT mode,
     * memory replicated as well -- one unprotected launch per memory copy + the exit vote of coast_sync_copies(scrub).
     * Without it the shims use the lane-replicated -noMemReplication engine. */
    COAST_F_HOST_MEMORY_REPLICATED = 0x100u,
    /* coast_mm_batch, side 256 on the matrix cores (the default engine and tiles), TMR / DWC: the global -> LDS staging loads are CLONED
     * (cloning.cpp:2187-2209; one address for the copies under -noMemReplication, :2247-2255) -- BY DEFAULT since ABI 8, as the pass
     * clones every load of a protected function with no opt-out short of __NO_xMR.  Every raw word of f and s is loaded a second time and
     * compared in front of its first use; TMR takes select(a == b, a, c) with a third load and counts one corrected error per word, DWC
     * counts a detected item and flags the first element the word reaches.  sync_count is unchanged (a cloned load is not a sync point of
     * the reference either; the votes stay where the stores are).  Price on an MI355X: + 6 % kernel time in the TMR kernel
     * (mm_mfma_blk4_kernel, profiles/r06_mm_blk4_ab.txt; + 10-12 % in mm_mfma_blk3_kernel), for the coverage of single-bit upsets drawn
     * uniformly from the wave's register state that docs/design/campaign.md tabulates (tools/campaign.py --reg-model uniform).
     *   COAST_F_CLONE_STAGING   (ABI 7's opt-in) is accepted and means the default.
     *   COAST_F_SINGLE_STAGING  opts out: every raw word is loaded ONCE into a staging register, converted once and written into the LDS
     *                           image all replicas read -- an upset of that register is common-mode, the wrong words come out with
     *                           TMR_ERROR_CNT unchanged (tests/test_gpu_parity.py::test_mm_physical_upsets_of_the_staging_registers).
     * Both are ignored where they have no meaning: other sides and engines (the lane-replicated kernels issue one load per replica lane
     * already), unprotected runs.  Neither selects the stepwise kernels the other flags select. */
    COAST_F_CLONE_STAGING = 0x200u,
    COAST_F_SINGLE_STAGING = 0x400u
};

/* Counters.  errors_corrected is TMR_ERROR_CNT (synchronization.cpp:269-294,1391-1443: +1 per voted value whose
 * copies are not all equal); sync_count is __SYNC_COUNT (:103-121,1415-1425); dwc_detected counts the work items on
 * which a DWC compare failed (each would have called FAULT_DETECTED_DWC(), :1299-1302). */
typedef struct coast_stats {
    uint64_t errors_corrected;
    uint64_t sync_count;
    uint64_t dwc_detected;
    uint64_t launches;
    double kernel_ms; /* GPU time of the protected launches since the last reset, measured with HIP events on the context's
                       * stream around every launch; 0 unless coast_set_profiling(ctx, 1) */
    double hbm_bytes; /* ALGORITHMIC HBM bytes of those launches (each input byte read once, each output byte written
                       * once; per-unit figures in DESIGN.md section 4) -- hbm_bytes / kernel_ms is the roofline numerator */
} coast_stats;

/* What the most recent protected launch on a context dispatched to (a debugging / test aid: e.g. that a faulted side-256
 * matrix_multiply was voted on the matrix cores and not handed to the stepwise VALU kernel). */
enum {
    COAST_ENGINE_NONE = 0,
    COAST_ENGINE_VALU = 1,        /* lean lane-replicated kernel (+ the stepwise kernel beside it for `general_blocks`) */
    COAST_ENGINE_MATRIX_CORE = 2, /* mm side 256: int8-MFMA limb kernel, injector hooks inside */
    COAST_ENGINE_STEPWISE = 3,    /* every workgroup in the stepwise kernel (sync_every != 0, flags, unaligned rows ...) */
    COAST_ENGINE_VOTE = 4         /* coast_sync_copies */
};
typedef struct coast_launch_info {
    uint32_t engine;         /* COAST_ENGINE_* */
    uint32_t hooked_blocks;  /* of fast_blocks: tiles / workgroups of the lean kernel that own an armed fault and applied it themselves */
    uint64_t general_blocks; /* workgroups / tiles run by a stepwise kernel (sync_every, flags, unaligned input) */
    uint64_t fast_blocks;    /* workgroups / tiles run by the lean kernel */
    uint64_t armed_faults;   /* single-bit flips the launch consumed */
    double algorithmic_bytes;
} coast_launch_info;

/* Fault sites, replacing the QEMU/GDB injector's "random register" targets
 * (simulation/platform/resources/injector.py:163-167,237-260). */
enum {
    COAST_SITE_MM_ACC = 0, /* accumulator before the MAC of k == step (step == n: after the loop) */
    COAST_SITE_MM_OPA = 1, /* loaded f[i][k], k == step */
    COAST_SITE_MM_OPB = 2, /* loaded s[k][j], k == step */
    COAST_SITE_MM_I = 3,   /* COAST_F_BRANCH_SYNC / ADDR_SYNC (the item is the call: coast_fault.item / n^2 names the matrix): loop counter
                            * i of a replica, flipped right before loop condition number `step` of the call is evaluated */
    COAST_SITE_MM_J = 4,   /* ... j */
    COAST_SITE_MM_K = 5,   /* ... k; COAST_SITE_MM_ACC is `sum` with the same timing in that mode */
    /* A PHYSICAL upset of the side-256 register-block kernel mm_mfma_blk3_kernel (rounds 4-5's TMR default; a launch that arms this site runs there): one bit of one lane of a named vector register is
     * flipped by a real exclusive-or while the kernel computes.  item = b n^2 + i n + j names the 64-row panel (i / 64), the wave's row half
     * (i / 32), the 16-row block (i / 16) and the 16-column tile (j / 16); replica the replica whose register it is; step = k-slab of the
     * tile (bits 1:0: k / 64) | lane << 8 | dword of the 4-dword fragment << 16 | register << 24 -- 0-3: the A-operand fragment of byte plane p
     * of that replica's set of ten MFMAs (flipped in front of the set; the next set reads its own), 4-7: the B-operand fragment of plane q of
     * that replica (flipped at the start of the slab's step, used by both row blocks), 8-11: the limb-sum accumulator t of (row block,
     * replica) (flipped at the start of the step, k-slabs 1-3: it stays until the tile's vote), 12-19: a raw word of s in the wave's staging
     * registers (round 0: 12-15, round 1: 16-19; dwords 0-1) and 20: a raw word of an f panel staged ahead (dwords 0-3) on their way into LDS --
     * state every replica reads: common-mode, no voter can see it; the word can belong to a LATER matrix of the workgroup (s: the next one
     * in a matrix's last steps; f: the next one, or the one after it for the piece requested in a matrix's last column tile); bit = the bit.  The effect is whatever the
     * hardware computes from the flipped register: TMR out-votes it, DWC flags the items it reaches, an unprotected run returns the wrong
     * words (tests/test_gpu_parity.py::test_mm_physical_register_upsets).  Rejected by every other mm engine. */
    COAST_SITE_MM_VGPR = 6,
    /* A physical upset of ANY register of a wave of a matrix-core kernel, named by its physical number (round 5): what the reference's injector
     * does when it draws a register of the core (simulation/platform/resources/injector.py:70-72, 237-260).  Hooked into the kernel the launch
     * would run anyway: mm_mfma_blk4_kernel (TMR default: 128-row panels, 32 steps per item, 60 slots), mm_mfma_blk3_kernel (DWC, unprotected,
     * COAST_MM_TILE=blocks3: 64-row panels, 16 steps, 20 x replicas slots) or mm_mfma_panel_kernel (COAST_MM_TILE=lanes: 64-row panels, a wave's
     * 48 / 56 steps, 20 slots).  item = b n^2 + i n names the matrix and the panel (i / panel rows: the workgroup); step = slot (bits 5:0: the
     * upset sits in front of that MFMA slot of the step) | step % 16 << 6 | lane << 10 (6 bits, vector registers) | wave of the workgroup << 16
     * (3 bits) | register file << 19 (0: vector, v0..v255; 1: scalar, s0..s101) | register number << 20 (9 bits) | step / 16 << 29 (2 bits; round
     * 6); bit = the bit.  The compiler knows nothing of the exclusive-or: accumulators, operand fragments, staging words and their clones, address
     * registers, lane constants, loop counters, descriptors -- whatever the allocator put there at that moment.  One upset per (wave, matrix).
     * Scalar upsets can send a descriptor or a kernel-argument pointer anywhere in the address space: a memory fault ends the process
     * (tools/campaign.py runs them in children).  Side 256, no sync_every / flags; replica is ignored.  (Round 5's kernel read the selector from
     * the wrong bits -- ADVICE r5; tests/test_gpu_parity.py::test_mm_preg_upset_lands_on_the_register_it_names pins the decode.) */
    COAST_SITE_MM_PREG = 7,
    COAST_SITE_SHA_M = 8,  /* schedule word m[step%64] of compression step/64, right after it is produced */
    COAST_SITE_SHA_WV = 9, /* working variable index 0..7 (a..h) before round step%64 of compression step/64 */
    COAST_SITE_SHA_STATE = 10, /* ctx_state[index] before compression `step` (== ncompress: before the digest) */
    COAST_SITE_SHA_DATALEN = 11, /* COAST_F_BRANCH_SYNC / ADDR_SYNC: ctx_datalen before the loop condition of byte-loop iteration `step` */
    COAST_SITE_SHA_I = 12,       /* the byte loop's counter i, same timing */
    COAST_SITE_AES_STATE = 16, /* state dword `index` at the start of main-loop round `step` (10: after the loop) */
    COAST_SITE_AES_KEY = 17,   /* running round-key dword `index`, same timing */
    COAST_SITE_AES_ROUND = 18, /* COAST_F_BRANCH_SYNC / ADDR_SYNC: aes_enc_dec's loop counter `round` (8 bits live) before loop condition
                                * `step` of the call (all loops of the function count) */
    COAST_SITE_AES_I = 19,     /* ... its loop counter `i`, same timing */
    COAST_SITE_CRC_CRC = 24,   /* crc register before byte `step` (== length: after the loop) */
    COAST_SITE_CRC_X = 25,     /* temporary x of byte `step` after x ^= x>>4 */
    COAST_SITE_CRC_LEN = 26,   /* COAST_F_BRANCH_SYNC: the `length` register (8 bits live) before the loop condition of iteration `step` */
    COAST_SITE_CT_SUM = 32,    /* cache_test: running sum before element `step` is added (step == n: after the loop) */
    COAST_SITE_CT_VAL = 33,    /* the loaded array[step], right after the load */
    COAST_SITE_CT_NERR = 34,   /* numberOfErrors before element `step` (step == n: after the loop) */
    COAST_SITE_CT_I = 35,      /* COAST_F_BRANCH_SYNC / ADDR_SYNC: calc_sum's loop counter i; `step` of all four cache_test sites then counts
                                * the loop conditions the call has evaluated (_CT_VAL: the element loaded in the iteration it entered) */
    COAST_SITE_CHSHA_W = 40,      /* CHStone sha: schedule word W[step%80] of transform step/80, right after it is produced */
    COAST_SITE_CHSHA_WV = 41,     /* working variable index 0..4 (A..E) before round step%80 of transform step/80 */
    COAST_SITE_CHSHA_DIGEST = 42, /* sha_info_digest[index] before transform `step` */
    COAST_SITE_CHSHA_I = 43,      /* COAST_F_BRANCH_SYNC / ADDR_SYNC: sha_transform's loop counter i before loop condition `step` of the call
                                   * (the W / A..E sites are those of the default schedule only) */
    COAST_SITE_CHSHA_COUNT = 44,  /* ... sha_update's `count`, same timing */
    /* quicksort: `step` counts the branch conditions the sort of this array has evaluated; the flip lands right before
     * condition number `step` is evaluated (after the load that feeds it) */
    COAST_SITE_QS_I = 48,     /* the left scan index i */
    COAST_SITE_QS_J = 49,     /* the right scan index j */
    COAST_SITE_QS_PIVOT = 50, /* the pivot */
    COAST_SITE_QS_VI = 51,    /* the value last loaded from A[i] */
    COAST_SITE_QS_VJ = 52,    /* the value last loaded from A[j] */
    /* CHStone aes: packed state column `index` (0 .. Nb-1; row r in byte r) at round boundary `step` -- 0: region entry,
     * r = 1 .. Nr: before the r-th (Invers)ShiftRow / ByteSub, Nr + 1: before the result is stored */
    COAST_SITE_CHAES_STATE = 64,
    COAST_SITE_CHAES_WORD = 65, /* expanded-key column `step` (word[0..3][step], packed), right after KeySchedule produced it */
    /* coast_chaes_batch under COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: a loop counter (32 bits live) of one replica, flipped right before
     * loop condition number `step` of the call reads it; STATE then lands before the `step`-th (Invers)ShiftRow call, WORD after the
     * `step`-th key-schedule column */
    COAST_SITE_CHAES_RND = 66, /* encrypt's / decrypt's round counter `i` (aes_enc.c:113, aes_dec.c:121) */
    COAST_SITE_CHAES_J = 67,   /* the running callee's `j` (KeySchedule, AddRoundKey, the two MixColumn functions) */
    COAST_SITE_CHAES_I = 68,   /* the running callee's `i` (KeySchedule, AddRoundKey_InversMixColumn) */
    /* coast_crazycf_xmr_batch: a register of one replica, flipped right before branch condition number `step` of the run reads it */
    COAST_SITE_CCF_I = 72,     /* main's i (crazyCF.c:42) */
    COAST_SITE_CCF_TOTAL = 73, /* total */
    COAST_SITE_CCF_TIMES = 74, /* timesThroughWhile */
    COAST_SITE_CCF_FI = 75,    /* fillArray's i (:22) */
    /* control-flow signatures (coast_crazycf_batch): `step` = how many block transitions the item has made; replica = 0 */
    COAST_SITE_CFC_PC = 56,   /* the branch target of transition `step`: execution lands at the START of block (target ^ 1<<bit) */
    COAST_SITE_CFC_RTS = 57,  /* BasicBlockSignatureTracker, after the leaving block stored it, before the next block checks it */
    COAST_SITE_CFC_RTSA = 58  /* RunTimeSignatureAdjuster, same timing */
};

/* One single-event upset: new = old XOR (1 << bit) on the 32-bit register holding the value
 * (FaultInjector.flipOneBit, injector.py:202-207).  replica = 0 .. replicas-1 hits that replica's private copy.
 * replica = COAST_REPLICA_ALL is a COMMON-MODE upset: the same flip in every replica's copy of the value -- what an upset of
 * state that the replicas of a lane group SHARE does (on the matrix-core mm engines: an MFMA A-operand fragment register, a raw
 * or converted word of `s` on its way into the LDS slab; everywhere: an LDS table word, a wave-uniform scalar).  No voter can see
 * it (all copies agree): expected outcome silent data corruption, as for a memory upset under -noMemReplication.  It is armed
 * as one flip per replica, so coast_launch_info.armed_faults counts `replicas` flips for it.  tools/campaign.py samples such
 * sites next to the replica-private ones, weighted by the register census of the kernel. */
#define COAST_REPLICA_ALL 255
typedef struct coast_fault {
    uint64_t item;   /* mm: b*n*n + i*n + j ; sha256: message ; aes: block ; crc16: block ; cache_test: array */
    uint32_t step;
    uint8_t replica; /* 0 .. replicas-1, or COAST_REPLICA_ALL */
    uint8_t site;    /* COAST_SITE_* */
    uint8_t bit;     /* 0..31 */
    uint8_t index;
} coast_fault;

typedef struct coast_ctx coast_ctx;

/* ---- context ---- */
int coast_create(coast_ctx **out, int device);
void coast_destroy(coast_ctx *ctx);
const char *coast_last_error(const coast_ctx *ctx);
int coast_abi_version(void);
/* content hash of the sources the library was compiled from (coast_amd/build.py compares it with the tree's) */
const char *coast_source_hash(void);
/* protected kernels run on `hip_stream` (a hipStream_t; NULL = the null stream) */
int coast_set_stream(coast_ctx *ctx, void *hip_stream);
/* Optional: totals are accumulated into the caller's device buffer of 4 x uint64
 * {errors_corrected, sync_count, dwc_detected, launches} (e.g. a tensor that RCCL all-reduces across GPUs,
 * replacing the single global TMR_ERROR_CNT of synchronization.cpp:1428-1431).  NULL restores the internal one. */
int coast_bind_counters(coast_ctx *ctx, uint64_t *d_totals);
/* fold the per-workgroup counter slots into the totals (async; coast_read_stats does it implicitly).  Launches nothing when no
 * protected launch has left counts in the slots since the last fold (with COAST_AES_FOLD=1 the persistent aes-128 kernels fold in
 * their own exit path). */
int coast_reduce_counters(coast_ctx *ctx);
/* Multi-GPU, C hosts: coast_reduce_counters + ncclAllReduce(SUM, 4 x uint64, in place) of the totals over `rccl_comm` (an
 * ncclComm_t the host created for this context's device: one rank per GPU), on the context's stream.  Afterwards every
 * rank's totals -- the bound buffer, or what coast_read_stats returns -- are the job-wide {errors_corrected, sync_count,
 * dwc_detected, launches}: the single global TMR_ERROR_CNT of synchronization.cpp:1428-1431, across GPUs.  Call it once per
 * batch and reset afterwards (the totals are now global sums).  librccl.so is loaded at first use. */
int coast_allreduce_counters(coast_ctx *ctx, void *rccl_comm);
int coast_read_stats(coast_ctx *ctx, coast_stats *out); /* synchronises the stream */
int coast_reset_stats(coast_ctx *ctx);
/* bracket every protected launch with HIP timing events on the context's stream -> coast_stats.kernel_ms.  enable = n > 1 (ABI 7): only
 * every n-th launch is bracketed and its time counted n times -- for launches of tens of microseconds, whose pair of event packets costs the
 * stream a fifth of the launch (profiles/r05_aes_step.txt); kernel_ms is then an estimate that is exact when the launches are alike. */
int coast_set_profiling(coast_ctx *ctx, int enable);
int coast_last_launch_info(const coast_ctx *ctx, coast_launch_info *out);

/* ---- on-device fault injector (replaces simulation/platform/supervisor.py + injector.py) ----
 * Arms `k` single-bit flips for the NEXT protected launch on this context.  The descriptor table is written on
 * a side stream and event-ordered before the consuming kernel; it is consumed by exactly one launch. */
int coast_inject_faults(coast_ctx *ctx, const coast_fault *faults, size_t k);

/* ---- protected regions (batch entry points) ---- */

/* matrix_multiply (tests/mm_common/mm_common_tmr.c:3-20; LANL variant tests/matrixMultiply/matrixMultiply.c:95-112):
 * `batch` independent n x n row-major uint32 products, r = (uint32) sum_k f[i][k]*s[k][j].
 * n == 256 (the benchmark's side) runs on the int8 matrix cores (exact signed-byte limb decomposition of the 32-bit
 * products; TMR: the three replicas in three accumulator blocks of one lane, voted in-lane; DWC / unprotected: replicas in
 * adjacent lanes; armed upsets are applied to the replica's accumulator and out-voted inside that kernel); every other side,
 * and n == 256 under COAST_MM_ENGINE=valu, on the VALU kernels.  Same words, counters and flags.  f, s, r: 16-byte aligned
 * when n is a multiple of 4.
 * d_detected (all batch entry points): optional, one byte per work item, set to 1 where a sync point of that item saw
 * unequal copies -- DWC: the compare that would have called FAULT_DETECTED_DWC(); TMR: a value was out-voted (the per-item
 * view of TMR_ERROR_CNT, what a campaign needs to classify a run as "fault corrected", jsonParser.py:162-186). */
int coast_mm_batch(coast_ctx *ctx, const uint32_t *d_f, const uint32_t *d_s, uint32_t *d_r, int n, size_t batch,
                   const coast_cfg *cfg, uint8_t *d_detected);

/* sha256_hash (tests/sha256_common/sha256_common_tmr.c:100-179): n_msgs messages of `len` bytes, message m at
 * d_msgs + m*stride; digests big-endian, 32 bytes each. */
int coast_sha256_batch(coast_ctx *ctx, const uint8_t *d_msgs, size_t stride, uint32_t len, size_t n_msgs,
                       uint8_t *d_digests, const coast_cfg *cfg, uint8_t *d_detected);

/* aes_enc_dec (tests/aes/TI_aes_128.c:107-231): n blocks, 16-byte state and 16-byte key each, both updated IN PLACE
 * exactly like the reference (encrypt leaves the last round key in `key`, decrypt restores the cipher key).
 * dir = 0 encrypt, != 0 decrypt. */
int coast_aes128_batch(coast_ctx *ctx, uint8_t *d_states, uint8_t *d_keys, size_t n, int dir, const coast_cfg *cfg,
                       uint8_t *d_detected);

/* crc16 (tests/crc16/crc16.c:21-31): n_blocks independent blocks of block_len bytes (the reference's `length` is an
 * unsigned char, so one call covers <= 255 bytes; larger streams are batches of blocks), crc per block. */
int coast_crc16_batch(coast_ctx *ctx, const uint8_t *d_data, uint32_t block_len, size_t n_blocks, uint16_t *d_crcs,
                      const coast_cfg *cfg, uint8_t *d_detected);

/* calc_sum (tests/cache_test/cacheTest.c:101-177, the memory-scrub benchmark of unittest/cfg/full.yml:13): n_arrays arrays
 * of n_elems ints, array a at d_arrays + a*n_elems.  Per array: d_sums[a] = the sum of its elements as found (:108),
 * d_nerrs[a] = how many were not equal to their index (:110-111); those are rewritten IN PLACE (:134).  Sync points: every
 * element's branch condition, the returned sum, the stored error count. */
int coast_cache_test_batch(coast_ctx *ctx, int32_t *d_arrays, uint32_t n_elems, size_t n_arrays, int32_t *d_sums,
                           uint32_t *d_nerrs, const coast_cfg *cfg, uint8_t *d_detected);

/* CHStone sha (tests/chstone/sha/sha.c: sha_init + sha_update + sha_final; unittest/cfg/full.yml:5): n_msgs messages of
 * `len` bytes (a multiple of 64 -- the only lengths the reference's sha_final pads as intended), message m at
 * d_msgs + m*stride; d_digests receives the five sha_info_digest words per message.  Not FIPS SHA-1 (no rotate in the
 * schedule, little-endian input words) -- bit-exact with the reference, its own test vector included. */
int coast_chsha_batch(coast_ctx *ctx, const uint8_t *d_msgs, size_t stride, uint32_t len, size_t n_msgs,
                      uint32_t *d_digests, const coast_cfg *cfg, uint8_t *d_detected);

/* CHStone aes (tests/chstone/aes; unittest/cfg/full.yml:6): encrypt (aes_enc.c:67-134) / decrypt (aes_dec.c:66-140) with
 * KeySchedule (aes_key.c:77-165) -- full Rijndael, `type` = key bits * 1000 + block bits, all nine combinations of
 * 128 / 192 / 256 the switch in KeySchedule knows (the benchmark's own main() runs 128128, aes.c:93-94).  n blocks: block b's
 * state = 4 Nb bytes at d_states + 4 Nb b (column-major like statemt[], updated IN PLACE), its key = 4 Nk bytes at
 * d_keys + 4 Nk b (left alone: KeySchedule expands it into replica-private storage).  dir = 0 encrypt, != 0 decrypt.
 * Sync points: the result block's stores (one vote per packed column); sync_every != 0 adds the state after every round.
 * Both arrays 4-byte aligned. */
int coast_chaes_batch(coast_ctx *ctx, uint8_t *d_states, const uint8_t *d_keys, size_t n, int type, int dir,
                      const coast_cfg *cfg, uint8_t *d_detected);

/* quick_sort (tests/quicksort/quicksort.c:109-129, the LANL quicksort benchmark): n_arrays arrays of n_elems ints, array a at
 * d_arrays + a*n_elems, sorted IN PLACE (ascending).  The first workload whose loop trip counts depend on the data: every
 * evaluated branch condition (`len < 2`, `A[i] < pivot`, `A[j] > pivot`, `i >= j`) is a sync point -- the replicas of an array
 * follow the voted direction and stay convergent -- and so are the GEP offsets of its loads / stores and the data of both
 * stores of a swap (the reference's -noMemReplication rule set; COAST_F_NO_LOAD_SYNC / _NO_STORE_ADDR_SYNC / _NO_STORE_DATA_SYNC
 * switch the three classes off, COAST_F_BRANCH_SYNC / _ADDR_SYNC are implied).  d_status (optional, one byte per array):
 * 0 = sorted to the end, 1 = cut by the watchdog (a corrupted index kept a loop alive: 64 n + 1024 conditions), 2 = more than
 * 48 pending parts (the reference's supervisor classes: timeout / stack overflow). */
enum { COAST_QS_OK = 0, COAST_QS_WATCHDOG = 1, COAST_QS_STACK = 2 };
int coast_quicksort_batch(coast_ctx *ctx, int32_t *d_arrays, uint32_t n_elems, size_t n_arrays, const coast_cfg *cfg,
                          uint8_t *d_detected, uint8_t *d_status);

/* ---- CFCSS: control-flow checking by software signatures (projects/CFCSS/CFCSS.cpp, docs/source/cfcss.rst) ----
 * The reference's second detector, a pass of its own (`opt -CFCSS`, tests/crazyCF/Makefile:3): every basic block gets a 16-bit
 * signature, a signature difference and -- for the predecessors of branch fan-in blocks -- a run-time adjuster; two globals
 * (BasicBlockSignatureTracker, RunTimeSignatureAdjuster) are stored at the end of every block and checked at the start of the
 * next; a mismatch calls FAULT_DETECTED_CFC() -> abort() (CFCSS.cpp:88-105, 107-126).
 * coast_cfcss_assign is the compile-time half (host only, no device work): the pass's signature generation for a control-flow
 * graph -- unseeded rand() % 65536 in ascending order over the blocks (:185-200, 218-240; the glibc sequence of the
 * reference's Linux hosts is restated inside), sigDiff / sigAdj (:438-457, 459-471), buffer blocks where the adjuster of a
 * block with several fan-in successors cannot serve them all (:348-436, docs/source/cfcss.rst "Modifications"), the call /
 * return handling of runOnModule (:551-643, 737-768).  The run-time half lives in the kernels: one tracker register pair per
 * lane, checked for the whole wave with one compare per block. */
enum { COAST_CFC_MAX_NODES = 256, COAST_CFC_MAX_SUCC = 1024, COAST_CFC_MAX_CALLS = 64 };
enum {
    COAST_CFC_FAN_IN = 1,  /* out: isBranchFanIn (CFCSS.cpp:234-236, 381) -- the check XORs the adjuster in */
    COAST_CFC_CHECKED = 2, /* out: the block starts with a signature check (successors of instrumented blocks, called entries) */
    COAST_CFC_BUFFER = 4,  /* out: an inserted buffer block */
    COAST_CFC_SKIP = 8,    /* in/out: an error-handler block of the pass's skipList (CFCSS.h:64-66): numbered, never instrumented */
    COAST_CFC_RET = 16     /* in/out: ends in a return */
};
typedef struct coast_cfc_graph {
    uint32_t n_nodes;           /* basic blocks in module order (function by function; CFCSS.cpp:154-183) */
    const uint8_t *flags;       /* per block: COAST_CFC_SKIP | COAST_CFC_RET */
    const uint16_t *func;       /* per block: index of its function */
    const uint32_t *succ_begin; /* n_nodes + 1 offsets into succ */
    const uint16_t *succ;       /* successors in terminator operand order (br: true, false; switch: default, cases) */
    uint32_t n_calls;           /* calls of functions defined in the module, in (block, position) order */
    const uint16_t *call_node;  /* the calling block */
    const uint16_t *call_entry; /* the callee's entry block */
    uint32_t main_func;         /* returns of this function are not tracked (CFCSS.cpp:173-178) */
} coast_cfc_graph;
typedef struct coast_cfc_tables {
    uint32_t n_nodes; /* input blocks + buffer blocks (appended) */
    uint32_t n_buffers;
    uint16_t sig[COAST_CFC_MAX_NODES];      /* stored into the tracker before the block's terminator */
    uint16_t sig_diff[COAST_CFC_MAX_NODES]; /* XORed with the tracker by the block's entry check */
    uint16_t sig_adj[COAST_CFC_MAX_NODES];  /* stored into the adjuster before the block's terminator */
    uint8_t flags[COAST_CFC_MAX_NODES];
    uint32_t succ_begin[COAST_CFC_MAX_NODES + 1];
    uint16_t succ[COAST_CFC_MAX_SUCC]; /* terminator operands after buffer insertion */
    uint16_t call_pre_adj[COAST_CFC_MAX_CALLS];  /* adjuster stored right before call c (after verifyCallSignatures, :645-690) */
    uint16_t call_post_adj[COAST_CFC_MAX_CALLS]; /* adjuster re-stored after call c returns (:620-626) */
} coast_cfc_tables;
int coast_cfcss_assign(const coast_cfc_graph *g, coast_cfc_tables *out);

/* crazyCF (tests/crazyCF/crazyCF.c, the reference's CFCSS test program: a for / switch / while / goto tangle over rand()).
 * One work item = one run of its main() with (seed, size, timesThroughWhile) taken from d_params instead of the constants
 * 42 / 20 / 10 (crazyCF.c:11,36,41); one lane per item, the wave executes the union of its lanes' paths block by block.
 * result.total = `total` when the run ended, .printed / .n_prints = the value and count of "total so far" lines (:54),
 * .blocks = block transitions made.  cfcss != 0 runs the program under the signatures of coast_crazycf_tables();
 * d_status: COAST_CFC_OK, _DETECTED (a check failed, or the jump landed in an error handler: FAULT_DETECTED_CFC -> abort),
 * _WATCHDOG (16 (size + times) + 256 transitions), _WILD (the jump left the program).  libc's srand / rand are the glibc
 * TYPE_3 generator, restated on the device. */
typedef struct coast_crazycf_params {
    int32_t seed, size, times;
} coast_crazycf_params;
typedef struct coast_crazycf_result {
    int32_t total, printed;
    uint32_t n_prints, blocks;
} coast_crazycf_result;
enum { COAST_CFC_OK = 0, COAST_CFC_DETECTED = 1, COAST_CFC_WATCHDOG = 2, COAST_CFC_WILD = 3 };
int coast_crazycf_graph(coast_cfc_graph *out);   /* the -O0 control-flow graph of crazyCF.c as this library encodes it */
int coast_crazycf_tables(coast_cfc_tables *out); /* = coast_cfcss_assign(coast_crazycf_graph) */
int coast_crazycf_batch(coast_ctx *ctx, const coast_crazycf_params *d_params, size_t n, coast_crazycf_result *d_results,
                        uint8_t *d_status, int cfcss);

/* The same program under -TMR / -DWC (unittest/cfg/full_tmr.yml:8 runs crazyCF with `-TMR`): NREP lanes per run, main's i, total,
 * timesThroughWhile and fillArray's i replica-private; `size` (a global), srand / rand / printf (library calls, rand's value fans out)
 * single.  Sync points: the arguments of the two printf calls (`total`) always -- values handed to an unprotected call,
 * synchronization.cpp:951-1100 --; COAST_F_BRANCH_SYNC the three loop conditions, the operand of `switch (i)` and main's return value;
 * COAST_F_ADDR_SYNC the offset of `array[i] = ..`; COAST_F_LOCAL_STORE_SYNC (with the two) the data of every store of a computed value.
 * With all three a run of the program's constants (42, 20, 10) has the 245 sync points of the reference's -O0 IR
 * (tools/ir_sync_counts.py crazycf).  result.total / .printed = what the printf calls received (TMR: the voted value, DWC: replica
 * 0's), .blocks = branch conditions evaluated; d_status: COAST_CFC_OK or COAST_CFC_WATCHDOG; d_detected (optional) one byte per run.
 * Fault sites COAST_SITE_CCF_*.  sync_every and -noLoadSync are rejected (nothing to act on). */
int coast_crazycf_xmr_batch(coast_ctx *ctx, const coast_crazycf_params *d_params, size_t n, coast_crazycf_result *d_results,
                            uint8_t *d_status, const coast_cfg *cfg, uint8_t *d_detected);

/* ---- default-mode TMR / DWC: memory replicated as well (docs/source/passes.rst:329,337; cloning.cpp:2417-2462) ----
 * In COAST's default mode the clones of a region run on their own copies of the data and stores are not voted
 * (synchronization.cpp:211-215); values are voted where the copies re-converge -- return values, arguments of unprotected
 * calls, stores to unprotected globals (synchronization.cpp:741-949, verification.cpp:625-682).  Here: run the batch
 * entry points with replicas = 1 once per memory copy, then call coast_sync_copies on the result arrays.
 *   ncopies = 3: vote = (a==b)?a:c per 32-bit word, errors_corrected += 1 per word whose copies differ, sync_count += 1
 *                per word; `scrub` != 0 writes the voted word back into the copies (they re-converge, :527-529).
 *   ncopies = 2: DWC compare, dwc_detected += 1 per differing word.
 * d_voted (optional) receives the voted single copy; d_detected (optional) one byte per word.  nbytes % 4 == 0,
 * all arrays 16-byte aligned.  Memory-resident upsets -- uncorrectable in the -noMemReplication mode of the lane-replicated
 * kernels -- are corrected here, at 3x the memory traffic. */
int coast_sync_copies(coast_ctx *ctx, void *const *d_copies, int ncopies, size_t nbytes, void *d_voted, int scrub,
                      uint8_t *d_detected);
/* The same vote with the pass's operand-type rules (synchronization.cpp:57-62, 70-88, 1380-1443, 1469-1530):
 *   elem = COAST_ELEM_F32   the words are floats, compared with `fcmp oeq` -- a NaN equals nothing (itself included), -0.0 == +0.0;
 *   vector_width > 1        the words are IR vectors of that many lanes: lane-wise select; TMR_ERROR_CNT += the number of lanes
 *                           with (a ne b) | (a ne c) under `icmp ne` / `fcmp one` (a NaN lane is not counted); __SYNC_COUNT is NOT
 *                           incremented (the vector path returns before the -countSyncs code, :1394-1396).
 * scrub re-converges every copy that is not bitwise the voted value.  coast_sync_copies = (COAST_ELEM_U32, 1). */
enum { COAST_ELEM_U32 = 0, COAST_ELEM_F32 = 1 };
int coast_sync_copies_typed(coast_ctx *ctx, void *const *d_copies, int ncopies, size_t nbytes, void *d_voted, int scrub,
                            uint8_t *d_detected, int elem, uint32_t vector_width);
/* injectFaultMem (simulation/platform/resources/injector.py:209-235): flip bit `bit` (0..7) of one byte of device memory,
 * ordered on the context's stream */
int coast_flip_memory(coast_ctx *ctx, void *d_ptr, size_t byte_offset, unsigned bit);

/* ---- single-call host shims with the reference's data contract (host pointers, synchronous) ---- */
/* matrix_multiply's `side` is a macro in the reference (mm_tmr.c:10), so the glue TU passes it explicitly */
int coast_matrix_multiply_host(const uint32_t *f, const uint32_t *s, uint32_t *r, int side, const coast_cfg *cfg);
int coast_sha256_host(const uint8_t *data, uint32_t len, uint8_t hash[32], uint32_t state_out[8], const coast_cfg *cfg);
int coast_aes_enc_dec_host(uint8_t *state, uint8_t *key, uint8_t dir, const coast_cfg *cfg);
int coast_crc16_host(const uint8_t *data, uint32_t length, uint16_t *crc, const coast_cfg *cfg);
/* calc_sum's `data_array_elements` is a macro in the reference (cacheTest.c:78), so the glue TU passes it explicitly */
int coast_cache_test_host(int32_t *array, uint32_t n_elems, int32_t *sum, uint32_t *nerr, const coast_cfg *cfg);
int coast_chsha_host(const uint8_t *data, uint32_t len, uint32_t digest[5], const coast_cfg *cfg);
int coast_quicksort_host(int32_t *array, uint32_t n_elems, const coast_cfg *cfg);
/* CHStone encrypt / decrypt on one block: state 4 Nb bytes (in place), key 4 Nk bytes */
int coast_chaes_host(uint8_t *state, const uint8_t *key, int type, int dir, const coast_cfg *cfg);
/* arm single-bit flips for the NEXT single-call shim (they run on a library-owned context): lets an external harness
 * inject into an unmodified driver the way supervisor.py + GDB inject into the running benchmark */
int coast_host_inject_faults(const coast_fault *faults, size_t k);
/* counters accumulated by the host shims since the last call (what TMR_ERROR_CNT / __SYNC_COUNT expose) */
int coast_host_stats(coast_stats *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* COAST_HIP_H */
