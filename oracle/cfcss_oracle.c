/*
 * cfcss_oracle.c -- CPU ORACLE for the control-flow-signature detector.  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see
 * coast_oracle.h): only tests/, __graft_entry__.smoke() and tools that check results may load it.
 *
 * Restates (paths relative to the reference checkout):
 *   - the CFCSS pass's static half, projects/CFCSS/CFCSS.cpp: populateGraph (:154-183), generateSignatures (:185-200),
 *     sortGraph (:218-240), calcSigDiff (:438-457), sigDiffGen (:459-471), verifySignatures (:348-404), insertBufferBlock
 *     (:306-346), and the constants runOnModule freezes into the inserted instructions (:737-768 with updateCallInsts
 *     :585-631, verifyCallSignatures :645-690, updateRetInsts :692-706).  Kept in the pass's own shape: an array of BBNode
 *     with edge lists, walked the way the pass walks its `graph` vector, drawing from libc's rand() after srand(1) -- the
 *     state an unseeded program has;
 *   - its run-time half: the stores of insertStoreInsts (:494-506), the compare of insertCompInsts (:508-549), the branch to
 *     the error block of splitBlocks (:708-731) -> FAULT_DETECTED_CFC() -> abort() (:88-105);
 *   - the program the reference tests the pass on, tests/crazyCF/crazyCF.c (Makefile: OPT_PASSES = -CFCSS), as a walk over
 *     its -O0 basic blocks with the three constants (srand(42), size = 20, timesThroughWhile = 10) as parameters;
 *   - libc srand()/rand(): glibc's TYPE_3 generator (stdlib/random_r.c), checked against this host's libc in the tests.
 *
 * Parity pins: the program's arithmetic (total, the "total so far" line) is pinned by oracle/_ref -- crazyCF.c itself
 * compiled unmodified, run for a grid of seeds and sizes (tests/golden/gen_golden.py).  The SIGNATURE VALUES and the
 * detection outcomes under upsets are PARITY UNPINNED: the pass needs LLVM 7.0 to run (projects/CMakeLists.txt:11) and the
 * reference holds no expected output for it; they are pinned by the algorithm's own invariant (every legal edge checks clean,
 * docs/source/cfcss.rst) and by the rules cited above.
 */
#include "coast_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- glibc random (TYPE_3) */
typedef struct {
    int32_t ring[34]; /* the textbook unrolled form r[i] = r[i-31] + r[i-3]: a sliding window of the last 34 values */
    int n;
} orc_glibc_rand;

/* o_k = r[k+344] where r[0..30] come from the Lehmer step, r[31..33] = r[0..2], r[i] = r[i-31] + r[i-3]; result = o_k >> 1.
 * Kept as a sliding window of the last 34 values (a different formulation from the product's in-place 31-word ring). */
void orc_glibc_srand(orc_glibc_rand *g, uint32_t seed)
{
    int64_t w;
    int i;
    if (seed == 0)
        seed = 1;
    g->ring[0] = (int32_t)seed;
    w = (int64_t)(int32_t)seed; /* glibc keeps `word` in an int32_t: seeds >= 2^31 start negative */
    for (i = 1; i < 31; ++i) {
        w = (16807 * (w % 127773)) - (2836 * (w / 127773));
        if (w < 0)
            w += 2147483647;
        g->ring[i] = (int32_t)w;
    }
    for (i = 31; i < 34; ++i)
        g->ring[i] = g->ring[i - 31];
    g->n = 34;
    for (i = 34; i < 344; ++i) {
        const uint32_t v = (uint32_t)g->ring[(g->n - 31) % 34] + (uint32_t)g->ring[(g->n - 3) % 34];
        g->ring[g->n % 34] = (int32_t)v;
        g->n += 1;
    }
}

uint32_t orc_glibc_rand_next(orc_glibc_rand *g)
{
    const uint32_t v = (uint32_t)g->ring[(g->n - 31) % 34] + (uint32_t)g->ring[(g->n - 3) % 34];
    g->ring[g->n % 34] = (int32_t)v;
    g->n += 1;
    if (g->n >= 34 * 1000000)
        g->n -= 34 * 999999;
    return v >> 1;
}

/* test hook: k outputs after srand(seed) */
void orc_glibc_rand_seq(uint32_t seed, uint32_t *out, size_t k)
{
    orc_glibc_rand g;
    orc_glibc_srand(&g, seed);
    for (size_t q = 0; q < k; ++q)
        out[q] = orc_glibc_rand_next(&g);
}

/* ---------------------------------------------------------------- the pass: static half */
#define MAXN ORC_CFC_MAX_NODES
#define MAXE 16

typedef struct {
    unsigned short sig, sigDiff, sigAdj;
    int isBranchFanIn, isBuffer, skip, ret, func;
    int nEdges, edgeNums[MAXE];   /* BBNode::edgeNums */
    int nTerm, termSucc[MAXE];    /* the terminator's successor operands */
    /* frozen constants */
    int hasComp;
    unsigned short compDiff, storeAdj;
    int compFanIn;
} BBNode;

typedef struct {
    BBNode graph[MAXN];
    int size;
    unsigned char sigSeen[65536]; /* the std::set<unsigned short> `signatures` */
    int sigCount;
    int fixBranchCount;
} Pass;

static unsigned short calcSigDiff(Pass *P, int pred, int succ) /* :438-457 */
{
    BBNode *pn = &P->graph[pred], *sn = &P->graph[succ];
    unsigned short sd;
    if (sn->skip)
        sd = 0;
    else if (sn->sigDiff == 0)
        sd = pn->sig ^ sn->sig;
    else {
        sd = sn->sigDiff;
        pn->sigAdj = pn->sig ^ sn->sigDiff ^ sn->sig;
    }
    return sd;
}

static int insertBufferBlock(Pass *P, int pred, int succ) /* :306-346 with updateBranchInst / updateEdgeNums */
{
    BBNode *b;
    int bi = P->size, i;
    unsigned short newSig = 0;
    int want = P->sigCount + 1;
    if (bi >= MAXN)
        return -1;
    b = &P->graph[bi];
    memset(b, 0, sizeof *b);
    P->size += 1;
    b->func = P->graph[pred].func;
    while (P->sigCount < want) { /* getSingleSig :292-300 */
        newSig = (unsigned short)(rand() % 65536);
        if (!P->sigSeen[newSig]) {
            P->sigSeen[newSig] = 1;
            P->sigCount += 1;
        }
    }
    b->sig = newSig;
    b->sigDiff = calcSigDiff(P, pred, bi);
    P->graph[succ].sigDiff = calcSigDiff(P, bi, succ);
    b->isBuffer = 1;
    for (i = 0; i < P->graph[pred].nTerm; ++i)
        if (P->graph[pred].termSucc[i] == succ) {
            P->graph[pred].termSucc[i] = bi;
            break;
        }
    b->termSucc[b->nTerm++] = succ;
    P->fixBranchCount += 1;
    P->graph[succ].isBranchFanIn = 1;
    /* updateEdgeNums: the old edge is erased, the new one appended */
    for (i = 0; i < P->graph[pred].nEdges; ++i)
        if (P->graph[pred].edgeNums[i] == succ)
            break;
    if (i < P->graph[pred].nEdges) {
        memmove(&P->graph[pred].edgeNums[i], &P->graph[pred].edgeNums[i + 1],
                sizeof(int) * (size_t)(P->graph[pred].nEdges - i - 1));
        P->graph[pred].nEdges -= 1;
    }
    P->graph[pred].edgeNums[P->graph[pred].nEdges++] = bi;
    b->edgeNums[b->nEdges++] = succ;
    return bi;
}

static int verifySignatures(Pass *P) /* :348-404; 1 = verified, 0 = a buffer went in, -1 = out of room */
{
    int n, k;
    for (n = 0; n < P->size; ++n) {
        BBNode *bn = &P->graph[n];
        for (k = 0; k < bn->nEdges; ++k) {
            BBNode *ch = &P->graph[bn->edgeNums[k]];
            unsigned short XOR1, XOR2;
            if (ch->skip)
                continue;
            XOR1 = bn->sig ^ ch->sigDiff;
            XOR2 = XOR1 ^ bn->sigAdj;
            if (bn->isBranchFanIn && XOR1 == ch->sig && bn->sigAdj != 0 && !ch->isBuffer)
                return insertBufferBlock(P, n, bn->edgeNums[k]) < 0 ? -1 : 0;
            else if (XOR2 != ch->sig && !ch->isBuffer)
                return insertBufferBlock(P, n, bn->edgeNums[k]) < 0 ? -1 : 0;
        }
    }
    return 1;
}

int orc_cfcss_assign(const orc_cfc_graph *in, orc_cfc_tables *out)
{
    static Pass Pst; /* large: kept off the stack; the oracle is single-threaded */
    Pass *P = &Pst;
    int n = (int)in->n_nodes, i, k, c, r;
    int indeg[MAXN];
    int visited[MAXN];
    unsigned short preAdj[ORC_CFC_MAX_CALLS], postAdj[ORC_CFC_MAX_CALLS];
    int retSeen[MAXN];
    unsigned short retAdj[MAXN];
    unsigned short sorted[MAXN];
    if (n <= 0 || n > MAXN || in->n_calls > ORC_CFC_MAX_CALLS)
        return -1;
    memset(P, 0, sizeof *P);
    memset(indeg, 0, sizeof indeg);
    P->size = n;
    for (i = 0; i < n; ++i) {
        BBNode *bn = &P->graph[i];
        bn->skip = (in->flags[i] & ORC_CFC_SKIP) != 0;
        bn->ret = (in->flags[i] & ORC_CFC_RET) != 0;
        bn->func = in->func[i];
        for (k = (int)in->succ_begin[i]; k < (int)in->succ_begin[i + 1]; ++k) {
            if (bn->nEdges >= MAXE)
                return -1;
            bn->edgeNums[bn->nEdges++] = in->succ[k];
            bn->termSucc[bn->nTerm++] = in->succ[k];
            indeg[in->succ[k]] += 1;
        }
    }
    /* generateSignatures (:185-200): unseeded rand() */
    srand(1);
    {
        int loopNum = 0;
        while (P->sigCount < n) {
            unsigned short sig = (unsigned short)(rand() % 65536);
            if (sig != 0) {
                if (!P->sigSeen[sig]) {
                    P->sigSeen[sig] = 1;
                    P->sigCount += 1;
                }
                loopNum++;
            }
            if (loopNum > 32000)
                break;
        }
        if (P->sigCount < n)
            return -1;
    }
    /* sortGraph (:218-240): the set iterates in ascending order */
    k = 0;
    for (i = 1; i < 65536 && k < n; ++i)
        if (P->sigSeen[i])
            sorted[k++] = (unsigned short)i;
    for (i = 0; i < n; ++i) {
        P->graph[i].sig = sorted[i];
        if (indeg[i] != 1)
            P->graph[i].isBranchFanIn = 1;
    }
    /* sigDiffGen (:459-471) */
    for (i = 0; i < n; ++i)
        for (k = 0; k < P->graph[i].nEdges; ++k) {
            int e = P->graph[i].edgeNums[k];
            P->graph[e].sigDiff = calcSigDiff(P, i, e);
        }
    for (;;) {
        int v = verifySignatures(P);
        if (v < 0)
            return -2;
        if (v)
            break;
    }
    /* runOnModule (:737-768) */
    memset(visited, 0, sizeof visited);
    memset(retSeen, 0, sizeof retSeen);
    memset(preAdj, 0, sizeof preAdj);
    memset(postAdj, 0, sizeof postAdj);
    memset(retAdj, 0, sizeof retAdj);
    for (i = 0; i < P->size; ++i) {
        BBNode *bn = &P->graph[i];
        if (bn->skip)
            continue;
        bn->storeAdj = bn->sigAdj; /* insertStoreInsts(bn, ..., terminator) */
        for (k = 0; k < bn->nEdges; ++k) {
            BBNode *ch = &P->graph[bn->edgeNums[k]];
            if (!visited[bn->edgeNums[k]] && !ch->skip) { /* insertCompInsts */
                ch->hasComp = 1;
                ch->compDiff = ch->sigDiff;
                ch->compFanIn = ch->isBranchFanIn;
                visited[bn->edgeNums[k]] = 1;
            }
        }
        for (c = 0; c < (int)in->n_calls; ++c) { /* bn->callList */
            BBNode *fn;
            if (in->call_node[c] != i)
                continue;
            fn = &P->graph[in->call_entry[c]];
            if (fn->skip)
                continue;
            /* updateCallInsts (:585-631) */
            fn->sigDiff = calcSigDiff(P, i, in->call_entry[c]);
            for (r = 0; r < n; ++r)
                if (P->graph[r].ret && P->graph[r].func == fn->func && P->graph[r].func != (int)in->main_func) {
                    bn->sigDiff = calcSigDiff(P, r, i);
                    if (!retSeen[r]) { /* std::map::insert keeps the first */
                        retSeen[r] = 1;
                        retAdj[r] = P->graph[r].sigAdj;
                    }
                }
            preAdj[c] = bn->sigAdj;  /* insertStoreInsts(callBB, ..., callI) */
            postAdj[c] = bn->sigAdj; /* multipleFunctionCalls holds every callee (:165-171): stores again, no compare */
            fn->hasComp = 1;
            fn->compDiff = fn->sigDiff;
            fn->compFanIn = fn->isBranchFanIn;
            visited[in->call_entry[c]] = 1;
        }
    }
    for (c = 0; c < (int)in->n_calls; ++c) { /* verifyCallSignatures (:645-690) */
        BBNode *cb = &P->graph[in->call_node[c]], *fn = &P->graph[in->call_entry[c]];
        unsigned short XOR2;
        if (cb->skip || fn->skip)
            continue;
        XOR2 = cb->sig ^ fn->sigDiff ^ preAdj[c];
        if (XOR2 != fn->sig)
            preAdj[c] = cb->sig ^ fn->sigDiff ^ fn->sig;
    }
    for (r = 0; r < n; ++r) /* updateRetInsts (:692-706) */
        if (retSeen[r])
            P->graph[r].storeAdj = retAdj[r];

    memset(out, 0, sizeof *out);
    out->n_nodes = (uint32_t)P->size;
    out->n_buffers = (uint32_t)P->fixBranchCount;
    k = 0;
    for (i = 0; i < P->size; ++i) {
        BBNode *bn = &P->graph[i];
        int fan = bn->hasComp ? bn->compFanIn : bn->isBranchFanIn;
        out->sig[i] = bn->sig;
        out->sig_diff[i] = bn->hasComp ? bn->compDiff : bn->sigDiff;
        out->sig_adj[i] = bn->storeAdj;
        out->flags[i] = (uint8_t)((fan ? ORC_CFC_FAN_IN : 0) | (bn->hasComp ? ORC_CFC_CHECKED : 0) | (bn->isBuffer ? ORC_CFC_BUFFER : 0) |
                                  (bn->skip ? ORC_CFC_SKIP : 0) | (bn->ret ? ORC_CFC_RET : 0));
        out->succ_begin[i] = (uint32_t)k;
        for (c = 0; c < bn->nTerm; ++c)
            out->succ[k++] = (uint16_t)bn->termSucc[c];
    }
    out->succ_begin[P->size] = (uint32_t)k;
    for (c = 0; c < (int)in->n_calls; ++c) {
        out->call_pre_adj[c] = preAdj[c];
        out->call_post_adj[c] = postAdj[c];
    }
    return 0;
}

/* ---------------------------------------------------------------- crazyCF.c: its -O0 control-flow graph */
/* Blocks by name, in module order after the pass has added its error blocks (createErrorBlocks, :107-126, appends one
 * "CFerrorHandler.<fn>" block to every function; insertErrorFunction, :88-105, appends the function FAULT_DETECTED_CFC).
 * Block names are clang's for this source; tools/cfg_from_ir.py re-derives the same graph from `clang -O0 -emit-llvm`. */
enum {
    GG_ENTRY, GG_ERR,
    FA_ENTRY, FA_COND, FA_BODY, FA_INC, FA_END, FA_ERR,
    M_ENTRY, M_LOOP, M_FOR_COND, M_FOR_BODY, M_SW0, M_SW5, M_SW17, M_SW25, M_SW37, M_DEFAULT, M_EPILOG, M_WHILE, M_WHILE_COND,
    M_WHILE_BODY, M_WHILE_END, M_FOR_INC, M_FOR_END, M_ERR,
    FD_BODY, FD_ERR,
    CCF_NBLOCKS
};
static const struct {
    int func, flags, nsucc, succ[6];
} ccf_blocks[CCF_NBLOCKS] = {
    [GG_ENTRY] = {0, ORC_CFC_RET, 0, {0}},
    [GG_ERR] = {0, ORC_CFC_SKIP, 0, {0}},
    [FA_ENTRY] = {1, 0, 1, {FA_COND}},
    [FA_COND] = {1, 0, 2, {FA_BODY, FA_END}},
    [FA_BODY] = {1, 0, 1, {FA_INC}},
    [FA_INC] = {1, 0, 1, {FA_COND}},
    [FA_END] = {1, ORC_CFC_RET, 0, {0}},
    [FA_ERR] = {1, ORC_CFC_SKIP, 0, {0}},
    [M_ENTRY] = {2, 0, 1, {M_LOOP}},
    [M_LOOP] = {2, 0, 1, {M_FOR_COND}},
    [M_FOR_COND] = {2, 0, 2, {M_FOR_BODY, M_FOR_END}},
    [M_FOR_BODY] = {2, 0, 6, {M_DEFAULT, M_SW0, M_SW5, M_SW17, M_SW25, M_SW37}},
    [M_SW0] = {2, 0, 1, {M_EPILOG}},
    [M_SW5] = {2, 0, 1, {M_EPILOG}},
    [M_SW17] = {2, 0, 1, {M_EPILOG}},
    [M_SW25] = {2, 0, 1, {M_SW37}},
    [M_SW37] = {2, 0, 1, {M_WHILE}},
    [M_DEFAULT] = {2, 0, 1, {M_EPILOG}},
    [M_EPILOG] = {2, 0, 1, {M_WHILE}},
    [M_WHILE] = {2, 0, 1, {M_WHILE_COND}},
    [M_WHILE_COND] = {2, 0, 2, {M_WHILE_BODY, M_WHILE_END}},
    [M_WHILE_BODY] = {2, 0, 1, {M_LOOP}},
    [M_WHILE_END] = {2, 0, 1, {M_FOR_INC}},
    [M_FOR_INC] = {2, 0, 1, {M_FOR_COND}},
    [M_FOR_END] = {2, ORC_CFC_RET, 0, {0}},
    [M_ERR] = {2, ORC_CFC_SKIP, 0, {0}},
    [FD_BODY] = {3, 0, 0, {0}},
    [FD_ERR] = {3, ORC_CFC_SKIP, 0, {0}},
};

static uint8_t ccf_flags[CCF_NBLOCKS];
static uint16_t ccf_func[CCF_NBLOCKS], ccf_succ[64];
static uint32_t ccf_succ_begin[CCF_NBLOCKS + 1];
static const uint16_t ccf_call_node[2] = {M_ENTRY, M_ENTRY}, ccf_call_entry[2] = {GG_ENTRY, FA_ENTRY};

void orc_crazycf_graph(orc_cfc_graph *g)
{
    int i, k, e = 0;
    for (i = 0; i < CCF_NBLOCKS; ++i) {
        ccf_flags[i] = (uint8_t)ccf_blocks[i].flags;
        ccf_func[i] = (uint16_t)ccf_blocks[i].func;
        ccf_succ_begin[i] = (uint32_t)e;
        for (k = 0; k < ccf_blocks[i].nsucc; ++k)
            ccf_succ[e++] = (uint16_t)ccf_blocks[i].succ[k];
    }
    ccf_succ_begin[CCF_NBLOCKS] = (uint32_t)e;
    g->n_nodes = CCF_NBLOCKS;
    g->flags = ccf_flags;
    g->func = ccf_func;
    g->succ_begin = ccf_succ_begin;
    g->succ = ccf_succ;
    g->n_calls = 2;
    g->call_node = ccf_call_node;
    g->call_entry = ccf_call_entry;
    g->main_func = 2;
}

/* ---------------------------------------------------------------- crazyCF.c under the signatures: one run */
/* The machine: a program counter that names a block, a return address (main's entry block is the only caller, `phase` says how
 * far it got), the program's variables, the two signature globals.  An upset on the branch target lands execution at the START
 * of another block (the corrupted-branch model the checks are built for, docs/source/cfcss.rst fig. 1); upsets on RTS / RTSA
 * hit the globals between the store and the next check. */
typedef struct {
    int total, times, i, fi, printed;
    uint32_t nprints;
    orc_glibc_rand rng;
} ccf_vars;

static int fault_mask(const orc_fault *fl, size_t nf, uint64_t item, uint32_t tick, int site, uint32_t *mask)
{
    int hit = 0;
    *mask = 0;
    for (size_t q = 0; q < nf; ++q)
        if (fl[q].item == item && fl[q].step == tick && fl[q].site == site) {
            *mask ^= 1u << (fl[q].bit & 31);
            hit = 1;
        }
    return hit;
}

void orc_crazycf_run(const orc_cfc_tables *T, int cfcss, int32_t seed, int32_t size, int32_t times, uint64_t item,
                     const orc_fault *fl, size_t nf, orc_crazycf_result *res, uint8_t *status)
{
    ccf_vars v;
    unsigned short RTS = 0, RTSA = 0; /* BasicBlockSignatureTracker, RunTimeSignatureAdjuster */
    int pc = M_ENTRY, phase = 0, inCall = 0, st = -1;
    uint32_t tick = 0;
    int64_t span = (int64_t)(size > 0 ? size : 0) + (int64_t)(times > 0 ? times : 0);
    int64_t cap64 = 16 * span + 256;
    uint32_t cap = (uint32_t)(cap64 < (1ll << 28) ? cap64 : (1ll << 28));
    memset(&v, 0, sizeof v);
    orc_glibc_srand(&v.rng, 1);

    while (st < 0) {
        int next = -1, isCall = 0, isRet = 0, callIx = 0, jumped;
        uint32_t m;
        /* the block's instructions, and where its terminator goes */
        switch (pc) {
        case GG_ENTRY: isRet = 1; break;
        case FA_ENTRY: v.fi = 0; next = 0; break;
        case FA_COND: next = (v.fi < size) ? 0 : 1; break;
        case FA_BODY: (void)(orc_glibc_rand_next(&v.rng) % 100); next = 0; break; /* array[i]: written, never read */
        case FA_INC: v.fi++; next = 0; break;
        case FA_END: isRet = 1; break;
        case M_ENTRY:
            if (phase == 0) {
                isCall = 1, callIx = 0;
            } else if (phase == 1) {
                v.total = 0;
                orc_glibc_srand(&v.rng, (uint32_t)seed);
                isCall = 1, callIx = 1;
            } else {
                v.times = times;
                v.i = 0;
                next = 0;
            }
            break;
        case M_LOOP: next = 0; break;
        case M_FOR_COND: next = (v.i < size) ? 0 : 1; break;
        case M_FOR_BODY:
            switch (v.i) {
            case 0: next = 1; break;
            case 5: next = 2; break;
            case 17: next = 3; break;
            case 25: next = 4; break;
            case 37: next = 5; break;
            default: next = 0;
            }
            break;
        case M_SW0: v.total += (int)(orc_glibc_rand_next(&v.rng) % 10); next = 0; break;
        case M_SW5: v.total += 127; next = 0; break;
        case M_SW17: v.printed = v.total; v.nprints++; next = 0; break;
        case M_SW25: v.total += 25; next = 0; break;
        case M_SW37: next = 0; break;
        case M_DEFAULT: v.total -= 10; next = 0; break;
        case M_EPILOG: next = 0; break;
        case M_WHILE: next = 0; break;
        case M_WHILE_COND: next = (v.times > 0) ? 0 : 1; break;
        case M_WHILE_BODY: v.total -= 1; v.times--; next = 0; break;
        case M_WHILE_END: next = 0; break;
        case M_FOR_INC: v.i++; next = 0; break;
        case M_FOR_END: st = ORC_CFC_OK; break;
        default: next = 0; break; /* buffer block */
        }
        if (st >= 0)
            break;
        /* stores at the end of the block / before the call */
        uint32_t target;
        if (isCall) {
            if (cfcss) {
                RTS = T->sig[pc];
                RTSA = T->call_pre_adj[callIx];
            }
            target = ccf_call_entry[callIx];
        } else {
            if (cfcss) {
                RTS = T->sig[pc];
                RTSA = T->sig_adj[pc];
            }
            target = isRet ? (uint32_t)M_ENTRY : T->succ[T->succ_begin[pc] + (uint32_t)next];
        }
        jumped = fault_mask(fl, nf, item, tick, ORC_SITE_CFC_PC, &m);
        target ^= m;
        if (fault_mask(fl, nf, item, tick, ORC_SITE_CFC_RTS, &m))
            RTS ^= (unsigned short)m;
        if (fault_mask(fl, nf, item, tick, ORC_SITE_CFC_RTSA, &m))
            RTSA ^= (unsigned short)m;
        tick += 1;
        if (isRet && !jumped) {
            if (!inCall) {
                st = ORC_CFC_WILD;
                break;
            }
            inCall = 0;
            if (cfcss) {
                RTS = T->sig[M_ENTRY];
                RTSA = T->call_post_adj[phase - 1];
            }
            pc = M_ENTRY;
        } else {
            if (target >= T->n_nodes) {
                st = ORC_CFC_WILD;
                break;
            }
            if ((T->flags[target] & ORC_CFC_SKIP) || target == FD_BODY) { /* error handler -> FAULT_DETECTED_CFC -> abort */
                st = ORC_CFC_DETECTED;
                break;
            }
            if (cfcss && (T->flags[target] & ORC_CFC_CHECKED)) {
                unsigned short x = RTS ^ T->sig_diff[target];
                if (T->flags[target] & ORC_CFC_FAN_IN)
                    x ^= RTSA;
                if (x != T->sig[target]) {
                    st = ORC_CFC_DETECTED;
                    break;
                }
            }
            if (isCall) {
                inCall = 1;
                phase += 1;
            } else if (isRet)
                inCall = 0;
            if (jumped && target == M_ENTRY)
                phase = 0;
            pc = (int)target;
        }
        if (tick >= cap)
            st = ORC_CFC_WATCHDOG;
    }
    res->total = v.total;
    res->printed = v.printed;
    res->n_prints = v.nprints;
    res->blocks = tick;
    *status = (uint8_t)st;
}

void orc_crazycf_batch(const orc_cfc_tables *T, int cfcss, const int32_t *params, size_t n, const orc_fault *fl, size_t nf,
                       orc_crazycf_result *res, uint8_t *status)
{
    for (size_t q = 0; q < n; ++q)
        orc_crazycf_run(T, cfcss, params[3 * q], params[3 * q + 1], params[3 * q + 2], (uint64_t)q, fl, nf, &res[q], &status[q]);
}

/* the program as the C source states it, no blocks, no signatures: what pins the block walk above (and is itself pinned by
 * oracle/_ref's compile of crazyCF.c) */
void orc_crazycf_plain(int32_t seed, int32_t size, int32_t timesThroughWhile, orc_crazycf_result *res)
{
    orc_glibc_rand g;
    int total = 0, i, printed = 0;
    uint32_t nprints = 0;
    orc_glibc_srand(&g, (uint32_t)seed);
    for (i = 0; i < size; i++)
        (void)orc_glibc_rand_next(&g);
    i = 0;
LOOP:
    for (; i < size; i++) {
        switch (i) {
        case 0: total += (int)(orc_glibc_rand_next(&g) % 10); break;
        case 5: total += 127; break;
        case 17: printed = total; nprints++; break;
        case 25: total += 25; /* falls through */
        case 37: goto WHILE;
        default: total -= 10;
        }
    WHILE:
        while (timesThroughWhile > 0) {
            total -= 1;
            timesThroughWhile--;
            goto LOOP;
        }
    }
    res->total = total;
    res->printed = printed;
    res->n_prints = nprints;
    res->blocks = 0;
}
