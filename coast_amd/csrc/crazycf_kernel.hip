// crazycf_kernel.hip -- tests/crazyCF/crazyCF.c, the reference's CFCSS test program, as a batch kernel under control-flow
// signatures.  The program is "a nightmare of a control flow graph" (crazyCF.c:1-6): a for loop whose body is a switch with
// fall-through and gotos into a while loop that jumps back to the for statement, fed by rand().
//
// Work item = one run of main() (crazyCF.c:29-72) with its three constants -- srand(42), size = 20, timesThroughWhile = 10
// (:36, :11, :41) -- replaced by the item's parameters.  One lane per item.  The wave walks the program's -O0 basic blocks
// (the units the pass instruments; Makefile.common compiles at -O0 before opt): each turn of the outer loop every lane executes
// the body of the block it is in, stores the block's signature / adjuster, picks its branch target, and the wave checks the 64
// targets' signatures with one compare.  Lanes in different blocks are different EXEC masks of the same switch.
//
// Block numbering = module order of the instrumented module, error-handler blocks included (they take a signature each,
// CFCSS.cpp:154-183):  generateGolden {0 entry, 1 CFerrorHandler}  fillArray {2 entry, 3 for.cond, 4 for.body, 5 for.inc,
// 6 for.end, 7 CFerrorHandler}  main {8 entry, 9 LOOP, 10 for.cond, 11 for.body, 12 case 0, 13 case 5, 14 case 17, 15 case 25,
// 16 case 37, 17 default, 18 sw.epilog, 19 WHILE, 20 while.cond, 21 while.body, 22 while.end, 23 for.inc, 24 for.end,
// 25 CFerrorHandler}  FAULT_DETECTED_CFC {26 body, 27 CFerrorHandler} (insertErrorFunction, :88-105);  28.. buffer blocks.
#include "cfcss.hpp"
#include "xmr.hpp"

namespace coast {

enum { SITE_CFC_PC = 56, SITE_CFC_RTS = 57, SITE_CFC_RTSA = 58 };
enum { kCfcOk = 0, kCfcDetected = 1, kCfcWatchdog = 2, kCfcWild = 3, kCfcRunning = 255 };

// ---- the program's control-flow graph (host side: input of coast_cfcss_assign) ----
constexpr uint32_t kCcfNodes = 28;
constexpr uint16_t kCcfMainFunc = 2;
static const uint8_t kCcfFlags[kCcfNodes] = {
    COAST_CFC_RET, COAST_CFC_SKIP,                                     // generateGolden
    0, 0, 0, 0, COAST_CFC_RET, COAST_CFC_SKIP,                         // fillArray
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, COAST_CFC_RET, COAST_CFC_SKIP, // main
    0, COAST_CFC_SKIP};                                                // FAULT_DETECTED_CFC: abort(); unreachable
static const uint16_t kCcfFunc[kCcfNodes] = {0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3};
static const uint32_t kCcfSuccBegin[kCcfNodes + 1] = {0, 0, 0, 1, 3, 4, 5, 5, 5, 6, 7, 9, 15, 16, 17, 18, 19, 20, 21, 22, 23, 25, 26, 27, 28, 28, 28, 28, 28};
static const uint16_t kCcfSucc[28] = {
    3,                      // 2  fillArray entry -> for.cond                                   crazyCF.c:22
    4, 6,                   // 3  i < size ? for.body : for.end
    5,                      // 4  array[i] = rand() % 100                                       :23
    3,                      // 5  i++
    9,                      // 8  main entry -> LOOP                                            :30-42
    10,                     // 9  LOOP: -> for.cond                                             :43
    11, 24,                 // 10 i < size ? for.body : for.end
    17, 12, 13, 14, 15, 16, // 11 switch (i): default, 0, 5, 17, 25, 37                         :44-60
    18,                     // 12 total += rand() % 10; break
    18,                     // 13 total += 127; break
    18,                     // 14 printf("total so far"); break
    16,                     // 15 total += 25; falls through
    19,                     // 16 goto WHILE
    18,                     // 17 total -= 10
    19,                     // 18 sw.epilog -> WHILE
    20,                     // 19 WHILE: -> while.cond                                          :61
    21, 22,                 // 20 timesThroughWhile > 0 ? while.body : while.end
    9,                      // 21 total -= 1; timesThroughWhile--; goto LOOP                    :62-64
    23,                     // 22 while.end -> for.inc
    10};                    // 23 i++ -> for.cond
static const uint16_t kCcfCallNode[2] = {8, 8};  // generateGolden() (:31), fillArray(array) (:39); srand / rand / printf are declarations
static const uint16_t kCcfCallEntry[2] = {0, 2};

// glibc TYPE_3 random state of one lane: 31 words, lane-interleaved in LDS (conflict-free for any per-lane position)
struct LaneRand {
    int32_t *st; // &ring[0][lane], stride 64
    uint32_t f, r;
    __device__ __forceinline__ uint32_t next()
    {
        const uint32_t v = (uint32_t)st[f * 64u] + (uint32_t)st[r * 64u];
        st[f * 64u] = (int32_t)v;
        f = f == 30u ? 0u : f + 1u;
        r = r == 30u ? 0u : r + 1u;
        return v >> 1;
    }
    __device__ __forceinline__ void seed(uint32_t s) // __srandom_r
    {
        if (s == 0u)
            s = 1u;
        st[0] = (int32_t)s;
        long long word = (long long)(int32_t)s; // glibc's `word` is an int32_t
        for (uint32_t i = 1; i < 31u; ++i) {
            const long long hi = word / 127773, lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0)
                word += 2147483647;
            st[i * 64u] = (int32_t)word;
        }
        f = 3u;
        r = 0u;
        for (int i = 0; i < 310; ++i)
            (void)next();
    }
};

template <bool CFCSS>
__global__ __launch_bounds__(64) void crazycf_kernel(const coast_crazycf_params *__restrict__ params, uint64_t nitems,
                                                     coast_crazycf_result *__restrict__ results, uint8_t *__restrict__ status,
                                                     const CfcDevTables *__restrict__ dTab, FaultTab ft)
{
    __shared__ CfcDevTables tab;
    __shared__ int32_t ring[31][64];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dTab);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&tab);
        for (uint32_t w = threadIdx.x; w < sizeof(CfcDevTables) / 4u; w += 64u)
            dst[w] = src[w];
    }
    wave_lds_sync();
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const uint64_t item = (uint64_t)tile * 64u + lane;
    const bool live = item < nitems;
    coast_crazycf_params pr = {1, 0, 0};
    if (live)
        pr = params[item];
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];

    LaneRand rnd{&ring[0][lane], 3u, 0u};
    rnd.seed(1u); // a program that has not called srand() yet draws from the srand(1) state (a jump can reach rand() first)
    CfcTracker trk;
    // the program's variables (all of them live in this lane's registers)
    int32_t total = 0, times = 0, i = 0, fi = 0, printed = 0;
    uint32_t nprints = 0u;
    uint32_t pc = 8u, phase = 0u, tick = 0u, st = live ? (uint32_t)kCfcRunning : (uint32_t)kCfcOk;
    bool inCall = false;
    const long long span = (long long)(pr.size > 0 ? pr.size : 0) + (long long)(pr.times > 0 ? pr.times : 0);
    const uint32_t cap = (uint32_t)(16ll * span + 256ll < (1ll << 28) ? 16ll * span + 256ll : (1ll << 28));
    const uint32_t nNodes = tab.nNodes;

    while (__ballot(st == kCfcRunning) != 0ull) {
        if (st != kCfcRunning)
            continue;
        // ---- the block's own instructions ----
        enum { BR, CALL, RET, EXIT };
        uint32_t kind = BR, choice = 0u, callee = 0u, callIx = 0u;
        switch (pc) {
        case 0: kind = RET; break;                                   // generateGolden: ;                             :16-18
        case 2: fi = 0; break;                                       // for (int i = 0;                               :22
        case 3: choice = fi < pr.size ? 0u : 1u; break;              //      i < size;
        case 4: (void)rnd.next(); break;                             // array[i] = rand() % 100  (array is never read) :23
        case 5: ++fi; break;                                         //      i++)
        case 6: kind = RET; break;                                   // return                                        :26
        case 8:                                                      // main's entry block                            :30-42
            if (phase == 0u) {
                kind = CALL, callee = 0u, callIx = 0u;               // generateGolden();
            } else if (phase == 1u) {
                total = 0;
                rnd.seed((uint32_t)pr.seed);                         // srand(42)
                kind = CALL, callee = 2u, callIx = 1u;               // fillArray(array);
            } else {
                times = pr.times;                                    // int timesThroughWhile = 10; int i = 0;
                i = 0;
            }
            break;
        case 9: break;                                               // LOOP:
        case 10: choice = i < pr.size ? 0u : 1u; break;              // for (; i < size;
        case 11:                                                     // switch (i)                                    :44
            choice = i == 0 ? 1u : i == 5 ? 2u : i == 17 ? 3u : i == 25 ? 4u : i == 37 ? 5u : 0u;
            break;
        case 12: total += (int32_t)(rnd.next() % 10u); break;        // total += rand() % 10                          :46
        case 13: total += 127; break;                                //                                               :49
        case 14: printed = total, ++nprints; break;                  // printf("total so far: %d\n", total)           :52
        case 15: total += 25; break;                                 //                                               :55
        case 16: break;                                              // goto WHILE                                    :57
        case 17: total -= 10; break;                                 // default                                       :59
        case 18: break;
        case 19: break;                                              // WHILE:
        case 20: choice = times > 0 ? 0u : 1u; break;                // while (timesThroughWhile > 0)                 :61
        case 21: total -= 1, --times; break;                         // total -= 1; timesThroughWhile--; goto LOOP    :62-64
        case 22: break;
        case 23: ++i; break;                                         // i++)
        case 24: kind = EXIT; break;                                 // printf("Total = %d\n", total); return 0       :69-71
        default: break;                                              // a buffer block: nothing but the signature code
        }
        if (kind == EXIT) {
            st = kCfcOk;
            continue;
        }
        // ---- leave: stores before the terminator / before the call (CFCSS.cpp:494-506, 617) ----
        uint32_t target;
        if (kind == CALL) {
            if (CFCSS) {
                trk.rts = tab.sig[pc];
                trk.rtsa = tab.callPreAdj[callIx];
            }
            target = callee;
        } else {
            if (CFCSS)
                trk.leave(tab, pc);
            target = kind == RET ? 8u : (uint32_t)tab.succ[tab.succBegin[pc] + choice];
        }
        // ---- the upset: a corrupted branch target lands at the start of another block; or the tracker pair is hit ----
        bool jumped = false;
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step != tick || df.local != lane)
                continue;
            const uint32_t m = 1u << (df.bit & 31u);
            if (df.site == SITE_CFC_PC)
                target ^= m, jumped = true;
            else if (df.site == SITE_CFC_RTS)
                trk.rts ^= m;
            else if (df.site == SITE_CFC_RTSA)
                trk.rtsa ^= m;
        }
        ++tick;
        if (kind == RET && !inCall && !jumped) { // a return nobody called for: the address on the stack is not ours
            st = kCfcWild;
            continue;
        }
        if (kind == RET && !jumped) { // back in main's entry block: signature and adjuster are stored again (:620-626)
            inCall = false;
            if (CFCSS) {
                trk.rts = tab.sig[8];
                trk.rtsa = tab.callPostAdj[phase - 1u];
            }
            pc = 8u;
        } else {
            if (target >= nNodes) {
                st = kCfcWild;
                continue;
            }
            const uint32_t fl = tab.flags[target];
            if ((fl & kCfcSkip) || target == 26u) { // an error-handler block or FAULT_DETECTED_CFC itself: abort()
                st = kCfcDetected;
                continue;
            }
            if (CFCSS && trk.enter_bad(tab, target)) {
                st = kCfcDetected;
                continue;
            }
            if (kind == CALL) // the call pushed its return address, wherever execution landed
                inCall = true, ++phase;
            else if (kind == RET)
                inCall = false;
            if (jumped && target == 8u)
                phase = 0u; // main starts over
            pc = target;
        }
        if (tick >= cap)
            st = kCfcWatchdog;
    }
    if (live) {
        coast_crazycf_result rs;
        rs.total = total;
        rs.printed = printed;
        rs.n_prints = nprints;
        rs.blocks = tick;
        results[item] = rs;
        status[item] = (uint8_t)st;
    }
}

// crazyCF under -TMR / -DWC (unittest/cfg/full_tmr.yml:8 runs the program with `-TMR`): main() (crazyCF.c:29-69) and fillArray() (:20-27)
// statement by statement, one lane group per run (NREP adjacent lanes, one per replica).  Replica-private: main's i, total,
// timesThroughWhile and fillArray's i.  `size` is a global (one copy); srand / rand / printf are library calls -- rand's value fans out to
// the copies (every replica lane keeps the same generator state and steps it on the same, group-uniform path).  array[] is filled and
// never read: its stores have an address and a datum to vote, nothing else.  Sync points:
//   always                   the arguments of the two printf calls (`total`): values handed to an unprotected call
//                            (processCallSync, synchronization.cpp:951-1100) -- the frozen schedule
//   COAST_F_BRANCH_SYNC      the three loop conditions, the operand of `switch (i)`, main's return value (through %retval)  :741-949
//   COAST_F_ADDR_SYNC        the offset of `array[i] = ..` (a store address: off with -noStoreAddrSync)                     :413-474
//   COAST_F_LOCAL_STORE_SYNC the data of every store of a computed value: the VLA's element count, array[i], fillArray's i++, every
//                            update of total, timesThroughWhile--, main's i++                                    :197-224, 476-561
// All of them on the program's own constants: 82 + 30 + 1 + 20 + 20 + 90 + the 2 printf arguments = 245, the counts of the reference's
// -O0 IR (tools/ir_sync_counts.py crazycf).  Fault sites SITE_CCF_I / _TOTAL / _TIMES / _FI: a bit of that register of one replica right
// before branch condition `step` of the run.  Watchdog as crazycf_kernel's.  Oracle: oracle/crazycf_xmr.inc.
enum { SITE_CCF_I = 72, SITE_CCF_TOTAL = 73, SITE_CCF_TIMES = 74, SITE_CCF_FI = 75 };
template <int NREP>
__global__ __launch_bounds__(64) void crazycf_xmr_kernel(const coast_crazycf_params *__restrict__ params, uint64_t nitems,
                                                         coast_crazycf_result *__restrict__ results, uint8_t *__restrict__ status,
                                                         Counters ctr, FaultTab ft, uint8_t *__restrict__ detected)
{
    __shared__ int32_t ring[31][64];
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u;
    const bool ss = (ctr.flags & kFlagAddrSync) != 0u && !(ctr.flags & kFlagNoStoreAddrSync);
    const bool lss = xmr_local_sync_on(ctr.flags);
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < nitems;
    const bool cnt = live && lm.r == 0;
    if (lane < 4)
        sCnt[lane] = 0;
    coast_crazycf_params pr = params[live ? item : 0];
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    Tally tl;
    LaneRand rnd{&ring[0][lane], 3u, 0u};
    rnd.seed((uint32_t)pr.seed);                                                     // srand(seed)                            :35
    int32_t i = 0, total = 0, times = pr.times, fi = 0, printed = 0, final = 0;
    uint32_t nprints = 0u, tick = 0u;
    bool wd = false;
    const long long span = (long long)(pr.size > 0 ? pr.size : 0) + (long long)(pr.times > 0 ? pr.times : 0);
    const uint32_t cap = (uint32_t)(16ll * span + 256ll < (1ll << 28) ? 16ll * span + 256ll : (1ll << 28));
    if (lm.live) { // (the idle lane of a TMR wave has no run of its own)
        auto cond = [&](const int32_t &reg, int32_t lim, bool gt) { // one evaluated loop condition
            for (uint32_t q = 0; q < fr.y; ++q) { // the registers' upsets land right before the condition reads them
                const DevFault df = ft.list[fr.x + q];
                if (df.step != tick || (int)df.local != slot || (int)df.replica != lm.r)
                    continue;
                const int32_t m = (int32_t)(1u << (df.bit & 31u));
                if (df.site == SITE_CCF_I)
                    i ^= m;
                else if (df.site == SITE_CCF_TOTAL)
                    total ^= m;
                else if (df.site == SITE_CCF_TIMES)
                    times ^= m;
                else if (df.site == SITE_CCF_FI)
                    fi ^= m;
            }
            if (tick >= cap) {
                wd = true;
                return false;
            }
            ++tick;
            return xmr_steer<NREP>((gt ? reg > lim : reg < lim) ? 1u : 0u, lm, bs, cnt, tl) != 0u;
        };
        auto lsy = [&](int32_t v) { return (int32_t)xmr_local_sync<NREP>((uint32_t)v, lm, lss, cnt, tl); }; // the data of a store (L)
        (void)lsy(pr.size);                                                          // int array[size]: the VLA's element count :32
        for (; cond(fi, pr.size, false); fi = lsy((int32_t)((uint32_t)fi + 1u))) {   // fillArray: for (i = 0; i < size; i++)  :22
            const uint32_t rv = rnd.next() % 100u;                                   //   array[i] = rand() % 100             :23
            (void)xmr_steer<NREP>((uint32_t)fi, lm, ss, cnt, tl);
            (void)lsy((int32_t)rv);
        }
        times = pr.times, i = 0;                                                     // int timesThroughWhile = 10; int i = 0; :40-41
        for (;;) {                                                                   // LOOP: for (; i < size; i++)            :43
            if (!cond(i, pr.size, false))
                break;
            const int32_t sw = (int32_t)xmr_steer<NREP>((uint32_t)i, lm, bs, cnt, tl); // switch (i): the voted operand steers  :44
            if (sw == 0)
                total = lsy((int32_t)((uint32_t)total + rnd.next() % 10u));          //   case 0: total += rand() % 10         :46
            else if (sw == 5)
                total = lsy((int32_t)((uint32_t)total + 127u));                      //   case 5                               :49
            else if (sw == 17) {                                                     //   case 17: printf("total so far: %d")  :52
                const uint32_t v = xmr_sync<NREP>((uint32_t)total, lm, cnt, tl);     //   (the argument of an unprotected call is voted)
                printed = (int32_t)(NREP == 3 ? v : xmr_rep0<NREP>((uint32_t)total, lm));
                ++nprints;
            } else if (sw == 25)
                total = lsy((int32_t)((uint32_t)total + 25u));                       //   case 25: falls into case 37: goto WHILE :55-58
            else if (sw != 37)
                total = lsy((int32_t)((uint32_t)total - 10u));                       //   default                              :60
            if (cond(times, 0, true)) {                                              // WHILE: while (timesThroughWhile > 0)   :62
                total = lsy((int32_t)((uint32_t)total - 1u));
                times = lsy((int32_t)((uint32_t)times - 1u));
                continue;                                                            //   goto LOOP: back to the condition, no i++ :65
            }
            i = lsy((int32_t)((uint32_t)i + 1u));
        }
        {                                                                            // printf("Total = %d\n", total)          :67
            const uint32_t v = xmr_sync<NREP>((uint32_t)total, lm, cnt, tl);
            final = (int32_t)(NREP == 3 ? v : xmr_rep0<NREP>((uint32_t)total, lm));
        }
        if (bs)
            (void)xmr_sync<NREP>(0u, lm, cnt, tl);                                   // return 0, through %retval              :69
    }
    uint32_t detItems = 0;
    if (cnt) {
        coast_crazycf_result rs;
        rs.total = final;
        rs.printed = printed;
        rs.n_prints = nprints;
        rs.blocks = tick;
        results[item] = rs;
        status[item] = (uint8_t)(wd ? kCfcWatchdog : kCfcOk);
        if (tl.det) {
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
