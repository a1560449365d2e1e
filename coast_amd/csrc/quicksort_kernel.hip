// quicksort_kernel.hip -- protected quick_sort of tests/quicksort/quicksort.c:109-129 (the LANL quicksort benchmark; Hoare
// partition from Rosetta code) over a batch of int arrays: the first workload here whose LOOP TRIP COUNTS depend on the data.
//
// Work item = one array, sorted by a lane group (NREP adjacent lanes, one per replica).  Everything the function computes with is
// replica-private and lives in the lane's registers: i, j, the pivot, the two values the scans loaded last, base and length of
// the part being sorted; the pending right parts of the recursion are a per-lane stack in LDS, walked in the reference's order
// (left part first).  The array is memory: ONE copy (-noMemReplication), staged into LDS for the duration of the region when the
// tile's arrays fit (one coalesced copy in, one out), otherwise left in HBM.
//
// Sync points = the reference's -noMemReplication rule set applied to the source as written (frozen in oracle/coast_oracle.c:
// qs_item): every evaluated branch condition -- a terminator sync on an i1 (synchronization.cpp:146-155, 741-949): all replicas
// of an array continue in the voted direction, which is what keeps three lanes with data-dependent trip counts convergent; every
// GEP offset (A[len/2], A[i], A[j]; :333-372, 413-474 -- loads off with -noLoadSync, stores off with -noStoreAddrSync); the data
// of both stores of a swap (:197-224, off with -noStoreDataSync).  Different arrays diverge freely (the wave executes the union
// of their paths); cross-lane traffic only ever happens between the lanes of one array, which are always at the same point.
//
// A corrupted index can leave the array: such loads return the replica's pivot (both scans stop), such stores are dropped; a
// sort that does not end within 64 n + 1024 conditions, or nests deeper than kQsMaxDepth pending parts, is cut and reported in
// the status array (the reference's supervisor files those runs under timeout / stack overflow, jsonParser.py:162-186).
#include "xmr.hpp"

namespace coast {

enum { SITE_QS_I = 48, SITE_QS_J = 49, SITE_QS_PIVOT = 50, SITE_QS_VI = 51, SITE_QS_VJ = 52 };
enum { kQsOk = 0, kQsWatchdog = 1, kQsStack = 2 };
constexpr int kQsMaxDepth = 48;

template <int NREP> struct QsGeom {
    static constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    static constexpr int STACK_BYTES = 64 * kQsMaxDepth * 8;                       // per-lane stack of (base, len)
    static constexpr int SLAB_BUDGET = 64 * 1024 - STACK_BYTES - 64;               // what the arrays may take in LDS
    static size_t lds_bytes(uint32_t n) { return (size_t)STACK_BYTES + 64 + ((size_t)n * 4 * IPW <= (size_t)SLAB_BUDGET ? (size_t)n * 4 * IPW : 0); }
    static bool staged(uint32_t n) { return (size_t)n * 4 * IPW <= (size_t)SLAB_BUDGET; }
};

// one wave (64-thread workgroup) per tile of IPW arrays
template <int NREP>
__global__ __launch_bounds__(64) void quicksort_kernel(int32_t *__restrict__ arrays, uint32_t n, uint64_t narrays, Counters ctr,
                                                       FaultTab ft, uint8_t *__restrict__ detected, uint8_t *__restrict__ status,
                                                       int staged)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smemQ[];
    using G = QsGeom<NREP>;
    uint2 *stk = reinterpret_cast<uint2 *>(smemQ) + threadIdx.x * kQsMaxDepth; // this lane's pending right parts
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(smemQ + G::STACK_BYTES);
    int32_t *slab = reinterpret_cast<int32_t *>(smemQ + G::STACK_BYTES + 64);
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool ls = !(ctr.flags & kFlagNoLoadSync), ss = !(ctr.flags & kFlagNoStoreAddrSync);
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * G::IPW + (uint64_t)slot;
    const bool live = lm.live && item < narrays;
    const bool cnt = live && lm.r == 0;
    const bool writer = cnt; // the single memory copy is written by the original store (replica 0's lane)
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;

    // the tile's arrays are contiguous in HBM: stage them with coalesced 4-byte-per-lane strips (any n)
    const uint64_t tileItems = (uint64_t)tile * G::IPW + G::IPW <= narrays ? (uint64_t)G::IPW : narrays - (uint64_t)tile * G::IPW;
    int32_t *tileBase = arrays + (uint64_t)tile * G::IPW * n;
    if (staged) {
        for (uint64_t e = threadIdx.x; e < tileItems * n; e += 64)
            slab[e] = tileBase[e];
    }
    wave_lds_sync();
    int32_t *A = staged ? slab + (size_t)slot * n : tileBase + (size_t)(live ? slot : 0) * n;

    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];

    Tally tl;
    uint32_t base = 0u, len = live ? n : 0u, sp = 0u, tick = 0u, st = kQsOk;
    const uint32_t cap = 64u * n + 1024u; // the watchdog (the supervisor's timeout): a sort is cut after this many branch conditions
    uint32_t i = 0u, j = 0u, pv = 0u, vi = 0u, vj = 0u;
    auto hook = [&]() __attribute__((always_inline)) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if (df.step != tick || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            const uint32_t m = 1u << (df.bit & 31u);
            if (df.site == SITE_QS_I)
                i ^= m;
            else if (df.site == SITE_QS_J)
                j ^= m;
            else if (df.site == SITE_QS_PIVOT)
                pv ^= m;
            else if (df.site == SITE_QS_VI)
                vi ^= m;
            else if (df.site == SITE_QS_VJ)
                vj ^= m;
        }
    };
    auto cond = [&](bool c) __attribute__((always_inline)) { // one evaluated branch condition: always a sync point
        ++tick;
        // The idle lane of a TMR wave (lane 63: 64 = 3 x 21 + 1) has no item: its "replica group" wraps around to lanes 0 and 1,
        // so a voted condition made it follow item 0's decisions on its own garbage state -- bounded only by the watchdog
        // (found in round 3: lifting the watchdog for clean arrays hung the kernel).  It follows its own condition instead:
        // len = 0 < 2, it leaves at once.  Measured effect with the watchdog in place: 54.7 -> 53.0 ms per 16 Ki x 580 (TMR).
        if (!lm.live)
            return c;
        return xmr_steer<NREP>(c ? 1u : 0u, lm, true, cnt, tl) != 0u;
    };
    auto load = [&](uint32_t off) __attribute__((always_inline)) { return off < n ? (uint32_t)A[off] : pv; };

    for (;;) {
        if (tick >= cap) {
            st = kQsWatchdog;
            break;
        }
        hook();
        if (cond(len < 2u)) {                                        // if (len < 2) return;                         :110
            if (sp == 0u)
                break;
            --sp;
            const uint2 f = stk[sp];
            base = f.x;
            len = f.y;
            continue;
        }
        {
            const uint32_t po = xmr_steer<NREP>(base + len / 2u, lm, ls, cnt, tl); // pivot = A[len / 2]             :112
            pv = po < n ? (uint32_t)A[po] : 0u;
        }
        i = base;
        j = base + len - 1u;
        for (;;) {                                                   // for (i = 0, j = len - 1; ; i++, j--)         :115
            for (;;) {                                               // while (A[i] < pivot) i++;                    :116
                vi = load(xmr_steer<NREP>(i, lm, ls, cnt, tl));
                hook();
                if (!cond((int32_t)vi < (int32_t)pv) || tick >= cap)
                    break;
                i += 1u;
            }
            for (;;) {                                               // while (A[j] > pivot) j--;                    :117
                vj = load(xmr_steer<NREP>(j, lm, ls, cnt, tl));
                hook();
                if (!cond((int32_t)vj > (int32_t)pv) || tick >= cap)
                    break;
                j -= 1u;
            }
            hook();
            if (cond((int32_t)i >= (int32_t)j) || tick >= cap)       // if (i >= j) break;                           :119
                break;
            {                                                        // temp = A[i]; A[i] = A[j]; A[j] = temp;  :121-123
                const uint32_t oi = xmr_steer<NREP>(i, lm, ss, cnt, tl);
                uint32_t d = xmr_store_sync<NREP>(vj, lm, cnt, tl);
                if (NREP != 3 || !lm.storeSync)
                    d = xmr_rep0<NREP>(d, lm);
                wave_lds_sync();
                if (writer && oi < n)
                    A[oi] = (int32_t)d;
                const uint32_t oj = xmr_steer<NREP>(j, lm, ss, cnt, tl);
                uint32_t e = xmr_store_sync<NREP>(vi, lm, cnt, tl);
                if (NREP != 3 || !lm.storeSync)
                    e = xmr_rep0<NREP>(e, lm);
                if (writer && oj < n)
                    A[oj] = (int32_t)e;
                wave_lds_sync(); // the next loads of the other replica lanes must see both stores
            }
            i += 1u;
            j -= 1u;
        }
        if (tick >= cap) {
            st = kQsWatchdog;
            break;
        }
        if (sp == (uint32_t)kQsMaxDepth) {
            st = kQsStack;
            break;
        }
        stk[sp] = make_uint2(i, base + len - i);                     // quick_sort(A, i); quick_sort(A + i, len - i); :126-127
        ++sp;
        len = i - base;
    }

    wave_lds_sync();
    if (staged) {
        __builtin_amdgcn_s_barrier(); // one wave: orders the divergent sorts before the cooperative copy-out
        for (uint64_t e = threadIdx.x; e < tileItems * n; e += 64)
            tileBase[e] = slab[e];
    }
    uint32_t detItems = 0;
    if (cnt) {
        if (status)
            status[item] = (uint8_t)st;
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
