// cache_test_kernel.hip -- protected calc_sum of tests/cache_test/cacheTest.c:101-177: a batch of int arrays is summed,
// every element that is not its own index is counted and rewritten (the benchmark's memory scrub).
//
// Work item = one array, walked by a lane group (NREP adjacent lanes, one per replica) exactly as the reference walks it;
// replicated registers: the running sum, the loaded element, numberOfErrors.  Sync points (frozen schedule, oracle
// ct_item): the data-dependent branch condition `array[i] != i` of every element -- a terminator sync on an i1
// (synchronization.cpp:146-155, 741-949): all copies continue on the voted outcome, so the replica lanes cannot diverge;
// the returned sum (ReturnInst sync); the error count where it is stored (store-data sync).  `array[i] = i` stores the loop
// index, a scalar outside the sphere of replication.
//
// The i1 votes are exact but not one exchange each: 32 conditions travel as one mask, the voter is the per-bit form of
// select(a == b, a, c) = (a & ~(a^b)) | (c & (a^b)), and TMR_ERROR_CNT grows by popcount((a^b) | (a^c)) -- one per
// condition whose copies were not all equal, as :1391-1443 counts them.
//
// Roofline: HBM.  Algorithmic bytes = 4 n per array read (+ 8 per array written; rewrites only where memory was corrupt).
// A lane group walking its own 2400-byte array is a 2400-byte stride across the wave: left alone, every 16-byte load touches
// its own cache line (first version: 2.7 TB/s in every mode).  So the wave fetches cooperatively -- eight lanes read the 128
// contiguous bytes (32 elements) of one array, eight arrays per load instruction -- into a wave-private LDS slab, and each
// replica lane then reads its array's 32 elements from there ("one load feeds all replicas").  The next group's loads are
// in flight while the current one is summed and compared; no workgroup barrier, the waves are independent.
#include "xmr.hpp"

namespace coast {

enum { SITE_CT_SUM = 32, SITE_CT_VAL = 33, SITE_CT_NERR = 34, SITE_CT_I = 35 };

// vote a mask of up to 32 branch conditions (bit k = condition k of the group, `gmask` = the bits in use)
template <int NREP>
__device__ __forceinline__ uint32_t xmr_vote_conditions(uint32_t m, uint32_t gmask, const LaneMap<NREP> &lm, bool count, Tally &t)
{
    if constexpr (NREP == 1) {
        return m;
    } else {
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4, (int)m);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4 + 4, (int)m);
        if constexpr (NREP == 2) {
            if (count) {
                t.syncs += (uint32_t)__builtin_popcount(gmask);
                t.det |= ((a ^ b) & gmask) ? 1u : 0u;
            }
            return a; // the region goes on with replica 0's outcome (the reference would be in the error handler)
        } else {
            const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4 + 8, (int)m);
            const uint32_t ab = a ^ b, bad = (ab | (a ^ c)) & gmask;
            if (count) {
                t.syncs += (uint32_t)__builtin_popcount(gmask);
                t.miss += (uint32_t)__builtin_popcount(bad);
                t.det |= bad ? 1u : 0u;
            }
            return (a & ~ab) | (c & ab);
        }
    }
}

__device__ __forceinline__ void wave_lds_fence()
{ // LDS operations of one wave execute in issue order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// four waves per workgroup, one tile of IPW arrays per wave
template <int NREP>
__global__ __launch_bounds__(256) void cache_test_kernel(uint32_t *__restrict__ arrays, uint32_t n, uint64_t narrays,
                                                         uint64_t ntiles, int32_t *__restrict__ sums,
                                                         uint32_t *__restrict__ nerrs, Counters ctr, FaultTab ft,
                                                         const uint32_t *__restrict__ tileList, uint32_t nListed,
                                                         uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    // one launch covers every tile (round 3): a tile an armed fault points into is walked element by element with the injector
    // hooks by its own wave (wave-uniform branch below) -- rounds 1-2 ran those tiles as a second launch on the side stream
    const uint64_t widx = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t tile = tileList ? (widx < nListed ? tileList[widx] : ntiles) : widx;
    const bool tileOk = tile < ntiles;
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range && tileOk)
        fr = ft.range[tile];
    const int slot = lm.q;
    const uint64_t item = tile * IPW + (uint64_t)slot;
    const bool live = tileOk && lm.live && item < narrays;
    uint32_t *a = arrays + (live ? item : 0) * (uint64_t)n;
    const bool vec = ((n & 3u) == 0u) && ((reinterpret_cast<uintptr_t>(arrays) & 15u) == 0u);

    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();

    const bool stepwise = fr.y != 0u; // wave-uniform: an armed fault points into this tile
    const bool cnt = live && lm.r == 0;
    Tally tl;
    uint32_t sum = 0, nerr = 0;

    // injector hook: the running sum (site SITE_CT_SUM), the loaded element (SITE_CT_VAL) or numberOfErrors (SITE_CT_NERR)
    // before / at element `step` (step == n: after the loop).  A tile rarely owns more than a few faults: they are read
    // once into registers instead of being fetched from the table 3 x n times.
    constexpr uint32_t kLocal = 4;
    DevFault lf[kLocal];
    const uint32_t nLocal = fr.y <= kLocal ? fr.y : 0u; // more than that: scan the table
#pragma unroll
    for (uint32_t q = 0; q < kLocal; ++q)
        if (q < nLocal)
            lf[q] = ft.list[fr.x + q];
    auto regHook = [&](uint32_t step, uint32_t site, uint32_t &reg) {
        auto hit = [&](const DevFault &df) {
            if (df.step == step && df.site == site && (int)df.local == slot && (int)df.replica == lm.r && lm.live)
                reg = flip_bit(reg, df.bit, 0xffffffffu);
        };
        if (nLocal) {
#pragma unroll
            for (uint32_t q = 0; q < kLocal; ++q)
                if (q < nLocal)
                    hit(lf[q]);
        } else {
            for (uint32_t q = 0; q < fr.y; ++q)
                hit(ft.list[fr.x + q]);
        }
    };

    // LDS slab of this wave: IPW rows of 8 x 16 bytes + 16 bytes of padding (bank spread)
    __shared__ uint4 sStage[4][IPW * 9];
    uint4 *stage = sStage[threadIdx.x >> 6];
    const int cl = lm.lane & 7, ca = lm.lane >> 3; // cooperative fetch: chunk of the group, array within the instruction
    constexpr int ROUNDS = (IPW + 7) / 8;
    const bool coop = vec; // wave-uniform
    uint4 pre[ROUNDS];
    auto fetch = [&](uint32_t base) { // 32 elements (fewer in the last group) of every array of the tile
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int arr = u * 8 + ca;
            const uint64_t it = tile * IPW + (uint64_t)arr;
            pre[u] = make_uint4(0u, 0u, 0u, 0u);
            if (tileOk && arr < IPW && it < narrays && base + 4u * (uint32_t)cl < n)
                pre[u] = *reinterpret_cast<const uint4 *>(arrays + it * (uint64_t)n + base + 4u * (uint32_t)cl);
        }
    };
    if (coop)
        fetch(0);

    for (uint32_t base = 0; base < n; base += 32u) {
        const uint32_t gcount = (n - base) < 32u ? (n - base) : 32u;
        const uint32_t gmask = gcount == 32u ? 0xffffffffu : ((1u << gcount) - 1u);
        uint32_t mask = 0;
        if (coop) {
            wave_lds_fence(); // the previous group's reads are done
#pragma unroll
            for (int u = 0; u < ROUNDS; ++u)
                if (u * 8 + ca < IPW)
                    stage[(u * 8 + ca) * 9 + cl] = pre[u];
            wave_lds_fence();
            if (base + 32u < n)
                fetch(base + 32u); // in flight under the compares
        }
        const uint4 *row = stage + (lm.live ? slot : 0) * 9;
        if (coop && !stepwise) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = row[u]; // chunks past the end of the array were stored as zeros
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i0 = base + 4u * (uint32_t)u;
                sum += v[u].x + v[u].y + v[u].z + v[u].w;
                mask |= (v[u].x != i0 ? 1u : 0u) << (4 * u);
                mask |= (v[u].y != i0 + 1u ? 1u : 0u) << (4 * u + 1);
                mask |= (v[u].z != i0 + 2u ? 1u : 0u) << (4 * u + 2);
                mask |= (v[u].w != i0 + 3u ? 1u : 0u) << (4 * u + 3);
            }
            mask &= gmask;
        } else { // element by element: ragged / misaligned arrays, and tiles with an armed fault (injector hooks)
            for (uint32_t e = 0; e < gcount; ++e) {
                const uint32_t i = base + e;
                if (stepwise)
                    regHook(i, SITE_CT_SUM, sum);
                uint32_t v = coop ? reinterpret_cast<const uint32_t *>(row)[e] : a[i];
                if (stepwise)
                    regHook(i, SITE_CT_VAL, v);
                sum += v;
                mask |= (v != i ? 1u : 0u) << e;
            }
        }
        const uint32_t voted = xmr_vote_conditions<NREP>(mask, gmask, lm, cnt, tl) & gmask;
        if (stepwise) { // an upset of numberOfErrors lands between the same two increments as in the reference's order
            for (uint32_t e = 0; e < gcount; ++e) {
                regHook(base + e, SITE_CT_NERR, nerr);
                nerr += (voted >> e) & 1u;
            }
        } else {
            nerr += (uint32_t)__builtin_popcount(voted);
        }
        if (cnt && voted) { // the taken branches: array[i] = i (single memory copy, written once)
            for (uint32_t m = voted; m; m &= m - 1u) {
                const uint32_t e = (uint32_t)__builtin_ctz(m);
                a[base + e] = base + e;
            }
        }
    }
    if (stepwise) {
        regHook(n, SITE_CT_SUM, sum);
        regHook(n, SITE_CT_NERR, nerr);
    }
    sum = xmr_sync<NREP>(sum, lm, cnt, tl);         // return-value sync
    nerr = xmr_store_sync<NREP>(nerr, lm, cnt, tl); // stored error count
    uint32_t detItems = 0;
    if (cnt) {
        sums[item] = (int32_t)sum;
        nerrs[item] = nerr;
        if (tl.det) {
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, blockIdx.x);
}

// calc_sum with its loop as written (cacheTest.c:107-131), for COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the loop counter i is a
// replica-private lane register beside sum and numberOfErrors -- one lane per (array, replica), one sequential walk.  Sync points, the
// reference's rule set for -TMR -noMemReplication on the source as written:
//   `i < data_array_elements` at every evaluation (n + 1 per clean array)                                     synchronization.cpp:146-155
//   the GEP offsets: array[i] of `sum += array[i]` and of `array[i] != i` (two loads: off with -noLoadSync), array[i] of the
//     scrub `array[i] = i` (a store: off with -noStoreAddrSync)                                               :333-372, 413-474
//   `array[i] != i` (the data-dependent condition every schedule votes), the data of `array[i] = i` -- the counter itself now
//     (off with -noStoreDataSync), the returned sum, the stored error count
// The printf block of the error branch is I/O outside the batch model, as in the default schedule.  The array is memory: one copy,
// a load uses the original instruction's address in every copy (the voted offset, or replica 0's); the original store (replica
// 0's lane) writes it.  Fault sites: SITE_CT_I / _SUM / _NERR of a replica with `step` = how many loop conditions the call has
// evaluated (the flip lands right before the next one); SITE_CT_VAL = the element loaded in the iteration that condition `step`
// entered.  A wild index reads 0 / stores nothing; a walk that a corrupted counter keeps alive is cut after 4 (n + 1) + 1024
// conditions.  The sync-point-parity form of the kernel, not the throughput form.
template <int NREP>
__global__ __launch_bounds__(64) void cache_test_indexed_kernel(uint32_t *__restrict__ arrays, uint32_t n, uint64_t narrays,
                                                                int32_t *__restrict__ sums, uint32_t *__restrict__ nerrs,
                                                                Counters ctr, FaultTab ft, uint8_t *__restrict__ detected)
{
    __shared__ uint32_t sCnt[4];
    constexpr int IPW = LaneMap<NREP>::kItemsPerWave;
    LaneMap<NREP> lm;
    lm.storeSync = !(ctr.flags & kFlagNoStoreDataSync);
    const bool bs = (ctr.flags & kFlagBranchSync) != 0u, as = (ctr.flags & kFlagAddrSync) != 0u;
    const bool ls = as && !(ctr.flags & kFlagNoLoadSync), ss = as && !(ctr.flags & kFlagNoStoreAddrSync);
    const bool lss = xmr_local_sync_on(ctr.flags); // COAST_F_LOCAL_STORE_SYNC: sum += .., numberOfErrors++, local_errors++, i++ are stores at -O0
    const uint32_t tile = blockIdx.x;
    const int slot = lm.q;
    const uint64_t item = (uint64_t)tile * IPW + (uint64_t)slot;
    const bool live = lm.live && item < narrays;
    const bool cnt = live && lm.r == 0;
    const bool writer = cnt; // the single memory copy is written by the original store (replica 0's lane)
    uint32_t *a = arrays + (live ? item : 0) * (uint64_t)n;
    if (threadIdx.x < 4)
        sCnt[threadIdx.x] = 0;
    __syncthreads();
    uint2 fr = make_uint2(0u, 0u);
    if (ft.range)
        fr = ft.range[tile];
    const uint32_t N = live ? n : 0u;
    const uint64_t cap = 4ull * ((uint64_t)n + 1ull) + 1024ull;
    Tally tl;
    uint32_t i = 0u, sum = 0u, nerr = 0u;
    bool firstError = false, inBlock = false; // the report block's state: equal in every copy (no fault site)
    uint32_t localErrors = 0u;
    uint64_t tick = 0;
    auto hook = [&](uint64_t step, bool loaded, uint32_t &v) __attribute__((always_inline)) {
        for (uint32_t q = 0; q < fr.y; ++q) {
            const DevFault df = ft.list[fr.x + q];
            if ((uint64_t)df.step != step || (int)df.local != slot || (int)df.replica != lm.r || !lm.live)
                continue;
            const uint32_t m = 1u << (df.bit & 31u);
            if (loaded) {
                if (df.site == SITE_CT_VAL)
                    v ^= m;
            } else if (df.site == SITE_CT_I)
                i ^= m;
            else if (df.site == SITE_CT_SUM)
                sum ^= m;
            else if (df.site == SITE_CT_NERR)
                nerr ^= m;
        }
    };
    uint32_t none = 0u;
    for (;;) {                                                       // for (i = 0; i < data_array_elements; i++)   :107
        hook(tick, false, none);
        if (tick >= cap)
            break;
        const uint64_t t = tick++;
        // (the idle lane of a TMR wave has no array of its own -- its replica group would wrap to lanes 0, 1: N = 0, it leaves)
        const bool go = lm.live ? xmr_steer<NREP>(i < N ? 1u : 0u, lm, bs, cnt, tl) != 0u : i < N;
        if (!go)
            break;
        const uint32_t o1 = xmr_steer<NREP>(i, lm, ls, cnt, tl);     // sum += array[i]                             :108
        uint32_t v = o1 < N ? a[o1] : 0u;
        hook(t, true, v);
        sum = xmr_local_sync<NREP>(sum + v, lm, lss, cnt, tl);
        (void)xmr_steer<NREP>(i, lm, ls, cnt, tl);                   // if (array[i] != i): the same offset, voted again :110
        const bool taken = xmr_steer<NREP>(v != i ? 1u : 0u, lm, true, cnt, tl) != 0u; // (the element as loaded above)
        if (taken) {
            nerr = xmr_local_sync<NREP>(nerr + 1u, lm, lss, cnt, tl); // numberOfErrors++                            :111
            // the report block (:114-131), its printing aside: `if (!first_error)`, for the first bad element `if (!in_block && ..)`
            // (in_block and local_errors are the program's globals: 0 when the call starts, in this batch model), and the
            // `array[i]` argument of the printf -- one more load offset
            if (xmr_steer<NREP>(!firstError ? 1u : 0u, lm, bs, cnt, tl) != 0u) {
                (void)xmr_steer<NREP>(!inBlock ? 1u : 0u, lm, bs, cnt, tl);
                firstError = true, inBlock = true;
            }
            (void)xmr_steer<NREP>(i, lm, ls, cnt, tl);
            localErrors += 1u;
            const uint32_t os = xmr_steer<NREP>(i, lm, ss, cnt, tl); // array[i] = i                                :127
            uint32_t d = xmr_store_sync<NREP>(i, lm, cnt, tl);
            if (NREP != 3 || !lm.storeSync)
                d = xmr_rep0<NREP>(d, lm);
            if (writer && os < N)
                a[os] = d;
            (void)xmr_local_sync<NREP>(localErrors, lm, lss, cnt, tl); // local_errors++ (a global: equal in every copy)     :128
        }
        i = xmr_local_sync<NREP>(i + 1u, lm, lss, cnt, tl);
    }
    // after the loop: `if (first_error && robust_printing)` (:139) and `if (sum != golden)` (:157), golden = n (n - 1) / 2; a wrong sum
    // looks at `local_errors == 0` (:161) and, with no element error behind it, at `!in_block` (:165)
    if (lm.live) {
        (void)xmr_steer<NREP>(firstError ? 1u : 0u, lm, bs, cnt, tl);
        const uint32_t golden = (uint32_t)(((uint64_t)N * (uint64_t)(N - 1u)) / 2u);
        if (xmr_steer<NREP>(sum != golden ? 1u : 0u, lm, bs, cnt, tl) != 0u)
            if (xmr_steer<NREP>(localErrors == 0u ? 1u : 0u, lm, bs, cnt, tl) != 0u) {
                (void)xmr_local_sync<NREP>(1u, lm, lss, cnt, tl); // sum_errors++; local_errors++ (globals)           :162-163
                (void)xmr_local_sync<NREP>(1u, lm, lss, cnt, tl);
                (void)xmr_steer<NREP>(!inBlock ? 1u : 0u, lm, bs, cnt, tl);
            }
    }
    sum = xmr_sync<NREP>(sum, lm, cnt, tl);          // return sum
    nerr = xmr_store_sync<NREP>(nerr, lm, cnt, tl);  // stored to the caller's error count
    uint32_t detItems = 0;
    if (cnt) {
        sums[item] = (int32_t)sum;
        nerrs[item] = nerr;
        if (tl.det) { // unequal copies at a sync point of this array (DWC: detected, TMR: corrected)
            if (NREP == 2)
                detItems = 1;
            if (detected)
                detected[item] = 1;
        }
    }
    block_tally(tl.miss, tl.syncs, detItems, sCnt, ctr, tile);
}

} // namespace coast
