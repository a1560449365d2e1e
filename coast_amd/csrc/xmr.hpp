// xmr.hpp -- lane-replicated execution primitives for gfx950 (wave64).
//
// The reference replicates every instruction of a protected region 2 or 3 times in the IR of a single CPU thread
// (cloning.cpp:2187-2209) and inserts voters at sync points (synchronization.cpp:741-949).  Here the replicas of one
// logical work item are NREP ADJACENT LANES of a wavefront: lane = NREP*q + r holds replica r of the wave's q-th item,
// a wave carries 64/NREP items (21 for TMR + one idle lane, 32 for DWC), and a sync point is a cross-lane exchange.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace coast {

constexpr int kWave = 64;
constexpr int kCounterSlots = 256; // per-workgroup counter slots, one 64-byte line each
constexpr int kSlotStride = 8;     // uint64 per slot: {errors, syncs, dwc_items, pad...}

// decoded fault descriptor written by the injector (side stream), sorted by workgroup
struct DevFault {
    uint32_t block;   // workgroup (logical id) that owns the item
    uint32_t local;   // kernel specific: item slot inside the workgroup (+ element inside a register tile)
    uint32_t step;
    uint8_t replica, site, bit, index;
};

struct FaultTab {
    const DevFault *list;   // sorted by block
    const uint2 *range;     // per logical block: {first index, count}
};

struct Counters {
    unsigned long long *slots; // [kCounterSlots][kSlotStride]
    uint32_t flags;            // coast_cfg.flags of the launch (kFlagNoStoreDataSync), read by the general kernels
    // In-kernel fold (block_fold; aes-128's persistent kernels): with `totals` set, the last workgroup to leave folds the slots into the
    // totals itself -- no reduce_counters_kernel behind the launch, one dependent dispatch less per step.  Null everywhere else.
    uint32_t foldLaunches;      // what the fold adds to totals[3]: the launches since the last fold, this one included
    unsigned long long *totals; // {errors, syncs, dwc_items, launches}
    uint32_t *ticket;           // kTicketWords words: the top ticket and kTicketGroups group tickets, a 128-byte line each (block_fold)
};
// block_fold's completion count is two-level: a workgroup draws the ticket of its group (blockIdx.x % kTicketGroups), the last one of a group
// draws the top ticket, the last of those folds.  Same-address device-scope atomics retire at ~10 ns each (tools/aes_fixed_probe.hip): one
// word for 256 / 512 workgroups that leave together was 2.7 / 5.2 us of every launch; 16 lines take 16-32 arrivals each, side by side.
constexpr uint32_t kTicketGroups = 16;
constexpr uint32_t kTicketStride = 32; // uint32 words between two tickets: one 128-byte line each
constexpr uint32_t kTicketWords = (1 + kTicketGroups) * kTicketStride;
constexpr uint32_t kFlagNoStoreDataSync = 1u; // == COAST_F_NO_STORE_DATA_SYNC
constexpr uint32_t kFlagBranchSync = 2u;      // == COAST_F_BRANCH_SYNC: loop / byte counters are replica-private, their branch conditions voted
constexpr uint32_t kFlagAddrSync = 4u;        // == COAST_F_ADDR_SYNC: GEP offsets built from them are voted ...
constexpr uint32_t kFlagNoLoadSync = 8u;      // == COAST_F_NO_LOAD_SYNC: ... except load addresses
constexpr uint32_t kFlagNoStoreAddrSync = 16u; // == COAST_F_NO_STORE_ADDR_SYNC: ... except store addresses
constexpr uint32_t kFlagIndexed = kFlagBranchSync | kFlagAddrSync;
constexpr uint32_t kFlagO0Shape = 128u;       // == COAST_F_O0_SHAPE: sha256's walk in the -O0 IR's shape
constexpr uint32_t kFlagLocalStoreSync = 64u; // == COAST_F_LOCAL_STORE_SYNC: the -O0 IR's stores into locals / in-place arrays are data votes

template <int NREP> struct LaneMap {
    static constexpr int kItemsPerWave = kWave / NREP;
    int lane;  // 0..63
    int q;     // item slot in the wave
    int r;     // replica id
    bool live; // false for the idle lane(s) when 64 % NREP != 0
    int base4; // byte address (lane*4) of replica 0 of this item, for ds_bpermute
    bool storeSync = true; // false under -noStoreDataSync (general kernels only): xmr_store_sync passes values through
    __device__ __forceinline__ LaneMap() : LaneMap((int)(threadIdx.x & (kWave - 1))) {}
    // from an explicit lane id -- xmr_fresh_lane(): a map for a kernel's epilogue that keeps no register alive across its main loop
    __device__ __forceinline__ explicit LaneMap(int laneId)
    {
        lane = laneId;
        q = lane / NREP;
        r = lane - q * NREP;
        live = q < kItemsPerWave;
        base4 = (lane - r) * 4;
    }
};

// the lane id, recomputed where it is used (v_mbcnt inside an asm volatile: nothing is hoisted or kept from an earlier copy)
__device__ __forceinline__ int xmr_fresh_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Running tallies of one lane; only replica 0 of a live item tallies, so every sync point counts once.
struct Tally {
    uint32_t miss = 0;  // voted values whose copies were not all equal      (TMR_ERROR_CNT)
    uint32_t syncs = 0; // sync points executed                               (__SYNC_COUNT)
    uint32_t det = 0;   // a sync point of the current item saw unequal copies (DWC: detected, TMR: corrected)
};

// One sync point on a 32-bit value.
//   TMR  vote = (a == b) ? a : c, whole-value compare (synchronization.cpp:934-938); miss = !((a==b)&&(a==c))
//        (:1391-1400); all three replicas continue from the voted value (:527-529).
//   DWC  mismatch = (a != b) (:1117-1192); replicas keep their own value, the item is flagged.
// `count` gates the tallies (false for padding elements and idle lanes); the exchange itself is wave-wide.
template <int NREP>
__device__ __forceinline__ uint32_t xmr_sync(uint32_t v, const LaneMap<NREP> &lm, bool count, Tally &t)
{
    if constexpr (NREP == 3) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4, (int)v);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4 + 4, (int)v);
        const uint32_t c = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4 + 8, (int)v);
        const bool e01 = (a == b), e02 = (a == c);
        if (count) {
            const uint32_t m = (e01 && e02) ? 0u : 1u;
            t.syncs += 1;
            t.miss += m;
            t.det |= m;
        }
        return e01 ? a : c;
    } else if constexpr (NREP == 2) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4, (int)v);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4 + 4, (int)v);
        if (count) {
            t.syncs += 1;
            t.det |= (a != b) ? 1u : 0u;
        }
        return v;
    } else {
        return v;
    }
}

// Final (store-data / return-value) sync point, DPP form: only replica 0 of each item consumes the result, so it needs
// just its two neighbours -- lane+1 and lane+2 via whole-wave DPP shifts (wave_shl:1: lane i reads lane i+1) -- instead of
// three ds_bpermute round trips through the LDS crossbar (24 cycles each, tools/valu_microbench).  Same voter and counter
// rules as xmr_sync; the returned value is meaningful in replica-0 lanes only.
// Sync point on the DATA of a store (synchronization.cpp:476-561).  -noStoreDataSync (:197-224, :324) drops it: every
// replica keeps its own value, replica 0's reaches memory, nothing is counted.
template <int NREP>
__device__ __forceinline__ uint32_t xmr_store_sync(uint32_t v, const LaneMap<NREP> &lm, bool count, Tally &t)
{
    if (!lm.storeSync)
        return v;
    return xmr_sync<NREP>(v, lm, count, t);
}

// COAST_F_LOCAL_STORE_SYNC: the data vote of a store into a local's alloca or into an array in place, on the -O0 IR
// (synchronization.cpp:197-224, 476-561).  `on` = the flag is set and -noStoreDataSync is not.  TMR: every copy continues from the
// voted value (what the single memory copy holds and every copy reloads); DWC: compared, the copies keep their values.
__device__ __forceinline__ bool xmr_local_sync_on(uint32_t flags)
{
    return (flags & kFlagLocalStoreSync) != 0u && (flags & kFlagNoStoreDataSync) == 0u;
}
template <int NREP>
__device__ __forceinline__ uint32_t xmr_local_sync(uint32_t v, const LaneMap<NREP> &lm, bool on, bool count, Tally &t)
{
    return on ? xmr_sync<NREP>(v, lm, count, t) : v;
}

// replica 0's copy of a value: what the ORIGINAL instruction consumes when a use is not a sync point
template <int NREP> __device__ __forceinline__ uint32_t xmr_rep0(uint32_t v, const LaneMap<NREP> &lm)
{
    if constexpr (NREP == 1)
        return v;
    else
        return (uint32_t)__builtin_amdgcn_ds_bpermute(lm.base4, (int)v);
}

// A sync point that STEERS: a branch condition (syncTerminator, synchronization.cpp:741-949) or a GEP offset (syncGEP,
// :413-474).  Synchronised, TMR: every copy continues with the voted value.  DWC: the copies are compared (a mismatch flags
// the item) and the original instruction's operand -- replica 0's -- is used; not synchronised: replica 0's as well.
template <int NREP>
__device__ __forceinline__ uint32_t xmr_steer(uint32_t v, const LaneMap<NREP> &lm, bool synced, bool count, Tally &t)
{
    if (synced) {
        const uint32_t voted = xmr_sync<NREP>(v, lm, count, t);
        if (NREP == 3)
            return voted;
    }
    return xmr_rep0<NREP>(v, lm);
}

// voter of a final sync point on explicit copies: v = this replica, b / c = the next two (replica-0 lanes consume it)
template <int NREP>
__device__ __forceinline__ uint32_t xmr_final_vote_vals(uint32_t v, uint32_t b, uint32_t c, bool count, Tally &t)
{
    if constexpr (NREP == 1) {
        return v;
    } else if constexpr (NREP == 2) {
        if (count) {
            t.syncs += 1;
            t.det |= (v != b) ? 1u : 0u;
        }
        return v;
    } else {
        const bool e01 = (v == b), e02 = (v == c);
        if (count) {
            const uint32_t m = (e01 && e02) ? 0u : 1u;
            t.syncs += 1;
            t.miss += m;
            t.det |= m;
        }
        return e01 ? v : c;
    }
}

template <int NREP>
__device__ __forceinline__ uint32_t xmr_final_vote_dpp(uint32_t v, bool count, Tally &t)
{
    if constexpr (NREP == 1) {
        return v;
    } else {
        constexpr int kWaveShl1 = 0x130; // DPP_WF_SL1
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kWaveShl1, 0xf, 0xf, false);
        const uint32_t c = NREP == 3 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, kWaveShl1, 0xf, 0xf, false) : 0u;
        return xmr_final_vote_vals<NREP>(v, b, c, count, t);
    }
}

// Lanes of one wave that hand data to each other through LDS: the hardware executes a wave's LDS operations in issue order,
// but the compiler treats lanes as independent threads and may forward / reorder a lane's own stores and loads (e.g. lay out
// the readers' side of a branch before the writer's).  This pins the order at the hand-over point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += (uint32_t)__shfl_xor((int)v, o, kWave);
    return v;
}

// Mismatch / sync / detected-item counters: lane -> wave (shuffle) -> LDS (one atomic per wave) -> one global
// atomic per workgroup into the workgroup's slot.  `s_cnt` is 3 x uint32 of LDS, zeroed by the caller before a barrier.
__device__ __forceinline__ void block_tally(uint32_t miss, uint32_t syncs, uint32_t detItems, uint32_t *s_cnt,
                                            const Counters &ctr, uint32_t slotKey)
{
    const uint32_t wm = wave_sum(miss), ws = wave_sum(syncs), wd = wave_sum(detItems);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (wm)
            atomicAdd(&s_cnt[0], wm);
        if (ws)
            atomicAdd(&s_cnt[1], ws);
        if (wd)
            atomicAdd(&s_cnt[2], wd);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *slot = ctr.slots + (size_t)(slotKey % kCounterSlots) * kSlotStride;
        if (s_cnt[0])
            atomicAdd(slot + 0, (unsigned long long)s_cnt[0]);
        if (s_cnt[1])
            atomicAdd(slot + 1, (unsigned long long)s_cnt[1]);
        if (s_cnt[2])
            atomicAdd(slot + 2, (unsigned long long)s_cnt[2]);
    }
}

// The counter fold inside the protected kernel (Counters::totals set): every workgroup calls this behind block_tally, all threads.  The
// workgroup's slot atomics are released in front of its ticket (two-level: kTicketGroups); the workgroup that draws the last ticket reads and clears every slot with
// atomics (the slots were written with atomics from every XCD: the exchange runs where they ran), sums them per wave and adds the sums to
// the totals.  `slotKey`: block_tally's; `s_flag`: one uint32 of LDS nobody else writes between block_tally's barrier and the kernel's end.
__device__ __forceinline__ void block_fold(const Counters &ctr, uint32_t slotKey, uint32_t *s_flag)
{
    if (!ctr.totals)
        return;
    if (threadIdx.x == 0) {
        // (no __threadfence: on gfx950 that is a write-back of the XCD's L2, full of the launch's own dirty output -- measured + 8 us per
        // launch.  Slots, ticket and totals are only ever touched by device-scope atomics, which execute past the L2s: it is enough that
        // this thread's slot atomics have been acknowledged before its ticket is drawn.)
        // Acknowledged is taken literally: a RETURNING read-modify-write on the slot's own 64-byte line (its first pad word; the line's
        // atomics execute in one L2 channel, in order) has its value back before the ticket is drawn.  (+ 0 on the counter words themselves
        // would do, but the compiler turns an idempotent atomic into a load.)
        unsigned long long *mine = ctr.slots + (size_t)(slotKey % kCounterSlots) * kSlotStride;
        // (an exchange with 0: returning, not idempotent for the compiler, and the pad word stays 0 -- ADVICE r5: an add of 1 accumulated there.
        // On gfx9 the s_waitcnt vmcnt(0) behind it also covers the non-returning slot atomics by itself -- they count in vmcnt until the L2
        // acknowledges them --, so the order holds twice over.)
        const unsigned long long r = atomicExch(mine + 3, 0ull);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(r) : "memory");
        // two-level ticket: group g = blockIdx.x % kTicketGroups holds the workgroups g, g + 16, ...; whoever sees the group complete clears
        // the group's word (nobody of this launch touches it again) and draws the top ticket.  The order argument is transitive: every
        // member's slot atomics were acknowledged in front of its group ticket, the group's last ticket in front of the top one.
        const uint32_t g = blockIdx.x % kTicketGroups;
        const uint32_t members = (gridDim.x - g + kTicketGroups - 1u) / kTicketGroups;
        const uint32_t groups = gridDim.x < kTicketGroups ? gridDim.x : kTicketGroups;
        uint32_t *gt = ctr.ticket + (size_t)kTicketStride * (1u + g);
        uint32_t last = 0u;
        if (atomicAdd(gt, 1u) == members - 1u) {
            atomicExch(gt, 0u);
            last = atomicAdd(ctr.ticket, 1u) == groups - 1u ? 1u : 0u;
        }
        *s_flag = last;
    }
    __syncthreads();
    if (*s_flag == 0u)
        return;
    unsigned long long v[3] = {0ull, 0ull, 0ull};
    for (uint32_t t = threadIdx.x; t < (uint32_t)kCounterSlots; t += blockDim.x) {
        unsigned long long *slot = ctr.slots + (size_t)t * kSlotStride;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v[c] += atomicExch(slot + c, 0ull);
    }
    if (threadIdx.x < (uint32_t)kCounterSlots) { // (waves past the slots hold zeros)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                v[c] += (unsigned long long)__shfl_xor((long long)v[c], o, kWave);
            if ((threadIdx.x & (kWave - 1)) == 0 && v[c])
                atomicAdd(&ctr.totals[c], v[c]);
        }
    }
    if (threadIdx.x == 0) {
        atomicExch(ctr.ticket, 0u);
        if (ctr.foldLaunches)
            atomicAdd(&ctr.totals[3], (unsigned long long)ctr.foldLaunches);
    }
}

// XCD-aware logical block id: the dispatcher places hardware block b on XCD b % 8 (each XCD has a private L2), so
// give every XCD one contiguous range of logical blocks -- neighbours that share inputs then share an L2.
__device__ __forceinline__ uint32_t xcd_logical_block(uint32_t hw, uint32_t nblocks)
{
    constexpr uint32_t X = 8;
    const uint32_t per = nblocks / X, rem = nblocks % X; // XCD x owns per + (x < rem) blocks
    const uint32_t x = hw % X, i = hw / X;
    return x * per + (x < rem ? x : rem) + i;
}

__device__ __forceinline__ uint32_t flip_bit(uint32_t v, uint32_t bit, uint32_t liveMask)
{
    return v ^ ((1u << (bit & 31u)) & liveMask); // flipOneBit, injector.py:202-207
}

} // namespace coast
