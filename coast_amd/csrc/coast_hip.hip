// coast_hip.hip -- C ABI of libcoast_hip.so (declared in include/coast_hip.h) and the host side of every launch.
// Unity build: the kernels are included so that one hipcc invocation produces the whole gfx950 code object.
#include "../../include/coast_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "xmr.hpp"
#include "injector.hip"
#include "mm_kernel.hip"
#include "mm_mfma_kernel.hip"
#include "mm_mfma_blk2_kernel.hip"
#include "mm_mfma_blk3_kernel.hip"
#include "mm_mfma_blk4_kernel.hip"
namespace coast { // compiled in mm_phys_instances.hip
#define X(...) extern template __global__ void __VA_ARGS__(COAST_MMARGS);
#include "mm_phys_instances.inc"
#undef X
} // namespace coast
#include "sha256_kernel.hip"
#include "aes_kernel.hip"
#include "crc16_kernel.hip"
#include "vote_kernel.hip"
#include "cache_test_kernel.hip"
#include "chsha_kernel.hip"
#include "quicksort_kernel.hip"
#include "crazycf_kernel.hip"
#include "chaes_kernel.hip"

using namespace coast;

static_assert(sizeof(coast_fault) == 16, "coast_fault layout");
static_assert(sizeof(DevFault) == 16, "DevFault layout");

struct coast_ctx {
    int device = 0;
    hipStream_t stream = nullptr; // protected kernels
    hipStream_t side = nullptr;   // injector: descriptor upload + indexing
    hipEvent_t evArmed = nullptr;    // side -> main: fault table ready
    hipEvent_t evMainReady = nullptr; // main -> side: inputs of this launch are complete
    hipEvent_t evSideDone = nullptr;  // side -> main: stepwise (injector) kernel finished

    unsigned long long *dSlots = nullptr;  // [kCounterSlots][kSlotStride]
    unsigned long long *dTotals = nullptr; // internal totals
    unsigned long long *dBound = nullptr;  // caller-owned totals (optional)
    unsigned long long pendingLaunches = 0;
    bool slotsDirty = false; // a launch since the last fold left its counts in the slots (a kernel that folds in its own exit path does not)

    std::vector<coast_fault> armed; // host copy of the faults waiting for the next launch
    // Fault tables are double-buffered: launch i uploads into buffer i & 1 while launch i-1 may still be reading the
    // other one, so the injector never waits for the kernel it follows.
    struct FaultBuf {
        DevFault *hPinned = nullptr;
        size_t pinnedCap = 0;
        DevFault *dList = nullptr;
        size_t listCap = 0;
        uint2 *dRange = nullptr;
        size_t rangeCap = 0;
        uint32_t *hBlocks = nullptr; // pinned: distinct workgroups / tiles that own >= 1 armed fault
        uint32_t *dBlocks = nullptr;
        size_t blocksCap = 0;
        hipEvent_t evConsumed = nullptr; // main -> side: the kernels reading this buffer have finished
        bool consumedPending = false;
        hipEvent_t evUploaded = nullptr; // side: the H2D copies out of hPinned / hBlocks have finished
        bool uploadPending = false;
        // what dList / dRange / dBlocks hold (their host images are hPinned / hBlocks): a launch that arms the very table this buffer
        // already carries -- a campaign or a bench step that repeats its upsets -- reuses it instead of uploading it again
        size_t residentK = 0, residentBlocks = 0;
        uint32_t residentNblocks = 0;
    } fb[2];
    unsigned armCount = 0; // armed launches so far; selects the buffer
    int curBuf = 0;

    uint16_t *dCrcTable = nullptr; // 64 Ki x u16 two-byte-step table of the crc16 stream kernel
    coast::CfcDevTables *dCfcTables = nullptr; // crazyCF's signature tables (coast_crazycf_batch)
    int numCUs = 256;

    coast_launch_info last = {};   // what the most recent protected launch dispatched to
    double hbmBytes = 0.0;         // algorithmic HBM bytes of the launches since the last reset
    // Optional kernel timing (coast_set_profiling): a pair of timing events around every protected launch on the
    // context's stream, folded into kernelMs whenever the stream is known to be idle.
    bool profiling = false;
    unsigned profEvery = 1, profSeen = 0; // coast_set_profiling(ctx, n > 1): every n-th launch is bracketed, its time counted n times
    struct EvPair {
        hipEvent_t a, b;
        unsigned weight;
    };
    std::vector<EvPair> evPending, evFree;
    double kernelMs = 0.0;
    bool evOpen = false;

    std::string err;
};

namespace {

int fail(coast_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                                    \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess)                                                                                \
            return fail((ctx), COAST_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                        __LINE__);                                                                            \
    } while (0)

// the counter slots, and behind them the ticket words (kTicketWords: top + group tickets) of the kernels that fold the slots themselves (Counters::ticket)
constexpr size_t kSlotBytes = sizeof(unsigned long long) * (size_t)kCounterSlots * kSlotStride + sizeof(uint32_t) * (size_t)kTicketWords;
uint32_t *ticket_of(coast_ctx *c) { return reinterpret_cast<uint32_t *>(c->dSlots + (size_t)kCounterSlots * kSlotStride); }
unsigned long long *totals_of(coast_ctx *c) { return c->dBound ? c->dBound : c->dTotals; }

// `indexedOk`: this kernel implements the index-in-the-sphere-of-replication flags (mm, sha256, crc16)
int check_cfg(coast_ctx *ctx, const coast_cfg *cfg, bool indexedOk = false, bool copiesOk = false, bool o0Ok = false)
{
    if (!ctx)
        return COAST_EINVAL;
    if (!cfg || cfg->replicas < 1 || cfg->replicas > 3)
        return fail(ctx, COAST_EINVAL, "coast_cfg.replicas must be 1 (none), 2 (DWC) or 3 (TMR)");
    const uint32_t indexed = COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC | COAST_F_NO_LOAD_SYNC | COAST_F_NO_STORE_ADDR_SYNC;
    if (cfg->flags & ~((uint32_t)COAST_F_NO_STORE_DATA_SYNC | indexed | (uint32_t)COAST_F_MEMORY_COPIES | (uint32_t)COAST_F_LOCAL_STORE_SYNC |
                       (uint32_t)COAST_F_O0_SHAPE))
        return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: unknown bits", cfg->flags);
    if ((cfg->flags & COAST_F_O0_SHAPE) && (!o0Ok || (cfg->flags & (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC)) != (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC)))
        return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: COAST_F_O0_SHAPE selects sha256's -O0 walk under COAST_F_BRANCH_SYNC | "
                                       "COAST_F_ADDR_SYNC (every other statement-by-statement walk IS the -O0 shape)", cfg->flags);
    if ((cfg->flags & COAST_F_LOCAL_STORE_SYNC) &&
        (cfg->flags & (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC)) != (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC))
        return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: COAST_F_LOCAL_STORE_SYNC qualifies COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC "
                                       "(the statement-by-statement walks)", cfg->flags);
    if (cfg->flags & COAST_F_MEMORY_COPIES) {
        if (!copiesOk)
            return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: COAST_F_MEMORY_COPIES is implemented for sha256, aes128 and crc16",
                        cfg->flags);
        if (cfg->flags != COAST_F_MEMORY_COPIES || cfg->sync_every)
            return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: COAST_F_MEMORY_COPIES runs on the lean kernels: no sync_every, no "
                                           "other flag", cfg->flags);
    }
    if ((cfg->flags & indexed) && !indexedOk)
        return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC are implemented for mm, sha256, "
                                       "aes128, crc16, cache_test, CHStone sha and CHStone aes (quicksort votes its indices by default)", cfg->flags);
    if ((cfg->flags & (COAST_F_NO_LOAD_SYNC | COAST_F_NO_STORE_ADDR_SYNC)) && !(cfg->flags & COAST_F_ADDR_SYNC))
        return fail(ctx, COAST_EINVAL, "coast_cfg.flags 0x%x: -noLoadSync / -noStoreAddrSync qualify COAST_F_ADDR_SYNC", cfg->flags);
    return COAST_OK;
}

// Kernel-specific decode of one armed fault: returns false when the fault addresses nothing in this launch
// (item out of range, replica >= replicas) -- such a fault is dropped exactly as the oracle ignores it.
typedef bool (*decode_fn)(const coast_fault &, const void *geom, DevFault &);

// Upload the armed faults for a launch of `nblocks` workgroups.  Runs entirely on the side stream; the main stream
// waits on evArmed.  Returns haveFaults (0/1) through *have.
int arm_faults_impl(coast_ctx *c, uint32_t nblocks, decode_fn dec, const void *geom, FaultTab *ft, int *have,
                    const uint32_t **dBlockList, uint32_t *nFaultBlocks, bool fullKeyOrder)
{
    *have = 0;
    if (dBlockList)
        *dBlockList = nullptr;
    if (nFaultBlocks)
        *nFaultBlocks = 0;
    ft->list = nullptr;
    ft->range = nullptr;
    if (c->armed.empty())
        return COAST_OK;
    // The armed list is consumed by exactly one launch -- but only once its table is safely enqueued: if anything below
    // fails, the faults stay armed for the caller's retry.
    struct Restore {
        coast_ctx *c;
        std::vector<coast_fault> taken;
        bool commit = false;
        ~Restore()
        {
            if (!commit)
                c->armed.insert(c->armed.begin(), taken.begin(), taken.end());
        }
    } guard{c, {}};
    guard.taken.swap(c->armed);
    std::vector<DevFault> dv;
    dv.reserve(guard.taken.size());
    for (const coast_fault &f : guard.taken) {
        DevFault d{};
        if (f.replica == COAST_REPLICA_ALL) { // common-mode upset: the same flip in every replica's copy (the decoder drops
            for (uint8_t r = 0; r < 3; ++r) { // replica numbers the launch does not have)
                coast_fault fr = f;
                fr.replica = r;
                if (dec(fr, geom, d))
                    dv.push_back(d);
            }
        } else if (dec(f, geom, d)) {
            dv.push_back(d);
        }
    }
    if (dv.empty()) {
        guard.commit = true; // nothing in the list addresses this launch: dropped, as the oracle ignores them
        return COAST_OK;
    }
    if (fullKeyOrder) // consumers that walk a workgroup's faults element by element, in program order of the upsets
        std::stable_sort(dv.begin(), dv.end(), [](const DevFault &a, const DevFault &b) {
            if (a.block != b.block)
                return a.block < b.block;
            if (a.local != b.local)
                return a.local < b.local;
            if (a.step != b.step)
                return a.step < b.step;
            return a.site < b.site;
        });
    else
        std::stable_sort(dv.begin(), dv.end(), [](const DevFault &a, const DevFault &b) { return a.block < b.block; });
    std::vector<uint32_t> blocks;
    for (const DevFault &d : dv)
        if (blocks.empty() || blocks.back() != d.block)
            blocks.push_back(d.block);

    HIP_TRY(c, hipSetDevice(c->device));
    const int nextBuf = (int)(c->armCount & 1u);
    coast_ctx::FaultBuf *b = &c->fb[nextBuf];
    if (b->uploadPending) { // this buffer's pinned staging area was last read by the upload two armed launches ago
        HIP_TRY(c, hipEventSynchronize(b->evUploaded));
        b->uploadPending = false;
    }
    if (b->residentK == dv.size() && b->residentNblocks == nblocks && b->residentBlocks == blocks.size() && b->residentK <= b->pinnedCap &&
        !memcmp(b->hPinned, dv.data(), dv.size() * sizeof(DevFault)) && !memcmp(b->hBlocks, blocks.data(), blocks.size() * sizeof(uint32_t))) {
        // the table is resident (uploaded two armed launches ago, in front of a launch this stream has long passed): nothing to send
        c->curBuf = nextBuf;
        c->armCount += 1;
        guard.commit = true;
        c->last.armed_faults = dv.size();
        ft->list = b->dList;
        ft->range = b->dRange;
        *have = 1;
        if (dBlockList)
            *dBlockList = b->dBlocks;
        if (nFaultBlocks)
            *nFaultBlocks = (uint32_t)blocks.size();
        return COAST_OK;
    }
    b->residentK = 0; // (set again below, once the new table is enqueued)
    if (dv.size() > b->pinnedCap) {
        if (b->hPinned)
            HIP_TRY(c, hipHostFree(b->hPinned));
        b->pinnedCap = std::max<size_t>(dv.size() * 2, 1024);
        HIP_TRY(c, hipHostMalloc((void **)&b->hPinned, b->pinnedCap * sizeof(DevFault), hipHostMallocDefault));
    }
    if (blocks.size() > b->blocksCap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (b->hBlocks)
            HIP_TRY(c, hipHostFree(b->hBlocks));
        if (b->dBlocks)
            HIP_TRY(c, hipFree(b->dBlocks));
        b->blocksCap = std::max<size_t>(blocks.size() * 2, 1024);
        HIP_TRY(c, hipHostMalloc((void **)&b->hBlocks, b->blocksCap * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(c, hipMalloc((void **)&b->dBlocks, b->blocksCap * sizeof(uint32_t)));
    }
    if (b->consumedPending) { // the launch before last may still be reading this buffer on the main stream
        HIP_TRY(c, hipStreamWaitEvent(c->side, b->evConsumed, 0));
        b->consumedPending = false;
    }
    if (dv.size() > b->listCap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (b->dList)
            HIP_TRY(c, hipFree(b->dList));
        b->listCap = std::max<size_t>(dv.size() * 2, 1024);
        HIP_TRY(c, hipMalloc((void **)&b->dList, b->listCap * sizeof(DevFault)));
    }
    if ((size_t)nblocks > b->rangeCap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (b->dRange)
            HIP_TRY(c, hipFree(b->dRange));
        b->rangeCap = std::max<size_t>((size_t)nblocks * 2, 4096);
        HIP_TRY(c, hipMalloc((void **)&b->dRange, b->rangeCap * sizeof(uint2)));
    }
    memcpy(b->hPinned, dv.data(), dv.size() * sizeof(DevFault));
    HIP_TRY(c, hipMemcpyAsync(b->dList, b->hPinned, dv.size() * sizeof(DevFault), hipMemcpyHostToDevice, c->side));
    memcpy(b->hBlocks, blocks.data(), blocks.size() * sizeof(uint32_t));
    HIP_TRY(c, hipMemcpyAsync(b->dBlocks, b->hBlocks, blocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->side));
    HIP_TRY(c, hipEventRecord(b->evUploaded, c->side));
    b->uploadPending = true;
    HIP_TRY(c, hipMemsetAsync(b->dRange, 0, (size_t)nblocks * sizeof(uint2), c->side));
    const uint32_t k = (uint32_t)dv.size();
    hipLaunchKernelGGL(fault_range_kernel, dim3((k + 255) / 256), dim3(256), 0, c->side, b->dList, k, b->dRange);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->evArmed, c->side));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evArmed, 0));
    c->curBuf = nextBuf;
    c->armCount += 1;
    guard.commit = true;
    b->residentK = dv.size();
    b->residentBlocks = blocks.size();
    b->residentNblocks = nblocks;
    c->last.armed_faults = dv.size();
    ft->list = b->dList;
    ft->range = b->dRange;
    *have = 1;
    if (dBlockList)
        *dBlockList = b->dBlocks;
    if (nFaultBlocks)
        *nFaultBlocks = (uint32_t)blocks.size();
    return COAST_OK;
}

// Kernel timing (coast_set_profiling): fold every finished event pair into kernelMs.  `wait`: the stream is known to be idle
// (or the pool is full), so block on the stragglers.
int profile_fold(coast_ctx *c, bool wait)
{
    size_t keep = 0, i = 0;
    int rc = COAST_OK;
    for (; i < c->evPending.size(); ++i) {
        coast_ctx::EvPair e = c->evPending[i];
        if (wait && hipEventSynchronize(e.b) != hipSuccess) {
            rc = fail(c, COAST_EHIP, "hipEventSynchronize failed on a timing event");
            break;
        }
        float ms = 0.f;
        const hipError_t q = hipEventElapsedTime(&ms, e.a, e.b);
        if (q == hipSuccess) {
            c->kernelMs += (double)ms * (double)e.weight;
            c->evFree.push_back(e);
        } else if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            c->evPending[keep++] = e;
        } else {
            rc = fail(c, COAST_EHIP, "hipEventElapsedTime failed: %s", hipGetErrorString(q));
            break;
        }
    }
    // on an error the unprocessed tail stays pending (compacted): nothing is listed twice, nothing is reused while pending
    for (; i < c->evPending.size(); ++i)
        c->evPending[keep++] = c->evPending[i];
    c->evPending.resize(keep);
    return rc;
}

int profile_begin(coast_ctx *c)
{
    if (c->evOpen && !c->evPending.empty()) { // the previous launch failed between its bracket's two records: the pair's `b`
        c->evFree.push_back(c->evPending.back()); // was never recorded and would poison every later fold
        c->evPending.pop_back();
    }
    c->evOpen = false;
    if (!c->profiling)
        return COAST_OK;
    if (c->profEvery > 1 && (c->profSeen++ % c->profEvery) != 0)
        return COAST_OK; // (a sampled bracket: this launch runs without timing events in front of and behind it)
    if (c->evPending.size() >= 1024) {
        int rc = profile_fold(c, true);
        if (rc)
            return rc;
    }
    coast_ctx::EvPair e{};
    if (!c->evFree.empty()) {
        e = c->evFree.back();
        c->evFree.pop_back();
    } else {
        HIP_TRY(c, hipEventCreate(&e.a));
        if (hipEventCreate(&e.b) != hipSuccess) {
            (void)hipEventDestroy(e.a);
            return fail(c, COAST_EHIP, "hipEventCreate failed");
        }
    }
    if (hipEventRecord(e.a, c->stream) != hipSuccess) {
        c->evFree.push_back(e);
        return fail(c, COAST_EHIP, "hipEventRecord failed on a timing event");
    }
    e.weight = c->profEvery > 1 ? c->profEvery : 1u;
    c->evPending.push_back(e);
    c->evOpen = true;
    return COAST_OK;
}

// Every protected launch starts here: reset the launch record, upload the armed faults on the side stream (the main stream
// waits for the table), open the timing bracket behind that wait.
int arm_faults(coast_ctx *c, uint32_t nblocks, decode_fn dec, const void *geom, FaultTab *ft, int *have,
               const uint32_t **dBlockList = nullptr, uint32_t *nFaultBlocks = nullptr, bool fullKeyOrder = false)
{
    c->last = coast_launch_info{};
    int rc = arm_faults_impl(c, nblocks, dec, geom, ft, have, dBlockList, nFaultBlocks, fullKeyOrder);
    if (rc)
        return rc;
    return profile_begin(c);
}

// ... and ends here.  engine / generalBlocks / fastBlocks: what was dispatched (coast_last_launch_info); algBytes: the
// algorithmic HBM bytes of the launch (DESIGN.md section 4 gives the per-unit figures), accumulated into coast_stats.
int after_launch(coast_ctx *c, int haveFaults, uint32_t engine = COAST_ENGINE_NONE, uint64_t generalBlocks = 0,
                 uint64_t fastBlocks = 0, double algBytes = 0.0)
{
    HIP_TRY(c, hipGetLastError());
    c->pendingLaunches += 1;
    c->slotsDirty = true;
    if (c->evOpen) {
        HIP_TRY(c, hipEventRecord(c->evPending.back().b, c->stream));
        c->evOpen = false;
    }
    c->last.engine = engine;
    c->last.general_blocks = generalBlocks;
    c->last.fast_blocks = fastBlocks;
    c->last.algorithmic_bytes = algBytes;
    c->hbmBytes += algBytes;
    if (haveFaults) {
        coast_ctx::FaultBuf *b = &c->fb[c->curBuf];
        HIP_TRY(c, hipEventRecord(b->evConsumed, c->stream));
        b->consumedPending = true;
    }
    return COAST_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------ context
extern "C" int coast_abi_version(void) { return COAST_HIP_ABI_VERSION; }

#ifndef COAST_SOURCE_HASH
#define COAST_SOURCE_HASH "unknown"
#endif
extern "C" const char *coast_source_hash(void) { return COAST_SOURCE_HASH; }

extern "C" int coast_create(coast_ctx **out, int device)
{
    if (!out)
        return COAST_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return COAST_ENODEV;
    coast_ctx *c = new coast_ctx();
    c->device = device;
    auto bail = [&](hipError_t e) {
        if (e == hipSuccess)
            return false;
        delete c;
        return true;
    };
    if (bail(hipSetDevice(device)) || bail(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)) ||
        bail(hipEventCreateWithFlags(&c->evArmed, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->fb[0].evConsumed, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->fb[1].evConsumed, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->fb[0].evUploaded, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->fb[1].evUploaded, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->evMainReady, hipEventDisableTiming)) ||
        bail(hipEventCreateWithFlags(&c->evSideDone, hipEventDisableTiming)) ||
        bail(hipMalloc((void **)&c->dSlots, kSlotBytes)) ||
        bail(hipMalloc((void **)&c->dTotals, sizeof(unsigned long long) * 4)) ||
        bail(hipMemset(c->dSlots, 0, kSlotBytes)) ||
        bail(hipMemset(c->dTotals, 0, sizeof(unsigned long long) * 4)))
        return COAST_EHIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->numCUs = prop.multiProcessorCount;
    *out = c;
    return COAST_OK;
}

extern "C" void coast_destroy(coast_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->side);
    for (coast_ctx::FaultBuf &b : c->fb) {
        if (b.hPinned)
            (void)hipHostFree(b.hPinned);
        if (b.dList)
            (void)hipFree(b.dList);
        if (b.dRange)
            (void)hipFree(b.dRange);
        if (b.hBlocks)
            (void)hipHostFree(b.hBlocks);
        if (b.dBlocks)
            (void)hipFree(b.dBlocks);
        if (b.evConsumed)
            (void)hipEventDestroy(b.evConsumed);
        if (b.evUploaded)
            (void)hipEventDestroy(b.evUploaded);
    }
    for (auto *v : {&c->evPending, &c->evFree})
        for (coast_ctx::EvPair &e : *v) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
    if (c->dCrcTable)
        (void)hipFree(c->dCrcTable);
    if (c->dCfcTables)
        (void)hipFree(c->dCfcTables);
    (void)hipFree(c->dSlots);
    (void)hipFree(c->dTotals);
    (void)hipEventDestroy(c->evArmed);
    (void)hipEventDestroy(c->evMainReady);
    (void)hipEventDestroy(c->evSideDone);
    (void)hipStreamDestroy(c->side);
    delete c;
}

extern "C" const char *coast_last_error(const coast_ctx *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int coast_set_stream(coast_ctx *c, void *hip_stream)
{
    if (!c)
        return COAST_EINVAL;
    if (c->stream != (hipStream_t)hip_stream)
        for (auto &b : c->fb) // a resident fault table is ordered before the launches of the stream it was armed for: a new stream
            b.residentK = 0;  // has no dependency on it -- the next armed launch uploads and orders its table again
    c->stream = (hipStream_t)hip_stream;
    return COAST_OK;
}

extern "C" int coast_bind_counters(coast_ctx *c, uint64_t *d_totals)
{
    if (!c)
        return COAST_EINVAL;
    c->dBound = (unsigned long long *)d_totals;
    return COAST_OK;
}

extern "C" int coast_reduce_counters(coast_ctx *c)
{
    if (!c)
        return COAST_EINVAL;
    if (!c->slotsDirty && c->pendingLaunches == 0)
        return COAST_OK; // nothing launched since the last fold, or the last launch folded in its own exit path (block_fold)
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(kCounterSlots), 0, c->stream, c->dSlots, totals_of(c),
                       c->pendingLaunches);
    HIP_TRY(c, hipGetLastError());
    c->pendingLaunches = 0;
    c->slotsDirty = false;
    return COAST_OK;
}

// The one collective of the path, for C hosts (one process per GPU, or one process driving several contexts): fold this
// context's counter slots, then ncclAllReduce(SUM) the four uint64 totals IN PLACE on the context's stream.  RCCL is
// resolved at first use (dlopen) so that single-GPU hosts carry no dependency on it.
extern "C" int coast_allreduce_counters(coast_ctx *c, void *rccl_comm)
{
    if (!c || !rccl_comm)
        return COAST_EINVAL;
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static allreduce_fn fn = nullptr;
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!fn) {
            void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h)
                h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h)
                return fail(c, COAST_EHIP, "coast_allreduce_counters: cannot load librccl.so (%s)", dlerror());
            fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
            if (!fn)
                return fail(c, COAST_EHIP, "coast_allreduce_counters: librccl.so has no ncclAllReduce");
        }
    }
    int rc = coast_reduce_counters(c);
    if (rc)
        return rc;
    enum { kNcclUint64 = 5, kNcclSum = 0 }; // rccl.h: ncclDataType_t / ncclRedOp_t
    const int nr = fn(totals_of(c), totals_of(c), 4, kNcclUint64, kNcclSum, rccl_comm, c->stream);
    if (nr != 0)
        return fail(c, COAST_EHIP, "ncclAllReduce failed with %d", nr);
    return COAST_OK;
}

extern "C" int coast_read_stats(coast_ctx *c, coast_stats *out)
{
    if (!c || !out)
        return COAST_EINVAL;
    int rc = coast_reduce_counters(c);
    if (rc)
        return rc;
    unsigned long long h[4];
    HIP_TRY(c, hipMemcpyAsync(h, totals_of(c), sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->errors_corrected = h[0];
    out->sync_count = h[1];
    out->dwc_detected = h[2];
    out->launches = h[3];
    if (c->evOpen && !c->evPending.empty()) { // a failed launch left its bracket open: drop it (see profile_begin)
        c->evFree.push_back(c->evPending.back());
        c->evPending.pop_back();
        c->evOpen = false;
    }
    rc = profile_fold(c, true); // the stream is idle: every bracket has closed
    if (rc)
        return rc;
    out->kernel_ms = c->kernelMs;
    out->hbm_bytes = c->hbmBytes;
    return COAST_OK;
}

extern "C" int coast_set_profiling(coast_ctx *c, int enable)
{
    if (!c)
        return COAST_EINVAL;
    c->profiling = enable != 0;
    c->profEvery = enable > 1 ? (unsigned)enable : 1u;
    c->profSeen = 0;
    return COAST_OK;
}

extern "C" int coast_last_launch_info(const coast_ctx *c, coast_launch_info *out)
{
    if (!c || !out)
        return COAST_EINVAL;
    *out = c->last;
    return COAST_OK;
}

extern "C" int coast_reset_stats(coast_ctx *c)
{
    if (!c)
        return COAST_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->dSlots, 0, kSlotBytes, c->stream));
    HIP_TRY(c, hipMemsetAsync(totals_of(c), 0, sizeof(unsigned long long) * 4, c->stream));
    c->pendingLaunches = 0;
    c->slotsDirty = false;
    // timing brackets still in flight belong to the period being discarded
    for (coast_ctx::EvPair &e : c->evPending)
        c->evFree.push_back(e);
    c->evPending.clear();
    c->evOpen = false;
    c->kernelMs = 0.0;
    c->hbmBytes = 0.0;
    return COAST_OK;
}

extern "C" int coast_inject_faults(coast_ctx *c, const coast_fault *faults, size_t k)
{
    if (!c || (k && !faults))
        return COAST_EINVAL;
    c->armed.insert(c->armed.end(), faults, faults + k);
    return COAST_OK;
}

// ------------------------------------------------------------------------------------------------ mm
namespace {

struct MmHostGeom {
    MmGeom g;
    size_t batch;
    int replicas;
    int tpb;
    int panelRows = 64; // matrix-core kernels: rows of a matrix per workgroup (mm_mfma_blk4_kernel: 128)
};

bool decode_mm(const coast_fault &f, const void *gp, DevFault &d)
{
    const MmHostGeom &h = *(const MmHostGeom *)gp;
    const uint64_t nn = (uint64_t)h.g.n * h.g.n;
    if (f.item >= nn * h.batch || f.replica >= h.replicas)
        return false;
    if (f.site > COAST_SITE_MM_OPB || f.step > (uint32_t)h.g.n)
        return false;
    const uint64_t mat = f.item / nn, e = f.item % nn;
    const int i = (int)(e / h.g.n), j = (int)(e % h.g.n);
    const int tile = (i / 4) * h.g.tc + (j / 4);
    const int bim = tile / h.tpb, tb = tile % h.tpb;
    d.block = (uint32_t)(mat * h.g.bpm + bim);
    d.local = ((uint32_t)tb << 4) | (uint32_t)((i & 3) << 2) | (uint32_t)(j & 3);
    d.step = f.step;
    d.replica = f.replica;
    d.site = f.site;
    d.bit = f.bit;
    d.index = f.index;
    return true;
}

// side 256 on the matrix cores: workgroup = 64 (mm_mfma_blk4_kernel: 128) rows of one matrix, element = (row in the panel, column)
bool decode_mm_mfma(const coast_fault &f, const void *gp, DevFault &d)
{
    const MmHostGeom &h = *(const MmHostGeom *)gp;
    const uint64_t nn = (uint64_t)h.g.n * h.g.n;
    if (f.item >= nn * h.batch || f.replica >= h.replicas)
        return false;
    if (f.site != COAST_SITE_MM_VGPR && f.site != COAST_SITE_MM_PREG && (f.site > COAST_SITE_MM_OPB || f.step > (uint32_t)h.g.n))
        return false;
    if (f.site == COAST_SITE_MM_PREG && ((f.step & 63u) >= 60u || (((f.step >> 19) & 1u) ? ((f.step >> 20) & 511u) >= 102u : ((f.step >> 20) & 511u) >= 256u)))
        return false; // slot 0..59; v0..v255 / s0..s101
    const uint64_t mat = f.item / nn, e = f.item % nn;
    const uint32_t i = (uint32_t)(e / h.g.n), j = (uint32_t)(e % h.g.n);
    const uint32_t pr = (uint32_t)h.panelRows;
    d.block = (uint32_t)(mat * (uint64_t)((uint32_t)h.g.n / pr) + i / pr);
    d.local = ((i % pr) << 8) | j;
    d.step = f.step;
    d.replica = f.replica;
    d.site = f.site;
    d.bit = f.bit;
    d.index = f.index;
    return true;
}

// counters inside the sphere of replication: the item is the call (one matrix per lane group); sites i / j / k / sum
bool decode_mm_indexed(const coast_fault &f, const void *gp, DevFault &d)
{
    const MmHostGeom &h = *(const MmHostGeom *)gp;
    const uint64_t nn = (uint64_t)h.g.n * h.g.n;
    if (f.item >= nn * h.batch || f.replica >= h.replicas)
        return false;
    if (f.site != COAST_SITE_MM_ACC && (f.site < COAST_SITE_MM_I || f.site > COAST_SITE_MM_K))
        return false;
    const uint64_t mat = f.item / nn;
    const uint64_t ipw = 64u / (uint64_t)h.replicas;
    d.block = (uint32_t)(mat / ipw);
    d.local = (uint32_t)(mat % ipw);
    d.step = f.step;
    d.replica = f.replica;
    d.site = f.site;
    d.bit = f.bit;
    d.index = f.index;
    return true;
}

} // namespace

extern "C" int coast_mm_batch(coast_ctx *c, const uint32_t *d_f, const uint32_t *d_s, uint32_t *d_r, int n,
                              size_t batch, const coast_cfg *cfgIn, uint8_t *d_detected)
{
    // COAST_F_CLONE_STAGING / COAST_F_SINGLE_STAGING are properties of the matrix-core kernels' staging path, not sync-point rules: they
    // select no stepwise kernel and the rule checks below do not see them.  Cloned is the default for replicas >= 2 (ABI 8): the pass clones
    // every load (cloning.cpp:2187-2209, 2247-2255)
    coast_cfg cfgRules{};
    bool cloneStaging = false;
    if (cfgIn) {
        cfgRules = *cfgIn;
        cloneStaging = cfgIn->replicas > 1 && (cfgIn->flags & COAST_F_SINGLE_STAGING) == 0;
        cfgRules.flags &= ~(uint32_t)(COAST_F_CLONE_STAGING | COAST_F_SINGLE_STAGING);
    }
    const coast_cfg *cfg = cfgIn ? &cfgRules : nullptr;
    int rc = check_cfg(c, cfg, true);
    if (rc)
        return rc;
    if ((cfg->flags & (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC)) && cfg->sync_every)
        return fail(c, COAST_EINVAL, "coast_mm_batch: sync_every belongs to the per-element schedule; with the loop counters inside "
                                     "the sphere of replication every loop condition is a sync point already");
    if (batch == 0)
        return COAST_OK; // an empty batch is a no-op (armed faults stay armed for the next real launch)
    if (!d_f || !d_s || !d_r || n < 1 || n > 4096)
        return fail(c, COAST_EINVAL, "coast_mm_batch: bad pointers or side %d (1..4096)", n);
    if ((((uintptr_t)d_f | (uintptr_t)d_s | (uintptr_t)d_r) & ((n & 3) == 0 ? 15u : 3u)) != 0)
        return fail(c, COAST_EINVAL, "coast_mm_batch: f, s, r must be %d-byte aligned for side %d (16-byte vector accesses "
                                     "when the side is a multiple of 4)", (n & 3) == 0 ? 16 : 4, n);
    HIP_TRY(c, hipSetDevice(c->device));
    if (cfg->flags & (COAST_F_BRANCH_SYNC | COAST_F_ADDR_SYNC)) {
        // the three loops as written, i / j / k / sum replica-private: one call per lane group (mm_indexed_kernel)
        MmHostGeom hi;
        hi.g.n = n;
        hi.batch = batch;
        hi.replicas = (int)cfg->replicas;
        hi.tpb = 0;
        const uint64_t ipw = 64u / cfg->replicas;
        const uint64_t ntiles = ((uint64_t)batch + ipw - 1) / ipw;
        if (ntiles > 0x7fffffffull)
            return fail(c, COAST_EINVAL, "coast_mm_batch: %llu tiles exceed the grid limit", (unsigned long long)ntiles);
        FaultTab fti;
        int havei = 0;
        rc = arm_faults(c, (uint32_t)ntiles, decode_mm_indexed, &hi, &fti, &havei);
        if (rc)
            return rc;
        if (!havei)
            fti.list = nullptr, fti.range = nullptr;
        Counters ctri{c->dSlots, cfg->flags};
        if (cfg->replicas == 3)
            hipLaunchKernelGGL(mm_indexed_kernel<3>, dim3((uint32_t)ntiles), dim3(64), 0, c->stream, d_f, d_s, d_r, (uint32_t)n,
                               (uint64_t)batch, ctri, fti, d_detected);
        else if (cfg->replicas == 2)
            hipLaunchKernelGGL(mm_indexed_kernel<2>, dim3((uint32_t)ntiles), dim3(64), 0, c->stream, d_f, d_s, d_r, (uint32_t)n,
                               (uint64_t)batch, ctri, fti, d_detected);
        else
            hipLaunchKernelGGL(mm_indexed_kernel<1>, dim3((uint32_t)ntiles), dim3(64), 0, c->stream, d_f, d_s, d_r, (uint32_t)n,
                               (uint64_t)batch, ctri, fti, d_detected);
        return after_launch(c, havei, COAST_ENGINE_STEPWISE, ntiles, 0, 12.0 * (double)n * (double)n * (double)batch);
    }

    MmHostGeom h;
    MmGeom &g = h.g;
    const int ipw = 64 / (int)cfg->replicas;
    h.tpb = 4 * ipw;
    h.batch = batch;
    h.replicas = (int)cfg->replicas;
    g.n = n;
    g.tc = (n + 3) / 4;
    g.tiles = g.tc * g.tc;
    g.bpm = (g.tiles + h.tpb - 1) / h.tpb;
    g.npad = 4 * g.tc;
    int maxTileRows = 1;
    for (int b = 0; b < g.bpm; ++b) {
        const int t0 = b * h.tpb, t1 = std::min(t0 + h.tpb, g.tiles) - 1;
        maxTileRows = std::max(maxTileRows, t1 / g.tc - t0 / g.tc + 1);
    }
    g.rs = 4 * maxTileRows;
    // k-chunk: largest power of two <= 16 whose panels a workgroup can stage with its fixed per-thread slots and
    // whose two LDS buffers leave room for >= 3 workgroups per CU
    g.kt = 16; // the fast kernel is instantiated for kt in {16, 4, 1}
    while (g.kt > 1 && ((size_t)g.kt * (g.npad / 4) > 256u * kMmMaxB || (size_t)g.kt * g.rs > 256u * kMmMaxA ||
                        (size_t)2 * g.kt * (g.rs + g.npad) * 4 > 52 * 1024))
        g.kt >>= 2;
    if ((size_t)g.kt * (g.npad / 4) > 256u * kMmMaxB || (size_t)g.kt * g.rs > 256u * kMmMaxA)
        return fail(c, COAST_EINVAL, "coast_mm_batch: side %d exceeds the staging geometry", n);
    g.ktLog2 = 0;
    while ((1 << g.ktLog2) < g.kt)
        ++g.ktLog2;
    const size_t lds = (size_t)2 * g.kt * (g.rs + g.npad) * 4 + 16;
    const uint64_t nb = (uint64_t)g.bpm * batch;
    if (nb > 0x7fffffffull)
        return fail(c, COAST_EINVAL, "coast_mm_batch: %llu workgroups exceed the grid limit", (unsigned long long)nb);
    g.nblocks = (uint32_t)nb;

    // extra sync points, or -noStoreDataSync: every workgroup takes the stepwise kernel
    const bool allGeneral = cfg->sync_every != 0 || cfg->flags != 0;
    // side 256: the int8-MFMA limb kernel, injector hooks included; COAST_MM_ENGINE=valu selects the v_mad_u64_u32 kernels
    const char *eng = getenv("COAST_MM_ENGINE");
    const bool mfma = n == 256 && !allGeneral && !(eng && !strcmp(eng, "valu"));
    // TMR: replicas in register blocks, two waves per SIMD (mm_mfma_blk3_kernel); COAST_MM_TILE=lanes selects the lane-replica kernel
    // (mm_mfma_panel_kernel: north_star's layout, three adjacent lanes and a cross-lane voter), which also serves DWC and the unprotected
    // mode there; COAST_MM_TILE=blocks (the one-wave-per-SIMD predecessor) was retired in round 5
    const char *tileEnv = getenv("COAST_MM_TILE");
    const bool mmBlocks = !(tileEnv && !strcmp(tileEnv, "lanes"));
    const bool mmBlocks2 = !(tileEnv && !strcmp(tileEnv, "blocks"));
    // default (round 4): mm_mfma_blk3_kernel -- blocks2's geometry with every loaded operand replicated (a replica's MFMAs read their
    // own A fragments); COAST_MM_TILE=blocks2 selects mm_mfma_blk2_kernel (one A fragment set for the three replicas)
    const bool mmBlocks3 = !(tileEnv && !strcmp(tileEnv, "blocks2"));
    // mm_mfma_blk4_kernel (round 6): a workgroup owns 128 rows (two column-tile lanes x four row quarters), s is converted twice per matrix
    // instead of four times.  TMR without physical-register upsets (those name mm_mfma_blk3_kernel's registers)
    // (COAST_SITE_MM_VGPR names mm_mfma_blk3_kernel's registers: such a launch runs there; COAST_SITE_MM_PREG -- any physical register -- has
    // an instantiation in both kernels)
    bool haveVgprEarly = false;
    for (const coast_fault &af : c->armed)
        haveVgprEarly = haveVgprEarly || af.site == COAST_SITE_MM_VGPR;
    // (the default TMR kernel since round 6; COAST_MM_TILE=blocks3 | blocks2 | lanes select the older ones)
    const bool mmPanel128 = mfma && cfg->replicas == 3 && (!tileEnv || !*tileEnv || !strcmp(tileEnv, "panel128")) && !haveVgprEarly;
    if (mmPanel128)
        h.panelRows = MmBlk4::BM;
    const uint64_t nbm = (uint64_t)(n / h.panelRows) * batch; // workgroups' panels: 64 (128) rows of one matrix each
    if (mfma && nbm > 0x7fffffffull)
        return fail(c, COAST_EINVAL, "coast_mm_batch: %llu workgroups exceed the grid limit", (unsigned long long)nbm);

    // COAST_SITE_MM_VGPR: a physical upset of a named vector register of the register-block matrix-core kernel (its PHYS instantiation)
    bool havePhys = false;
    for (const coast_fault &af : c->armed)
        havePhys = havePhys || af.site == COAST_SITE_MM_VGPR || af.site == COAST_SITE_MM_PREG;
    bool havePreg = false; // ... of ANY register, by physical number: the instantiation with a hook in front of every MFMA slot
    for (const coast_fault &af : c->armed)
        havePreg = havePreg || af.site == COAST_SITE_MM_PREG;
    if (haveVgprEarly && !(mfma && mmBlocks && mmBlocks2 && mmBlocks3))
        return fail(c, COAST_EINVAL, "coast_mm_batch: COAST_SITE_MM_VGPR names a register of mm_mfma_blk3_kernel: side 256, no sync_every / "
                                     "flags, COAST_MM_ENGINE at its default, COAST_MM_TILE unset or blocks3");
    if (havePreg && !(mfma && mmBlocks2 && mmBlocks3))
        return fail(c, COAST_EINVAL, "coast_mm_batch: COAST_SITE_MM_PREG is hooked into mm_mfma_blk4_kernel, mm_mfma_blk3_kernel and "
                                     "mm_mfma_panel_kernel: side 256, no sync_every / flags, COAST_MM_TILE unset, panel128, blocks3 or lanes");
    (void)havePhys;
    FaultTab ft;
    int have = 0;
    const uint32_t *dBlockList = nullptr;
    uint32_t nFaultBlocks = 0;
    if (mfma)
        rc = arm_faults(c, (uint32_t)nbm, decode_mm_mfma, &h, &ft, &have, nullptr, &nFaultBlocks, true);
    else
        rc = arm_faults(c, g.nblocks, decode_mm, &h, &ft, &have, &dBlockList, &nFaultBlocks);
    if (rc)
        return rc;
    Counters ctr{c->dSlots, cfg->flags};
    const dim3 block(256);
    uint32_t engine = COAST_ENGINE_VALU;
    uint64_t generalBlocks = 0, fastBlocks = g.nblocks;
    uint32_t hookedBlocks = 0;
#define LAUNCH_FAST(R, V, K)                                                                                    \
    do {                                                                                                        \
        if (lds > 64 * 1024)                                                                                    \
            HIP_TRY(c, hipFuncSetAttribute((const void *)mm_fast_kernel<R, V, K>,                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
        FaultTab ftk = ft;                                                                                      \
        if (!have)                                                                                              \
            ftk.list = nullptr, ftk.range = nullptr;                                                            \
        hipLaunchKernelGGL((mm_fast_kernel<R, V, K>), dim3(g.nblocks), block, lds, c->stream, d_f, d_s, d_r, g, \
                           ctr, ftk, d_detected);                                                               \
    } while (0)
#define LAUNCH_FAST_K(R, V)                                                                                     \
    do {                                                                                                        \
        if (g.kt == 16)                                                                                         \
            LAUNCH_FAST(R, V, 16);                                                                              \
        else if (g.kt == 4)                                                                                     \
            LAUNCH_FAST(R, V, 4);                                                                               \
        else                                                                                                    \
            LAUNCH_FAST(R, V, 1);                                                                               \
    } while (0)
    /* mm_mfma_blk3_kernel<R, FLAGS, PHYS, CLONE>: FLAGS = per-item flags wanted (the physical-upset instantiations always carry them; the  \
     * unprotected mode has none), PHYS = 1: COAST_SITE_MM_VGPR hooks, 2: + COAST_SITE_MM_PREG, CLONE = COAST_F_CLONE_STAGING (replicas > 1) */ \
#define LAUNCH_BLK3_ONE(R, FL, PH, CL)                                                                          \
    do {                                                                                                        \
        using G3 = MmBlk2<R>;                                                                                   \
        HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_blk3_kernel<R, FL, PH, CL>,                        \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G3::LDS_BYTES));        \
        hipLaunchKernelGGL((mm_mfma_blk3_kernel<R, FL, PH, CL>), dim3(gridB), dim3(G3::NTHR), G3::LDS_BYTES,    \
                           c->stream, d_f, d_s, d_r, (uint32_t)batch, ctr, ftm, d_detected);                    \
    } while (0)
#define LAUNCH_BLK3(R)                                                                                          \
    do {                                                                                                        \
        constexpr bool canClone = (R) > 1;                                                                      \
        const bool wantFlags = d_detected != nullptr && (R) > 1;                                                \
        if (cloneStaging && canClone) {                                                                         \
            if (havePreg)                                                                                       \
                LAUNCH_BLK3_ONE(R, true, 2, canClone);                                                          \
            else if (havePhys)                                                                                  \
                LAUNCH_BLK3_ONE(R, true, 1, canClone);                                                          \
            else if (wantFlags)                                                                                 \
                LAUNCH_BLK3_ONE(R, true, 0, canClone);                                                          \
            else                                                                                                \
                LAUNCH_BLK3_ONE(R, false, 0, canClone);                                                         \
        } else if (havePreg)                                                                                    \
            LAUNCH_BLK3_ONE(R, true, 2, false);                                                                 \
        else if (havePhys)                                                                                      \
            LAUNCH_BLK3_ONE(R, true, 1, false);                                                                 \
        else if (wantFlags)                                                                                     \
            LAUNCH_BLK3_ONE(R, true, 0, false);                                                                 \
        else                                                                                                    \
            LAUNCH_BLK3_ONE(R, false, 0, false);                                                                \
    } while (0)
#define LAUNCH_BLK4_ONE(FL, CL, PH)                                                                             \
    do {                                                                                                        \
        HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_blk4_kernel<FL, CL, PH>,                           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)MmBlk4::LDS_BYTES));    \
        hipLaunchKernelGGL((mm_mfma_blk4_kernel<FL, CL, PH>), dim3(gridP), dim3(MmBlk4::NTHR), MmBlk4::LDS_BYTES, \
                           c->stream, d_f, d_s, d_r, (uint32_t)batch, ctr, ftm, d_detected);                    \
    } while (0)
#define LAUNCH_MM(R)                                                                                            \
    do {                                                                                                        \
        if (mmPanel128 && R == 3) { /* TMR on a 128-row panel: two workgroups (one XCD) share a matrix */       \
            FaultTab ftm = ft;                                                                                  \
            if (!have)                                                                                          \
                ftm.list = nullptr, ftm.range = nullptr;                                                        \
            const uint32_t gridP = 2u * (uint32_t)std::min<uint64_t>((uint64_t)batch, (uint64_t)std::max(1, c->numCUs / 2)); \
            if (have)                                                                                           \
                hookedBlocks = nFaultBlocks;                                                                    \
            if (cloneStaging) {                                                                                 \
                if (havePreg) /* (the physical-upset instantiation always carries the per-item flags) */        \
                    LAUNCH_BLK4_ONE(true, true, 2);                                                             \
                else if (d_detected)                                                                            \
                    LAUNCH_BLK4_ONE(true, true, 0);                                                             \
                else                                                                                            \
                    LAUNCH_BLK4_ONE(false, true, 0);                                                            \
            } else if (havePreg)                                                                                \
                LAUNCH_BLK4_ONE(true, false, 2);                                                                \
            else if (d_detected)                                                                                \
                LAUNCH_BLK4_ONE(true, false, 0);                                                                \
            else                                                                                                \
                LAUNCH_BLK4_ONE(false, false, 0);                                                               \
            engine = COAST_ENGINE_MATRIX_CORE;                                                                  \
            fastBlocks = nbm;                                                                                   \
            break;                                                                                              \
        }                                                                                                       \
        if (mfma && R == 3 && mmBlocks) { /* TMR: replicas in register blocks */              \
            FaultTab ftm = ft;                                                                                  \
            if (!have)                                                                                          \
                ftm.list = nullptr, ftm.range = nullptr;                                                        \
            /* one workgroup per CU; four of them (one XCD) share a matrix */                                    \
            const uint32_t gridB = 4u * (uint32_t)std::min<uint64_t>((uint64_t)batch, (uint64_t)std::max(1, c->numCUs / 4)); \
            if (have) /* the armed upsets are applied inside the matrix-core kernels: the panels that do it */    \
                hookedBlocks = nFaultBlocks;                                                                    \
            if (mmBlocks2 && mmBlocks3) {                                                                       \
                LAUNCH_BLK3(3);                                                                                 \
            } else if (mmBlocks2) {                                                                             \
                using G2 = MmBlk2<3>;                                                                           \
                if (d_detected) {                                                                               \
                    HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_blk2_kernel<3, true>,                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2::LDS_BYTES)); \
                    hipLaunchKernelGGL((mm_mfma_blk2_kernel<3, true>), dim3(gridB), dim3(G2::NTHR), G2::LDS_BYTES, \
                                       c->stream, d_f, d_s, d_r, (uint32_t)batch, ctr, ftm, d_detected);        \
                } else {                                                                                        \
                    HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_blk2_kernel<3, false>,                 \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2::LDS_BYTES)); \
                    hipLaunchKernelGGL((mm_mfma_blk2_kernel<3, false>), dim3(gridB), dim3(G2::NTHR), G2::LDS_BYTES, \
                                       c->stream, d_f, d_s, d_r, (uint32_t)batch, ctr, ftm, d_detected);        \
                }                                                                                               \
            } else {                                                                                            \
                return fail(c, COAST_EINVAL, "coast_mm_batch: COAST_MM_TILE=blocks (mm_mfma_blk_kernel, one wave per SIMD) was retired in round 5: " \
                                             "blocks3 (default), blocks2 or lanes");                          \
            }                                                                                                   \
            engine = COAST_ENGINE_MATRIX_CORE;                                                                  \
            fastBlocks = nbm;                                                                                   \
            break;                                                                                              \
        }                                                                                                       \
        if (mfma && mmBlocks && mmBlocks2 && mmBlocks3) { /* DWC / unprotected (round 4): the register-block kernel with four / two sets per step */ \
            using G2 = MmBlk2<R>;                                                                               \
            FaultTab ftm = ft;                                                                                  \
            if (!have)                                                                                          \
                ftm.list = nullptr, ftm.range = nullptr;                                                        \
            const uint32_t gridB = 4u * (uint32_t)std::min<uint64_t>((uint64_t)batch, (uint64_t)std::max(1, c->numCUs / 4)); \
            if (have)                                                                                           \
                hookedBlocks = nFaultBlocks;                                                                    \
            LAUNCH_BLK3(R);                                                                                     \
            engine = COAST_ENGINE_MATRIX_CORE;                                                                  \
            fastBlocks = nbm;                                                                                   \
            break;                                                                                              \
        }                                                                                                       \
        if (mfma) { /* armed upsets are applied and out-voted inside the panel kernel: no VALU workgroup runs */ \
            using GP = MmPanel<R>;                                                                              \
            static_assert(GP::BPM == 256 / 64, "panel geometry");                                               \
            FaultTab ftm = ft;                                                                                  \
            if (!have)                                                                                          \
                ftm.list = nullptr, ftm.range = nullptr;                                                        \
            if (havePreg) { /* the instantiation with a hook in front of every MFMA slot */                     \
                HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_panel_kernel<R, 2>,                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GP::LDS_BYTES)); \
                hipLaunchKernelGGL((mm_mfma_panel_kernel<R, 2>), dim3((uint32_t)nbm), dim3(GP::NTHR), GP::LDS_BYTES, \
                                   c->stream, d_f, d_s, d_r, (uint32_t)nbm, ctr, ftm, d_detected);              \
            } else {                                                                                            \
                HIP_TRY(c, hipFuncSetAttribute((const void *)mm_mfma_panel_kernel<R>,                           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GP::LDS_BYTES)); \
                hipLaunchKernelGGL(mm_mfma_panel_kernel<R>, dim3((uint32_t)nbm), dim3(GP::NTHR), GP::LDS_BYTES, \
                                   c->stream, d_f, d_s, d_r, (uint32_t)nbm, ctr, ftm, d_detected);              \
            }                                                                                                   \
            if (have)                                                                                           \
                hookedBlocks = nFaultBlocks;                                                                    \
            engine = COAST_ENGINE_MATRIX_CORE;                                                                  \
            fastBlocks = nbm;                                                                                   \
            break;                                                                                              \
        }                                                                                                       \
        if (lds > 64 * 1024)                                                                                    \
            HIP_TRY(c, hipFuncSetAttribute((const void *)mm_general_kernel<R>,                                  \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
        if (allGeneral) {                                                                                       \
            hipLaunchKernelGGL(mm_general_kernel<R>, dim3(g.nblocks), block, lds, c->stream, d_f, d_s, d_r, g,  \
                               cfg->sync_every, ctr, ft, (const uint32_t *)nullptr, d_detected);                \
            engine = COAST_ENGINE_STEPWISE;                                                                     \
            generalBlocks = g.nblocks;                                                                          \
            fastBlocks = 0;                                                                                     \
        } else {                                                                                                \
            /* side 256 on the VALU engine (COAST_MM_ENGINE=valu): the specialised kernel keeps its register budget, armed  */ \
            /* workgroups go to the stepwise kernel on the side stream; every other side: mm_fast_kernel walks them itself  */ \
            const bool fast256 = n == 256 && g.rs == Mm256<R>::RS && g.bpm == Mm256<R>::BPM;                    \
            const bool sideGeneral = have && nFaultBlocks && fast256;                                           \
            if (have && nFaultBlocks && !fast256)                                                               \
                hookedBlocks = nFaultBlocks;                                                                    \
            if (sideGeneral) { /* faulted workgroups: stepwise kernel on the side stream, beside the fast one */ \
                HIP_TRY(c, hipEventRecord(c->evMainReady, c->stream));                                          \
                HIP_TRY(c, hipStreamWaitEvent(c->side, c->evMainReady, 0));                                     \
                hipLaunchKernelGGL(mm_general_kernel<R>, dim3(nFaultBlocks), block, lds, c->side, d_f, d_s,     \
                                   d_r, g, 0u, ctr, ft, dBlockList, d_detected);                                \
                HIP_TRY(c, hipEventRecord(c->evSideDone, c->side));                                             \
                generalBlocks = nFaultBlocks;                                                                   \
                fastBlocks = g.nblocks - nFaultBlocks;                                                          \
            }                                                                                                   \
            if (fast256)                                                                                        \
                hipLaunchKernelGGL(mm_fast256_kernel<R>, dim3(g.nblocks), block, Mm256<R>::LDS_BYTES, c->stream, \
                                   d_f, d_s, d_r, g, ctr, have ? ft.range : (const uint2 *)nullptr, d_detected); \
            else if ((n & 3) == 0)                                                                              \
                LAUNCH_FAST_K(R, true);                                                                         \
            else                                                                                                \
                LAUNCH_FAST_K(R, false);                                                                        \
            if (sideGeneral)                                                                                    \
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evSideDone, 0));                                    \
        }                                                                                                       \
    } while (0)
    if (cfg->replicas == 3)
        LAUNCH_MM(3);
    else if (cfg->replicas == 2)
        LAUNCH_MM(2);
    else
        LAUNCH_MM(1);
#undef LAUNCH_FAST
#undef LAUNCH_FAST_K
#undef LAUNCH_MM
    c->last.hooked_blocks = hookedBlocks;
    return after_launch(c, have, engine, generalBlocks, fastBlocks, 12.0 * (double)n * (double)n * (double)batch);
}

// ------------------------------------------------------------------------------------------------ default mode
extern "C" int coast_sync_copies_typed(coast_ctx *c, void *const *d_copies, int ncopies, size_t nbytes, void *d_voted, int scrub,
                                       uint8_t *d_detected, int elem, uint32_t vector_width)
{
    if (!c)
        return COAST_EINVAL;
    if (!d_copies || (ncopies != 2 && ncopies != 3) || (nbytes & 3u))
        return fail(c, COAST_EINVAL, "coast_sync_copies: 2 or 3 copies, byte count a multiple of 4");
    if ((elem != COAST_ELEM_U32 && elem != COAST_ELEM_F32) || vector_width < 1 || (nbytes / 4) % vector_width)
        return fail(c, COAST_EINVAL, "coast_sync_copies_typed: elem is COAST_ELEM_U32 or COAST_ELEM_F32, the word count a multiple "
                                     "of vector_width >= 1");
    if (nbytes == 0)
        return COAST_OK;
    for (int i = 0; i < ncopies; ++i)
        if (!d_copies[i] || ((uintptr_t)d_copies[i] & 15u))
            return fail(c, COAST_EINVAL, "coast_sync_copies: copies must be non-NULL and 16-byte aligned");
    if (d_voted && ((uintptr_t)d_voted & 15u))
        return fail(c, COAST_EINVAL, "coast_sync_copies: output must be 16-byte aligned");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t nwords = nbytes / 4;
    const uint64_t nvec = nwords / 4;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)c->numCUs * 8, std::max<uint64_t>(1, (nvec + 255) / 256));
    Counters ctr{c->dSlots, 0u};
    c->last = coast_launch_info{};
    {
        int prc = profile_begin(c);
        if (prc)
            return prc;
    }
    uint32_t *c0 = (uint32_t *)d_copies[0], *c1 = (uint32_t *)d_copies[1];
    uint32_t *c2 = ncopies == 3 ? (uint32_t *)d_copies[2] : nullptr;
    const bool fp = elem == COAST_ELEM_F32, vec = vector_width > 1;
#define LAUNCH_VOTE(NC, FP, VEC)                                                                                \
    hipLaunchKernelGGL((sync_copies_kernel<NC, FP, VEC>), dim3(grid), dim3(256), 0, c->stream, c0, c1, c2, nwords, \
                       (uint32_t *)d_voted, (NC) == 3 ? scrub : 0, ctr, d_detected)
    if (ncopies == 3) {
        if (fp && vec)
            LAUNCH_VOTE(3, true, true);
        else if (fp)
            LAUNCH_VOTE(3, true, false);
        else if (vec)
            LAUNCH_VOTE(3, false, true);
        else
            LAUNCH_VOTE(3, false, false);
    } else {
        if (fp)
            LAUNCH_VOTE(2, true, false);
        else
            LAUNCH_VOTE(2, false, false);
    }
#undef LAUNCH_VOTE
    return after_launch(c, 0, COAST_ENGINE_VOTE, 0, grid,
                        (double)nbytes * (ncopies + (d_voted ? 1 : 0) + (scrub && ncopies == 3 ? ncopies : 0)));
}

extern "C" int coast_sync_copies(coast_ctx *c, void *const *d_copies, int ncopies, size_t nbytes, void *d_voted,
                                 int scrub, uint8_t *d_detected)
{
    return coast_sync_copies_typed(c, d_copies, ncopies, nbytes, d_voted, scrub, d_detected, COAST_ELEM_U32, 1u);
}

extern "C" int coast_flip_memory(coast_ctx *c, void *d_ptr, size_t byte_offset, unsigned bit)
{
    if (!c || !d_ptr || bit > 7)
        return COAST_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(flip_memory_kernel, dim3(1), dim3(1), 0, c->stream, (uint8_t *)d_ptr + byte_offset, bit);
    HIP_TRY(c, hipGetLastError());
    return COAST_OK;
}

#include "launch_others.inc"
#include "cfcss_assign.inc"
#include "launch_cfcss.inc"
#include "host_shims.inc"
