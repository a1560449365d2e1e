// cfcss.hpp -- run-time half of CFCSS (projects/CFCSS/CFCSS.cpp:494-572) for gfx950 kernels.
//
// The pass keeps two 16-bit globals, BasicBlockSignatureTracker (RTS) and RunTimeSignatureAdjuster (RTSA, :723-734): stored at the
// end of a block (insertStoreInsts, :494-506), XORed with the next block's signature difference and compared with its signature at
// the top of that block (insertCompInsts, :508-549); a mismatch branches to the function's error block -> FAULT_DETECTED_CFC() ->
// abort() (:107-126, splitBlocks :708-731).
//
// Here every lane runs its own work item, so the two globals are one VGPR pair per wave (a 16-bit value per lane): leaving a block
// is two v_mov from the signature table, entering one is two XORs and ONE v_cmp for the 64 control-flow paths the wave holds --
// the wave branches to the handler on vcc != 0.  The tables (coast_cfc_tables, built by coast_cfcss_assign) are staged in LDS:
// lanes of a wave sit in different blocks, the lookups are per-lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace coast {

constexpr int kCfcMaxNodes = 64; // blocks of one protected kernel (crazyCF: 28 + buffer blocks)
constexpr int kCfcMaxSucc = 128;
constexpr int kCfcMaxCalls = 4;
enum { kCfcFanIn = 1, kCfcChecked = 2, kCfcBuffer = 4, kCfcSkip = 8, kCfcRet = 16 };

struct CfcDevTables { // device copy of the coast_cfc_tables of one program, compact
    uint16_t sig[kCfcMaxNodes], diff[kCfcMaxNodes], adj[kCfcMaxNodes];
    uint8_t flags[kCfcMaxNodes];
    uint8_t succBegin[kCfcMaxNodes + 4];
    uint8_t succ[kCfcMaxSucc];
    uint16_t callPreAdj[kCfcMaxCalls], callPostAdj[kCfcMaxCalls];
    uint32_t nNodes, pad;
};
static_assert(sizeof(CfcDevTables) % 4 == 0, "staged with dword copies");

struct CfcTracker {
    uint32_t rts = 0u, rtsa = 0u; // both start at zero (setUpGlobal, :483-492)
    // end of block b: insertStoreInsts
    __device__ __forceinline__ void leave(const CfcDevTables &t, uint32_t b)
    {
        rts = t.sig[b];
        rtsa = t.adj[b];
    }
    // top of block b: insertCompInsts; true = signature mismatch
    __device__ __forceinline__ bool enter_bad(const CfcDevTables &t, uint32_t b) const
    {
        const uint32_t fl = t.flags[b];
        uint32_t gsig = rts ^ t.diff[b];
        if (fl & kCfcFanIn)
            gsig ^= rtsa;
        return (fl & kCfcChecked) != 0u && (gsig & 0xffffu) != t.sig[b];
    }
};

} // namespace coast
