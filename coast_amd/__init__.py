"""coast_amd -- MI355X-native redundant-execution engine for COAST's dataflowProtection hot path.

The product is libcoast_hip.so (hand-written gfx950 kernels behind the C ABI of include/coast_hip.h).  This package is
the thin host-side mirror of the reference's interface for that path: the protected kernels under their reference
names (matrix_multiply, sha256_hash, aes_enc_dec, crc16), the batch engine, the fault injector and the multi-GPU
counter reduction.  Nothing here computes on the CPU.
"""
from .engine import (DWC, F_ADDR_SYNC, F_BRANCH_SYNC, F_CLONE_STAGING, F_SINGLE_STAGING, F_LOCAL_STORE_SYNC, F_MEMORY_COPIES, F_NO_LOAD_SYNC, F_NO_STORE_ADDR_SYNC, F_O0_SHAPE,  # noqa: F401
                     F_NO_STORE_DATA_SYNC, TMR, UNPROTECTED, Engine, XmrConfig, make_faults)
from .hostapi import (FaultDetectedDWC, aes_enc_dec, crc16, host_stats, matrix_multiply,  # noqa: F401
                      sha256_hash)
from ._lib import FAULT_DTYPE, CoastLibraryError  # noqa: F401

REPLICA_ALL = 255  # COAST_REPLICA_ALL: a common-mode upset (state the replicas of a lane group share)
SITE_MM_ACC, SITE_MM_OPA, SITE_MM_OPB = 0, 1, 2
SITE_MM_VGPR = 6  # a physical register upset of the side-256 matrix-core kernel (step packs slab | lane << 8 | dword << 16 | register << 24)
SITE_MM_PREG = 7  # ... of ANY register of a wave, by physical number (step packs slot | step % 16 << 6 | lane << 10 | wave << 16 | file << 19 | register << 20 | step / 16 << 29)
SITE_MM_I, SITE_MM_J, SITE_MM_K = 3, 4, 5  # COAST_F_BRANCH_SYNC / COAST_F_ADDR_SYNC: the loop counters, one item per call
SITE_SHA_M, SITE_SHA_WV, SITE_SHA_STATE, SITE_SHA_DATALEN, SITE_SHA_I = 8, 9, 10, 11, 12
SITE_AES_STATE, SITE_AES_KEY, SITE_AES_ROUND, SITE_AES_I = 16, 17, 18, 19
SITE_CRC_CRC, SITE_CRC_X, SITE_CRC_LEN = 24, 25, 26
SITE_CT_SUM, SITE_CT_VAL, SITE_CT_NERR, SITE_CT_I = 32, 33, 34, 35
SITE_CHSHA_W, SITE_CHSHA_WV, SITE_CHSHA_DIGEST, SITE_CHSHA_I, SITE_CHSHA_COUNT = 40, 41, 42, 43, 44
SITE_QS_I, SITE_QS_J, SITE_QS_PIVOT, SITE_QS_VI, SITE_QS_VJ = 48, 49, 50, 51, 52
SITE_CFC_PC, SITE_CFC_RTS, SITE_CFC_RTSA = 56, 57, 58
SITE_CHAES_STATE, SITE_CHAES_WORD = 64, 65
SITE_CCF_I, SITE_CCF_TOTAL, SITE_CCF_TIMES, SITE_CCF_FI = 72, 73, 74, 75  # crazyCF under -TMR / -DWC (crazycf_xmr_batch)
SITE_CHAES_RND, SITE_CHAES_J, SITE_CHAES_I = 66, 67, 68  # CHStone aes under F_BRANCH_SYNC / F_ADDR_SYNC: the loop counters
CFC_OK, CFC_DETECTED, CFC_WATCHDOG, CFC_WILD = 0, 1, 2, 3
