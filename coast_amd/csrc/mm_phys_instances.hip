// mm_phys_instances.hip -- second translation unit of libcoast_hip.so: the physical-upset instantiations of the matrix-core kernels
// (mm_phys_instances.inc says why).  Templates only in these headers: nothing here is defined twice in the library.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "xmr.hpp"
#include "mm_kernel.hip"
#include "mm_mfma_kernel.hip"
#include "mm_mfma_blk2_kernel.hip"
#include "mm_mfma_blk3_kernel.hip"
#include "mm_mfma_blk4_kernel.hip"

namespace coast {
#define X(...) template __global__ void __VA_ARGS__(COAST_MMARGS);
#include "mm_phys_instances.inc"
#undef X
} // namespace coast
