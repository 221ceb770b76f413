// kernel_common.h -- shared device helpers of the gfx950 kernels.  Internal to libmiunet.so.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace miunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// bijective XCD remap (cdna_hip_programming.md §5): blocks b and b+8 share an XCD; give XCD x the logical range
// [start_x, start_x + count_x).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// Kernels that need more than 64 KB of dynamic LDS must opt in once per (kernel, device).
template <typename K>
inline hipError_t ensure_dynamic_lds(K kernel, size_t bytes)
{
    static bool done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    return hipSuccess;
}

}  // namespace miunet
