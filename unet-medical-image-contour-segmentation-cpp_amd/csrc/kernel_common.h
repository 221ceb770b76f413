// kernel_common.h -- shared device helpers of the gfx950 kernels.  Internal to libmiunet.so.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "kernels.h"

namespace miunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// gfx950: a VMEM store of more than 64 bits reads its data registers AFTER it issues; a VALU write to one of them within two
// wait states corrupts the stored value.  hipcc pads that itself inside a basic block, but not when the overwriting
// instruction is the first of the block that follows a branch join (found in conv_lpr.hip: the pooled 16-byte store at
// the end of an `if (do_pool)`, then the next row block's v_add into the same register -- first dword of random lanes
// garbage, tests/test_gpu_bf16.py::test_conv3x3_resident_weights_fused_pooling).  Call this right after such a store.
__device__ __forceinline__ void wide_store_guard()
{
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}

// bijective XCD remap (cdna_hip_programming.md §5): blocks b and b+8 share an XCD; give XCD x the logical range
// [start_x, start_x + count_x).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// Kernels that need more than 64 KB of dynamic LDS must opt in once per (kernel, device).  Keyed by the kernel's address:
// template instantiations share one function-pointer TYPE, so a per-type flag would cover only the first of them.
template <typename K>
inline hipError_t ensure_dynamic_lds(K kernel, size_t bytes)
{
    struct Slot { const void *fn; size_t bytes[64]; };           // largest size opted in so far, per device
    static Slot slots[16] = {};
    static std::mutex guard;                                     // engines of different host threads launch concurrently
    std::lock_guard<std::mutex> lk(guard);
    const void *fn = reinterpret_cast<const void *>(kernel);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    Slot *sl = nullptr;
    for (Slot &c : slots)
        if (c.fn == fn || c.fn == nullptr) { sl = &c; break; }
    if (sl != nullptr && sl->fn == fn && sl->bytes[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (sl != nullptr) { sl->fn = fn; sl->bytes[dev] = bytes; }     // table full: just set the attribute every time
    return hipSuccess;
}

}  // namespace miunet
