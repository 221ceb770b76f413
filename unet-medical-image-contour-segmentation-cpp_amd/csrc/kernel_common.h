// kernel_common.h -- shared device helpers of the gfx950 kernels.  Internal to libmiunet.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include "kernels.h"

namespace miunet {

// cache policy of the hipcc fp32 kernels' OUTPUT stores (the aux operand of the raw buffer store builtins; gfx940+: bit 0 = sc0,
// bit 1 = nt, bit 4 = sc1).  Non-temporal stores pay in the assembly two-block kernel (gen_wino4_asm.py) and not here: same card,
// -DMIUNET_ST_AUX=2 against 0: inc.c2 1.105 -> 1.13 ms, the transposed convs +- 0.008 (profiles/r04_ab_store_policy.txt).
#ifndef MIUNET_ST_AUX
#define MIUNET_ST_AUX 0
#endif
constexpr int ST_AUX = MIUNET_ST_AUX;
// ... and of the 16-bit kernels' 16-byte output stores (-DMIUNET_LP_ST_AUX=2 for an A/B)
#ifndef MIUNET_LP_ST_AUX
#define MIUNET_LP_ST_AUX 0
#endif
constexpr int LP_ST_AUX = MIUNET_LP_ST_AUX;


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
// The table is read without a lock on the launch path (an eager launch used to take a process-wide mutex here: contexts of
// different host threads launch concurrently); the mutex only serialises the rare first opt-in of a (kernel, device, size).
template <typename K>
inline hipError_t ensure_dynamic_lds(K kernel, size_t bytes)
{
    struct Slot { std::atomic<const void *> fn; std::atomic<size_t> bytes[64]; };           // largest size opted in so far, per device
    static Slot slots[32];
    static std::mutex guard;
    const void *fn = reinterpret_cast<const void *>(kernel);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);                               // (a thread-local read inside the runtime)
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    for (Slot &c : slots) {
        const void *f = c.fn.load(std::memory_order_acquire);
        if (f == nullptr) break;
        if (f == fn) {
            if (c.bytes[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
            break;
        }
    }
    std::lock_guard<std::mutex> lk(guard);
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    for (Slot &c : slots) {
        const void *f = c.fn.load(std::memory_order_relaxed);
        if (f == fn || f == nullptr) {
            if (c.bytes[dev].load(std::memory_order_relaxed) < bytes) c.bytes[dev].store(bytes, std::memory_order_release);
            if (f == nullptr) c.fn.store(fn, std::memory_order_release);
            break;
        }
    }                                                               // table full: the attribute is simply set again next time
    return hipSuccess;
}

}  // namespace miunet
