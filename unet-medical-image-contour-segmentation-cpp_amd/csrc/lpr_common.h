// lpr_common.h -- shared by the resident-weight kernels of the 16-bit pipelines (conv_lpr.hip, convt_lpr.hip).  Internal.
#pragma once
#include "kernel_common.h"

namespace miunet {

template <typename T> struct LprVec { typedef T x8 __attribute__((ext_vector_type(8))); };

__device__ __forceinline__ f32x16 mfma_lpr(LprVec<__bf16>::x8 a, LprVec<__bf16>::x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_lpr(LprVec<_Float16>::x8 a, LprVec<_Float16>::x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void lpr_wait_vm()
{
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));      // vmcnt(N); expcnt / lgkmcnt untouched
}
__device__ __forceinline__ void lpr_wait_vm_n(int n)      // uniform n
{
    switch (n) {
    case 0: lpr_wait_vm<0>(); break;
    case 1: lpr_wait_vm<1>(); break;
    case 2: lpr_wait_vm<2>(); break;
    case 3: lpr_wait_vm<3>(); break;
    case 4: lpr_wait_vm<4>(); break;
    case 5: lpr_wait_vm<5>(); break;
    case 6: lpr_wait_vm<6>(); break;
    case 7: lpr_wait_vm<7>(); break;
    case 8: lpr_wait_vm<8>(); break;
    case 9: lpr_wait_vm<9>(); break;
    case 10: lpr_wait_vm<10>(); break;
    case 11: lpr_wait_vm<11>(); break;
    case 12: lpr_wait_vm<12>(); break;
    default: lpr_wait_vm<0>(); break;
    }
}

}  // namespace miunet
