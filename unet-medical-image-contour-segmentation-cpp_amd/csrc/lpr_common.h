// lpr_common.h -- shared by the 16-bit kernels whose input patches arrive by LDS-DMA (conv_lpr.hip, conv_lprk.hip, convt_lpr.hip,
// conv_lp2.hip).  Internal.
#pragma once
#include "kernel_common.h"

namespace miunet {

template <typename T> struct LprVec { typedef T x8 __attribute__((ext_vector_type(8))); };

// ---- v_mfma_f32_16x16x32_{bf16,f16}: the shape of the 16-bit kernels
// Under dense 16-bit MFMA work the chip holds its clock down (1.7-1.9 GHz on these layers against 2.4 nominal), so cycles
// saved in the issue stream return only partly as wall time, while the MFMA SHAPE is a lever of its own: at equal cycles per
// FLOP the 16x16x32 form holds a higher clock than 32x32x16 (MI355X_MICROARCH.md, 'DVFS give-back' item 7; same card, round 3:
// the wide kernel's layers 2.5-10 % faster, profiles/r03_ab_mfma_shape.txt).  Lane (i16 = lane & 15, kq = lane >> 4) supplies
// A[i16][8 kq .. + 8] and B[8 kq .. + 8][i16] and holds D[4 kq + r][i16], r = 0..3.
// Accumulation is in place through inline asm: the builtin's 4-register destination is not tied to its accumulator operand
// (only the wider MFMAs get the tied form), and hipcc then rotates accumulators through other registers -- copies at every
// loop head, spills inside the K loops.  Hazards: the callers never touch an accumulator within four instructions of the MFMA
// that wrote it except as the accumulator of the next MFMA on the same registers (interlocked), and reach their epilogue
// through a wait or a barrier.
__device__ __forceinline__ void mfma16_lpr(f32x4 &c, LprVec<__bf16>::x8 a, LprVec<__bf16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_lpr(f32x4 &c, LprVec<_Float16>::x8 a, LprVec<_Float16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// The first MFMA of a chain takes the literal 0 as its accumulator: no v_mov of zeros in front of it (a VALU write needs two
// wait states before an MFMA reads it, and hipcc pads nothing in front of an asm).
__device__ __forceinline__ void mfma16_lpr_first(f32x4 &c, LprVec<__bf16>::x8 a, LprVec<__bf16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_lpr_first(f32x4 &c, LprVec<_Float16>::x8 a, LprVec<_Float16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// ---- v_mfma_f32_32x32x2_f32 in place, for a kernel that threads a second, fp32 GEMM between its 16-bit MFMAs (conv_lpr.hip, FIRST):
// volatile asms keep their program order, a builtin would be clustered away from them.  Lane (i = lane & 31, h = lane >> 5)
// supplies A[i][h] and B[h][i] and holds D[(r & 3) + 8 (r >> 2) + 4 h][i] in register r.  A chain starts from the literal 0;
// before the first VALU read of the result: mfma32_settle (a 16-pass MFMA needs 18 wait states there; the asm hides it from hipcc).
__device__ __forceinline__ void mfma32_f32(f32x16 &c, float a, float b) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma32_f32_first(f32x16 &c, float a, float b) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma32_settle(f32x16 &c) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(c)); }
// The compiler does not see an MFMA behind the asm, so it pads nothing between the last of them and the first VALU read of
// an accumulator (a 4-pass MFMA needs 7 wait states there): a kernel whose epilogue follows its MFMAs without a barrier calls
// mfma16_drain() once and mfma16_settled(acc) on every accumulator -- volatile asms keep their order, and every later read
// of `acc` depends on the (empty) second one.
__device__ __forceinline__ void mfma16_drain() { asm volatile("s_nop 7\n\ts_nop 7"); }
__device__ __forceinline__ void mfma16_settled(f32x4 &c) { asm volatile("" : "+v"(c)); }

// ---- where the four 16-byte pieces of a pixel sit inside its 64 bytes of an LDS image
// An LDS-DMA load places lane l's 16 bytes at base + 16 l, so rows cannot be padded; bank spreading comes from the slot a
// piece takes inside its pixel instead: piece q of the pixel in column `col` of a patch row lives in slot q ^ swz(col), and
// the permutation costs nothing -- it is the per-lane GLOBAL offset of the load.  What "spread" has to mean is set by how
// gfx950 services a ds_read_b128: in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32,
// over 64 banks of 4 bytes (MI355X_MICROARCH.md, LDS) -- a group takes one LDS cycle when its 16 pieces fall into 16 distinct
// 16-byte bank groups (byte address / 16 mod 16), and one more for every extra address on a busy group.  A 16x16x32 fragment
// is one patch row x 16 columns with the piece index in the lane (lane = (column i16, piece kq)): a group holds columns
// {0-3, 12-15} with one piece index and {4-11} with the next, and swz = 2 ((col >> 2) & 1) separates them for every tap
// displacement (tools/dev/lds_bank_model.py enumerates every fragment read of every kernel; tests/test_tools_cpu.py runs it).
// The round-2 layout, (pixel >> 1) & 3 under 32x32x16 fragments, was derived for groups of eight consecutive lanes and is 2-way
// conflicted in EVERY group under the real grouping: 8 LDS cycles per read instead of 4.  Round 3 measured both on the same
// card (profiles/r03_ab_lds_swizzle.txt): SQ_LDS_BANK_CONFLICT fell from 0.45-0.47 to 0.04-0.11 of SQ_LDS_IDX_ACTIVE and no
// layer's time moved by more than 1 % -- these kernels are bound by the clock the chip holds, not by LDS cycles.
__device__ __forceinline__ int lds_swz_row16(int col) { return 2 * ((col >> 2) & 1); }

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
