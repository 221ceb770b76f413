// conv_lp2.hip -- the 16-bit-operand conv3x3 for the WIDE layers (Cout a multiple of 128): a workgroup of four waves owns 16 rows x 32
// columns x 128 output channels, 256 accumulator registers per wave.  bf16 or fp16 operands, fp32 accumulation on
// v_mfma_f32_16x16x32_{bf16,f16}; gfx950 only.  Two kernels share that tile: conv3x3_lp2 (first below: the ROWS split over the waves,
// 4 rows x 32 columns x 128 channels per wave) and conv3x3_lp2n (further down: the CHANNELS split over the waves, 16 rows x 32 columns x
// 32 channels per wave) -- the wide layers run on the second since round 3; the first serves fp32 outputs and unaligned channel
// offsets, and carries the timing-only builds that led to the second.
//
// Why a second kernel.  conv_mfma_bf16 (conv_lp.hip) gives a wave 64 pixels x 64 channels: 8 fragment reads from LDS per 16
// MFMAs of 16 cycles, i.e. 128 B/clk per CU of the LDS's 256 for the fragments alone, with the ds_writes of the register
// staging beside them -- its matrix pipe is a third busy on the deep layers.  Here a wave owns 4 image rows x 32 columns x 128
// output channels (8 x 8 blocks of 16 x 16: 256 accumulator registers, one wave per SIMD): 8 patch + 8 weight fragments per
// 64 MFMAs, and the weights do not go through LDS at all -- all four waves of a workgroup need the same weight fragments, but
// they are small and hot (one kernel's worth per 32 input channels: 72 KB), so every lane loads its 16-byte fragment straight
// from L1/L2 with a buffer load whose (chunk, tap) displacement is a scalar offset -- the U ring of the fp32 Winograd kernels
// -- SIX groups (192 MFMAs = 3 072 cycles) ahead.  LDS carries only the input patch: 8 reads per 64 MFMAs and wave.
//
// What the K loop waits for (timing-only builds, same card, profiles/r03_ab_lp2_k_loop.txt): without the patch DMA inside the loop
// config 3's step is 5 % shorter, without the weight refills 8 %, without both 10 % -- a wave's vmcnt counts its loads IN ORDER, so a wait
// for a weight fragment is also a wait for every patch DMA issued before it (HBM latency against a ring lead of three groups = 1 536
// cycles in round 2).  A ring of six groups: config 3 +2.8 %, config 5 +1.6 % (nine: no better); it fits since the accumulators are
// pinned in AGPRs (96 of the 256 VGPRs).  The rest of that 10 % needs the operand streams off the MFMA waves' own counters (loader
// waves, or weights staged through LDS for the whole workgroup): next round.
//
// The MFMA shape.  Under dense 16-bit MFMA work the chip holds its clock down (1.7-1.9 GHz on these layers against 2.4
// nominal: `clock_ghz_from_sq_busy` of the bench), so cycles saved by a tighter issue stream come back only partly as wall time
// -- round 3 halved this kernel's LDS cycles by removing a 2-way bank conflict from every fragment read (counters:
// SQ_LDS_BANK_CONFLICT 0.47 -> 0.08 of SQ_LDS_IDX_ACTIVE) and no layer moved by more than 1 % -- while the MFMA SHAPE is a lever
// of its own: at equal cycles per FLOP the 16x16x32 form holds a higher clock than 32x32x16 (MI355X_MICROARCH.md, 'DVFS
// give-back' item 7).  Same card, same tile, round 3 (profiles/r03_ab_mfma_shape.txt): every layer of this kernel 2.5-10 %
// faster on 16x16x32, BASELINE config 3 2812 -> 3000 images/s, config 5 1840 -> 1898.  All 16-bit kernels are on that shape now.
//
// Workgroup = 4 waves = 16 rows x 32 columns of pixels x 128 output channels.  K is walked in chunks of 32 input channels:
//   LDS: the 18 x 34 input patch of the chunk, 64 bytes per pixel, double-buffered (2 x 39,936 B); the next chunk's patch
//        arrives by LDS-DMA (buffer_load ... lds, 16 bytes per lane: no staging registers, no ds_write), one load per group;
//        rows cannot be padded that way, so the four 16-byte pieces of a pixel are permuted inside its 64 bytes for the 16-lane
//        service groups of ds_read_b128 (lpr_common.h: lds_swz_row16).  ONE barrier per chunk = per 576 MFMAs of a wave.
//   a wave's 4 rows x 32 columns are 8 pixel blocks m = (row m >> 1, column half m & 1), its 128 channels 8 channel blocks;
//   one MFMA contracts a tap's whole 32-channel chunk: lane (i16 = lane & 15, kq = lane >> 4) supplies the pixel's / the output
//        channel's input channels 8 kq .. + 8 -- piece kq of the pixel's 64 bytes, one ds_read_b128 / buffer load;
//   a group is (tap, channel half): 4 weight fragments x the tap's 8 patch fragments = 32 MFMAs; the patch fragments are
//        refilled one by one right after their last use in the tap's second group (28 MFMAs ahead of their next use); the weight
//        ring is six groups deep.
// Arithmetic: the same products and the same fp32 accumulation chain as conv_mfma_bf16 (chunks in order, taps in raster order,
// one MFMA per tap and chunk): bit-identical results; same epilogue semantics: + folded-BN shift, ReLU, one
// round-to-nearest-even to the 16-bit output, optional fused 2x2 max pooling (each wave holds rows 4w .. 4w+3: both row pairs
// in-lane).
#include <cstdlib>
#include <type_traits>

// this file's 16-byte output stores are non-temporal (kernel_common.h: LP_ST_AUX).  Same-card A/B of every 16-bit kernel
// (profiles/r04_ab_store_policy.txt): the wide layers of config 3 3.05 -> 2.99 ms with it -- their consumers of the two smallest
// tensors, which had found them in L2, + 0.016 --; neutral or worse in the resident-weight kernels, the transposed convs and the
// first layer, which keep the default policy.
#ifndef MIUNET_LP_ST_AUX
#define MIUNET_LP_ST_AUX 2
#endif
#include "kernel_common.h"
#include "lpr_common.h"

namespace miunet {

template <typename T> struct Lp2Vec { typedef T x8 __attribute__((ext_vector_type(8))); };

struct LP2 {
    static constexpr int TH = 16, MT = 4;
    static constexpr int ROW = KC_BF16;                      // 16-bit elements per pixel (64 bytes, unpadded: pieces are swizzled)
    static constexpr int PW = 34, PH = TH + 2, NPIX = PW * PH;
    static constexpr int A_LOADS = (NPIX + 15) / 16;         // wave-wide LDS-DMA loads of 16 pixels
    static constexpr int A_ELEMS = A_LOADS * 16 * ROW;       // per patch buffer (39 KB)
    static constexpr size_t LDS_BYTES = 2 * 2 * (size_t)A_ELEMS;      // two patch buffers of 16-bit elements
};

// In-place accumulation through inline asm (lpr_common.h): "+a" pins each block to its own AGPR quad.  Hazards: no accumulator
// is touched again for 32 MFMAs inside the loop, and a wait plus a barrier separate the last MFMA from the epilogue's first
// accumulator read.
__device__ __forceinline__ void mfma_lp2s(f32x4 &c, Lp2Vec<__bf16>::x8 a, Lp2Vec<__bf16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_lp2s(f32x4 &c, Lp2Vec<_Float16>::x8 a, Lp2Vec<_Float16>::x8 b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// EXP != 0: timing-only experiment builds (MIUNET_LP2_EXP; results are WRONG, never routed by default; DESIGN 5.1 "what the K loop
// of the wide kernel waits for"): 1 = no patch DMA inside the K loop (every chunk multiplies chunk 0's patch), 2 = the weight ring
// is never refilled, 3 = both
template <typename T, bool OUT_LP, int EXP = 0, int WD = 6>
__global__ __launch_bounds__(256, 1) void conv3x3_lp2(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                       const int m_tiles, const int nwg)
{
    typedef typename Lp2Vec<T>::x8 x8;
    constexpr int ROW = LP2::ROW, PW = LP2::PW, MT = LP2::MT, BN = 128, TH = LP2::TH;
    constexpr int MB = 2 * MT, NB = BN / 16;                 // 8 pixel blocks x 8 channel blocks per wave; WD = weight ring depth in groups
    static_assert(18 % WD == 0, "the ring depth must divide the groups of a chunk");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    T *const As = reinterpret_cast<T *>(lds);                // [2][NPIX][ROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, n0 = n_tile * BN;
    const T *in_img = reinterpret_cast<const T *>(a.in) + (size_t)b * a.H * a.W * a.ldc;

    // ---- patch loads by LDS-DMA, as in conv3x3_lp2 (the two low bits of an offset carry the piece index)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int DMA_ITERS = (LP2::A_LOADS + 3) / 4;
    unsigned dvoff[DMA_ITERS];
#pragma unroll
    for (int k = 0; k < DMA_ITERS; ++k) {
        const int i = wave + 4 * k;
        const int p = 16 * i + (lane >> 2);
        const int py = p / PW, px = p - py * PW;
        const int q = (lane & 3) ^ lds_swz_row16(px);
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inb = i < LP2::A_LOADS && p < LP2::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        dvoff[k] = inb ? ((unsigned)(((gy * a.W + gx) * a.ldc + 8 * q) * 2) | (unsigned)q) : 0xFFFFFFFFu;
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in_img), 0, a.H * a.W * a.ldc * 2, 0x00020000);
    auto dma_a = [&](int chunk, int buf, int k) {
        if (wave + 4 * k < LP2::A_LOADS) {
            const unsigned dv = dvoff[k];
            const unsigned voff = (chunk * KC_BF16 + 8 * (int)(dv & 3u) < a.Cin) ? (dv & ~15u) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(As + buf * LP2::A_ELEMS + (wave + 4 * k) * 16 * ROW), 16, voff, chunk * KC_BF16 * 2, 0, 0);
        }
    };

    // ---- weight fragments straight from global memory: packed [chunk][tap][CoutPad][32]; lane (i16, kq) of channel block j
    // wants input channels 8 kq .. + 8 of output channel n0 + 16 j + i16
    const int nchunks = (a.Cin + KC_BF16 - 1) / KC_BF16;
    const unsigned tap_bytes = (unsigned)a.CoutPad * KC_BF16 * 2;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk), 0, (int)((size_t)nchunks * 9 * tap_bytes), 0x00020000);
    const unsigned w_voff = (unsigned)(((n0 + i16) * KC_BF16 + 8 * kq) * 2);
    auto w_load = [&](int chunk, int grp, int jj) {          // grp = 2 tap + channel half; jj = block inside the half
        return __builtin_bit_cast(x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff + (unsigned)((4 * (grp & 1) + jj) * 16 * KC_BF16 * 2),
                                                                             (unsigned)(chunk * 9 + (grp >> 1)) * tap_bytes, 0));
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    unsigned aoff[6][3];                          // byte offset of piece kq of pixel (4 wave + r) * 34 + i16 + dx (column half 1: + 1024)
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int p = (wave * MT + r) * PW + i16 + dx;
            aoff[r][dx] = (unsigned)(p * 64 + ((kq ^ lds_swz_row16(i16 + dx)) << 4));
        }
#pragma unroll
    for (int k = 0; k < DMA_ITERS; ++k) dma_a(0, 0, k);
    x8 wf[WD][4];
#pragma unroll
    for (int k = 0; k < WD; ++k)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wf[k][jj] = w_load(0, k, jj);
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((WD * 4) & 15) | (((WD * 4) >> 4) << 14));        // the patch is older than the ring
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int abuf = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        const int nxt = more ? chunk + 1 : chunk;                          // the last chunk prefetches itself: straight-line code
        const unsigned abase = (unsigned)(abuf * LP2::A_ELEMS * 2);
        auto read_a = [&](int tap, int mb) {      // patch fragment of block mb = (row mb >> 1, column half mb & 1) for a tap
            const int dy = tap / 3, dx = tap - 3 * dy;
            return *reinterpret_cast<const x8 *>(reinterpret_cast<const char *>(As) + (abase + aoff[(mb >> 1) + dy][dx]) + (mb & 1) * 1024);
        };
        x8 af[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[mb] = read_a(0, mb);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            const int tap = g >> 1, jh = g & 1;
            // the next patch, one load per group from the chunk's first group on (two per group in its first five groups, or one per
            // group in its last ten, measured the same or 1 % slower: profiles/r03_ab_lp2_k_loop.txt)
            if (more && g < DMA_ITERS && !(EXP & 1)) dma_a(chunk + 1, abuf ^ 1, g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) mfma_lp2s(acc[mb][4 * jh + jj], af[mb], wf[g % WD][jj]);
                if (jh == 1 && tap + 1 < 9) {     // last use of this fragment in the tap: refill it for the next one
                    af[mb] = read_a(tap + 1, mb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int gn = g + WD;                // refill the ring slot WD groups ahead (into the next chunk at the end)
            if constexpr (!(EXP & 2)) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) wf[g % WD][jj] = w_load(gn < 18 ? chunk : nxt, gn % 18, jj);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((WD * 4) & 15) | (((WD * 4) >> 4) << 14));    // this wave's patch loads have landed
        __syncthreads();
    }

    // ---- epilogue: + shift, ReLU, (16-bit rounding), stores.  Lane = channel i16 of block j; register r of block (row, h) =
    // pixel column 16 h + 4 kq + r of image row y0 + 4 wave + row.
    typedef typename std::conditional<OUT_LP, T, float>::type OutT;
    constexpr unsigned ES = sizeof(OutT);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<OutT *>(a.out) + (size_t)b * a.H * a.W * a.ldo, 0, (int)((size_t)a.H * a.W * a.ldo * ES), 0x00020000);
    const bool do_pool = a.pool_out != nullptr;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        do_pool ? reinterpret_cast<OutT *>(a.pool_out) + (size_t)b * Hp * Wp * a.pool_ld : reinterpret_cast<OutT *>(a.out), 0,
        do_pool ? (int)((size_t)Hp * Wp * a.pool_ld * ES) : 0, 0x00020000);
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const int yw = y0 + wave * MT;
    const bool interior = x0 + 32 <= a.W && y0 + TH <= a.H;
    float shj[NB];                                // loaded BEFORE the first store (conv3x3_lp2)
#pragma unroll
    for (int j = 0; j < NB; ++j) shj[j] = n0 + 16 * j + i16 < a.Cout ? a.bias[n0 + 16 * j + i16] : 0.f;
    if constexpr (OUT_LP) {
        if (a.Cout % 8 == 0 && a.ldo % 8 == 0 && a.co_off % 8 == 0 && (!do_pool || a.pool_ld % 8 == 0)) {
            // 16-bit outputs leave through a wave-private LDS tile of 32 pixels x 32 channels as 16-byte pieces (conv3x3_lp2)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr int TROW = 40;
            T *const Ts = As + wave * (48 * TROW);       // [32 pixels][TROW] + pooled [16][TROW], wave-private
            T *const Ps = Ts + 32 * TROW;
            auto lds_epilogue = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int jp = 0; jp < NB / 2; ++jp) {         // 32 channels = blocks 2 jp, 2 jp + 1
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                Ts[(16 * h + 4 * kq + r) * TROW + 16 * jl + i16] = (T)fmaxf(acc[2 * i + h][2 * jp + jl][r] + shj[2 * jp + jl], relu_lo);
                    if (do_pool && (i & 1)) {     // rows i - 1 and i, column pairs (r, r + 1): pooled column 8 h + 2 kq + (r >> 1)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                                for (int r = 0; r < 4; r += 2) {
                                    const f32x4 &u = acc[2 * (i - 1) + h][2 * jp + jl], &v = acc[2 * i + h][2 * jp + jl];
                                    const float mx = fmaxf(fmaxf(u[r], u[r + 1]), fmaxf(v[r], v[r + 1]));
                                    Ps[(8 * h + 2 * kq + (r >> 1)) * TROW + 16 * jl + i16] = (T)fmaxf(mx + shj[2 * jp + jl], relu_lo);
                                }
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int e = lane + 64 * it, m = e >> 2, q = e & 3;
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + m * TROW + 8 * q);
                        const bool ok = INTERIOR || (yw + i < a.H && x0 + m < a.W);
                        __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc,
                            ok ? (unsigned)((((yw + i) * a.W + x0 + m) * a.ldo + a.co_off + n0 + 32 * jp + 8 * q) * 2) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                        wide_store_guard();
                    }
                    if (do_pool && (i & 1)) {
                        const int m = lane >> 2, q = lane & 3;              // 16 pooled pixels x 4 pieces = 64 lanes
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(Ps + m * TROW + 8 * q);
                        const bool ok = INTERIOR || (yw + i < a.H && x0 + 2 * m + 1 < a.W);
                        __builtin_amdgcn_raw_buffer_store_b128(v, pool_rsrc,
                            ok ? (unsigned)(((((yw + i) >> 1) * Wp + (x0 >> 1) + m) * a.pool_ld + n0 + 32 * jp + 8 * q) * 2) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                        wide_store_guard();
                    }
                }
            }
            };
            if (interior) lds_epilogue(std::true_type{});
            else lds_epilogue(std::false_type{});
            return;
        }
    }
    // fp32 outputs (the layer in front of an unfused head) and unaligned channel offsets: element stores
    auto store_out = [&](const __amdgpu_buffer_rsrc_t &rs, float v, unsigned voff, unsigned soff) {
        if constexpr (OUT_LP) {
            const T t = (T)v;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, t), rs, voff, soff, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
        }
    };
    const unsigned pix_bytes = (unsigned)a.ldo * ES, ppix_bytes = (unsigned)a.pool_ld * ES;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int co = n0 + 16 * j + i16;
        const bool n_ok = co < a.Cout;
        const float sh = shj[j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xc = x0 + 16 * h + 4 * kq;  // first of this lane's four columns
            const unsigned vbase = n_ok ? (unsigned)(((yw * a.W + xc) * a.ldo + a.co_off + co) * ES) : 0xFFFFFFFFu;
            if (do_pool) {
                const unsigned pbase = n_ok ? (unsigned)((((yw >> 1) * Wp + (xc >> 1)) * a.pool_ld + co) * ES) : 0xFFFFFFFFu;
#pragma unroll
                for (int ip = 0; ip < MT / 2; ++ip)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x4 &u = acc[4 * ip + h][j], &v = acc[4 * ip + 2 + h][j];
                        const float mx = fmaxf(fmaxf(fmaxf(u[r], u[r + 1]), fmaxf(v[r], v[r + 1])) + sh, relu_lo);
                        const bool ok = interior || (yw + 2 * ip + 1 < a.H && xc + r + 1 < a.W);
                        store_out(pool_rsrc, mx, ok ? pbase : 0xFFFFFFFFu, (unsigned)(ip * Wp + (r >> 1)) * ppix_bytes);
                    }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(acc[2 * i + h][j][r] + sh, relu_lo);
                    const bool ok = interior || (yw + i < a.H && xc + r < a.W);
                    store_out(out_rsrc, v, ok ? vbase : 0xFFFFFFFFu, (unsigned)(i * a.W + r) * pix_bytes);
                }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------
// conv3x3_lp2n -- the same workgroup tile with the CHANNELS, not the rows, split over the four waves: the kernel the wide layers
// run on (16-bit outputs with 16-byte aligned channel offsets; anything else, and MIUNET_LP2_NSPLIT=0, takes conv3x3_lp2).  Same
// card, round 3 (profiles/r03_ab_lp2_k_loop.txt block 4): every wide layer 2-10 % faster.  Why: the timing-only builds of conv3x3_lp2 say
// its largest in-loop wait is for the weight fragments, which each of its four waves fetches in full (8 per tap).  Here wave w
// owns channels 32 w .. + 32 of all 16 rows x 32 columns (32 pixel blocks x 2 channel blocks: the same 256 accumulator registers):
// 2 weight fragments per tap and wave -- a quarter of the L1 / L2 traffic -- and a ring NINE taps deep in 72 registers, i.e. one
// whole chunk (9 216 cycles) of lead, beyond any patch DMA's latency in the wave's in-order vmcnt.  The price is on the LDS side:
// every wave reads every patch fragment (32 per tap and wave instead of 8; 128 B/clk per CU of the LDS's 256, conflict-free).
template <typename T, int AD = 8>
__global__ __launch_bounds__(256, 1) void conv3x3_lp2n(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                        const int m_tiles, const int nwg)
{
    typedef typename Lp2Vec<T>::x8 x8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int ROW = LP2::ROW, PW = LP2::PW, BN = 128, TH = LP2::TH, TROW = 40;
    constexpr int MB = 2 * TH, WT = 9;                       // 32 pixel blocks (row m >> 1, column half m & 1); weight ring in taps; AD = patch-fragment ring
    static_assert(288 % AD == 0, "the fragment ring must divide a chunk's 288 fragments");       // (4 and 16 measured the same as 8)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    T *const As = reinterpret_cast<T *>(lds);                // [2][NPIX][ROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, c0 = n_tile * BN + 32 * wave;       // this wave's first output channel
    const T *in_img = reinterpret_cast<const T *>(a.in) + (size_t)b * a.H * a.W * a.ldc;

    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int DMA_ITERS = (LP2::A_LOADS + 3) / 4;
    static_assert(DMA_ITERS == 10, "one patch load per tap and one more in the first");
    unsigned dvoff[DMA_ITERS];
#pragma unroll
    for (int k = 0; k < DMA_ITERS; ++k) {
        const int i = wave + 4 * k;
        const int p = 16 * i + (lane >> 2);
        const int py = p / PW, px = p - py * PW;
        const int q = (lane & 3) ^ lds_swz_row16(px);
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inb = i < LP2::A_LOADS && p < LP2::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        dvoff[k] = inb ? ((unsigned)(((gy * a.W + gx) * a.ldc + 8 * q) * 2) | (unsigned)q) : 0xFFFFFFFFu;
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in_img), 0, a.H * a.W * a.ldc * 2, 0x00020000);
    auto dma_a = [&](int chunk, int buf, int k) {
        if (wave + 4 * k < LP2::A_LOADS) {
            const unsigned dv = dvoff[k];
            const unsigned voff = (chunk * KC_BF16 + 8 * (int)(dv & 3u) < a.Cin) ? (dv & ~15u) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(As + buf * LP2::A_ELEMS + (wave + 4 * k) * 16 * ROW), 16, voff, chunk * KC_BF16 * 2, 0, 0);
        }
    };

    const int nchunks = (a.Cin + KC_BF16 - 1) / KC_BF16;
    const unsigned tap_bytes = (unsigned)a.CoutPad * KC_BF16 * 2;
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk), 0, (int)((size_t)nchunks * 9 * tap_bytes), 0x00020000);
    const unsigned w_voff = (unsigned)(((c0 + i16) * KC_BF16 + 8 * kq) * 2);
    auto w_load = [&](int chunk, int tap, int jl) {
        return __builtin_bit_cast(x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff + (unsigned)(jl * 16 * KC_BF16 * 2), (unsigned)(chunk * 9 + tap) * tap_bytes, 0));
    };

    f32x4 acc[MB][2];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) acc[i][jl] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    unsigned aoff[3];                             // piece kq of patch pixel (row 0, column i16 + dx); patch row and column half are immediates
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) aoff[dx] = (unsigned)((i16 + dx) * 64 + ((kq ^ lds_swz_row16(i16 + dx)) << 4));
#pragma unroll
    for (int k = 0; k < DMA_ITERS; ++k) dma_a(0, 0, k);
    x8 wf[WT][2];
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) wf[t][jl] = w_load(0, t, jl);
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((WT * 2) & 15) | (((WT * 2) >> 4) << 14));      // the patch is older than the ring
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int abuf = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        const int nxt = more ? chunk + 1 : chunk;                          // the last chunk prefetches itself: straight-line code
        const unsigned abase = (unsigned)(abuf * LP2::A_ELEMS * 2);
        auto read_a = [&](int q) {                // fragment q = 32 tap + m of the chunk
            const int tap = q >> 5, mb = q & 31, dy = tap / 3, dx = tap - 3 * dy;
            return *reinterpret_cast<const x8 *>(reinterpret_cast<const char *>(As) + (abase + aoff[dx]) + (((mb >> 1) + dy) * PW * 64 + (mb & 1) * 1024));
        };
        x8 af[AD];
#pragma unroll
        for (int q = 0; q < AD; ++q) af[q] = read_a(q);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (more) {                           // the next patch: one load per tap (two in the first)
                dma_a(chunk + 1, abuf ^ 1, tap);
                if (tap == 0) dma_a(chunk + 1, abuf ^ 1, 9);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int q = 32 * tap + mb;
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) mfma_lp2s(acc[mb][jl], af[q % AD], wf[tap][jl]);
                if (q + AD < 288) af[q % AD] = read_a(q + AD);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) wf[tap][jl] = w_load(nxt, tap, jl);             // this tap of the next chunk: nine taps ahead
            __builtin_amdgcn_sched_barrier(0);
        }
        // this wave's patch loads are older than the two youngest weight loads: landed
        __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
        __syncthreads();
    }

    // ---- epilogue: + shift, ReLU, 16-bit rounding; a row's 32 pixels x this wave's 32 channels through the wave's LDS tile, 16-byte stores
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<T *>(a.out) + (size_t)b * a.H * a.W * a.ldo, 0, (int)((size_t)a.H * a.W * a.ldo * 2), 0x00020000);
    const bool do_pool = a.pool_out != nullptr;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        do_pool ? reinterpret_cast<T *>(a.pool_out) + (size_t)b * Hp * Wp * a.pool_ld : reinterpret_cast<T *>(a.out), 0,
        do_pool ? (int)((size_t)Hp * Wp * a.pool_ld * 2) : 0, 0x00020000);
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const bool interior = x0 + 32 <= a.W && y0 + TH <= a.H;
    float shj[2];
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) shj[jl] = c0 + 16 * jl + i16 < a.Cout ? a.bias[c0 + 16 * jl + i16] : 0.f;
    T *const Ts = As + wave * (48 * TROW);       // [32 pixels][TROW] + pooled [16][TROW], wave-private
    T *const Ps = Ts + 32 * TROW;
    auto lds_epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Ts[(16 * h + 4 * kq + r) * TROW + 16 * jl + i16] = (T)fmaxf(acc[2 * i + h][jl][r] + shj[jl], relu_lo);
            if (do_pool && (i & 1)) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x4 &u = acc[2 * (i - 1) + h][jl], &v = acc[2 * i + h][jl];
                            const float mx = fmaxf(fmaxf(u[r], u[r + 1]), fmaxf(v[r], v[r + 1]));
                            Ps[(8 * h + 2 * kq + (r >> 1)) * TROW + 16 * jl + i16] = (T)fmaxf(mx + shj[jl], relu_lo);
                        }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = lane + 64 * it, mm = e >> 2, q = e & 3;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + mm * TROW + 8 * q);
                const bool ok = (INTERIOR || (y0 + i < a.H && x0 + mm < a.W)) && c0 + 8 * q < a.Cout;
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc,
                    ok ? (unsigned)((((y0 + i) * a.W + x0 + mm) * a.ldo + a.co_off + c0 + 8 * q) * 2) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                wide_store_guard();
            }
            if (do_pool && (i & 1)) {
                const int mm = lane >> 2, q = lane & 3;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(Ps + mm * TROW + 8 * q);
                const bool ok = (INTERIOR || (y0 + i < a.H && x0 + 2 * mm + 1 < a.W)) && c0 + 8 * q < a.Cout;
                __builtin_amdgcn_raw_buffer_store_b128(v, pool_rsrc,
                    ok ? (unsigned)(((((y0 + i) >> 1) * Wp + (x0 >> 1) + mm) * a.pool_ld + c0 + 8 * q) * 2) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                wide_store_guard();
            }
        }
    };
    if (interior) lds_epilogue(std::true_type{});
    else lds_epilogue(std::false_type{});
}

template <typename T, bool OUT_LP>
static hipError_t launch_lp2_cfg(const ConvArgs &a, hipStream_t s)
{
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + LP2::TH - 1) / LP2::TH;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int nwg = m_tiles * ((a.Cout + 127) / 128);
    auto launch = [&](auto kern) {
        if (hipError_t e = ensure_dynamic_lds(kern, LP2::LDS_BYTES); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LP2::LDS_BYTES, s, a, tiles_x, tiles_y, m_tiles, nwg);
        return hipGetLastError();
    };
#ifdef MIUNET_EXPERIMENTS                              // lab build only (libmiunet_exp.so, tools/dev/ab*.sh): never in libmiunet.so
    static const int exp = [] { const char *e = getenv("MIUNET_LP2_EXP"); return e ? atoi(e) : 0; }();      // timing-only builds (see the kernel)
    static const int wd = [] { const char *e = getenv("MIUNET_LP2_WD"); return e ? atoi(e) : 6; }();           // A/B: =3, the ring depth of round 2
    static const int nsplit = [] { const char *e = getenv("MIUNET_LP2_NSPLIT"); return e ? atoi(e) : 1; }();     // A/B: =0, rows split over the waves (conv3x3_lp2)
#else
    constexpr int exp = 0, nsplit = 1;
#endif
    if constexpr (OUT_LP) {
        if (nsplit && exp == 0 && a.Cout % 8 == 0 && a.ldo % 8 == 0 && a.co_off % 8 == 0 && (a.pool_out == nullptr || a.pool_ld % 8 == 0)) {
            return launch(conv3x3_lp2n<T>);
        }
    }
#ifdef MIUNET_EXPERIMENTS
    if (exp == 1) return launch(conv3x3_lp2<T, OUT_LP, 1>);
    if (exp == 2) return launch(conv3x3_lp2<T, OUT_LP, 2>);
    if (exp == 3) return launch(conv3x3_lp2<T, OUT_LP, 3>);
    if (wd == 3) return launch(conv3x3_lp2<T, OUT_LP, 0, 3>);
#endif
    hipError_t e = launch(conv3x3_lp2<T, OUT_LP, 0>);
    return e;
}

// Which layers it takes (measured per layer at batch 16, r02): faster than the 2 x 2 kernel of conv_lp.hip from Cin = 256 up
// (down4.c2 0.280 -> 0.226 ms, up1.c1 0.552 -> 0.459), 3-5 % faster at Cin = 128 with the LDS-transposed stores (level with the
// 256-store epilogue it had first), slower below.  MIUNET_LP2: 0 = never; 2 = every Cout % 128 == 0 layer whatever its size (parity tests).
bool conv3x3_lp2_takes(const ConvArgs &a)
{
    const int mode = routing_of(a).lp2;
    if (mode == 0) return false;
    if (a.head_w != nullptr || a.Cout % 128 != 0 || a.Cin % 8 || a.ldc % 8 || a.CoutPad % NPAD) return false;
    if (mode == 2) return true;
    const long long nwg = (long long)((a.W + 31) / 32) * ((a.H + LP2::TH - 1) / LP2::TH) * a.B * (a.Cout / 128);
    // from Cin = 128 since the 16-byte-store epilogue (same card, config 3: down1.c2 0.333 -> 0.322 ms, up3.c2 0.313 -> 0.298, down2.c1
    // 0.159 -> 0.154; config 5 unchanged); MIUNET_LP2_MINCIN moves the threshold
#ifdef MIUNET_EXPERIMENTS
    static const int min_cin = [] { const char *m = getenv("MIUNET_LP2_MINCIN"); return m ? atoi(m) : 128; }();
#else
    constexpr int min_cin = 128;
#endif
    return a.Cin >= min_cin && nwg >= 192;
}

hipError_t launch_conv3x3_lp2(const ConvArgs &a, bool fp16, hipStream_t s)
{
    if (a.Cout % 128 != 0 || a.head_w != nullptr || a.ldc % 8) return hipErrorInvalidValue;
    if (fp16) return a.out_lp ? launch_lp2_cfg<_Float16, true>(a, s) : launch_lp2_cfg<_Float16, false>(a, s);
    return a.out_lp ? launch_lp2_cfg<__bf16, true>(a, s) : launch_lp2_cfg<__bf16, false>(a, s);
}

}  // namespace miunet
