// conv_lp.hip -- the direct implicit GEMM with 16-bit operands (bf16 / fp16) and fp32 accumulation, gfx950 only.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"
#include "lpr_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// conv_mfma_bf16 -- the same implicit GEMM with bf16 operands and fp32 accumulation (BASELINE config 3).
// Layout and schedule are those of conv_mfma_f32; what changes:
//   * a K-chunk is 32 channels = ONE v_mfma_f32_16x16x32 per tap (lpr_common.h: the shape the chip clocks highest under dense
//     16-bit MFMA work); LDS rows are 64 bytes, unpadded, the four 16-byte pieces of a row permuted for the 16-lane service
//     groups of ds_read_b128 (lds_swz_row16: a group reads rows {0-3, 12-15} with one piece index and {4-11} with the next);
//   * activations are 16-bit in HBM as well (every tensor but the network's last conv output, which feeds the fp32 head):
//     the producing kernel rounds its fp32 result once (round-to-nearest-even) instead of every consumer rounding it while
//     staging -- the same values (rounding commutes with max pooling and is idempotent across the skip connections), half the
//     HBM and L2 traffic, no conversion VALU in the loader: one 16-byte load and one 16-byte ds_write per 8 channels;
//   * epilogue: buffer stores on a per-image descriptor (one per-lane byte offset per output-row group, pixel displacement
//     in the scalar offset) -- 64-bit address arithmetic per store was most of the kernel's instructions at small K;
//   * lane (i16 = lane & 15, kq = lane >> 4) supplies A[i16][8 kq .. + 8] and B[8 kq .. + 8][i16]: one ds_read_b128 per
//     fragment, 8 fragment reads per 16 MFMAs of 16 cycles; the fragments of tap t + 1 are requested before the MFMAs of tap t
//     (the MFMAs are volatile asms: the software pipeline is written out) -- the kernel is bound by its staging and LDS traffic
//     and by HBM, not by the matrix pipe.
// The same kernel serves fp16 operands (BASELINE config 5's arithmetic): T = __bf16 or _Float16, 16 bits either way.
template <typename T> struct LpVec { typedef T x8 __attribute__((ext_vector_type(8))); };


// OUT_LP: the output tensor (and the pooled one) is 16-bit like the input; false = fp32 output (the layer in front of the head)
// HEAD: the layer feeds the network's fp32 1x1 head + argmax (ConvArgs::head_w): the post-ReLU fp32 tile crosses LDS instead
// of HBM (the staging images are dead by then), thread = pixel; `out` is never written.  Needs one n-tile (Cout <= BN).
template <typename T, int TAPS, int TH, int BN, bool NFAST, bool OUT_LP, bool HEAD = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_bf16(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                         const int m_tiles, const int nwg)
{
    typedef typename LpVec<T>::x8 bf16x8;
    static_assert(BN == 32 || BN % 64 == 0, "n-tile of 32 (narrow layers: base 32) or a multiple of 64 output channels");
    constexpr int ROW = KC_BF16;                         // 16-bit elements per LDS row (64 bytes, pieces permuted)
    constexpr int HEAD_ROW = BN + 4;                     // floats per pixel of the fused head's LDS tile
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * HALO, PH = TH + 2 * HALO, NPIX = PW * PH;
    constexpr int NA8 = NPIX * (KC_BF16 / 8);            // 8-channel pieces of the A patch
    constexpr int A_ITERS = (NA8 + 255) / 256;
    constexpr int MT = TH / 4;
    constexpr int MB16 = 2 * MT, NB16 = BN / 16;         // a wave's blocks of 16 pixels (row i = m >> 1, column half h = m & 1) x 16 channels
    constexpr int B_PARTS = BN >= 64 ? BN / 64 : 1;      // 64 rows x 64 bytes = 4 KB = 256 threads x 16 bytes (BN = 32: half of them)
    constexpr int B_ITERS = TAPS * B_PARTS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    T *const As = reinterpret_cast<T *>(lds);
    T *const Bs = As + NPIX * ROW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tiles = nwg / m_tiles;
    const int n_tile = NFAST ? L % n_tiles : L / m_tiles;
    int m = NFAST ? L / n_tiles : L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, n0 = n_tile * BN;
    const T *in_img = reinterpret_cast<const T *>(a.in) + (size_t)b * a.H * a.W * a.ldc;
    const T *wpk = reinterpret_cast<const T *>(a.wpk);

    int a_goff[A_ITERS], a_loff[A_ITERS];
#pragma unroll
    for (int s = 0; s < A_ITERS; ++s) {
        const int e = tid + 256 * s;
        const int pix = e >> 2, q = e & 3;               // 4 pieces of 8 channels per pixel
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 - HALO + py, gx = x0 - HALO + px;
        const bool live = e < NA8;
        const bool inb = live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        a_goff[s] = inb ? (gy * a.W + gx) * a.ldc + 8 * q : -1;
        a_loff[s] = live ? pix * ROW + 8 * (q ^ lds_swz_row16(px)) : -1;
    }
    const int bq = tid & 3, bn = tid >> 2;               // 16-byte piece (8 bf16) / cout row inside a 64-cout slab
    const bool b_live = BN >= 64 || bn < BN;             // BN = 32: rows 32..63 of the slab belong to nobody
    const T *w_base = wpk + ((size_t)n0 + bn) * KC_BF16 + 8 * bq;
    const int b_loff = bn * ROW + 8 * (bq ^ lds_swz_row16(bn));

    bf16x8 a_reg[A_ITERS];
    bf16x8 b_reg[B_ITERS];
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * KC_BF16;
#pragma unroll
        for (int s = 0; s < A_ITERS; ++s) {
            const int q8 = 8 * ((tid + 256 * s) & 3);
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (T)0.f;
            if (a_goff[s] >= 0 && c0 + q8 < a.Cin) v = *reinterpret_cast<const bf16x8 *>(in_img + a_goff[s] + c0);
            a_reg[s] = v;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            if (b_live) b_reg[it] = *reinterpret_cast<const bf16x8 *>(w_base + (((size_t)chunk * TAPS + tap) * a.CoutPad + part * 64) * KC_BF16);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < A_ITERS; ++s)
            if (a_loff[s] >= 0) *reinterpret_cast<bf16x8 *>(As + a_loff[s]) = a_reg[s];
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            if (b_live) *reinterpret_cast<bf16x8 *>(Bs + (tap * BN + part * 64) * ROW + b_loff) = b_reg[it];
        }
    };

    f32x4 acc[MB16][NB16];
#pragma unroll
    for (int m = 0; m < MB16; ++m)
#pragma unroll
        for (int j = 0; j < NB16; ++j) acc[m][j] = f32x4{ 0.f, 0.f, 0.f, 0.f };      // (a staging round and a barrier ahead of the first MFMA)

    // fragment addresses: patch pixel (row wave MT + i + dy, column 16 h + i16 + dx), piece kq; weight row tap BN + 16 j + i16
    int a_off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_off[dx] = (i16 + dx) * ROW + 8 * (kq ^ lds_swz_row16(i16 + dx));
    const T *const a_frag = As + (wave * MT) * PW * ROW;
    const T *const b_frag = Bs + i16 * ROW + 8 * (kq ^ lds_swz_row16(i16));
    auto read_tap = [&](const int tap, bf16x8 (&af)[MB16], bf16x8 (&bf)[NB16]) {
        const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
        for (int m = 0; m < MB16; ++m)
            af[m] = *reinterpret_cast<const bf16x8 *>(a_frag + (((m >> 1) + dy) * PW + 16 * (m & 1)) * ROW + a_off[dx]);
#pragma unroll
        for (int j = 0; j < NB16; ++j)
            bf[j] = *reinterpret_cast<const bf16x8 *>(b_frag + (tap * BN + 16 * j) * ROW);
    };
    const int nchunks = (a.Cin + KC_BF16 - 1) / KC_BF16;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) load_chunk(chunk + 1);
        bf16x8 af[2][MB16], bf[2][NB16];
        read_tap(0, af[0], bf[0]);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap + 1 < TAPS) read_tap(tap + 1, af[(tap + 1) & 1], bf[(tap + 1) & 1]);
#pragma unroll
            for (int m = 0; m < MB16; ++m)
#pragma unroll
                for (int j = 0; j < NB16; ++j) mfma16_lpr(acc[m][j], af[tap & 1][m], bf[tap & 1][j]);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }
    mfma16_drain();                               // (lpr_common.h: a barrier alone does not cover the last MFMA's latency)
#pragma unroll
    for (int m = 0; m < MB16; ++m)
#pragma unroll
        for (int j = 0; j < NB16; ++j) mfma16_settled(acc[m][j]);

    // ---- epilogue: + shift, ReLU, (16-bit rounding), buffer stores.  Lane = channel i16 of the 16-channel block j; register r
    // of pixel block (i, h) = pixel column 16 h + 4 kq + r of image row y0 + wave*MT + i.
    typedef typename std::conditional<OUT_LP, T, float>::type OutT;
    constexpr unsigned ES = sizeof(OutT);
    const int OH = (TAPS == 9) ? a.H : 2 * a.H, OW = (TAPS == 9) ? a.W : 2 * a.W;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<OutT *>(a.out) + (size_t)b * OH * OW * a.ldo, 0, (int)((size_t)OH * OW * a.ldo * ES), 0x00020000);
    const bool do_pool = TAPS == 9 && MT == 2 && a.pool_out != nullptr;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        do_pool ? reinterpret_cast<OutT *>(a.pool_out) + (size_t)b * Hp * Wp * a.pool_ld : reinterpret_cast<OutT *>(a.out), 0,
        do_pool ? (int)((size_t)Hp * Wp * a.pool_ld * ES) : 0, 0x00020000);
    const unsigned pix_bytes = (unsigned)a.ldo * ES, ppix_bytes = (unsigned)a.pool_ld * ES;
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const int yw = y0 + wave * MT;
    auto store_out = [&](const __amdgpu_buffer_rsrc_t &rs, float v, unsigned voff, unsigned soff) {
        if constexpr (OUT_LP) {
            const T t = (T)v;                     // round-to-nearest-even: the rounding the consumer used to do while staging
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, t), rs, voff, soff, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
        }
    };
    const bool interior = x0 + 32 <= a.W && y0 + TH <= a.H;
    // 16-bit conv3x3 outputs leave through LDS (the staging images are dead by now: the K loop ended with a barrier): every
    // lane drops its rounded values into a [pixel][BN + 8 pad] tile with 2-byte writes, then every thread stores 16-byte
    // pieces of 8 consecutive channels -- a wave's store is 8 pixels x 128 contiguous bytes instead of 2 bytes per lane.  Per
    // tile and wave that is 10 store instructions instead of 80.  Measured (same card, A/B): inc.c2 0.418 -> 0.402 ms, down1.c1
    // 0.190 -> 0.180, the other layers within 1 %: the store path is a small part of what holds these layers back.
    constexpr bool VIA_LDS = OUT_LP && TAPS == 9 && !HEAD;
    const bool via_lds = VIA_LDS && a.Cout % 8 == 0 && a.ldo % 8 == 0 && a.co_off % 8 == 0 && (!do_pool || a.pool_ld % 8 == 0);
    constexpr int TROW = BN + 8;                            // 16-bit elements per pixel row of the output tile
    T *const Ts = reinterpret_cast<T *>(lds);               // [TH * 32][TROW], then the pooled tile [TH * 8][TROW]
    T *const Ps = Ts + TH * 32 * TROW;
    // the shifts of this lane's channels, loaded BEFORE the first store (a load issued behind stores makes hipcc wait for
    // vmcnt(0): for every store in flight)
    float shj[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int n = n0 + 16 * j + i16;
        const bool n_ok = (TAPS == 9) ? (n < a.Cout) : (n < 4 * a.Cout);
        shj[j] = n_ok ? a.bias[(TAPS == 9) ? n : n % a.Cout] : 0.f;
    }
    auto epilogue = [&](auto lds_tag) {           // one straight-line copy per route: no per-store branches
    constexpr bool TO_LDS = decltype(lds_tag)::value;
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int n = n0 + 16 * j + i16;
        int co, oy_off = 0, ox_off = 0;
        if (TAPS == 9) {
            co = n;
        } else {
            const int kidx = n / a.Cout;
            co = n - kidx * a.Cout;
            oy_off = kidx >> 1; ox_off = kidx & 1;
        }
        const bool n_ok = (TAPS == 9) ? (n < a.Cout) : (n < 4 * a.Cout);
        const float sh = shj[j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int xc = x0 + 16 * h + 4 * kq;  // first of this lane's four columns
            // per-lane byte offset of (row yw, column xc) [conv] or of its 2x2 output block's (oy_off, ox_off) pixel [convT]
            const unsigned vbase = !n_ok ? 0xFFFFFFFFu
                : (TAPS == 9) ? (unsigned)(((yw * a.W + xc) * a.ldo + a.co_off + co) * ES)
                              : (unsigned)((((2 * yw + oy_off) * OW + 2 * xc + ox_off) * a.ldo + a.co_off + co) * ES);
            if (do_pool) {
                const unsigned pbase = n_ok ? (unsigned)((((yw >> 1) * Wp + (xc >> 1)) * a.pool_ld + co) * ES) : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x4 &u = acc[h][j], &v = acc[2 * (MT - 1) + h][j];
                    const float mx = fmaxf(fmaxf(fmaxf(u[r], u[r + 1]), fmaxf(v[r], v[r + 1])) + sh, relu_lo);
                    if constexpr (TO_LDS) { Ps[(wave * 16 + 8 * h + 2 * kq + (r >> 1)) * TROW + 16 * j + i16] = (T)mx; continue; }
                    const bool ok = interior || (yw + 1 < a.H && xc + r + 1 < a.W);
                    store_out(pool_rsrc, mx, ok ? pbase : 0xFFFFFFFFu, (unsigned)(r >> 1) * ppix_bytes);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaxf(acc[2 * i + h][j][r] + sh, relu_lo);
                    const int xt = 16 * h + 4 * kq + r;       // pixel column inside the 32-wide tile
                    if constexpr (HEAD) {                     // pixel (wave*MT + i, xt) of the TH x 32 tile, channel n
                        lds[((wave * MT + i) * 32 + xt) * HEAD_ROW + 16 * j + i16] = n_ok ? v : 0.f;
                        continue;
                    }
                    if constexpr (TO_LDS) { Ts[((wave * MT + i) * 32 + xt) * TROW + 16 * j + i16] = (T)v; continue; }
                    const bool ok = interior || (yw + i < a.H && xc + r < a.W);
                    const unsigned soff = (TAPS == 9) ? (unsigned)(i * a.W + r) * pix_bytes : (unsigned)(2 * i * OW + 2 * r) * pix_bytes;
                    store_out(out_rsrc, v, ok ? vbase : 0xFFFFFFFFu, soff);
                }
            }
        }
    }
    };
    if (VIA_LDS && via_lds) epilogue(std::integral_constant<bool, VIA_LDS>{});
    else epilogue(std::false_type{});
    if constexpr (VIA_LDS) {
        if (via_lds) {
            __syncthreads();
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr int PIECES = BN / 8;                  // 16-byte pieces per pixel
            for (int e = tid; e < TH * 32 * PIECES; e += 256) {
                const int px = e / PIECES, q = e - px * PIECES;
                const int py = px >> 5, pxx = px & 31;
                const bool ok = y0 + py < a.H && x0 + pxx < a.W && n0 + 8 * q < a.Cout;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + px * TROW + 8 * q);
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc,
                    ok ? (unsigned)((((y0 + py) * a.W + x0 + pxx) * a.ldo + a.co_off + n0 + 8 * q) * ES) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                wide_store_guard();
            }
            if (do_pool) {
                for (int e = tid; e < TH * 8 * PIECES; e += 256) {
                    const int px = e / PIECES, q = e - px * PIECES;
                    const int py = px >> 4, pxx = px & 15;          // pooled row (one per wave: MT = 2), pooled column
                    const bool ok = y0 + 2 * py + 1 < a.H && x0 + 2 * pxx + 1 < a.W && n0 + 8 * q < a.Cout;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(Ps + px * TROW + 8 * q);
                    __builtin_amdgcn_raw_buffer_store_b128(v, pool_rsrc,
                        ok ? (unsigned)(((((y0 >> 1) + py) * Wp + (x0 >> 1) + pxx) * a.pool_ld + n0 + 8 * q) * ES) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                    wide_store_guard();
                }
            }
        }
    }
    if constexpr (HEAD) {
        static_assert(TAPS == 9 && TH * 32 == 256 && (BN == 64 || BN == 32) && !OUT_LP, "one pixel per thread, every channel in the workgroup");
        float *const Wh = lds + 256 * HEAD_ROW;            // [classes][BN]
        if (tid < a.head_classes * BN) Wh[tid] = (tid % BN) < a.Cout ? a.head_w[(tid / BN) * a.Cout + (tid % BN)] : 0.f;
        __syncthreads();
        f32x4 d4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d4[k] = f32x4{ 0.f, 0.f, 0.f, 0.f };
        const float *yrow = lds + tid * HEAD_ROW;
#pragma unroll
        for (int c4 = 0; c4 < BN / 4; ++c4) {
            const f32x4 yv = *reinterpret_cast<const f32x4 *>(yrow + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.head_classes) d4[k] += yv * *reinterpret_cast<const f32x4 *>(Wh + BN * k + 4 * c4);
        }
        const int py = y0 + (tid >> 5), px = x0 + (tid & 31);
        if (py < a.H && px < a.W) {
            const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
            float best = -3.402823466e+38f;
            int idx = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < a.head_classes) {
                    const float d = ((d4[k].x + d4[k].y) + (d4[k].z + d4[k].w)) + a.head_b[k];
                    if (a.head_logits != nullptr) a.head_logits[((size_t)b * a.head_classes + k) * hw + pin] = d;
                    if (d > best) { best = d; idx = k; }          // first maximum wins (src/process.cpp:158-170)
                }
            }
            a.head_labels[(size_t)b * hw + pin] = (uint8_t)idx;
        }
    }
}

template <typename T, int TAPS, int TH, int BN, bool NFAST, bool OUT_LP, bool HEAD = false>
static hipError_t launch_bf16_cfg(const ConvArgs &a, hipStream_t s)
{
    const int n_total = (TAPS == 9) ? a.Cout : 4 * a.Cout;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (n_total + BN - 1) / BN;
    const int nwg = m_tiles * n_tiles;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr size_t lds_stage = 2 * (size_t)KC_BF16 * ((32 + 2 * HALO) * (TH + 2 * HALO) + TAPS * BN);
    constexpr size_t lds_head = HEAD ? sizeof(float) * (256 * (size_t)(BN + 4) + 4 * BN) : 0;     // [256 px][BN + 4] + [4 classes][BN]
    constexpr size_t lds = lds_stage > lds_head ? lds_stage : lds_head;
    auto kern = conv_mfma_bf16<T, TAPS, TH, BN, NFAST, OUT_LP, HEAD>;
    if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

// Cout <= 32 (the top level of a base-32 network, BASELINE config 5) takes a 32-wide n-tile: the 64-wide one would stage,
// read and multiply a half-empty weight slab.
template <typename T>
static hipError_t launch_conv3x3_lp(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 8 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    const bool narrow = a.Cout <= 32;
    if (a.head_w != nullptr) {
        if (a.out_lp || a.Cout > 64 || a.head_classes < 1 || a.head_classes > 4 || a.pool_out != nullptr || a.head_labels == nullptr)
            return hipErrorInvalidValue;
        return narrow ? launch_bf16_cfg<T, 9, 8, 32, false, false, true>(a, s) : launch_bf16_cfg<T, 9, 8, 64, false, false, true>(a, s);
    }
    if (narrow) return a.out_lp ? launch_bf16_cfg<T, 9, 8, 32, false, true>(a, s) : launch_bf16_cfg<T, 9, 8, 32, false, false>(a, s);
    return a.out_lp ? launch_bf16_cfg<T, 9, 8, 64, false, true>(a, s) : launch_bf16_cfg<T, 9, 8, 64, false, false>(a, s);
}

hipError_t launch_conv3x3_bf16(const ConvArgs &a, hipStream_t s) { return launch_conv3x3_lp<__bf16>(a, s); }

hipError_t launch_convT2x2_bf16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 8 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return a.out_lp ? launch_bf16_cfg<__bf16, 1, 8, 64, true, true>(a, s) : launch_bf16_cfg<__bf16, 1, 8, 64, true, false>(a, s);
}

hipError_t launch_conv3x3_fp16(const ConvArgs &a, hipStream_t s) { return launch_conv3x3_lp<_Float16>(a, s); }

hipError_t launch_convT2x2_fp16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 8 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return a.out_lp ? launch_bf16_cfg<_Float16, 1, 8, 64, true, true>(a, s) : launch_bf16_cfg<_Float16, 1, 8, 64, true, false>(a, s);
}


}  // namespace miunet
