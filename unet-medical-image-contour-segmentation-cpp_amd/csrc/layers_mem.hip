// layers_mem.hip -- the HBM-bound layers: first conv (K = 9), 2x2 max pooling, 1x1 head + argmax.  gfx950 only.
#include <cstdlib>

#include "kernel_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// First layer (K = 9*Cin with Cin <= 4: HBM-bound on its output, 1 GiB at batch 16).  A workgroup owns PPB consecutive
// pixels of one image row (64 at Cout = 64, 128 at Cout = 32); thread = (pixel slot, 4 output channels) and walks FOUR
// consecutive pixels.  The three input rows of the workgroup -- (PPB + 2) x Cin bytes each -- are loaded ONCE, converted
// u8 -> fp32 through the host-built 256-entry table (so the input equals float(x)/255.0f bit for bit, src/process.cpp:36-39)
// and kept in LDS; every thread then reads its 3 x 6 x Cin floats from there with 16-byte reads.  (Round 1 had every thread
// load and look up its own bytes: the Cout/4 threads of a pixel slot repeated the same 18 x Cin byte loads, which at
// Cin = 3, Cout = 32 -- BASELINE config 5 -- made the layer 1.05 ms at 0.5 TB/s, the slowest kernel of that network.)
// The Cout/4 threads of a pixel write one contiguous NHWC row (256 bytes at Cout = 64).
// OT = float, or __bf16 / _Float16 for the 16-bit pipelines (the tensor is then rounded here, once, instead of by its consumer)
template <int CIN, typename OT>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const uint8_t *__restrict__ img, const float *__restrict__ lut,
                                                            const float *__restrict__ w, const float *__restrict__ shift,
                                                            OT *__restrict__ out, int H, int W, int Cout, int ldo,
                                                            int quads, int xblocks)
{
    __shared__ float s_lut[256];
    extern __shared__ __attribute__((aligned(16))) float s_in[];      // planar: [CIN][3 rows][rowlen], rowlen = ppb + 2 (+ pad)
    s_lut[threadIdx.x] = lut[threadIdx.x];
    const int q = threadIdx.x % quads;            // which 4 couts
    const int pl = threadIdx.x / quads;           // pixel slot in block
    const int ppb = 4 * (256 / quads);            // pixels per block (4 per slot)
    const int xb = blockIdx.x % xblocks;
    const int row = blockIdx.x / xblocks;         // b * H + y
    const int y = row % H;
    const int xb0 = xb * ppb;                     // first pixel of the block
    const int rowlen = (ppb + 2 + 3) & ~3;        // floats per staged row, a 16-byte multiple
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 4 * q);
    f32x4 w0[9];                                  // Cin = 1: the nine weight quads are issued here and land under the staging
    if constexpr (CIN == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t) w0[t] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * Cout + 4 * q);
    }
    __syncthreads();                              // the table is complete
    const uint8_t *rowp = img + (size_t)row * W * CIN;
    const int rowbytes = (ppb + 2) * CIN;
#pragma unroll
    for (int r = 0; r < 3; ++r) {                 // no runtime divisions: CIN is a compile-time constant
        const int yy = y + r - 1;
        const bool yok = yy >= 0 && yy < H;
        for (int k = threadIdx.x; k < rowbytes; k += 256) {
            const int slot = k / CIN, c = k - slot * CIN, xx = xb0 - 1 + slot;
            const bool ok = yok && xx >= 0 && xx < W;
            s_in[(c * 3 + r) * rowlen + slot] = ok ? s_lut[rowp[((long long)(r - 1) * W + xb0 - 1) * CIN + k]] : 0.f;
        }
    }
    __syncthreads();
    const int x0 = xb0 + 4 * pl;
    if (x0 >= W) return;
    // One input channel at a time: its nine weight quads (36 registers) and its 3 x 6 pixel window; the four pixels'
    // accumulators stay live across the channels.  (Holding all 9 x Cin weight quads at once -- 108 registers at Cin = 3 --
    // left two waves per SIMD; this order leaves five.)  Sum order: channel-major, taps in raster order inside a channel.
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    auto channel = [&](int c) {
        f32x4 wc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if constexpr (CIN == 1) wc[t] = w0[t];
            else wc[t] = *reinterpret_cast<const f32x4 *>(w + (size_t)(t * CIN + c) * Cout + 4 * q);
        }
        float v[3][6];                            // rows y-1..y+1, columns x0-1..x0+4 of channel c
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float *src = s_in + (c * 3 + r) * rowlen + 4 * pl;      // 16-byte aligned
#pragma unroll
            for (int k = 0; k < 6; ++k) v[r][k] = src[k];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[i] += v[t / 3][i + t % 3] * wc[t];
    };
    if constexpr (CIN == 1) {
        channel(0);
    } else {
#pragma unroll 1                                  // a real loop: unrolled, hipcc hoists all 9 x Cin weight loads again
        for (int c = 0; c < CIN; ++c) channel(c);
    }
    OT *orow = out + ((size_t)row * W + x0) * ldo + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x0 + i >= W) break;
        const f32x4 a4 = acc[i] + sh;
        f32x4 r;
        r.x = a4.x > 0.f ? a4.x : 0.f; r.y = a4.y > 0.f ? a4.y : 0.f;
        r.z = a4.z > 0.f ? a4.z : 0.f; r.w = a4.w > 0.f ? a4.w : 0.f;
        if constexpr (sizeof(OT) == 4) {
            *reinterpret_cast<f32x4 *>(orow + (size_t)i * ldo) = r;
        } else {
            typedef OT ot4 __attribute__((ext_vector_type(4)));
            ot4 t;
            t[0] = (OT)r.x; t[1] = (OT)r.y; t[2] = (OT)r.z; t[3] = (OT)r.w;
            *reinterpret_cast<ot4 *>(orow + (size_t)i * ldo) = t;
        }
    }
}

template <typename OT>
static hipError_t launch_first_t(const uint8_t *img, const float *lut256, const float *w, const float *shift, OT *out,
                                 int B, int H, int W, int Cin, int Cout, int ldo, hipStream_t s)
{
    const int quads = Cout / 4;
    if (Cout % 4 || quads > 256 || 256 % quads || ldo % 4) return hipErrorInvalidValue;
    const int ppb = 4 * (256 / quads);
    const int xblocks = (W + ppb - 1) / ppb;
    const long long blocks = (long long)B * H * xblocks;
    if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * 3 * (size_t)Cin * (((size_t)ppb + 2 + 3) & ~(size_t)3);
    switch (Cin) {
    case 1: hipLaunchKernelGGL((conv3x3_first_kernel<1, OT>), dim3((unsigned)blocks), dim3(256), lds, s, img, lut256, w, shift, out, H, W, Cout, ldo, quads, xblocks); break;
    case 3: hipLaunchKernelGGL((conv3x3_first_kernel<3, OT>), dim3((unsigned)blocks), dim3(256), lds, s, img, lut256, w, shift, out, H, W, Cout, ldo, quads, xblocks); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// The same layer on the fp32 MFMA, for the 16-bit pipelines.  The kernel above is VALU-bound at Cout = 32 x K = 27 (BASELINE
// config 5: 0.35 ms for a 0.54 GB write, 41 TFLOP/s) and at Cout = 64 x K = 9 (config 3: 0.18 ms).  As a GEMM it is tiny:
// M = pixels, N = Cout, K = 9 Cin <= 27 -- K / 2 steps of v_mfma_f32_32x32x2_f32 per 32 pixels x 32 channels, operands and
// accumulation still fp32 (only the ORDER of the K sum differs from the VALU kernel: tap-major instead of channel-major).
//   * workgroup = 4 RBW rows x 32 columns, wave w = rows RBW w .. + RBW (row blocks of 1 x 32 pixels) x NBK channel blocks;
//   * the (4 RBW + 2) x 34-pixel patch goes through the /255 table into LDS as [pixel][Cin] floats (+ one zero word for the
//     padded K step); lane (pixel i, k half h) reads A[i][2s + h] with one ds_read_b32 at patch offset + koff[s];
//   * B[2s + h][n] = w[2s + h][n] sits in K / 2 x NBK registers per lane for the whole tile;
//   * epilogue: + shift, ReLU, one rounding, [pixel][channel] tile in the wave's LDS scratch, 16-byte stores.
template <int CIN, int NBK, int RBW, typename OT>
__global__ __launch_bounds__(256) void conv3x3_first_mfma(const uint8_t *__restrict__ img, const float *__restrict__ lut,
                                                          const float *__restrict__ w, const float *__restrict__ shift,
                                                          OT *__restrict__ out, int H, int W, int Cout, int ldo,
                                                          int tiles_x, int tiles_y)
{
    static_assert(sizeof(OT) == 2, "16-bit outputs (the fp32 plan keeps the VALU kernel: its tensor is HBM-bound there)");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int K = 9 * CIN, KS = (K + 1) / 2;
    constexpr int TH = 4 * RBW, PW = 34, NPIX = (TH + 2) * PW;
    constexpr int TROW = 40;                                  // 16-bit elements per pixel of the output scratch (32 + 8 pad)
    __shared__ float s_lut[256];
    __shared__ __attribute__((aligned(16))) float s_in[NPIX * CIN + 4];       // [pixel][CIN], then zeros
    __shared__ __attribute__((aligned(16))) OT s_out[4][32 * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    s_lut[tid] = lut[tid];
    int L = blockIdx.x;
    const int tx = L % tiles_x; L /= tiles_x;
    const int ty = L % tiles_y;
    const int b = L / tiles_y;
    const int y0 = ty * TH, x0 = tx * 32;

    // B fragments and the per-step patch offsets of this lane's k = 2 s + lh
    float bw[KS][NBK];
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lh;
        const int tap = k / CIN, c = k - tap * CIN, dy = tap / 3, dx = tap - 3 * dy;
        koff[s] = k < K ? ((dy * PW + dx) * CIN + c) : -1;
#pragma unroll
        for (int j = 0; j < NBK; ++j) bw[s][j] = (k < K && 32 * j + li < Cout) ? w[(size_t)k * Cout + 32 * j + li] : 0.f;
    }
    // the patch bytes are requested BEFORE the barrier that completes the table (one global round trip per workgroup, not two)
    const uint8_t *imgb = img + (size_t)b * H * W * CIN;
    constexpr int SITERS = (NPIX * CIN + 255) / 256;
    int sbyte[SITERS];                            // the byte, or -1 for zero padding / past the patch
#pragma unroll
    for (int it = 0; it < SITERS; ++it) {
        const int e = tid + 256 * it;
        const int p = e / CIN, c = e - p * CIN, py = p / PW, px = p - py * PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        sbyte[it] = (e < NPIX * CIN && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (int)imgb[((size_t)gy * W + gx) * CIN + c] : -1;
    }
    __syncthreads();                              // the table is complete
#pragma unroll
    for (int it = 0; it < SITERS; ++it) {
        const int e = tid + 256 * it;
        if (e < NPIX * CIN) s_in[e] = sbyte[it] >= 0 ? s_lut[sbyte[it]] : 0.f;
    }
    if (tid < 4) s_in[NPIX * CIN + tid] = 0.f;
    __syncthreads();

    f32x16 acc[RBW][NBK];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < NBK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int pbase = ((RBW * wave) * PW + li) * CIN;           // row block i adds i * PW * CIN
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float av[RBW];
#pragma unroll
        for (int i = 0; i < RBW; ++i) av[i] = s_in[koff[s] >= 0 ? pbase + i * PW * CIN + koff[s] : NPIX * CIN];
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int j = 0; j < NBK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[s][j], acc[i][j], 0, 0, 0);
    }

    // epilogue: accumulator register r = pixel column (r & 3) + 8 (r >> 2) + 4 lh of row y0 + 4 wave + i, lane = channel
    OT *const Ts = s_out[wave];
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * H * W * ldo, 0, H * W * ldo * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int y = y0 + RBW * wave + i;
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            const float sh = 32 * j + li < Cout ? shift[32 * j + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] + sh;
                Ts[((r & 3) + 8 * (r >> 2) + 4 * lh) * TROW + li] = (OT)(v > 0.f ? v : 0.f);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = lane + 64 * it, m = e >> 2, q = e & 3;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + m * TROW + 8 * q);
                const bool ok = y < H && x0 + m < W && 32 * j + 8 * q < Cout;
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, ok ? (unsigned)((((y * W) + x0 + m) * ldo + 32 * j + 8 * q) * 2) : 0xFFFFFFFFu, 0, LP_ST_AUX);
                wide_store_guard();
            }
        }
    }
}

template <typename OT>
static hipError_t launch_first_mfma_t(const uint8_t *img, const float *lut256, const float *w, const float *shift, OT *out,
                                      int B, int H, int W, int Cin, int Cout, int ldo, hipStream_t s)
{
    // row blocks per wave: 2 (8-row tiles: 68-112 registers, five to seven waves per SIMD) measured 0.242 -> 0.223 ms on config
    // 5's first layer against 4 (16-row tiles, 116-192 registers), same card; MIUNET_FIRST_RBW=4 keeps the taller tile
#ifdef MIUNET_EXPERIMENTS                              // lab build only: the product library has one route per shape
    static const int rbw = [] { const char *e = getenv("MIUNET_FIRST_RBW"); return (e && atoi(e) == 4) ? 4 : 2; }();
#else
    constexpr int rbw = 2;
#endif
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 4 * rbw - 1) / (4 * rbw);
    const long long blocks = (long long)B * tiles_x * tiles_y;
    if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    const dim3 g((unsigned)blocks), t(256);
    if (Cin == 1 && Cout <= 32) { if (rbw == 2) hipLaunchKernelGGL((conv3x3_first_mfma<1, 1, 2, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); else hipLaunchKernelGGL((conv3x3_first_mfma<1, 1, 4, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); }
    else if (Cin == 1) { if (rbw == 2) hipLaunchKernelGGL((conv3x3_first_mfma<1, 2, 2, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); else hipLaunchKernelGGL((conv3x3_first_mfma<1, 2, 4, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); }
    else if (Cout <= 32) { if (rbw == 2) hipLaunchKernelGGL((conv3x3_first_mfma<3, 1, 2, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); else hipLaunchKernelGGL((conv3x3_first_mfma<3, 1, 4, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); }
    else { if (rbw == 2) hipLaunchKernelGGL((conv3x3_first_mfma<3, 2, 2, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); else hipLaunchKernelGGL((conv3x3_first_mfma<3, 2, 4, OT>), g, t, 0, s, img, lut256, w, shift, out, H, W, Cout, ldo, tiles_x, tiles_y); }
    return hipGetLastError();
}

// the MFMA form takes the 16-bit outputs it was built for (MIUNET_FIRST_MFMA=0: never)
static bool first_mfma_takes(int Cin, int Cout, int ldo, int H, int W, int out_kind, const Routing *rt)
{
    if (!(rt && rt->resolved ? *rt : Routing::from_env()).first_mfma) return false;
    return out_kind != 0 && (Cin == 1 || Cin == 3) && Cout % 8 == 0 && Cout <= 64 && ldo % 8 == 0 && (long long)H * W * ldo * 2 < (1ll << 31);
}

// out_kind: 0 = fp32, 1 = bf16, 2 = fp16 output tensor
hipError_t launch_conv3x3_first(const uint8_t *img, const float *lut256, const float *w, const float *shift, float *out,
                                int B, int H, int W, int Cin, int Cout, int ldo, int out_kind, hipStream_t s, const Routing *rt)
{
    if (first_mfma_takes(Cin, Cout, ldo, H, W, out_kind, rt))
        return out_kind == 1 ? launch_first_mfma_t(img, lut256, w, shift, reinterpret_cast<__bf16 *>(out), B, H, W, Cin, Cout, ldo, s)
                             : launch_first_mfma_t(img, lut256, w, shift, reinterpret_cast<_Float16 *>(out), B, H, W, Cin, Cout, ldo, s);
    if (out_kind == 1) return launch_first_t(img, lut256, w, shift, reinterpret_cast<__bf16 *>(out), B, H, W, Cin, Cout, ldo, s);
    if (out_kind == 2) return launch_first_t(img, lut256, w, shift, reinterpret_cast<_Float16 *>(out), B, H, W, Cin, Cout, ldo, s);
    return launch_first_t(img, lut256, w, shift, out, B, H, W, Cin, Cout, ldo, s);
}

// --------------------------------------------------------------------------------------------------------------------
// 2x2 max pooling, stride 2; input may be the lower half of a concat buffer (channel stride ldc).  16 B per lane.
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float *__restrict__ in, int ldc, float *__restrict__ out,
                                                         int Ho, int Wo, int C4, long long total)
{
    const int W = 2 * Wo;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        const long long p = e / C4;                 // output pixel index over [B][Ho][Wo]
        const int xo = (int)(p % Wo);
        const long long by = p / Wo;                // b*Ho + yo
        const float *src = in + ((size_t)(2 * by) * W + 2 * xo) * ldc + 4 * c4;
        const f32x4 v00 = *reinterpret_cast<const f32x4 *>(src);
        const f32x4 v01 = *reinterpret_cast<const f32x4 *>(src + ldc);
        const f32x4 v10 = *reinterpret_cast<const f32x4 *>(src + (size_t)W * ldc);
        const f32x4 v11 = *reinterpret_cast<const f32x4 *>(src + (size_t)W * ldc + ldc);
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = v00[k];
            t = v01[k] > t ? v01[k] : t;
            t = v10[k] > t ? v10[k] : t;
            t = v11[k] > t ? v11[k] : t;
            m[k] = t;
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)p * (4 * C4) + 4 * c4) = m;
    }
}

// The 16-bit pipelines pool post-ReLU tensors: non-negative bf16 / fp16 values order like their bit patterns, so the
// maximum is an integer maximum on 16-bit lanes (8 channels = 16 bytes per thread).
__global__ __launch_bounds__(256) void maxpool2x2_u16_kernel(const uint16_t *__restrict__ in, int ldc, uint16_t *__restrict__ out,
                                                             int Ho, int Wo, int C8, long long total)
{
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const int W = 2 * Wo;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        const long long p = e / C8;
        const int xo = (int)(p % Wo);
        const long long by = p / Wo;
        const uint16_t *src = in + ((size_t)(2 * by) * W + 2 * xo) * ldc + 8 * c8;
        const u16x8 v00 = *reinterpret_cast<const u16x8 *>(src), v01 = *reinterpret_cast<const u16x8 *>(src + ldc);
        const u16x8 v10 = *reinterpret_cast<const u16x8 *>(src + (size_t)W * ldc), v11 = *reinterpret_cast<const u16x8 *>(src + (size_t)W * ldc + ldc);
        u16x8 m;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned short t = v00[k];
            t = v01[k] > t ? v01[k] : t;
            t = v10[k] > t ? v10[k] : t;
            t = v11[k] > t ? v11[k] : t;
            m[k] = t;
        }
        *reinterpret_cast<u16x8 *>(out + (size_t)p * (8 * C8) + 8 * c8) = m;
    }
}

hipError_t launch_maxpool2x2_u16(const void *in, int ldc, void *out, int B, int H, int W, int C, hipStream_t s)
{
    if (C % 8 || ldc % 8 || H % 2 || W % 2) return hipErrorInvalidValue;
    const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
    const long long total = (long long)B * Ho * Wo * C8;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool2x2_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(in), ldc,
                       static_cast<uint16_t *>(out), Ho, Wo, C8, total);
    return hipGetLastError();
}

hipError_t launch_maxpool2x2(const float *in, int ldc, float *out, int B, int H, int W, int C, hipStream_t s)
{
    if (C % 4 || ldc % 4 || H % 2 || W % 2) return hipErrorInvalidValue;
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const long long total = (long long)B * Ho * Wo * C4;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, ldc, out, Ho, Wo, C4, total);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// 1x1 head + argmax.  LPP = Cin/4 lanes share a pixel (each holds 4 channels = one 16-byte load, so a wave reads
// 64/LPP whole NHWC rows, fully coalesced); partial dot products are combined with xor-shuffles inside the lane group.
// argmax: strict '>' against -FLT_MAX in class order (src/process.cpp:158-170): ties and NaN keep the lower index.
template <int CLASSES>
__global__ __launch_bounds__(256) void head_argmax_kernel(const float *__restrict__ in, int Cin, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ logits,
                                                          uint8_t *__restrict__ labels, long long npix, int HW)
{
    const int lpp = Cin / 4;
    const int q = threadIdx.x % lpp;
    const int ppb = 256 / lpp;
    f32x4 wr[CLASSES];
    float bs[CLASSES];
#pragma unroll
    for (int k = 0; k < CLASSES; ++k) {
        wr[k] = *reinterpret_cast<const f32x4 *>(w + (size_t)k * Cin + 4 * q);
        bs[k] = bias[k];
    }
    // every lane of a group iterates together (npix is padded to the group count by the loop bound on the group id)
    for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / lpp; p < npix; p += (long long)gridDim.x * ppb) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (size_t)p * Cin + 4 * q);
        float d[CLASSES];
#pragma unroll
        for (int k = 0; k < CLASSES; ++k) {
            // sequential within the lane, then a butterfly over the group: a fixed, data-independent order
            float t = v.x * wr[k].x;
            t += v.y * wr[k].y;
            t += v.z * wr[k].z;
            t += v.w * wr[k].w;
            for (int o = 1; o < lpp; o <<= 1) t += __shfl_xor(t, o, 64);
            d[k] = t + bs[k];
        }
        if (q == 0) {
            const long long bimg = p / HW, pin = p % HW;
            float best = -3.402823466e+38f;
            int idx = 0;
#pragma unroll
            for (int k = 0; k < CLASSES; ++k) {
                if (logits) logits[((size_t)bimg * CLASSES + k) * HW + pin] = d[k];
                if (d[k] > best) { best = d[k]; idx = k; }
            }
            labels[p] = (uint8_t)idx;
        }
    }
}

hipError_t launch_head_argmax(const float *in, int Cin, const float *w, const float *bias, int classes, float *logits,
                              uint8_t *labels, int B, int HW, hipStream_t s)
{
    const int lpp = Cin / 4;
    if (Cin % 4 || lpp > 64 || (lpp & (lpp - 1))) return hipErrorInvalidValue;
    const long long npix = (long long)B * HW;
    const int ppb = 256 / lpp;
    // the shuffle needs whole lane groups active: npix must be a multiple of the pixels per wave
    if (npix % (64 / lpp)) return hipErrorInvalidValue;
    long long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 256 * 16) blocks = 256 * 16;
#define HEAD_CASE(K) case K: hipLaunchKernelGGL(head_argmax_kernel<K>, dim3((unsigned)blocks), dim3(256), 0, s, in, Cin, w, bias, logits, labels, npix, HW); break;
    switch (classes) {
        HEAD_CASE(2) HEAD_CASE(3) HEAD_CASE(4) HEAD_CASE(5) HEAD_CASE(6)
    default: return hipErrorInvalidValue;
    }
#undef HEAD_CASE
    return hipGetLastError();
}


}  // namespace miunet
