// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the UNet forward pass.
//
// Hot kernel: conv_mfma_f32 -- implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain,
// 157 TFLOP/s peak; MI355X_MICROARCH.md "Matrix cores").  GEMM view: M = pixels, N = output channels,
// K = taps x input channels.
//   * a 256-thread workgroup (4 waves, one per SIMD) owns a spatial tile of TH rows x 32 columns (M = 32*TH)
//     and BN output channels; wave w owns rows [w*TH/4, (w+1)*TH/4) x all BN  -> (TH/4)*(BN/32) accumulators of 32x32;
//   * K is walked in chunks of KC = 16 input channels.  Per chunk the (TH+2) x 34 halo patch of the input (NHWC, so
//     16 channels = 64 contiguous bytes per pixel) and the 9 x BN x 16 weight slab are staged ONCE into LDS and all
//     9 taps read shifted windows of the same patch: global->LDS traffic is ~1.3x the tile, not 9x;
//   * LDS rows are padded 16 -> 20 floats so the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots
//     (stride 80 B: 5*p mod 16 is a bijection) -- conflict-free fragment reads for A (pixel-major) and B (cout-major);
//   * the 32x32x2 MFMA consumes k = {k0, k1} from lane halves 0/1.  K order inside a GEMM is free, so each lane reads
//     FOUR consecutive k (one b128) and feeds MFMA step s with k = 8g + 4h + s: one ds_read_b128 per operand per
//     4 MFMAs instead of four ds_read_b32;
//   * the next chunk is prefetched global->registers before the 9-tap MFMA phase and written to LDS after it, so HBM/L2
//     latency hides under >= 18k cycles of matrix work; two workgroups per CU cover each other's barriers;
//   * epilogue fuses the folded-BatchNorm shift (scale is folded into the weights), ReLU and the channel-offset store
//     that makes torch.cat a no-op (skip and upsampled halves share one NHWC buffer).
//   * blockIdx is remapped so the 8 XCDs (private L2s) each walk a contiguous range of (n-tile, m-tile) pairs with the
//     m-tile fastest: neighbouring workgroups on one XCD share the weight slab and halo rows in L2.
//
// The other kernels are HBM-bound (first layer K = 9, pooling, 1x1 head + argmax) and are written for coalesced
// 16-byte-per-lane NHWC access.
#include "kernels.h"

namespace miunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_ROW = KC + 4;   // padded floats per (pixel | cout) row in LDS

template <int TAPS, int TH>
struct TileGeom {
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = 32 + 2 * HALO;
    static constexpr int PH = TH + 2 * HALO;
    static constexpr int NPIX = PW * PH;
    static constexpr int NA4 = NPIX * (KC / 4);               // float4 pieces of the A patch
    static constexpr int A_ITERS = (NA4 + 255) / 256;
    static constexpr int A_FLOATS = NPIX * LDS_ROW;
};

template <int TAPS, int TH, int BN>
constexpr size_t conv_lds_bytes()
{
    return sizeof(float) * (size_t)(TileGeom<TAPS, TH>::A_FLOATS + TAPS * BN * LDS_ROW);
}

// bijective XCD remap (cdna_hip_programming.md §5): blocks b and b+8 share an XCD; give XCD x the logical range
// [start_x, start_x + count_x).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

template <int TAPS, int TH, int BN>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                        const int m_tiles, const int nwg)
{
    using G = TileGeom<TAPS, TH>;
    constexpr int MT = TH / 4;            // 32-row MFMA tiles per wave (one image row each)
    constexpr int NT = BN / 32;           // 32-col MFMA tiles per wave
    constexpr int B_PARTS = BN / 64;      // 64-cout slabs per tap staged by 256 threads x float4
    constexpr int B_ITERS = TAPS * B_PARTS;
    static_assert(TH % 4 == 0 && BN % 64 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const As = lds;
    float *const Bs = lds + G::A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;             // row (pixel x) index for A, column (cout) index for B
    const int lh = lane >> 5;             // k half

    // ---- which tile
    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, n0 = n_tile * BN;

    const float *in_img = a.in + (size_t)b * a.H * a.W * a.ldc;

    // ---- per-thread staging descriptors (chunk invariant)
    int a_goff[G::A_ITERS];               // float offset inside the image, -1 = zero (padding / dead slot)
    int a_loff[G::A_ITERS];               // float offset inside As, -1 = dead slot
#pragma unroll
    for (int s = 0; s < G::A_ITERS; ++s) {
        const int e = tid + 256 * s;
        const int pix = e >> 2, q = e & 3;
        const int py = pix / G::PW, px = pix - py * G::PW;
        const int gy = y0 - G::HALO + py, gx = x0 - G::HALO + px;
        const bool live = e < G::NA4;
        const bool inb = live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        a_goff[s] = inb ? (gy * a.W + gx) * a.ldc + 4 * q : -1;
        a_loff[s] = live ? pix * LDS_ROW + 4 * q : -1;
    }
    const int bq = tid & 3, bn = tid >> 2;                       // float4 piece / cout row inside a 64-cout slab
    const float *w_base = a.wpk + ((size_t)n0 + bn) * KC + 4 * bq;   // + ((chunk*TAPS + tap)*CoutPad + part*64) * KC
    const int b_loff = bn * LDS_ROW + 4 * bq;

    f32x4 a_reg[G::A_ITERS];
    f32x4 b_reg[B_ITERS];

    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * KC;
#pragma unroll
        for (int s = 0; s < G::A_ITERS; ++s) {
            const int q4 = 4 * ((tid + 256 * s) & 3);
            f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (a_goff[s] >= 0 && c0 + q4 < a.Cin) v = *reinterpret_cast<const f32x4 *>(in_img + a_goff[s] + c0);
            a_reg[s] = v;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            b_reg[it] = *reinterpret_cast<const f32x4 *>(
                w_base + (((size_t)chunk * TAPS + tap) * a.CoutPad + part * 64) * KC);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < G::A_ITERS; ++s)
            if (a_loff[s] >= 0) *reinterpret_cast<f32x4 *>(As + a_loff[s]) = a_reg[s];
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            *reinterpret_cast<f32x4 *>(Bs + (tap * BN + part * 64) * LDS_ROW + b_loff) = b_reg[it];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float *a_frag = As + ((wave * MT) * G::PW + li) * LDS_ROW + 4 * lh;
    const float *b_frag = Bs + li * LDS_ROW + 4 * lh;

    const int nchunks = (a.Cin + KC - 1) / KC;
    load_chunk(0);
    store_chunk();
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bool more = chunk + 1 < nchunks;
        if (more) load_chunk(chunk + 1);

#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int g = 0; g < KC / 8; ++g) {
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const f32x4 *>(a_frag + ((i + dy) * G::PW + dx) * LDS_ROW + 8 * g);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[j] = *reinterpret_cast<const f32x4 *>(b_frag + (tap * BN + 32 * j) * LDS_ROW + 8 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                  // every wave is done reading this chunk's LDS image
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31 (cout), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + 32 * j + li;   // GEMM column
        int co, oy_off = 0, ox_off = 0;
        if (TAPS == 9) {
            co = n;
        } else {                          // convT: n = kidx * Cout + co, kidx = dy * 2 + dx
            const int kidx = n / a.Cout;
            co = n - kidx * a.Cout;
            oy_off = kidx >> 1; ox_off = kidx & 1;
        }
        const bool n_ok = (TAPS == 9) ? (n < a.Cout) : (n < 4 * a.Cout);
        const float sh = n_ok ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int y = y0 + wave * MT + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + sh;
                if (a.relu) v = v > 0.f ? v : 0.f;
                if (n_ok && y < a.H && x < a.W) {
                    size_t o;
                    if (TAPS == 9)
                        o = (((size_t)b * a.H + y) * a.W + x) * a.ldo + a.co_off + co;
                    else
                        o = (((size_t)b * 2 * a.H + 2 * y + oy_off) * (2 * a.W) + 2 * x + ox_off) * a.ldo + a.co_off + co;
                    a.out[o] = v;
                }
            }
        }
    }
}

template <int TAPS, int TH, int BN>
static hipError_t launch_conv_cfg(const ConvArgs &a, hipStream_t s)
{
    const int n_total = (TAPS == 9) ? a.Cout : 4 * a.Cout;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (n_total + BN - 1) / BN;
    const int nwg = m_tiles * n_tiles;
    constexpr size_t lds = conv_lds_bytes<TAPS, TH, BN>();
    static bool attr_set = false;
    auto kern = conv_mfma_f32<TAPS, TH, BN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

hipError_t launch_conv3x3_mfma(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_conv_cfg<9, 8, 64>(a, s);
}

hipError_t launch_convT2x2_mfma(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    // a GEMM column tile must not straddle two (dy,dx) taps unless masked per lane: co/kidx are per lane, so any Cout works
    return launch_conv_cfg<1, 8, 64>(a, s);
}

// --------------------------------------------------------------------------------------------------------------------
// First layer (K = 9*Cin with Cin <= 4: HBM-bound on its output).  One thread = one pixel x 4 output channels; the
// Cout/4 threads of a pixel write one contiguous NHWC row.  u8 -> fp32 through the host-built 256-entry table so the
// input equals float(x)/255.0f bit for bit (src/process.cpp:36-39).
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const uint8_t *__restrict__ img, const float *__restrict__ lut,
                                                            const float *__restrict__ w, const float *__restrict__ shift,
                                                            float *__restrict__ out, int B, int H, int W, int Cout,
                                                            int ldo, int quads)
{
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const int q = threadIdx.x % quads;            // which 4 couts
    const int pl = threadIdx.x / quads;           // pixel slot in block
    const int ppb = 256 / quads;
    f32x4 wr[9 * CIN];
#pragma unroll
    for (int t = 0; t < 9 * CIN; ++t) wr[t] = *reinterpret_cast<const f32x4 *>(w + (size_t)t * Cout + 4 * q);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(shift + 4 * q);
    const long long npix = (long long)B * H * W;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < npix; p += (long long)gridDim.x * ppb) {
        const int x = (int)(p % W);
        const int y = (int)((p / W) % H);
        const uint8_t *base = img + (size_t)(p - x - (long long)y * W) * CIN;   // image start
        f32x4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float v = ok ? s_lut[base[((size_t)yy * W + xx) * CIN + c]] : 0.f;
                acc += v * wr[t * CIN + c];
            }
        }
        acc += sh;
        f32x4 r;
        r.x = acc.x > 0.f ? acc.x : 0.f; r.y = acc.y > 0.f ? acc.y : 0.f;
        r.z = acc.z > 0.f ? acc.z : 0.f; r.w = acc.w > 0.f ? acc.w : 0.f;
        *reinterpret_cast<f32x4 *>(out + (size_t)p * ldo + 4 * q) = r;
    }
}

hipError_t launch_conv3x3_first(const uint8_t *img, const float *lut256, const float *w, const float *shift, float *out,
                                int B, int H, int W, int Cin, int Cout, int ldo, hipStream_t s)
{
    const int quads = Cout / 4;
    if (Cout % 4 || quads > 256 || 256 % quads || ldo % 4) return hipErrorInvalidValue;
    const long long npix = (long long)B * H * W;
    const int ppb = 256 / quads;
    long long blocks = (npix + ppb - 1) / ppb;
    if (blocks > 256 * 32) blocks = 256 * 32;
    switch (Cin) {
    case 1: hipLaunchKernelGGL(conv3x3_first_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, img, lut256, w, shift, out, B, H, W, Cout, ldo, quads); break;
    case 3: hipLaunchKernelGGL(conv3x3_first_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, img, lut256, w, shift, out, B, H, W, Cout, ldo, quads); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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
