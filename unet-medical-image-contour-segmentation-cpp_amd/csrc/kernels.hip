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

#include <cstdlib>

namespace miunet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KCT = input channels staged per K-chunk (a multiple of KC = 16, the granule of the packed weights)
template <int TAPS, int TH, int KCT>
struct TileGeom {
    static constexpr int LDS_ROW = KCT + 4;                   // padded floats per (pixel | cout) row in LDS
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PW = 32 + 2 * HALO;
    static constexpr int PH = TH + 2 * HALO;
    static constexpr int NPIX = PW * PH;
    static constexpr int QPP = KCT / 4;                       // float4 pieces per pixel
    static constexpr int NA4 = NPIX * QPP;                    // float4 pieces of the A patch
    static constexpr int A_ITERS = (NA4 + 255) / 256;
    static constexpr int A_FLOATS = NPIX * LDS_ROW;
};

template <int TAPS, int TH, int BN, int KCT>
constexpr size_t conv_lds_bytes()
{
    return sizeof(float) * (size_t)(TileGeom<TAPS, TH, KCT>::A_FLOATS + TAPS * BN * TileGeom<TAPS, TH, KCT>::LDS_ROW);
}

// bijective XCD remap (cdna_hip_programming.md §5): blocks b and b+8 share an XCD; give XCD x the logical range
// [start_x, start_x + count_x).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// NFAST: walk the n-tiles fastest (the workgroups that share an input tile run together on one XCD).  Used for the
// transposed conv, whose K is short and whose N = 4*Cout is wide: the input tile is then fetched from HBM once and
// re-read from L2 by its 8..32 column tiles, instead of once per column tile.
template <int TAPS, int TH, int BN, int KCT, bool NFAST>
__global__ __launch_bounds__(256, 2) void conv_mfma_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                        const int m_tiles, const int nwg)
{
    using G = TileGeom<TAPS, TH, KCT>;
    constexpr int LDS_ROW = G::LDS_ROW;
    constexpr int MT = TH / 4;            // 32-row MFMA tiles per wave (one image row each)
    constexpr int NT = BN / 32;           // 32-col MFMA tiles per wave
    constexpr int B_PARTS = BN / 64;      // 64-cout slabs per tap staged by 256 threads x float4
    constexpr int SUBS = KCT / KC;        // 16-channel weight granules per K-chunk
    constexpr int B_ITERS = TAPS * B_PARTS * SUBS;
    static_assert(TH % 4 == 0 && BN % 64 == 0 && KCT % KC == 0, "tile shape");

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
    const int n_tiles = nwg / m_tiles;
    const int n_tile = NFAST ? L % n_tiles : L / m_tiles;
    int m = NFAST ? L / n_tiles : L - n_tile * m_tiles;
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
        const int pix = e / G::QPP, q = e % G::QPP;
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

    const int nsub16 = (a.Cin + KC - 1) / KC;     // 16-channel granules the packed weights hold
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * KCT;
#pragma unroll
        for (int s = 0; s < G::A_ITERS; ++s) {
            const int q4 = 4 * ((tid + 256 * s) % G::QPP);
            f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (a_goff[s] >= 0 && c0 + q4 < a.Cin) v = *reinterpret_cast<const f32x4 *>(in_img + a_goff[s] + c0);
            a_reg[s] = v;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int sub = it % SUBS, tp = it / SUBS;
            const int tap = tp / B_PARTS, part = tp % B_PARTS;
            const int c16 = chunk * SUBS + sub;
            f32x4 v = { 0.f, 0.f, 0.f, 0.f };
            if (SUBS == 1 || c16 < nsub16)
                v = *reinterpret_cast<const f32x4 *>(w_base + (((size_t)c16 * TAPS + tap) * a.CoutPad + part * 64) * KC);
            b_reg[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < G::A_ITERS; ++s)
            if (a_loff[s] >= 0) *reinterpret_cast<f32x4 *>(As + a_loff[s]) = a_reg[s];
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int sub = it % SUBS, tp = it / SUBS;
            const int tap = tp / B_PARTS, part = tp % B_PARTS;
            *reinterpret_cast<f32x4 *>(Bs + (tap * BN + part * 64) * LDS_ROW + KC * sub + b_loff) = b_reg[it];
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

    const int nchunks = (a.Cin + KCT - 1) / KCT;
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
            for (int g = 0; g < KCT / 8; ++g) {
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
        if (TAPS == 9 && MT == 2 && a.pool_out != nullptr) {
            // rows (y, y+1) are the wave's two MFMA tiles, columns (x, x+1) are registers (r, r+1), r even
            const int yp = (y0 + wave * MT) >> 1;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float m = fmaxf(fmaxf(acc[0][j][r], acc[0][j][r + 1]), fmaxf(acc[MT - 1][j][r], acc[MT - 1][j][r + 1])) + sh;
                if (a.relu) m = m > 0.f ? m : 0.f;             // max and (+shift, ReLU) commute: both are monotone
                if (n_ok && y0 + wave * MT + 1 < a.H && x + 1 < a.W)
                    a.pool_out[(((size_t)b * (a.H >> 1) + yp) * (a.W >> 1) + (x >> 1)) * a.pool_ld + co] = m;
            }
        }
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

template <int TAPS, int TH, int BN, int KCT, bool NFAST>
static hipError_t launch_conv_cfg(const ConvArgs &a, hipStream_t s)
{
    const int n_total = (TAPS == 9) ? a.Cout : 4 * a.Cout;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (n_total + BN - 1) / BN;
    const int nwg = m_tiles * n_tiles;
    constexpr size_t lds = conv_lds_bytes<TAPS, TH, BN, KCT>();
    static bool attr_set = false;
    auto kern = conv_mfma_f32<TAPS, TH, BN, KCT, NFAST>;
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
    return launch_conv_cfg<9, 8, 64, 16, false>(a, s);
}

hipError_t launch_convT2x2_mfma(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    // a GEMM column tile must not straddle two (dy,dx) taps unless masked per lane: co/kidx are per lane, so any Cout works
    static int cfg = -1;
    if (cfg < 0) { const char *e = getenv("MIUNET_CONVT_CFG"); cfg = e ? atoi(e) : 1; }
    switch (cfg) {
    case 0: return launch_conv_cfg<1, 8, 64, 16, false>(a, s);
    case 1: return launch_conv_cfg<1, 8, 64, 16, true>(a, s);
    case 2: return launch_conv_cfg<1, 8, 128, 16, true>(a, s);
    case 4: return launch_conv_cfg<1, 16, 64, 16, true>(a, s);
    default: return launch_conv_cfg<1, 8, 64, 32, true>(a, s);
    }
}

// --------------------------------------------------------------------------------------------------------------------
// conv_mfma_bf16 -- the same implicit GEMM with bf16 operands and fp32 accumulation (BASELINE config 3).
// Layout and schedule are those of conv_mfma_f32; what changes:
//   * a K-chunk is 32 channels; LDS rows hold 32 bf16 + 8 pad = 80 bytes (the same conflict-free stride);
//   * activations are fp32 in HBM: the loader fetches 2 x 16 bytes per 8 channels and rounds them to bf16
//     (v_cvt_pk_bf16_f32, round-to-nearest-even) on the way into LDS -- one 16-byte ds_write per 8 channels;
//   * v_mfma_f32_32x32x16_bf16 takes A[i][8h + j], j = 0..7 from lane (i, h): exactly one ds_read_b128 per operand per
//     MFMA, natural k order, 32 cycles per instruction (16x the fp32 rate) -- the kernel is bound by its staging and LDS
//     traffic and by HBM, not by the matrix pipe.
// The same kernel serves fp16 operands (BASELINE config 5's arithmetic): T = __bf16 or _Float16, 16 bits either way.
template <typename T> struct LpVec { typedef T x8 __attribute__((ext_vector_type(8))); };

template <typename T>
__device__ __forceinline__ typename LpVec<T>::x8 pack_lp8(const f32x4 lo, const f32x4 hi)
{
    typename LpVec<T>::x8 r;
    r[0] = (T)lo[0]; r[1] = (T)lo[1]; r[2] = (T)lo[2]; r[3] = (T)lo[3];
    r[4] = (T)hi[0]; r[5] = (T)hi[1]; r[6] = (T)hi[2]; r[7] = (T)hi[3];
    return r;
}

__device__ __forceinline__ f32x16 mfma_lp(LpVec<__bf16>::x8 a, LpVec<__bf16>::x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_lp(LpVec<_Float16>::x8 a, LpVec<_Float16>::x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <typename T, int TAPS, int TH, int BN, bool NFAST>
__global__ __launch_bounds__(256, 2) void conv_mfma_bf16(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                         const int m_tiles, const int nwg)
{
    typedef typename LpVec<T>::x8 bf16x8;
    constexpr int ROW = KC_BF16 + 8;                     // bf16 elements per LDS row (80 bytes)
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PW = 32 + 2 * HALO, PH = TH + 2 * HALO, NPIX = PW * PH;
    constexpr int NA8 = NPIX * (KC_BF16 / 8);            // 8-channel pieces of the A patch
    constexpr int A_ITERS = (NA8 + 255) / 256;
    constexpr int MT = TH / 4, NT = BN / 32;
    constexpr int B_PARTS = BN / 64;                     // 64 rows x 64 bytes = 4 KB = 256 threads x 16 bytes
    constexpr int B_ITERS = TAPS * B_PARTS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    T *const As = reinterpret_cast<T *>(lds);
    T *const Bs = As + NPIX * ROW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tiles = nwg / m_tiles;
    const int n_tile = NFAST ? L % n_tiles : L / m_tiles;
    int m = NFAST ? L / n_tiles : L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, n0 = n_tile * BN;
    const float *in_img = a.in + (size_t)b * a.H * a.W * a.ldc;
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
        a_loff[s] = live ? pix * ROW + 8 * q : -1;
    }
    const int bq = tid & 3, bn = tid >> 2;               // 16-byte piece (8 bf16) / cout row inside a 64-cout slab
    const T *w_base = wpk + ((size_t)n0 + bn) * KC_BF16 + 8 * bq;
    const int b_loff = bn * ROW + 8 * bq;

    f32x4 a_lo[A_ITERS], a_hi[A_ITERS];
    bf16x8 b_reg[B_ITERS];
    auto load_chunk = [&](int chunk) {
        const int c0 = chunk * KC_BF16;
#pragma unroll
        for (int s = 0; s < A_ITERS; ++s) {
            const int q8 = 8 * ((tid + 256 * s) & 3);
            f32x4 lo = { 0.f, 0.f, 0.f, 0.f }, hi = { 0.f, 0.f, 0.f, 0.f };
            if (a_goff[s] >= 0) {
                if (c0 + q8 < a.Cin) lo = *reinterpret_cast<const f32x4 *>(in_img + a_goff[s] + c0);
                if (c0 + q8 + 4 < a.Cin) hi = *reinterpret_cast<const f32x4 *>(in_img + a_goff[s] + c0 + 4);
            }
            a_lo[s] = lo; a_hi[s] = hi;
        }
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            b_reg[it] = *reinterpret_cast<const bf16x8 *>(w_base + (((size_t)chunk * TAPS + tap) * a.CoutPad + part * 64) * KC_BF16);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < A_ITERS; ++s)
            if (a_loff[s] >= 0) *reinterpret_cast<bf16x8 *>(As + a_loff[s]) = pack_lp8<T>(a_lo[s], a_hi[s]);
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) {
            const int tap = it / B_PARTS, part = it % B_PARTS;
            *reinterpret_cast<bf16x8 *>(Bs + (tap * BN + part * 64) * ROW + b_loff) = b_reg[it];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const T *a_frag = As + ((wave * MT) * PW + li) * ROW + 8 * lh;
    const T *b_frag = Bs + li * ROW + 8 * lh;
    const int nchunks = (a.Cin + KC_BF16 - 1) / KC_BF16;
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
            for (int g = 0; g < KC_BF16 / 16; ++g) {
                bf16x8 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const bf16x8 *>(a_frag + ((i + dy) * PW + dx) * ROW + 16 * g);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8 *>(b_frag + (tap * BN + 32 * j) * ROW + 16 * g);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = mfma_lp(af[i], bf[j], acc[i][j]);
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue (identical to the fp32 kernel's)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + 32 * j + li;
        int co, oy_off = 0, ox_off = 0;
        if (TAPS == 9) {
            co = n;
        } else {
            const int kidx = n / a.Cout;
            co = n - kidx * a.Cout;
            oy_off = kidx >> 1; ox_off = kidx & 1;
        }
        const bool n_ok = (TAPS == 9) ? (n < a.Cout) : (n < 4 * a.Cout);
        const float sh = n_ok ? a.bias[co] : 0.f;
        if (TAPS == 9 && MT == 2 && a.pool_out != nullptr) {
            const int yp = (y0 + wave * MT) >> 1;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float mx = fmaxf(fmaxf(acc[0][j][r], acc[0][j][r + 1]), fmaxf(acc[MT - 1][j][r], acc[MT - 1][j][r + 1])) + sh;
                if (a.relu) mx = mx > 0.f ? mx : 0.f;
                if (n_ok && y0 + wave * MT + 1 < a.H && x + 1 < a.W)
                    a.pool_out[(((size_t)b * (a.H >> 1) + yp) * (a.W >> 1) + (x >> 1)) * a.pool_ld + co] = mx;
            }
        }
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

template <typename T, int TAPS, int TH, int BN, bool NFAST>
static hipError_t launch_bf16_cfg(const ConvArgs &a, hipStream_t s)
{
    const int n_total = (TAPS == 9) ? a.Cout : 4 * a.Cout;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TH - 1) / TH;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (n_total + BN - 1) / BN;
    const int nwg = m_tiles * n_tiles;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr size_t lds = 2 * (size_t)(KC_BF16 + 8) * ((32 + 2 * HALO) * (TH + 2 * HALO) + TAPS * BN);
    static bool attr_set = false;
    auto kern = conv_mfma_bf16<T, TAPS, TH, BN, NFAST>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

hipError_t launch_conv3x3_bf16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_bf16_cfg<__bf16, 9, 8, 64, false>(a, s);
}

hipError_t launch_convT2x2_bf16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_bf16_cfg<__bf16, 1, 8, 64, true>(a, s);
}

hipError_t launch_conv3x3_fp16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_bf16_cfg<_Float16, 9, 8, 64, false>(a, s);
}

hipError_t launch_convT2x2_fp16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 8 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_bf16_cfg<_Float16, 1, 8, 64, true>(a, s);
}

// --------------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) convolution on the fp32 MFMA (Lavin & Gray's minimal filtering: 16 multiplies per 2x2 output tile
// and channel pair instead of 36 -- 2.25x fewer MACs, all arithmetic still fp32).
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A
// The 16 element-wise products are 16 independent GEMMs  M_p[tile][co] = sum_ci V_p[tile][ci] * U_p[ci][co]
// (p = 4*xi + nu).  Mapping to CDNA4:
//   * a workgroup = 4 waves (one per SIMD, the whole 512-entry register file each) owns 32*WM tiles (16 x 8*WM output
//     pixels) x 32*WN output channels; wave (wm, wn) owns ONE 32-tile x 32-channel MFMA tile for ALL 16 positions
//     = 16 accumulators of 32x32 (256 registers).  Because a lane then holds M_p for every p of the same (tile, channel),
//     the inverse transform A^T M A is pure in-lane arithmetic: no LDS exchange, no second pass;
//   * U fragments are wave-private (each wave has its own 32 output channels), so they never touch LDS: every lane
//     loads its 16-byte fragment straight from global memory (one fully coalesced 1 KB wave-load per position), issued
//     one whole K-chunk (64 MFMAs = 4096 cycles) ahead into the register the previous chunk just released;
//   * V = B^T d B is shared by the waves of a tile row, so it is built once per chunk into a double-buffered LDS image
//     [pos][tile][8 + 4 pad] (48-byte rows: 3i mod 16 is a bijection -> conflict-free ds_read_b128): 256 threads =
//     64 tiles x 2 channel quads x 2 halves of xi; each thread loads its 3 x 4 pixels x 4 channels directly from the
//     NHWC input (zero padding by predication), applies B^T .. B in registers and writes 8 x 16 bytes.  The loads for
//     chunk c+1 are in flight during the MFMAs of chunk c; one barrier per chunk;
//   * K order inside a chunk is permuted exactly as in the direct kernel (MFMA step s consumes k = 4h + s).
template <int WM, int WN>
struct WinoGeom {
    static constexpr int TMB = 32 * WM;                    // 2x2 output tiles per workgroup (8 wide x 4*WM tall)
    static constexpr int VROW = WINO_KC + 4;               // padded floats per tile row of V
    static constexpr int VBUF = 16 * TMB * VROW;           // floats per V buffer
    static constexpr int PROWS = 8 * WM + 2;               // raw patch rows (16 x 8*WM output pixels + halo)
    static constexpr int RAWPIX = PROWS * 18;
    static constexpr int RAW_P = WINO_SC + 4;              // padded floats per raw pixel (2-way worst case on ds_read_b128)
    static constexpr int RAW_FLOATS = RAWPIX * RAW_P;
    static constexpr int RAW_ITERS = (RAWPIX * (WINO_SC / 4) + 255) / 256;
    static constexpr size_t LDS_BYTES = sizeof(float) * (2 * VBUF + RAW_FLOATS);
};

template <int WM, int WN>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                           const int m_tiles, const int nwg)
{
    static_assert(WM * WN == 4, "four waves");
    using G = WinoGeom<WM, WN>;
    constexpr int TMB = G::TMB, VROW = G::VROW, VBUF = G::VBUF, RAW_P = G::RAW_P;
    constexpr int CPS = WINO_SC / WINO_KC;    // K-chunks per super-chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Vs = lds;                    // [2][16][TMB][VROW]
    float *const Raw = lds + 2 * VBUF;        // [RAWPIX][RAW_P]: the input halo patch of ONE 32-channel super-chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int bx0 = tx * 16, by0 = ty * 8 * WM, n0 = n_tile * 32 * WN;
    const float *in_img = a.in + (size_t)b * a.H * a.W * a.ldc;

    // ---- stage 1 of the input path: the raw halo patch, 32 channels at a time, global -> registers -> LDS.
    // 8 consecutive lanes fetch one pixel's 128 contiguous bytes (whole cache lines, each fetched once per workgroup);
    // out-of-image pixels (zero padding), channels past Cin and dead slots get voffset 0xFFFFFFFF, which the buffer
    // range check turns into zeros: no branches.
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_img), 0, a.H * a.W * a.ldc * 4, 0x00020000);
    unsigned raw_voff[G::RAW_ITERS];          // byte offset in the image (channel 4*q8 of super-chunk 0), or 0xFFFFFFFF
    int raw_loff[G::RAW_ITERS];               // float offset in Raw, -1 = dead slot
#pragma unroll
    for (int s = 0; s < G::RAW_ITERS; ++s) {
        const int e = tid + 256 * s;
        const int pix = e >> 3, q8 = e & 7;
        const int py = pix / 18, px = pix - py * 18;
        const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
        const bool live = pix < G::RAWPIX;
        const bool inb = live && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        raw_voff[s] = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * q8) * 4) : 0xFFFFFFFFu;
        raw_loff[s] = live ? pix * RAW_P + 4 * q8 : -1;
    }
    f32x4 raw_reg[G::RAW_ITERS];
    auto raw_load = [&](int super) {
        const int c0 = super * WINO_SC;
#pragma unroll
        for (int s = 0; s < G::RAW_ITERS; ++s) {
            const int q8 = (tid + 256 * s) & 7;
            const unsigned voff = (c0 + 4 * q8 < a.Cin) ? raw_voff[s] : 0xFFFFFFFFu;
            raw_reg[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff, c0 * 4, 0));
        }
    };
    auto raw_store = [&]() {
#pragma unroll
        for (int s = 0; s < G::RAW_ITERS; ++s)
            if (raw_loff[s] >= 0) *reinterpret_cast<f32x4 *>(Raw + raw_loff[s]) = raw_reg[s];
    };

    // ---- stage 2: V = B^T d B for one K-chunk (8 channels).  256 threads share 32*WM tiles x 2 channel quads:
    //   WM = 2: thread = (tile, quad, HALF of xi): half 0 builds xi = 0,1 from patch rows (0,1,2), half 1 builds xi = 2,3
    //           from rows (2,3,1); with the rows in that order both halves use  ta = l0 - l2,  tb = sgn*l1 + l2.
    //   WM = 1: thread = (tile, quad, ONE xi): row xi of B^T d is  la + sgn*lb  with (la, lb, sgn) =
    //           xi 0: (d0, d2, -) | xi 1: (d1, d2, +) | xi 2: (d2, d1, -) | xi 3: (d1, d3, -).
    // Either way there is no per-lane select (multiplying by +-1 is exact) and every thread has work.
    constexpr bool XI_MODE = (WM == 1);
    constexpr int NROWS = XI_MODE ? 2 : 3;        // patch rows a thread reads
    constexpr int NXI = XI_MODE ? 1 : 2;          // xi rows a thread produces
    constexpr int NPIECES = 4 + 4 * NXI;          // transform pieces threaded between the MFMAs
    const int t_tile = XI_MODE ? (tid >> 3) : (tid >> 2);
    const int t_quad = tid & 1;
    const int t_sel = XI_MODE ? ((tid >> 1) & 3) : ((tid >> 1) & 1);      // xi (WM = 1) or half (WM = 2)
    const float t_sgn = XI_MODE ? (t_sel == 1 ? 1.f : -1.f) : (t_sel ? -1.f : 1.f);
    int p_loff[NROWS];                        // float offset in Raw of its patch row r, column 0, channel 4*quad
    {
        const int i = t_tile & 31, mt = t_tile >> 5;
        const int pr0 = 2 * ((i >> 3) + 4 * mt), pc0 = 2 * (i & 7);
#pragma unroll
        for (int r = 0; r < NROWS; ++r) {
            int prow;
            if (XI_MODE) prow = r == 0 ? (t_sel == 0 ? 0 : t_sel == 2 ? 2 : 1) : (t_sel == 0 ? 2 : t_sel == 1 ? 2 : t_sel == 2 ? 1 : 3);
            else prow = t_sel ? (r == 0 ? 2 : r == 1 ? 3 : 1) : r;
            p_loff[r] = ((pr0 + prow) * 18 + pc0) * RAW_P + 4 * t_quad;
        }
    }
    f32x4 patch[NROWS][4];
    auto patch_read = [&](int r, int chunk_in_super) {       // one patch row (4 pixels) of this thread, LDS -> registers
#pragma unroll
        for (int c = 0; c < 4; ++c)
            patch[r][c] = *reinterpret_cast<const f32x4 *>(Raw + p_loff[r] + c * RAW_P + chunk_in_super * WINO_KC);
    };
    float *const v_wr = Vs + ((XI_MODE ? t_sel * 4 : t_sel * 8) * TMB + t_tile) * VROW + 4 * t_quad;   // + buf*VBUF + (xi_local*4 + nu)*TMB*VROW
    // pieces 0-3 = B^T d for patch column k, pieces 4.. = one V position each (.. B, then its 16-byte store)
    f32x4 t_rows[NXI][4];
    auto transform_piece = [&](int k, int buf) {
        if (k < 4) {
            if constexpr (XI_MODE) {
                t_rows[0][k] = t_sgn * patch[1][k] + patch[0][k];
            } else {
                t_rows[0][k] = patch[0][k] - patch[2][k];
                t_rows[1][k] = t_sgn * patch[1][k] + patch[2][k];
            }
        } else {
            const int x = (k - 4) >> 2, nu = (k - 4) & 3;
            const f32x4 *t = t_rows[x];
            const f32x4 v = nu == 0 ? t[0] - t[2] : nu == 1 ? t[1] + t[2] : nu == 2 ? t[2] - t[1] : t[1] - t[3];
            *reinterpret_cast<f32x4 *>(v_wr + buf * VBUF + (x * 4 + nu) * TMB * VROW) = v;
        }
    };

    // ---- MFMA role: wave (wm, wn), all 16 positions
    // U fragments: buffer loads whose per-position / per-chunk displacement is a SCALAR offset (no per-lane address
    // arithmetic in the loop); the descriptor covers this layer's whole packed U (< 4 GB).
    const int ncol = n0 + 32 * wn + li;
    const unsigned u_pos_bytes = (unsigned)a.CoutPad * WINO_KC * 4;
    const unsigned u_voff = (unsigned)(ncol * WINO_KC + 4 * lh) * 4;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.wpk), 0, (int)(((a.Cin + WINO_KC - 1) / WINO_KC) * 16 * u_pos_bytes), 0x00020000);
    auto u_load = [&](int chunk, int p) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff, (chunk * 16 + p) * u_pos_bytes, 0));
    };
    const float *v_rd = Vs + (32 * wm + li) * VROW + 4 * lh;                 // + buf*VBUF + pos*TMB*VROW

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nchunks = (a.Cin + WINO_KC - 1) / WINO_KC;
    const int nsuper = (a.Cin + WINO_SC - 1) / WINO_SC;
    f32x4 u[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] = u_load(0, p);
    // prologue: raw patch of super-chunk 0 -> LDS, V of chunk 0
    raw_load(0);
    raw_store();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NROWS; ++r) patch_read(r, 0);
#pragma unroll
    for (int k = 0; k < NPIECES; ++k) transform_piece(k, 0);
    __syncthreads();

    // Per K-chunk c (super-chunk S = c / 4, j = c % 4):
    //   j == 0 : issue the buffer loads of super-chunk S+1's raw patch (registers; they have three chunks to land)
    //   j == 3 : write them to Raw (its last reader, the transform of chunk (S,3), ran during chunk (S,2)) + one extra barrier
    //   always : 64 MFMAs; the V fragment one position ahead; the U refill one chunk ahead; during positions 4-6 the 12
    //            ds_reads of the NEXT chunk's patch, during positions 8-13 its transform, one piece after every other MFMA
    //            (a lone wave per SIMD issues in order: a filler only hides if it sits BETWEEN two MFMAs); one barrier.
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int j = chunk & (CPS - 1), S = chunk / CPS;
        if (j == 0 && S + 1 < nsuper) raw_load(S + 1);
        if (j == CPS - 1 && S + 1 < nsuper) {
            raw_store();
            __syncthreads();
        }
        const int nxt = (chunk + 1 < nchunks) ? chunk + 1 : chunk;       // last iteration re-does itself: straight-line code
        const int nxt_j = nxt & (CPS - 1);
        const float *vb = v_rd + (chunk & 1) * VBUF;
        const int wbuf = (chunk + 1) & 1;
        f32x4 av = *reinterpret_cast<const f32x4 *>(vb);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f32x4 avn = av;
            if (p + 1 < 16) avn = *reinterpret_cast<const f32x4 *>(vb + (p + 1) * TMB * VROW);   // V fragment one position ahead
            if (p >= 4 && p < 4 + NROWS) patch_read(p - 4, nxt_j);
            __builtin_amdgcn_sched_barrier(0);        // ... issued BEFORE this position's MFMAs (hipcc would sink them to their use)
            const f32x4 bv = u[p];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[p], 0, 0, 0);
                const int idx = (p - 8) * 4 + s;
                if (p >= 8 && (idx & 1) == 0 && idx / 2 < NPIECES) {
                    transform_piece(idx / 2, wbuf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            u[p] = u_load(nxt, p);                                                              // refill a full chunk ahead
            av = avn;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in-lane, + shift, ReLU, 2x2 store.  Lane = channel, register r = tile row of the MFMA tile.
    const bool n_ok = ncol < a.Cout;
    const float sh = n_ok ? a.bias[ncol] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int oy = by0 + 2 * ((i >> 3) + 4 * wm), ox = bx0 + 2 * (i & 7);
        float s0[4], s1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float m0 = acc[0 + nu][r], m1 = acc[4 + nu][r], m2 = acc[8 + nu][r], m3 = acc[12 + nu][r];
            s0[nu] = m0 + m1 + m2;
            s1[nu] = m1 - m2 - m3;
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2]; y[1][1] = s1[1] - s1[2] - s1[3];
        float vmax = -3.402823466e+38f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v = y[dy][dx] + sh;
                if (a.relu) v = v > 0.f ? v : 0.f;
                vmax = fmaxf(vmax, v);
                if (n_ok && oy + dy < a.H && ox + dx < a.W)
                    a.out[(((size_t)b * a.H + oy + dy) * a.W + ox + dx) * a.ldo + a.co_off + ncol] = v;
            }
        if (a.pool_out != nullptr && n_ok && oy + 1 < a.H && ox + 1 < a.W)    // the lane's 2x2 tile IS one pooling window
            a.pool_out[(((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * a.pool_ld + ncol] = vmax;
    }
}

// --------------------------------------------------------------------------------------------------------------------
// conv3x3_wino16_f32 -- the same Winograd F(2x2,3x3) algorithm re-tiled for TWO waves per SIMD.
// The 4-wave kernel above gives each wave a 32x32 MFMA tile for all 16 positions = 256 accumulator registers, which
// leaves one wave per SIMD: every in-order issue stall (a VMEM issue, an LDS wait, the per-chunk barrier) idles the
// matrix pipe.  Here a workgroup is 8 waves and each wave owns 32 tiles x 16 output channels on
// v_mfma_f32_16x16x4_f32 (two 16x16 blocks per position -> 128 accumulator registers), so two waves share a SIMD and
// cover each other's stalls, while a lane STILL holds all 16 positions of its (tile, channel) pairs: the inverse
// transform stays in-lane.  Same workgroup tile (64 tiles x 64 channels), same LDS images (raw patch + double-buffered
// V), same register-streamed U -- packed so that one 16-byte load per lane covers a pair of positions.
//   MFMA operand maps (16x16x4): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
//   C/D col = lane & 15, row = 4 * (lane >> 4) + reg.  Lane group kq = lane >> 4 feeds k = 2*kq + s in step s (K order
//   inside a chunk is free), so a lane's two k come from one ds_read_b64 / one half of its 16-byte U fragment.
constexpr int W16_THREADS = 512;

__global__ __launch_bounds__(W16_THREADS, 2) void conv3x3_wino16_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                                    const int m_tiles, const int nwg)
{
    using G = WinoGeom<2, 2>;
    constexpr int TMB = G::TMB, VROW = G::VROW, VBUF = G::VBUF, RAW_P = G::RAW_P;
    constexpr int CPS = WINO_SC / WINO_KC;
    constexpr int RAW_ITERS = (G::RAWPIX * (WINO_SC / 4) + W16_THREADS - 1) / W16_THREADS;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Vs = lds;
    float *const Raw = lds + 2 * VBUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wq = wave & 3;          // 32-tile group, 16-channel group
    const int i16 = lane & 15, kq = lane >> 4;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int bx0 = tx * 16, by0 = ty * 16, n0 = n_tile * 64;
    const float *in_img = a.in + (size_t)b * a.H * a.W * a.ldc;

    // ---- raw halo patch, 32 channels at a time (whole 128-byte lines; zero padding through the buffer range check)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_img), 0, a.H * a.W * a.ldc * 4, 0x00020000);
    unsigned raw_voff[RAW_ITERS];
#pragma unroll
    for (int s = 0; s < RAW_ITERS; ++s) {
        const int e = tid + W16_THREADS * s;
        const int pix = e >> 3, q8 = e & 7;
        const int py = pix / 18, px = pix - py * 18;
        const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
        const bool inb = pix < G::RAWPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        raw_voff[s] = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * q8) * 4) : 0xFFFFFFFFu;
    }
    f32x4 raw_reg[RAW_ITERS];
    auto raw_load = [&](int super) {
        const int c0 = super * WINO_SC;
#pragma unroll
        for (int s = 0; s < RAW_ITERS; ++s) {
            const int q8 = (tid + W16_THREADS * s) & 7;
            const unsigned voff = (c0 + 4 * q8 < a.Cin) ? raw_voff[s] : 0xFFFFFFFFu;
            raw_reg[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, voff, c0 * 4, 0));
        }
    };
    auto raw_store = [&]() {
#pragma unroll
        for (int s = 0; s < RAW_ITERS; ++s) {
            const int e = tid + W16_THREADS * s;
            if ((e >> 3) < G::RAWPIX) *reinterpret_cast<f32x4 *>(Raw + (e >> 3) * RAW_P + 4 * (e & 7)) = raw_reg[s];
        }
    };

    // ---- input transform role: thread = (tile, channel quad, xi).  Row xi of B^T d is  la + sgn * lb  with
    //      xi 0: d0 - d2 | xi 1: d1 + d2 | xi 2: d2 - d1 | xi 3: d1 - d3   (one fma per element, no select)
    const int t_tile = tid >> 3, t_quad = tid & 1, t_xi = (tid >> 1) & 3;
    const float t_sgn = t_xi == 1 ? 1.f : -1.f;
    int p_la, p_lb;
    {
        const int i = t_tile & 31, mt = t_tile >> 5;
        const int pr0 = 2 * ((i >> 3) + 4 * mt), pc0 = 2 * (i & 7);
        const int ra = t_xi == 0 ? 0 : t_xi == 2 ? 2 : 1;
        const int rb = t_xi == 0 ? 2 : t_xi == 1 ? 2 : t_xi == 2 ? 1 : 3;
        p_la = ((pr0 + ra) * 18 + pc0) * RAW_P + 4 * t_quad;
        p_lb = ((pr0 + rb) * 18 + pc0) * RAW_P + 4 * t_quad;
    }
    float *const v_wr = Vs + (t_xi * 4 * TMB + t_tile) * VROW + 4 * t_quad;     // + buf*VBUF + nu*TMB*VROW
    f32x4 pa[4], pb[4], tcol[4];
    auto xf_read = [&](int c, int cj) {
        pa[c] = *reinterpret_cast<const f32x4 *>(Raw + p_la + c * RAW_P + cj * WINO_KC);
        pb[c] = *reinterpret_cast<const f32x4 *>(Raw + p_lb + c * RAW_P + cj * WINO_KC);
    };
    auto xf_rows = [&](int c) { tcol[c] = t_sgn * pb[c] + pa[c]; };
    auto xf_out = [&](int nu, int buf) {
        const f32x4 v = nu == 0 ? tcol[0] - tcol[2] : nu == 1 ? tcol[1] + tcol[2] : nu == 2 ? tcol[2] - tcol[1] : tcol[1] - tcol[3];
        *reinterpret_cast<f32x4 *>(v_wr + buf * VBUF + nu * TMB * VROW) = v;
    };

    // ---- MFMA role
    const int ncol = n0 + 16 * wq + i16;
    const size_t u_pp_stride = (size_t)a.CoutPad * 16;                         // floats per position pair
    const float *u_lane = a.wpk + (size_t)ncol * 16 + 4 * kq;                  // + (chunk*8 + pp) * u_pp_stride
    const float *v_rd = Vs + (32 * wm + i16) * VROW + 2 * kq;                  // + buf*VBUF + pos*TMB*VROW + mb*16*VROW

    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc[p][mb] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    const int nchunks = (a.Cin + WINO_KC - 1) / WINO_KC;
    const int nsuper = (a.Cin + WINO_SC - 1) / WINO_SC;
    f32x4 u[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) u[pp] = *reinterpret_cast<const f32x4 *>(u_lane + (size_t)pp * u_pp_stride);
    raw_load(0);
    raw_store();
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) { xf_read(c, 0); xf_rows(c); }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) xf_out(nu, 0);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int j = chunk & (CPS - 1), S = chunk / CPS;
        if (j == 0 && S + 1 < nsuper) raw_load(S + 1);
        if (j == CPS - 1 && S + 1 < nsuper) {
            raw_store();
            __syncthreads();
        }
        const int nxt = (chunk + 1 < nchunks) ? chunk + 1 : chunk;
        const int nxt_j = nxt & (CPS - 1);
        const float *vb = v_rd + (chunk & 1) * VBUF;
        const float *un = u_lane + (size_t)nxt * 8 * u_pp_stride;
        const int wbuf = (chunk + 1) & 1;
        f32x2 a0 = *reinterpret_cast<const f32x2 *>(vb), a1 = *reinterpret_cast<const f32x2 *>(vb + 16 * VROW);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f32x2 n0v = a0, n1v = a1;
            if (p + 1 < 16) {
                n0v = *reinterpret_cast<const f32x2 *>(vb + (p + 1) * TMB * VROW);
                n1v = *reinterpret_cast<const f32x2 *>(vb + (p + 1) * TMB * VROW + 16 * VROW);
            }
            // next chunk's input transform, spread over the positions: LDS reads first, then rows, then outputs
            if (p == 1) xf_read(0, nxt_j);
            if (p == 2) xf_read(1, nxt_j);
            if (p == 3) { xf_rows(0); xf_read(2, nxt_j); }
            if (p == 4) { xf_rows(1); xf_read(3, nxt_j); }
            if (p == 5) xf_rows(2);
            if (p == 6) xf_rows(3);
            if (p >= 8 && p < 12) xf_out(p - 8, wbuf);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 uv = u[p >> 1];
            const float b0 = (p & 1) ? uv[2] : uv[0], b1 = (p & 1) ? uv[3] : uv[1];
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b0, acc[p][0], 0, 0, 0);
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b0, acc[p][1], 0, 0, 0);
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b1, acc[p][0], 0, 0, 0);
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b1, acc[p][1], 0, 0, 0);
            if (p & 1) u[p >> 1] = *reinterpret_cast<const f32x4 *>(un + (size_t)(p >> 1) * u_pp_stride);   // refill one chunk ahead
            a0 = n0v; a1 = n1v;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in-lane.  Lane = channel, (mb, reg) = tile.
    const bool n_ok = ncol < a.Cout;
    const float sh = n_ok ? a.bias[ncol] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * mb + 4 * kq + r;
            const int oy = by0 + 2 * ((i >> 3) + 4 * wm), ox = bx0 + 2 * (i & 7);
            float s0[4], s1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float m0 = acc[0 + nu][mb][r], m1 = acc[4 + nu][mb][r], m2 = acc[8 + nu][mb][r], m3 = acc[12 + nu][mb][r];
                s0[nu] = m0 + m1 + m2;
                s1[nu] = m1 - m2 - m3;
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s0[1] - s0[2] - s0[3];
            y[1][0] = s1[0] + s1[1] + s1[2]; y[1][1] = s1[1] - s1[2] - s1[3];
            float vmax = -3.402823466e+38f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float v = y[dy][dx] + sh;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    vmax = fmaxf(vmax, v);
                    if (n_ok && oy + dy < a.H && ox + dx < a.W)
                        a.out[(((size_t)b * a.H + oy + dy) * a.W + ox + dx) * a.ldo + a.co_off + ncol] = v;
                }
            if (a.pool_out != nullptr && n_ok && oy + 1 < a.H && ox + 1 < a.W)
                a.pool_out[(((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1)) * a.pool_ld + ncol] = vmax;
        }
}

static hipError_t launch_wino16(const ConvArgs &a, hipStream_t s)
{
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 63) / 64;
    const int nwg = m_tiles * n_tiles;
    constexpr size_t lds = WinoGeom<2, 2>::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wino16_f32),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_wino16_f32, dim3(nwg), dim3(W16_THREADS), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_wino16(a, s);
}

template <int WM, int WN>
static hipError_t launch_wino_cfg(const ConvArgs &a, hipStream_t s)
{
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 8 * WM - 1) / (8 * WM);
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 32 * WN - 1) / (32 * WN);
    const int nwg = m_tiles * n_tiles;
    constexpr size_t lds = WinoGeom<WM, WN>::LDS_BYTES;
    static bool attr_set = false;
    auto kern = conv3x3_wino_f32<WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    // Cout >= 128: 32 tiles x 128 channels per workgroup (half the input-transform work per MFMA); else 64 x 64
    if (a.Cout > 64) return launch_wino_cfg<1, 4>(a, s);
    return launch_wino_cfg<2, 2>(a, s);
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

// --------------------------------------------------------------------------------------------------------------------
// RAW16 preprocessing on the device: HBM-bound integer scan + a 262144-pixel fp64 gather.
__global__ __launch_bounds__(256) void minmax_init_kernel(unsigned *mnmx)
{
    if (threadIdx.x == 0) { mnmx[0] = 65535u; mnmx[1] = 0u; }
}

__global__ __launch_bounds__(256) void minmax_u16_kernel(const uint16_t *__restrict__ raw, size_t n, unsigned *mnmx)
{
    unsigned lo = 65535u, hi = 0u;
    const size_t n8 = n / 8;                                   // 16 bytes = 8 samples per lane
    const uint4 *v = reinterpret_cast<const uint4 *>(raw);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 q = v[i];
        const unsigned ws[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = ws[k] & 0xFFFFu, b = ws[k] >> 16;
            lo = min(lo, min(a, b));
            hi = max(hi, max(a, b));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {            // ragged tail
        const unsigned a = raw[n8 * 8 + threadIdx.x];
        lo = min(lo, a); hi = max(hi, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                          // wave64 butterfly
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mnmx[0], lo);
        atomicMax(&mnmx[1], hi);
    }
}

hipError_t launch_minmax_u16(const uint16_t *raw, size_t n, unsigned *mnmx, hipStream_t s)
{
    if (n == 0 || (reinterpret_cast<uintptr_t>(raw) & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, mnmx);
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(minmax_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, raw, n, mnmx);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void resample_u8_kernel(const uint16_t *__restrict__ raw, int w, int h,
                                                          const unsigned *__restrict__ mnmx, uint8_t *__restrict__ dst,
                                                          int outW, int outH)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= outW || y >= outH) return;
    const unsigned short mn = (unsigned short)mnmx[0];
    unsigned short mx = (unsigned short)mnmx[1];
    if (mn == mx) mx = (unsigned short)(mn + 1);                // evaluated in uint16_t: wraps to 0 at 65535 (src/preprocess.cpp:92)
    const double scale8 = 255.0 / (double)((int)mx - (int)mn);
    const double stepX = (double)w / (double)outW, stepY = (double)h / (double)outH;
    const double fx = __dmul_rn((double)x, stepX), fy = __dmul_rn((double)y, stepY);
    const int ix = (int)fx, iy = (int)fy;
    const int ix1 = ix + 1 < w - 1 ? ix + 1 : w - 1;
    const int iy1 = iy + 1 < h - 1 ? iy + 1 : h - 1;
    const double dx = __dsub_rn(fx, (double)ix), dy = __dsub_rn(fy, (double)iy);
    const double v00 = raw[(size_t)iy * w + ix], v01 = raw[(size_t)iy * w + ix1];
    const double v10 = raw[(size_t)iy1 * w + ix], v11 = raw[(size_t)iy1 * w + ix1];
    const double omdx = __dsub_rn(1.0, dx), omdy = __dsub_rn(1.0, dy);
    // (1-dx)*(1-dy)*v00 + dx*(1-dy)*v01 + (1-dx)*dy*v10 + dx*dy*v11, left to right, one rounding per operation
    double v = __dmul_rn(__dmul_rn(omdx, omdy), v00);
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(dx, omdy), v01));
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(omdx, dy), v10));
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(dx, dy), v11));
    const double q = __dadd_rn(__dmul_rn(__dsub_rn(v, (double)mn), scale8), 0.5);
    dst[(size_t)y * outW + x] = (uint8_t)(int)q;
}

hipError_t launch_resample_u8(const uint16_t *raw, int w, int h, const unsigned *mnmx, uint8_t *dst, int outW, int outH,
                              hipStream_t s)
{
    if (w <= 0 || h <= 0 || outW <= 0 || outH <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((outW + 63) / 64, (outH + 3) / 4), dim3(256), 0, s, raw, w, h, mnmx, dst, outW, outH);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// postprocess_mask on the device.  Byte/integer work, HBM/L2-bound and tiny next to the network:
// the point is to keep the label maps on the device and to replace the reference's O(components x H x W) loops
// (src/postprocess.cpp:41, :71) by one union-find labelling.
namespace pp {

__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int find_root(const int *parent, int x)
{
    int p = ld(parent + x);
    while (p != x) { x = p; p = ld(parent + x); }
    return x;
}

// parents only ever decrease, roots satisfy parent[r] == r; atomicMin at L2 makes concurrent unions safe
__device__ __forceinline__ void unite(int *parent, int a, int b)
{
    for (;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang a under b
        const int old = atomicMin(parent + a, b);
        if (old == a) return;
        a = old;                                                // somebody re-parented a meanwhile: retry from there
    }
}

// fg[i] != 0 marks foreground.  parent = own index for fg, -1 for bg; stats cleared.
__global__ __launch_bounds__(256) void cc_init(const uint8_t *__restrict__ fg, int *parent, int *area, int *minx, int *miny,
                                               int *maxx, int *maxy, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    parent[i] = fg[i] ? (int)i : -1;
    area[i] = 0; minx[i] = 0x7FFFFFFF; miny[i] = 0x7FFFFFFF; maxx[i] = -1; maxy[i] = -1;
}

__global__ __launch_bounds__(256) void cc_merge(const uint8_t *__restrict__ fg, int *parent, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !fg[i]) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    if (x > 0 && fg[i - 1]) unite(parent, (int)i, (int)i - 1);
    if (y > 0) {
        if (fg[i - W]) unite(parent, (int)i, (int)i - W);
        if (x > 0 && fg[i - W - 1]) unite(parent, (int)i, (int)i - W - 1);
        if (x + 1 < W && fg[i - W + 1]) unite(parent, (int)i, (int)i - W + 1);
    }
}

__global__ __launch_bounds__(256) void cc_stats(int *parent, int *area, int *minx, int *miny, int *maxx, int *maxy, int H, int W,
                                                long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || parent[i] < 0) return;
    const int r = find_root(parent, (int)i);
    parent[i] = r;                                              // flatten (only this thread writes parent[i] in this kernel...
    const int hw = H * W;                                       // ... and a non-root's value is never used as a union target)
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    atomicAdd(area + r, 1);
    atomicMin(minx + r, x); atomicMax(maxx + r, x);
    atomicMin(miny + r, y); atomicMax(maxy + r, y);
}

__global__ __launch_bounds__(256) void k_inv(const uint8_t *__restrict__ labels, uint8_t *inv, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[i] = labels[i] == 2 ? 0 : 255;               // src/postprocess.cpp:18-22
}

// bin = 255 where the pixel is foreground after hole filling (src/postprocess.cpp:30-43, :57)
__global__ __launch_bounds__(256) void k_fill_bin(const uint8_t *__restrict__ labels, const int *__restrict__ parent,
                                                  const int *__restrict__ area, const int *__restrict__ minx,
                                                  const int *__restrict__ miny, const int *__restrict__ maxx,
                                                  const int *__restrict__ maxy, uint8_t *bin, int H, int W, int min_area, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool fgd = labels[i] == 2;
    const int r = parent[i];
    if (!fgd && r >= 0)
        fgd = minx[r] > 0 && miny[r] > 0 && maxx[r] < W - 1 && maxy[r] < H - 1 && area[r] < min_area;
    bin[i] = fgd ? 255 : 0;
}

template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph3(const uint8_t *__restrict__ src, uint8_t *dst, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    unsigned v = DILATE ? 0u : 255u;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;          // the border never constrains / never seeds
            const unsigned sv = src[i + dy * W + dx];
            v = DILATE ? max(v, sv) : min(v, sv);
        }
    dst[i] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_filter(const int *__restrict__ parent, const int *__restrict__ area, uint8_t *out,
                                                int min_area, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = parent[i];
    out[i] = (r >= 0 && area[r] >= min_area) ? 2 : 0;          // src/postprocess.cpp:70, :75-76
}

}  // namespace pp

size_t postprocess_workspace_bytes(int B, int H, int W)
{
    const size_t n = (size_t)B * H * W;
    return n * (6 * sizeof(int) + 3);                           // parent, area, 4 x bbox, three u8 planes
}

hipError_t launch_postprocess_masks(const uint8_t *labels_in, uint8_t *labels_out, int B, int H, int W, int min_area, void *ws,
                                    hipStream_t s)
{
    const long long n = (long long)B * H * W;
    if (n <= 0 || n > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    int *parent = static_cast<int *>(ws), *area = parent + n, *minx = area + n, *miny = minx + n, *maxx = miny + n, *maxy = maxx + n;
    uint8_t *u0 = reinterpret_cast<uint8_t *>(maxy + n), *u1 = u0 + n, *u2 = u1 + n;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    auto label = [&](const uint8_t *fg) {
        hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, fg, parent, area, minx, miny, maxx, maxy, n);
        hipLaunchKernelGGL(pp::cc_merge, g, b, 0, s, fg, parent, H, W, n);
        hipLaunchKernelGGL(pp::cc_stats, g, b, 0, s, parent, area, minx, miny, maxx, maxy, H, W, n);
    };
    hipLaunchKernelGGL(pp::k_inv, g, b, 0, s, labels_in, u0, n);
    label(u0);
    hipLaunchKernelGGL(pp::k_fill_bin, g, b, 0, s, labels_in, parent, area, minx, miny, maxx, maxy, u1, H, W, min_area, n);
    hipLaunchKernelGGL(pp::k_morph3<false>, g, b, 0, s, u1, u2, H, W, n);
    hipLaunchKernelGGL(pp::k_morph3<true>, g, b, 0, s, u2, u1, H, W, n);
    label(u1);
    hipLaunchKernelGGL(pp::k_filter, g, b, 0, s, parent, area, labels_out, min_area, n);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// extract_contours on the device.
// OpenCV's sequential Suzuki-Abe scan interleaves "find the next start pixel" with "follow that border".  For
// RETR_EXTERNAL both halves separate cleanly:
//   * the start pixel of a component's outer border is its first pixel in raster order = the ROOT of the union-find
//     labelling above (parents always point to smaller indices);
//   * the border is external iff the background region just above that pixel reaches the image frame (OpenCV's
//     `img0[lnbd] > 0` test says the same thing through the sign of the last border label on the row): the pixel above
//     the root is background by construction, and a 4-connected background component reaches the frame iff its
//     bounding box touches the image edge;
//   * the trace itself (first neighbour clockwise from west, then counter-clockwise from the arrival direction, a point
//     wherever the step direction changes) is inherently serial per contour, so every external contour gets its own
//     lane: contours and images run in parallel, the mask is L2-resident (256 KB per image);
//   * OpenCV returns contours newest-first = descending raster order of the start pixels: a 64-bit-free trick -- the
//     per-image list of external roots is sorted by one thread per image (a handful of entries after postprocess_mask).
namespace ct {

__global__ __launch_bounds__(256) void k_threshold(const uint8_t *__restrict__ mask, uint8_t *fg, uint8_t *bg, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool f = mask[i] > 127;                               // cv::threshold(127, 255, THRESH_BINARY)
    fg[i] = f ? 255 : 0;
    bg[i] = f ? 0 : 255;
}

// 4-connected union (background regions)
__global__ __launch_bounds__(256) void cc_merge4(const uint8_t *__restrict__ fg, int *parent, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !fg[i]) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    if (x > 0 && fg[i - 1]) pp::unite(parent, (int)i, (int)i - 1);
    if (y > 0 && fg[i - W]) pp::unite(parent, (int)i, (int)i - W);
}

__global__ __launch_bounds__(256) void k_zero_counts(int *counts, int B)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) counts[i] = 0;
}

// one entry per external component: its root (= start pixel).  fparent: flattened fg labelling; bparent + bbox: background.
__global__ __launch_bounds__(256) void k_collect(const int *__restrict__ fparent, const int *__restrict__ bparent,
                                                 const int *__restrict__ bminx, const int *__restrict__ bminy,
                                                 const int *__restrict__ bmaxx, const int *__restrict__ bmaxy, int *roots,
                                                 int *counts, int cap, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || fparent[i] != (int)i) return;                // roots only
    const int hw = H * W, img = (int)(i / hw);
    const int p = (int)(i % hw), y = p / W;
    bool external = (y == 0);
    if (!external) {
        const int r = bparent[i - W];                           // the pixel above a component's first pixel is background
        external = r >= 0 && (bminx[r] == 0 || bminy[r] == 0 || bmaxx[r] == W - 1 || bmaxy[r] == H - 1);
    }
    if (external) {
        const int slot = atomicAdd(counts + img, 1);
        if (slot < cap) roots[(size_t)img * cap + slot] = p;
    }
}

// per image: sort the start pixels descending (newest contour first).  Few entries: insertion sort by one lane.
__global__ __launch_bounds__(64) void k_sort_roots(int *roots, const int *counts, int cap, int B)
{
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= B) return;
    const int n = counts[img] < cap ? counts[img] : cap;
    int *r = roots + (size_t)img * cap;
    for (int a = 1; a < n; ++a) {
        const int v = r[a];
        int b = a - 1;
        while (b >= 0 && r[b] < v) { r[b + 1] = r[b]; --b; }
        r[b + 1] = v;
    }
}

// one lane per (image, contour): follow the border from its start pixel, emit CHAIN_APPROX_SIMPLE points.
// pass 0 counts points (npts), pass 1 writes them at the offsets computed in between.
__global__ __launch_bounds__(64) void k_trace(const uint8_t *__restrict__ fg, const int *__restrict__ roots,
                                              const int *__restrict__ counts, int cap, int H, int W, int B, int *npts,
                                              const int *__restrict__ offs, int *out_xy, int cap_points, int write)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    const int img = t / cap, c = t - img * cap;
    if (img >= B || c >= counts[img] || counts[img] > cap) return;
    const uint8_t *im = fg + (size_t)img * H * W;
    auto at = [&](int x, int y) -> bool { return x >= 0 && x < W && y >= 0 && y < H && im[(size_t)y * W + x] != 0; };
    const int DX[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };            // 0 = E, then counter-clockwise on screen (y grows down)
    const int DY[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };
    const int start = roots[(size_t)img * cap + c];
    const int x0 = start % W, y0 = start / W;
    int *dst = nullptr;
    int room = 0;
    if (write) {
        const int o = offs[(size_t)img * (cap + 1) + c];
        room = cap_points - o;
        dst = out_xy + ((size_t)img * cap_points + o) * 2;
    }
    int n = 0;
    auto emit = [&](int x, int y) {
        if (write && n < room) { dst[2 * n] = x; dst[2 * n + 1] = y; }
        ++n;
    };
    int dir = 4, first = -1;                                    // first neighbour: clockwise, starting after west
    for (int k = 0; k < 8; ++k) {
        dir = (dir + 7) & 7;
        if (at(x0 + DX[dir], y0 + DY[dir])) { first = dir; break; }
    }
    if (first < 0) {
        emit(x0, y0);                                           // isolated pixel
    } else {
        const int x1 = x0 + DX[first], y1 = y0 + DY[first];
        int cx = x0, cy = y0, came = first, last_step = first ^ 4;
        for (long long guard = 0; guard < 4LL * H * W + 16; ++guard) {      // a border has at most 4 visits per pixel
            int s = came, nx, ny;
            do { ++s; nx = cx + DX[s & 7]; ny = cy + DY[s & 7]; } while (!at(nx, ny));
            const int step = s & 7;
            if (step != last_step) { emit(cx, cy); last_step = step; }
            const bool closing = (nx == x0 && ny == y0 && cx == x1 && cy == y1);
            cx = nx; cy = ny;
            if (closing) break;
            came = (step + 4) & 7;
        }
    }
    if (!write) npts[(size_t)img * cap + c] = n;
}

// per image: exclusive scan of the point counts -> out_start, and the verdict (count or -1 on overflow)
__global__ __launch_bounds__(64) void k_offsets(const int *__restrict__ npts, const int *__restrict__ counts, int cap,
                                                int cap_points, int B, int *out_start, int *out_count)
{
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= B) return;
    int *st = out_start + (size_t)img * (cap + 1);
    const int nc = counts[img];
    if (nc > cap) { out_count[img] = -1; st[0] = 0; return; }
    int o = 0;
    for (int c = 0; c < nc; ++c) { st[c] = o; o += npts[(size_t)img * cap + c]; }
    st[nc] = o;
    out_count[img] = o > cap_points ? -1 : nc;
}

}  // namespace ct

__global__ __launch_bounds__(256) void mask_to_image_kernel(const uint8_t *__restrict__ labels, uint8_t *vis, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned v = labels[i];
    vis[i] = v == 1 ? 128 : v == 2 ? 255 : 0;
}

hipError_t launch_mask_to_image(const uint8_t *labels, uint8_t *vis, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(mask_to_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, vis, n);
    return hipGetLastError();
}

size_t contour_workspace_bytes(int B, int H, int W, int cap_contours)
{
    const size_t n = (size_t)B * H * W;
    return n * (7 * sizeof(int) + 2) + sizeof(int) * ((size_t)B * (2 * cap_contours + 1) + 64);
}

hipError_t launch_extract_contours(const uint8_t *masks, int B, int H, int W, int *out_xy, int cap_points, int *out_start,
                                   int cap_contours, int *out_count, void *ws, hipStream_t s)
{
    const long long n = (long long)B * H * W;
    if (n <= 0 || n > 0x7FFFFFFFLL || cap_contours <= 0 || cap_points <= 0) return hipErrorInvalidValue;
    int *fparent = static_cast<int *>(ws), *bparent = fparent + n, *area = bparent + n, *minx = area + n, *miny = minx + n,
        *maxx = miny + n, *maxy = maxx + n;
    uint8_t *fg = reinterpret_cast<uint8_t *>(maxy + n), *bg = fg + n;
    int *roots = reinterpret_cast<int *>(bg + n + ((16 - (2 * n) % 16) % 16));
    int *npts = roots + (size_t)B * cap_contours, *counts = npts + (size_t)B * cap_contours;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    hipLaunchKernelGGL(ct::k_threshold, g, b, 0, s, masks, fg, bg, n);
    // foreground labelling (8-connected); its stats are not needed, the arrays are reused by the background pass
    hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, fg, fparent, area, minx, miny, maxx, maxy, n);
    hipLaunchKernelGGL(pp::cc_merge, g, b, 0, s, fg, fparent, H, W, n);
    hipLaunchKernelGGL(pp::cc_stats, g, b, 0, s, fparent, area, minx, miny, maxx, maxy, H, W, n);
    // background labelling (4-connected) with bounding boxes
    hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, bg, bparent, area, minx, miny, maxx, maxy, n);
    hipLaunchKernelGGL(ct::cc_merge4, g, b, 0, s, bg, bparent, H, W, n);
    hipLaunchKernelGGL(pp::cc_stats, g, b, 0, s, bparent, area, minx, miny, maxx, maxy, H, W, n);
    hipLaunchKernelGGL(ct::k_zero_counts, dim3((B + 255) / 256), b, 0, s, counts, B);
    hipLaunchKernelGGL(ct::k_collect, g, b, 0, s, fparent, bparent, minx, miny, maxx, maxy, roots, counts, cap_contours, H, W, n);
    hipLaunchKernelGGL(ct::k_sort_roots, dim3((B + 63) / 64), dim3(64), 0, s, roots, counts, cap_contours, B);
    const dim3 gt((unsigned)(((long long)B * cap_contours + 63) / 64)), bt(64);
    hipLaunchKernelGGL(ct::k_trace, gt, bt, 0, s, fg, roots, counts, cap_contours, H, W, B, npts, out_start, out_xy, cap_points, 0);
    hipLaunchKernelGGL(ct::k_offsets, dim3((B + 63) / 64), dim3(64), 0, s, npts, counts, cap_contours, cap_points, B, out_start, out_count);
    hipLaunchKernelGGL(ct::k_trace, gt, bt, 0, s, fg, roots, counts, cap_contours, H, W, B, npts, out_start, out_xy, cap_points, 1);
    return hipGetLastError();
}

}  // namespace miunet
