// conv_direct.hip -- direct implicit-GEMM convolution (3x3 and the 2x2 transposed conv) on the fp32 MFMA, gfx950 only.
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
#include "kernel_common.h"

#include <cstdlib>

namespace miunet {

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
    auto kern = conv_mfma_f32<TAPS, TH, BN, KCT, NFAST>;
    if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
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
#ifdef MIUNET_EXPERIMENTS                              // lab build only: the product library has one route per shape
    static const int cfg = [] { const char *e = getenv("MIUNET_CONVT_CFG"); return e ? atoi(e) : 1; }();
#else
    constexpr int cfg = 1;
#endif
    switch (cfg) {
    case 0: return launch_conv_cfg<1, 8, 64, 16, false>(a, s);
    case 1: return launch_conv_cfg<1, 8, 64, 16, true>(a, s);
    case 2: return launch_conv_cfg<1, 8, 128, 16, true>(a, s);
    case 4: return launch_conv_cfg<1, 16, 64, 16, true>(a, s);
    default: return launch_conv_cfg<1, 8, 64, 32, true>(a, s);
    }
}


}  // namespace miunet
