// convt_lpr.hip -- the large transposed convolutions of the 16-bit pipelines with the weights resident in registers, gfx950 only.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"
#include "lpr_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// out[b][2y+dy][2x+dx][co] = bias[co] + sum_ci x[b][y][x][ci] * w[ci][co][dy][dx]: per pixel a [Cin] x [4 Cout] product with NO
// halo -- 4 Cout / (1 + 4 Cout / Cin) FLOP per byte, a quarter of the narrow 3x3 layers' intensity (conv_lpr.hip): the top
// two transposed convolutions of a network are pure HBM streams (Cin = 128 -> 4 x 64 at 256 x 256: 0.27 GB in, 0.54 GB out
// per batch of 16).  conv_mfma_bf16 with N = 4 Cout runs them at 2.4-3.4 TB/s (one tile per workgroup, weights restaged per
// tile, 2-byte stores).  The same scheme as conv_lpr.hip:
//   * a PERSISTENT workgroup of eight waves per CU; wave w owns ONE tap (dy, dx) = w & 3 and either half of the tile's
//     32-pixel row blocks (Cin <= 128: all Cout / 32 channel blocks of the tap) or half of the tap's channels (Cin = 256):
//     Cin / 16 x NBW B-fragments = 16, 64 or 128 weight registers, loaded once;
//   * LDS holds only input tiles: 32 KB each (256, 128 or 64 pixels x Cin), a ring of three filled by LDS-DMA loads two
//     tiles ahead, pieces permuted inside a pixel's 64 bytes for the 16-lane service groups of ds_read_b128 (lpr_common.h: lds_swz_row16);
//   * a row block's accumulators (NBW x 16 registers) are rounded once and leave through a wave-private LDS tile as 16-byte
//     stores: every output pixel of the tap gets its 64 or 128 contiguous bytes in one piece;
//   * one barrier per tile, waits counted as in conv_lpr.hip (a wave waits for ITS loads of tile n + 1 after the MFMAs of
//     tile n and before its stores).
// v_mfma_f32_16x16x32, one MFMA per 32-channel chunk as in every 16-bit kernel: the same products and fp32 accumulation chain as
// conv_mfma_bf16<TAPS = 1>: bit-identical outputs
// (tests/test_gpu_bf16.py::test_convT_resident_weights).
//
// CIN = 64 / 128 / 256, NBW = 32-channel blocks per wave, COSPLIT = waves w >> 2 split the tap's channels (else its row blocks)
template <typename T, int CIN, int NBW, bool COSPLIT>
__global__ __launch_bounds__(512, 1) void convT2x2_lpr(const ConvArgs a, const int tiles_x, const int tiles_y, const int ntiles)
{
    typedef typename LprVec<T>::x8 x8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int PLANES = CIN / 32, NB16 = 2 * NBW;
    constexpr int TILE_BYTES = 32 * 1024, TILE_LOADS = 32;    // every tile is 32 KB: 32 wave-wide loads, four per wave
    constexpr int M = TILE_BYTES / (CIN * 2);                  // pixels per tile: 256 / 128 / 64
    constexpr int TR = M / 32;                                 // image rows per tile (row block = 1 row x 32 columns)
    constexpr int PLANE_BYTES = M * 64, PLANE_LOADS = M / 16;
    constexpr int NBUF = 3, LEAD = 2;
    constexpr int MBW = COSPLIT ? TR : TR / 2;                 // row blocks per wave
    constexpr int PIECES = 4 * NBW;                            // 16-byte pieces per output pixel of this wave
    constexpr int TROW = 32 * NBW + 8;                         // 16-bit elements per pixel of the output scratch
    constexpr int SCR_BYTES = 32 * TROW * 2;
    static_assert(MBW >= 1 && TILE_LOADS == PLANES * PLANE_LOADS, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;                 // v_mfma_f32_16x16x32 lane roles (lpr_common.h)
    const int kidx = wave & 3, half = wave >> 2;               // tap dy = kidx >> 1, dx = kidx & 1
    const int dy = kidx >> 1, dx = kidx & 1;
    const int mb0 = COSPLIT ? 0 : half * MBW;                  // first row block
    const int co0 = COSPLIT ? half * 32 * NBW : 0;             // first channel of the tap
    T *const Ts = reinterpret_cast<T *>(smem + NBUF * TILE_BYTES + wave * SCR_BYTES);

    // ---- weights: [Cin / 32][N = 4 Cout (n = kidx Cout + co)][32]; lane (i16, kq) holds w[32 c + 8 kq .. + 8][n]
    const T *const wpk = reinterpret_cast<const T *>(a.wpk);
    x8 wreg[PLANES][NB16];
    float bias[NB16];
#pragma unroll
    for (int j = 0; j < NB16; ++j) {
        const int co = co0 + 16 * j + i16;
        bias[j] = a.bias[co];
#pragma unroll
        for (int c = 0; c < PLANES; ++c)
            wreg[c][j] = *reinterpret_cast<const x8 *>(wpk + ((size_t)c * a.CoutPad + (size_t)kidx * a.Cout + co) * KC_BF16 + 8 * kq);
    }

    // ---- A fragments: pixel 32 mb + 16 h + i16 of the tile (row block mb = one image row x 32 columns, column half h), piece kq
    // in slot kq ^ lds_swz_row16(column) (lpr_common.h; the same for every mb and h)
    const unsigned aoff = (unsigned)(i16 * 64 + ((kq ^ lds_swz_row16(i16)) << 4));

    // ---- this wave's four patch loads: load i = wave + 8 k = pixels 16 (i % PLANE_LOADS) .. + 16 of plane i / PLANE_LOADS
    unsigned dvoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = wave + 8 * k, c = i / PLANE_LOADS, j = i - c * PLANE_LOADS;
        const int p = 16 * j + (lane >> 2), q = (lane & 3) ^ lds_swz_row16(p & 31);
        dvoff[k] = (unsigned)((((p >> 5) * a.W + (p & 31)) * a.ldc + 32 * c + 8 * q) * 2);
    }

    // ---- output pieces: piece e = lane + 64 it of the [32 pixels][PIECES] tile of a row block; pixel m -> column 2 m + dx
    const int OW = 2 * a.W;
    constexpr int OITERS = 32 * PIECES / 64;
    unsigned ovoff[OITERS];
#pragma unroll
    for (int it = 0; it < OITERS; ++it) {
        const int e = lane + 64 * it, m = e / PIECES, q = e - m * PIECES;
        ovoff[it] = (unsigned)((((dy * OW) + 2 * m + dx) * a.ldo + a.co_off + co0 + 8 * q) * 2);
    }
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;

    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slots = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q_ = ntiles >> 3, r_ = ntiles & 7;
    const int t_start = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    const int t_count = q_ + (xcd < r_ ? 1 : 0);
    const int nt = slot < t_count ? (t_count - slot + slots - 1) / slots : 0;
    const T *const in = reinterpret_cast<const T *>(a.in);

    auto issue_dma = [&](const int n) {
        int L = t_start + slot + n * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * TR, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(in + (size_t)b * a.H * a.W * a.ldc), 0, a.H * a.W * a.ldc * 2, 0x00020000);
        char *const dst = smem + (n % NBUF) * TILE_BYTES;
        const unsigned org = (unsigned)(((y0 * a.W + x0) * a.ldc) * 2);
        const bool full = y0 + TR <= a.H && x0 + 32 <= a.W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned voff = dvoff[k];
            if (!full) {                      // a tile over the image edge: pixels past it read zeros through the range check
                const int i = wave + 8 * k, j = i % PLANE_LOADS, p = 16 * j + (lane >> 2);
                if (y0 + (p >> 5) >= a.H || x0 + (p & 31) >= a.W) voff = 0xFFFFFFFFu;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + (wave + 8 * k) * 1024), 16, voff, org, 0, 0);
        }
    };

    if (nt == 0) return;
#pragma unroll
    for (int n = 0; n < LEAD; ++n)
        if (n < nt) issue_dma(n);
    lpr_wait_vm_n(nt > 1 ? 4 : 0);            // tile 0 has landed when at most tile 1's four loads are outstanding

    for (int n = 0; n < nt; ++n) {
        __syncthreads();                      // tile n is complete in LDS; tile n - 1 is consumed
        if (n + LEAD < nt) issue_dma(n + LEAD);
        const unsigned base = (unsigned)((n % NBUF) * TILE_BYTES);
        int L = t_start + slot + n * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * TR, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<T *>(a.out) + (size_t)b * 4 * a.H * a.W * a.ldo, 0, 4 * a.H * a.W * a.ldo * 2, 0x00020000);
        const bool edge = y0 + TR > a.H || x0 + 32 > a.W;
#pragma unroll
        for (int mi = 0; mi < MBW; ++mi) {
            const int mb = mb0 + mi;
            f32x4 acc[2][NB16];                   // [column half][16-channel block]; first written by plane 0
            x8 af[2][2];                          // the fragments of plane c + 1 are requested before the MFMAs of plane c (conv_lpr.hip)
            auto read_plane = [&](const int c, x8 (&dst)[2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    dst[h] = *reinterpret_cast<const x8 *>(smem + (base + aoff) + c * PLANE_BYTES + mb * 2048 + h * 1024);
            };
            read_plane(0, af[0]);
#pragma unroll
            for (int c = 0; c < PLANES; ++c) {
                if (c + 1 < PLANES) read_plane(c + 1, af[(c + 1) & 1]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < NB16; ++j) {
                        if (c == 0) mfma16_lpr_first(acc[h][j], af[c & 1][h], wreg[c][j]);
                        else mfma16_lpr(acc[h][j], af[c & 1][h], wreg[c][j]);
                    }
            }
            mfma16_drain();                       // (lpr_common.h: the stores below read the accumulators with no barrier in between)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NB16; ++j) mfma16_settled(acc[h][j]);
            // this wave's loads of tile n + 1: older than the four it issued for tile n + 2, and waited for before ANY store of
            // tile n (a store issued first would sit in front of them in the count)
            if (mi == 0) lpr_wait_vm_n(n + 2 < nt ? 4 : 0);
            // register r of block (h, j) = pixel column 16 h + 4 kq + r of row y0 + mb, lane = channel i16 of the 16-channel block j
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NB16; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Ts[(16 * h + 4 * kq + r) * TROW + 16 * j + i16] = (T)fmaxf(acc[h][j][r] + bias[j], relu_lo);
            const unsigned osoff = (unsigned)(((2 * (y0 + mb) * OW + 2 * x0) * a.ldo) * 2);
#pragma unroll
            for (int it = 0; it < OITERS; ++it) {
                const int e = lane + 64 * it, m = e / PIECES, q = e - m * PIECES;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + m * TROW + 8 * q);
                unsigned voff = ovoff[it];
                if (edge && !(y0 + mb < a.H && x0 + m < a.W)) voff = 0xFFFFFFFFu;
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, voff, osoff, LP_ST_AUX);
                wide_store_guard();
            }
        }
    }
}

template <typename T, int CIN, int NBW, bool COSPLIT>
static hipError_t launch_convt_lpr_cfg(const ConvArgs &a, hipStream_t s)
{
    constexpr int TR = 32 * 1024 / (CIN * 2) / 32;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + TR - 1) / TR;
    const int ntiles = tiles_x * tiles_y * a.B;
    const int cus = routing_of(a).cus;
    const int grid = ntiles < cus ? ntiles : cus;
    constexpr size_t lds = 3 * 32 * 1024 + 8 * (size_t)(32 * (32 * NBW + 8) * 2);
    static_assert(lds <= 160 * 1024, "LDS of one CU");
    auto kern = convT2x2_lpr<T, CIN, NBW, COSPLIT>;
    if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}

// the shapes with at most 128 weight registers per wave: Cin -> Cout = 64 -> 32, 128 -> 64, 256 -> 128 (the three largest
// transposed convolutions of a base-32 or base-64 network); 16-bit output
static bool convt_lpr_shape_ok(const ConvArgs &a)
{
    if (a.wpk == nullptr || !a.out_lp || a.head_w != nullptr || a.pool_out != nullptr) return false;
    if (!((a.Cin == 64 && a.Cout == 32) || (a.Cin == 128 && a.Cout == 64) || (a.Cin == 256 && a.Cout == 128))) return false;
    if (a.ldc % 8 || a.ldo % 8 || a.co_off % 8 || a.CoutPad < 4 * a.Cout) return false;
    return (long long)a.H * a.W * a.ldc * 2 < (1ll << 31) && 4ll * a.H * a.W * a.ldo * 2 < (1ll << 31);
}

// MIUNET_CONVT_LPR = 0: never; 1 (default): those shapes when the tiles fill the chip four times over; 2: whatever the grid
bool convT2x2_lpr_takes(const ConvArgs &a)
{
    const Routing rt = routing_of(a);
    const int mode = rt.convt_lpr;
    if (mode == 0 || !convt_lpr_shape_ok(a)) return false;
    const int tr = 32 * 1024 / (a.Cin * 2) / 32;
    const long long ntiles = (long long)((a.W + 31) / 32) * ((a.H + tr - 1) / tr) * a.B;
    return mode == 2 || ntiles >= 4 * rt.cus;
}

template <typename T>
static hipError_t launch_convt_lpr(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin == 64) return launch_convt_lpr_cfg<T, 64, 1, false>(a, s);
    if (a.Cin == 128) return launch_convt_lpr_cfg<T, 128, 2, false>(a, s);
    return launch_convt_lpr_cfg<T, 256, 2, true>(a, s);
}

hipError_t launch_convT2x2_lpr(const ConvArgs &a, bool fp16, hipStream_t s)
{
    if (!convt_lpr_shape_ok(a)) return hipErrorInvalidValue;
    return fp16 ? launch_convt_lpr<_Float16>(a, s) : launch_convt_lpr<__bf16>(a, s);
}

}  // namespace miunet
