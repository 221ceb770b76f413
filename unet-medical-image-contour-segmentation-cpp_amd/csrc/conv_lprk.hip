// conv_lprk.hip -- the 128 -> 64 channel 3x3 layer of the 16-bit pipelines (`up4.c1` of BASELINE configs 3 and 5: the first
// convolution behind the top-level concat) with the weights resident in registers and K SPLIT OVER A WAVE PAIR, gfx950 only.
#include <cstdlib>

#include "kernel_common.h"
#include "lpr_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// conv3x3_lpr (conv_lpr.hip) keeps a layer's weights in registers and streams input patches through an LDS ring; it stops
// at Cin = 64 because a 32-channel block of a 128 -> 64 layer is 9 taps x 8 k-steps x 4 registers = 288 registers per lane.
// Here the reduction is cut in two: of a wave pair, wave kh = 0 holds the weights of input channels 0..63 and wave kh = 1
// those of 64..127 (144 registers each) for the SAME 32 output channels and the SAME 2 x 32 pixels; each runs half the
// MFMAs of the block and the two partial fp32 sums meet in LDS.  The exchange is symmetric -- a wave hands its partner the
// partial block of one 16-column half and finishes the other half itself (shift, ReLU, one rounding, 16-byte stores) -- so
// both waves of a pair do the same amount of epilogue work.
//   * persistent workgroup of eight waves per CU = 2 row pairs x 2 channel blocks x 2 K halves; a tile is 4 rows x 32 columns
//     (a 128-channel patch of 6 x 34 pixels is 52 KB: a ring of two fits the CU next to the exchange buffer; 8-row tiles
//     would need 174 KB);
//   * LDS: 2 x 52 KB patch ring (LDS-DMA loads, pieces permuted as in conv_lpr.hip: lpr_common.h) + 32 KB exchange + 20 KB output scratch;
//   * two barriers per tile (patch complete / partial sums published).
// Arithmetic (v_mfma_f32_16x16x32, one MFMA per tap and 32-channel chunk as in every 16-bit kernel): the same products as
// conv_mfma_bf16, fp32 accumulation inside each K half in its order, then ONE extra fp32
// add of the two halves -- (c0 + c1) + (c2 + c3) instead of ((c0 + c1) + c2) + c3: not bit-identical to the 2 x 2 kernel
// (the other resident-weight kernels are), identical to fp32 re-association noise before the single 16-bit rounding
// (tests/test_gpu_bf16.py::test_conv3x3_resident_weights_k_split, tests/test_gpu_insitu.py).
struct LPRK {
    static constexpr int CIN = 128, COUT = 64, TH = 4, PW = 34, PH = TH + 2, NPIX = PW * PH;
    static constexpr int PLANES = CIN / 32;
    static constexpr int PLANE_LOADS = (NPIX + 15) / 16;          // 13 wave-wide LDS-DMA loads per 32-channel plane
    static constexpr int PLANE_BYTES = PLANE_LOADS * 1024;
    static constexpr int TILE_BYTES = PLANES * PLANE_BYTES;       // 52 KB
    static constexpr int TILE_LOADS = PLANES * PLANE_LOADS;       // 52
    static constexpr int DMA_ITERS = (TILE_LOADS + 7) / 8;
    static constexpr int NBUF = 2;
    static constexpr int X_BYTES = 16 * 64 * 4;                   // per wave: one 32x32 fp32 accumulator block
    static constexpr int TROW = 40;                               // 16-bit elements per pixel of the output tile (32 + 8 pad)
    static constexpr int SCR_BYTES = 32 * TROW * 2;
    static constexpr size_t LDS_BYTES = (size_t)NBUF * TILE_BYTES + 8 * X_BYTES + 8 * SCR_BYTES;
};

template <typename T>
__global__ __launch_bounds__(512, 1) void conv3x3_lprk(const ConvArgs a, const int tiles_x, const int tiles_y, const int ntiles)
{
    typedef typename LprVec<T>::x8 x8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int TROW = LPRK::TROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;                // v_mfma_f32_16x16x32 lane roles (lpr_common.h)
    const int kh = wave & 1;                                  // K half: input channels 64 kh .. + 64 (patch planes 2 kh, 2 kh + 1)
    const int blk = (wave >> 1) & 1;                          // 32-channel output block
    const int rp = wave >> 2;                                 // row pair of the tile: image rows y0 + 2 rp, + 1
    char *const Xs = smem + LPRK::NBUF * LPRK::TILE_BYTES;    // [8 waves][16 registers][64 lanes] fp32
    T *const Ts = reinterpret_cast<T *>(smem + LPRK::NBUF * LPRK::TILE_BYTES + 8 * LPRK::X_BYTES + wave * LPRK::SCR_BYTES);

    // ---- this wave's half of the block's weights, as MFMA B fragments: lane (i16, kq) holds
    // w[tap][64 kh + 32 c + 8 kq .. + 8][32 blk + 16 jb + i16]
    const T *const wpk = reinterpret_cast<const T *>(a.wpk);
    x8 wreg[9][2][2];
    float bias[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const int n = 32 * blk + 16 * jb + i16;
        bias[jb] = a.bias[n];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                wreg[tap][c][jb] = *reinterpret_cast<const x8 *>(wpk + ((size_t)((2 * kh + c) * 9 + tap) * a.CoutPad + n) * KC_BF16 + 8 * kq);
    }

    // ---- per-lane LDS byte offsets of the A fragments inside a plane: pixel (row 2 rp + mr, column i16) displaced by the tap,
    // piece kq in slot kq ^ lds_swz_row16(column) (first 16-column half; the second is + 1024)
    unsigned aoff[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr)
            aoff[tap][mr] = (unsigned)(((2 * rp + mr + dy) * LPRK::PW + i16 + dx) * 64 + ((kq ^ lds_swz_row16(i16 + dx)) << 4));
    }

    // ---- per-lane global byte offsets of this wave's patch loads, relative to the patch origin (y0 - 1, x0 - 1)
    unsigned dvoff[LPRK::DMA_ITERS];
#pragma unroll
    for (int k = 0; k < LPRK::DMA_ITERS; ++k) {
        const int i = wave + 8 * k, c = i / LPRK::PLANE_LOADS, j = i - c * LPRK::PLANE_LOADS;
        const int p = 16 * j + (lane >> 2);
        const int py = p / LPRK::PW, px = p - py * LPRK::PW;
        const int q = (lane & 3) ^ lds_swz_row16(px);
        dvoff[k] = (i < LPRK::TILE_LOADS && p < LPRK::NPIX) ? (unsigned)(((py * a.W + px) * a.ldc + 32 * c + 8 * q) * 2) : 0xFFFFFFFFu;
    }

    // ---- per-lane byte offsets of the 16-byte output pieces of THIS wave's finished half (column half kh), relative to the tile origin
    unsigned ovoff[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = lane + 64 * it, m = e >> 2, q = e & 3;
        ovoff[it] = (unsigned)((((2 * rp + (m >> 4)) * a.W + 16 * kh + (m & 15)) * a.ldo + a.co_off + 32 * blk + 8 * q) * 2);
    }
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;

    // ---- this workgroup's tiles: its XCD's logical range, walked with a stride (conv_lpr.hip)
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slots = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q_ = ntiles >> 3, r_ = ntiles & 7;
    const int t_start = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    const int t_count = q_ + (xcd < r_ ? 1 : 0);
    const int nt = slot < t_count ? (t_count - slot + slots - 1) / slots : 0;
    const T *const in = reinterpret_cast<const T *>(a.in);

    auto issue_dma = [&](const int t) {       // the patch of this workgroup's t-th tile -> ring slot t % NBUF
        int L = t_start + slot + t * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * LPRK::TH, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(in + (size_t)b * a.H * a.W * a.ldc), 0, a.H * a.W * a.ldc * 2, 0x00020000);
        char *const dst = smem + (t % LPRK::NBUF) * LPRK::TILE_BYTES;
        if (y0 >= 1 && y0 + LPRK::TH + 1 <= a.H && x0 >= 1 && x0 + 33 <= a.W) {
            const unsigned org = (unsigned)((((y0 - 1) * a.W + x0 - 1) * a.ldc) * 2);
#pragma unroll
            for (int k = 0; k < LPRK::DMA_ITERS; ++k)
                if (wave + 8 * k < LPRK::TILE_LOADS) {
                    const unsigned voff = dvoff[k];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + (wave + 8 * k) * 1024), 16, voff, org, 0, 0);
                }
        } else {                              // a tile on the image border: zero padding through the range check
#pragma unroll
            for (int k = 0; k < LPRK::DMA_ITERS; ++k) {
                const int i = wave + 8 * k, c = i / LPRK::PLANE_LOADS, j = i - c * LPRK::PLANE_LOADS;
                const int p = 16 * j + (lane >> 2);
                const int py = p / LPRK::PW, px = p - py * LPRK::PW;
                const int q = (lane & 3) ^ lds_swz_row16(px);
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool inb = p < LPRK::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned voff = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 32 * c + 8 * q) * 2) : 0xFFFFFFFFu;
                if (i < LPRK::TILE_LOADS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + i * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    if (nt == 0) return;
    issue_dma(0);
    lpr_wait_vm<0>();

    for (int t = 0; t < nt; ++t) {
        __syncthreads();                      // tile t is complete in LDS; tile t - 1's patch and exchange slots are consumed
        if (t + 1 < nt) issue_dma(t + 1);     // ... into the slot tile t - 1 just left

        // ---- tile t: this wave's K half -- 9 taps x 4 k-steps for both 16-column halves of its row pair
        const unsigned base = (unsigned)((t % LPRK::NBUF) * LPRK::TILE_BYTES + 2 * kh * LPRK::PLANE_BYTES);
        f32x4 acc[2][2][2];                       // [column half][row of the pair][16-channel block]; first written by step 0
        // 144 weight + 32 accumulator registers leave no room for a second fragment set: each fragment is refilled for step t + 1
        // right after its two MFMAs of step t (six MFMAs = 96 cycles ahead of its next use; the SIMD's other wave covers the rest)
        x8 af[2][2];
        auto read_frag = [&](const int t, const int mb, const int mr) {
            const int c = t / 9, tap = t - 9 * c;
            return *reinterpret_cast<const x8 *>(smem + (base + aoff[tap][mr]) + c * LPRK::PLANE_BYTES + mb * 1024);
        };
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr) af[mb][mr] = read_frag(0, mb, mr);
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int c = t / 9, tap = t - 9 * c;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        if (t == 0) mfma16_lpr_first(acc[mb][mr][jb], af[mb][mr], wreg[tap][c][jb]);
                        else mfma16_lpr(acc[mb][mr][jb], af[mb][mr], wreg[tap][c][jb]);
                    }
                    if (t + 1 < 18) af[mb][mr] = read_frag(t + 1, mb, mr);
                }
        }
        mfma16_drain();                           // (lpr_common.h: the exchange below reads the accumulators with no barrier in between)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) mfma16_settled(acc[mb][mr][jb]);

        // ---- exchange: hand the partner (the other K half of the same block) the half it finishes, keep column half kh
        {
            f32x4 *const xw = reinterpret_cast<f32x4 *>(Xs + wave * LPRK::X_BYTES) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) xw[64 * q] = acc[1 - kh][q >> 1][q & 1];
        }
        lpr_wait_vm<0>();                     // this wave's share of tile t + 1's patch has landed (and tile t - 1's stores are out)
        __syncthreads();                      // partial sums published
        f32x4 sum[2][2];
        {
            const f32x4 *const xr = reinterpret_cast<const f32x4 *>(Xs + (wave ^ 1) * LPRK::X_BYTES) + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = xr[64 * q];
                const f32x4 mine = acc[kh][q >> 1][q & 1];
                // channels 0..63 first, then 64..127: the order a single chain would have added them in
                sum[q >> 1][q & 1] = kh == 0 ? mine + v : v + mine;
            }
        }

        // ---- epilogue of column half kh: + shift, ReLU, one rounding to 16 bits, [pixel][channel] tile in LDS, 16-byte stores
        int L = t_start + slot + t * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * LPRK::TH, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<T *>(a.out) + (size_t)b * a.H * a.W * a.ldo, 0, a.H * a.W * a.ldo * 2, 0x00020000);
        const unsigned osoff = (unsigned)(((y0 * a.W + x0) * a.ldo) * 2);
        const bool edge = y0 + LPRK::TH > a.H || x0 + 32 > a.W;
        // register r of block (row mr, 16-channel block jb) = pixel m = 16 mr + 4 kq + r of the row block, channel 16 jb + i16
#pragma unroll
        for (int mr = 0; mr < 2; ++mr)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ts[(16 * mr + 4 * kq + r) * TROW + 16 * jb + i16] = (T)fmaxf(sum[mr][jb][r] + bias[jb], relu_lo);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = lane + 64 * it, m = e >> 2, q = e & 3;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + m * TROW + 8 * q);
            unsigned voff = ovoff[it];
            if (edge && !(y0 + 2 * rp + (m >> 4) < a.H && x0 + 16 * kh + (m & 15) < a.W)) voff = 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, voff, osoff, LP_ST_AUX);
            wide_store_guard();
        }
    }
}

static bool lprk_shape_ok(const ConvArgs &a)
{
    if (a.wpk == nullptr || !a.out_lp || a.head_w != nullptr || a.pool_out != nullptr) return false;
    if (a.Cin != LPRK::CIN || a.Cout != LPRK::COUT) return false;
    if (a.ldc % 8 || a.ldo % 8 || a.co_off % 8 || a.CoutPad < a.Cout) return false;
    return (long long)a.H * a.W * a.ldc * 2 < (1ll << 31) && (long long)a.H * a.W * a.ldo * 2 < (1ll << 31);
}

// MIUNET_LPRK (Routing::lprk) = 0: never; 1 (default): 128 -> 64, 16-bit output, no pooling, when the tiles fill the chip four
// times over; 2: whatever the grid (parity tests on small inputs)
bool conv3x3_lprk_takes(const ConvArgs &a)
{
    const Routing rt = routing_of(a);
    if (rt.lprk == 0 || !lprk_shape_ok(a)) return false;
    const long long ntiles = (long long)((a.W + 31) / 32) * ((a.H + LPRK::TH - 1) / LPRK::TH) * a.B;
    return rt.lprk == 2 || ntiles >= 4 * rt.cus;
}

template <typename T>
static hipError_t launch_lprk(const ConvArgs &a, hipStream_t s)
{
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + LPRK::TH - 1) / LPRK::TH;
    const int ntiles = tiles_x * tiles_y * a.B;
    const int cus = routing_of(a).cus;
    const int grid = ntiles < cus ? ntiles : cus;
    static_assert(LPRK::LDS_BYTES <= 160 * 1024, "LDS of one CU");
    auto kern = conv3x3_lprk<T>;
    if (hipError_t e = ensure_dynamic_lds(kern, LPRK::LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LPRK::LDS_BYTES, s, a, tiles_x, tiles_y, ntiles);
    return hipGetLastError();
}

hipError_t launch_conv3x3_lprk(const ConvArgs &a, bool fp16, hipStream_t s)
{
    if (!lprk_shape_ok(a)) return hipErrorInvalidValue;
    return fp16 ? launch_lprk<_Float16>(a, s) : launch_lprk<__bf16>(a, s);
}

}  // namespace miunet
