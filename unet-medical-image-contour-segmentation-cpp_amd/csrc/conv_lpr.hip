// conv_lpr.hip -- the narrow 16-bit 3x3 layers (Cin, Cout <= 64) with the weights resident in registers, gfx950 only.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"
#include "lpr_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// The top levels of the 16-bit pipelines (512 x 512 x 64 in BASELINE config 3, 1024 x 1024 x 32 and 512 x 512 x 64 in
// config 5) sit at the ridge of the roofline: 64 -> 64 channels is 288 FLOP per byte moved, 32 -> 32 half of that, against
// 1.89 PFLOP/s / 6.3 TB/s = 300 (what this card SUSTAINS: a register-only bf16 MFMA loop, an HBM copy -- the practical ceilings
// this design argues against; every REPORTED fraction is priced at the nominal 2.5 PFLOP/s / 8 TB/s of bench.py).  conv_mfma_bf16 (conv_lp.hip) runs them at 40-48 % of either roof: one tile per workgroup,
// and every tile stages the layer's WHOLE weight set into LDS again (73 KB for 64 -> 64: more bytes than its input patch),
// waits for its patch with nothing else to do, and reads one LDS fragment per MFMA.  Here:
//   * one PERSISTENT workgroup of eight waves per CU walks its XCD's share of the 8 x 32-pixel tiles;
//   * the weights never touch LDS: a layer's 9 x Cin x Cout 16-bit weights are 9 x Cin/16 MFMA B-fragments of four
//     registers per 32 output channels -- 72 registers for 32 -> 32, 144 for 64 -> 32, 32 -> 64, and for 64 -> 64 when a wave
//     keeps one 32-channel block (its partner wave the other).  Loaded once per workgroup, they stay for every tile;
//   * LDS holds only input patches: a ring of NBUF (8+2) x (32+2)-pixel patches, 64 bytes per pixel and 32-channel plane,
//     filled by LDS-DMA loads (buffer_load ... lds, 16 bytes per lane, no staging registers) LEAD = NBUF - 1 tiles ahead of
//     the MFMAs -- one to two patches (22-87 KB) in flight per CU at any time, which is what 1/256 of the HBM bandwidth
//     needs at its latency.  An LDS-DMA load places lane l's 16 bytes at base + 16 l, so rows cannot be padded; instead
//     the four 16-byte pieces of a pixel are permuted inside its 64 bytes so that every 16-lane service group of a
//     ds_read_b128 covers all sixteen 16-byte bank groups (lpr_common.h: lds_swz_row16), and the permutation costs nothing --
//     it is the per-lane global offset of the load;
//   * the MFMA is v_mfma_f32_16x16x32 (lpr_common.h: under dense 16-bit MFMA work the chip holds a higher clock on this shape
//     than on 32x32x16): a pixel block is one image row x 16 columns, lane (i16 = lane & 15, kq = lane >> 4) supplies the
//     pixel's / the output channel's input channels 8 kq .. + 8 of a 32-channel plane, and holds output columns 4 kq .. + 4 of
//     channel i16.  A wave's unit of work is still a ROW BLOCK of 2 image rows x 16 columns (two pixel blocks): every 2 x 2
//     output block sits in ONE lane (registers r, r + 1 of both rows), so the fused max pooling stays in-lane;
//   * one barrier per tile.  A wave waits for ITS loads of tile n+1 after the MFMAs of tile n and before its stores
//     (vmcnt <= the loads it issued for the tiles after n+1: loads return in order, so that bound holds whatever the
//     stores of tile n-1 do), the barrier at the head of the next tile publishes them;
//   * outputs leave through a wave-private LDS tile as 16-byte stores (the conv_lp.hip epilogue, without its barrier).
// Same products and the same fp32 accumulation chain as the other 16x16x32 kernels (chunks of 32 channels in order, taps in
// raster order, one MFMA per tap and chunk): bit-identical results (tests/test_gpu_bf16.py::test_conv3x3_resident_weights).
// RB = row blocks (2 image rows x 16 columns each) a wave owns per tile: the tile is 8 RB rows x 32 columns.  RB = 2 halves
// the halo re-read (18 x 34 pixels for 16 x 32 outputs: 1.20 x instead of 1.33 x), the barriers and the per-tile address
// work per pixel, and gives a wave two independent accumulator chains; it needs a 39 KB patch per 32-channel plane, so it is
// for the Cin = 32 layers.
template <int RB>
struct LprGeom {
    static constexpr int TH = 8 * RB, PW = 34, PH = TH + 2, NPIX = PW * PH;
    static constexpr int PLANE_LOADS = (NPIX + 15) / 16;          // wave-wide LDS-DMA loads per 32-channel plane: 16 pixels each
    static constexpr int PLANE_BYTES = PLANE_LOADS * 1024;
    static constexpr int RB_BYTES = 8 * PW * 64;                  // the next row block of the same wave: 8 patch rows further
};
// FIRST (the network's first layer computed in this kernel's loader, ConvArgs::first_img; C0 = channels of the u8 image): a window of
// (PH + 2) x 36 normalised image pixels per tile in LDS as [pixel][C0] floats (conv3x3_first_mfma's layout, layers_mem.hip), and the
// layer itself on v_mfma_f32_32x32x2_f32 exactly as that kernel runs it -- same K order, same accumulation: the same bits.  A unit
// of work is 32 patch pixels x 32 channels: the first 32 columns of one patch row, or (the last units) 32 pixels of columns 32, 33.
template <int RB, int CIN, int C0>
struct LprFirst {
    using GEO = LprGeom<RB>;
    static constexpr int K = 9 * C0, KS = (K + 1) / 2;            // K steps of two
    static constexpr int WP = GEO::PW + 2;                        // window pitch
    static constexpr int WIN_PIX = (GEO::PH + 2) * WP;
    static constexpr int WIN_FLOATS = WIN_PIX * C0 + 4;           // + zeros for the padded K step
    static constexpr int WIN_BYTES = (WIN_FLOATS * 4 + 15) & ~15;
    static constexpr int FILL_ITERS = (GEO::PH + 2 + 3) / 4;      // window rows in fours: thread = (row tid >> 7 of the four, byte tid & 127 of its 36 C0)
    static constexpr int EDGE_BLOCKS = (2 * GEO::PH + 31) / 32;   // units of the columns 32, 33
    static constexpr int UNITS = GEO::PH + EDGE_BLOCKS;           // (CIN = 32: one 32-channel block); wave w takes units w, w + 8, w + 16
    static_assert(UNITS <= 24, "three units per wave");
    static constexpr int BYTES = 2 * WIN_BYTES + 1024 + KS * 256 + FILL_ITERS * 2048;      // two windows, the /255 table, the B fragments [KS][64 lanes], the byte staging
};
struct LPR {
    static constexpr int TROW = 40;                               // 16-bit elements per pixel of the output tile (32 + 8 pad)
    static constexpr int SCR_BYTES = (32 + 8) * TROW * 2;         // per wave: [32 pixels][TROW] + pooled [8][TROW]
    static constexpr int HEAD_ROW = 32 + 4;                       // floats per pixel of the fused head's tile (conflict-free b128 rows)
    static constexpr int HEAD_SCR_BYTES = 32 * HEAD_ROW * 4;      // per wave: [32 pixels][HEAD_ROW] fp32
    static constexpr size_t lds_bytes(int plane_bytes, int cin, int nbuf, bool head, int first_bytes = 0)
    {
        return (size_t)nbuf * (cin / 32) * plane_bytes + 8 * (head ? HEAD_SCR_BYTES : SCR_BYTES) + (head ? 3 * 32 * 4 : 0) + first_bytes;
    }
};

// CIN = 32 or 64 input channels (exactly), NBT = 1 or 2 blocks of 32 output channels (Cout = 32 NBT exactly), NBUF = patch ring
// HEAD (Cout = 32, at most three classes): the layer feeds the network's fp32 1x1 head + argmax (ConvArgs::head_w).  The
// post-ReLU fp32 row block crosses the wave's LDS scratch, lane = pixel, the head's weights stay in registers (96), the sums
// run in the order of conv_mfma_bf16's fused head (four interleaved partial sums per class, folded at the end): the same
// logits bit for bit.  `out` is never written.
// FIRST = C0 (1 or 3): the input tensor is the network's first layer, computed here from the u8 image (LprFirst above): instead of
// the LDS-DMA of tile n + LEAD, the workgroup computes the patch of tile n + 1 BETWEEN the MFMA steps of tile n -- conv3x3_first_mfma's
// arithmetic (fp32 MFMA over k = tap C0 + c, + shift, ReLU, one rounding to 16 bits), zero outside the image: the same bits as the
// stand-alone first layer -- and stores it where the DMA would have.  The layer is bound by the matrix pipe either way (the first
// layer's fp32 MFMAs are twice the conv's 16-bit ones in pipe time); interleaved, the dependent fp32 chain of a unit runs in the
// shadow of the conv's independent MFMAs.  Built as a separate phase behind the epilogue it cost more than the launch it replaced
// (same card, config 5: 0.51 ms against 0.21 + 0.27; timing-only builds: the phase 0.25 ms, of which stores 0.05).  Cin = 32 only:
// the 64 -> 64 layer's 144 weight registers leave no room for a second accumulator set (256 registers per wave at 8 waves per CU).
// The first layer's activation tensor (0.5 GB at 1024^2 x 32 x 8 images) is neither written nor read back.
template <typename T, int CIN, int NBT, int NBUF, int RB, bool HEAD = false, int FIRST = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_lpr(const ConvArgs a, const int tiles_x, const int tiles_y, const int ntiles, const int exp_arg)
{
#ifdef MIUNET_EXPERIMENTS
    const int exp = exp_arg;                  // timing-only switches of the lab build (MIUNET_LPR_EXP)
#else
    constexpr int exp = 0;                    // the product library has one behaviour
    (void)exp_arg;
#endif
    typedef typename LprVec<T>::x8 x8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) void *lds_ptr;
    static_assert((CIN == 32 || CIN == 64) && (NBT == 1 || NBT == 2) && (NBUF >= 3 || FIRST != 0), "narrow layers only");
    static_assert(!HEAD || NBT == 1, "the fused head needs every channel of a pixel in one wave");
    static_assert(RB == 1 || CIN == 32, "two row blocks per wave: one 32-channel plane only (immediate LDS offsets, LDS size)");
    static_assert(FIRST == 0 || ((FIRST == 1 || FIRST == 3) && !HEAD && NBUF == 2 && CIN == 32), "fused first layer: one or three image channels, two patch slots, 32 channels");
    using GEO = LprGeom<RB>;
    using FG = LprFirst<RB, CIN, FIRST ? FIRST : 1>;
    constexpr int PLANES = CIN / 32;
    constexpr bool SPLITN = CIN * NBT > 64;                   // 64 -> 64: a wave keeps ONE 32-channel block (144 weight registers) ...
    constexpr int MB = SPLITN ? 2 : 1;                        // ... for both column halves of its row pair; otherwise one row block,
    constexpr int NB = SPLITN ? 1 : NBT;                      // every channel block
    constexpr int TILE_BYTES = PLANES * GEO::PLANE_BYTES;
    constexpr int TILE_LOADS = PLANES * GEO::PLANE_LOADS;
    constexpr int DMA_ITERS = (TILE_LOADS + 7) / 8;
    constexpr int LEAD = NBUF - 1;
    constexpr int TROW = LPR::TROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int rp = wave >> 1;                                 // row pair of the tile: image rows y0 + 2 rp, + 1
    const int ch0 = SPLITN ? 0 : (wave & 1);                  // first 16-column half
    const int blk0 = SPLITN ? (wave & 1) : 0;                 // first 32-channel block
    T *const Ts = reinterpret_cast<T *>(smem + NBUF * TILE_BYTES + wave * (HEAD ? LPR::HEAD_SCR_BYTES : LPR::SCR_BYTES));
    T *const Ps = Ts + 32 * TROW;
    float *const Ys = reinterpret_cast<float *>(Ts);          // HEAD: [32 pixels][HEAD_ROW] fp32

    // ---- the layer's weights, as MFMA B fragments: lane (i16, kq) holds w[tap][32 c + 8 kq .. + 8][32 blk0 + 16 jb + i16]
    constexpr int NB16 = 2 * NB;                              // 16-channel blocks of this wave
    const T *const wpk = reinterpret_cast<const T *>(a.wpk);
    x8 wreg[9][PLANES][NB16];
    float bias[NB16];
#pragma unroll
    for (int jb = 0; jb < NB16; ++jb) {
        const int n = 32 * blk0 + 16 * jb + i16;
        bias[jb] = a.bias[n];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < PLANES; ++c)
                wreg[tap][c][jb] = *reinterpret_cast<const x8 *>(wpk + ((size_t)(c * 9 + tap) * a.CoutPad + n) * KC_BF16 + 8 * kq);
    }

    // ---- per-lane LDS byte offsets of the A fragments inside a plane: pixel (row 2 rp + mr, column 16 ch0 + i16) of the
    // tile displaced by the tap, 16-byte piece kq in slot kq ^ lds_swz_row16(column) (lpr_common.h; the second row block of a
    // wave is 8 rows down and the second column half 16 columns right: the same slots)
    // (FIRST: the register file is full -- three column offsets, the row displacement goes into the reads' immediate offsets)
    unsigned aoff[9][2], acol[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pcol = 16 * ch0 + i16 + dx;
        acol[dx] = (unsigned)((2 * rp * GEO::PW + pcol) * 64 + ((kq ^ lds_swz_row16(pcol)) << 4));
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
            const int prow = 2 * rp + mr + dy, pcol = 16 * ch0 + i16 + dx;
            aoff[tap][mr] = (unsigned)((prow * GEO::PW + pcol) * 64 + ((kq ^ lds_swz_row16(pcol)) << 4));
        }
    }

    // ---- per-lane global byte offsets of this wave's patch loads, relative to the patch origin (y0 - 1, x0 - 1): load i =
    // wave + 8 k covers pixels 16 (i % 22) .. + 16 of plane i / 22, lane l = (pixel l >> 2, slot l & 3)
    unsigned dvoff[DMA_ITERS];
#pragma unroll
    for (int k = 0; k < (FIRST != 0 ? 0 : DMA_ITERS); ++k) {
        const int i = wave + 8 * k, c = i / GEO::PLANE_LOADS, j = i - c * GEO::PLANE_LOADS;
        const int p = 16 * j + (lane >> 2);
        const int py = p / GEO::PW, px = p - py * GEO::PW;
        const int q = (lane & 3) ^ lds_swz_row16(px);
        dvoff[k] = (i < TILE_LOADS && p < GEO::NPIX) ? (unsigned)(((py * a.W + px) * a.ldc + 32 * c + 8 * q) * 2) : 0xFFFFFFFFu;
    }
    const int my_loads = (TILE_LOADS - wave + 7) / 8;        // loads this wave issues per tile

    // ---- per-lane byte offsets of the 16-byte output pieces, relative to the tile origin: piece e = lane + 64 it of the
    // [32 pixels][4 pieces] tile of row block mb, channel block j
    unsigned ovoff[MB][NB][2], pvoff[MB][NB];
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const bool do_pool = a.pool_out != nullptr;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = lane + 64 * it, m = e >> 2, q = e & 3;
                ovoff[mb][j][it] = (unsigned)((((2 * rp + (m >> 4)) * a.W + 16 * (ch0 + mb) + (m & 15)) * a.ldo + a.co_off + 32 * (blk0 + j) + 8 * q) * 2);
            }
            const int m = lane >> 2, q = lane & 3;           // pooled: row rp, columns 8 (ch0 + mb) + m, lanes 0..31
            pvoff[mb][j] = lane < 32 ? (unsigned)(((rp * Wp + 8 * (ch0 + mb) + m) * a.pool_ld + 32 * (blk0 + j) + 8 * q) * 2) : 0xFFFFFFFFu;
        }
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    // HEAD: head_w[k][c] (classes past head_classes = 0) goes through LDS once.  Read straight from global its address is
    // uniform, so hipcc keeps the 96 values in SGPRs, spills them into VGPR lanes and restores each with a v_readlane per use
    // (149 per tile); read back from LDS they are ordinary per-lane registers.
    f32x4 wh[HEAD ? 3 : 1][8];
    float hb[3] = { 0.f, 0.f, 0.f };
    if constexpr (HEAD) {
        float *const WhL = reinterpret_cast<float *>(smem + NBUF * TILE_BYTES + 8 * LPR::HEAD_SCR_BYTES);      // [3][32]
        if (tid < 3 * 32) WhL[tid] = tid / 32 < a.head_classes ? a.head_w[tid] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < a.head_classes) hb[k] = a.head_b[k];
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) wh[k][c4] = *reinterpret_cast<const f32x4 *>(WhL + k * 32 + 4 * c4);
        }
    }

    // ---- this workgroup's tiles: its XCD's logical range (blocks b and b + 8 share an XCD), walked with a stride, so the
    // 32 CUs of an XCD work on neighbouring tiles at any time and share their halos through that XCD's L2
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int slots = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    const int q_ = ntiles >> 3, r_ = ntiles & 7;
    const int t_start = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    const int t_count = q_ + (xcd < r_ ? 1 : 0);
    const int nt = slot < t_count ? (t_count - slot + slots - 1) / slots : 0;
    const T *const in = reinterpret_cast<const T *>(a.in);

    auto issue_dma = [&](const int n) {       // the patch of this workgroup's n-th tile -> ring slot n % NBUF
        int L = t_start + slot + n * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * GEO::TH, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(in + (size_t)b * a.H * a.W * a.ldc), 0, a.H * a.W * a.ldc * 2, 0x00020000);
        char *const dst = smem + (n % NBUF) * TILE_BYTES;
        if (y0 >= 1 && y0 + GEO::TH + 1 <= a.H && x0 >= 1 && x0 + 33 <= a.W) {
            // interior: the precomputed offsets, the patch origin in the scalar offset (dead lanes keep 0xFFFFFFFF: the range
            // check looks at the vector offset only and returns zeros)
            const unsigned org = (unsigned)((((y0 - 1) * a.W + x0 - 1) * a.ldc) * 2);
#pragma unroll
            for (int k = 0; k < DMA_ITERS; ++k)
                if (wave + 8 * k < TILE_LOADS) {
                    const unsigned voff = dvoff[k];       // (a prvalue: hipcc's host pass silently drops the kernel when the array element is passed directly)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + (wave + 8 * k) * 1024), 16, voff, org, 0, 0);
                }
        } else {                              // a tile on the image border: zero padding through the range check
#pragma unroll
            for (int k = 0; k < DMA_ITERS; ++k) {
                const int i = wave + 8 * k, c = i / GEO::PLANE_LOADS, j = i - c * GEO::PLANE_LOADS;
                const int p = 16 * j + (lane >> 2);
                const int py = p / GEO::PW, px = p - py * GEO::PW;
                const int q = (lane & 3) ^ lds_swz_row16(px);
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool inb = p < GEO::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned voff = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 32 * c + 8 * q) * 2) : 0xFFFFFFFFu;
                if (i < TILE_LOADS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + i * 1024), 16, voff, 0, 0, 0);
            }
        }
    };

    if (nt == 0) return;

    // ---- FIRST: the fused first layer (see LprFirst and the kernel's header)
    constexpr int C0 = FIRST ? FIRST : 1;
    char *const WinL = smem + LPR::lds_bytes(GEO::PLANE_BYTES, CIN, NBUF, HEAD);
    float *const LutL = reinterpret_cast<float *>(WinL + 2 * FG::WIN_BYTES);
    const int li = lane & 31, lh = lane >> 5;
    float *const FbwL = LutL + 256;                                            // B fragments: [s][lane] = w[2 s + lh][li] (LDS: the register file is full)
    float fsh = 0.f;                                                           // shift of channel li
    // window offset of k = tap C0 + c: the 3 C0 values of a kernel row are consecutive floats of the window.  A lane's k is 2 s + lh:
    // its offset is that of k = 2 s plus lh -- folded into the lane's unit base -- except where 2 s + 1 starts the next kernel row
    // (+ lhx more); the padded step past K reads a finite neighbour (its weight is 0).  So a read is base + an immediate.
    auto fkoff = [](const int k) constexpr { return (k / (3 * C0)) * FG::WP * C0 + k % (3 * C0); };
    const int lhx = lh * (FG::WP * C0 - 3 * C0);
    auto tile_of = [&](const int n, int &b, int &y0, int &x0) {
        int L = t_start + slot + n * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        b = L / tiles_y; y0 = ty * GEO::TH; x0 = tx * 32;
    };
    // the window's rows are runs of 36 C0 bytes of the image: four rows per pass, thread = (row, byte).  The bytes come by LDS-DMA
    // (one byte per lane into a dword slot of the staging area: no registers, and a whole MFMA loop between request and use); outside
    // the image the range check returns the byte 0, which the /255 table maps to 0.0f: the first layer's zero padding.  Every thread
    // converts the slots its own wave requested, so the wave's own vmcnt wait is all the ordering the staging area needs.
    char *const StageL = reinterpret_cast<char *>(FbwL + FG::KS * 64);
    const int wr0 = tid >> 7, wbx = tid & 127, wgx = wbx / C0;
    auto win_request = [&](const int n) {
        int b, y0, x0;
        tile_of(n, b, y0, x0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.first_img) + (size_t)b * a.H * a.W * C0, 0, a.H * a.W * C0, 0x00020000);
        const int gx = x0 - 2 + wgx;
        const bool xok = wbx < FG::WP * C0 && (unsigned)gx < (unsigned)a.W;
#pragma unroll
        for (int it = 0; it < FG::FILL_ITERS; ++it) {
            const int gy = y0 - 2 + 4 * it + wr0;
            const bool ok = xok && (unsigned)gy < (unsigned)a.H && 4 * it + wr0 < GEO::PH + 2;
            const unsigned voff = ok ? (unsigned)((gy * a.W + x0 - 2) * C0 + wbx) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(StageL + it * 2048 + wave * 256), 1, voff, 0, 0, 0);
        }
    };
    auto win_commit = [&](const int buf) {    // (after this wave's vmcnt wait) ... through the /255 table into window `buf`
        float *const w = reinterpret_cast<float *>(WinL + buf * FG::WIN_BYTES) + wr0 * FG::WP * C0 + wbx;
#pragma unroll
        for (int it = 0; it < FG::FILL_ITERS; ++it)
            if (wbx < FG::WP * C0 && 4 * it + wr0 < GEO::PH + 2) w[4 * it * FG::WP * C0] = LutL[*reinterpret_cast<const unsigned *>(StageL + it * 2048 + tid * 4) & 255u];
        if (tid < 4) reinterpret_cast<float *>(WinL + buf * FG::WIN_BYTES)[FG::WIN_PIX * C0 + tid] = 0.f;
    };
    // ---- the units of this wave: un = 0, 1 are the patch rows `wave` and `wave + 8`; un = 2 is row wave + 16, or (past the
    // rows) the block eb = wave + 16 - PH of the columns 32, 33, or nothing.  Accumulator register r of a unit is its pixel
    // c_r + 4 lh, c_r = (r & 3) + 8 (r >> 2): a row unit's column (slot swizzle = the lane constant 2 lh), an edge block's pixel
    // (row 16 eb + 2 lh + (c_r >> 1), column 32 + (c_r & 1): swizzle 0) -- a store address is a lane constant plus an immediate.
    const bool has3 = wave + 16 < FG::UNITS, edge3 = wave + 16 >= GEO::PH;      // (uniform)
    const int eb3 = wave + 16 - GEO::PH;
    const int ee = 32 * eb3 + li;
    const int upb_c[3] = { (wave * FG::WP + li) * C0 + lh, ((wave + 8) * FG::WP + li) * C0 + lh,
                           (edge3 ? (((ee >> 1) < GEO::PH ? (ee >> 1) : GEO::PH - 1) * FG::WP + 32 + (ee & 1)) * C0 : ((wave + 16) * FG::WP + li) * C0) + lh };
    const unsigned frow0 = (unsigned)((li & 7) * 2 + 4 * lh * 64 + (((li >> 3) ^ (2 * lh)) << 4));
    const unsigned ust_c[3] = { (unsigned)(wave * GEO::PW * 64) + frow0, (unsigned)((wave + 8) * GEO::PW * 64) + frow0,
                                edge3 ? (unsigned)((li & 7) * 2 + ((li >> 3) << 4) + ((16 * eb3 + 2 * lh) * GEO::PW + 32) * 64)
                                      : (unsigned)((wave + 16) * GEO::PW * 64) + frow0 };
    int fy0 = 0, fx0 = 0;                     // the tile whose patch is being computed: origin,
    bool finterior = true;                    // no patch pixel outside the image,
    const float *fwin = nullptr;              // window,
    char *fdst = nullptr;                     // patch slot
    f32x16 facc;
    // border tiles: bit r of fcolm = column c_r + 4 lh of a row unit is inside the image (the row itself is a uniform test), bit r
    // of fedgm = pixel r of this wave's edge block is
    unsigned fcolm = 0xFFFFu, fedgm = 0xFFFFu;
    auto first_setup = [&](const int n, const int into) {
        int b;
        tile_of(n, b, fy0, fx0);
        finterior = fy0 >= 1 && fy0 + GEO::PH - 1 <= a.H && fx0 >= 1 && fx0 + 33 <= a.W;
        fwin = reinterpret_cast<const float *>(WinL + (n & 1) * FG::WIN_BYTES);
        fdst = smem + (into % NBUF) * TILE_BYTES;
        fcolm = 0xFFFFu; fedgm = 0xFFFFu;
        if (!finterior) {
            unsigned cm = 0, em = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);
                cm |= (unsigned)(fx0 - 1 + cr + 4 * lh) < (unsigned)a.W ? (1u << r) : 0u;
                em |= ((unsigned)(fy0 - 1 + 16 * eb3 + 2 * lh + (cr >> 1)) < (unsigned)a.H && (unsigned)(fx0 + 31 + (cr & 1)) < (unsigned)a.W) ? (1u << r) : 0u;
            }
            fcolm = cm; fedgm = em;
        }
    };
    auto unit_store = [&](const int un) {     // + shift, ReLU, one rounding; zero outside the image
        char *const d = fdst + ust_c[un];
        const unsigned mk = (un == 2 && edge3) ? fedgm : ((unsigned)(fy0 - 1 + wave + 8 * un) < (unsigned)a.H ? fcolm : 0u);
        if (!(un == 2 && edge3)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = facc[r] + fsh;
                *reinterpret_cast<T *>(d + ((r & 3) + 8 * (r >> 2)) * 64) = (mk >> r & 1u) ? (T)(v > 0.f ? v : 0.f) : (T)0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2);
                const float v = facc[r] + fsh;
                if (16 * eb3 + 2 * lh + (cr >> 1) < GEO::PH)
                    *reinterpret_cast<T *>(d + ((cr >> 1) * GEO::PW + (cr & 1)) * 64) = (mk >> r & 1u) ? (T)(v > 0.f ? v : 0.f) : (T)0.f;
            }
        }
    };
    // K step q % KS of unit q / KS (q is a constant after unrolling): the operands are READ one conv step ahead of the MFMA that
    // takes them (LDS reads do not move across volatile asms, so a read next to its MFMA would expose the LDS latency 42 times per tile)
    constexpr int MS = (3 * FG::KS + 9 * PLANES - 1) / (9 * PLANES);        // first-layer K steps per conv step
    float fav[MS] = {}, fbv[MS] = {};
    auto first_read = [&](const int q, const int i) {
        const int un = q / FG::KS, s_ = q - un * FG::KS;
        if (un == 2 && !has3) return;
        const bool cross = (2 * s_ + 1) % (3 * C0) == 0 && 2 * s_ + 1 < FG::K;
        fav[i] = fwin[upb_c[un] + fkoff(2 * s_) + (cross ? lhx : 0)];
        fbv[i] = FbwL[s_ * 64 + lane];
    };
    auto first_micro = [&](const int q, const int i) {
        const int un = q / FG::KS, s_ = q - un * FG::KS;
        if (un == 2 && !has3) return;
        if (s_ == 0) mfma32_f32_first(facc, fav[i], fbv[i]);
        else mfma32_f32(facc, fav[i], fbv[i]);
        if (s_ == FG::KS - 1 && !(exp & 8)) { mfma32_settle(facc); unit_store(un); }      // (exp & 8, lab build: no stores -- timing only)
    };
    if constexpr (FIRST != 0) {
        for (int e = tid; e < FG::KS * 64; e += 512) {
            const int k = 2 * (e >> 6) + ((e >> 5) & 1);
            FbwL[e] = k < FG::K ? a.first_w[(size_t)k * CIN + (e & 31)] : 0.f;
        }
        fsh = a.first_shift[li];
        if (tid < 256) LutL[tid] = a.first_lut[tid];
        win_request(0);
        __syncthreads();                      // the table
        lpr_wait_vm<0>(); asm volatile("" ::: "memory");
        win_commit(0);
        if (nt > 1) { win_request(1); lpr_wait_vm<0>(); asm volatile("" ::: "memory"); win_commit(1); }
        __syncthreads();                      // both windows
        first_setup(0, 0);                    // the patch of tile 0; every later one is computed under the MFMAs of the tile before
#pragma unroll
        for (int q = 0; q < 3 * FG::KS; ++q) { first_read(q, 0); first_micro(q, 0); }
    } else {
#pragma unroll
        for (int n = 0; n < LEAD; ++n)
            if (n < nt) issue_dma(n);
        // tile 0 has landed when at most the loads of tiles 1 .. LEAD - 1 are outstanding
        int later = 0;
#pragma unroll
        for (int n = 1; n < LEAD; ++n) later += (n < nt) ? my_loads : 0;
        lpr_wait_vm_n(later);
    }

    for (int n = 0; n < nt; ++n) {
        __syncthreads();                      // tile n is complete in LDS (every wave waited for its share); tile n - 1 is consumed
        // (exp: timing-only switches, MIUNET_LPR_EXP, results WRONG -- 1 = no patch DMA after the ring's first fill, 2 = no output stores)
        if constexpr (FIRST != 0) {
            // window n + 1 is complete, patch slot (n + 1) % 2 is free: the patch of tile n + 1 is computed between the MFMA steps below;
            // the bytes of window n + 2 are requested now (LDS-DMA) and converted behind the MFMAs, into the window tile n's patch
            // was computed from
            if (n + 2 < nt && !(exp & 16)) win_request(n + 2);
            first_setup(n + 1 < nt ? n + 1 : n, n + 1);       // (past the last tile: that tile once more, into the free slot -- no branch in the steps)
        } else {
            if (n + LEAD < nt && !(exp & 1)) issue_dma(n + LEAD);      // ... into the slot tile n - 1 just left
        }

        // ---- tile n: 9 taps x Cin / 16 MFMAs per row block and channel block
        const unsigned base = (unsigned)((n % NBUF) * TILE_BYTES);
        f32x4 acc[RB][MB][2][NB16];               // [row block][column half][row of the pair][16-channel block]; first written by tap 0
        // the MFMAs are volatile asms (lpr_common.h) and LDS reads do not move across those: the software pipeline is written
        // out -- the fragments of step t + 1 are requested before the MFMAs of step t (a step = one tap of one 32-channel plane)
        x8 af[2][RB][MB][2];
        auto read_step = [&](const int t, x8 (&dst)[RB][MB][2]) {
            const int c = t / 9, tap = t - 9 * c;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int mr = 0; mr < 2; ++mr)
                        dst[rb][mb][mr] = FIRST != 0
                            ? *reinterpret_cast<const x8 *>(smem + (base + acol[tap % 3]) + (mr + tap / 3) * GEO::PW * 64 + c * GEO::PLANE_BYTES + rb * GEO::RB_BYTES + mb * 1024)
                            : *reinterpret_cast<const x8 *>(smem + (base + aoff[tap][mr]) + c * GEO::PLANE_BYTES + rb * GEO::RB_BYTES + mb * 1024);
        };
        read_step(0, af[0]);
#pragma unroll
        for (int t = 0; t < 9 * PLANES; ++t) {
            const int c = t / 9, tap = t - 9 * c;
            if constexpr (FIRST != 0) {           // the operands of this step's share of the next tile's patch
                if (!(exp & 4)) {
#pragma unroll
                    for (int i = 0; i < MS; ++i)
                        if (t * MS + i < 3 * FG::KS && !(exp & 32)) first_read(t * MS + i, i);      // (exp & 32, lab build: stale operands -- timing only)
                }
            }
            if (t + 1 < 9 * PLANES) read_step(t + 1, af[(t + 1) & 1]);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                        for (int jb = 0; jb < NB16; ++jb) {
                            if (t == 0) mfma16_lpr_first(acc[rb][mb][mr][jb], af[t & 1][rb][mb][mr], wreg[tap][c][jb]);
                            else mfma16_lpr(acc[rb][mb][mr][jb], af[t & 1][rb][mb][mr], wreg[tap][c][jb]);
                        }
            if constexpr (FIRST != 0) {           // this step's share of the next tile's patch (exp & 4, lab build: none -- timing only)
                if (!(exp & 4)) {
#pragma unroll
                    for (int i = 0; i < MS; ++i)
                        if (t * MS + i < 3 * FG::KS) first_micro(t * MS + i, i);
                }
            }
        }
        mfma16_drain();                           // (lpr_common.h: the epilogue below reads the accumulators with no barrier in between)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                    for (int jb = 0; jb < NB16; ++jb) mfma16_settled(acc[rb][mb][mr][jb]);

        if constexpr (FIRST != 0) {
            if (n + 2 < nt && !(exp & 16)) { lpr_wait_vm<0>(); asm volatile("" ::: "memory"); win_commit(n & 1); }      // (the DMA wrote LDS behind the compiler's back)
        } else {
            // ---- this wave's loads of tile n + 1 (older than everything it issued for tiles n + 2 .. n + LEAD)
            int later = 0;
#pragma unroll
            for (int d = 2; d <= LEAD; ++d) later += (n + d < nt) ? my_loads : 0;
            lpr_wait_vm_n(later);
        }

        // ---- epilogue: + shift, ReLU, one rounding to 16 bits, [pixel][channel] tile in this wave's LDS scratch, 16-byte stores
        int L = t_start + slot + n * slots;
        const int tx = L % tiles_x; L /= tiles_x;
        const int ty = L % tiles_y;
        const int b = L / tiles_y;
        const int y0 = ty * GEO::TH, x0 = tx * 32;
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<T *>(a.out) + (size_t)b * a.H * a.W * a.ldo, 0, a.H * a.W * a.ldo * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            do_pool ? reinterpret_cast<T *>(a.pool_out) + (size_t)b * Hp * Wp * a.pool_ld : reinterpret_cast<T *>(a.out), 0,
            do_pool ? Hp * Wp * a.pool_ld * 2 : 0, 0x00020000);
        const unsigned osoff = (unsigned)(((y0 * a.W + x0) * a.ldo) * 2);
        const unsigned psoff = (unsigned)((((y0 >> 1) * Wp + (x0 >> 1)) * a.pool_ld) * 2);
        const bool edge = y0 + GEO::TH > a.H || x0 + 32 > a.W;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int yb = y0 + 8 * rb;                   // first image row of this row block's group of four row pairs
                // register r of block (row mr, 16-channel block jl of this 32-channel group) = pixel m = 16 mr + 4 kq + r of the
                // row block (row m >> 4, column m & 15), channel 16 jl + i16
                if constexpr (HEAD) {
#pragma unroll
                    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                        for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                Ys[(16 * mr + 4 * kq + r) * LPR::HEAD_ROW + 16 * jl + i16] = fmaxf(acc[rb][mb][mr][2 * j + jl][r] + bias[2 * j + jl], relu_lo);
                    // lane = pixel (lanes 32..63 repeat 0..31 and store nothing); first maximum wins (src/process.cpp:158-170)
                    const int m = lane & 31;
                    f32x4 d4[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) d4[k] = f32x4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const f32x4 yv = *reinterpret_cast<const f32x4 *>(Ys + m * LPR::HEAD_ROW + 4 * c4);
#pragma unroll
                        for (int k = 0; k < 3; ++k) d4[k] += yv * wh[k][c4];
                    }
                    const int py = yb + 2 * rp + (m >> 4), px = x0 + 16 * (ch0 + mb) + (m & 15);
                    if (lane < 32 && py < a.H && px < a.W) {
                        const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
                        float best = -3.402823466e+38f;
                        int idx = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            if (k < a.head_classes) {
                                const float d = ((d4[k].x + d4[k].y) + (d4[k].z + d4[k].w)) + hb[k];
                                if (a.head_logits != nullptr) a.head_logits[((size_t)b * a.head_classes + k) * hw + pin] = d;
                                if (d > best) { best = d; idx = k; }
                            }
                        }
                        a.head_labels[(size_t)b * hw + pin] = (uint8_t)idx;
                    }
                    continue;
                }
#pragma unroll
                for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Ts[(16 * mr + 4 * kq + r) * TROW + 16 * jl + i16] = (T)fmaxf(acc[rb][mb][mr][2 * j + jl][r] + bias[2 * j + jl], relu_lo);
                if (do_pool) {                // the 2 x 2 block of registers (r, r + 1) of both rows, r even: pooled column 2 kq + (r >> 1)
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const f32x4 &u = acc[rb][mb][0][2 * j + jl], &v = acc[rb][mb][1][2 * j + jl];
                            const float mx = fmaxf(fmaxf(u[r], u[r + 1]), fmaxf(v[r], v[r + 1]));
                            Ps[(2 * kq + (r >> 1)) * TROW + 16 * jl + i16] = (T)fmaxf(mx + bias[2 * j + jl], relu_lo);
                        }
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int e = lane + 64 * it, m = e >> 2, q = e & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(Ts + m * TROW + 8 * q);
                    unsigned voff = ovoff[mb][j][it];
                    if ((edge && !(yb + 2 * rp + (m >> 4) < a.H && x0 + 16 * (ch0 + mb) + (m & 15) < a.W)) || (exp & 2)) voff = 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, voff, osoff + (unsigned)(8 * rb * a.W * a.ldo * 2), LP_ST_AUX);
                    wide_store_guard();
                }
                if (do_pool) {
                    const int m = (lane >> 2) & 7, q = lane & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(Ps + m * TROW + 8 * q);
                    unsigned voff = pvoff[mb][j];
                    if (edge && !(yb + 2 * rp + 1 < a.H && x0 + 16 * (ch0 + mb) + 2 * m + 1 < a.W)) voff = 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b128(v, pool_rsrc, voff, psoff + (unsigned)(4 * rb * Wp * a.pool_ld * 2), LP_ST_AUX);
                    wide_store_guard();
                }
            }
    }
}

template <typename T, int CIN, int NBT, int NBUF, int RB, bool HEAD = false, int FIRST = 0>
static hipError_t launch_lpr_cfg(const ConvArgs &a, hipStream_t s)
{
    using GEO = LprGeom<RB>;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + GEO::TH - 1) / GEO::TH;
    const int ntiles = tiles_x * tiles_y * a.B;
    const int cus = routing_of(a).cus;
    const int grid = ntiles < cus ? ntiles : cus;
    constexpr size_t lds = LPR::lds_bytes(GEO::PLANE_BYTES, CIN, NBUF, HEAD, FIRST ? LprFirst<RB, CIN, FIRST ? FIRST : 1>::BYTES : 0);
    static_assert(lds <= 160 * 1024, "LDS of one CU");
    auto kern = conv3x3_lpr<T, CIN, NBT, NBUF, RB, HEAD, FIRST>;
    if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
#ifdef MIUNET_EXPERIMENTS                              // lab build only (libmiunet_exp.so, tools/dev/ab*.sh): never in libmiunet.so
    static const int exp = [] { const char *e = getenv("MIUNET_LPR_EXP"); return e ? atoi(e) : 0; }();          // timing-only switches (see the kernel)
#else
    constexpr int exp = 0;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, tiles_x, tiles_y, ntiles, exp);
    return hipGetLastError();
}

// The fused first layer (ConvArgs::first_img): 32 -> 32 channels behind a three-channel image (BASELINE config 5's inc.c2);
// `first_cin` = channels of the image.  Not 64 -> 64: see the kernel's header.
bool conv3x3_lpr_can_fuse_first(const ConvArgs &a, int first_cin)
{
    if (a.head_w != nullptr || !a.out_lp) return false;
    return first_cin == 3 && a.Cin == 32 && a.Cout == 32;
}

static bool lpr_shape_ok(const ConvArgs &a)
{
    if (a.wpk == nullptr) return false;
    if (a.head_w != nullptr) {                // fused head: 32 -> 32 channels, at most three classes, fp32 tile (never stored), no pooling
        if (a.Cin != 32 || a.Cout != 32 || a.head_classes < 1 || a.head_classes > 3 || a.out_lp || a.pool_out != nullptr || a.head_labels == nullptr ||
            a.head_b == nullptr)
            return false;
    } else if (!a.out_lp) {
        return false;
    }
    if ((a.Cin != 32 && a.Cin != 64) || (a.Cout != 32 && a.Cout != 64)) return false;
    if (a.ldc % 8 || a.ldo % 8 || a.co_off % 8 || a.CoutPad < a.Cout) return false;
    if (a.pool_out != nullptr && (a.pool_ld % 8 || (a.H & 1) || (a.W & 1))) return false;
    // 32-bit byte offsets inside one image
    return (long long)a.H * a.W * a.ldc * 2 < (1ll << 31) && (long long)a.H * a.W * a.ldo * 2 < (1ll << 31);
}

// MIUNET_LPR = 0: never; 1 (default): the shapes above when the tiles fill the chip four times over; 2: whatever the grid
// (parity tests on small inputs)
bool conv3x3_lpr_takes(const ConvArgs &a)
{
    const Routing rt = routing_of(a);
    const int mode = rt.lpr;
    if (mode == 0 || !lpr_shape_ok(a)) return false;
    const long long ntiles = (long long)((a.W + 31) / 32) * ((a.H + 15) / 16) * a.B;     // 16-row tiles (Cin = 32); twice as many of 8 rows
    return mode == 2 || ntiles >= 4 * rt.cus;
}

template <typename T>
static hipError_t launch_lpr(const ConvArgs &a, hipStream_t s)
{
    const int rb_env = routing_of(a).lpr_rb;                // MIUNET_LPR_RB = 1: 8-row tiles for every shape (A/B, parity tests)
    // the fused head keeps 8-row tiles: with two row blocks its 96 head-weight registers spill, and scratch traffic shares
    // vmcnt with the patch DMA (the compiler's waits for it drain the ring: measured 0.27 -> 0.53 ms)
    if (a.first_img != nullptr)               // (the caller asked conv3x3_lpr_can_fuse_first)
        return (a.Cin == 32 && a.Cout == 32 && a.first_cin == 3) ? launch_lpr_cfg<T, 32, 1, 2, 2, false, 3>(a, s) : hipErrorInvalidValue;
    if (a.head_w != nullptr) return launch_lpr_cfg<T, 32, 1, 4, 1, true>(a, s);
    if (rb_env == 2) {
        if (a.Cin == 32 && a.Cout == 32) return launch_lpr_cfg<T, 32, 1, 3, 2>(a, s);
        if (a.Cin == 32) return launch_lpr_cfg<T, 32, 2, 4, 1>(a, s);      // 32 -> 64 with two row blocks spills (144 weight + 64 accumulator registers)
    } else {
        if (a.Cin == 32) return a.Cout == 32 ? launch_lpr_cfg<T, 32, 1, 4, 1>(a, s) : launch_lpr_cfg<T, 32, 2, 4, 1>(a, s);
    }
    return a.Cout == 32 ? launch_lpr_cfg<T, 64, 1, 3, 1>(a, s) : launch_lpr_cfg<T, 64, 2, 3, 1>(a, s);
}

hipError_t launch_conv3x3_lpr(const ConvArgs &a, bool fp16, hipStream_t s)
{
    if (!lpr_shape_ok(a)) return hipErrorInvalidValue;
    return fp16 ? launch_lpr<_Float16>(a, s) : launch_lpr<__bf16>(a, s);
}

}  // namespace miunet
