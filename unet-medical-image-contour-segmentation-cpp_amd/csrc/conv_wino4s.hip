// conv_wino4s.hip -- Winograd F(4x4,3x3) for the layers with at most 64 output channels per workgroup, built so that TWO
// workgroups share a compute unit.  fp32 arithmetic on v_mfma_f32_16x16x4_f32, gfx950 only.
//
// Why a second kernel.  The one-block variant of conv_wino4.hip (NB = 1) holds one 144 KB workgroup per CU and threads the
// forward transform, the raw-patch DMA and the U ring between its MFMAs.  On the three top-level layers (inc.c2, up4.c1,
// up4.c2 + head: K = 4 or 8 chunks of 16 channels, 1-2 GiB of activations each) its matrix pipe is only 37-46 % busy
// (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES, profiles/r02a_pmc_fp32.json): a tile spends about as long in its prologue, in the
// stalls of chunks that open new 128-byte lines, and in its epilogue as in its MFMAs, and with one wave per SIMD nothing
// else can issue meanwhile.  On gfx950 the fp32 MFMA never co-executes with VALU work anyway (SQ_VALU_MFMA_COEXEC_CYCLES
// = 0 on every fp32 kernel of this library), so interleaving buys latency hiding only -- and a second resident workgroup
// buys the same, for every kind of stall at once, with much simpler code.
//
// Shape.  Workgroup = 4 waves = a 16x16 block of output pixels (16 tiles of 4x4) x 64 output channels; wave w owns 16
// channels for all 36 positions: 36 accumulators of 4 registers = 144, the whole kernel inside 256 registers
// (__launch_bounds__(256, 2)).  LDS = ONE V image [36][16 tiles][16] (36,864 B) + ONE raw patch image
// [18 rows][20 slots][16 channels] (23,552 B) = 60,416 B, twice per CU.  A K-chunk of 16 input channels is three phases:
//      wait for the chunk's raw patch (LDS-DMA, vmcnt) | barrier | forward transform raw -> V | barrier |
//      144 MFMAs per wave, with the DMA of the NEXT chunk's patch and the U ring (16-byte buffer loads, three positions
//      ahead) issued between them
// Single buffers cost two barriers per chunk and serialise transform and MFMAs inside a workgroup; the co-resident
// workgroup runs its MFMAs meanwhile.  Everything else -- the slot-permuted raw image, the row-split forward transform,
// the in-lane inverse transform, the free bias at position (1,1), buffer stores, fused pooling, the fused 1x1 head -- is
// the arithmetic of conv_wino4.hip, so the two kernels give bit-identical results.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"
#include "wino4_common.h"

namespace miunet {

struct W4S {
    static constexpr int HEAD_ROW = 64 + 4;                 // floats per pixel of the head's LDS tile (conflict-free b128 rows)
    static constexpr size_t LDS_BYTES = sizeof(float) * (W4::VBUF + W4::RAW_FLOATS);          // 60,416
    static constexpr size_t LDS_BYTES_HEAD = sizeof(float) * (256 * HEAD_ROW + 4 * 64);       // the head tile (larger than V + raw) + its weights
    static constexpr int FIRST_WIN = 20;                    // fused first layer: the normalised (16 + 4)^2 window of the image behind V + raw
    static constexpr int FIRST_WMAX = 64;                   // ... and the first layer's weights [9][Cin] + shift [Cin] (Cin <= 64: no long-lived registers)
    static constexpr size_t LDS_BYTES_FIRST = LDS_BYTES + sizeof(float) * (FIRST_WIN * FIRST_WIN + 10 * FIRST_WMAX);
    static_assert(LDS_BYTES <= sizeof(float) * 256 * HEAD_ROW, "the head launch's allocation must cover the K loop's V + raw images");
};

// UD = U prefetch distance in positions (36 % UD == 0).  The ring is carried across the chunk boundary: its first UD loads of
// chunk c+1 fly during the barriers and the transform of that chunk, at 4 UD live registers there.  Round 3 priced the K loop's
// operand streams with timing-only builds (EXP below, profiles/r03_ab_wino4s_k_loop.txt): never refilling the U ring makes these
// layers 15-25 % faster, no patch DMA inside the loop 12-20 % -- a ring of three positions is 384 cycles of lead against an L2
// round trip, and a wave's vmcnt counts its loads in order.  UD = 6 is the deepest ring that compiles without spills next to the
// transform's temporaries since the branch-free role code (round 2: three): these layers -3 ... -10 %, the step 17.19 -> 16.89 ms
// same card.  Nine and twelve positions fit only with the two-row waves building their rows one after the other (24 fewer live
// registers, the patch column read twice): slower than six with the one-pass transform.  (Round 2: refilling a deeper ring AFTER the
// transform instead measured 15-40 % slower -- spills, and the refill's latency at the head of every MFMA phase.)
// EXP != 0: timing-only experiment builds (MIUNET_W4S_EXP; results are WRONG, never routed by default; switch the numeric guard off,
// MIUNET_WINO4_GUARD=0, or it sends the whole plan to F(2x2)): 1 = the U ring is never refilled, 2 = no raw-patch DMA inside the K
// loop, 3 = both
// LATE: the next chunk's raw patch is requested in the LAST six positions of the MFMA phase instead of at every fourth position from
// the first on: a wave's vmcnt is in order, so every U fragment younger than a patch request waits for HBM with it; requested late,
// the patch's latency falls into the wait in front of the next transform, which the co-resident workgroup's MFMAs cover
// (inc.c2 -2 %, up4.c1 -2.8 %, the others unchanged: profiles/r03_ab_wino4s_k_loop.txt).
// FIRST: the layer's input is the network's FIRST layer, computed here (ConvArgs::first_img): instead of the LDS-DMA of a raw chunk,
// every thread builds its 16-byte piece of the patch image -- the same (slot, channel quad) piece the DMA would have written --
// from a 20x20 window of the u8 image that the workgroup normalises once: acc = sum over the nine taps (raster order, fma chain from
// 0), + shift, ReLU: bit for bit what conv3x3_first_kernel stores, zero where the patch leaves the image (this layer's padding).
template <bool HEAD, int UD, int EXP = 0, bool LATE = true, bool FIRST = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wino4s_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                             const int m_tiles, const int nwg)
{
    constexpr int VROW = W4::VROW, VPOS = W4::VPOS;
    constexpr int HEAD_ROW = W4S::HEAD_ROW;
    static_assert(36 % UD == 0, "ring depth must divide the position count");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Vs = lds;                    // [36][16][VROW]
    float *const Raw = lds + W4::VBUF;        // [18 rows][20 slots][16]
    float *const Win = Raw + W4::RAW_FLOATS;  // FIRST: [20][20] normalised image window, origin (by0 - 2, bx0 - 2)
    float *const Wf = Win + W4S::FIRST_WIN * W4S::FIRST_WIN;      // FIRST: [9][Cin] weights, then [Cin] shifts

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, kq = lane >> 4;

    // ---- this workgroup's tile (bijective XCD remap: the workgroups of one XCD walk neighbouring tiles, kernel_common.h)
    int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int bx0 = tx * 16, by0 = ty * 16;
    const int ncol0 = n_tile * 64 + 16 * wave + j16;
    const unsigned u_voff = (unsigned)(ncol0 * WINO4_KC + 4 * kq) * 4;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in + (size_t)b * a.H * a.W * a.ldc), 0, a.H * a.W * a.ldc * 4, 0x00020000);

    // ---- raw halo patch of one 16-channel chunk: global -> LDS by DMA (4 lanes = one pixel's 64 bytes; zero padding, channels
    // past Cin and dead slots through the buffer range check: voffset 0xFFFFFFFF reads zeros)
    unsigned raw_voff[W4::RAW_ITERS];
#pragma unroll
    for (int s = 0; s < W4::RAW_ITERS; ++s) {
        const int g = (wave + 4 * s) * 16 + (lane >> 2);                  // linear slot of this lane in load wave + 4s
        const int py = g / W4::RAW_ROW, sl = g - py * W4::RAW_ROW;
        const int px = 4 * (sl % 5) + sl / 5;
        const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
        const bool inb = g < W4::RAW_SLOTS && px < 18 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        raw_voff[s] = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * (lane & 3)) * 4) : 0xFFFFFFFFu;
    }
    auto first_patch = [&](int chunk) {                        // FIRST: this thread's pieces of a chunk's patch, computed
        const int cq = chunk * WINO4_KC + 4 * (lane & 3);
        const float *wq = Wf + cq;                             // tap t: wq[t * Cin .. + 3]; the shifts behind the ninth tap
#pragma unroll 1                                               // a real loop: unrolled, hipcc hoists 6 x 18 LDS reads and spills accumulators
        for (int s = 0; s < W4::RAW_ITERS; ++s) {
            if (wave + 4 * s >= W4::RAW_LOADS) break;
            const int g = (wave + 4 * s) * 16 + (lane >> 2);
            const int py = g / W4::RAW_ROW, sl = g - py * W4::RAW_ROW;
            const int px = 4 * (sl % 5) + sl / 5;
            const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
            const bool inb = g < W4::RAW_SLOTS && px < 18 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            f32x4 r = f32x4{ 0.f, 0.f, 0.f, 0.f };
            if (inb) {
                f32x4 acc = f32x4{ 0.f, 0.f, 0.f, 0.f };
                const float *wp = Win + py * W4S::FIRST_WIN + px;      // tap (dy, dx) of patch pixel (py, px): window (py + dy, px + dx)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc += wp[(t / 3) * W4S::FIRST_WIN + t % 3] * *reinterpret_cast<const f32x4 *>(wq + t * a.Cin);
                const f32x4 a4 = acc + *reinterpret_cast<const f32x4 *>(wq + 9 * a.Cin);
                r.x = a4.x > 0.f ? a4.x : 0.f; r.y = a4.y > 0.f ? a4.y : 0.f; r.z = a4.z > 0.f ? a4.z : 0.f; r.w = a4.w > 0.f ? a4.w : 0.f;
            }
            *reinterpret_cast<f32x4 *>(Raw + (size_t)g * WINO4_KC + 4 * (lane & 3)) = r;
        }
    };
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto raw_dma_one = [&](int chunk, int s) {                 // the wave's s-th load of a chunk's patch
        const int c0 = chunk * WINO4_KC;
        const bool c_ok = c0 + 4 * (lane & 3) < a.Cin;
        if (wave + 4 * s < W4::RAW_LOADS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(Raw + (wave + 4 * s) * 16 * WINO4_KC), 16,
                                                     c_ok ? raw_voff[s] : 0xFFFFFFFFu, c0 * 4, 0, 0);
    };

    // ---- forward transform V = B^T d B of one chunk; lane = (tile, channel quad), wave = row group of B^T:
    //   wave 0: xi 1, 2 = (d4 - 4 d2) +- (d3 - 4 d1)        wave 1: xi 3, 4 = (d4 - d2) +- (2 d3 - 2 d1)
    //   wave 2: xi 0    = 4 d0 - 5 d2 + d4                  wave 3: xi 5    = 4 d1 - 5 d3 + d5
    const bool two = wave < 2;
    const int t_tile = lane >> 2, t_quad = lane & 3;
    const int row0 = two ? 1 : wave - 2, rstep = two ? 1 : 2;
    const float *const p_rd = Raw + (((4 * (t_tile >> 2) + row0) * W4::RAW_ROW + (t_tile & 3)) * 4 + t_quad) * 4;
    const int p_rstride = rstep * W4::RAW_ROW * WINO4_KC;
    const int xi_a = two ? (wave == 0 ? 1 : 3) : (wave == 2 ? 0 : 5);
    const float c_alpha = wave == 0 ? -4.f : -1.f, c_beta = wave == 0 ? 1.f : 2.f;
    float *const v_wr_a = Vs + xi_a * 6 * VPOS + t_tile * VROW + 4 * (t_quad ^ v_swz(t_tile));      // + nu*VPOS; row b = + 6*VPOS
    auto column_pass = [&](const f32x4 *c, float *dst) {
        const f32x4 e0 = c[4] - 4.f * c[2], e1 = c[3] - 4.f * c[1], e2 = pk_sub(c[4], c[2]), e3 = pk_sub(c[3], c[1]);
        *reinterpret_cast<f32x4 *>(dst + 0 * VPOS) = 4.f * c[0] - 5.f * c[2] + c[4];
        *reinterpret_cast<f32x4 *>(dst + 1 * VPOS) = e0 + e1;
        *reinterpret_cast<f32x4 *>(dst + 2 * VPOS) = pk_sub(e0, e1);
        *reinterpret_cast<f32x4 *>(dst + 3 * VPOS) = e2 + 2.f * e3;
        *reinterpret_cast<f32x4 *>(dst + 4 * VPOS) = e2 - 2.f * e3;
        *reinterpret_cast<f32x4 *>(dst + 5 * VPOS) = 4.f * c[1] - 5.f * c[3] + c[5];
    };
    auto transform = [&]() {
        f32x4 cR[2][6];                       // rows of B^T d (row b only on the two-row waves)
#pragma unroll
        for (int k = 0; k < 6; ++k) {         // patch column k of this lane's rows: pixel 4*tx + k -> slot 5*(k&3) + tx + (k>>2)
            const float *src = p_rd + ((k & 3) * 5 + (k >> 2)) * WINO4_KC;
            f32x4 d[4];
            d[0] = *reinterpret_cast<const f32x4 *>(src);
            d[1] = *reinterpret_cast<const f32x4 *>(src + p_rstride);
            d[2] = *reinterpret_cast<const f32x4 *>(src + 2 * p_rstride);
            d[3] = *reinterpret_cast<const f32x4 *>(src + 3 * p_rstride);       // (unused by the one-row waves: patch rows 6, 7)
            f32x4 ra, rb;
            if (two) {                        // waves 0 and 1: one instruction sequence, wave-uniform coefficients (conv_wino4.hip)
                const f32x4 ta = d[3] + c_alpha * d[1];
                const f32x4 tb = d[2] + c_alpha * d[0];
                ra = ta + c_beta * tb;
                rb = ta - c_beta * tb;
            } else {
                ra = 4.f * d[0] - 5.f * d[1] + d[2];
                rb = ra;
            }
            cR[0][k] = ra;
            cR[1][k] = rb;
        }
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            if (row == 1 && !two) break;
            column_pass(cR[row], v_wr_a + row * 6 * VPOS);
        }
    };

    // ---- MFMA role: wave w, channels n0 + 16 w .. + 15, all 36 positions
    const unsigned u_pos_bytes = (unsigned)a.CoutPad * WINO4_KC * 4;
    const int nchunks = (a.Cin + WINO4_KC - 1) / WINO4_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk4), 0, (int)(nchunks * 36 * u_pos_bytes), 0x00020000);
    auto u_load = [&](int chunk, int p) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff, (chunk * 36 + p) * u_pos_bytes, 0));
    };
    const float *const v_rd = Vs + j16 * VROW + 4 * (kq ^ v_swz(j16));                       // + pos*VPOS

    // the loads that open the tile: the raw patch of chunk 0 straight into LDS, then the U ring
    if constexpr (FIRST) {
        const uint8_t *imgb = a.first_img + (size_t)b * a.H * a.W;
        for (int i = tid; i < W4S::FIRST_WIN * W4S::FIRST_WIN; i += 256) {
            const int y = i / W4S::FIRST_WIN, x = i - y * W4S::FIRST_WIN, gy = by0 - 2 + y, gx = bx0 - 2 + x;
            Win[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? a.first_lut[imgb[(size_t)gy * a.W + gx]] : 0.f;
        }
        for (int i = tid; i < 10 * a.Cin; i += 256) Wf[i] = i < 9 * a.Cin ? a.first_w[i] : a.first_shift[i - 9 * a.Cin];
    } else {
#pragma unroll
        for (int s = 0; s < W4::RAW_ITERS; ++s) raw_dma_one(0, s);
    }
    f32x4 u[UD];
#pragma unroll
    for (int p = 0; p < UD; ++p) u[p] = u_load(0, p);

    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) {
        // position (xi, nu) = (1, 1) starts at the shift: A^T e_1 e_1^T A is the all-ones tile, so the bias add is free
        const float init = (p == 7 && ncol0 < a.Cout) ? a.bias[ncol0] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[p][r] = init;
    }

    int chunk = 0;                            // at least one chunk: a do-while has no zero-trip path to merge accumulators with
    do {
        // the patch DMA of this chunk is older than the UD youngest loads (the U ring): wait for everything but those
        constexpr int LEFT = LATE ? 0 : UD;           // (late patch requests interleave with the ring's refills: wait for everything)
        if constexpr (FIRST) {
            if (chunk == 0) __syncthreads();  // the image window is complete
            first_patch(chunk);               // (the raw image is free: its last reader, the transform, is behind a barrier)
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70 | (LEFT & 15) | ((LEFT >> 4) << 14));
        }
        __syncthreads();                      // every wave's part of the patch has landed; nobody reads V any more
        transform();
        __syncthreads();                      // V complete, the raw image is free again
        const bool more = chunk + 1 < nchunks;
        const int nxt = more ? chunk + 1 : chunk;                          // the last chunk prefetches itself: straight-line code
        f32x4 av = *reinterpret_cast<const f32x4 *>(v_rd);
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            f32x4 avn = av;
            if (p + 1 < 36) avn = *reinterpret_cast<const f32x4 *>(v_rd + (p + 1) * VPOS);   // V fragment one position ahead
            // one DMA load every fourth position, not a burst: 16 line misses at a time keep the VMEM queue moving
            if (!FIRST && more && !(EXP & 2)) {   // one DMA load per position (16 line misses at a time keep the VMEM queue moving)
                if (!LATE && p % 4 == 0 && p / 4 < W4::RAW_ITERS) raw_dma_one(chunk + 1, p / 4);
                if (LATE && p >= 36 - W4::RAW_ITERS) raw_dma_one(chunk + 1, p - (36 - W4::RAW_ITERS));
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 bv = u[p % UD];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc[p], 0, 0, 0);
            const int pn = p + UD;                                                              // refill the ring slot
            if constexpr (!(EXP & 1)) u[p % UD] = u_load(pn < 36 ? chunk : nxt, pn % 36);
            av = avn;
            __builtin_amdgcn_sched_barrier(0);
        }
    } while (++chunk < nchunks);

    // ---- epilogue: Y = A^T M A in-lane on all four tiles of a lane at once (the accumulator's four registers = tile
    // columns r = 0..3 of tile row kq, lane = channel), ReLU, 4x4 stores (+ the 2x2 pooled maxima).  Stores are buffer
    // stores on a per-image descriptor; pixels past the image edge and masked channels get voffset 0xFFFFFFFF (dropped).
    // (Swapping the MFMA operand roles would give every lane four consecutive CHANNELS of one tile and so 16-byte stores,
    // a quarter of the store instructions: measured 2-4 % slower on these layers -- the stores are bound by the 64-byte
    // segments they scatter, not by their count.)
    const float relu_lo = a.relu ? 0.f : -3.402823466e+38f;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const bool do_pool = a.pool_out != nullptr;
    const unsigned pix_bytes = (unsigned)a.ldo * 4, row_bytes = (unsigned)a.W * pix_bytes;
    const unsigned ppix_bytes = (unsigned)a.pool_ld * 4, prow_bytes = (unsigned)Wp * ppix_bytes;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * a.H * a.W * a.ldo, 0,
                                                                              a.H * a.W * a.ldo * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        do_pool ? a.pool_out + (size_t)b * Hp * Wp * a.pool_ld : a.out, 0, do_pool ? Hp * Wp * a.pool_ld * 4 : 0, 0x00020000);
    const int oy = by0 + 4 * kq;
    if constexpr (HEAD) __syncthreads();      // the head tile overwrites V: every wave must be past its last MFMA phase
    auto epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        const int ncol = ncol0;
        const bool n_ok = ncol < a.Cout;
        const unsigned vbase = n_ok ? (unsigned)((oy * a.W + bx0) * a.ldo + a.co_off + ncol) * 4 : 0xFFFFFFFFu;
        const unsigned pbase = n_ok ? (unsigned)(((oy >> 1) * Wp + (bx0 >> 1)) * a.pool_ld + ncol) * 4 : 0xFFFFFFFFu;
        unsigned vcol[4][4], pcol[4][2];          // edge workgroups: per-column offsets (dead columns -> dropped stores)
        if constexpr (!INTERIOR) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) vcol[r][k] = bx0 + 4 * r + k < a.W ? vbase : 0xFFFFFFFFu;
                pcol[r][0] = bx0 + 4 * r + 1 < a.W ? pbase : 0xFFFFFFFFu;
                pcol[r][1] = bx0 + 4 * r + 3 < a.W ? pbase : 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {          // two tile columns (r = 2 h2, 2 h2 + 1) at a time
            auto half = [&](const f32x4 &v) { return h2 ? v.hi : v.lo; };
            f32x2 t[4][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const f32x2 m0 = half(acc[nu]), m1 = half(acc[6 + nu]), m2 = half(acc[12 + nu]),
                            m3 = half(acc[18 + nu]), m4 = half(acc[24 + nu]), m5 = half(acc[30 + nu]);
                const f32x2 s12 = m1 + m2, d12 = pk_sub2(m1, m2), s34 = m3 + m4, d34 = pk_sub2(m3, m4);
                t[0][nu] = m0 + s12 + s34;
                t[1][nu] = d12 + 2.f * d34;
                t[2][nu] = s12 + 4.f * s34;
                t[3][nu] = d12 + 8.f * d34 + m5;
            }
            f32x2 carry0, carry1;                 // horizontal maxima of the even row, for the 2x2 pooling
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 s12 = t[i][1] + t[i][2], d12 = pk_sub2(t[i][1], t[i][2]), s34 = t[i][3] + t[i][4], d34 = pk_sub2(t[i][3], t[i][4]);
                f32x2 y[4];
                y[0] = t[i][0] + s12 + s34;
                y[1] = d12 + 2.f * d34;
                y[2] = s12 + 4.f * s34;
                y[3] = d12 + 8.f * d34 + t[i][5];
                const bool row_ok = INTERIOR || oy + i < a.H;      // per-lane (kq) row predicate of an edge workgroup
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int r = 2 * h2 + rr;
                        const float v = fmaxf(y[k][rr], relu_lo);
                        y[k][rr] = v;
                        if constexpr (HEAD) {     // pixel (4 kq + i, 4 r + k) of the 16x16 block, channel ncol -> LDS
                            lds[((4 * kq + i) * 16 + 4 * r + k) * HEAD_ROW + (ncol & 63)] = v;
                            continue;
                        }
                        unsigned voff = vbase;
                        if constexpr (!INTERIOR) voff = row_ok ? vcol[r][k] : 0xFFFFFFFFu;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, voff,
                                                              i * row_bytes + (4 * r + k) * pix_bytes, ST_AUX);
                    }
                if (!HEAD && do_pool) {
                    f32x2 hm0, hm1;               // horizontal maxima of this row: pooled columns 2r and 2r + 1
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) { hm0[rr] = fmaxf(y[0][rr], y[1][rr]); hm1[rr] = fmaxf(y[2][rr], y[3][rr]); }
                    if ((i & 1) == 0) { carry0 = hm0; carry1 = hm1; }
                    else {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const int r = 2 * h2 + rr;
                            const float p0 = fmaxf(hm0[rr], carry0[rr]), p1 = fmaxf(hm1[rr], carry1[rr]);
                            unsigned v0 = pbase, v1 = pbase;
                            if constexpr (!INTERIOR) { v0 = row_ok ? pcol[r][0] : 0xFFFFFFFFu; v1 = row_ok ? pcol[r][1] : 0xFFFFFFFFu; }
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p0), pool_rsrc, v0,
                                                                  (i >> 1) * prow_bytes + (2 * r) * ppix_bytes, ST_AUX);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p1), pool_rsrc, v1,
                                                                  (i >> 1) * prow_bytes + (2 * r + 1) * ppix_bytes, ST_AUX);
                        }
                    }
                }
            }
        }
    };
    if (HEAD || (by0 + 16 <= a.H && bx0 + 16 <= a.W)) epilogue(std::true_type{});   // workgroup-uniform: no per-pixel predicates
    else epilogue(std::false_type{});
    if constexpr (HEAD) {
        // 1x1 head + argmax on the tile: the four waves hold 16 channels each, so the tile crossed LDS above
        // ([256 pixels][64 + 4 pad]); thread = pixel, a fixed summation order, first-max-wins argmax (src/process.cpp:158-170).
        float *const Wh = lds + 256 * HEAD_ROW;            // [classes][64]
        if (tid < a.head_classes * 64) Wh[tid] = (tid & 63) < a.Cout ? a.head_w[(tid >> 6) * a.Cout + (tid & 63)] : 0.f;
        __syncthreads();
        f32x4 d4[4];                            // four interleaved partial sums per class (packed fma), folded at the end
#pragma unroll
        for (int k = 0; k < 4; ++k) d4[k] = f32x4{ 0.f, 0.f, 0.f, 0.f };
        const float *yrow = lds + tid * HEAD_ROW;
#pragma unroll
        for (int c4 = 0; c4 < 16; ++c4) {
            const f32x4 yv = *reinterpret_cast<const f32x4 *>(yrow + 4 * c4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.head_classes) d4[k] += yv * *reinterpret_cast<const f32x4 *>(Wh + 64 * k + 4 * c4);
        }
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = k < a.head_classes ? ((d4[k].x + d4[k].y) + (d4[k].z + d4[k].w)) + a.head_b[k] : 0.f;
        const int py = by0 + (tid >> 4), px = bx0 + (tid & 15);
        if (py < a.H && px < a.W) {
            const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
            float best = -3.402823466e+38f;
            int idx = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < a.head_classes) {
                    if (a.head_logits != nullptr) a.head_logits[((size_t)b * a.head_classes + k) * hw + pin] = d[k];
                    if (d[k] > best) { best = d[k]; idx = k; }
                }
            }
            a.head_labels[(size_t)b * hw + pin] = (uint8_t)idx;
        }
    }
}

template <bool HEAD>
static hipError_t launch_wino4s_cfg(const ConvArgs &a, hipStream_t s)
{
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 63) / 64;
    const int nwg = m_tiles * n_tiles;
    size_t lds_bytes = HEAD ? W4S::LDS_BYTES_HEAD : W4S::LDS_BYTES;
    const bool first = a.first_img != nullptr;
    if (first) {
        if (HEAD || !conv3x3_wino4s_can_fuse_first(a, 1)) return hipErrorInvalidValue;
        lds_bytes = W4S::LDS_BYTES_FIRST;
    }
    // experiment switch: MIUNET_WINO4S_ONE_WG=1 asks for more than half a CU's LDS, so only ONE workgroup is
    // resident per CU -- the same kernel, same instructions, without its co-resident partner.  What that costs is what any
    // one-workgroup-per-CU re-tiling of these layers (a 32-tile block on v_mfma_f32_32x32x2_f32 needs 56 % of the register
    // file for its accumulators alone) would first have to win back.
#ifdef MIUNET_EXPERIMENTS                              // lab build only (libmiunet_exp.so, tools/dev/ab*.sh): never in libmiunet.so
    static const bool one_wg = [] { const char *e = getenv("MIUNET_WINO4S_ONE_WG"); return e && e[0] == '1'; }();
    if (one_wg) lds_bytes = 96 * 1024;
    static const int exp = [] { const char *e = getenv("MIUNET_W4S_EXP"); return e ? atoi(e) : 0; }();        // timing-only builds (see the kernel)
    static const int ud = [] { const char *e = getenv("MIUNET_W4S_UD"); return e ? atoi(e) : 6; }();            // A/B: =3 round 2's ring and early patch requests, =60 ring of six with early requests
#endif
    auto launch = [&](auto kern) {
        if (hipError_t e = ensure_dynamic_lds(kern, lds_bytes); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds_bytes, s, a, tiles_x, tiles_y, m_tiles, nwg);
        return hipGetLastError();
    };
#ifdef MIUNET_EXPERIMENTS
    if (exp == 1) return launch(conv3x3_wino4s_f32<HEAD, 3, 1, false>);
    if (exp == 2) return launch(conv3x3_wino4s_f32<HEAD, 3, 2, false>);
    if (exp == 3) return launch(conv3x3_wino4s_f32<HEAD, 3, 3, false>);
    if (ud == 3) return launch(conv3x3_wino4s_f32<HEAD, 3, 0, false>);
    if (ud == 60) return launch(conv3x3_wino4s_f32<HEAD, 6, 0, false>);
#endif
    if constexpr (!HEAD) { if (first) return launch(conv3x3_wino4s_f32<false, 6, 0, true, true>); }
    return launch(conv3x3_wino4s_f32<HEAD, 6>);
}

// the fused first layer: one input channel, whole 16-channel chunks of its output (= this layer's input), no fused head
bool conv3x3_wino4s_can_fuse_first(const ConvArgs &a, int first_cin)
{
    return first_cin == 1 && a.Cin % WINO4_KC == 0 && a.Cin >= WINO4_KC && a.Cin <= W4S::FIRST_WMAX && a.head_w == nullptr && a.wpk4 != nullptr;
}

// Same contract as the one-block variant of launch_conv3x3_wino4 (a.wpk4 = U packed [Cin/16][36][CoutPad][16]); no split-K.
hipError_t launch_conv3x3_wino4s(const ConvArgs &a, hipStream_t s)
{
    if (a.wpk4 == nullptr || a.Cin % 4 || (a.first_img == nullptr && a.ldc % 4) || a.CoutPad % NPAD) return hipErrorInvalidValue;
    if (a.head_w != nullptr) {
        if (a.Cout > 64 || a.head_classes < 1 || a.head_classes > 4 || a.pool_out != nullptr || a.head_labels == nullptr)
            return hipErrorInvalidValue;
        return launch_wino4s_cfg<true>(a, s);
    }
    return launch_wino4s_cfg<false>(a, s);
}

}  // namespace miunet
