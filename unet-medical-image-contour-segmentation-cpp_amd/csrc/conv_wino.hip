// conv_wino.hip -- Winograd F(2x2,3x3) convolution on the fp32 MFMA (two tilings), gfx950 only.
#include "kernel_common.h"

namespace miunet {

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

// SPLITK = false compiles the split-K paths out entirely (the full-K kernel's register allocation and schedule are
// exactly those of a kernel without them: measured, a runtime ksplit == 1 path costs 4 % at batch 16).
template <int WM, int WN, bool SPLITK>
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

    int L = xcd_remap(blockIdx.x, nwg);
    int ks = 0;                               // K slice of this workgroup
    if constexpr (SPLITK) { ks = L % a.ksplit; L /= a.ksplit; }
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

    // this workgroup's K range [c_begin, nchunks): whole super-chunks (c_begin is a multiple of 4), never empty
    const int all_chunks = (a.Cin + WINO_KC - 1) / WINO_KC;
    int c_begin = 0, nchunks = all_chunks;
    if constexpr (SPLITK) {
        const int per_slice = ((all_chunks + a.ksplit - 1) / a.ksplit + CPS - 1) / CPS * CPS;
        c_begin = ks * per_slice;
        nchunks = (c_begin + per_slice < all_chunks) ? c_begin + per_slice : all_chunks;
    }
    const int nsuper = (nchunks + CPS - 1) / CPS;
    f32x4 u[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] = u_load(c_begin, p);
    // prologue: raw patch of the first super-chunk -> LDS, V of the first chunk
    raw_load(c_begin / CPS);
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
    for (int chunk = c_begin; chunk < nchunks; ++chunk) {
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
        if constexpr (SPLITK) {               // partial sums of this K slice: the reduce kernel finishes the layer
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    if (n_ok && oy + dy < a.H && ox + dx < a.W)
                        a.ksplit_ws[((((size_t)ks * a.B + b) * a.H + oy + dy) * a.W + ox + dx) * a.Cout + ncol] = y[dy][dx];
            continue;
        }
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
    if (hipError_t e = ensure_dynamic_lds(conv3x3_wino16_f32, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv3x3_wino16_f32, dim3(nwg), dim3(W16_THREADS), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino16(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    return launch_wino16(a, s);
}

// sum the K slices in slice order, then shift, ReLU, channel-offset store and the optional 2x2 max pooling.
// One thread = one 2x2 pixel block x 4 channels (16-byte accesses).
__global__ __launch_bounds__(256) void wino_splitk_reduce(const ConvArgs a, long long total)
{
    const int C4 = a.Cout / 4, Hb = (a.H + 1) / 2, Wb = (a.W + 1) / 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long long t = e / C4;
        const int xb = (int)(t % Wb); t /= Wb;
        const int yb = (int)(t % Hb);
        const int b = (int)(t / Hb);
        const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.bias + 4 * c4);
        f32x4 vmax = { -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f };
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * yb + dy, x = 2 * xb + dx;
                if (y >= a.H || x >= a.W) continue;
                const size_t px = ((size_t)b * a.H + y) * a.W + x;
                f32x4 acc = *reinterpret_cast<const f32x4 *>(a.ksplit_ws + px * a.Cout + 4 * c4);
                for (int k = 1; k < a.ksplit; ++k)
                    acc += *reinterpret_cast<const f32x4 *>(a.ksplit_ws + ((size_t)k * a.B * a.H * a.W + px) * a.Cout + 4 * c4);
                acc += sh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (a.relu) acc[q] = acc[q] > 0.f ? acc[q] : 0.f;
                    vmax[q] = fmaxf(vmax[q], acc[q]);
                }
                *reinterpret_cast<f32x4 *>(a.out + px * a.ldo + a.co_off + 4 * c4) = acc;
            }
        if (a.pool_out != nullptr && 2 * yb + 1 < a.H && 2 * xb + 1 < a.W)
            *reinterpret_cast<f32x4 *>(a.pool_out + (((size_t)b * (a.H >> 1) + yb) * (a.W >> 1) + xb) * a.pool_ld + 4 * c4) = vmax;
    }
}

// second half of a split-K layer (either Winograd kernel): a.ksplit slabs [slice][B][H][W][Cout] in a.ksplit_ws -> a.out
hipError_t launch_wino_splitk_reduce(const ConvArgs &a, hipStream_t s)
{
    const long long total = (long long)a.B * ((a.H + 1) / 2) * ((a.W + 1) / 2) * (a.Cout / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, s, a, total);
    return hipGetLastError();
}

template <int WM, int WN>
static hipError_t launch_wino_cfg(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 8 * WM - 1) / (8 * WM);
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 32 * WN - 1) / (32 * WN);
    // split K only when the grid cannot fill half of the 256 CUs (one workgroup per CU): each slice keeps >= 2 super-chunks
    a.ksplit = 1;
    const int chunks = (a.Cin + WINO_KC - 1) / WINO_KC, tiles = m_tiles * n_tiles;
    if (a.ksplit_ws != nullptr && tiles <= 128 && chunks >= 16 && a.Cout % 4 == 0 && a.ldo % 4 == 0 && a.co_off % 4 == 0) {
        int ks = 256 / tiles;
        if (ks > 8) ks = 8;
        if (ks > chunks / 8) ks = chunks / 8;
        while (ks > 1 && (size_t)ks * a.B * a.H * a.W * a.Cout * sizeof(float) > a.ksplit_ws_bytes) --ks;
        // every slice must own at least one chunk after rounding its share up to whole super-chunks
        while (ks > 1 && (ks - 1) * (((chunks + ks - 1) / ks + 3) / 4 * 4) >= chunks) --ks;
        a.ksplit = ks;
    }
    const int nwg = tiles * a.ksplit;
    constexpr size_t lds = WinoGeom<WM, WN>::LDS_BYTES;
    if (a.ksplit > 1) {
        auto kern = conv3x3_wino_f32<WM, WN, true>;
        if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    } else {
        auto kern = conv3x3_wino_f32<WM, WN, false>;
        if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg);
    }
    if (a.ksplit > 1) {
        const long long total = (long long)a.B * ((a.H + 1) / 2) * ((a.W + 1) / 2) * (a.Cout / 4);
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(wino_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, s, a, total);
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvArgs &a, hipStream_t s)
{
    if (a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    // Cout >= 128: 32 tiles x 128 channels per workgroup (half the input-transform work per MFMA); else 64 x 64
    if (a.Cout > 64) return launch_wino_cfg<1, 4>(a, s);
    return launch_wino_cfg<2, 2>(a, s);
}


}  // namespace miunet
