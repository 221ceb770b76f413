// conv_wino4.hip -- Winograd F(4x4,3x3) convolution on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950 only.
#include "kernel_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// F(4x4, 3x3): a 6x6 input tile gives a 4x4 output tile with 36 multiplies per channel pair instead of 144 -- 4x fewer
// MACs than the direct form, 1.78x fewer than F(2x2,3x3); all arithmetic stays fp32.
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          (Lavin & Gray's matrices, interpolation points 0, +-1, +-2, inf)
// The 36 element-wise products are 36 independent GEMMs  M_p[tile][co] = sum_ci V_p[tile][ci] * U_p[ci][co], p = 6*xi + nu.
// Whole-network error of this form against fp64 is 2e-5 on logits of magnitude 4 (direct fp32: 1e-5; the tolerance of
// the path is 1e-3): the transforms only multiply by small integers, U is built in double on the host.
// Mapping to CDNA4:
//   * a workgroup = 4 waves (one per SIMD, the whole register file each) owns 16 tiles (a 16x16 block of output
//     pixels) x 128 output channels; wave w owns 16 tiles x 32 channels for ALL 36 positions on the 16x16x4 MFMA
//     = 36 x 2 accumulators of 4 registers = 288 registers.  A lane then holds every position of its (tile, channel)
//     pairs, so the inverse transform A^T M A is in-lane arithmetic: no LDS exchange, no second pass;
//   * U fragments are wave-private: 16-byte buffer loads straight from global / L2 with scalar offsets, six positions
//     ahead in a register ring;
//   * the raw 18x18 halo patch is staged through a double-buffered LDS image, 16 channels (64 bytes per pixel) at a
//     time, two chunks ahead of the MFMAs; V = B^T d B is built from it into a double-buffered LDS image
//     [pos][tile][16 + 4 pad] (80-byte rows: 5i mod 16 is a bijection -> conflict-free ds_read_b128);
//   * the forward transform is cut by ROWS of B^T: waves 0 and 1 build two rows each (xi = 1,2 and 3,4, which share
//     their sub-expressions), waves 2 and 3 one row each (xi = 0 and 5); every lane = (tile, channel quad); the pieces
//     are threaded between the MFMAs of the chunk that precedes their use;
//   * K order inside a 16-channel chunk: MFMA step s of lane group kq consumes channel 4*kq + s (one ds_read_b128 / one
//     buffer_load_b128 per lane feeds four MFMA steps).
struct W4 {
    static constexpr int TMB = 16;                          // 4x4 output tiles per workgroup (4 wide x 4 tall)
    static constexpr int VROW = WINO4_KC + 4;               // padded floats per tile row of V
    static constexpr int VPOS = TMB * VROW;                 // floats per position
    static constexpr int VBUF = 36 * VPOS;                  // floats per V buffer
    static constexpr int RAWPIX = 18 * 18;
    static constexpr int RAW_P = WINO4_KC + 4;              // padded floats per raw pixel (80-byte stride: the 4 tiles x 4 quads
                                                            // of a 16-lane group hit 64 distinct banks)
    static constexpr int RAW_FLOATS = RAWPIX * RAW_P;       // one 16-channel chunk of the halo patch
    static constexpr int RAW_ITERS = (RAWPIX * (WINO4_KC / 4) + 255) / 256;     // 6 (the last one: 16 live lanes)
    static constexpr size_t LDS_BYTES = sizeof(float) * (2 * VBUF + 2 * RAW_FLOATS);
    static constexpr int UD = 6;                            // U prefetch distance in positions (36 % UD == 0)
};

__global__ __launch_bounds__(256, 1) void conv3x3_wino4_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                            const int m_tiles, const int nwg)
{
    constexpr int VROW = W4::VROW, VPOS = W4::VPOS, VBUF = W4::VBUF, RAW_P = W4::RAW_P, UD = W4::UD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Vs = lds;                    // [2][36][16][VROW]
    float *const Raw = lds + 2 * VBUF;        // [2][RAWPIX][RAW_P]: the input halo patch of one 16-channel chunk, double-buffered

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, kq = lane >> 4;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int bx0 = tx * 16, by0 = ty * 16, n0 = n_tile * 128;
    const float *in_img = a.in + (size_t)b * a.H * a.W * a.ldc;

    // ---- stage 1: raw halo patch, one 16-channel chunk at a time, global -> registers -> LDS (4 lanes = one pixel's 64
    // bytes; zero padding, channels past Cin and dead slots through the buffer range check: voffset 0xFFFFFFFF reads zeros)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in_img), 0, a.H * a.W * a.ldc * 4, 0x00020000);
    unsigned raw_voff[W4::RAW_ITERS];
#pragma unroll
    for (int s = 0; s < W4::RAW_ITERS; ++s) {
        const int pix = (tid >> 2) + 64 * s;
        const int py = pix / 18, px = pix - py * 18;
        const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
        const bool inb = pix < W4::RAWPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        raw_voff[s] = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * (tid & 3)) * 4) : 0xFFFFFFFFu;
    }
    float *const raw_wr = Raw + (tid >> 2) * RAW_P + 4 * (tid & 3);          // + buf*RAW_FLOATS + s*64*RAW_P
    f32x4 raw_reg[W4::RAW_ITERS];
    auto raw_load = [&](int chunk) {
        const int c0 = chunk * WINO4_KC;
        const bool c_ok = c0 + 4 * (tid & 3) < a.Cin;
#pragma unroll
        for (int s = 0; s < W4::RAW_ITERS; ++s)
            raw_reg[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, c_ok ? raw_voff[s] : 0xFFFFFFFFu, c0 * 4, 0));
    };
    auto raw_store = [&](int buf) {
#pragma unroll
        for (int s = 0; s < W4::RAW_ITERS; ++s)
            if (s + 1 < W4::RAW_ITERS || (tid >> 2) + 64 * s < W4::RAWPIX)
                *reinterpret_cast<f32x4 *>(raw_wr + buf * W4::RAW_FLOATS + s * 64 * RAW_P) = raw_reg[s];
    };

    // ---- stage 2: V = B^T d B for one 16-channel chunk; lane = (tile, channel quad), wave = row group of B^T:
    //   wave 0: xi 1, 2 = (d4 - 4 d2) +- (d3 - 4 d1)        wave 1: xi 3, 4 = (d4 - d2) +- (2 d3 - 2 d1)
    //   wave 2: xi 0    = 4 d0 - 5 d2 + d4                  wave 3: xi 5    = 4 d1 - 5 d3 + d5
    const bool two = wave < 2;
    const float c_al = wave == 0 ? 4.f : 1.f, c_be = wave == 0 ? 1.f : 2.f, c_ga = wave == 0 ? 4.f : 2.f;
    const int t_tile = lane >> 2, t_quad = lane & 3;
    const int row0 = two ? 1 : wave - 2, rstep = two ? 1 : 2;
    const float *const p_rd = Raw + ((4 * (t_tile >> 2) + row0) * 18 + 4 * (t_tile & 3)) * RAW_P + 4 * t_quad;
    const int p_rstride = rstep * 18 * RAW_P;
    const int xi_a = two ? (wave == 0 ? 1 : 3) : (wave == 2 ? 0 : 5);
    float *const v_wr_a = Vs + xi_a * 6 * VPOS + t_tile * VROW + 4 * t_quad;      // + buf*VBUF + nu*VPOS; row b = + 6*VPOS
    f32x4 px_[4];                             // patch column k of this lane's rows
    f32x4 cR[2][6];                           // rows of B^T d (row b only on the two-row waves)
    f32x4 e_[4];
    auto piece_load = [&](int k, int rbuf) {
        const float *src = p_rd + rbuf * W4::RAW_FLOATS + k * RAW_P;
        px_[0] = *reinterpret_cast<const f32x4 *>(src);
        px_[1] = *reinterpret_cast<const f32x4 *>(src + p_rstride);
        px_[2] = *reinterpret_cast<const f32x4 *>(src + 2 * p_rstride);
        if (two) px_[3] = *reinterpret_cast<const f32x4 *>(src + 3 * p_rstride);
    };
    auto piece_col = [&](int k) {
        const f32x4 *d = px_;
        if (two) {
            const f32x4 ta = d[3] - c_al * d[1];
            const f32x4 tb = c_be * d[2] - c_ga * d[0];
            cR[0][k] = ta + tb;
            cR[1][k] = ta - tb;
        } else {
            cR[0][k] = 4.f * d[0] - 5.f * d[1] + d[2];
        }
    };
    auto piece_prep = [&](int row) {
        const f32x4 *c = cR[row];
        e_[0] = c[4] - 4.f * c[2]; e_[1] = c[3] - 4.f * c[1]; e_[2] = c[4] - c[2]; e_[3] = c[3] - c[1];
    };
    auto piece_store = [&](int row, float *dst, int nu) {
        const f32x4 *c = cR[row];
        const f32x4 v = nu == 0 ? 4.f * c[0] - 5.f * c[2] + c[4]
                      : nu == 1 ? e_[0] + e_[1]
                      : nu == 2 ? e_[0] - e_[1]
                      : nu == 3 ? e_[2] + 2.f * e_[3]
                      : nu == 4 ? e_[2] - 2.f * e_[3]
                                : 4.f * c[1] - 5.f * c[3] + c[5];
        *reinterpret_cast<f32x4 *>(dst + nu * VPOS) = v;
    };
    // the transform of one chunk as 23 numbered pieces (0-5 load, 6-11 column pass, 12 prep a, 13-18 row a, 19 prep b ...)
    auto transform_all = [&](int rbuf, int buf) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { piece_load(k, rbuf); piece_col(k); }
        piece_prep(0);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) piece_store(0, v_wr_a + buf * VBUF, nu);
        if (two) {
            piece_prep(1);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) piece_store(1, v_wr_a + buf * VBUF + 6 * VPOS, nu);
        }
    };

    // ---- MFMA role: wave w, channels n0 + 32 w .. + 31 (two 16-column blocks), all 36 positions
    const int ncol0 = n0 + 32 * wave + j16;
    const unsigned u_pos_bytes = (unsigned)a.CoutPad * WINO4_KC * 4;
    const unsigned u_voff = (unsigned)(ncol0 * WINO4_KC + 4 * kq) * 4;
    const int all_chunks = (a.Cin + WINO4_KC - 1) / WINO4_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk4), 0, (int)(all_chunks * 36 * u_pos_bytes), 0x00020000);
    auto u_load = [&](int chunk, int p, int blk) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff + blk * 16 * WINO4_KC * 4,
                                                                              (chunk * 36 + p) * u_pos_bytes, 0));
    };
    const float *const v_rd = Vs + j16 * VROW + 4 * kq;                       // + buf*VBUF + pos*VPOS

    f32x4 acc[36][2];
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[p][0][r] = 0.f; acc[p][1][r] = 0.f; }

    const int nchunks = all_chunks;
    f32x4 u[UD][2];
#pragma unroll
    for (int p = 0; p < UD; ++p) { u[p][0] = u_load(0, p, 0); u[p][1] = u_load(0, p, 1); }
    // prologue: raw patches of chunks 0 and 1 -> LDS, V of chunk 0
    raw_load(0);
    raw_store(0);
    raw_load(1);
    raw_store(1);
    __syncthreads();
    transform_all(0, 0);
    __syncthreads();

    // Per chunk c, around its 288 MFMAs (one barrier per chunk):
    //   positions 0 / 24  : buffer loads of chunk c+2's raw patch / their LDS stores into Raw[c & 1] (whose last reader, the
    //                       transform of chunk c, ran during chunk c-1)
    //   positions 2 - 22  : the transform of chunk c+1 (Raw[(c+1) & 1] -> V[(c+1) & 1]) in pieces between the MFMAs
    //   every position    : the V fragment one position ahead, the U ring six positions ahead
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int nxt = (chunk + 1 < nchunks) ? chunk + 1 : chunk;          // the last chunk prefetches itself: straight-line code
        const int rbuf = (chunk + 1) & 1;
        const float *vb = v_rd + (chunk & 1) * VBUF;
        float *const wr = v_wr_a + ((chunk + 1) & 1) * VBUF;
        f32x4 av = *reinterpret_cast<const f32x4 *>(vb);
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            f32x4 avn = av;
            if (p + 1 < 36) avn = *reinterpret_cast<const f32x4 *>(vb + (p + 1) * VPOS);   // V fragment one position ahead
#ifndef W4_ABL_NO_RAW
            if (p == 0) raw_load(chunk + 2);
            if (p == 24) raw_store(chunk & 1);
#endif
#ifndef W4_ABL_NO_TRANSFORM
            if (p == 2) piece_load(0, rbuf);
#endif
            __builtin_amdgcn_sched_barrier(0);        // ... issued BEFORE this position's MFMAs (hipcc would sink them to their use)
            const f32x4 b0 = u[p % UD][0], b1 = u[p % UD][1];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b0[s], acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b1[s], acc[p][1], 0, 0, 0);
#ifndef W4_ABL_NO_TRANSFORM
                if (s == 1) {
                    if (p >= 3 && p < 9) piece_col(p - 3);
                    if (p == 9) piece_prep(0);
                    if (p >= 10 && p < 16) piece_store(0, wr, p - 10);
                    if (two) {
                        if (p == 16) piece_prep(1);
                        if (p >= 17 && p < 23) piece_store(1, wr + 6 * VPOS, p - 17);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == 2 && p >= 3 && p < 8) {      // the next patch column, once the column pass above has consumed this one
                    piece_load(p - 2, rbuf);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
            }
            const int pn = p + UD;                                                              // refill the ring slot
#ifndef W4_ABL_NO_ULOAD
            u[p % UD][0] = u_load(pn < 36 ? chunk : nxt, pn % 36, 0);
            u[p % UD][1] = u_load(pn < 36 ? chunk : nxt, pn % 36, 1);
#endif
            av = avn;
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in-lane, + shift, ReLU, 4x4 store (+ the 2x2 pooled maxima).
    // Lane = channel j16 of block blk, register r = tile (row kq, column r) of the 4x4 tile block.
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int ncol = ncol0 + 16 * blk;
        const bool n_ok = ncol < a.Cout;
        const float sh = n_ok ? a.bias[ncol] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = by0 + 4 * kq, ox = bx0 + 4 * r;
            float t[4][6];
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const float m0 = acc[nu][blk][r], m1 = acc[6 + nu][blk][r], m2 = acc[12 + nu][blk][r], m3 = acc[18 + nu][blk][r],
                            m4 = acc[24 + nu][blk][r], m5 = acc[30 + nu][blk][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                t[0][nu] = m0 + s12 + s34;
                t[1][nu] = d12 + 2.f * d34;
                t[2][nu] = s12 + 4.f * s34;
                t[3][nu] = d12 + 8.f * d34 + m5;
            }
            float y[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float s12 = t[i][1] + t[i][2], d12 = t[i][1] - t[i][2], s34 = t[i][3] + t[i][4], d34 = t[i][3] - t[i][4];
                y[i][0] = t[i][0] + s12 + s34;
                y[i][1] = d12 + 2.f * d34;
                y[i][2] = s12 + 4.f * s34;
                y[i][3] = d12 + 8.f * d34 + t[i][5];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = y[i][k] + sh;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    y[i][k] = v;
                    if (n_ok && oy + i < a.H && ox + k < a.W)
                        a.out[(((size_t)b * a.H + oy + i) * a.W + ox + k) * a.ldo + a.co_off + ncol] = v;
                }
            if (a.pool_out != nullptr && n_ok) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (oy + 2 * i + 1 < a.H && ox + 2 * k + 1 < a.W)
                            a.pool_out[(((size_t)b * (a.H >> 1) + ((oy >> 1) + i)) * (a.W >> 1) + ((ox >> 1) + k)) * a.pool_ld + ncol] =
                                fmaxf(fmaxf(y[2 * i][2 * k], y[2 * i][2 * k + 1]), fmaxf(y[2 * i + 1][2 * k], y[2 * i + 1][2 * k + 1]));
            }
        }
    }
}

hipError_t launch_conv3x3_wino4(const ConvArgs &a, hipStream_t s)
{
    if (a.wpk4 == nullptr || a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 127) / 128;
    const int nwg = m_tiles * n_tiles;
    if (hipError_t e = ensure_dynamic_lds(conv3x3_wino4_f32, W4::LDS_BYTES); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv3x3_wino4_f32, dim3(nwg), dim3(256), W4::LDS_BYTES, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

}  // namespace miunet
