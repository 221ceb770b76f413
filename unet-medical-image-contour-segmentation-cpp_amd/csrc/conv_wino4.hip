// conv_wino4.hip -- Winograd F(4x4,3x3) convolution on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950 only.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"
#include "wino4_common.h"

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
//     time, two chunks ahead of the MFMAs, by LDS-DMA loads (no staging registers, no ds_write; one load every fourth
//     position rather than a burst); V = B^T d B is built from it into a double-buffered LDS image
//     [pos][tile][16] (64-byte rows, channel quads permuted for the 16-lane service groups of ds_read_b128: wino4_common.h);
//   * the forward transform is cut by ROWS of B^T: waves 0 and 1 build two rows each (xi = 1,2 and 3,4, which share
//     their sub-expressions), waves 2 and 3 one row each (xi = 0 and 5); every lane = (tile, channel quad); the pieces
//     are threaded between the MFMAs of the chunk that precedes their use;
//   * K order inside a 16-channel chunk: MFMA step s of lane group kq consumes channel 4*kq + s (one ds_read_b128 / one
//     buffer_load_b128 per lane feeds four MFMA steps).
// 72 accumulators of 4 registers do not fit the 256 AGPRs: hipcc then shuttles the overflow through AGPRs around every
// MFMA (64 v_accvgpr moves per chunk).  The last four positions of the two-block kernel therefore use the VGPR form of the
// instruction directly; their results are only ever re-read as SrcC of the next MFMA on the same registers (the
// hardware-interlocked accumulate chain, one independent MFMA in between) until the epilogue, which starts with a barrier.
__device__ __forceinline__ void mfma16_vgpr(f32x4 &acc, const float a, const float b)
{
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}


// NB = 16-channel blocks per wave: 2 (workgroup = 128 output channels) or 1 (64 channels, for the Cout = 64 layers: half
// the MFMA work per transformed tile, but still 1.6x the F(2x2) kernel there).
// HEAD (one-block variant only): the layer feeds the network's 1x1 head; see ConvArgs::head_w.
// SPLITK: the grid carries a.ksplit workgroups per tile, each walks one slice of the K chunks and writes its raw 4x4
// outputs into slab [slice][B][H][W][Cout] of a.ksplit_ws; wino_splitk_reduce (conv_wino.hip) sums the slabs in slice order
// (deterministic) and applies shift / ReLU / pooling.  For grids that cannot fill the chip (single images, deep levels).
// EXP != 0: timing-only experiment builds (MIUNET_W4_EXP, results are WRONG, never routed by default; DESIGN 4.3 "what the
// transform's role split costs"): 1 = every wave runs the one-row transform role (no role branches, half the transform
// VALU), 2 = every wave runs wave 0's two-row role (no role branches, same VALU as the heaviest wave), 3 = no forward
// transform at all, 4 = the inverse transform and the stores replaced by a checksum of the accumulators.
// EXP = 5 (correct results): the two-block kernel PERSISTENT -- one workgroup per CU walks its XCD's tiles, the raw-patch DMA of
// tile T+1 is issued before the epilogue of tile T, its U ring after it (hipcc spills 33 loop-invariant registers around the tile
// loop; reloaded once per tile, outside the K loop).
#ifndef W4_UD2
#define W4_UD2 6                              // U ring depth of the two-block kernel in positions (A/B builds: -DW4_UD2=9, 12, 18)
#endif
template <int NB, bool HEAD, bool SPLITK, int EXP = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_wino4_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                            const int m_tiles, const int nwg)
{
    constexpr int VROW = W4::VROW, VPOS = W4::VPOS, VBUF = W4::VBUF;
    constexpr int HEAD_ROW = 64 + 4;          // floats per pixel of the head's LDS tile (conflict-free b128 rows)
    static_assert(!HEAD || NB == 1, "the fused head needs every channel of a pixel in one workgroup");
    constexpr int UD = NB == 1 ? 9 : W4_UD2;  // U prefetch distance in positions (36 % UD == 0); the one-block variant has registers to spare
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const Vs = lds;                    // [2][36][16][VROW]
    float *const Raw = lds + 2 * VBUF;        // [2][18 rows][20 slots][16]: the input halo patch of one 16-channel chunk, double-buffered

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j16 = lane & 15, kq = lane >> 4;

    // ---- per-tile state.  The workgroup is persistent: it walks the tiles of its XCD's logical range with a stride of that
    // XCD's workgroup count (the 32 CUs of an XCD work on neighbouring tiles of one channel group at a time), and the loads
    // that open tile T+1 (its first two raw chunks, its first U fragments) are issued before the epilogue of tile T.
    int bx0 = 0, by0 = 0, b = 0, ncol0 = 0;
    unsigned u_voff = 0;
    __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, 0, 0x00020000);
    // stage 1: raw halo patch, one 16-channel chunk at a time, global -> registers -> LDS (4 lanes = one pixel's 64
    // bytes; zero padding, channels past Cin and dead slots through the buffer range check: voffset 0xFFFFFFFF reads zeros)
    unsigned raw_voff[W4::RAW_ITERS];
    int ks = 0;                               // K slice of this workgroup (SPLITK)
    auto setup_tile = [&](int L) {
        if constexpr (SPLITK) { ks = L % a.ksplit; L /= a.ksplit; }
        const int n_tile = L / m_tiles;
        int m = L - n_tile * m_tiles;
        const int tx = m % tiles_x; m /= tiles_x;
        const int ty = m % tiles_y;
        b = m / tiles_y;
        bx0 = tx * 16; by0 = ty * 16;
        ncol0 = n_tile * 64 * NB + 16 * NB * wave + j16;
        u_voff = (unsigned)(ncol0 * WINO4_KC + 4 * kq) * 4;
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in + (size_t)b * a.H * a.W * a.ldc), 0,
                                                    a.H * a.W * a.ldc * 4, 0x00020000);
        // the two-block kernel's persistent build has no registers for loop-invariant per-lane values across the tile loop (hipcc
        // hoists these divisions out of it and then spills them): the lane index is laundered through an empty asm, so they
        // are recomputed per tile (a few dozen VALU against a 100 k-cycle tile)
        int ln = lane;
        if constexpr (NB == 2 && EXP == 5) asm volatile("" : "+v"(ln));
#pragma unroll
        for (int s = 0; s < W4::RAW_ITERS; ++s) {
            const int g = (wave + 4 * s) * 16 + (ln >> 2);                    // linear slot of this lane in load wave + 4s
            const int py = g / W4::RAW_ROW, sl = g - py * W4::RAW_ROW;
            const int px = 4 * (sl % 5) + sl / 5;
            const int gy = by0 - 1 + py, gx = bx0 - 1 + px;
            const bool inb = g < W4::RAW_SLOTS && px < 18 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            raw_voff[s] = inb ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * (ln & 3)) * 4) : 0xFFFFFFFFu;
        }
    };
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto raw_dma_one = [&](int chunk, int buf, int s) {        // the wave's s-th load of a chunk's patch
        const int c0 = chunk * WINO4_KC;
        const bool c_ok = c0 + 4 * (lane & 3) < a.Cin;
        if (wave + 4 * s < W4::RAW_LOADS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(Raw + buf * W4::RAW_FLOATS + (wave + 4 * s) * 16 * WINO4_KC), 16,
                                                     c_ok ? raw_voff[s] : 0xFFFFFFFFu, c0 * 4, 0, 0);
    };
    auto raw_dma = [&](int chunk, int buf) {   // global -> LDS directly: no registers, no ds_write; completion = vmcnt
        const int c0 = chunk * WINO4_KC;
        const bool c_ok = c0 + 4 * (lane & 3) < a.Cin;
#pragma unroll
        for (int s = 0; s < W4::RAW_ITERS; ++s)
            if (wave + 4 * s < W4::RAW_LOADS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_ptr)(Raw + buf * W4::RAW_FLOATS + (wave + 4 * s) * 16 * WINO4_KC), 16,
                                                         c_ok ? raw_voff[s] : 0xFFFFFFFFu, c0 * 4, 0, 0);
    };

    // ---- stage 2: V = B^T d B for one 16-channel chunk; lane = (tile, channel quad), wave = row group of B^T:
    //   wave 0: xi 1, 2 = (d4 - 4 d2) +- (d3 - 4 d1)        wave 1: xi 3, 4 = (d4 - d2) +- (2 d3 - 2 d1)
    //   wave 2: xi 0    = 4 d0 - 5 d2 + d4                  wave 3: xi 5    = 4 d1 - 5 d3 + d5
    const int wrole = EXP == 1 ? 2 : EXP == 2 ? 0 : wave;      // the transform role this wave plays
    const bool two = wrole < 2;
    const int t_tile = lane >> 2, t_quad = lane & 3;
    const int row0 = two ? 1 : wrole - 2, rstep = two ? 1 : 2;
    int p_rd_off = (((4 * (t_tile >> 2) + row0) * W4::RAW_ROW + (t_tile & 3)) * 4 + t_quad) * 4;     // (laundered per tile in the persistent two-block build)
    const int p_rstride = rstep * W4::RAW_ROW * WINO4_KC;
    const int xi_a = two ? (wrole == 0 ? 1 : 3) : (wrole == 2 ? 0 : 5);
    const float c_alpha = wrole == 0 ? -4.f : -1.f, c_beta = wrole == 0 ? 1.f : 2.f;
    float *const v_wr_a = Vs + xi_a * 6 * VPOS + t_tile * VROW + 4 * (t_quad ^ v_swz(t_tile));      // + buf*VBUF + nu*VPOS; row b = + 6*VPOS
    f32x4 px_[4];                             // patch column k of this lane's rows
    f32x4 cR[2][6];                           // rows of B^T d (row b only on the two-row waves)
    f32x4 e_[4];
    auto piece_load = [&](int k, int rbuf) {
        const float *src = Raw + p_rd_off + rbuf * W4::RAW_FLOATS + ((k & 3) * 5 + (k >> 2)) * WINO4_KC;      // pixel 4*tx + k -> slot 5*(k&3) + tx + (k>>2)
        px_[0] = *reinterpret_cast<const f32x4 *>(src);
        px_[1] = *reinterpret_cast<const f32x4 *>(src + p_rstride);
        px_[2] = *reinterpret_cast<const f32x4 *>(src + 2 * p_rstride);
        px_[3] = *reinterpret_cast<const f32x4 *>(src + 3 * p_rstride);      // (the one-row waves read a fourth row they never use: patch rows 6, 7 -- cheaper than a branch per column)
    };
    auto piece_col = [&](int k) {
        const f32x4 *d = px_;
        f32x4 ra, rb;
        if (two) {
            // waves 0 and 1 run ONE instruction sequence with wave-uniform coefficients (alpha, beta) = (-4, 1) and (-1, 2):
            // ta = d3 + alpha d1, tb = d2 + alpha d0, rows ta +- beta tb.  The same values bit for bit as the literal forms (a
            // product by 1 or -1 is exact), without the role branches in front of every column (same-card A/B of a build where
            // every wave took one role: -0.28 ms of the 10.9 ms this kernel takes per step)
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
    };
    auto piece_prep = [&](int row) {
        const f32x4 *c = cR[row];
        e_[0] = c[4] - 4.f * c[2]; e_[1] = c[3] - 4.f * c[1]; e_[2] = pk_sub(c[4], c[2]); e_[3] = pk_sub(c[3], c[1]);
    };
    auto piece_store = [&](int row, float *dst, int nu) {
        const f32x4 *c = cR[row];
        const f32x4 v = nu == 0 ? 4.f * c[0] - 5.f * c[2] + c[4]
                      : nu == 1 ? e_[0] + e_[1]
                      : nu == 2 ? pk_sub(e_[0], e_[1])
                      : nu == 3 ? e_[2] + 2.f * e_[3]
                      : nu == 4 ? e_[2] - 2.f * e_[3]
                                : 4.f * c[1] - 5.f * c[3] + c[5];
        *reinterpret_cast<f32x4 *>(dst + nu * VPOS) = v;
    };
    // the transform of one chunk as 23 numbered pieces (0-5 load, 6-11 column pass, 12 prep a, 13-18 row a, 19 prep b ...)
    auto transform_all = [&](int rbuf, int buf) {
        if constexpr (EXP == 3) return;
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

    // ---- MFMA role: wave w, channels n0 + 16 NB w .. (NB 16-column blocks), all 36 positions
    const unsigned u_pos_bytes = (unsigned)a.CoutPad * WINO4_KC * 4;
    const int all_chunks = (a.Cin + WINO4_KC - 1) / WINO4_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk4), 0, (int)(all_chunks * 36 * u_pos_bytes), 0x00020000);
    auto u_load = [&](int chunk, int p, int blk) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff + blk * 16 * WINO4_KC * 4,
                                                                              (chunk * 36 + p) * u_pos_bytes, 0));
    };
    const float *const v_rd = Vs + j16 * VROW + 4 * (kq ^ v_swz(j16));                       // + buf*VBUF + pos*VPOS

    int c_begin = 0, nchunks = all_chunks;    // this workgroup's K range [c_begin, nchunks)
    // SPLITK: raw partial sums into this slice's slab [B][H][W][Cout]; shift, ReLU and pooling belong to the reduce kernel
    const float relu_lo = (a.relu && !SPLITK) ? 0.f : -3.402823466e+38f;
    const int Hp = a.H >> 1, Wp = a.W >> 1;
    const bool do_pool = !SPLITK && a.pool_out != nullptr;
    const int ld_e = SPLITK ? a.Cout : a.ldo, co_e = SPLITK ? 0 : a.co_off;
    const unsigned pix_bytes = (unsigned)ld_e * 4, row_bytes = (unsigned)a.W * pix_bytes;
    const unsigned ppix_bytes = (unsigned)a.pool_ld * 4, prow_bytes = (unsigned)Wp * ppix_bytes;
    f32x4 u[UD][NB];
    // the loads that open a tile: the raw patches of chunks 0 and 1 straight into LDS, then the U ring
    auto open_tile = [&]() {
        if constexpr (SPLITK) {
            const int per_slice = (all_chunks + a.ksplit - 1) / a.ksplit;
            c_begin = ks * per_slice;
            nchunks = c_begin + per_slice < all_chunks ? c_begin + per_slice : all_chunks;
        }
        raw_dma(c_begin, 0);
        raw_dma(c_begin + 1, 1);
        if constexpr (!((NB == 2 && EXP == 5) && !SPLITK)) {
#pragma unroll
            for (int p = 0; p < UD; ++p)
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) u[p][blk] = u_load(c_begin, p, blk);
        }
    };
    auto open_tile_u = [&]() {
#pragma unroll
        for (int p = 0; p < UD; ++p)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) u[p][blk] = u_load(c_begin, p, blk);
    };

    // PERSIST (one-block variant): this workgroup's share of its XCD's logical tile range (the bijective XCD remap of
    // kernel_common.h, walked with a stride).  The two-block variant has no registers left to hold a second tile's opening
    // loads across its epilogue: one tile per workgroup there (the loop below runs once and folds away).
    constexpr bool PERSIST = (NB == 1 || EXP == 5) && !SPLITK;
    constexpr bool U_LATE = PERSIST && NB == 2;      // no registers for a second tile's U ring across the epilogue
    const int G = gridDim.x, xcd = blockIdx.x & 7;
    const int slot = PERSIST ? (int)(blockIdx.x >> 3) : 0;
    const int slots = PERSIST ? (G >> 3) + (xcd < (G & 7) ? 1 : 0) : 1;
    const int q_ = nwg >> 3, r_ = nwg & 7;
    const int t_start = PERSIST ? ((xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) : xcd_remap(blockIdx.x, nwg);
    const int t_count = PERSIST ? q_ + (xcd < r_ ? 1 : 0) : 1;
    if (slot < t_count) {
        setup_tile(t_start + slot);
        open_tile();
        if constexpr (U_LATE) open_tile_u();
    }
    for (int tt = slot; tt < t_count; tt += slots) {
    if constexpr (NB == 2 && EXP == 5) asm volatile("" : "+v"(p_rd_off));     // recompute the first transform's 24 LDS addresses per tile: no registers to keep them
    f32x4 acc[36][NB];
#pragma unroll
    for (int p = 0; p < 36; ++p)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            // position (xi, nu) = (1, 1) starts at the shift: A^T e_1 e_1^T A is the all-ones tile, so the bias add is free
            const float init = (!SPLITK && p == 7 && ncol0 + 16 * blk < a.Cout) ? a.bias[ncol0 + 16 * blk] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][blk][r] = init;
        }
    // prologue: the raw patches of chunks 0 and 1 are landing in LDS.  Issue order was chunk 0, chunk 1, U ring, so the
    // first transform only waits for chunk 0 (everything older than the RAW_ITERS + NB*UD youngest loads) and chunk 1 lands
    // under it.  The persistent variant also has the previous tile's stores in the queue (issued after those loads, vmcnt
    // does not order stores against loads): it waits for everything.
    if constexpr (PERSIST) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    } else {
        constexpr int N1 = W4::RAW_ITERS - 1 + NB * UD;      // wave 3 issues one load fewer per chunk: be exact for it
        __builtin_amdgcn_s_waitcnt(0x0F70 | (N1 & 15) | ((N1 >> 4) << 14));
    }
    __syncthreads();
    transform_all(0, 0);
    if constexpr (!PERSIST) __builtin_amdgcn_s_waitcnt(0x0F70 | ((NB * UD) & 15) | (((NB * UD) >> 4) << 14));
    __syncthreads();

    // Per chunk c, around its 288 MFMAs (one barrier per chunk):
    //   position 0        : LDS-DMA loads of chunk c+2's raw patch into Raw[c & 1] (whose last reader, the transform of
    //                       chunk c, ran during chunk c-1)
    //   positions 2 - 22  : the transform of chunk c+1 (Raw[(c+1) & 1] -> V[(c+1) & 1]) in pieces between the MFMAs
    //   every position    : the V fragment one position ahead, the U ring six positions ahead
    for (int chunk = c_begin; chunk < nchunks; ++chunk) {
        const int nxt = (chunk + 1 < nchunks) ? chunk + 1 : chunk;          // the last chunk prefetches itself: straight-line code
        const int par = (chunk - c_begin) & 1;                              // buffer parity of this chunk
        const int rbuf = par ^ 1;
        const float *vb = v_rd + par * VBUF;
        float *const wr = v_wr_a + (par ^ 1) * VBUF;
        f32x4 av = *reinterpret_cast<const f32x4 *>(vb);
#pragma unroll
        for (int p = 0; p < 36; ++p) {
            f32x4 avn = av;
            if (p + 1 < 36) avn = *reinterpret_cast<const f32x4 *>(vb + (p + 1) * VPOS);   // V fragment one position ahead
            // one load every fourth position, not a burst: 16 line misses at a time keep the VMEM queue moving, so the U loads
            // issued behind each of them are delayed by less than the U ring covers
            if (p % 4 == 0 && p / 4 < W4::RAW_ITERS) raw_dma_one(chunk + 2, par, p / 4);
            if (EXP != 3 && p == 2) piece_load(0, rbuf);
            __builtin_amdgcn_sched_barrier(0);        // ... issued BEFORE this position's MFMAs (hipcc would sink them to their use)
            f32x4 bv[NB];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) bv[blk] = u[p % UD][blk];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    if (NB == 2 && p >= 32) mfma16_vgpr(acc[p][blk], av[s], bv[blk][s]);
                    else acc[p][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[blk][s], acc[p][blk], 0, 0, 0);
                }
                if (EXP != 3 && s == 1) {
                    if (p >= 3 && p < 9) piece_col(p - 3);
                    if (p == 9) piece_prep(0);
                    if (p >= 10 && p < 16) piece_store(0, wr, p - 10);
                    if (two) {
                        if (p == 16) piece_prep(1);
                        if (p >= 17 && p < 23) piece_store(1, wr + 6 * VPOS, p - 17);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (EXP != 3 && s == 2 && p >= 3 && p < 8) {      // the next patch column, once the column pass above has consumed this one
                    piece_load(p - 2, rbuf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int pn = p + UD;                                                              // refill the ring slot
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) u[p % UD][blk] = u_load(pn < 36 ? chunk : nxt, pn % 36, blk);
            av = avn;
            __builtin_amdgcn_sched_barrier(0);
        }
        // the DMA of position 0 is older than every U load of this chunk; NB*UD of those are still allowed in flight
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((NB * UD) & 15) | (((NB * UD) >> 4) << 14));
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in-lane on all four tiles of a lane at once (the accumulator's four registers = tile
    // columns r = 0..3 of tile row kq, lane = channel), ReLU, 4x4 stores (+ the 2x2 pooled maxima).  The shift was the
    // initial value of position (1,1), whose inverse transform is the all-ones tile.  Stores are buffer stores on a
    // per-image descriptor: one per-lane byte offset for the whole tile row, the pixel displacement in the scalar offset;
    // pixels past the image edge and masked channels get voffset 0xFFFFFFFF, which the range check drops.
    if constexpr (EXP == 4) {
        f32x4 sum = f32x4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int p = 0; p < 36; ++p)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) sum += acc[p][blk];
        if (sum.x + sum.y + sum.z + sum.w == 12345.678f) a.out[tid] = sum.x;
        return;
    }
    const int e_b = b, e_by0 = by0, e_bx0 = bx0, e_ncol0 = ncol0;
    if constexpr (PERSIST) {
        if (tt + slots < t_count) {           // open the next tile: its loads fly during this tile's epilogue
            setup_tile(t_start + tt + slots);
            open_tile();
        }
    }
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (SPLITK ? a.ksplit_ws + (size_t)ks * a.B * a.H * a.W * a.Cout : a.out) + (size_t)e_b * a.H * a.W * ld_e, 0,
        a.H * a.W * ld_e * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t pool_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        do_pool ? a.pool_out + (size_t)e_b * Hp * Wp * a.pool_ld : a.out, 0, do_pool ? Hp * Wp * a.pool_ld * 4 : 0, 0x00020000);
    const int oy = e_by0 + 4 * kq;
    auto epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const int ncol = e_ncol0 + 16 * blk;
            const bool n_ok = ncol < a.Cout;
            const unsigned vbase = n_ok ? (unsigned)((oy * a.W + e_bx0) * ld_e + co_e + ncol) * 4 : 0xFFFFFFFFu;
            const unsigned pbase = n_ok ? (unsigned)(((oy >> 1) * Wp + (e_bx0 >> 1)) * a.pool_ld + ncol) * 4 : 0xFFFFFFFFu;
            unsigned vcol[4][4], pcol[4][2];          // edge workgroups: per-column offsets (dead columns -> dropped stores)
            if constexpr (!INTERIOR) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) vcol[r][k] = e_bx0 + 4 * r + k < a.W ? vbase : 0xFFFFFFFFu;
                    pcol[r][0] = e_bx0 + 4 * r + 1 < a.W ? pbase : 0xFFFFFFFFu;
                    pcol[r][1] = e_bx0 + 4 * r + 3 < a.W ? pbase : 0xFFFFFFFFu;
                }
            }
            // two tile columns (r = 2 h2, 2 h2 + 1) at a time: the same packed instructions as four at once, half the live
            // temporaries (the epilogue runs with every accumulator still resident)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                auto half = [&](const f32x4 &v) { return h2 ? v.hi : v.lo; };
                f32x2 t[4][6];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) {
                    const f32x2 m0 = half(acc[nu][blk]), m1 = half(acc[6 + nu][blk]), m2 = half(acc[12 + nu][blk]),
                                m3 = half(acc[18 + nu][blk]), m4 = half(acc[24 + nu][blk]), m5 = half(acc[30 + nu][blk]);
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
                                lds[((4 * kq + i) * 16 + 4 * r + k) * HEAD_ROW + ncol] = v;
                                continue;
                            }
                            unsigned voff = vbase;
                            if constexpr (!INTERIOR) voff = row_ok ? vcol[r][k] : 0xFFFFFFFFu;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, voff,
                                                                  i * row_bytes + (4 * r + k) * pix_bytes, 0);
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
                                                                      (i >> 1) * prow_bytes + (2 * r) * ppix_bytes, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p1), pool_rsrc, v1,
                                                                      (i >> 1) * prow_bytes + (2 * r + 1) * ppix_bytes, 0);
                            }
                        }
                    }
                }
            }
        }
    };
    if (HEAD || (e_by0 + 16 <= a.H && e_bx0 + 16 <= a.W)) epilogue(std::true_type{});   // workgroup-uniform: no per-pixel predicates
    else epilogue(std::false_type{});
    if constexpr (HEAD) {
        // 1x1 head + argmax on the tile: the four waves hold 16 channels each, so the tile crossed LDS above
        // ([256 pixels][64 + 4 pad], in the V region, which is dead until the next tile's first transform -- behind the
        // next prologue's barrier); thread = pixel, a fixed summation order, first-max-wins argmax (src/process.cpp:158-170).
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
        const int py = e_by0 + (tid >> 4), px = e_bx0 + (tid & 15);
        if (py < a.H && px < a.W) {
            const size_t hw = (size_t)a.H * a.W, pin = (size_t)py * a.W + px;
            float best = -3.402823466e+38f;
            int idx = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < a.head_classes) {
                    if (a.head_logits != nullptr) a.head_logits[((size_t)e_b * a.head_classes + k) * hw + pin] = d[k];
                    if (d[k] > best) { best = d[k]; idx = k; }
                }
            }
            a.head_labels[(size_t)e_b * hw + pin] = (uint8_t)idx;
        }
    }
    if constexpr (U_LATE) { if (tt + slots < t_count) open_tile_u(); }
    }   // persistent tile loop
}

template <int NB, bool HEAD>
static hipError_t launch_wino4_cfg(const ConvArgs &a0, hipStream_t s)
{
    ConvArgs a = a0;
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 64 * NB - 1) / (64 * NB);
    const int nwg = m_tiles * n_tiles;
    // split K when the (tile, channel) grid alone leaves most CUs idle: every slice keeps at least four chunks
    a.ksplit = 1;
    const int chunks = (a.Cin + WINO4_KC - 1) / WINO4_KC;
    if (!HEAD && a.ksplit_ws != nullptr && nwg <= 128 && chunks >= 8 && a.Cout % 4 == 0 && a.ldo % 4 == 0 && a.co_off % 4 == 0) {
        int ks = 256 / nwg;
        if (ks > 8) ks = 8;
        if (ks > chunks / 4) ks = chunks / 4;
        while (ks > 1 && (size_t)ks * a.B * a.H * a.W * a.Cout * sizeof(float) > a.ksplit_ws_bytes) --ks;
        while (ks > 1 && (ks - 1) * ((chunks + ks - 1) / ks) >= chunks) --ks;      // no empty slice
        a.ksplit = ks;
    }
    if (a.ksplit > 1) {
        auto kern = conv3x3_wino4_f32<NB, false, true>;
        if (hipError_t e = ensure_dynamic_lds(kern, W4::LDS_BYTES); e != hipSuccess) return e;
        const int grid = nwg * a.ksplit;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4::LDS_BYTES, s, a, tiles_x, tiles_y, m_tiles, grid);
        return launch_wino_splitk_reduce(a, s);
    }
#ifdef MIUNET_EXPERIMENTS                              // lab build only (libmiunet_exp.so, tools/dev/ab*.sh): never in libmiunet.so
    if constexpr (NB == 2 && !HEAD) {                  // timing-only experiment builds of the two-block kernel (see the kernel's EXP)
        static const int exp = [] { const char *e = getenv("MIUNET_W4_EXP"); return e ? atoi(e) : 0; }();
        if (exp >= 1 && exp <= 5) {
            auto ke = exp == 1 ? conv3x3_wino4_f32<2, false, false, 1> : exp == 2 ? conv3x3_wino4_f32<2, false, false, 2>
                    : exp == 3 ? conv3x3_wino4_f32<2, false, false, 3> : exp == 4 ? conv3x3_wino4_f32<2, false, false, 4>
                    : conv3x3_wino4_f32<2, false, false, 5>;
            if (hipError_t e = ensure_dynamic_lds(ke, W4::LDS_BYTES); e != hipSuccess) return e;
            const int cus = routing_of(a).cus;
            const int ge = (exp == 5 && nwg > cus) ? cus : nwg;
            hipLaunchKernelGGL(ke, dim3(ge), dim3(256), W4::LDS_BYTES, s, a, tiles_x, tiles_y, m_tiles, nwg);
            return hipGetLastError();
        }
    }
#endif
    auto kern = conv3x3_wino4_f32<NB, HEAD, false>;
    if (hipError_t e = ensure_dynamic_lds(kern, W4::LDS_BYTES); e != hipSuccess) return e;
    // one-block variant: persistent, one workgroup per CU (144 KB of LDS each); two-block variant: one tile per workgroup
    const int cus = routing_of(a).cus;
    const int grid = (NB == 1 && nwg > cus) ? cus : nwg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4::LDS_BYTES, s, a, tiles_x, tiles_y, m_tiles, nwg);
    return hipGetLastError();
}

// one-block cases whose grid fills the chip twice over go to the two-workgroups-per-CU kernel (conv_wino4s.hip); small
// grids keep the persistent kernel and its split-K
// (MIUNET_WINO4S = 0: never; 2: every one-block case whatever its grid -- for parity tests on small shapes)
bool conv3x3_wino4_runs_staged(const ConvArgs &a)
{
    const Routing rt = routing_of(a);
    const int staged = rt.wino4s;
    const int rem = a.Cout % 128;
    const long long wg1 = (long long)((a.W + 15) / 16) * ((a.H + 15) / 16) * a.B * ((a.Cout + 63) / 64);
    // ... and wider layers whose K loop is at most four chunks (down1.c1, 64 -> 128: 0.521 -> 0.488 ms); with eight chunks and
    // more the two-block kernel's shared forward transform wins (measured on every such layer: 7-20 % slower staged)
    // ... unless the assembly kernel can take the layer (down1.c1, 64 -> 128: 0.459 -> 0.415 ms same card, profiles/r04_ab_asm_routing.txt;
    // MIUNET_WINO4_ASM = 3 keeps such a layer on the staged kernel -- A/B switch)
    const bool asm_takes_k4 = rt.wino4_asm != 0 && rt.wino4_asm != 3 && a.Cin == 64 && conv3x3_wino4a_shape_ok(a);
    const bool one_block = a.head_w != nullptr || !(a.Cout >= 128 && (rem == 0 || rem > 64)) || (a.Cin <= 64 && !asm_takes_k4);
    // a.ksplit_ws == nullptr is the batch-invariant mode (MIUNET_SPLITK=0): there the choice must not depend on the batch
    return one_block && (staged == 2 || (staged == 1 && (wg1 >= 2 * rt.cus || a.ksplit_ws == nullptr)));
}

// the hand-scheduled persistent form of the two-block kernel (csrc/wino4_asm.cpp) takes the layer when its shape fits the assembly's
// contract and the grid is not one the launcher below would split K for (single images, deep levels: those keep the hipcc kernel)
bool conv3x3_wino4_runs_asm(const ConvArgs &a)
{
    if (routing_of(a).wino4_asm == 0 || !conv3x3_wino4a_shape_ok(a) || conv3x3_wino4_runs_staged(a)) return false;
    const long long nwg = (long long)(a.W / 16) * (a.H / 16) * a.B * (a.Cout / 128);
    const bool split_k = a.ksplit_ws != nullptr && nwg <= 128 && a.Cin / WINO4_KC >= 8;
    return !split_k;
}

// ... and conv3x3_wino4b for the layers conv3x3_wino4s would take (64 output channels per workgroup, big grids) when the shape fits
bool conv3x3_wino4_runs_asm_b(const ConvArgs &a)
{
    return routing_of(a).wino4_asm_b != 0 && conv3x3_wino4b_shape_ok(a) && conv3x3_wino4_runs_staged(a) && a.Cout % 128 != 0;
}

hipError_t launch_conv3x3_wino4(const ConvArgs &a, hipStream_t s)
{
    if (a.wpk4 == nullptr || a.Cin % 4 || a.ldc % 4 || a.CoutPad % NPAD) return hipErrorInvalidValue;
    // 128 output channels per workgroup when Cout fills them; 64 for the Cout = 64 layers (and any Cout % 128 in (0, 64])
    const int rem = a.Cout % 128;
    if (conv3x3_wino4_runs_asm_b(a)) return launch_conv3x3_wino4b(a, s);
    if (conv3x3_wino4_runs_staged(a)) return launch_conv3x3_wino4s(a, s);
    if (a.head_w != nullptr) {
        if (a.Cout > 64 || a.head_classes < 1 || a.head_classes > 4 || a.pool_out != nullptr || a.head_labels == nullptr)
            return hipErrorInvalidValue;
        return launch_wino4_cfg<1, true>(a, s);
    }
    if (conv3x3_wino4_runs_asm(a)) return launch_conv3x3_wino4a(a, s);
    if (a.Cout >= 128 && (rem == 0 || rem > 64)) return launch_wino4_cfg<2, false>(a, s);
    return launch_wino4_cfg<1, false>(a, s);
}

}  // namespace miunet
