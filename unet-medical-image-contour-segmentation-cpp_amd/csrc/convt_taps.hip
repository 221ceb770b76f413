// convt_taps.hip -- the 2x2 / stride-2 transposed convolution as four GEMMs that share one A operand, gfx950 only.
#include <cstdlib>
#include <type_traits>

#include "kernel_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// out[b][2y+dy][2x+dx][co] = bias[co] + sum_ci x[b][y][x][ci] * w[ci][co][dy][dx]: for each of the four taps a plain GEMM
// M = pixels, K = Cin, N = Cout, all four over the SAME pixels.  The direct kernel (conv_direct.hip, TAPS = 1) treats it
// as one GEMM with N = 4*Cout and stages A and B through LDS every 16 channels: 32 MFMAs per barrier, 58-77 % of the
// fp32 MFMA peak.  Here the structure is the one of the Winograd kernels (conv_wino.hip) with the taps in the role of
// the positions:
//   * wave w = tap w (dy = w >> 1, dx = w & 1): it owns MB x 32 input pixels (MB image rows x 32 columns) x NBK x 32
//     output channels = MB * NBK = 16 accumulators of 32x32 (256 registers, one wave per SIMD);
//   * B (the tap's weights) never touches LDS: packed [Cin/8][tap][Cout^128][8], every lane loads its 16-byte fragment
//     with a scalar-offset buffer load one K-chunk (64 MFMAs) ahead into the register the previous chunk released;
//   * A (the input pixels, no transform) is staged 32 channels at a time (whole 128-byte lines) into a double-buffered
//     LDS image [pixel][32 + 4 pad], so there is ONE barrier per 256 MFMAs, and all four waves read the same fragments;
//   * per 8-channel chunk and wave: 64 MFMAs, MB ds_read_b128, NBK buffer loads -- 10 to 17 other vector instructions
//     (the fp32 MFMA shares its issue slots with them);
//   * epilogue: + bias, pixel-shuffle store (32 lanes = 32 consecutive channels = 128 bytes) into the upper half of the
//     concat buffer, buffer stores with scalar pixel offsets.
constexpr int CT_KC = 8;
constexpr int CT_SC = 32;

template <int MB>
struct CTGeom {
    static constexpr int PIX = 32 * MB;
    static constexpr int A_P = CT_SC + 4;                 // padded floats per pixel (2-way worst case on ds_read_b128)
    static constexpr int A_FLOATS = PIX * A_P;
    static constexpr size_t LDS_BYTES = sizeof(float) * 2 * A_FLOATS;
};

// WPS = waves per SIMD the kernel is built for: 1 = sixteen 32x32 accumulators per wave (256 registers, one workgroup per
// CU), 2 = eight (two workgroups per CU: one runs its MFMAs while the other waits for its first loads or stores its tile)
template <int MB, int NBK, int WPS>
__global__ __launch_bounds__(256, WPS) void convT2x2_taps_f32(const ConvArgs a, const int tiles_x, const int tiles_y,
                                                              const int m_tiles, const int nwg, const int cpad)
{
    static_assert(MB * NBK * WPS <= 16, "sixteen (WPS = 1), eight (2) or at most four (4) 32x32 accumulators per wave");
    using G = CTGeom<MB>;
    constexpr int A_P = G::A_P;
    constexpr int CPS = CT_SC / CT_KC;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tap = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int L = xcd_remap(blockIdx.x, nwg);
    const int n_tile = L / m_tiles;
    int m = L - n_tile * m_tiles;
    const int tx = m % tiles_x; m /= tiles_x;
    const int ty = m % tiles_y;
    const int b = m / tiles_y;
    const int x0 = tx * 32, y0 = ty * MB, n0 = n_tile * 32 * NBK;

    // ---- A staging: MB rows x 32 columns x 32 channels per super-chunk; 8 lanes = one pixel's 128 bytes
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in + (size_t)b * a.H * a.W * a.ldc), 0, a.H * a.W * a.ldc * 4, 0x00020000);
    unsigned raw_voff[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) {
        const int pix = (tid >> 3) + 32 * s;              // row s, column tid >> 3
        const int gy = y0 + s, gx = x0 + (pix & 31);
        raw_voff[s] = (gy < a.H && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.ldc + 4 * (tid & 7)) * 4) : 0xFFFFFFFFu;
    }
    float *const raw_wr = lds + (tid >> 3) * A_P + 4 * (tid & 7);                // + buf*A_FLOATS + s*32*A_P
    f32x4 raw_reg[MB];
    auto raw_load = [&](int super) {
        const int c0 = super * CT_SC;
        const bool c_ok = c0 + 4 * (tid & 7) < a.Cin;
#pragma unroll
        for (int s = 0; s < MB; ++s)
            raw_reg[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, c_ok ? raw_voff[s] : 0xFFFFFFFFu, c0 * 4, 0));
    };
    auto raw_store = [&](int buf) {
#pragma unroll
        for (int s = 0; s < MB; ++s) *reinterpret_cast<f32x4 *>(raw_wr + buf * G::A_FLOATS + s * 32 * A_P) = raw_reg[s];
    };

    // ---- B fragments of this wave's tap
    const int ncol0 = n0 + li;
    const unsigned u_tap_bytes = (unsigned)cpad * CT_KC * 4;
    const unsigned u_voff = (unsigned)(ncol0 * CT_KC + 4 * lh) * 4;
    const int nsuper = (a.Cin + CT_SC - 1) / CT_SC;
    const int nchunks = nsuper * CPS;                     // the packed weights are zero-padded to whole super-chunks
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wpk4), 0, (int)(nchunks * 4 * u_tap_bytes), 0x00020000);
    auto u_load = [&](int chunk, int nb) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_voff + nb * 32 * CT_KC * 4,
                                                                              (chunk * 4 + tap) * u_tap_bytes, 0));
    };
    const float *const a_rd = lds + li * A_P + 4 * lh;                          // + buf*A_FLOATS + mb*32*A_P + j*8

    f32x16 acc[MB][NBK];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    f32x4 u[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) u[nb] = u_load(0, nb);
    raw_load(0);
    raw_store(0);
    __syncthreads();

    f32x4 av[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const f32x4 *>(a_rd + mb * 32 * A_P);
    auto super_chunk = [&](const int S) {
        const int buf = S & 1;
        if (S + 1 < nsuper) raw_load(S + 1);              // registers; they land during this super-chunk's 256 MFMAs
#pragma unroll
        for (int j = 0; j < CPS; ++j) {
            const int chunk = S * CPS + j;
            const int nxt = chunk + 1 < nchunks ? chunk + 1 : chunk;
            f32x4 avn[MB];
            if (j + 1 < CPS) {                            // A fragments of the next chunk of this super-chunk
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    avn[mb] = *reinterpret_cast<const f32x4 *>(a_rd + buf * G::A_FLOATS + mb * 32 * A_P + (j + 1) * CT_KC);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                const f32x4 bv = u[nb];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][s], bv[s], acc[mb][nb], 0, 0, 0);
                u[nb] = u_load(nxt, nb);                  // refill a full chunk ahead
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + 1 < CPS) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = avn[mb];
            }
        }
        if (S + 1 < nsuper) raw_store(buf ^ 1);
        __syncthreads();
        if (S + 1 < nsuper) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const f32x4 *>(a_rd + (buf ^ 1) * G::A_FLOATS + mb * 32 * A_P);
        }
    };
    // at least one super-chunk: as a do-while there is no zero-trip path whose accumulators hipcc merges with the loop's
    // (150 register moves per tile); the 256-accumulator shapes spill in that form and keep the plain loop
    if constexpr (WPS == 1) {
        for (int S = 0; S < nsuper; ++S) super_chunk(S);
    } else {
        int S = 0;
        do super_chunk(S); while (++S < nsuper);
    }

    // ---- epilogue: + bias, pixel-shuffle store.  Lane = channel (li) of block nb, register r = pixel column
    // (r & 3) + 8 (r >> 2) + 4 lh of image row y0 + mb; output pixel (2y + dy, 2x + dx).
    const int dy = tap >> 1, dx = tap & 1;
    const int W2 = 2 * a.W;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + (size_t)b * 4 * a.H * a.W * a.ldo, 0, 4 * a.H * a.W * a.ldo * 4, 0x00020000);
    const unsigned pix_bytes = (unsigned)a.ldo * 4;
    // this lane's biases, loaded BEFORE the first store: a load issued behind stores makes hipcc wait for vmcnt(0), i.e. for
    // every store in flight, once per channel block
    float biasv[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) biasv[nb] = ncol0 + 32 * nb < a.Cout ? a.bias[ncol0 + 32 * nb] : 0.f;
    auto epilogue = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
            const int ncol = ncol0 + 32 * nb;
            const bool n_ok = ncol < a.Cout;
            const float bias = biasv[nb];
            // per-lane part: its column group (4 lh), the tap's (dy, dx) displacement, the channel
            const unsigned vbase = n_ok ? (unsigned)((((2 * y0 + dy) * W2 + 2 * (x0 + 4 * lh) + dx) * a.ldo + a.co_off + ncol) * 4) : 0xFFFFFFFFu;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bool row_ok = INTERIOR || y0 + mb < a.H;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xr = (r & 3) + 8 * (r >> 2);
                    unsigned voff = vbase;
                    if constexpr (!INTERIOR) voff = (row_ok && x0 + xr + 4 * lh < a.W) ? vbase : 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[mb][nb][r] + bias), out_rsrc, voff,
                                                          (2 * mb * W2 + 2 * xr) * pix_bytes, ST_AUX);
                }
            }
        }
    };
    if (x0 + 32 <= a.W && y0 + MB <= a.H) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

template <int MB, int NBK, int WPS = 1>
static hipError_t launch_taps_cfg(const ConvArgs &a, int cpad, hipStream_t s)
{
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + MB - 1) / MB;
    const int m_tiles = tiles_x * tiles_y * a.B;
    const int n_tiles = (a.Cout + 32 * NBK - 1) / (32 * NBK);
    const int nwg = m_tiles * n_tiles;
    auto kern = convT2x2_taps_f32<MB, NBK, WPS>;
    constexpr size_t lds = CTGeom<MB>::LDS_BYTES;
    if (hipError_t e = ensure_dynamic_lds(kern, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a, tiles_x, tiles_y, m_tiles, nwg, cpad);
    return hipGetLastError();
}

// The shape the launcher gives a layer: {MB image rows, NBK 32-channel blocks, waves per SIMD}.  Whole batches get the
// measured-best shapes of DESIGN.md 4.4; when those leave the chip short of ~one workgroup per CU (single images, the deep
// levels: 32 x 32 pixels x 512 channels is 64 of the large tiles) the tile shrinks to one row x 128 or 64 channels with
// four workgroups per CU, which is latency cover for a K loop of 1024 channels rather than operand reuse.
struct TapsShape { int mb, nbk, wps; };
static long long taps_grid(const ConvArgs &a, const TapsShape &t)
{
    return (long long)((a.W + 31) / 32) * ((a.H + t.mb - 1) / t.mb) * a.B * ((a.Cout + 32 * t.nbk - 1) / (32 * t.nbk));
}
static TapsShape taps_shape(const ConvArgs &a)
{
#ifdef MIUNET_EXPERIMENTS                              // lab build only: the product library has one route per shape
    static const int mode = [] { const char *e = getenv("MIUNET_CONVT_WPS"); return e ? atoi(e) : 2; }();
#else
    constexpr int mode = 2;
#endif
    if (mode != 2) {
        if (a.Cout > 256) return { 1, 16, 1 };
        if (a.Cout > 128) return { 2, 8, 1 };
        if (a.Cout > 64) return { 4, 4, 1 };
        return { 8, 2, 1 };
    }
    const TapsShape big = a.Cout > 256 ? TapsShape{ 1, 8, 2 } : a.Cout > 64 ? TapsShape{ 2, 4, 2 } : TapsShape{ 4, 2, 2 };
    // MIUNET_CONVT_SMALL = 0: never shrink (parity tests of the large shapes on small inputs)
    if (taps_grid(a, big) >= 192 || !routing_of(a).convt_small) return big;
    if (a.Cout > 64 && taps_grid(a, { 1, 4, 4 }) >= 192) return { 1, 4, 4 };
    return { 1, 2, 4 };
}
long long convT_taps_grid(const ConvArgs &a) { return taps_grid(a, taps_shape(a)); }

// a.wpk4 = the per-tap packing [Cin^32 / 8][4 taps][convT_taps_cpad(Cout)][8]; everything else as launch_convT2x2_mfma
hipError_t launch_convT2x2_taps(const ConvArgs &a, hipStream_t s)
{
    if (a.wpk4 == nullptr || a.Cin % 4 || a.ldc % 4) return hipErrorInvalidValue;
    const int cpad = convT_taps_cpad(a.Cout);
    // whole batches: half-size tiles, eight accumulators per wave, TWO workgroups per CU (measured 3-10 % faster than the
    // sixteen-accumulator tiles on all four layers at batch 16: up1.t 0.501 -> 0.488 ms, up2.t 0.511 -> 0.499, up3.t 0.576 ->
    // 0.529, up4.t 0.706 -> 0.630; the other eight-accumulator shapes -- <1,8> at Cout 256, <4,2> at 128, and four workgroups
    // per CU with <2,2> at 64 -- were each within 2 % or slower); MIUNET_CONVT_WPS=1 keeps the large tiles
    const TapsShape t = taps_shape(a);
    const int key = t.mb * 100 + t.nbk * 10 + t.wps;
    switch (key) {
    case 182: return launch_taps_cfg<1, 8, 2>(a, cpad, s);
    case 242: return launch_taps_cfg<2, 4, 2>(a, cpad, s);
    case 422: return launch_taps_cfg<4, 2, 2>(a, cpad, s);
    case 144: return launch_taps_cfg<1, 4, 4>(a, cpad, s);
    case 124: return launch_taps_cfg<1, 2, 4>(a, cpad, s);
    case 281: return launch_taps_cfg<2, 8>(a, cpad, s);
    case 441: return launch_taps_cfg<4, 4>(a, cpad, s);
    case 821: return launch_taps_cfg<8, 2>(a, cpad, s);
    default: return launch_taps_cfg<1, 16>(a, cpad, s);
    }
}

}  // namespace miunet
