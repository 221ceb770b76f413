// kernels.h -- launch interface of the gfx950 kernels (conv_direct.hip, conv_lp.hip, conv_wino.hip, layers_mem.hip, image_stages.hip).  Internal to libmiunet.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace miunet {

// Channel counts of packed weights are padded to these granules with zeros.
constexpr int KC = 16;        // input channels staged per K-chunk
constexpr int NPAD = 128;     // packed Cout granule (covers both BN = 64 and BN = 128 tiles)

// Kernel-routing switches (the MIUNET_* A/B variables) and the device's CU count, resolved ONCE per engine handle at
// mi_unet_create and carried in every launch's arguments: a launch never calls getenv or touches shared tables, so cloned
// contexts launching from several threads share nothing mutable, and an engine's routing cannot change under it.
// A default-constructed Routing is unresolved: routing_of() then reads the environment at the call (mi_unet_layer_debug).
struct Routing {
    bool resolved = false;
    int cus = 256;            // compute units of the launch device
    int lp2 = 1;              // MIUNET_LP2: 0 never, 1 default thresholds, 2 every Cout % 128 == 0 layer
    int lpr = 1;              // MIUNET_LPR: 0 never, 1 when the tiles fill the chip four times over, 2 whatever the grid
    int lpr_rb = 2;           // MIUNET_LPR_RB: 1 = 8-row tiles for every shape
    int lprk = 1;             // MIUNET_LPRK: the 128 -> 64 K-split resident-weight kernel (conv_lprk.hip), as lpr
    int convt_lpr = 1;        // MIUNET_CONVT_LPR: as lpr
    int wino4s = 1;           // MIUNET_WINO4S: 0 never, 1 grids that fill the chip twice over, 2 every one-block case
    int fuse_first = 1;       // MIUNET_FUSE_FIRST=0: the first layer stays a kernel of its own (A/B, parity checks)
    int wino4_asm_b = 1;      // MIUNET_WINO4_ASM_B=0: the 64-channel layers stay on conv3x3_wino4s
    int wino4_asm = 1;        // MIUNET_WINO4_ASM: 0 never, 1 the hand-scheduled persistent two-block kernel for the shapes it takes
    bool convt_small = true;  // MIUNET_CONVT_SMALL=0: the per-tap transposed conv never shrinks its tile
    bool first_mfma = true;   // MIUNET_FIRST_MFMA=0: the 16-bit pipelines' first layer stays on the VALU kernel
    static Routing from_env();            // reads the environment and the current device's properties (engine.cpp)
};

// One implicit-GEMM launch: conv3x3 (taps = 9) or the 2x2-stride-2 transposed conv viewed as a 1-tap GEMM with
// N = 4*Cout (taps = 1).  Activations are NHWC fp32; `ldc`/`ldo` are the channel strides of the input / output pixel
// (so a tensor can live in one half of a concat buffer), `co_off` the first output channel written.
struct ConvArgs {
    const float *in;      // [B][H][W][ldc]
    const float *wpk;     // packed weights [nchunks][taps][CoutPad][KC]  (CoutPad counts N = 4*Cout for convT)
    const float *wpk4;    // optional second packing of the same conv3x3 for the F(4x4,3x3) kernel (see launch_conv3x3_wino4), or nullptr
    const float *bias;    // [Cout] folded BN shift (conv) or convT bias
    float *out;           // conv: [B][H][W][ldo]; convT: [B][2H][2W][ldo]
    int B, H, W;          // INPUT spatial size
    int Cin, ldc;
    int Cout;             // real output channels (convT: per-tap channels, N = 4*Cout)
    int CoutPad;          // padded N of wpk
    int ldo, co_off;
    int relu;
    // optional fused 2x2/stride-2 max pooling of the (post-ReLU) output: a second store of [B][H/2][W/2][pool_ld] at
    // channel 0.  Both conv kernels hold every 2x2 output block inside one lane, so this costs one max3 + one store.
    float *pool_out;
    int pool_ld;
    // optional split-K workspace (Winograd kernel only): when the (tile, channel) grid alone cannot fill the chip (single
    // images, deep levels) the launcher cuts K = Cin into up to 8 slices, each workgroup writes its partial 2x2 outputs into
    // slab [slice][B][H][W][Cout] and a second kernel sums the slabs in a fixed order (deterministic) and applies the
    // shift / ReLU / pooling.  nullptr or too small = never split.
    float *ksplit_ws;
    size_t ksplit_ws_bytes;
    int ksplit;           // set by the launcher
    // optional fused 1x1 head + argmax (F(4x4) one-block kernel only, Cout <= 64 so one workgroup holds every channel of
    // its pixels, no pooling): the post-ReLU tile goes through LDS instead of HBM and `out` is never written.
    //   head_w [classes][Cout], head_b [classes] (classes <= 4), planar logits [B][classes][H*W] (may be null), u8 labels
    // 16-bit kernels (conv_lp.hip): `in` always points at 16-bit activations (bf16 / fp16, NHWC, ldc in elements); out_lp
    // selects a 16-bit `out` / `pool_out` (everything but the layer in front of the fp32 head)
    int out_lp;
    Routing rt;           // resolved by the engine; unresolved = read the environment at the call
    const float *head_w;
    const float *head_b;
    int head_classes;
    float *head_logits;
    uint8_t *head_labels;
    // optional fused FIRST layer (conv_wino4s.hip, the fp32 plan's inc.c2): `in` is never read; the kernel builds each 16-channel
    // chunk of its 18x18 input patch from the u8 image instead -- /255 table, conv3x3 (first_cin = 1 input channel, 64 output
    // channels = this layer's Cin) + shift + ReLU, the arithmetic of conv3x3_first_kernel -- so the first layer's 1 GiB tensor is
    // neither written nor read back.  first_w is [9][1][Cin] (BN scale folded), first_shift [Cin].
    const uint8_t *first_img;
    int first_cin;                  // channels of first_img (conv_lpr.hip; conv_wino4s.hip takes one channel only)
    const float *first_lut, *first_w, *first_shift;
};

inline Routing routing_of(const ConvArgs &a) { return a.rt.resolved ? a.rt : Routing::from_env(); }

hipError_t launch_conv3x3_mfma(const ConvArgs &a, hipStream_t s);

// Winograd F(2x2,3x3) form of the same layer: a.wpk holds the TRANSFORMED weights U = G g G^T packed as
// [Cin/8][16 positions][CoutPad][8]  (WINO_KC = 8 input channels per K-chunk).  Same ConvArgs otherwise.
constexpr int WINO_KC = 8;
constexpr int WINO_SC = 32;       // input channels per raw-patch staging step (4 K-chunks: one whole 128-byte line per pixel)
hipError_t launch_conv3x3_wino(const ConvArgs &a, hipStream_t s);
// 8-wave / two-waves-per-SIMD re-tiling of the same algorithm on v_mfma_f32_16x16x4_f32; a.wpk is packed as
// [Cin/8][8 position pairs][CoutPad][16] with element 4*kq + 2*(pos & 1) + s = U_pos[k = 2*kq + s].
hipError_t launch_conv3x3_wino16(const ConvArgs &a, hipStream_t s);
// Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32: 16 tiles (16x16 output pixels) x 128 channels per workgroup.  a.wpk4 holds
// U = G g G^T (6x6) packed as [Cin/16][36 positions][CoutPad][16]; no split-K (small grids stay on the F(2x2) kernel).
constexpr int WINO4_KC = 16;
constexpr int WINO4_SC = 32;
hipError_t launch_conv3x3_wino4(const ConvArgs &a, hipStream_t s);
// The same algorithm for Cout <= 64 per workgroup with single-buffered 70 KB of LDS and <= 256 registers, so that two
// workgroups share a CU (conv_wino4s.hip); same packing (a.wpk4), bit-identical results, no split-K.  launch_conv3x3_wino4
// routes its one-block cases here unless MIUNET_WINO4S=0.
hipError_t launch_conv3x3_wino4s(const ConvArgs &a, hipStream_t s);
bool conv3x3_wino4_runs_staged(const ConvArgs &a);   // the routing decision of launch_conv3x3_wino4 (for the launch log)
bool conv3x3_wino4s_can_fuse_first(const ConvArgs &a, int first_cin);   // shape contract of the fused first layer (conv_wino4s.hip)
bool conv3x3_lpr_can_fuse_first(const ConvArgs &a, int first_cin);      // ... of the 16-bit resident-weight kernel (conv_lpr.hip)
// The two-block kernel hand-scheduled in gfx950 assembly and persistent (csrc/asm/gen_wino4_asm.py, csrc/wino4_asm.cpp): same
// packing (a.wpk4), same tensors.  shape_ok = the contract of the assembly (whole 16x16 blocks, Cin % 32 == 0 and >= 64, Cout % 128
// == 0, fp32, no fused head); runs_asm = what launch_conv3x3_wino4 decides (shape, MIUNET_WINO4_ASM, not a split-K grid).
bool conv3x3_wino4a_shape_ok(const ConvArgs &a);
bool conv3x3_wino4_runs_asm(const ConvArgs &a);
hipError_t launch_conv3x3_wino4a(const ConvArgs &a, hipStream_t s);
// ... and its sibling for the layers with 64 output channels per workgroup: blocks of 16 x 32 pixels (32 tiles) x 64 channels, a wave
// = 32 tiles x 16 channels, V single-buffered with a transform phase and an MFMA phase per chunk (csrc/asm/gen_wino4b_asm.py)
bool conv3x3_wino4b_shape_ok(const ConvArgs &a);
bool conv3x3_wino4_runs_asm_b(const ConvArgs &a);
hipError_t launch_conv3x3_wino4b(const ConvArgs &a, hipStream_t s);
hipError_t launch_wino_splitk_reduce(const ConvArgs &a, hipStream_t s);   // sums a.ksplit slabs of a.ksplit_ws into a.out
hipError_t launch_convT2x2_mfma(const ConvArgs &a, hipStream_t s);
// The transposed conv as four per-tap GEMMs sharing one A operand (convt_taps.hip): a.wpk4 holds the weights packed
// [ceil(Cin/32)*4 chunks of 8][4 taps (dy*2+dx)][convT_taps_cpad(Cout)][8], zero-padded; other fields as above.
inline int convT_taps_cpad(int cout) { return (cout + NPAD - 1) / NPAD * NPAD; }
// workgroups launch_convT2x2_taps would start with the tile shape it picks for this layer (the engine keeps the direct
// kernel below half a workgroup per CU)
long long convT_taps_grid(const ConvArgs &a);
hipError_t launch_convT2x2_taps(const ConvArgs &a, hipStream_t s);

// BASELINE config 3: bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16 (lpr_common.h: the shape the chip clocks highest).  Activations are bf16 in HBM too (rounded
// once, RNE, by the kernel that produces them); a.wpk holds bf16 weights packed [Cin/32][taps][CoutPad][32].
constexpr int KC_BF16 = 32;
hipError_t launch_conv3x3_bf16(const ConvArgs &a, hipStream_t s);
hipError_t launch_convT2x2_bf16(const ConvArgs &a, hipStream_t s);
// the same kernels on v_mfma_f32_16x16x32_f16 (BASELINE config 5's arithmetic); a.wpk holds IEEE half weights, same packing
hipError_t launch_conv3x3_fp16(const ConvArgs &a, hipStream_t s);
hipError_t launch_convT2x2_fp16(const ConvArgs &a, hipStream_t s);
// The wide layers (Cout % 128 == 0) of the same pipelines on a 4 x 4 register tile per wave (conv_lp2.hip): half the LDS bytes
// per MFMA.  Same packing (a.wpk), same arithmetic; conv3x3_lp2_takes says whether the layer's grid fills the chip.
bool conv3x3_lp2_takes(const ConvArgs &a);
hipError_t launch_conv3x3_lp2(const ConvArgs &a, bool fp16, hipStream_t s);
// The narrow layers (Cin, Cout in {32, 64}, 16-bit output, no fused head) with the weights resident in registers and a
// persistent workgroup per CU streaming input patches through an LDS ring (conv_lpr.hip).  Same packing, same arithmetic.
bool conv3x3_lpr_takes(const ConvArgs &a);
hipError_t launch_conv3x3_lpr(const ConvArgs &a, bool fp16, hipStream_t s);
// 128 -> 64 channels (the first convolution behind the top-level concat): the same scheme with the reduction split over a wave
// pair, partial sums through LDS (conv_lprk.hip).  Same packing; fp32 re-association differs from conv_mfma_bf16 by one add.
bool conv3x3_lprk_takes(const ConvArgs &a);
hipError_t launch_conv3x3_lprk(const ConvArgs &a, bool fp16, hipStream_t s);
// ... and the three largest transposed convolutions (Cin -> Cout = 64 -> 32, 128 -> 64, 256 -> 128; convt_lpr.hip)
bool convT2x2_lpr_takes(const ConvArgs &a);
hipError_t launch_convT2x2_lpr(const ConvArgs &a, bool fp16, hipStream_t s);

// First layer: u8 image -> (LUT /255) -> conv3x3 (Cin = 1..4) + shift + ReLU.  w is [9][Cin][Cout] (BN scale folded).
// out_kind: 0 = fp32 output, 1 = bf16, 2 = fp16 (the 16-bit pipelines keep every activation tensor 16-bit in HBM)
hipError_t launch_conv3x3_first(const uint8_t *img, const float *lut256, const float *w, const float *shift, float *out,
                                int B, int H, int W, int Cin, int Cout, int ldo, int out_kind, hipStream_t s,
                                const Routing *rt = nullptr);
// Same arithmetic on an fp32 NHWC input (layer_debug / small-Cin fallback is not needed elsewhere).

hipError_t launch_maxpool2x2(const float *in, int ldc, float *out, int B, int H, int W, int C, hipStream_t s);
// the same on 16-bit post-ReLU tensors (bf16 or fp16: non-negative values order like their bit patterns)
hipError_t launch_maxpool2x2_u16(const void *in, int ldc, void *out, int B, int H, int W, int C, hipStream_t s);

// 1x1 head + first-max-wins argmax: in [npix][Cin] -> planar logits [B][classes][H*W] (may be null) + u8 labels.
hipError_t launch_head_argmax(const float *in, int Cin, const float *w, const float *bias, int classes, float *logits,
                              uint8_t *labels, int B, int HW, hipStream_t s);

// Device form of the RAW16 preprocessing arithmetic (reference: src/preprocess.cpp:65-118), bit-exact:
//   minmax   : exact u16 min / max of n samples into mnmx[0], mnmx[1] (u32 words, pre-set to 65535 / 0 by the launcher)
//   resample : top-left aligned 4-tap bilinear in fp64 with the reference's operand order and NO fma contraction,
//              quantised with (uchar)(int)((v - mn) * (255.0 / (mx - mn)) + 0.5); mx = (u16)(mn + 1) when mn == mx.
hipError_t launch_minmax_u16(const uint16_t *raw, size_t n, unsigned *mnmx, hipStream_t s);
// dst_stride = bytes between consecutive output pixels (1 = planar tile; C = plane c of an interleaved HWC tile at dst + c)
hipError_t launch_resample_u8(const uint16_t *raw, int w, int h, const unsigned *mnmx, uint8_t *dst, int outW, int outH,
                              int dst_stride, hipStream_t s);

// Device form of postprocess_mask (reference: src/postprocess.cpp:13-79), integer-exact.  Workspace `ws` must hold
// postprocess_workspace_bytes(B, H, W) bytes; labels_in/out are u8 [B][H][W] (in-place allowed).
//   hole fill : 8-connected components of (label != 2) by lock-free union-find, per-root area + bbox by atomics; a
//               component is filled iff its bbox touches no image edge and area < min_area
//   open      : 3x3 erode then dilate, windows clipped to the image
//   filter    : 8-connected components of the opened mask, kept iff area >= min_area;  output in {0, 2}
size_t postprocess_workspace_bytes(int B, int H, int W);
hipError_t launch_postprocess_masks(const uint8_t *labels_in, uint8_t *labels_out, int B, int H, int W, int min_area,
                                    void *ws, hipStream_t s);

// Device form of Mask2Polygon::extract_contours (reference: src/mask2polygon.cpp:29-36 = threshold 127 +
// findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)), exact point sequences and contour order.
//   masks u8 [B][H][W] (any values; > 127 = foreground).  Per image: at most cap_contours contours and cap_points points.
//   out_xy     int32 [B][cap_points][2]   points of all contours of the image, contour after contour (newest first)
//   out_start  int32 [B][cap_contours+1]  first point of contour c; entry n_contours = total points
//   out_count  int32 [B]                  number of contours, or -1 when a capacity was too small
// Workspace: contour_workspace_bytes(B, H, W, cap_contours).
// mask_to_image (src/process.cpp:178-185): 0 -> 0, 1 -> 128, 2 -> 255, anything else -> 0
hipError_t launch_mask_to_image(const uint8_t *labels, uint8_t *vis, size_t n, hipStream_t s);
size_t contour_workspace_bytes(int B, int H, int W, int cap_contours);
hipError_t launch_extract_contours(const uint8_t *masks, int B, int H, int W, int *out_xy, int cap_points, int *out_start,
                                   int cap_contours, int *out_count, void *ws, hipStream_t s);

}  // namespace miunet
