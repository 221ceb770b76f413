/*
 * mi_unet.h -- C-ABI of the MI355X-native UNet segmentation engine (libmiunet.so).
 *
 * This is the drop-in seam for the reference's TensorRT call: everything between the normalised 8-bit tile and the
 * u8 label map in MedicalSeg::execute_inference (/root/reference/src/process.cpp:123-175), i.e.
 *     preprocess_image  u8 -> f32 /255.0f            src/process.cpp:22-42
 *     H2D + cudaGraphLaunch(engine) + D2H            src/process.cpp:143-155
 *     3-class first-max-wins argmax                  src/process.cpp:158-170
 * plus the lifecycle around it (engine load: src/initialize.cpp:26-77; per-thread context with device buffers, stream
 * and captured graph: src/process.cpp:45-120; teardown: src/cleanup.cpp:10-64).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns 0 on success or an MI_UNET_E* code and
 * leaves a human-readable message retrievable through mi_unet_last_error() (thread local).  There is no CPU fallback:
 * without a HIP device every entry point that needs one fails with MI_UNET_ENODEVICE.
 */
#ifndef MI_UNET_H
#define MI_UNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_UNET_OK 0
#define MI_UNET_EARG 1        /* bad argument / unsupported configuration */
#define MI_UNET_ENODEVICE 2   /* no usable HIP device */
#define MI_UNET_EHIP 3        /* a HIP runtime call failed (message carries hipGetErrorString) */
#define MI_UNET_EFILE 4       /* weight file missing / malformed / topology mismatch */
#define MI_UNET_ESTATE 5      /* call order violated (e.g. infer before load_weights) */

typedef struct mi_unet mi_unet_t;

typedef struct mi_unet_config {
    int height;      /* input tile height; the reference fixes 512 (src/process.cpp:70, :126) */
    int width;       /* input tile width;  512 */
    int in_ch;       /* 1 (grayscale, src/process.cpp:70) */
    int base;        /* channels of the first level, 64 */
    int levels;      /* number of 2x down/up steps, 4 */
    int classes;     /* 3 (src/process.cpp:162) */
    int max_batch;   /* images processed per micro-batch; device buffers are sized for this */
    int device;      /* HIP device ordinal (the reference uses implicit device 0) */
    int conv_algo;   /* 3x3 convolution algorithm, fp32 arithmetic on the fp32 MFMA unless stated:
                        MI_UNET_CONV_AUTO (environment MIUNET_CONV_ALGO=direct|winograd|winograd16|bf16|fp16, else the default),
                        MI_UNET_CONV_DIRECT (implicit GEMM, 9 taps), MI_UNET_CONV_WINOGRAD (the default: Winograd F(4x4,3x3), 4x
                        fewer MACs, on every layer whose grid fills the chip or can split K, F(2x2,3x3) on the rest; the
                        transposed convs as four per-tap GEMMs; head fused into the last conv.  DESIGN.md 4.2-4.4) */
} mi_unet_config;

#define MI_UNET_CONV_AUTO 0
#define MI_UNET_CONV_DIRECT 1
#define MI_UNET_CONV_WINOGRAD 2
#define MI_UNET_CONV_WINOGRAD16 3   /* same algorithm, 8-wave tiling on v_mfma_f32_16x16x4_f32 (two waves per SIMD) */
#define MI_UNET_CONV_BF16 4         /* BASELINE config 3: bf16 conv operands (weights packed bf16, activations rounded to
                                       bf16 once by the kernel that produces them and kept bf16 in HBM), fp32 accumulate on
                                       v_mfma_f32_16x16x32_bf16.  NOT the fp32 metric: logits follow the bf16-operand oracle. */
#define MI_UNET_CONV_FP16 5         /* BASELINE config 5's arithmetic: the same kernels with IEEE half operands
                                       (v_mfma_f32_16x16x32_f16), fp32 accumulate */
#define MI_UNET_CONV_DEFAULT MI_UNET_CONV_WINOGRAD

/* Fills *cfg with the reference's constants: 512x512x1, base 64, 4 levels, 3 classes, max_batch 16, device 0. */
void mi_unet_default_config(mi_unet_config *cfg);

/* Replaces createInferRuntime + per-thread context creation (src/initialize.cpp:48, src/process.cpp:45-120):
 * allocates all device buffers and the stream.  No weights yet. */
int mi_unet_create(const mi_unet_config *cfg, mi_unet_t **out);

/* Replaces reading + deserialising the .trt engine (src/initialize.cpp:49-60).  File format: miunet/spec.py
 * ("MIUNETW1").  Folds eval-mode BatchNorm into the conv weights, repacks for the MFMA kernels, uploads. */
int mi_unet_load_weights(mi_unet_t *h, const char *path);
int mi_unet_load_weights_from_memory(mi_unet_t *h, const void *blob, size_t len);

/* The hot path = execute_inference (src/process.cpp:123-175) for B images at once, host buffers:
 *   imgs   u8  [B][H][W][in_ch]              (the 8-bit normalised tile the reference reads back at :217)
 *   labels u8  [B][H][W]          out        (class index per pixel, as pred_mask at :170)
 *   logits f32 [B][classes][H][W] out/NULL   (planar, the reference's output binding layout :81-85, :163)
 * B may exceed max_batch (processed in micro-batches). */
int mi_unet_infer_u8(mi_unet_t *h, const uint8_t *imgs, int B, uint8_t *labels, float *logits);

/* Same, with all three buffers already resident in device memory (HBM); asynchronous on the engine's stream.
 * Call mi_unet_sync() before reading results from another stream. */
int mi_unet_infer_u8_device(mi_unet_t *h, const uint8_t *d_imgs, int B, uint8_t *d_labels, float *d_logits);

/* SURVEY §8f row f1 -- the arithmetic of Preprocess::preprocess_raw (src/preprocess.cpp:65-118) on the device, fused in
 * front of the network: B headerless little-endian u16 RAW images (host pointers; image i is heights[i] x widths[i]) ->
 * exact min/max -> top-left-aligned bilinear resample to the engine's H x W in fp64 -> u8 tiles (bit-exact with the
 * reference's CPU loop) -> UNet -> labels.  tiles (u8 [B][H][W][in_ch], host) and logits may be NULL.
 * Engines with in_ch = C > 1 (BASELINE config 5: C = 3) take C planes per image: raws / widths / heights then hold B*C
 * entries, plane c of image i at index i*C + c.  Every plane is preprocessed on its own (own size, own min/max -- exactly
 * what preprocess_raw would do to it as a file) and becomes channel c of the interleaved tile.  The reference defines only
 * single-plane RAW (src/preprocess.cpp:86); a caller holding one plane passes its pointer C times, which is the grey ->
 * B,G,R replication of cv::imread(IMREAD_COLOR) (src/mask2polygon.cpp:117). */
int mi_unet_infer_raw16(mi_unet_t *h, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                        uint8_t *tiles, uint8_t *labels, float *logits);

/* SURVEY §8f row f2 -- postprocess_mask (src/postprocess.cpp:47-79) on the device, integer-exact: hole fill (8-connected
 * components of label != 2 that touch no image edge and are smaller than 6 % of the image), 3x3 open, keep components of
 * at least 6 % of the image; output in {0, 2}.
 *   mi_unet_set_postprocess(h, 1): every mi_unet_infer_* call returns the POSTPROCESSED masks instead of the raw label
 *                                  maps (the label maps never leave the device in between).
 *   mi_unet_postprocess_masks    : the stage alone on host buffers, u8 [B][H][W] in -> out (may alias). */
int mi_unet_set_postprocess(mi_unet_t *h, int on);
int mi_unet_postprocess_masks(mi_unet_t *h, const uint8_t *labels, int B, uint8_t *out);

/* SURVEY §8f row f3 -- Mask2Polygon::extract_contours (src/mask2polygon.cpp:29-36: threshold 127 +
 * findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)) on the device, same point sequences and contour order.
 *   masks  u8 [B][H][W] host (e.g. the 0/128/255 visualisation of mask_to_image; > 127 = foreground)
 *   xy     int32 [B][cap_points][2] out: x,y pairs of image b's contours, contour after contour (newest first)
 *   start  int32 [B][cap_contours + 1] out: first point of contour c; entry [counts[b]] = total points of image b
 *   counts int32 [B] out: number of contours of image b, or -1 if one of the two capacities was too small for it */
int mi_unet_extract_contours(mi_unet_t *h, const uint8_t *masks, int B, int32_t *xy, int cap_points, int32_t *start,
                             int cap_contours, int32_t *counts);

/* The whole device half of process_single_image (src/process.cpp:188-242) for B images in one call, nothing leaving the
 * device in between: RAW16 -> min/max + bilinear + quantise (f1) -> UNet + argmax -> postprocess_mask (f2) ->
 * mask_to_image -> extract_contours (f3).  Outputs (host): tiles u8 [B][H][W] (the _normalized.png pixels, may be NULL),
 * masks u8 [B][H][W] (the _mask.png pixels: 0 / 255), and the contours in the layout of mi_unet_extract_contours. */
int mi_unet_segment_raw16(mi_unet_t *h, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                          uint8_t *tiles, uint8_t *masks, int32_t *xy, int cap_points, int32_t *start, int cap_contours,
                          int32_t *counts);

/* Page-locked host memory.  RAW images handed to mi_unet_infer_raw16 / mi_unet_segment_raw16 (and their group forms) from such
 * a buffer are read by the DMA engine directly -- no staging copy on the calling thread (100 MB for sixteen 2048 x 1536 images:
 * 3 - 5 ms of memcpy that the pageable route pays).  Any hipHostMalloc'd / hipHostRegister'ed pointer is recognised, not only
 * these; ordinary pointers keep working through the engine's own pinned ring. */
int mi_unet_host_alloc(size_t bytes, void **p);
void mi_unet_host_free(void *p);

/* Device time of the stages of the LAST mi_unet_infer_raw16 / mi_unet_segment_raw16 call on this handle, in milliseconds, summed
 * over its micro-batches (hipEvent pairs on the streams the stages run on; the upload / preprocess stage of micro-batch k + 1
 * runs on a second stream under the network of micro-batch k, so the stages may add up to more than the call took).  The
 * reference logs two durations per image (src/process.cpp:223-228, :245-253); these are the terms of its "Inference time". */
#define MI_UNET_STAGE_UPLOAD_PRE 0   /* host staging copy + H2D + min/max + bilinear resample (f1) */
#define MI_UNET_STAGE_NETWORK 1      /* UNet + argmax (graph replay) */
#define MI_UNET_STAGE_POSTPROCESS 2  /* postprocess_mask (f2) */
#define MI_UNET_STAGE_CONTOURS 3     /* mask_to_image + extract_contours (f3) */
#define MI_UNET_STAGE_DOWNLOAD 4     /* D2H of tiles, masks / label maps, contours, logits */
#define MI_UNET_N_STAGES 5
int mi_unet_last_stage_ms(const mi_unet_t *h, float *ms /* [MI_UNET_N_STAGES] */);

/* Use an external hipStream_t (e.g. the caller framework's current stream) instead of the engine's own. NULL restores it. */
int mi_unet_set_stream(mi_unet_t *h, void *hip_stream);
int mi_unet_sync(mi_unet_t *h);

/* Time the last `mi_unet_infer_u8_device` calls: brackets with hipEvents on the engine's stream.
 * mi_unet_timer_begin/end return elapsed milliseconds through *ms at end (end synchronises the stop event). */
int mi_unet_timer_begin(mi_unet_t *h);
int mi_unet_timer_end(mi_unet_t *h, float *ms);

/* Per-launch accounting since profiling was last switched on: mi_unet_set_profiling(h,1) clears the log and from then on
 * brackets every kernel launch with a hipEvent pair recorded on the launch stream (no host wait at launch time).
 * mi_unet_get_kernel_stats synchronises the last event, fills up to `cap` entries (launch order) and returns the number
 * of launches logged in *n. */
typedef struct mi_unet_kernel_stat {
    char name[48];        /* layer name, e.g. "up4.c1" */
    char kernel[32];      /* kernel family: conv3x3_mfma, convT2x2_mfma, conv3x3_c1, maxpool2x2, head_argmax */
    double flops;         /* algorithmic FLOPs of this launch (2*MAC) */
    double bytes;         /* algorithmic HBM bytes of this launch (inputs + weights + outputs, each once) */
    float ms;             /* measured duration */
} mi_unet_kernel_stat;
int mi_unet_set_profiling(mi_unet_t *h, int on);
int mi_unet_get_kernel_stats(mi_unet_t *h, mi_unet_kernel_stat *stats, int cap, int *n);

/* Parity hook: run ONE layer kernel on host NHWC fp32 buffers (uploaded, run, downloaded).
 *   op = "conv3x3"  : in [B][H][W][Cin], w [Cout][Cin][3][3], scale/shift [Cout] (folded BN; NULL = 1/0), relu flag
 *   op = "conv3x3_wino" / "conv3x3_wino16" : the same layer through the Winograd F(2x2,3x3) kernels
 *   op = "conv3x3_bf16" / "convT2x2_bf16" / "conv3x3_fp16" / "convT2x2_fp16" : the 16-bit-operand kernels
 *   op = "convT2x2" : in [B][H][W][Cin], w [Cin][Cout][2][2], shift = bias [Cout]  -> out [B][2H][2W][Cout]
 *   op = "maxpool"  : in [B][H][W][Cin]                                          -> out [B][H/2][W/2][Cin]
 * Weights are given in PyTorch layout exactly as in the weight file. */
int mi_unet_layer_debug(int device, const char *op, const float *in, int B, int H, int W, int Cin, const float *w,
                        const float *scale, const float *shift, int Cout, int relu, float *out);

/* Numeric guard of the default fp32 plan (conv_algo auto / winograd).  Winograd F(4x4,3x3) is exact arithmetic re-associated:
 * its rounding error relative to a layer's operand range is about five times that of F(2x2,3x3) (2e-5 against 4e-6 on logits
 * of magnitude 4), and the path's bar is absolute (logits within 1e-3 of the fp32 reference, BASELINE north_star).  Whether
 * F(4x4) holds that bar therefore depends on the dynamic range of the loaded weights.  Guarantee: when weights are loaded, one
 * probe tile runs through the plan with every 3x3 layer on F(4x4) and again with every 3x3 layer on F(2x2); if the logits differ
 * by more than 5e-4 (half the bar; the difference of the two plans overstates either one's own error), this handle -- and its clones -- runs F(2x2,3x3) on every layer.  The returned
 * text says which (the facade logs it); *tripped / *diff (may be NULL) receive the decision and the measured difference.
 * Beyond logits of magnitude ~1e2 no fp32 algorithm holds an ABSOLUTE 1e-3 (fp32 itself resolves 6e-8 of the range per
 * operation); there the guard still picks the tighter algorithm and the meaningful bound is relative (~1e-6 of the range). */
const char *mi_unet_numeric_guard(const mi_unet_t *h, int *tripped, float *diff);

/* In-situ parity hook (tests): what the engine's OWN launch plan does to its OWN activations, layer by layer, at any size.
 * The opaque seam this opens is the reference's graph replay (src/process.cpp:143-155), whose intermediate tensors nobody
 * can see.  A "layer" is one step of the plan in launch order (inc.c1, inc.c2, down1.pool, down1.c1, ... outc+argmax);
 * steps whose work is fused into their producer (pooling, the head) are listed and flagged `skipped`.
 *   mi_unet_debug_layer_count : number of steps
 *   mi_unet_debug_layer_info  : static description of step `layer` (shapes are per image)
 *   mi_unet_debug_capture     : uploads B <= max_batch images, runs the plan EAGERLY with exactly the kernels a batch of B
 *       takes, stops after step `layer`, and returns for image `img` of the batch, converted to float, dense NHWC:
 *         in     [in_h][in_w][in_c]      the tensor the step's kernel read (u8 image values 0..255 for the first layer)
 *         out    [out_h][out_w][out_c]   what it stored; when the step ran the fused 1x1 head instead (info->fused_head),
 *                                        the planar logits [classes][out_h][out_w]
 *         pooled [out_h/2][out_w/2][out_c] the fused 2x2 max-pooled tensor (only when info->pooled; may be NULL)
 *         labels [out_h][out_w]          argmax labels (only for head / fused-head steps; may be NULL)
 *       and in *info the dynamic facts: kernel family launched, storage width of the tensors in HBM, flags. */
typedef struct mi_unet_layer_info {
    char name[48];
    char kernel[32];       /* kernel family launched (capture only) */
    int kind;              /* 0 first conv, 1 conv3x3, 2 convT2x2, 3 maxpool2x2, 4 head+argmax */
    int in_h, in_w, in_c;
    int out_h, out_w, out_c;
    int in_bits, out_bits; /* 8 = u8 image, 16 = bf16 / fp16 (per conv_algo), 32 = fp32: storage type in HBM (capture only) */
    int pooled;            /* the step also stored the 2x2 max-pooled tensor */
    int fused_head;        /* the step ran the 1x1 head + argmax in its epilogue; its own activations never reached HBM */
    int skipped;           /* not launched at this batch size: fused into its producer (pooling, head) or its consumer (first layer) */
    int fused_first;       /* the step computed the network's first layer in its loader: `in` of the capture is the u8 image */
} mi_unet_layer_info;
int mi_unet_debug_layer_count(const mi_unet_t *h);
int mi_unet_debug_layer_info(const mi_unet_t *h, int layer, mi_unet_layer_info *info);
int mi_unet_debug_capture(mi_unet_t *h, const uint8_t *imgs, int B, int layer, int img, float *in, float *out, float *pooled,
                          uint8_t *labels, mi_unet_layer_info *info);

/* A second context on the SAME device that shares the source engine's weight blob (no second copy, no re-packing) but
 * owns its activation buffers, stream and graphs -- the counterpart of the reference's per-thread TensorRTContext over one
 * shared ICudaEngine (include/process.h:13-26, src/process.cpp:15, :69).  max_batch <= 0 keeps the source's.  The weights
 * stay alive until the last handle that shares them is destroyed, in any order. */
int mi_unet_clone(const mi_unet_t *src, int max_batch, mi_unet_t **out);

void mi_unet_destroy(mi_unet_t *h);

/* ---- Multi-device group (the slot is the reference's sequential file loop, src/main.cpp:148-164) -------------
 * One engine handle + one host worker thread per device inside ONE process.  The path shards by image: a batch of B
 * images is cut into contiguous ranges (rank r of R owns [r*q + min(r, B%R), ...), the first B%R ranks one image more) and
 * every rank runs its range independently -- no collective inside the forward pass.  Two exchange steps exist:
 *   weights : parsed, BN-folded and packed ONCE on the host, uploaded to the first device, then sent to the other devices
 *             device-to-device: ncclBroadcast over xGMI (RCCL, loaded with dlopen when the group spans > 1 distinct device)
 *             or a peer copy when RCCL is unavailable / two ranks share a device;
 *   labels  : MI_UNET_GATHER_HOST (default): every rank copies its own range straight into the caller's host buffer (its own
 *             PCIe link, no collective);  MI_UNET_GATHER_XGMI: grouped ncclSend / ncclRecv of the u8 label maps into the
 *             first device (7 concurrent point-to-point transfers on an 8-GPU node), then one D2H.
 * `devices` lists HIP ordinals, one rank each (a repeated ordinal puts two ranks on one GPU: a test configuration, peer-copy
 * weights, HOST gather only); devices == NULL means ordinals cfg->device .. cfg->device + n_devices - 1, and n_devices <= 0
 * means every visible device.  cfg->max_batch is per rank.  N > 1 distinct devices has never run on hardware here; the RCCL code path
 * itself (communicators, broadcast, send / recv gather) runs in the tests on one card against a stand-in library (MIUNET_RCCL_LIB,
 * MIUNET_GROUP_RCCL=2: tests/cpu/fake_rccl.cpp). */
typedef struct mi_unet_group mi_unet_group_t;
#define MI_UNET_GATHER_HOST 0
#define MI_UNET_GATHER_XGMI 1
int mi_unet_group_create(const mi_unet_config *cfg, const int *devices, int n_devices, mi_unet_group_t **out);
/* A second set of contexts over the same devices that shares every rank's weight blob (mi_unet_clone per rank) and owns its
 * buffers, streams and worker threads: two groups can have two batches in flight at once -- the small serial stages of one
 * (RAW upload, preprocessing, labelling, the contour walk) overlap the network of the other.  The clone has no RCCL
 * communicator (host gather only). */
int mi_unet_group_clone(mi_unet_group_t *src, mi_unet_group_t **out);
int mi_unet_group_size(const mi_unet_group_t *g);
mi_unet_t *mi_unet_group_handle(mi_unet_group_t *g, int rank);            /* rank's engine (owned by the group) */
int mi_unet_group_load_weights(mi_unet_group_t *g, const char *path);
int mi_unet_group_load_weights_from_memory(mi_unet_group_t *g, const void *blob, size_t len);
int mi_unet_group_set_gather(mi_unet_group_t *g, int mode);               /* EARG when XGMI is asked for without RCCL */
int mi_unet_group_set_postprocess(mi_unet_group_t *g, int on);
/* how the weights reached ranks > 0 -- "rccl", "peer-copy" or "host-upload", with the reason when a faster transport failed
 * and the next one was taken -- and the gather mode in force */
const char *mi_unet_group_weight_transport(const mi_unet_group_t *g);
int mi_unet_group_gather(const mi_unet_group_t *g);
/* Sharded forms of mi_unet_infer_u8 / mi_unet_infer_raw16 / mi_unet_segment_raw16: same arguments, same results, the batch
 * split across the group's ranks. */
int mi_unet_group_infer_u8(mi_unet_group_t *g, const uint8_t *imgs, int B, uint8_t *labels, float *logits);
int mi_unet_group_infer_raw16(mi_unet_group_t *g, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                              uint8_t *tiles, uint8_t *labels, float *logits);
int mi_unet_group_segment_raw16(mi_unet_group_t *g, const uint16_t *const *raws, const int *widths, const int *heights, int B,
                                uint8_t *tiles, uint8_t *masks, int32_t *xy, int cap_points, int32_t *start, int cap_contours,
                                int32_t *counts);
void mi_unet_group_destroy(mi_unet_group_t *g);
/* The split itself (pure host arithmetic, needs no device): rank's range [*lo, *hi) of n_items over `world` ranks. */
int mi_unet_shard_range(int n_items, int rank, int world, int *lo, int *hi);

/* Message of the last failing call on this thread ("" if none). */
const char *mi_unet_last_error(void);

/* Number of visible HIP devices (0 when there is no driver); never fails. */
int mi_unet_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_UNET_H */
