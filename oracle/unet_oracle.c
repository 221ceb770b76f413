/*
 * oracle/unet_oracle.c -- CPU restatement of the inference half of the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libmiunet.so) never links, loads or calls it.
 *
 * PARITY UNPINNED for the network: the reference runs its UNet inside an unpublished TensorRT engine
 * (/root/reference/src/initialize.cpp:49-60, src/process.cpp:94-105,147) and holds no tests, weights,
 * golden vectors or sample I/O, and neither TensorRT nor OpenCV exists in this image, so nothing of the
 * reference's inference path can be compiled or run here.  What IS restated from reference code, line by line:
 *     orc_normalize_u8   <- preprocess_image            src/process.cpp:22-42   (u8 -> float / 255.0f, true division)
 *     orc_argmax_planar  <- the argmax loop             src/process.cpp:158-170 (strict '>' against -FLT_MAX, class 0 on ties/NaN)
 *     planar NCHW logits <- output binding layout       src/process.cpp:81-85, :163
 * The network itself follows BASELINE.json's topology with the free choices fixed in miunet/spec.py; it is
 * cross-checked in the build container against PyTorch-CPU (tests/golden/make_golden.py -> tests/golden/unet_*.npz).
 *
 * Arithmetic: fp32 throughout, one rounding per multiply and per add (built with -ffp-contract=off), accumulation
 * order per output element = tap-major (ky,kx), input-channel-minor, independent of the thread count.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_OK 0
#define ORC_EBADFILE 1
#define ORC_EALLOC 2
#define ORC_EARG 3

/* ---------------------------------------------------------------- elementwise pieces of the reference */

/* src/process.cpp:36-39: dst[i] = static_cast<float>(src[i]) / 255.0f */
void orc_normalize_u8(const uint8_t *src, size_t n, float *dst)
{
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i] / 255.0f;
}

/* src/process.cpp:158-170: max_prob starts at -FLT_MAX, class_idx at 0; per class c a pixel is updated only
 * where logit > max_prob (cv::compare CMP_GT: false for NaN).  The reference hard-codes 3 classes (:162). */
void orc_argmax_planar(const float *logits, int classes, size_t hw, uint8_t *labels)
{
    for (size_t i = 0; i < hw; ++i) {
        float best = -FLT_MAX;
        uint8_t idx = 0;
        for (int c = 0; c < classes; ++c) {
            float v = logits[(size_t)c * hw + i];
            if (v > best) { best = v; idx = (uint8_t)c; }
        }
        labels[i] = idx;
    }
}

/* ---------------------------------------------------------------- layers (NHWC fp32 activations) */

/* 3x3 conv, padding 1, stride 1, no bias.  w is PyTorch layout [Cout][Cin][3][3]. */
void orc_conv3x3(const float *in, int B, int H, int W, int Cin, const float *w, int Cout, float *out)
{
    /* repack to [tap][ci][co] so the innermost loop is contiguous in co */
    float *wt = (float *)malloc(sizeof(float) * 9 * (size_t)Cin * Cout);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t)
                wt[((size_t)t * Cin + ci) * Cout + co] = w[((size_t)co * Cin + ci) * 9 + t];
    enum { PX = 4, CB = 32 };
    const int nxb = (W + PX - 1) / PX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int by = 0; by < B * H; ++by) {
        for (int xb = 0; xb < nxb; ++xb) {
            const int b = by / H, y = by % H, x0 = xb * PX;
            const int npx = (W - x0) < PX ? (W - x0) : PX;
            const int interior = (x0 >= 1) && (x0 + PX + 1 <= W);
            for (int c0 = 0; c0 < Cout; c0 += CB) {
                const int ncb = (Cout - c0) < CB ? (Cout - c0) : CB;
                float acc[PX][CB];
                for (int p = 0; p < PX; ++p)
                    for (int c = 0; c < CB; ++c) acc[p][c] = 0.0f;
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1;
                    if (yy < 0 || yy >= H) continue;          /* zero padding contributes nothing */
                    const int dx = t % 3 - 1;
                    const float *irow = in + ((size_t)b * H + yy) * W * Cin;
                    if (interior && ncb == CB) {
                        /* same arithmetic as the generic branch, fixed trip counts so the compiler keeps acc in registers */
                        const float *ip = irow + (size_t)(x0 + dx) * Cin;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float *wr = wt + ((size_t)t * Cin + ci) * Cout + c0;
                            const float a0 = ip[ci], a1 = ip[Cin + ci], a2 = ip[2 * Cin + ci], a3 = ip[3 * Cin + ci];
                            for (int c = 0; c < CB; ++c) {
                                const float wv = wr[c];
                                acc[0][c] += a0 * wv;
                                acc[1][c] += a1 * wv;
                                acc[2][c] += a2 * wv;
                                acc[3][c] += a3 * wv;
                            }
                        }
                        continue;
                    }
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float *wr = wt + ((size_t)t * Cin + ci) * Cout + c0;
                        for (int p = 0; p < npx; ++p) {
                            const int xx = x0 + p + dx;
                            if (xx < 0 || xx >= W) continue;
                            const float a = irow[(size_t)xx * Cin + ci];
                            for (int c = 0; c < ncb; ++c) acc[p][c] += a * wr[c];
                        }
                    }
                }
                for (int p = 0; p < npx; ++p) {
                    float *o = out + (((size_t)b * H + y) * W + x0 + p) * Cout + c0;
                    for (int c = 0; c < ncb; ++c) o[c] = acc[p][c];
                }
            }
        }
    }
    free(wt);
}

/* eval-mode BatchNorm + ReLU in place: y = (x - mean) / sqrt(var + eps) * gamma + beta */
void orc_bn_relu(float *x, size_t npix, int C, const float *gamma, const float *beta, const float *mean,
                 const float *var, float eps, int relu)
{
    float *inv = (float *)malloc(sizeof(float) * C);
    for (int c = 0; c < C; ++c) inv[c] = 1.0f / sqrtf(var[c] + eps);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)npix; ++i) {
        float *p = x + (size_t)i * C;
        for (int c = 0; c < C; ++c) {
            float v = (p[c] - mean[c]) * inv[c] * gamma[c] + beta[c];
            p[c] = (relu && !(v > 0.0f)) ? 0.0f : v;
        }
    }
    free(inv);
}

void orc_maxpool2x2(const float *in, int B, int H, int W, int C, float *out)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int by = 0; by < B * Ho; ++by) {
        const int b = by / Ho, y = by % Ho;
        for (int x = 0; x < Wo; ++x) {
            const float *p00 = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C;
            const float *p01 = p00 + C, *p10 = p00 + (size_t)W * C, *p11 = p10 + C;
            float *o = out + (((size_t)b * Ho + y) * Wo + x) * C;
            for (int c = 0; c < C; ++c) {
                float m = p00[c];
                if (p01[c] > m) m = p01[c];
                if (p10[c] > m) m = p10[c];
                if (p11[c] > m) m = p11[c];
                o[c] = m;
            }
        }
    }
}

/* transposed conv 2x2 stride 2 with bias; w is PyTorch layout [Cin][Cout][2][2]; out is [B][2H][2W][Cout] written
 * with channel stride ldo at channel offset co_off (so it can land in the upper half of a concat buffer). */
void orc_convT2x2(const float *in, int B, int H, int W, int Cin, const float *w, const float *bias, int Cout,
                  float *out, int ldo, int co_off)
{
    float *wt = (float *)malloc(sizeof(float) * 4 * (size_t)Cin * Cout); /* [k][ci][co] */
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int k = 0; k < 4; ++k)
                wt[((size_t)k * Cin + ci) * Cout + co] = w[((size_t)ci * Cout + co) * 4 + k];
#pragma omp parallel for schedule(static)
    for (int by = 0; by < B * H; ++by) {
        const int b = by / H, y = by % H;
        float *acc = (float *)malloc(sizeof(float) * Cout);
        for (int x = 0; x < W; ++x) {
            const float *ip = in + (((size_t)b * H + y) * W + x) * Cin;
            for (int k = 0; k < 4; ++k) {
                for (int c = 0; c < Cout; ++c) acc[c] = 0.0f;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float a = ip[ci];
                    const float *wr = wt + ((size_t)k * Cin + ci) * Cout;
                    for (int c = 0; c < Cout; ++c) acc[c] += a * wr[c];
                }
                float *o = out + (((size_t)b * 2 * H + 2 * y + k / 2) * 2 * W + 2 * x + k % 2) * ldo + co_off;
                for (int c = 0; c < Cout; ++c) o[c] = acc[c] + bias[c];
            }
        }
        free(acc);
    }
    free(wt);
}

/* 1x1 head: NHWC in, PLANAR [B][classes][H][W] out (the reference's output binding layout, src/process.cpp:163) */
void orc_conv1x1_planar(const float *in, int B, int H, int W, int Cin, const float *w, const float *bias,
                        int classes, float *logits)
{
    const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)(B * hw); ++i) {
        const size_t b = (size_t)i / hw, p = (size_t)i % hw;
        const float *ip = in + (size_t)i * Cin;
        for (int k = 0; k < classes; ++k) {
            float acc = 0.0f;
            for (int ci = 0; ci < Cin; ++ci) acc += ip[ci] * w[(size_t)k * Cin + ci];
            logits[(b * classes + k) * hw + p] = acc + bias[k];
        }
    }
}

/* copy [npix][C] into a wider [npix][ldo] buffer at channel offset co_off */
static void copy_channels(const float *src, size_t npix, int C, float *dst, int ldo, int co_off)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)npix; ++i)
        memcpy(dst + (size_t)i * ldo + co_off, src + (size_t)i * C, sizeof(float) * C);
}

/* ---------------------------------------------------------------- weight file (miunet/spec.py) */

typedef struct {
    uint32_t in_ch, base, levels, classes;
    float eps;
    const float *payload;
    size_t n;
} orc_weights;

static int parse_weights(const void *blob, size_t len, orc_weights *w)
{
    const unsigned char *p = (const unsigned char *)blob;
    if (len < 36 || memcmp(p, "MIUNETW1", 8) != 0) return ORC_EBADFILE;
    uint32_t h[5];
    memcpy(h, p + 8, 20);
    if (h[0] != 1) return ORC_EBADFILE;
    w->in_ch = h[1]; w->base = h[2]; w->levels = h[3]; w->classes = h[4];
    memcpy(&w->eps, p + 28, 4);
    uint32_t n;
    memcpy(&n, p + 32, 4);
    if (len < 36 + (size_t)n * 4) return ORC_EBADFILE;
    w->payload = (const float *)(p + 36);
    w->n = n;
    return ORC_OK;
}

typedef struct { const float *p; size_t left; } cursor;
static const float *take(cursor *c, size_t n)
{
    if (c->left < n) return NULL;
    const float *r = c->p;
    c->p += n; c->left -= n;
    return r;
}

/* round-to-nearest-even to bfloat16, returned as float (what v_cvt_pk_bf16_f32 / a plain (__bf16) cast produce) */
static float bf16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return x;      /* NaN stays NaN */
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4);
    return x;
}

/* round-to-nearest-even to IEEE binary16 (with subnormals, overflow to infinity), returned as float */
static float fp16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    const int e = (int)((u >> 23) & 0xFF) - 127;
    if (e == 128) return x;                                       /* inf / NaN */
    if (e > 15) { u = sign | 0x7F800000u; memcpy(&x, &u, 4); return x; }
    if (e >= -14) {                                               /* keep 10 fraction bits */
        const uint32_t rem = u & 0x1FFFu;
        u &= ~0x1FFFu;
        if (rem > 0x1000u || (rem == 0x1000u && (u & 0x2000u))) u += 0x2000u;
        if (((u >> 23) & 0xFF) > 127 + 15) u = sign | 0x7F800000u;   /* rounded past 65504: infinity */
        memcpy(&x, &u, 4);
        return x;
    }
    if (e < -25) { u = sign; memcpy(&x, &u, 4); return x; }
    {                                                             /* subnormal half: multiples of 2^-24 */
        const float q = 5.9604644775390625e-08f;                  /* 2^-24 */
        float a = x < 0 ? -x : x;
        float n = a / q;                                          /* exact: a is a multiple of 2^-149 well below 2^-14 */
        float fl = (float)(long long)n;
        float r = n - fl;
        if (r > 0.5f || (r == 0.5f && ((long long)fl & 1))) fl += 1.0f;
        a = fl * q;
        return x < 0 ? -a : a;
    }
}

/* MODE 0 (default): conv -> BatchNorm -> ReLU exactly as written.
 * MODE 1 (BASELINE config 3 emulation): BatchNorm folded into the conv the way the engine folds it (scale in double ->
 *   float weights, shift added after the conv), and -- except for the first layer of the network, which the engine
 *   computes on the fp32 VALU -- BOTH conv operands rounded to bf16, products and sums in fp32.
 * MODE 2: the same with IEEE half operands (BASELINE config 5's arithmetic). */
static int g_mode = 0;
static float lp_round(float x) { return g_mode == 2 ? fp16_round(x) : bf16_round(x); }
static int g_first_conv = 0;

/* one [conv3x3 -> BN -> ReLU] x2 block; returns 0 or error; out must hold B*H*W*Cout floats */
static int double_conv(cursor *cur, const float *in, int B, int H, int W, int Cin, int Cout, float eps, float *tmp,
                       float *out)
{
    const float *src = in;
    int ci = Cin;
    float *dsts[2] = { tmp, out };
    for (int k = 0; k < 2; ++k) {
        const float *w = take(cur, (size_t)Cout * ci * 9);
        const float *g = take(cur, Cout), *be = take(cur, Cout), *mu = take(cur, Cout), *va = take(cur, Cout);
        if (!w || !g || !be || !mu || !va) return ORC_EBADFILE;
        if (g_mode == 0) {
            orc_conv3x3(src, B, H, W, ci, w, Cout, dsts[k]);
            orc_bn_relu(dsts[k], (size_t)B * H * W, Cout, g, be, mu, va, eps, 1);
        } else {
            const int low = !g_first_conv;                 /* first conv of the network stays fp32 */
            g_first_conv = 0;
            const size_t nw = (size_t)Cout * ci * 9, nx = (size_t)B * H * W * ci, npix = (size_t)B * H * W;
            float *wf = (float *)malloc(sizeof(float) * nw), *shift = (float *)malloc(sizeof(float) * Cout);
            float *xr = low ? (float *)malloc(sizeof(float) * nx) : NULL;
            for (int co = 0; co < Cout; ++co) {
                const double sc = (double)g[co] / sqrt((double)va[co] + (double)eps);
                shift[co] = (float)((double)be[co] - (double)mu[co] * sc);
                for (size_t q = 0; q < (size_t)ci * 9; ++q) {
                    const float f = (float)((double)w[(size_t)co * ci * 9 + q] * sc);
                    wf[(size_t)co * ci * 9 + q] = low ? lp_round(f) : f;
                }
            }
            if (low)
                for (size_t q = 0; q < nx; ++q) xr[q] = lp_round(src[q]);
            orc_conv3x3(low ? xr : src, B, H, W, ci, wf, Cout, dsts[k]);
            for (size_t p = 0; p < npix; ++p)
                for (int co = 0; co < Cout; ++co) {
                    float v = dsts[k][p * Cout + co] + shift[co];
                    dsts[k][p * Cout + co] = v > 0.0f ? v : 0.0f;
                }
            free(wf); free(shift); free(xr);
        }
        src = dsts[k];
        ci = Cout;
    }
    return ORC_OK;
}

/*
 * Whole inference step = preprocess_image + engine + argmax of execute_inference (src/process.cpp:123-175),
 * batched: imgs u8 [B][H][W][in_ch] -> logits f32 [B][classes][H][W] (may be NULL) and labels u8 [B][H][W] (may be NULL).
 */
int orc_unet_forward(const void *blob, size_t blob_len, const uint8_t *imgs, int B, int H, int W, float *logits,
                     uint8_t *labels, int nthreads)
{
    orc_weights wf;
    int rc = parse_weights(blob, blob_len, &wf);
    if (rc) return rc;
    const int L = (int)wf.levels;
    if (B <= 0 || L < 1 || L > 6 || (H % (1 << L)) || (W % (1 << L))) return ORC_EARG;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    cursor cur = { wf.payload, wf.n };
    int ch[8];
    for (int i = 0; i <= L; ++i) ch[i] = (int)wf.base << i;
    const size_t npix0 = (size_t)B * H * W;
    /* skip tensors live inside concat buffers: cat[i] = [B][H>>i][W>>i][2*ch[i]], skip in the lower half */
    float *cat[8] = { 0 };
    float *x = (float *)malloc(sizeof(float) * npix0 * ((int)wf.in_ch > ch[0] ? (int)wf.in_ch : ch[0]));
    float *t1 = (float *)malloc(sizeof(float) * npix0 * ch[0] * 2);
    float *t2 = (float *)malloc(sizeof(float) * npix0 * ch[0] * 2);
    if (!x || !t1 || !t2) return ORC_EALLOC;
    for (int i = 0; i < L; ++i) {
        cat[i] = (float *)malloc(sizeof(float) * (npix0 >> (2 * i)) * 2 * ch[i]);
        if (!cat[i]) return ORC_EALLOC;
    }
    orc_normalize_u8(imgs, npix0 * wf.in_ch, x);

    /* encoder */
    int h = H, w = W;
    rc = double_conv(&cur, x, B, h, w, (int)wf.in_ch, ch[0], wf.eps, t1, t2);
    if (rc) return rc;
    float *feat = t2; /* current feature map [B][h][w][ch[i]] */
    for (int i = 1; i <= L; ++i) {
        copy_channels(feat, (size_t)B * h * w, ch[i - 1], cat[i - 1], 2 * ch[i - 1], 0);
        orc_maxpool2x2(feat, B, h, w, ch[i - 1], x);
        h /= 2; w /= 2;
        rc = double_conv(&cur, x, B, h, w, ch[i - 1], ch[i], wf.eps, t1, t2);
        if (rc) return rc;
        feat = t2;
    }
    /* decoder: concat order = [skip, upsampled] */
    for (int i = 1; i <= L; ++i) {
        const int cin = ch[L - i + 1], cout = cin / 2, lvl = L - i;
        const float *tw = take(&cur, (size_t)cin * cout * 4), *tb = take(&cur, cout);
        if (!tw || !tb) return ORC_EBADFILE;
        if (g_mode == 0) {
            orc_convT2x2(feat, B, h, w, cin, tw, tb, cout, cat[lvl], 2 * cout, cout);
        } else {                                            /* bf16 operands, fp32 accumulate, fp32 bias */
            const size_t nw = (size_t)cin * cout * 4, nx = (size_t)B * h * w * cin;
            float *wr = (float *)malloc(sizeof(float) * nw), *xr = (float *)malloc(sizeof(float) * nx);
            for (size_t q = 0; q < nw; ++q) wr[q] = lp_round(tw[q]);
            for (size_t q = 0; q < nx; ++q) xr[q] = lp_round(feat[q]);
            orc_convT2x2(xr, B, h, w, cin, wr, tb, cout, cat[lvl], 2 * cout, cout);
            free(wr); free(xr);
        }
        h *= 2; w *= 2;
        rc = double_conv(&cur, cat[lvl], B, h, w, cin, cout, wf.eps, t1, t2);
        if (rc) return rc;
        feat = t2;
    }
    const float *ow = take(&cur, (size_t)wf.classes * ch[0]), *ob = take(&cur, wf.classes);
    if (!ow || !ob || cur.left != 0) return ORC_EBADFILE;
    float *lg = logits ? logits : (float *)malloc(sizeof(float) * npix0 * wf.classes);
    if (!lg) return ORC_EALLOC;
    orc_conv1x1_planar(feat, B, H, W, ch[0], ow, ob, (int)wf.classes, lg);
    if (labels)
        for (int b = 0; b < B; ++b)
            orc_argmax_planar(lg + (size_t)b * wf.classes * H * W, (int)wf.classes, (size_t)H * W,
                              labels + (size_t)b * H * W);
    if (!logits) free(lg);
    for (int i = 0; i < L; ++i) free(cat[i]);
    free(x); free(t1); free(t2);
    return ORC_OK;
}

/* Same network with bf16 conv operands (see MODE 1 above): the checker for BASELINE config 3. */
int orc_unet_forward_bf16(const void *blob, size_t blob_len, const uint8_t *imgs, int B, int H, int W, float *logits,
                          uint8_t *labels, int nthreads)
{
    g_mode = 1;
    g_first_conv = 1;
    const int rc = orc_unet_forward(blob, blob_len, imgs, B, H, W, logits, labels, nthreads);
    g_mode = 0;
    return rc;
}

int orc_unet_forward_fp16(const void *blob, size_t blob_len, const uint8_t *imgs, int B, int H, int W, float *logits,
                          uint8_t *labels, int nthreads)
{
    g_mode = 2;
    g_first_conv = 1;
    const int rc = orc_unet_forward(blob, blob_len, imgs, B, H, W, logits, labels, nthreads);
    g_mode = 0;
    return rc;
}

float orc_bf16_round(float x) { return bf16_round(x); }
float orc_fp16_round(float x) { return fp16_round(x); }

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
