/*
 * oracle/imgproc_oracle.c -- CPU restatement of the reference's CPU-side image stages.  TEST INFRASTRUCTURE ONLY
 * (see unet_oracle.c for who may load this library).
 *
 * Each function follows the cited lines of /root/reference.  Where the reference calls OpenCV (not installed here,
 * version unpinned: SURVEY.md §2.3) the documented OpenCV semantics are restated with the simplest possible
 * algorithm (flood fill instead of OpenCV's two-pass labeller, plain min/max windows for morphology); these parts are
 * PARITY UNPINNED against OpenCV itself and are cross-checked in the build container against scipy.ndimage
 * (tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- A2: compute_minmax, src/preprocess.cpp:65-74 ---- */
void orc_minmax_u16(const uint16_t *src, size_t n, uint16_t *mn, uint16_t *mx)
{
    uint16_t lo = 65535, hi = 0;
    for (size_t i = 0; i < n; ++i) {
        if (src[i] < lo) lo = src[i];
        if (src[i] > hi) hi = src[i];
    }
    *mn = lo; *mx = hi;
}

/* ---- A3: the resampling loop of Preprocess::preprocess_raw, src/preprocess.cpp:81-118 ----
 * stepX = w/outW (double), top-left aligned taps, ix1/iy1 clamped, 4-tap bilinear in double in the reference's
 * operand order, quantise with (uchar)((v - mn) * scale8 + 0.5).  `if (mn == mx) mx = mn + 1` is evaluated in
 * uint16_t as in the reference (:92), so mn == 65535 wraps mx to 0 and (mx - mn) is the int -65535. */
void orc_preprocess_raw(const uint16_t *src, int w, int h, uint8_t *dst, int outW, int outH)
{
    const double stepX = (double)w / outW, stepY = (double)h / outH;
    uint16_t mn, mx;
    orc_minmax_u16(src, (size_t)w * h, &mn, &mx);
    if (mn == mx) mx = (uint16_t)(mn + 1);
    const double scale8 = 255.0 / ((int)mx - (int)mn);
    for (int y = 0; y < outH; ++y) {
        for (int x = 0; x < outW; ++x) {
            double fx = x * stepX, fy = y * stepY;
            int ix = (int)fx, iy = (int)fy;
            int ix1 = ix + 1 < w - 1 ? ix + 1 : w - 1;
            int iy1 = iy + 1 < h - 1 ? iy + 1 : h - 1;
            double dx = fx - ix, dy = fy - iy;
            uint16_t v00 = src[(size_t)iy * w + ix], v01 = src[(size_t)iy * w + ix1];
            uint16_t v10 = src[(size_t)iy1 * w + ix], v11 = src[(size_t)iy1 * w + ix1];
            double v = (1 - dx) * (1 - dy) * v00 + dx * (1 - dy) * v01 + (1 - dx) * dy * v10 + dx * dy * v11;
            /* static_cast<uchar>(double): the reference relies on x86 behaviour (cvttsd2si then truncate to 8 bits) */
            dst[(size_t)y * outW + x] = (uint8_t)(int)((v - mn) * scale8 + 0.5);
        }
    }
}

/* ---- A10: mask_to_image, src/process.cpp:178-185 ---- */
void orc_mask_to_image(const uint8_t *mask, size_t n, uint8_t *vis)
{
    uint8_t lut[256] = { 0 };
    lut[1] = 128; lut[2] = 255;
    for (size_t i = 0; i < n; ++i) vis[i] = lut[mask[i]];
}

/* ---- 8-connected components with bbox + area (cv::connectedComponentsWithStats(..., 8) semantics that the
 * reference consumes: membership, LEFT/TOP/WIDTH/HEIGHT, AREA; label numbering is not observable) ---- */
typedef struct { int left, top, right, bottom, area; } orc_cc_stat;

/* labels: int32 [h][w], 0 = background (fg[i]==0); returns number of labels incl. background.  stats[0] unused. */
static int label8(const uint8_t *fg, int w, int h, int32_t *labels, orc_cc_stat **stats_out)
{
    size_t n = (size_t)w * h;
    memset(labels, 0, n * sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc(n * sizeof(int32_t));
    int cap = 16, nc = 1;
    orc_cc_stat *st = (orc_cc_stat *)malloc(cap * sizeof(orc_cc_stat));
    for (int y0 = 0; y0 < h; ++y0)
        for (int x0 = 0; x0 < w; ++x0) {
            size_t s = (size_t)y0 * w + x0;
            if (!fg[s] || labels[s]) continue;
            if (nc == cap) { cap *= 2; st = (orc_cc_stat *)realloc(st, cap * sizeof(orc_cc_stat)); }
            orc_cc_stat c = { x0, y0, x0, y0, 0 };
            size_t sp = 0;
            stack[sp++] = (int32_t)s; labels[s] = nc;
            while (sp) {
                int32_t p = stack[--sp];
                int y = p / w, x = p % w;
                c.area++;
                if (x < c.left) c.left = x;
                if (x > c.right) c.right = x;
                if (y < c.top) c.top = y;
                if (y > c.bottom) c.bottom = y;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int yy = y + dy, xx = x + dx;
                        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                        size_t q = (size_t)yy * w + xx;
                        if (fg[q] && !labels[q]) { labels[q] = nc; stack[sp++] = (int32_t)q; }
                    }
            }
            st[nc++] = c;
        }
    free(stack);
    *stats_out = st;
    return nc;
}

/* exported for tests: per-pixel labels + a flat stats array [nc][5] = left, top, width, height, area */
int orc_connected_components8(const uint8_t *fg, int w, int h, int32_t *labels, int32_t *stats, int max_stats)
{
    orc_cc_stat *st;
    int nc = label8(fg, w, h, labels, &st);
    for (int i = 1; i < nc && i < max_stats; ++i) {
        stats[i * 5 + 0] = st[i].left; stats[i * 5 + 1] = st[i].top;
        stats[i * 5 + 2] = st[i].right - st[i].left + 1; stats[i * 5 + 3] = st[i].bottom - st[i].top + 1;
        stats[i * 5 + 4] = st[i].area;
    }
    free(st);
    return nc;
}

#define FOREGROUND_VALUE 2            /* src/postprocess.cpp:5 */
#define MIN_AREA_RATIO 0.06f          /* src/postprocess.cpp:9 */

/* ---- A8: fill_holes_inside_foreground, src/postprocess.cpp:13-44 ---- */
void orc_fill_holes(uint8_t *mask, int w, int h)
{
    size_t n = (size_t)w * h;
    uint8_t *inv = (uint8_t *)malloc(n);
    int32_t *labels = (int32_t *)malloc(n * sizeof(int32_t));
    for (size_t i = 0; i < n; ++i) inv[i] = (mask[i] == FOREGROUND_VALUE) ? 0 : 255;   /* :18-22 */
    orc_cc_stat *st;
    int nc = label8(inv, w, h, labels, &st);                                         /* :26 */
    const int min_area = (int)(w * h * MIN_AREA_RATIO);                              /* :30, int*float -> float */
    for (int i = 1; i < nc; ++i) {
        if (st[i].left > 0 && st[i].top > 0 && st[i].right < w - 1 && st[i].bottom < h - 1 &&
            st[i].area < min_area)                                                   /* :40 */
            for (size_t p = 0; p < n; ++p)
                if (labels[p] == i) mask[p] = FOREGROUND_VALUE;                      /* :41 */
    }
    free(st); free(labels); free(inv);
}

/* 3x3 rectangular erode / dilate with cv::morphologyDefaultBorderValue(): pixels outside the image never
 * constrain an erosion and never seed a dilation, i.e. min / max over the in-bounds part of the window. */
static void morph3x3(const uint8_t *src, uint8_t *dst, int w, int h, int dilate)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t v = dilate ? 0 : 255;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    uint8_t s = src[(size_t)yy * w + xx];
                    if (dilate ? (s > v) : (s < v)) v = s;
                }
            dst[(size_t)y * w + x] = v;
        }
}

/* MORPH_OPEN with a 3x3 MORPH_RECT element, src/postprocess.cpp:58-60 */
void orc_open3x3(const uint8_t *src, uint8_t *dst, int w, int h)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
    morph3x3(src, tmp, w, h, 0);
    morph3x3(tmp, dst, w, h, 1);
    free(tmp);
}

/* ---- A9: postprocess_mask, src/postprocess.cpp:47-79 ---- */
void orc_postprocess_mask(const uint8_t *src, uint8_t *out, int w, int h)
{
    size_t n = (size_t)w * h;
    uint8_t *mask = (uint8_t *)malloc(n), *bin = (uint8_t *)malloc(n), *opened = (uint8_t *)malloc(n);
    int32_t *labels = (int32_t *)malloc(n * sizeof(int32_t));
    memcpy(mask, src, n);                                                            /* :51 clone */
    orc_fill_holes(mask, w, h);                                                      /* :54 */
    for (size_t i = 0; i < n; ++i) bin[i] = (mask[i] == FOREGROUND_VALUE) ? 255 : 0; /* :57 */
    orc_open3x3(bin, opened, w, h);                                                  /* :58-60 */
    orc_cc_stat *st;
    int nc = label8(opened, w, h, labels, &st);                                      /* :64 */
    const int min_area = (int)(w * h * MIN_AREA_RATIO);                              /* :66 */
    memset(out, 0, n);                                                               /* :75 */
    for (int i = 1; i < nc; ++i)
        if (st[i].area >= min_area)                                                  /* :70 */
            for (size_t p = 0; p < n; ++p)
                if (labels[p] == i) out[p] = FOREGROUND_VALUE;                       /* :71, :76 */
    free(st); free(labels); free(opened); free(bin); free(mask);
}

/* ---- A11: Mask2Polygon::extract_contours, src/mask2polygon.cpp:29-36 ----
 * threshold(mask, 127, 255, THRESH_BINARY) then findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE).
 * OpenCV is not available here (PARITY UNPINNED against it); this restates the Suzuki-Abe border following the way
 * OpenCV's classic implementation documents it: the image is zero-padded by one pixel, scanned in raster order, an
 * outer border starts at an unlabelled foreground pixel whose west neighbour is 0, and in RETR_EXTERNAL mode it is
 * accepted only if the last labelled border pixel seen on this row is not a "still inside" (positive) one.  Border pixels
 * are labelled 2, or -126 when the trace leaves them with background on their east side.  The trace looks for its first
 * neighbour CLOCKWISE starting after west, every later one COUNTER-CLOCKWISE starting after the direction it came
 * from; CHAIN_APPROX_SIMPLE emits a point only where the step direction changes.  Contours are returned newest first
 * (each new contour is linked in front of its siblings).
 *
 * out_xy receives x,y pairs, contour after contour; out_start[c] is the index (in points) of contour c's first point,
 * out_start[ncontours] the total.  Returns the number of contours, or -1 if a capacity is too small. */
int orc_find_contours(const uint8_t *mask, int w, int h, int32_t *out_xy, int cap_points, int32_t *out_start,
                      int cap_contours)
{
    static const int DX[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };     /* 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (y grows down) */
    static const int DY[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };
    const int pw = w + 2, ph = h + 2;
    int8_t *img = (int8_t *)calloc((size_t)pw * ph, 1);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img[(size_t)(y + 1) * pw + x + 1] = mask[(size_t)y * w + x] > 127 ? 1 : 0;
    int32_t *pts = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)cap_points);   /* discovery order */
    int32_t *starts = (int32_t *)malloc(sizeof(int32_t) * ((size_t)cap_contours + 1));
    int nc = 0, np = 0, overflow = 0;
    for (int y = 1; y < ph - 1 && !overflow; ++y) {
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x < pw - 1; ++x) {
            const int p = img[(size_t)y * pw + x];
            if (p != prev && prev == 0 && p == 1 && !(img[(size_t)y * pw + lnbd_x] > 0)) {
                if (nc == cap_contours) { overflow = 1; break; }
                starts[nc++] = np;
                /* ---- follow the border starting at (x, y) */
                int8_t *i0 = img + (size_t)y * pw + x;
                int s = 4, s_end = 4;
                int8_t *i1;
                do { s = (s - 1) & 7; i1 = i0 + DY[s] * pw + DX[s]; } while (*i1 == 0 && s != s_end);
                int px = x, py = y;
                if (s == s_end) {                     /* isolated pixel */
                    *i0 = (int8_t)(2 | -128);
                    if (np == cap_points) { overflow = 1; break; }
                    pts[2 * np] = px - 1; pts[2 * np + 1] = py - 1; ++np;
                } else {
                    int8_t *i3 = i0, *i4;
                    int prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        for (;;) { ++s; i4 = i3 + DY[s & 7] * pw + DX[s & 7]; if (*i4 != 0) break; }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(2 | -128);   /* east side is background */
                        else if (*i3 == 1) *i3 = 2;
                        if (s != prev_s) {
                            if (np == cap_points) { overflow = 1; break; }
                            pts[2 * np] = px - 1; pts[2 * np + 1] = py - 1; ++np;
                            prev_s = s;
                        }
                        px += DX[s]; py += DY[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                    if (overflow) break;
                }
            }
            prev = img[(size_t)y * pw + x];
            if (prev & -2) lnbd_x = x;
        }
    }
    int ret = -1;
    if (!overflow) {
        starts[nc] = np;
        int o = 0;
        for (int c = nc - 1; c >= 0; --c) {           /* newest first */
            out_start[nc - 1 - c] = o;
            for (int k = starts[c]; k < starts[c + 1]; ++k) { out_xy[2 * o] = pts[2 * k]; out_xy[2 * o + 1] = pts[2 * k + 1]; ++o; }
        }
        out_start[nc] = o;
        ret = nc;
    }
    free(pts); free(starts); free(img);
    return ret;
}

/* ---- A12: map_contour_points, src/mask2polygon.cpp:41-63: int(pt * scale), truncation toward zero, double math ---- */
void orc_map_points(const int32_t *xy, int n, double scale_x, double scale_y, int32_t *out)
{
    for (int i = 0; i < n; ++i) {
        out[2 * i] = (int)(xy[2 * i] * scale_x);
        out[2 * i + 1] = (int)(xy[2 * i + 1] * scale_y);
    }
}
