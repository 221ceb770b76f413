// image_stages.hip -- the reference's CPU-side image stages on the device: RAW16 preprocessing (f1), postprocess_mask
// (f2), mask_to_image + extract_contours (f3).  Integer / byte / fp64 work, bit-exact against the oracle.  gfx950 only.
#include "kernel_common.h"

namespace miunet {

// --------------------------------------------------------------------------------------------------------------------
// RAW16 preprocessing on the device: HBM-bound integer scan + a 262144-pixel fp64 gather.
__global__ __launch_bounds__(256) void minmax_init_kernel(unsigned *mnmx)
{
    if (threadIdx.x == 0) { mnmx[0] = 65535u; mnmx[1] = 0u; }
}

__global__ __launch_bounds__(256) void minmax_u16_kernel(const uint16_t *__restrict__ raw, size_t n, unsigned *mnmx)
{
    __shared__ unsigned s_lo[4], s_hi[4];
    unsigned lo = 65535u, hi = 0u;
    const size_t n8 = n / 8;                                   // 16 bytes = 8 samples per lane
    const uint4 *v = reinterpret_cast<const uint4 *>(raw);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 q = v[i];
        const unsigned ws[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = ws[k] & 0xFFFFu, b = ws[k] >> 16;
            lo = min(lo, min(a, b));
            hi = max(hi, max(a, b));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {            // ragged tail
        const unsigned a = raw[n8 * 8 + threadIdx.x];
        lo = min(lo, a); hi = max(hi, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                          // wave64 butterfly
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {                                     // ONE atomic pair per workgroup: same-address atomics serialise
        atomicMin(&mnmx[0], min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])));
        atomicMax(&mnmx[1], max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
    }
}

hipError_t launch_minmax_u16(const uint16_t *raw, size_t n, unsigned *mnmx, hipStream_t s)
{
    if (n == 0 || (reinterpret_cast<uintptr_t>(raw) & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, mnmx);
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(minmax_u16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, raw, n, mnmx);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void resample_u8_kernel(const uint16_t *__restrict__ raw, int w, int h,
                                                          const unsigned *__restrict__ mnmx, uint8_t *__restrict__ dst,
                                                          int outW, int outH, int dst_stride)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= outW || y >= outH) return;
    const unsigned short mn = (unsigned short)mnmx[0];
    unsigned short mx = (unsigned short)mnmx[1];
    if (mn == mx) mx = (unsigned short)(mn + 1);                // evaluated in uint16_t: wraps to 0 at 65535 (src/preprocess.cpp:92)
    const double scale8 = 255.0 / (double)((int)mx - (int)mn);
    const double stepX = (double)w / (double)outW, stepY = (double)h / (double)outH;
    const double fx = __dmul_rn((double)x, stepX), fy = __dmul_rn((double)y, stepY);
    const int ix = (int)fx, iy = (int)fy;
    const int ix1 = ix + 1 < w - 1 ? ix + 1 : w - 1;
    const int iy1 = iy + 1 < h - 1 ? iy + 1 : h - 1;
    const double dx = __dsub_rn(fx, (double)ix), dy = __dsub_rn(fy, (double)iy);
    const double v00 = raw[(size_t)iy * w + ix], v01 = raw[(size_t)iy * w + ix1];
    const double v10 = raw[(size_t)iy1 * w + ix], v11 = raw[(size_t)iy1 * w + ix1];
    const double omdx = __dsub_rn(1.0, dx), omdy = __dsub_rn(1.0, dy);
    // (1-dx)*(1-dy)*v00 + dx*(1-dy)*v01 + (1-dx)*dy*v10 + dx*dy*v11, left to right, one rounding per operation
    double v = __dmul_rn(__dmul_rn(omdx, omdy), v00);
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(dx, omdy), v01));
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(omdx, dy), v10));
    v = __dadd_rn(v, __dmul_rn(__dmul_rn(dx, dy), v11));
    const double q = __dadd_rn(__dmul_rn(__dsub_rn(v, (double)mn), scale8), 0.5);
    dst[((size_t)y * outW + x) * dst_stride] = (uint8_t)(int)q;      // dst_stride > 1: one plane of an interleaved (HWC) tile
}

hipError_t launch_resample_u8(const uint16_t *raw, int w, int h, const unsigned *mnmx, uint8_t *dst, int outW, int outH,
                              int dst_stride, hipStream_t s)
{
    if (w <= 0 || h <= 0 || outW <= 0 || outH <= 0 || dst_stride < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((outW + 63) / 64, (outH + 3) / 4), dim3(256), 0, s, raw, w, h, mnmx, dst, outW, outH,
                       dst_stride);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// postprocess_mask on the device.  Byte/integer work, HBM/L2-bound and tiny next to the network:
// the point is to keep the label maps on the device and to replace the reference's O(components x H x W) loops
// (src/postprocess.cpp:41, :71) by one union-find labelling.
namespace pp {

__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// root of x with path halving: every store writes an ancestor of the node, so racing finds / unions stay consistent
__device__ __forceinline__ int find_root(int *parent, int x)
{
    int p = ld(parent + x);
    while (p != x) {
        const int g = ld(parent + p);
        if (g != p) st(parent + x, g);
        x = p; p = g;
    }
    return x;
}

// read-only walk for the flatten pass: there every store must be the final root, so no halving stores may race with it
__device__ __forceinline__ int find_root_ro(const int *parent, int x)
{
    for (int p = ld(parent + x); p != x; p = ld(parent + x)) x = p;
    return x;
}

// parents only ever decrease, roots satisfy parent[r] == r; atomicMin at L2 makes concurrent unions safe
__device__ __forceinline__ void unite(int *parent, int a, int b)
{
    for (;;) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang a under b
        const int old = atomicMin(parent + a, b);
        if (old == a) return;
        a = old;                                                // somebody re-parented a meanwhile: retry from there
    }
}

// fg[i] != 0 marks foreground.  parent = first pixel of the horizontal run inside the lane's 64-pixel segment (so the
// forest starts with chains no longer than W/64 per row instead of one node per pixel), -1 for background; stats cleared.
__global__ __launch_bounds__(256) void cc_init(const uint8_t *__restrict__ fg, int *parent, int *area, int *minx, int *miny,
                                               int *maxx, int *maxy, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    const bool f = in && fg[i] != 0;
    const int x = in ? (int)(i % W) : 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long fm = __ballot(f);
    const unsigned long long prev = (fm << 1) & ~__ballot(x == 0);      // lanes whose left neighbour (same row, same wave) is fg
    const unsigned long long starts = fm & ~prev;                        // first pixel of each run in this segment
    if (in) {
        int p = -1;
        if (f) {
            const unsigned long long below = starts & ((2ull << lane) - 1ull);
            p = (int)i - (lane - (63 - __builtin_clzll(below)));
        }
        parent[i] = p;
        area[i] = 0; minx[i] = 0x7FFFFFFF; miny[i] = 0x7FFFFFFF; maxx[i] = -1; maxy[i] = -1;
    }
}

// 8-connected unions.  Horizontal: only where a run continues across a 64-pixel segment boundary.  Vertical: with the
// pixel above if it is set -- skipped when the left neighbour and the upper-left pixel are set too (that pair already made
// the same connection) -- otherwise with the two upper diagonals.
__global__ __launch_bounds__(256) void cc_merge(const uint8_t *__restrict__ fg, int *parent, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !fg[i]) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    const bool l = x > 0 && fg[i - 1];
    if (l && (threadIdx.x & 63) == 0) unite(parent, (int)i, (int)i - 1);
    if (y > 0) {
        const bool ul = x > 0 && fg[i - W - 1], u = fg[i - W] != 0, ur = x + 1 < W && fg[i - W + 1];
        if (u) {
            if (!(l && ul)) unite(parent, (int)i, (int)i - W);
        } else {
            if (ul && !l) unite(parent, (int)i, (int)i - W - 1);      // with l set, the left pixel makes this union (as its 'u' or 'ur')
            if (ur) unite(parent, (int)i, (int)i - W + 1);
        }
    }
}

// flatten + per-component area and bounding box.  Lanes of a wave mostly share one root, and same-address atomics serialise
// (a mask is a handful of components: every wave of the image would queue on the same five words -- 0.25 ms per call at
// 4 x 1024^2, profiles/r04_image_stage_kernels.txt), so a wave reduces per distinct root inside a 64-pixel segment AND carries
// that root's sums over CC_RUN consecutive segments, issuing the five atomics only when the root changes.
constexpr int CC_RUN = 16;
__global__ __launch_bounds__(256) void cc_stats(int *parent, int *area, int *minx, int *miny, int *maxx, int *maxy, int H, int W,
                                                long long n)
{
    const int lane = threadIdx.x & 63;
    const long long first = (((long long)blockIdx.x * 256 + threadIdx.x) >> 6) * (64LL * CC_RUN);
    int ar = -1, an = 0, alx = 0x7FFFFFFF, ahx = -1, aly = 0x7FFFFFFF, ahy = -1;      // the carried root and its sums (wave-uniform)
    auto flush = [&]() {
        if (ar >= 0 && lane == 0) {
            atomicAdd(area + ar, an);
            atomicMin(minx + ar, alx); atomicMax(maxx + ar, ahx);
            atomicMin(miny + ar, aly); atomicMax(maxy + ar, ahy);
        }
    };
    for (int sg = 0; sg < CC_RUN; ++sg) {
        const long long i = first + 64LL * sg + lane;
        if (first + 64LL * sg >= n) break;                          // (wave-uniform)
        int r = -1, x = 0, y = 0;
        if (i < n && parent[i] >= 0) {
            r = find_root_ro(parent, (int)i);
            parent[i] = r;
            const int p = (int)(i % ((long long)H * W));
            y = p / W; x = p - y * W;
        }
        unsigned long long todo = __ballot(r >= 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int r0 = __shfl(r, leader, 64);
            const bool mine = r == r0;
            const unsigned long long m = __ballot(mine);
            int lx = mine ? x : 0x7FFFFFFF, hx = mine ? x : -1, ly = mine ? y : 0x7FFFFFFF, hy = mine ? y : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lx = min(lx, __shfl_xor(lx, o, 64)); hx = max(hx, __shfl_xor(hx, o, 64));
                ly = min(ly, __shfl_xor(ly, o, 64)); hy = max(hy, __shfl_xor(hy, o, 64));
            }
            if (r0 != ar) { flush(); ar = r0; an = 0; alx = 0x7FFFFFFF; ahx = -1; aly = 0x7FFFFFFF; ahy = -1; }
            an += __builtin_popcountll(m);
            alx = min(alx, lx); ahx = max(ahx, hx); aly = min(aly, ly); ahy = max(ahy, hy);
            todo &= ~m;
        }
    }
    flush();
}

__global__ __launch_bounds__(256) void k_inv(const uint8_t *__restrict__ labels, uint8_t *inv, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[i] = labels[i] == 2 ? 0 : 255;               // src/postprocess.cpp:18-22
}

// bin = 255 where the pixel is foreground after hole filling (src/postprocess.cpp:30-43, :57)
__global__ __launch_bounds__(256) void k_fill_bin(const uint8_t *__restrict__ labels, const int *__restrict__ parent,
                                                  const int *__restrict__ area, const int *__restrict__ minx,
                                                  const int *__restrict__ miny, const int *__restrict__ maxx,
                                                  const int *__restrict__ maxy, uint8_t *bin, int H, int W, int min_area, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool fgd = labels[i] == 2;
    const int r = parent[i];
    if (!fgd && r >= 0)
        fgd = minx[r] > 0 && miny[r] > 0 && maxx[r] < W - 1 && maxy[r] < H - 1 && area[r] < min_area;
    bin[i] = fgd ? 255 : 0;
}

template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph3(const uint8_t *__restrict__ src, uint8_t *dst, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    unsigned v = DILATE ? 0u : 255u;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;          // the border never constrains / never seeds
            const unsigned sv = src[i + dy * W + dx];
            v = DILATE ? max(v, sv) : min(v, sv);
        }
    dst[i] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_filter(const int *__restrict__ parent, const int *__restrict__ area, uint8_t *out,
                                                int min_area, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = parent[i];
    out[i] = (r >= 0 && area[r] >= min_area) ? 2 : 0;          // src/postprocess.cpp:70, :75-76
}

}  // namespace pp

size_t postprocess_workspace_bytes(int B, int H, int W)
{
    const size_t n = (size_t)B * H * W;
    return n * (6 * sizeof(int) + 3);                           // parent, area, 4 x bbox, three u8 planes
}

hipError_t launch_postprocess_masks(const uint8_t *labels_in, uint8_t *labels_out, int B, int H, int W, int min_area, void *ws,
                                    hipStream_t s)
{
    const long long n = (long long)B * H * W;
    if (n <= 0 || n > 0x7FFFFFFFLL) return hipErrorInvalidValue;
    int *parent = static_cast<int *>(ws), *area = parent + n, *minx = area + n, *miny = minx + n, *maxx = miny + n, *maxy = maxx + n;
    uint8_t *u0 = reinterpret_cast<uint8_t *>(maxy + n), *u1 = u0 + n, *u2 = u1 + n;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    auto label = [&](const uint8_t *fg) {
        hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, fg, parent, area, minx, miny, maxx, maxy, W, n);
        hipLaunchKernelGGL(pp::cc_merge, g, b, 0, s, fg, parent, H, W, n);
        hipLaunchKernelGGL(pp::cc_stats, dim3((unsigned)((n + 256LL * pp::CC_RUN - 1) / (256LL * pp::CC_RUN))), b, 0, s, parent, area, minx, miny, maxx, maxy, H, W, n);
    };
    hipLaunchKernelGGL(pp::k_inv, g, b, 0, s, labels_in, u0, n);
    label(u0);
    hipLaunchKernelGGL(pp::k_fill_bin, g, b, 0, s, labels_in, parent, area, minx, miny, maxx, maxy, u1, H, W, min_area, n);
    hipLaunchKernelGGL(pp::k_morph3<false>, g, b, 0, s, u1, u2, H, W, n);
    hipLaunchKernelGGL(pp::k_morph3<true>, g, b, 0, s, u2, u1, H, W, n);
    label(u1);
    hipLaunchKernelGGL(pp::k_filter, g, b, 0, s, parent, area, labels_out, min_area, n);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------------
// extract_contours on the device.
// OpenCV's sequential Suzuki-Abe scan interleaves "find the next start pixel" with "follow that border".  For
// RETR_EXTERNAL both halves separate cleanly:
//   * the start pixel of a component's outer border is its first pixel in raster order = the ROOT of the union-find
//     labelling above (parents always point to smaller indices);
//   * the border is external iff the background region just above that pixel reaches the image frame (OpenCV's
//     `img0[lnbd] > 0` test says the same thing through the sign of the last border label on the row): the pixel above
//     the root is background by construction, and a 4-connected background component reaches the frame iff its
//     bounding box touches the image edge;
//   * the trace itself (first neighbour clockwise from west, then counter-clockwise from the arrival direction, a point
//     wherever the step direction changes) is inherently serial per contour, so every external contour gets its own
//     lane: contours and images run in parallel, the mask is L2-resident (256 KB per image);
//   * OpenCV returns contours newest-first = descending raster order of the start pixels: a 64-bit-free trick -- the
//     per-image list of external roots is sorted by one thread per image (a handful of entries after postprocess_mask).
namespace ct {

__global__ __launch_bounds__(256) void k_threshold(const uint8_t *__restrict__ mask, uint8_t *fg, uint8_t *bg, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool f = mask[i] > 127;                               // cv::threshold(127, 255, THRESH_BINARY)
    fg[i] = f ? 255 : 0;
    bg[i] = f ? 0 : 255;
}

// 4-connected unions (background regions): across segment boundaries horizontally; vertically unless the left pair
// already made the connection
__global__ __launch_bounds__(256) void cc_merge4(const uint8_t *__restrict__ fg, int *parent, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !fg[i]) return;
    const int hw = H * W;
    const int p = (int)(i % hw), y = p / W, x = p - y * W;
    const bool l = x > 0 && fg[i - 1];
    if (l && (threadIdx.x & 63) == 0) pp::unite(parent, (int)i, (int)i - 1);
    if (y > 0 && fg[i - W] && !(l && fg[i - W - 1])) pp::unite(parent, (int)i, (int)i - W);
}

__global__ __launch_bounds__(256) void k_zero_counts(int *counts, int B)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) counts[i] = 0;
}

// a background component reaches the image frame iff one of the frame's pixels belongs to it: every background pixel of the frame
// flags its root (`flag`: zeroed by cc_init).  2 (H + W) pixels per image instead of a statistics pass over all of them.
__global__ __launch_bounds__(256) void k_frame_flags(const int *__restrict__ bparent, int *flag, int H, int W, int B)
{
    const int per = 2 * (H + W), t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * per) return;
    const int img = t / per, k = t - img * per;
    const int y = k < W ? 0 : k < 2 * W ? H - 1 : k < 2 * W + H ? k - 2 * W : k - 2 * W - H;
    const int x = k < W ? k : k < 2 * W ? k - W : k < 2 * W + H ? 0 : W - 1;
    const int p = img * H * W + y * W + x;
    if (bparent[p] >= 0) flag[pp::find_root_ro(bparent, p)] = 1;
}

// one entry per external component: its root (= start pixel).  fparent: fg labelling (not flattened: only `is a root` is asked);
// bparent + flag: background labelling (not flattened either) and which of its roots reach the frame.
__global__ __launch_bounds__(256) void k_collect(const int *__restrict__ fparent, const int *__restrict__ bparent,
                                                 const int *__restrict__ flag, int *roots, int *counts, int cap, int H, int W, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || fparent[i] != (int)i) return;                // roots only
    const int hw = H * W, img = (int)(i / hw);
    const int p = (int)(i % hw), y = p / W;
    bool external = (y == 0);
    if (!external)                                              // the pixel above a component's first pixel is background
        external = bparent[i - W] >= 0 && flag[pp::find_root_ro(bparent, (int)(i - W))] != 0;
    if (external) {
        const int slot = atomicAdd(counts + img, 1);
        if (slot < cap) roots[(size_t)img * cap + slot] = p;
    }
}

// per image: sort the start pixels descending (newest contour first).  Few entries: insertion sort by one lane.
__global__ __launch_bounds__(64) void k_sort_roots(int *roots, const int *counts, int cap, int B)
{
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= B) return;
    const int n = counts[img] < cap ? counts[img] : cap;
    int *r = roots + (size_t)img * cap;
    for (int a = 1; a < n; ++a) {
        const int v = r[a];
        int b = a - 1;
        while (b >= 0 && r[b] < v) { r[b + 1] = r[b]; --b; }
        r[b + 1] = v;
    }
}

// one lane per (image, contour): follow the border from its start pixel, emit CHAIN_APPROX_SIMPLE points.
// pass 0 counts points (npts) -- and already WRITES contour 0, whose offset is 0 whatever the counts are; pass 1 writes the
// others at the offsets computed in between and returns at once for images with a single contour (the usual case behind
// postprocess_mask, which keeps components of at least 6 % of the tile).
// The walk is a chain of dependent pixel probes (2-8 per border step), so the probe latency IS the kernel time: a workgroup
// = one image first packs the thresholded mask into an LDS bit plane (H*W/8 bytes: 32 KB at 512x512; images past the LDS
// budget keep probing global memory), then the lanes that own a contour walk it at LDS latency.
// PAD (IN_LDS and W % 32 == 0, so that rows are whole words): the plane carries one zero row above the image and one below it.
// Every row of a 3 x 3 probe then exists, the window's first bit is never negative, and the three rows of a probe share their
// shift -- the probe of a border step loses its row tests and selects.
template <bool IN_LDS, bool PAD = false>
__global__ __launch_bounds__(256) void k_trace(const uint8_t *__restrict__ fg, const int *__restrict__ roots,
                                               const int *__restrict__ counts, int cap, int H, int W, int B, int *npts,
                                               const int *__restrict__ offs, int *out_xy, int cap_points, int write)
{
    extern __shared__ unsigned bits[];
    const int img = blockIdx.x;
    const uint8_t *im = fg + (size_t)img * H * W;
    const int nc = counts[img];
    if (nc <= 0 || nc > cap) return;                             // workgroup-uniform
    if (write && nc == 1) return;                                // contour 0 was written by the count pass
    static_assert(!PAD || IN_LDS, "the padded plane lives in LDS");
    const int padw = PAD ? (W >> 5) : 0;                         // words of the zero row in front of the image
    if constexpr (IN_LDS) {
        const int nwords = (H * W + 31) >> 5;
        if (threadIdx.x == 0) bits[nwords + 2 * padw] = 0;       // one word past the plane: the 64-bit windows below may touch it
        if constexpr (PAD)
            for (int w = threadIdx.x; w < padw; w += 256) { bits[w] = 0; bits[padw + nwords + w] = 0; }
        for (int w = threadIdx.x; w < nwords; w += 256) {
            unsigned m = 0;
            const int base = w << 5;
            if (base + 32 <= H * W && (((uintptr_t)(im + base)) & 15) == 0) {
                const uint4 q0 = *reinterpret_cast<const uint4 *>(im + base), q1 = *reinterpret_cast<const uint4 *>(im + base + 16);
                const unsigned ws[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) m |= (((ws[k] >> (8 * b4)) & 0xFFu) != 0 ? 1u : 0u) << (4 * k + b4);
            } else {
                for (int k = 0; k < 32 && base + k < H * W; ++k) m |= (im[base + k] != 0 ? 1u : 0u) << k;
            }
            bits[padw + w] = m;
        }
        __syncthreads();
    }
    auto at = [&](int x, int y) -> bool {
        if (x < 0 || x >= W || y < 0 || y >= H) return false;
        const int p = y * W + x;
        if constexpr (IN_LDS) return (bits[padw + (p >> 5)] >> (p & 31)) & 1u;
        else return im[p] != 0;
    };
    // direction d: 0 = E, then counter-clockwise on screen (y grows down): DX = {1,1,0,-1,-1,-1,0,1}, DY = {0,-1,-1,-1,0,1,1,1}.
    // Packed as nibbles of (value + 1): a dynamically indexed array would live in scratch memory, and every probe of the walk
    // would pay a memory round trip for its offsets.
    auto DX = [](int d) -> int { return (int)((0x21000122u >> (4 * d)) & 3u) - 1; };
    auto DY = [](int d) -> int { return (int)((0x22210001u >> (4 * d)) & 3u) - 1; };
    for (int c = threadIdx.x; c < nc; c += 256) {               // lanes own contours (a handful after postprocess_mask)
    if (write && c == 0) continue;                               // written by the count pass
    const int start = roots[(size_t)img * cap + c];
    const int x0 = start % W, y0 = start / W;
    int *dst = nullptr;
    int room = 0;
    const bool store = write || c == 0;
    if (store) {
        const int o = write ? offs[(size_t)img * (cap + 1) + c] : 0;
        room = cap_points - o;
        dst = out_xy + ((size_t)img * cap_points + o) * 2;
    }
    int n = 0;
    auto emit = [&](int x, int y) {
        if (store && n < room) { dst[2 * n] = x; dst[2 * n + 1] = y; }
        ++n;
    };
    // The walk is a chain of dependent steps; inside a step the eight neighbour probes are INDEPENDENT reads (issued together,
    // one LDS latency), and the search order -- counter-clockwise from the arrival direction -- is a rotate + find-first-set
    // on the resulting 8-bit mask instead of up to eight chained probes.
    // three pixels (x-1, x, x+1) of row y as bits 0..2: the rows of the plane are packed back to back, so they are three
    // CONSECUTIVE bits of one 64-bit window (two LDS words), whatever W is; columns outside the image are masked off
    // Branch-free: a row outside the image reads the plane's first words and is masked to zero afterwards, so the six LDS reads of a
    // step are issued together and waited for ONCE (with an early return per row hipcc waited three times per border pixel).
    auto nb8 = [&](int x, int y) -> unsigned {                  // bit d = neighbour in direction d is foreground
        if constexpr (PAD) {
            const unsigned cols = (x > 0 ? 7u : 6u) & (x + 1 < W ? 7u : 3u);
            const int q = (y + 1) * W + x - 1;                  // row y of the image is row y + 1 of the plane: q >= W - 1
            const unsigned *const b = bits + (q >> 5);
            const int sh = q & 31;
            unsigned r[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned *const bk = b + (k - 1) * padw;
                r[k] = (unsigned)((((unsigned long long)bk[1] << 32) | bk[0]) >> sh) & cols;
            }
            return ((r[1] >> 2) & 1u) | (((r[0] >> 2) & 1u) << 1) | (((r[0] >> 1) & 1u) << 2) | ((r[0] & 1u) << 3) | ((r[1] & 1u) << 4) |
                   ((r[2] & 1u) << 5) | (((r[2] >> 1) & 1u) << 6) | (((r[2] >> 2) & 1u) << 7);
        } else if constexpr (IN_LDS) {
            const unsigned cols = (x > 0 ? 7u : 6u) & (x + 1 < W ? 7u : 3u);
            unsigned r[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int yy = y - 1 + k;
                const bool ok = (unsigned)yy < (unsigned)H;
                const int q = yy * W + x - 1;                   // first bit of the window; -1 only at the image origin
                const int qc = ok ? (q > 0 ? q : 0) : 0;
                const unsigned lo = bits[qc >> 5], hi = bits[(qc >> 5) + 1];
                unsigned v = (unsigned)((((unsigned long long)hi << 32) | lo) >> (qc & 31));
                if (q < 0) v <<= 1;
                r[k] = ok ? (v & cols) : 0u;
            }
            return ((r[1] >> 2) & 1u) | (((r[0] >> 2) & 1u) << 1) | (((r[0] >> 1) & 1u) << 2) | ((r[0] & 1u) << 3) | ((r[1] & 1u) << 4) |
                   ((r[2] & 1u) << 5) | (((r[2] >> 1) & 1u) << 6) | (((r[2] >> 2) & 1u) << 7);
        } else {
            unsigned m = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) m |= (at(x + DX(d), y + DY(d)) ? 1u : 0u) << d;
            return m;
        }
    };
    int first = -1;                                             // first neighbour: clockwise, starting after west
    {
        const unsigned m = nb8(x0, y0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = (3 - k) & 7;                          // 3, 2, 1, 0, 7, 6, 5, 4
            if (first < 0 && ((m >> d) & 1u)) first = d;
        }
    }
    if (first < 0) {
        emit(x0, y0);                                           // isolated pixel
    } else {
        const int x1 = x0 + DX(first), y1 = y0 + DY(first);
        int cx = x0, cy = y0, came = first, last_step = first ^ 4;
        for (long long guard = 0; guard < 4LL * H * W + 16; ++guard) {      // a border has at most 4 visits per pixel
            const unsigned m = nb8(cx, cy);
            if (m == 0) break;                                  // cannot happen on a border pixel; keeps the loop finite on bad input
            const int base = (came + 1) & 7;                    // search came+1, came+2, ... (counter-clockwise)
            const unsigned rot = ((m | (m << 8)) >> base) & 0xFFu;
            const int step = (base + __builtin_ctz(rot)) & 7;
            const int nx = cx + DX(step), ny = cy + DY(step);
            if (step != last_step) { emit(cx, cy); last_step = step; }
            const bool closing = (nx == x0 && ny == y0 && cx == x1 && cy == y1);
            cx = nx; cy = ny;
            if (closing) break;
            came = (step + 4) & 7;
        }
    }
    if (!write) npts[(size_t)img * cap + c] = n;
    }
}

// per image: exclusive scan of the point counts -> out_start, and the verdict (count or -1 on overflow)
__global__ __launch_bounds__(64) void k_offsets(const int *__restrict__ npts, const int *__restrict__ counts, int cap,
                                                int cap_points, int B, int *out_start, int *out_count)
{
    const int img = blockIdx.x * 64 + threadIdx.x;
    if (img >= B) return;
    int *st = out_start + (size_t)img * (cap + 1);
    const int nc = counts[img];
    if (nc > cap) { out_count[img] = -1; st[0] = 0; return; }
    int o = 0;
    for (int c = 0; c < nc; ++c) { st[c] = o; o += npts[(size_t)img * cap + c]; }
    st[nc] = o;
    out_count[img] = o > cap_points ? -1 : nc;
}

}  // namespace ct

__global__ __launch_bounds__(256) void mask_to_image_kernel(const uint8_t *__restrict__ labels, uint8_t *vis, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned v = labels[i];
    vis[i] = v == 1 ? 128 : v == 2 ? 255 : 0;
}

hipError_t launch_mask_to_image(const uint8_t *labels, uint8_t *vis, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(mask_to_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, vis, n);
    return hipGetLastError();
}

size_t contour_workspace_bytes(int B, int H, int W, int cap_contours)
{
    const size_t n = (size_t)B * H * W;
    return n * (7 * sizeof(int) + 2) + sizeof(int) * ((size_t)B * (2 * cap_contours + 1) + 64);
}

hipError_t launch_extract_contours(const uint8_t *masks, int B, int H, int W, int *out_xy, int cap_points, int *out_start,
                                   int cap_contours, int *out_count, void *ws, hipStream_t s)
{
    const long long n = (long long)B * H * W;
    if (n <= 0 || n > 0x7FFFFFFFLL || cap_contours <= 0 || cap_points <= 0) return hipErrorInvalidValue;
    int *fparent = static_cast<int *>(ws), *bparent = fparent + n, *area = bparent + n, *minx = area + n, *miny = minx + n,
        *maxx = miny + n, *maxy = maxx + n;
    uint8_t *fg = reinterpret_cast<uint8_t *>(maxy + n), *bg = fg + n;
    int *roots = reinterpret_cast<int *>(bg + n + ((16 - (2 * n) % 16) % 16));
    int *npts = roots + (size_t)B * cap_contours, *counts = npts + (size_t)B * cap_contours;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    hipLaunchKernelGGL(ct::k_threshold, g, b, 0, s, masks, fg, bg, n);
    // foreground labelling (8-connected): only its roots are asked for (no statistics pass, no flattening)
    hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, fg, fparent, area, minx, miny, maxx, maxy, W, n);
    hipLaunchKernelGGL(pp::cc_merge, g, b, 0, s, fg, fparent, H, W, n);
    // background labelling (4-connected); `area` (zeroed by cc_init) becomes the reaches-the-frame flag of its roots
    hipLaunchKernelGGL(pp::cc_init, g, b, 0, s, bg, bparent, area, minx, miny, maxx, maxy, W, n);
    hipLaunchKernelGGL(ct::cc_merge4, g, b, 0, s, bg, bparent, H, W, n);
    hipLaunchKernelGGL(ct::k_frame_flags, dim3((unsigned)((B * 2 * (H + W) + 255) / 256)), b, 0, s, bparent, area, H, W, B);
    hipLaunchKernelGGL(ct::k_zero_counts, dim3((B + 255) / 256), b, 0, s, counts, B);
    hipLaunchKernelGGL(ct::k_collect, g, b, 0, s, fparent, bparent, area, roots, counts, cap_contours, H, W, n);
    hipLaunchKernelGGL(ct::k_sort_roots, dim3((B + 63) / 64), dim3(64), 0, s, roots, counts, cap_contours, B);
    // one workgroup per image; the mask as an LDS bit plane when it fits
    const size_t plane = (((size_t)H * W + 31) / 32) * 4 + 4;    // + one word of slack behind the plane (k_trace's windows)
    const bool in_lds = plane <= 160 * 1024;                     // the whole LDS of a CU: 1024x1024 is 128 KB + 4
    const size_t padded = plane + 2 * (size_t)(W / 32) * 4;      // + a zero row above and below (k_trace<true, true>)
    const bool pad = W % 32 == 0 && padded <= 160 * 1024;
    const dim3 gt((unsigned)B), bt(256);
    for (int pass = 0; pass < 2; ++pass) {
        if (pad) {
            if (hipError_t e = ensure_dynamic_lds(ct::k_trace<true, true>, padded); e != hipSuccess) return e;
            hipLaunchKernelGGL((ct::k_trace<true, true>), gt, bt, padded, s, fg, roots, counts, cap_contours, H, W, B, npts, out_start, out_xy, cap_points, pass);
        } else if (in_lds) {
            if (hipError_t e = ensure_dynamic_lds(ct::k_trace<true>, plane); e != hipSuccess) return e;
            hipLaunchKernelGGL(ct::k_trace<true>, gt, bt, plane, s, fg, roots, counts, cap_contours, H, W, B, npts, out_start, out_xy, cap_points, pass);
        } else {
            hipLaunchKernelGGL(ct::k_trace<false>, gt, bt, 0, s, fg, roots, counts, cap_contours, H, W, B, npts, out_start, out_xy, cap_points, pass);
        }
        if (pass == 0)
            hipLaunchKernelGGL(ct::k_offsets, dim3((B + 63) / 64), dim3(64), 0, s, npts, counts, cap_contours, cap_points, B, out_start, out_count);
    }
    return hipGetLastError();
}


}  // namespace miunet
