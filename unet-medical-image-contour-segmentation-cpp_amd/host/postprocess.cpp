// postprocess.cpp -- label map -> cleaned {0,2} mask.  Reference: src/postprocess.cpp:5-79 (OpenCV there).
// Own algorithms: run-based two-pass union-find labelling with per-component bbox/area (instead of one full-image
// `labels == i` pass per component, the reference's O(nc*H*W) loops at :41 and :71), separable 3x3 min/max.
#include "../../include/medseg/postprocess.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>

namespace {

constexpr int FOREGROUND_VALUE = 2;          // src/postprocess.cpp:5
constexpr float MIN_AREA_RATIO = 0.06f;      // src/postprocess.cpp:9

struct Stat { int left, top, right, bottom, area; };

struct Labelling {
    std::vector<int> labels;                 // per pixel, 0 = background, otherwise a ROOT id
    std::vector<Stat> stats;                 // indexed by root id (entries of non-roots unused)
};

int find(std::vector<int> &p, int a)
{
    while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; }
    return a;
}

// 8-connected components of fg != 0 (cv::connectedComponentsWithStats(..., 8): membership, bbox, area).
Labelling label8(const uint8_t *fg, int w, int h)
{
    Labelling L;
    L.labels.assign((size_t)w * h, 0);
    std::vector<int> parent(1, 0);
    for (int y = 0; y < h; ++y) {
        int *row = L.labels.data() + (size_t)y * w;
        const int *up = y ? row - w : nullptr;
        const uint8_t *f = fg + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (!f[x]) continue;
            int lab = 0;
            auto join = [&](int other) {
                if (!other) return;
                other = find(parent, other);
                if (!lab) lab = other;
                else if (lab != other) { const int a = std::min(lab, other), b = std::max(lab, other); parent[b] = a; lab = a; }
            };
            if (x) join(row[x - 1]);
            if (up) {
                if (x) join(up[x - 1]);
                join(up[x]);
                if (x + 1 < w) join(up[x + 1]);
            }
            if (!lab) { lab = (int)parent.size(); parent.push_back(lab); }
            row[x] = lab;
        }
    }
    L.stats.assign(parent.size(), Stat{ w, h, -1, -1, 0 });
    for (int y = 0; y < h; ++y) {
        int *row = L.labels.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (!row[x]) continue;
            const int r = find(parent, row[x]);
            row[x] = r;
            Stat &s = L.stats[r];
            s.left = std::min(s.left, x); s.right = std::max(s.right, x);
            s.top = std::min(s.top, y); s.bottom = std::max(s.bottom, y);
            ++s.area;
        }
    }
    return L;
}

// 3x3 rectangular erode (is_max = false) / dilate (true); windows are clipped to the image (OpenCV's default morphology
// border never constrains an erosion and never seeds a dilation).
void morph3x3(const std::vector<uint8_t> &src, std::vector<uint8_t> &dst, int w, int h, bool is_max)
{
    std::vector<uint8_t> tmp(src.size());
    auto pick = [is_max](uint8_t a, uint8_t b) { return is_max ? std::max(a, b) : std::min(a, b); };
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = &src[(size_t)y * w];
        uint8_t *t = &tmp[(size_t)y * w];
        for (int x = 0; x < w; ++x) {
            uint8_t v = s[x];
            if (x) v = pick(v, s[x - 1]);
            if (x + 1 < w) v = pick(v, s[x + 1]);
            t[x] = v;
        }
    }
    dst.resize(src.size());
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t v = tmp[(size_t)y * w + x];
            if (y) v = pick(v, tmp[(size_t)(y - 1) * w + x]);
            if (y + 1 < h) v = pick(v, tmp[(size_t)(y + 1) * w + x]);
            dst[(size_t)y * w + x] = v;
        }
}

}  // namespace

medseg::Image8 postprocess_mask(const medseg::Image8 &src)
{
    if (src.empty() || src.channels != 1) throw std::runtime_error("postprocess_mask: need a non-empty single-channel mask");
    const int w = src.cols, h = src.rows;
    const size_t n = (size_t)w * h;
    const int min_area = static_cast<int>(w * h * MIN_AREA_RATIO);          // evaluated in float, src/postprocess.cpp:30,:66
    std::vector<uint8_t> mask(src.data);

    // 1. fill holes: components of (mask != 2) whose bbox touches no edge and whose area < min_area
    {
        std::vector<uint8_t> inv(n);
        for (size_t i = 0; i < n; ++i) inv[i] = mask[i] == FOREGROUND_VALUE ? 0 : 255;
        const Labelling L = label8(inv.data(), w, h);
        std::vector<uint8_t> fill(L.stats.size(), 0);
        for (size_t r = 1; r < L.stats.size(); ++r) {
            const Stat &s = L.stats[r];
            fill[r] = s.area > 0 && s.left > 0 && s.top > 0 && s.right < w - 1 && s.bottom < h - 1 && s.area < min_area;
        }
        for (size_t i = 0; i < n; ++i)
            if (L.labels[i] && fill[L.labels[i]]) mask[i] = FOREGROUND_VALUE;
    }
    // 2. binarise + 3x3 open
    std::vector<uint8_t> bin(n), er, op;
    for (size_t i = 0; i < n; ++i) bin[i] = mask[i] == FOREGROUND_VALUE ? 255 : 0;
    morph3x3(bin, er, w, h, false);
    morph3x3(er, op, w, h, true);
    // 3. area filter, 4. map back to {0, 2}
    const Labelling L = label8(op.data(), w, h);
    medseg::Image8 out(h, w, 1, 0);
    for (size_t i = 0; i < n; ++i)
        if (L.labels[i] && L.stats[L.labels[i]].area >= min_area) out.data[i] = FOREGROUND_VALUE;
    return out;
}
