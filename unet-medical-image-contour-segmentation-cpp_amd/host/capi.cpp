// capi.cpp -- extern "C" wrappers of include/medseg_c.h around the C++ facade.
#include "../../include/medseg_c.h"

#include <cstring>
#include <string>

#include "../../include/medseg/cleanup.h"
#include "../../include/medseg/initialize.h"
#include "../../include/medseg/mask2polygon.h"
#include "../../include/medseg/postprocess.h"
#include "../../include/medseg/preprocess.h"
#include "../../include/medseg/process.h"
#include "png_io.h"

using medseg::Contour;
using medseg::Image8;

namespace {
Image8 wrap(const uint8_t *p, int w, int h, int ch = 1)
{
    Image8 m(h, w, ch);
    memcpy(m.data.data(), p, m.data.size());
    return m;
}
std::vector<Contour> unflatten(const int32_t *xy, const int32_t *start, int n)
{
    std::vector<Contour> cs(n);
    for (int c = 0; c < n; ++c)
        for (int k = start[c]; k < start[c + 1]; ++k) cs[c].emplace_back(xy[2 * k], xy[2 * k + 1]);
    return cs;
}
thread_local std::string t_log_path;
}  // namespace

extern "C" {

int medseg_initialize_engine(const char *weight_path, const char *log_dir)
{
    return MedicalSeg::initialize_engine(weight_path, log_dir) ? 0 : 1;
}
int medseg_process_single_image(const char *raw_path, int width, int height, const char *output_dir)
{
    return MedicalSeg::process_single_image(raw_path, width, height, output_dir) ? 0 : 1;
}
int medseg_process_image_batch(const char *const *raw_paths, const int *widths, const int *heights, int n, const char *output_dir)
{
    std::vector<std::string> p(raw_paths, raw_paths + n);
    return MedicalSeg::process_image_batch(p, std::vector<int>(widths, widths + n), std::vector<int>(heights, heights + n), output_dir);
}
void medseg_cleanup_resources(void) { MedicalSeg::cleanup_resources(); }
const char *medseg_get_log_path(void)
{
    t_log_path = MedicalSeg::get_log_path();
    return t_log_path.c_str();
}

int medseg_preprocess_raw(const char *raw_path, const char *png_path, const char *json_path, int w, int h)
{
    return Preprocess::preprocess_raw(raw_path, png_path, json_path, w, h) ? 0 : 1;
}
int medseg_resample_normalize(const uint16_t *src, int w, int h, uint8_t *dst, int out_w, int out_h)
{
    if (!src || !dst || w <= 0 || h <= 0 || out_w <= 0 || out_h <= 0) return 1;
    const Image8 r = Preprocess::resample_normalize(src, w, h, out_w, out_h);
    memcpy(dst, r.data.data(), r.data.size());
    return 0;
}

int medseg_postprocess_mask(const uint8_t *mask, int w, int h, uint8_t *out)
{
    try {
        const Image8 r = postprocess_mask(wrap(mask, w, h));
        memcpy(out, r.data.data(), r.data.size());
        return 0;
    } catch (...) { return 1; }
}
int medseg_mask_to_image(const uint8_t *mask, int w, int h, uint8_t *out)
{
    const Image8 r = MedicalSeg::mask_to_image(wrap(mask, w, h));
    memcpy(out, r.data.data(), r.data.size());
    return 0;
}

int medseg_extract_contours(const uint8_t *mask, int w, int h, int32_t *xy, int cap_points, int32_t *start, int cap_contours)
{
    const std::vector<Contour> cs = Mask2Polygon::extract_contours(wrap(mask, w, h));
    if ((int)cs.size() > cap_contours) return -1;
    int o = 0;
    for (size_t c = 0; c < cs.size(); ++c) {
        start[c] = o;
        if (o + (int)cs[c].size() > cap_points) return -1;
        for (const auto &p : cs[c]) { xy[2 * o] = p.x; xy[2 * o + 1] = p.y; ++o; }
    }
    start[cs.size()] = o;
    return (int)cs.size();
}
void medseg_map_points(const int32_t *xy, int n, double scale_x, double scale_y, int32_t *out)
{
    Contour c;
    for (int i = 0; i < n; ++i) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
    const auto m = Mask2Polygon::map_contour_points({ c }, scale_x, scale_y);
    for (int i = 0; i < n; ++i) { out[2 * i] = m[0][i].x; out[2 * i + 1] = m[0][i].y; }
}
int medseg_generate_json(const int32_t *xy, const int32_t *start, int ncontours, const char *json_path, const char *base_name,
                         int original_width, int original_height)
{
    try {
        Mask2Polygon::generate_json(unflatten(xy, start, ncontours), json_path, base_name, original_width, original_height);
        return 0;
    } catch (...) { return 1; }
}
int medseg_draw_overlay(const uint8_t *gray, int w, int h, const int32_t *xy, const int32_t *start, int ncontours, uint8_t *bgr_out)
{
    try {
        const Image8 r = Mask2Polygon::draw_overlay(wrap(gray, w, h), unflatten(xy, start, ncontours));
        memcpy(bgr_out, r.data.data(), r.data.size());
        return 0;
    } catch (...) { return 1; }
}
void medseg_process_single_mask(const char *mask_path, const char *output_dir, const char *json_path, const char *original_png,
                                const char *base_name)
{
    Mask2Polygon::process_single_mask(mask_path, output_dir, json_path, original_png ? original_png : "", base_name);
}
int medseg_write_png(const char *path, const uint8_t *data, int w, int h, int channels, int level0)
{
    return medseg::write_png(path, wrap(data, w, h, channels), level0 != 0) ? 0 : 1;
}
int medseg_read_png(const char *path, int as_color, uint8_t *data, int cap_bytes, int *w, int *h)
{
    const Image8 r = medseg::read_png(path, as_color != 0);
    if (r.empty() || (int)r.data.size() > cap_bytes) return 1;
    memcpy(data, r.data.data(), r.data.size());
    *w = r.cols; *h = r.rows;
    return 0;
}

}  // extern "C"
