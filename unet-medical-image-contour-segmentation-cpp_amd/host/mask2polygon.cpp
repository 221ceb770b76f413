// mask2polygon.cpp -- mask PNG -> external contours -> labelme-style JSON (+ overlay).  Reference: src/mask2polygon.cpp.
// cv::threshold / cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) / cv::drawContours are re-implemented here
// (no OpenCV in this build): Suzuki-Abe border following with OpenCV's start pixel, search directions, SIMPLE
// compression and newest-first contour order.
#include "../../include/medseg/mask2polygon.h"

#include <cstring>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "json_io.h"
#include "png_io.h"

namespace fs = std::filesystem;
using medseg::Contour;
using medseg::Image8;
using medseg::Point;

namespace Mask2Polygon {

namespace {

// direction codes: 0 = E, then counter-clockwise on screen (y grows downwards): NE, N, NW, W, SW, S, SE
constexpr int kDx[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
constexpr int kDy[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };
constexpr int8_t kBorder = 2;                      // visited border pixel
constexpr int8_t kBorderExit = (int8_t)(2 | -128); // visited, and the trace left it with background to its east

class BorderTracer {
public:
    BorderTracer(const Image8 &mask) : w_(mask.cols + 2), h_(mask.rows + 2), img_((size_t)w_ * h_, 0)
    {
        for (int y = 0; y < mask.rows; ++y) {
            const uint8_t *s = mask.ptr(y);
            int8_t *d = &img_[(size_t)(y + 1) * w_ + 1];
            for (int x = 0; x < mask.cols; ++x) d[x] = s[x] > 127 ? 1 : 0;     // cv::threshold(127, 255, THRESH_BINARY)
        }
    }

    std::vector<Contour> run()
    {
        std::vector<Contour> found;
        for (int y = 1; y < h_ - 1; ++y) {
            int prev = 0;
            int last_border_x = 0;                 // x of the last visited-border pixel on this row (column 0 is padding)
            for (int x = 1; x < w_ - 1; ++x) {
                const int p = px(x, y);
                // an unvisited foreground pixel right of background starts an outer border; in external mode it only
                // counts when we are not inside an already traced component (last border pixel seen is an "exit" one)
                if (prev == 0 && p == 1 && px(last_border_x, y) <= 0) found.push_back(trace(x, y));
                prev = px(x, y);
                if (prev != 0 && prev != 1) last_border_x = x;
            }
        }
        return { found.rbegin(), found.rend() };   // newest first
    }

private:
    int8_t &px(int x, int y) { return img_[(size_t)y * w_ + x]; }

    Contour trace(int x0, int y0)
    {
        Contour out;
        // first neighbour: clockwise from west
        int dir = 4, first = -1;
        for (int k = 0; k < 8; ++k) {
            dir = (dir + 7) & 7;
            if (px(x0 + kDx[dir], y0 + kDy[dir]) != 0) { first = dir; break; }
        }
        if (first < 0) {                           // isolated pixel
            px(x0, y0) = kBorderExit;
            out.emplace_back(x0 - 1, y0 - 1);
            return out;
        }
        const int x1 = x0 + kDx[first], y1 = y0 + kDy[first];
        int cx = x0, cy = y0, came = first, last_step = first ^ 4;
        for (;;) {
            // next neighbour: counter-clockwise, starting just after the direction we came from
            int s = came;
            int nx, ny;
            do { ++s; nx = cx + kDx[s & 7]; ny = cy + kDy[s & 7]; } while (px(nx, ny) == 0);
            const int step = s & 7;
            // the search swept over "east" (code 0 = 8) iff it wrapped past 7 or started at/below 0 ... same test as
            // OpenCV's `(unsigned)(s - 1) < (unsigned)s_end`
            if ((unsigned)(step - 1) < (unsigned)came) px(cx, cy) = kBorderExit;
            else if (px(cx, cy) == 1) px(cx, cy) = kBorder;
            if (step != last_step) { out.emplace_back(cx - 1, cy - 1); last_step = step; }
            const bool closing = (nx == x0 && ny == y0 && cx == x1 && cy == y1);
            cx = nx; cy = ny;
            if (closing) break;
            came = (step + 4) & 7;
        }
        return out;
    }

    int w_, h_;
    std::vector<int8_t> img_;
};

void draw_segment(Image8 &img, Point a, Point b, const uint8_t bgr[3])
{
    // 8-connected Bresenham; after CHAIN_APPROX_SIMPLE every segment is horizontal, vertical or an exact diagonal
    int dx = std::abs(b.x - a.x), dy = std::abs(b.y - a.y);
    const int sx = a.x < b.x ? 1 : -1, sy = a.y < b.y ? 1 : -1;
    int err = dx - dy, x = a.x, y = a.y;
    for (;;) {
        if (x >= 0 && x < img.cols && y >= 0 && y < img.rows)
            for (int c = 0; c < 3; ++c) img.at(y, x, c) = bgr[c];
        if (x == b.x && y == b.y) break;
        const int e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x += sx; }
        if (e2 < dx) { err += dx; y += sy; }
    }
}

}  // namespace

SizeJson load_size_json(const std::string &json_path)
{
    std::ifstream f(json_path);
    if (!f.is_open()) throw std::runtime_error("Fail to Open JSON File: " + json_path);
    std::stringstream ss;
    ss << f.rdbuf();
    SizeJson out;
    for (const auto &kv : medseg::parse_size_json(ss.str())) {
        SizeEntry e;
        auto get = [&](const char *k) {
            auto it = kv.second.find(k);
            if (it == kv.second.end()) throw std::runtime_error(std::string("size JSON: missing key ") + k);
            return (int)it->second;
        };
        e.original_width = get("original_width");
        e.original_height = get("original_height");
        e.scaled_width = get("scaled_width");
        e.scaled_height = get("scaled_height");
        out[kv.first] = e;
    }
    return out;
}

std::vector<Contour> extract_contours(const Image8 &mask)
{
    if (mask.empty() || mask.channels != 1) return {};
    return BorderTracer(mask).run();
}

std::vector<Contour> map_contour_points(const std::vector<Contour> &contours, double scale_x, double scale_y)
{
    std::vector<Contour> mapped;
    mapped.reserve(contours.size());
    for (const Contour &c : contours) {
        Contour m;
        m.reserve(c.size());
        for (const Point &pt : c) m.emplace_back(static_cast<int>(pt.x * scale_x), static_cast<int>(pt.y * scale_y));
        mapped.push_back(std::move(m));
    }
    return mapped;
}

std::string polygon_json_text(const std::vector<Contour> &contours, const std::string &base_name, int original_width,
                              int original_height)
{
    return medseg::polygon_json_text(contours, base_name, original_width, original_height);
}

void generate_json(const std::vector<Contour> &contours, const std::string &json_path, const std::string &base_name,
                   int original_width, int original_height)
{
    std::ofstream f(json_path);
    if (!f.is_open()) throw std::runtime_error("Fail to Create JSON File: " + json_path);
    f << medseg::polygon_json_text(contours, base_name, original_width, original_height);
    f.flush();
}

Image8 draw_overlay(const Image8 &src, const std::vector<Contour> &contours)
{
    Image8 img(src.rows, src.cols, 3);
    for (int y = 0; y < src.rows; ++y) {                              // cv::imread(IMREAD_COLOR) of a gray PNG: B = G = R = gray
        const uint8_t *s = src.ptr(y);
        uint8_t *d = img.ptr(y);
        if (src.channels == 3) std::memcpy(d, s, (size_t)src.cols * 3);
        else
            for (int x = 0; x < src.cols; ++x) { const uint8_t g = s[x]; d[3 * x] = g; d[3 * x + 1] = g; d[3 * x + 2] = g; }
    }
    const uint8_t red_bgr[3] = { 0, 0, 255 };                         // cv::Scalar(0, 0, 255), src/mask2polygon.cpp:10
    for (const Contour &c : contours)                                 // drawContours(-1, thickness 1): closed polylines
        for (size_t k = 0; k < c.size(); ++k) draw_segment(img, c[k], c[(k + 1) % c.size()], red_bgr);
    return img;
}

void create_overlay_image(const std::vector<Contour> &contours, const std::string &original_png_path,
                          const std::string &overlay_path)
{
    const Image8 img = medseg::read_png(original_png_path, /*as_color=*/true);
    if (img.empty()) throw std::runtime_error("Fail to Read Original Image: " + original_png_path);
    if (!medseg::write_png(overlay_path, draw_overlay(img, contours), /*level0=*/false))
        throw std::runtime_error("Fail to Save Overlay PNG: " + overlay_path);
}

void write_polygon_outputs(const std::vector<Contour> &contours, const Image8 &normalized_tile, const std::string &output_dir,
                           const std::string &base_name, int original_width, int original_height,
                           std::ostream &console)
{
    try {
        console << "Processing Mask: " << base_name + ".png" << std::endl;
        console << "Original Size: " << original_width << "x" << original_height << std::endl;
        console << "Scaled Size: " << normalized_tile.cols << "x" << normalized_tile.rows << std::endl;
        if (contours.empty()) {
            console << "Warning: No Contours Detected" << std::endl;
            return;
        }
        console << "Extracted " << contours.size() << " Contours" << std::endl;
        const std::string overlay_path = output_dir + "/" + base_name + "_contour_overlay.png";
        // stored blocks, like the two PNGs the reference itself writes with IMWRITE_PNG_COMPRESSION 0 (src/preprocess.cpp:116,
        // src/process.cpp:233): the pixels are the contract, and deflating the 786 KB overlay was the longest artefact of the
        // device route (1.1 ms of the 3.9 ms per image); create_overlay_image above -- the reference's own
        // stand-alone mask2polygon entry point -- keeps cv::imwrite's default level 1
        if (!medseg::write_png(overlay_path, draw_overlay(normalized_tile, contours), /*level0=*/true))
            throw std::runtime_error("Fail to Save Overlay PNG: " + overlay_path);
        console << "Overlay Image Saved to: " << overlay_path << std::endl;
        const double scale_x = static_cast<double>(original_width) / normalized_tile.cols;
        const double scale_y = static_cast<double>(original_height) / normalized_tile.rows;
        const std::string output_json_path = output_dir + "/" + base_name + ".json";
        generate_json(map_contour_points(contours, scale_x, scale_y), output_json_path, base_name, original_width, original_height);
        console << "JSON Saved to: " << output_json_path << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "Processing Failure: " << e.what() << std::endl;
    }
}

void process_single_mask(const std::string &mask_path, const std::string &output_dir, const std::string &json_path,
                         const std::string &original_png, const std::string &base_name)
{
    try {
        std::cout << "Processing Mask: " << base_name + ".png" << std::endl;
        const SizeJson sizes = load_size_json(json_path);
        auto it = sizes.find(base_name + ".raw");
        if (it == sizes.end()) it = sizes.find(base_name + ".tif");
        if (it == sizes.end()) throw std::runtime_error("Cannot Find Size Info in JSON: " + base_name + ".raw/.tif");
        const SizeEntry &sz = it->second;
        std::cout << "Original Size: " << sz.original_width << "x" << sz.original_height << std::endl;
        std::cout << "Scaled Size: " << sz.scaled_width << "x" << sz.scaled_height << std::endl;

        const Image8 mask = medseg::read_png(mask_path, /*as_color=*/false);
        if (mask.empty()) throw std::runtime_error("Fail to Read Mask File: " + mask_path);
        if (mask.cols != sz.scaled_width || mask.rows != sz.scaled_height)
            throw std::runtime_error("Mask size mismatch: " + std::to_string(mask.cols) + "x" + std::to_string(mask.rows) +
                                     " (actual) vs " + std::to_string(sz.scaled_width) + "x" + std::to_string(sz.scaled_height) +
                                     " (JSON)");

        const std::vector<Contour> contours = extract_contours(mask);
        if (contours.empty()) {
            std::cout << "Warning: No Contours Detected" << std::endl;
            return;                                                   // neither overlay nor JSON (src/mask2polygon.cpp:183-186)
        }
        std::cout << "Extracted " << contours.size() << " Contours" << std::endl;

        if (!original_png.empty()) {                                  // overlay uses the UN-mapped 512-space points (:189-193)
            const std::string overlay_path = output_dir + "/" + base_name + "_contour_overlay.png";
            create_overlay_image(contours, original_png, overlay_path);
            std::cout << "Overlay Image Saved to: " << overlay_path << std::endl;
        } else {
            std::cout << "Warning: Original PNG not provided, skipping overlay generation" << std::endl;
        }

        const double scale_x = static_cast<double>(sz.original_width) / sz.scaled_width;
        const double scale_y = static_cast<double>(sz.original_height) / sz.scaled_height;
        const std::vector<Contour> mapped = map_contour_points(contours, scale_x, scale_y);
        const std::string output_json_path = output_dir + "/" + base_name + ".json";
        generate_json(mapped, output_json_path, base_name, sz.original_width, sz.original_height);
        std::cout << "JSON Saved to: " << output_json_path << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "Processing Failure: " << e.what() << std::endl;   // swallowed: the image still counts as done (:219-221)
    }
}

}  // namespace Mask2Polygon
