// Reference: include/mask2polygon.h:7-23, src/mask2polygon.cpp.
#pragma once
#include <map>
#include <iostream>
#include <string>
#include <vector>

#include "image.h"

namespace Mask2Polygon {

// One entry of the size file written by Preprocess::preprocess_raw (src/preprocess.cpp:126-134).
struct SizeEntry {
    int original_width = 0, original_height = 0, scaled_width = 0, scaled_height = 0;
};
using SizeJson = std::map<std::string, SizeEntry>;       // stands in for nlohmann::json in load_size_json's signature

SizeJson load_size_json(const std::string &json_path);
std::vector<medseg::Contour> extract_contours(const medseg::Image8 &mask);
std::vector<medseg::Contour> map_contour_points(const std::vector<medseg::Contour> &contours, double scale_x, double scale_y);
void generate_json(const std::vector<medseg::Contour> &contours, const std::string &json_path, const std::string &base_name,
                   int original_width, int original_height);
void create_overlay_image(const std::vector<medseg::Contour> &contours, const std::string &original_png_path,
                          const std::string &overlay_path);
void process_single_mask(const std::string &mask_path, const std::string &output_dir, const std::string &json_path,
                         const std::string &original_png, const std::string &base_name);

// In-memory halves of the functions above (used by the device-first pipeline, which already holds the tile and the contours):
// the overlay picture (gray tile replicated to B,G,R, closed red polylines of thickness 1) and the steps 6-8 of
// process_single_mask (src/mask2polygon.cpp:188-207) for contours that were extracted elsewhere.
medseg::Image8 draw_overlay(const medseg::Image8 &gray_or_bgr, const std::vector<medseg::Contour> &contours);
void write_polygon_outputs(const std::vector<medseg::Contour> &contours, const medseg::Image8 &normalized_tile,
                           const std::string &output_dir, const std::string &base_name, int original_width,
                           int original_height, std::ostream &console = std::cout);

// The document generate_json writes, as a string (4-space indent, sorted keys, trailing newline).
std::string polygon_json_text(const std::vector<medseg::Contour> &contours, const std::string &base_name, int original_width,
                              int original_height);

}  // namespace Mask2Polygon
