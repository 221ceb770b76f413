// json_io.h -- byte-exact emitter for the two documents the reference writes through nlohmann::json 3.12
// (include/nlohmann/json.hpp in the reference; pinned in tests against oracle/_ref/json_probe, which is built from that
// header) and a small reader for the size file.  nlohmann specifics reproduced: std::map key order (byte-wise sorted),
// dump() without indent = no whitespace at all, dump(4) = every array element and object member on its own line,
// "{}" / "[]" for empty containers, string escaping of dump() with ensure_ascii = false.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/medseg/image.h"

namespace medseg {

std::string json_escape(const std::string &s);

// {"<raw filename>":{"original_height":h,"original_width":w,"scaled_height":sh,"scaled_width":sw}}\n  (src/preprocess.cpp:126-134)
std::string size_json_text(const std::string &raw_filename, int w, int h, int scaled_w, int scaled_h);

// src/mask2polygon.cpp:68-109 with std::setw(4)
std::string polygon_json_text(const std::vector<Contour> &contours, const std::string &base_name, int original_width,
                              int original_height);

// Parses an object of objects of integers (the size file).  Throws std::runtime_error on malformed input.
std::map<std::string, std::map<std::string, long long>> parse_size_json(const std::string &text);

}  // namespace medseg
