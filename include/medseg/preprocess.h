// Reference: include/preprocess.h:20-23, src/preprocess.cpp:76-141.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "image.h"

namespace Preprocess {

// mmap the headerless little-endian u16 RAW (w*h*2 bytes), min/max, top-left-aligned bilinear resample to 512x512 in
// double, quantise to u8, write the PNG (level 0) and the one-line size JSON.  false on failure.
bool preprocess_raw(const std::string &raw_path, const std::string &png_path, const std::string &json_path, int w, int h);

// Pieces of the above, exposed so the device-first pipeline can reuse them: the checked mmap read of the RAW file, and the
// PNG (level 0) + one-line size JSON writers (src/preprocess.cpp:121-134).
std::vector<uint16_t> read_raw16(const std::string &raw_path, int w, int h);
bool write_preprocess_outputs(const medseg::Image8 &tile, const std::string &raw_path, const std::string &png_path,
                              const std::string &json_path, int w, int h);

// The arithmetic of the above on memory (no files): src/preprocess.cpp:81-118.
medseg::Image8 resample_normalize(const uint16_t *src, int w, int h, int outW = 512, int outH = 512);

}  // namespace Preprocess
