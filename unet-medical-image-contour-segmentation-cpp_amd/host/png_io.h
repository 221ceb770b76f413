// png_io.h -- the two things the pipeline needs from cv::imwrite / cv::imread (src/preprocess.cpp:122,
// src/process.cpp:217,236-237, src/mask2polygon.cpp:117,126,166): write 8-bit gray or RGB PNGs (zlib level 0 = stored
// blocks when level0 is set), read 8-bit PNGs back as gray or as 3-channel colour.  Compressed bytes are not part of
// the contract; decoded pixels are.
#pragma once
#include <string>

#include "../../include/medseg/image.h"

namespace medseg {

bool write_png(const std::string &path, const Image8 &img, bool level0);
// A deflated PNG is compressed in up to n bands of rows on n threads (default 16).  Per calling thread: directory mode, which
// already writes its images on several host threads, sets 1 on those.
void set_png_threads(int n);
// as_color = false: cv::IMREAD_GRAYSCALE (colour inputs are reduced with OpenCV's integer BT.601 weights);
// as_color = true : default cv::imread -> 3 channels in B,G,R order (gray replicated).  Empty image on failure.
Image8 read_png(const std::string &path, bool as_color);

}  // namespace medseg
