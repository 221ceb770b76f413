// Reference: include/preprocess.h:20-23, src/preprocess.cpp:76-141.
#pragma once
#include <cstdint>
#include <memory>
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

// Read-only view of the RAW file -- the checked mmap itself (file length and MAP_FAILED verified, sizes in size_t; the
// reference's MMapFile, src/preprocess.cpp:28-61, checks neither).  The device-first pipeline copies from it straight into
// pinned staging memory: one copy between the page cache and the GPU.  Throws std::runtime_error.
class RawView {
public:
    RawView(const std::string &raw_path, int w, int h);
    const uint16_t *data() const { return static_cast<const uint16_t *>(map_.get()); }
    size_t samples() const { return samples_; }

private:
    std::shared_ptr<void> map_;
    size_t samples_ = 0;
};

// The arithmetic of the above on memory (no files): src/preprocess.cpp:81-118.
medseg::Image8 resample_normalize(const uint16_t *src, int w, int h, int outW = 512, int outH = 512);

}  // namespace Preprocess
