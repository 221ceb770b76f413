// image.h -- minimal dense image / point types standing in for cv::Mat (CV_8UC1 / CV_8UC3) and cv::Point in the
// public signatures of the reference (include/mask2polygon.h:9-17, src/postprocess.cpp:47): this build has no OpenCV.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace medseg {

struct Point {
    int x = 0, y = 0;
    Point() = default;
    Point(int x_, int y_) : x(x_), y(y_) {}
    bool operator==(const Point &o) const { return x == o.x && y == o.y; }
};

using Contour = std::vector<Point>;

// Row-major, interleaved channels, 8 bits per sample.
struct Image8 {
    int rows = 0, cols = 0, channels = 1;
    std::vector<uint8_t> data;

    Image8() = default;
    Image8(int r, int c, int ch = 1, uint8_t fill = 0) : rows(r), cols(c), channels(ch), data((size_t)r * c * ch, fill) {}
    bool empty() const { return data.empty(); }
    uint8_t *ptr(int y = 0) { return data.data() + (size_t)y * cols * channels; }
    const uint8_t *ptr(int y = 0) const { return data.data() + (size_t)y * cols * channels; }
    uint8_t &at(int y, int x, int c = 0) { return data[((size_t)y * cols + x) * channels + c]; }
    uint8_t at(int y, int x, int c = 0) const { return data[((size_t)y * cols + x) * channels + c]; }
};

}  // namespace medseg
