// preprocess.cpp -- RAW16 -> normalised 512x512 8-bit tile (+ PNG + size JSON).  Reference: src/preprocess.cpp.
// Own code on the same arithmetic contract: exact u16 min/max; top-left aligned bilinear taps;
// interpolation and quantisation in double in the reference's operand order; `mx = mn + 1` evaluated in uint16_t.
#include "../../include/medseg/preprocess.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "json_io.h"
#include "png_io.h"

namespace fs = std::filesystem;

namespace {

// Read-only mapping of the RAW file.  Unlike the reference (src/preprocess.cpp:40: mmap result unchecked, size in int)
// the size is computed in size_t and both the file length and MAP_FAILED are checked.
class MappedFile {
public:
    MappedFile(const std::string &path, size_t size) : size_(size)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("open failed");
        struct stat st {};
        if (::fstat(fd_, &st) != 0 || (size_t)st.st_size < size) {
            ::close(fd_);
            throw std::runtime_error("RAW file is smaller than width*height*2 bytes");
        }
        // MAP_POPULATE: the pages are mapped in one batch here.  Touched one by one they are 1 536 minor faults per 6 MB file, and the
        // faults of a process serialise on its mmap lock -- sixteen reader threads of directory mode spent 24 ms per 16 files in them
        data_ = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd_, 0);
        if (data_ == MAP_FAILED) {
            ::close(fd_);
            throw std::runtime_error("mmap failed");
        }
    }
    ~MappedFile()
    {
        if (data_ && data_ != MAP_FAILED) ::munmap(data_, size_);
        if (fd_ >= 0) ::close(fd_);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    const uint16_t *data() const { return static_cast<const uint16_t *>(data_); }

private:
    size_t size_;
    void *data_ = nullptr;
    int fd_ = -1;
};

}  // namespace

namespace Preprocess {

medseg::Image8 resample_normalize(const uint16_t *src, int w, int h, int outW, int outH)
{
    const size_t n = (size_t)w * h;
    uint16_t mn = 65535, mx = 0;
#pragma omp parallel for reduction(min : mn) reduction(max : mx) schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        mn = std::min(mn, src[i]);
        mx = std::max(mx, src[i]);
    }
    if (mn == mx) mx = (uint16_t)(mn + 1);                       // src/preprocess.cpp:92 (wraps at 65535, kept)
    const double scale8 = 255.0 / ((int)mx - (int)mn);
    const double stepX = (double)w / outW, stepY = (double)h / outH;
    medseg::Image8 dst(outH, outW, 1);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < outH; ++y) {
        const double fy = y * stepY;
        const int iy = (int)fy, iy1 = std::min(iy + 1, h - 1);
        const double dy = fy - iy;
        const uint16_t *r0 = src + (size_t)iy * w, *r1 = src + (size_t)iy1 * w;
        uint8_t *o = dst.ptr(y);
        for (int x = 0; x < outW; ++x) {
            const double fx = x * stepX;
            const int ix = (int)fx, ix1 = std::min(ix + 1, w - 1);
            const double dx = fx - ix;
            const double v = (1 - dx) * (1 - dy) * r0[ix] + dx * (1 - dy) * r0[ix1] + (1 - dx) * dy * r1[ix] + dx * dy * r1[ix1];
            o[x] = (uint8_t)(int)((v - mn) * scale8 + 0.5);
        }
    }
    return dst;
}

RawView::RawView(const std::string &raw_path, int w, int h)
{
    if (w <= 0 || h <= 0) throw std::runtime_error("width and height must be positive");
    auto file = std::make_shared<MappedFile>(raw_path, (size_t)w * h * 2);
    map_ = std::shared_ptr<void>(file, const_cast<uint16_t *>(file->data()));    // aliasing: the mapping lives as long as the view
    samples_ = (size_t)w * h;
}

std::vector<uint16_t> read_raw16(const std::string &raw_path, int w, int h)
{
    if (w <= 0 || h <= 0) throw std::runtime_error("width and height must be positive");
    MappedFile file(raw_path, (size_t)w * h * 2);
    return std::vector<uint16_t>(file.data(), file.data() + (size_t)w * h);
}

bool write_preprocess_outputs(const medseg::Image8 &tile, const std::string &raw_path, const std::string &png_path,
                              const std::string &json_path, int w, int h)
{
    try {
        const fs::path parent = fs::path(png_path).parent_path();
        if (!parent.empty()) fs::create_directories(parent);
        if (!medseg::write_png(png_path, tile, /*level0=*/true)) throw std::runtime_error("imwrite failed");
        std::ofstream jf(json_path);
        jf << medseg::size_json_text(fs::path(raw_path).filename().string(), w, h, tile.cols, tile.rows);
        jf.flush();
        return jf.good();
    } catch (const std::exception &e) {
        std::cerr << "preprocess_raw error: " << e.what() << '\n';
        return false;
    }
}

bool preprocess_raw(const std::string &raw_path, const std::string &png_path, const std::string &json_path, int w, int h)
{
    try {
        if (w <= 0 || h <= 0) throw std::runtime_error("width and height must be positive");
        const int outW = 512, outH = 512;                        // src/preprocess.cpp:81
        MappedFile file(raw_path, (size_t)w * h * 2);
        const medseg::Image8 dst8 = resample_normalize(file.data(), w, h, outW, outH);
        return write_preprocess_outputs(dst8, raw_path, png_path, json_path, w, h);
    } catch (const std::exception &e) {
        std::cerr << "preprocess_raw error: " << e.what() << '\n';
        return false;
    }
}

}  // namespace Preprocess
