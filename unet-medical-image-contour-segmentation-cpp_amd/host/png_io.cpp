#include "png_io.h"

#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace medseg {
namespace {

void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &body)
{
    put32(out, (uint32_t)body.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at)));
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// IDAT payload = the zlib stream of the filtered rows (filter byte 0 + samples), built in bands of whole rows that are
// converted and deflated independently and then concatenated (the pigz construction: every band but the last ends in a sync
// flush, i.e. on a byte boundary with no final block; the Adler-32 of the whole and the chunk's CRC-32 are combined from the
// bands').  Any inflater reads it as one ordinary stream.  Bands run on their own threads unless the caller already is one of
// several parallel writers (directory mode writes its images in parallel: `threads` = 1 there).
struct Band {
    int r0 = 0, r1 = 0;
    std::vector<uint8_t> out;
    uLong adler = 1, crc = 0;
    size_t raw_len = 0;
    bool ok = false;
};

void deflate_band(const Image8 &img, Band &bd, int level, int strategy, bool last)
{
    const size_t stride = (size_t)img.cols * img.channels;
    std::vector<uint8_t> raw((stride + 1) * (bd.r1 - bd.r0));
    for (int y = bd.r0; y < bd.r1; ++y) {
        uint8_t *d = &raw[(stride + 1) * (y - bd.r0)];
        *d++ = 0;                                                    // filter type None
        if (img.channels == 1) {
            memcpy(d, img.ptr(y), stride);
        } else {                                                     // memory is B,G,R (OpenCV order); PNG wants R,G,B
            const uint8_t *s = img.ptr(y);
            for (int x = 0; x < img.cols; ++x) { d[3 * x] = s[3 * x + 2]; d[3 * x + 1] = s[3 * x + 1]; d[3 * x + 2] = s[3 * x]; }
        }
    }
    bd.raw_len = raw.size();
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) return;
    bd.out.resize(deflateBound(&zs, (uLong)raw.size()) + 16);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size();
    zs.next_out = bd.out.data(); zs.avail_out = (uInt)bd.out.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    bd.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0);
    bd.out.resize(zs.total_out);
    deflateEnd(&zs);
    bd.adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    bd.crc = crc32(crc32(0L, Z_NULL, 0), bd.out.data(), (uInt)bd.out.size());
}

// A few persistent helper threads for the bands of one image.  Round 3 started (and joined) up to fifteen std::threads per PNG:
// for the 786 KB overlay of a 512 x 512 tile that start-up was most of the 1.1 ms the file took.  The pool
// is created on first use and lives as long as the library; a caller that is itself one of several parallel writers (directory
// mode: one image per OpenMP thread) still asks for one band and never touches it.
class BandPool {
public:
    static BandPool &get()
    {
        static BandPool pool;
        return pool;
    }
    // run fn(0) .. fn(n - 1): the caller takes job 0, the helpers the rest; returns when all are done
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 1) { if (n == 1) fn(0); return; }
        Batch b;
        b.fn = &fn;
        b.left = n - 1;
        {
            std::lock_guard<std::mutex> lk(m_);
            for (int i = 1; i < n; ++i) q_.push_back({ &b, i });
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {                                     // help with whatever is still queued (also other callers' bands), then wait
            if (b.left == 0) break;
            if (!q_.empty()) {
                Job j = q_.front();
                q_.pop_front();
                lk.unlock();
                (*j.b->fn)(j.i);
                lk.lock();
                if (--j.b->left == 0) done_.notify_all();
                continue;
            }
            done_.wait(lk);
        }
    }

private:
    struct Batch { const std::function<void(int)> *fn = nullptr; int left = 0; };
    struct Job { Batch *b; int i; };
    BandPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        const int n = (int)std::max(2u, std::min(12u, hw ? hw / 4 : 4u));
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~BandPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
            if (stop_) return;
            Job j = q_.front();
            q_.pop_front();
            lk.unlock();
            (*j.b->fn)(j.i);
            lk.lock();
            if (--j.b->left == 0) done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<Job> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

}  // namespace

static thread_local int t_png_threads = 16;
void set_png_threads(int n) { t_png_threads = n < 1 ? 1 : n; }

bool write_png(const std::string &path, const Image8 &img, bool level0)
{
    if (img.empty() || (img.channels != 1 && img.channels != 3)) return false;
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    std::vector<uint8_t> ihdr;
    put32(ihdr, img.cols); put32(ihdr, img.rows);
    ihdr.push_back(8); ihdr.push_back(img.channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    // level 0 = stored blocks (the reference's IMWRITE_PNG_COMPRESSION 0 files); otherwise OpenCV's imwrite defaults:
    // level 1 with the run-length strategy -- fast, and the compressed bytes are not part of any contract (the pixels are)
    const int level = level0 ? 0 : 1, strategy = level0 ? Z_DEFAULT_STRATEGY : Z_RLE;
    const size_t total = ((size_t)img.cols * img.channels + 1) * img.rows;
    // bands of >= 48 KB (level 1) / >= 96 KB (stored: only the copy and the two checksums are left to share)
    int nb = (int)std::min<size_t>((size_t)t_png_threads, std::max<size_t>(1, total / ((level0 ? 96u : 48u) << 10)));
    nb = std::max(1, std::min(nb, img.rows));
    std::vector<Band> bands(nb);
    for (int b = 0; b < nb; ++b) {
        bands[b].r0 = (int)((long long)img.rows * b / nb);
        bands[b].r1 = (int)((long long)img.rows * (b + 1) / nb);
    }
    BandPool::get().run(nb, [&](int b) { deflate_band(img, bands[b], level, strategy, b == nb - 1); });
    size_t zlen = 2 + 4;
    for (const Band &bd : bands) { if (!bd.ok) return false; zlen += bd.out.size(); }
    // the IDAT chunk, written in place: length, type, zlib header, bands, Adler-32, CRC-32 (combined from the pieces)
    out.reserve(out.size() + zlen + 12 + 12);
    put32(out, (uint32_t)zlen);
    const uint8_t head[6] = { 'I', 'D', 'A', 'T', 0x78, 0x01 };      // deflate, 32 KB window, no dictionary, check bits
    out.insert(out.end(), head, head + 6);
    uLong crc = crc32(crc32(0L, Z_NULL, 0), head, 6), ad = bands[0].adler;
    for (int b = 0; b < nb; ++b) {
        out.insert(out.end(), bands[b].out.begin(), bands[b].out.end());
        crc = crc32_combine(crc, bands[b].crc, (z_off_t)bands[b].out.size());
        if (b) ad = adler32_combine(ad, bands[b].adler, (z_off_t)bands[b].raw_len);
    }
    const uint8_t tail[4] = { (uint8_t)(ad >> 24), (uint8_t)(ad >> 16), (uint8_t)(ad >> 8), (uint8_t)ad };
    out.insert(out.end(), tail, tail + 4);
    crc = crc32(crc, tail, 4);
    put32(out, (uint32_t)crc);
    chunk(out, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

Image8 read_png(const std::string &path, bool as_color)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return {};
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) return {};
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t p = 8;
    auto rd32 = [&](size_t o) { return (uint32_t)buf[o] << 24 | (uint32_t)buf[o + 1] << 16 | (uint32_t)buf[o + 2] << 8 | buf[o + 3]; };
    while (p + 12 <= buf.size()) {
        const uint32_t len = rd32(p);
        if (p + 12 + len > buf.size()) return {};
        const char *type = reinterpret_cast<const char *>(&buf[p + 4]);
        const uint8_t *body = &buf[p + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = rd32(p + 8); h = rd32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + len;
    }
    if (!w || !h || interlace || (depth != 8 && !(depth == 16 && ctype == 0))) return {};
    int spp;                                                       // samples per pixel
    switch (ctype) {
    case 0: spp = 1; break;
    case 2: spp = 3; break;
    case 3: spp = 1; break;
    case 4: spp = 2; break;
    case 6: spp = 4; break;
    default: return {};
    }
    const int bpp = spp * depth / 8;
    const size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return {};
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const int ft = raw[(stride + 1) * y];
        const uint8_t *s = &raw[(stride + 1) * y + 1];
        uint8_t *d = &img[stride * y];
        const uint8_t *up = y ? d - stride : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? d[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int v = s[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return {};
            }
            d[i] = (uint8_t)v;
        }
    }
    Image8 out((int)h, (int)w, as_color ? 3 : 1);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *px = &img[stride * y + (size_t)x * bpp];
            int r, g, b;
            if (ctype == 0) { r = g = b = px[0]; }                   // 16-bit gray: high byte (cv::imread without ANYDEPTH)
            else if (ctype == 4) { r = g = b = px[0]; }
            else if (ctype == 3) { const size_t k = (size_t)px[0] * 3; if (k + 2 >= plte.size() + 0 && plte.size() < k + 3) return {}; r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; }
            else { r = px[0]; g = px[1]; b = px[2]; }
            if (as_color) { out.at(y, x, 0) = (uint8_t)b; out.at(y, x, 1) = (uint8_t)g; out.at(y, x, 2) = (uint8_t)r; }
            else out.at(y, x) = (r == g && g == b) ? (uint8_t)r : (uint8_t)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
        }
    return out;
}

}  // namespace medseg
